# census of torch-level ops in one training iteration (which aten ops launch the small kernels)
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L
from layout2img_amd.synthetic import make_batch
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
torch.manual_seed(1234)
netG = L.ResnetGenerator128_context(num_classes=184).finalize(dev, torch.bfloat16)
netD = L.CombineDiscriminator128_app(num_classes=184).finalize(dev, torch.bfloat16)
tr = L.GanTrainer(netG, netD)
real, label, bbox, z, z_im = make_batch(32, 128, "coco", seed=1234, device=dev)
for _ in range(3): tr.step(real, label, bbox, z, None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    tr.step(real, label, bbox, z, None)
torch.cuda.synchronize()
ev = prof.key_averages(group_by_stack_n=4)
rows = [(e.count, e.key, e.stack) for e in ev if e.key in ("aten::copy_", "aten::add_", "aten::add", "aten::zeros", "aten::fill_", "aten::zero_", "aten::mul", "aten::clone", "aten::contiguous", "aten::zeros_like", "aten::cat", "aten::sum")]
rows.sort(key=lambda r: -r[0])
for c, k, st in rows[:45]:
    loc = [s for s in st if "/root/repo" in s or "layout2img" in s]
    print(f"x{c:4d} {k:18s} {loc[0][-90:] if loc else (st[0][-90:] if st else '')}")
