"""Stock (aten) launches of one eager training iteration with their tensor shapes and GPU time, largest first: which fills, adds and
copies are still worth removing."""
import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("L2I_OVERLAP", "0")
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(real, label, bbox, z, None)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = getattr(e, "self_cuda_time_total", 0.0)
    if e.key.startswith("aten::") and t > 0:
        rows.append((t, e.count, e.key, str(e.input_shapes)[:100]))
print(f"aten launches: {sum(r[1] for r in rows)}, {sum(r[0] for r in rows) / 1e3:.3f} ms")
for t, c, n, sh in sorted(rows, reverse=True)[:45]:
    print(f"{t:8.1f} us  x{c:3d}  {n:22s} {sh}")
