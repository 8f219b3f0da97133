# census of every kernel-launching aten op of one training iteration, grouped by the repo call site that issued it
import sys, os, collections, torch
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.step(real, label, bbox, z, None)
    torch.cuda.synchronize()
# an op "launches" if a kernel event is correlated to it; attribute each kernel to the innermost aten op, then to the
# nearest repo frame of that op's stack
cnt = collections.Counter(); dur = collections.Counter()
for e in prof.events():
    if not e.kernels or not e.name.startswith("aten::"):
        continue
    if any(c.kernels for c in e.cpu_children if c.name.startswith("aten::")):
        continue   # a child aten op owns the kernels
    loc = [s for s in (e.stack or []) if "layout2img_amd" in s or "bench.py" in s]
    site = loc[0].split("layout2img_amd/")[-1][:70] if loc else "autograd-engine"
    shp = str([tuple(x) for x in (e.input_shapes or []) if x])[:60]
    k = (e.name, site + " " + shp)
    cnt[k] += len(e.kernels)
    dur[k] += sum(x.duration for x in e.kernels)
tot = sum(cnt.values())
print(f"{tot} kernel launches from aten ops, {sum(dur.values())/1e3:.3f} ms")
for k, c in sorted(cnt.items(), key=lambda kv: -dur[kv[0]])[:140]:
    print(f"x{c:4d} {dur[k]/1e3:7.3f} ms  {k[0]:28s} {k[1]}")
