"""Which convolutions still cast their incoming gradient (or input stream) with a separate launch: for every ops.cast_op call of one
training iteration, the caller (forward / backward of FusedConvFn) and the layer (Ci -> Co, kernel, up / pool) it serves."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["L2I_OVERLAP"] = "0"
import layout2img_amd as L
from layout2img_amd import ops
from layout2img_amd.synthetic import make_batch
dev = torch.device("cuda:0")
torch.manual_seed(1234)
netG = L.ResnetGenerator128_context(num_classes=184).finalize(dev, torch.bfloat16)
netD = L.CombineDiscriminator128_app(num_classes=184).finalize(dev, torch.bfloat16)
names = {}
for net, tag in ((netG, "G"), (netD, "D")):
    for n, m in net.named_modules():
        names[id(m)] = f"{tag}.{n}"
tr = L.GanTrainer(netG, netD)
batch = make_batch(32, 128, "coco", seed=1234, device=dev)[:4]
tr.step(*batch)
cnt = collections.Counter()
orig = ops.cast_op
def cast_op(x, *a, **k):
    f = sys._getframe(1)
    where, loc = f.f_code.co_name, f.f_locals
    h = loc.get("h") or loc.get("holder")
    desc = ""
    if h is not None:
        hh = getattr(h, "h", h)
        desc = f"{names.get(id(hh), '?')} {hh.ci}->{hh.co} k{hh.kh}"
    ctx = loc.get("ctx")
    if ctx is not None and where == "backward":
        desc += f" up{int(ctx.up2)} pool{int(ctx.pool2)} has_res{int(ctx.has_res)}"
    cnt[(where, tuple(x.shape), desc)] += 1
    return orig(x, *a, **k)
ops.cast_op = cast_op
tr.step(*batch)
torch.cuda.synchronize()
tot = 0
for (where, shape, desc), n in sorted(cnt.items(), key=lambda kv: -kv[1] * torch.Size(kv[0][1]).numel()):
    tot += n
    print(f"{n:3d} x {where:9s} {str(shape):24s} {desc}")
print("cast launches per iteration:", tot)
