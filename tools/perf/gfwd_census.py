"""Every kernel of ONE generator forward (train-mode statistics, no autograd tape) in launch order, with the aten op / user
range that issued it: the list the "stock launches out of the generator forward" work is driven by."""
import collections, functools, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L
from layout2img_amd import generator as G, ops
from layout2img_amd.synthetic import make_batch
from torch.profiler import profile, ProfilerActivity, record_function


def wrap(obj, name, tag):
    f = getattr(obj, name)

    @functools.wraps(f)
    def g(*a, **k):
        with record_function("L2I:" + tag):
            return f(*a, **k)
    setattr(obj, name, g)


for cls in (G.BoxMultiHeadedAttention, G.MaskRegressNetv2, G.PSPModule, G.ConvMaskHead, G.ResBlock):
    wrap(cls, "forward", cls.__name__)
for n in ("_stage_mask", "_project_isla", "_latent"):
    wrap(G.ResnetGenerator128_context, n, n)

dev = torch.device("cuda:0")
torch.manual_seed(1234)
layout = sys.argv[1] if len(sys.argv) > 1 else "coco"
if layout == "vg":
    netG = L.context_aware_generator(num_classes=179).finalize(dev, torch.bfloat16)
else:
    netG = L.ResnetGenerator128_context(num_classes=184).finalize(dev, torch.bfloat16)
netG.train()
real, label, bbox, z, z_im = make_batch(32, 128, layout, seed=1234, device=dev)
def run():
    with torch.no_grad(), ops.POOL.step(dev):
        return netG(z, bbox, z_im=z_im, y=label)
for _ in range(3):
    run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run()
    torch.cuda.synchronize()


def owner(e):
    p = e.cpu_parent
    names = []
    while p is not None:
        if p.name.startswith("L2I:"):
            names.append(p.name[4:])
        p = p.cpu_parent
    return "/".join(reversed(names)) or "(top)"


rows = []
for e in prof.events():
    if not e.kernels or any(c.kernels for c in e.cpu_children):
        continue
    for k in e.kernels:
        rows.append((e.time_range.start, owner(e), e.name, k.name, k.duration))
rows.sort()
tot = collections.Counter()
n_l2i = n_other = 0
t_l2i = t_other = 0.0
for _, o, op, kn, d in rows:
    stock = op.startswith("aten::") or "Memcpy" in op or "Memset" in op or "Cijk" in kn or "rocclr" in kn
    if stock:
        n_other += 1; t_other += d
    else:
        n_l2i += 1; t_l2i += d
    print(f"{d:8.1f} us  {'*' if stock else ' '} {o:45s} {op[:28]:28s} {kn[:70]}")
print(f"own launches {n_l2i} = {t_l2i / 1e3:.3f} ms; stock launches {n_other} = {t_other / 1e3:.3f} ms")
