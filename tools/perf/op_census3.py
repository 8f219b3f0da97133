# which parts of the model issue the small ATen / runtime launches: user ranges (record_function) wrapped around the host-side
# pieces of one training iteration, every kernel-launching aten op attributed to its innermost enclosing range
import sys, os, collections, functools, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L
from layout2img_amd import generator as G, discriminator as D, ops, arena, trainer
from layout2img_amd.synthetic import make_batch
from torch.profiler import profile, ProfilerActivity, record_function

def wrap(obj, name, tag=None):
    f = getattr(obj, name)
    @functools.wraps(f)
    def g(*a, **k):
        with record_function("L2I:" + (tag or name)):
            return f(*a, **k)
    setattr(obj, name, g)

for cls, names in ((G.BoxMultiHeadedAttention, ["forward"]), (G.MaskRegressNetv2, ["forward"]), (G.PSPModule, ["forward"]),
                   (G.ConvMaskHead, ["forward"]), (G.ResBlock, ["forward"]), (D.ResBlock, ["forward"]), (D.OptimizedBlock, ["forward"]),
                   (D.CombineDiscriminator128_app, ["prepare_layout", "_prepare"]), (arena.WeightArena, ["prepare", "flush_grads"])):
    for n in names:
        wrap(cls, n, cls.__name__ + "." + n)
for n in ("proj_head", "emb_dot", "gram_head", "roi_align", "hinge", "hinge_sum", "l1_loss", "stage_mask", "box_attention", "grouped_linear", "adam_step"):
    if hasattr(ops, n): wrap(ops, n, "ops." + n)
wrap(G.ResnetGenerator128_context, "_stage_mask", "G._stage_mask")
wrap(G.ResnetGenerator128_context, "_project_isla", "G._project_isla")
wrap(G.ResnetGenerator128_context, "_latent", "G._latent")

dev = torch.device('cuda:0')
torch.manual_seed(1234)
netG = L.ResnetGenerator128_context(num_classes=184).finalize(dev, torch.bfloat16)
netD = L.CombineDiscriminator128_app(num_classes=184).finalize(dev, torch.bfloat16)
tr = L.GanTrainer(netG, netD)
real, label, bbox, z, z_im = make_batch(32, 128, "coco", seed=1234, device=dev)
for _ in range(3): tr.step(real, label, bbox, z, None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(real, label, bbox, z, None)
    torch.cuda.synchronize()
cnt = collections.Counter(); dur = collections.Counter()
def owner(e):
    p = e.cpu_parent
    while p is not None:
        if p.name.startswith("L2I:"): return p.name[4:]
        if p.name.startswith(("autograd::engine", "torch::autograd")) or "Backward" in p.name: return "backward:" + p.name.split(":")[-1][:40]
        p = p.cpu_parent
    return "(top level: trainer / generator / discriminator forward)"
for e in prof.events():
    if not e.kernels: continue
    if any(c.kernels for c in e.cpu_children): continue
    if not (e.name.startswith("aten::") or "Memcpy" in e.name or "Memset" in e.name or e.name.startswith("hip")): continue
    k = (owner(e), e.name)
    cnt[k] += len(e.kernels); dur[k] += sum(x.duration for x in e.kernels)
by = collections.Counter(); byn = collections.Counter()
for (o, n), c in cnt.items(): by[o] += dur[(o, n)]; byn[o] += c
print("small launches by owner (ms, launches):")
for o, d in by.most_common(40): print(f"  {d/1e3:7.3f} ms  x{byn[o]:4d}  {o}")
print("detail:")
for k, c in sorted(cnt.items(), key=lambda kv: -dur[kv[0]])[:60]: print(f"  x{c:4d} {dur[k]/1e3:7.3f} ms  {k[1]:30s} {k[0]}")
