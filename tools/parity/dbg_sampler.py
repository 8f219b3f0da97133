import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import layout2img_amd as L
from layout2img_amd import _lib, arena as A
from layout2img_amd.sampling import GraphSampler
from layout2img_amd.synthetic import make_batch, make_layouts
DEV = "cuda:0"
torch.manual_seed(3)
g = L.ResnetGenerator128_context(num_classes=184).finalize(DEV, torch.float32)
d = L.CombineDiscriminator128_app(num_classes=184).finalize(DEV, torch.float32)
real, label, bbox, z, z_im = make_batch(4, 128, "coco", seed=9, device=DEV)
g.train()
with torch.no_grad():
    for _ in range(3):
        g(z, bbox, z_im, label)
lab1, box1 = make_layouts(1, "coco", seed=21, device=DEV)
calls = []
orig = _lib.call
def spy(name, *a):
    calls.append(name)
    return orig(name, *a)
A._lib.call = spy
g.eval()
s = GraphSampler(g, thres=2.0)
def check(tag):
    n0 = calls.count("l2i_weights_prepare")
    img, zs, zi = s(lab1, box1, return_latents=True)
    n1 = calls.count("l2i_weights_prepare")
    img, zs, zi = img.clone(), zs.clone(), zi.clone()
    with torch.no_grad():
        ref = g(zs, box1, z_im=zi, y=lab1)
        ref2 = g(zs, box1, z_im=zi, y=lab1)
    n2 = calls.count("l2i_weights_prepare")
    print(tag, "graph-vs-eager", float((img - ref).abs().max()), "eager-vs-eager", float((ref - ref2).abs().max()), "img absmean", float(img.abs().mean()), float(ref.abs().mean()),
          "prepare calls in sampler", n1 - n0, "in eager", n2 - n1, "stamp", g.arena._eval_stamp, flush=True)
check("first"); check("second")
with torch.no_grad():
    g.fc.weight_orig.mul_(1.25)
check("after mul_")
sd = {k: v.detach().cpu().clone() for k, v in g.state_dict().items()}
g.load_state_dict(sd)
check("after identical load_state_dict")
g.train()
tr = L.GanTrainer(g, d)
tr.step(real, label, bbox, z, z_im)
g.eval()
check("after a training step")
check("again")
sd = {k: v.detach().cpu().clone() for k, v in g.state_dict().items()}
g.load_state_dict(sd)
check("after identical load_state_dict 2")
torch.cuda.synchronize()
check("again 2")
