"""debug: states of the generator's cross-block GradJoins after one backward, gradient difference against JOIN_HEADS = False"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L
from layout2img_amd import generator as G, ops
from layout2img_amd.synthetic import make_batch
DEV = "cuda:0"
real, label, bbox, z, z_im = make_batch(3, 128, "coco", seed=5, device=torch.device(DEV))
grads = {}
made = []
J0 = ops.GradJoin
class J(J0):
    def __init__(self):
        super().__init__(); made.append(self); self.log = []
    def give(self, dx):
        r = super().give(dx); self.log.append(("give", type(dx).__name__, r is None)); return r
    def take(self):
        r = super().take(); self.log.append(("take", r is not None)); return r
    def leftover(self):
        r = super().leftover(); self.log.append(("leftover", r is not None)); return r
ops.GradJoin = J
for join in (True, False):
    G.JOIN_HEADS = join
    torch.manual_seed(0)
    g = L.ResnetGenerator128_context(num_classes=184).finalize(DEV, torch.float32).train()
    for m in g.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
    g.zero_grad()
    made.clear()
    img = g(z, bbox, z_im, label)
    (img * real).sum().backward()
    g.arena.flush_grads()
    torch.cuda.synchronize()
    grads[join] = g.flat.grad.clone()
    print("JOIN_HEADS", join, "joins:", len(made))
    for j in made:
        print("   ", j.state, j.log)
a, b = grads[True], grads[False]
print("rel", float((a - b).norm() / b.norm()))
named = dict(g.named_parameters())
off = 0
for n, p_ in named.items():
    if p_.grad is not None:
        pass
for n, p_ in g.named_parameters():
    o = g.flat.offset_of(p_) if hasattr(g.flat, "offset_of") else None
    if o is None: break
    da = a[o:o + p_.numel()]; db = b[o:o + p_.numel()]
    r = float((da - db).norm() / (db.norm() + 1e-20))
    if r > 1e-3: print(f"  {n}: rel {r:.3e}  |b| {float(db.norm()):.3e}")
