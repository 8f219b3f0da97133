"""Eight identical runs of two generator forwards + backward in one process: the matrix of pairwise gradient distances and, for the most distant
pair, the parameter tensors that differ (round 6: found the duplicate-destination multi-tensor add in flush_grads, DESIGN A.r06.2).
    python tools/parity/two_forwards_states.py [batch]"""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import layout2img_amd as L
from layout2img_amd.synthetic import make_batch
DEV = "cuda:0"
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = L.ResnetGenerator64_context(num_classes=184).finalize(DEV, torch.float32)
for m in g.modules():
    if hasattr(m, "dropout_p"):
        m.dropout_p = 0.0
g.train()
_, label, bbox, z1, z_im = make_batch(B, 64, "coco", seed=1, device=DEV)
z2 = torch.randn_like(z1)
state = {k: v.clone() for k, v in g.state_dict().items()}
sn = g.arena.sn_flat.data.clone()
def grads(pairs, together):
    g.load_state_dict(state)
    g.arena.sn_flat.data.copy_(sn)
    g.arena.drop_pending()
    g.zero_grad()
    if together:
        outs = [g(z, bbox, z_im, label) for z in pairs]
        sum(o.square().mean() for o in outs).backward()
    else:
        outs = []
        for z in pairs:
            o = g(z, bbox, z_im, label)
            outs.append(o.detach().clone())
            o.square().mean().backward()
    g.arena.flush_grads()
    torch.cuda.synchronize()
    return g.flat.grad.clone(), [o.detach().clone() for o in outs]
seq = "SSSSSSSS"
res = [grads((z1, z2), c == "T") for c in seq]
rel = lambda a, b: float((a - b).norm() / b.norm())
for i in range(len(seq)):
    print(seq[i], i, " ".join(f"{rel(res[i][0], res[j][0]):.1e}" for j in range(len(seq))), " | fwd diffs vs run0:", [float((x - y).abs().max()) for x, y in zip(res[i][1], res[0][1])])
base = g.flat.grad.data_ptr()
import itertools
i_, j_ = max(itertools.combinations(range(len(seq)), 2), key=lambda ij: rel(res[ij[0]][0], res[ij[1]][0]))
print('comparing runs', i_, j_)
a, b = res[i_][0], res[j_][0]
rows = []
for n, p in g.named_parameters():
    off = (p.grad.data_ptr() - base) // 4
    x, y = a[off:off + p.numel()], b[off:off + p.numel()]
    rows.append((rel(x, y) if float(y.norm()) > 0 else 0.0, float((x-y).norm()), n))
rows=[(r[1],r[0],r[2]) for r in rows]
for r in sorted(rows, reverse=True)[:25]:
    print(f"abs {r[0]:.2e} rel {r[1]:.2e} {r[2]}")
