"""Graph replay vs eager iterations from the same state, twice each: pairwise distances of gradients / parameters / losses and the parameter
tensors whose gradients differ (round 6: found the 4-way bias sum of the doubly applied block_obj4 on two streams, DESIGN A.r06.3).
    python tools/parity/graph_vs_eager.py [bf16|f32] [iterations]"""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import layout2img_amd as L
from layout2img_amd.synthetic import make_batch
from layout2img_amd.trainer import restore_state, snapshot_state
DEV = "cuda:0"
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
def nets():
    torch.manual_seed(0)
    g = L.ResnetGenerator128_context(num_classes=184).finalize(DEV, dt).train()
    d = L.CombineDiscriminator128_app(num_classes=184).finalize(DEV, dt).train()
    for m in g.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
    return g, d
def run(mode):
    g, d = nets()
    tr = L.GanTrainer(g, d)
    real, label, bbox, z, z_im = make_batch(8, 128, "coco", seed=3, device=torch.device(DEV))
    if mode == "graph":
        st = snapshot_state(tr)
        assert tr.capture(real, label, bbox, z, z_im)
        restore_state(tr, st)
        for _ in range(iters):
            r = tr.step_graphed(real, label, bbox, z, z_im)
    else:
        for _ in range(iters):
            r = tr.step(real, label, bbox, z, z_im)
    tr.flush(); torch.cuda.synchronize()
    return g, d, dict(gg=g.flat.grad.clone(), dg=d.flat.grad.clone(), gp=g.flat.data.clone(), dp=d.flat.data.clone(), fake=r["fake"].clone(), dl=float(r["d_loss"]), gl=float(r["g_loss"]))
res = {}
for m in ("graph", "eager", "graph2", "eager2"):
    g, d, res[m] = run(m.rstrip("2"))
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
for a, b in (("graph", "graph2"), ("eager", "eager2"), ("graph", "eager")):
    print(a, b, {k: (rel(res[a][k], res[b][k]) if torch.is_tensor(res[a][k]) else (res[a][k], res[b][k])) for k in res[a]})
for net, key in ((d, "dg"), (g, "gg")):
    base = net.flat.grad.data_ptr()
    rows = []
    for n, p in net.named_parameters():
        off = (p.grad.data_ptr() - base) // 4
        x, y = res["graph"][key][off:off + p.numel()], res["eager"][key][off:off + p.numel()]
        if not torch.equal(x, y):
            rows.append((rel(x, y), n))
    print(key, len(rows), "differ:", rows[:12])
