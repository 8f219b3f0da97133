import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L
from layout2img_amd.synthetic import make_batch
dev = torch.device("cuda:0")
def build():
    torch.manual_seed(5)
    g = L.ResnetGenerator64_context(num_classes=184).finalize(dev, torch.float32)
    d = L.CombineDiscriminator64(num_classes=184).finalize(dev, torch.float32)
    for m in g.modules():
        if hasattr(m, "dropout_p"): m.dropout_p = 0.0
    return g, d, L.GanTrainer(g, d)
real, label, bbox, z, z_im = make_batch(4, 64, "coco", seed=3, device=dev)
res = []
for mode in ("eager", "eager", "graph"):
    g, d, t = build()
    if mode == "graph":
        t.capture(real, label, bbox, z, z_im)
        for _ in range(3): o = t.step_graphed(real, label, bbox, z, z_im)
    else:
        for _ in range(5): o = t.step(real, label, bbox, z, z_im)
    torch.cuda.synchronize()
    res.append((mode, float(o["d_loss"]), float(o["g_loss"]), g.flat.data.clone(), d.flat.data.clone()))
for i in (1, 2):
    a, b = res[0], res[i]
    print(b[0], "vs eager: d_loss", a[1], b[1], "g_loss", a[2], b[2], "G maxdiff", float((a[3]-b[3]).abs().max()), "frac<1e-4", float(((a[3]-b[3]).abs()<1e-4).float().mean()),
          "D frac<1e-4", float(((a[4]-b[4]).abs()<1e-4).float().mean()))
