import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_models import _build_g, DEV
from tests.helpers import *
dt = torch.float32 if sys.argv[1] == 'f32' else torch.bfloat16
fx = load_fixture("g_coco.npz")
g = _build_g(fx, 11, dt)
inp = {k: v.to(DEV) for k, v in fixture_inputs(fx).items()}
g.train()
out1 = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
print('img maxdiff', maxdiff(out1, fx["out_train1"]))
proj = torch.randn(out1.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
g.zero_grad(); (out1 * proj).sum().backward(); g.arena.flush_grads()
named = dict(g.named_parameters())
names = [str(n) for n in fx["grad_names"]]
gn = np.array([float(named[n].grad.norm()) for n in names]); ref = fx["grad_norms"]
rel = np.abs(gn-ref)/(np.abs(ref)+1e-4*np.median(ref))
for i in np.argsort(-rel)[:45]:
    print(f"{names[i]:50s} {gn[i]:12.5f} {ref[i]:12.5f} {rel[i]:.2e}")
print('median rel', np.median(rel))
