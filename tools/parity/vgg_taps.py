import sys, os, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", "."))
from tests import test_gpu_00_models as T
from tests.helpers import vgg_inputs
x0, y0 = vgg_inputs()
taps, grads = {}, {}
for dt in (torch.float32, torch.bfloat16):
    m, fx = T._vgg(dt)
    x = x0.to(T.DEV).requires_grad_(True)
    l = m(x, y0.to(T.DEV)); l.backward()
    with torch.no_grad():
        taps[dt] = [torch.relu(t).float().cpu() for t in m.vgg(m._nhwc8(x0.to(T.DEV)))]
    grads[dt] = x.grad.detach().float().cpu()
print("tap rel L2:", [f"{float((a-b).norm()/b.norm()):.3g}" for a, b in zip(taps[torch.bfloat16], taps[torch.float32])])
ga, gb = grads[torch.bfloat16], grads[torch.float32]
print("grad cos", float((ga*gb).sum()/(ga.norm()*gb.norm())), "rel", float((ga-gb).norm()/gb.norm()))
