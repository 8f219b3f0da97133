import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L
from tests.helpers import vgg_inputs, vgg_state, load_fixture
DEV="cuda:0"
fx = load_fixture("vgg.npz")
ref = float(fx["loss"]); g = torch.from_numpy(fx["grad_x_sub"])
from layout2img_amd.synthetic import make_batch
# interleave with trainer steps (pool / scratch / other allocations churn memory)
netG = L.ResnetGenerator128_context(num_classes=184).finalize(DEV, torch.bfloat16)
netD = L.CombineDiscriminator128_app(num_classes=184).finalize(DEV, torch.bfloat16)
tr = L.GanTrainer(netG, netD)
batch = make_batch(8, 128, "coco", seed=1, device=torch.device(DEV))[:4]
for rep in range(12):
    for dt in (torch.float32, torch.bfloat16):
        m = L.VGGLoss(); m.load_state_dict(vgg_state(fx)); m.finalize(DEV, dt)
        x, y = vgg_inputs()
        x = x.to(DEV).requires_grad_(True)
        loss = m(x, y.to(DEV)); loss.backward()
        err = float((x.grad[:, :, ::2, ::2].cpu() - g).norm() / g.norm())
        print(rep, str(dt)[6:], "loss rel %.3e" % (abs(float(loss) - ref) / abs(ref)), "grad err %.3e" % err, flush=True)
    tr.step(*batch)
    junk = torch.full((64 << 20,), float("nan"), device=DEV); del junk   # poison freed memory
