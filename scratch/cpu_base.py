import sys, time, os, torch
sys.path.insert(0, '.')
from oracle import model as O
import layout2img_amd as L
from layout2img_amd.synthetic import make_batch
thr = int(sys.argv[1]); torch.set_num_threads(thr)
torch.manual_seed(0)
g = L.ResnetGenerator128_context(num_classes=184); d = L.CombineDiscriminator128_app(num_classes=184)
sd_g = O.make_trainable({k: v.detach().float().cpu() for k, v in g.state_dict().items()})
sd_d = O.make_trainable({k: v.detach().float().cpu() for k, v in d.state_dict().items()})
tr = O.OracleTrainer(sd_g, sd_d)
b = int(sys.argv[2])
real, label, bbox, z, z_im = make_batch(b, 128, "coco", seed=99)
for i in range(3):
    t0 = time.time(); tr.step(real, label, bbox, z, z_im); print(thr, b, i, round(time.time()-t0, 2), flush=True)
