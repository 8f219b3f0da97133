"""Generator forward alone (train-mode statistics, no autograd tape), eager, for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L
from layout2img_amd.synthetic import make_batch
dev = torch.device("cuda:0")
torch.manual_seed(1234)
netG = L.ResnetGenerator128_context(num_classes=184).finalize(dev, torch.bfloat16)
netG.train()
real, label, bbox, z, z_im = make_batch(32, 128, "coco", seed=1234, device=dev)
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        netG(z, bbox, z_im=z_im, y=label)
torch.cuda.synchronize()
