import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L
from layout2img_amd.synthetic import make_batch
dev = torch.device('cuda:0')
torch.manual_seed(1234)
netG = L.ResnetGenerator128_context(num_classes=184).finalize(dev, torch.bfloat16)
netD = L.CombineDiscriminator128_app(num_classes=184).finalize(dev, torch.bfloat16)
tr = L.GanTrainer(netG, netD)
real, label, bbox, z, z_im = make_batch(32, 128, "coco", seed=1234, device=dev)
for _ in range(3): tr.step(real, label, bbox, z, None)
torch.cuda.synchronize()
n = 6
t0 = time.perf_counter()
for _ in range(n): tr.step(real, label, bbox, z, None)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"CPU enqueue {1e3*(t1-t0)/n:.1f} ms/step; wall {1e3*(t2-t0)/n:.1f} ms/step; cpus {os.cpu_count()}")
