import os, sys, torch, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import layout2img_amd as L
from layout2img_amd.synthetic import make_batch
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
torch.manual_seed(1234)
netG = L.ResnetGenerator128_context(num_classes=184).finalize(dev, torch.bfloat16)
netG.train()
real, label, bbox, z, z_im = make_batch(32, 128, "coco", seed=1234, device=dev)
with torch.no_grad():
    for _ in range(3): netG(z, bbox, z_im=z_im, y=label)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        netG(z, bbox, z_im=z_im, y=label)
        torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA and ("emcpy" in e.name or "emset" in e.name or "copyBuffer" in e.name):
        c[e.name] += 1
print(c)
c2 = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("hipMem"):
        p = e.cpu_parent
        c2[(e.name, p.name if p else None)] += 1
print(c2.most_common(20))
