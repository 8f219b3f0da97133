"""Which call sites of ONE generator forward (train mode) and one full training iteration launch a device-to-device copy
(__amd_rocclr_copyBuffer / aten::copy_): torch profiler with python stacks, grouped by the innermost repository frame and tensor shape."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L
from layout2img_amd.synthetic import make_batch
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
torch.manual_seed(1234)
netG = L.ResnetGenerator128_context(num_classes=184).finalize(dev, torch.bfloat16).train()
netD = L.CombineDiscriminator128_app(num_classes=184).finalize(dev, torch.bfloat16).train()
tr = L.GanTrainer(netG, netD)
real, label, bbox, z, z_im = make_batch(32, 128, "coco", seed=1234, device=dev)
for _ in range(2): tr.step(real, label, bbox, z, None)
torch.cuda.synchronize()


def census(fn, title):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        fn()
        torch.cuda.synchronize()
    cnt = collections.Counter()
    mem = collections.Counter()
    for e in prof.events():
        if "emcpy" in e.name or "copyBuffer" in e.name:
            mem[e.name[:60]] += 1
        ks = [k.name for k in getattr(e, "kernels", [])]
        if any("emcpy" in k or "copyBuffer" in k for k in ks):
            site = "?"
            for fr in e.stack:
                if "/layout2img_amd/" in fr or "bench.py" in fr:
                    site = fr.split("/layout2img_amd/")[-1]
                    break
            mem[("op", e.name, site[:60], str(e.input_shapes)[:50])] += 1
    print("   memcpy-like events:", dict(mem))
    for e in prof.events():
        if e.name == "aten::copy_":
            site = "?"
            for fr in e.stack:
                if "/layout2img_amd/" in fr or "bench.py" in fr:
                    site = fr.split("/layout2img_amd/")[-1] if "/layout2img_amd/" in fr else fr
                    break
            cnt[(site[:70], str(e.input_shapes)[:60])] += 1
    print(f"== {title}: {sum(cnt.values())} aten::copy_ calls")
    for (site, sh), n in cnt.most_common(40):
        print(f"  x{n:3d}  {site:70s} {sh}")


with torch.no_grad():
    pass
census(lambda: netG(z, bbox, z_im, label), "generator forward (train mode, grad on)")
census(lambda: tr.step(real, label, bbox, z, None), "training iteration")
