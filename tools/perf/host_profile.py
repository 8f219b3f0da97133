"""Host cost of enqueueing one eager training iteration, without back-pressure from the GPU: the same launches at batch 2 (the GPU
finishes long before the host), wall time per step and the cProfile top of the Python side.  usage: python tools/perf/host_profile.py"""
import cProfile, io, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L
from layout2img_amd.synthetic import make_batch
dev = torch.device("cuda:0")
torch.manual_seed(1234)
netG = L.ResnetGenerator128_context(num_classes=184).finalize(dev, torch.bfloat16)
netD = L.CombineDiscriminator128_app(num_classes=184).finalize(dev, torch.bfloat16)
tr = L.GanTrainer(netG, netD)
for b in (32, 2):
    real, label, bbox, z, z_im = make_batch(b, 128, "coco", seed=1234, device=dev)
    for _ in range(3):
        tr.step(real, label, bbox, z, None)
    torch.cuda.synchronize()
    n = 8
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(real, label, bbox, z, None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"batch {b}: host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, wall {1e3 * (t2 - t0) / n:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    tr.step(real, label, bbox, z, None)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:60]))
