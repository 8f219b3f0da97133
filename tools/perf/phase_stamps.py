"""Timeline of ONE graph replay of the training iteration without a profiler: wall-clock stamps (l2i_debug_stamp) launched at the phase
boundaries of GanTrainer._step are part of the captured graph; after replays the slots hold the device times of the last one.
    python tools/perf/phase_stamps.py [replays=20]
Prints the phases (main stream) in us, the iteration's span by the stamps, and the wall time per replay measured around the loop."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L
from layout2img_amd.synthetic import make_batch
from layout2img_amd.trainer import PhaseStamps

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.manual_seed(1234)
netG = L.ResnetGenerator128_context(num_classes=184).finalize(dev, torch.bfloat16)
netD = L.CombineDiscriminator128_app(num_classes=184).finalize(dev, torch.bfloat16)
tr = L.GanTrainer(netG, netD)
tr.stamps = PhaseStamps(dev)
real, label, bbox, z, z_im = make_batch(32, 128, "coco", seed=1234, device=dev)
assert tr.capture(real, label, bbox, None, None)
for _ in range(5):
    tr.step_graphed(real, label, bbox, None, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    tr.step_graphed(real, label, bbox, None, None)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
st = tr.stamps.read_us()
print(f"wall per replay {wall:.3f} ms ({n} replays, 11 stamp kernels inside); stamps of the last replay:")
for (a, ta), (b, tb) in zip(st, st[1:]):
    print(f"  {tb - ta:9.1f} us   {a}")
print(f"  {st[-1][1]:9.1f} us   start -> end (the rest of the wall time is the gap between replays)")
