"""The Python side of one eager training iteration WITHOUT a GPU: meta-device tensors, the C ABI replaced by a recorder (tests/dryrun.py).
What it times is the host work of this package and of torch's autograd (Function.apply, the engine thread, argument marshalling) -- not
hipLaunchKernel, not the ctypes foreign call itself; stock torch ops run through the meta device's Python shape functions, which are SLOWER
than their CUDA dispatch, so read the package's own rows of the profile, not the total.
usage: python tools/perf/host_dryrun.py [tottime|cumtime] [coco|vg]"""
import cProfile, collections, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import dryrun

order = sys.argv[1] if len(sys.argv) > 1 else "tottime"
kind = sys.argv[2] if len(sys.argv) > 2 else "coco"
with dryrun.dry_run() as trace:
    tr, (real, label, bbox, z, z_im) = dryrun.build(kind, torch.bfloat16)
    step = lambda: tr.step(real, label, bbox, z, z_im if kind == "vg" else None)
    for _ in range(3):
        del trace[:]
        step()
    c = collections.Counter(n for n, _ in trace)
    print(f"{len(trace)} C-ABI calls per iteration; the ten most frequent: " + ", ".join(f"{n} x{k}" for n, k in c.most_common(10)))
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        del trace[:]
        step()
    print(f"dry host time {1e3 * (time.perf_counter() - t0) / n:.2f} ms per iteration on {os.cpu_count()} cores (this machine, no GPU)")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        del trace[:]
        step()
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats(order).print_stats(60)
rows = [l[:170] for l in s.getvalue().splitlines()]
print("\n".join(l for l in rows if "dist-packages" not in l and "/usr/lib" not in l))
