"""Timeline of the graph-replayed iteration (two streams) from a rocprofv3 kernel trace of bench.py:
    python timeline.py <kernel_trace.csv> [iters]
Reports, over the last `iters` iterations: wall per iteration, time with 0 / 1 / >= 2 kernels resident, and per kernel
class the summed duration (which, under overlap, exceeds what the kernel costs alone)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
ends = adam[1::2]
lo, hi = ends[-iters - 1] + 1, ends[-1] + 1
seg = rows[lo:hi]
t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
ev = []
for r in seg:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
lvl = 0; last = t0; occ = collections.Counter()
for t, d in ev:
    occ[min(lvl, 3)] += t - last; last = t; lvl += d
print(f"{iters} iterations: wall {(t1 - t0) / 1e6 / iters:.3f} ms/it; resident kernels 0: {occ[0] / 1e6 / iters:.3f}  1: {occ[1] / 1e6 / iters:.3f}  2: {occ[2] / 1e6 / iters:.3f}  >=3: {occ[3] / 1e6 / iters:.3f} ms/it; kernels/it {len(seg) / iters:.0f}")
cls = collections.Counter(); cnt = collections.Counter()
def klass(n):
    n = n.split("(")[0]
    for k, v in (("conv_halo", "conv"), ("conv_igemm", "conv"), ("conv_wstat", "conv"), ("conv_wgrad", "wgrad"), ("wgrad_reduce", "wgrad_reduce"), ("sn_", "spectral norm"), ("norm_", "norm"),
                 ("channel_stats", "norm"), ("ws_fold", "norm"), ("adam", "adam"), ("cast_kernel", "cast"), ("at::native", "aten"), ("rocclr", "rocclr"), ("Cijk", "rocblas")):
        if k in n: return v
    return "other own"
for r in seg:
    c = klass(r["Kernel_Name"]); cls[c] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[c] += 1
for c, d in cls.most_common():
    print(f"  {d / 1e6 / iters:7.3f} ms/it  x{cnt[c] / iters:6.1f}  {c}")
