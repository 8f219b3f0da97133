"""Idle gaps between kernels of the graph-replayed iteration: python gap_census.py <kernel_trace.csv> [iters]
Takes the trace of `bench.py` (one stream), finds the last `iters` adam_kernel pairs and reports wall span, busy time and
the gaps by the kernel that FOLLOWS them."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
# an iteration = from just after an G-step adam to the next G-step adam (every second adam)
ends = adam[1::2]
lo, hi = ends[-iters - 1] + 1, ends[-1] + 1
seg = rows[lo:hi]
t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
busy = 0; cur_end = t0; gaps = collections.Counter(); gcnt = collections.Counter(); big = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_end:
        g = s - cur_end
        name = r["Kernel_Name"].split("(")[0][:60]
        gaps[name] += g; gcnt[name] += 1
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
print(f"{iters} iterations: wall {(t1-t0)/1e6/iters:.3f} ms/it, busy {busy/1e6/iters:.3f} ms/it, idle {(t1-t0-busy)/1e6/iters:.3f} ms/it, kernels/it {len(seg)/iters:.0f}")
tot = sum(gaps.values())
for n, g in gaps.most_common(25):
    print(f"  {g/1e6/iters:7.3f} ms/it  x{gcnt[n]/iters:6.1f}  avg {g/gcnt[n]/1e3:6.2f} us  before {n}")
