"""Small kernels of the replayed iteration that run while NOTHING else is resident (the side stream idle): each one costs the iteration
its whole duration plus a dispatch boundary.   python tiny_critical.py <kernel_trace.csv> [iters] [max_us=8]
Groups them by kernel name: launches per iteration, total us per iteration; and the gap time (no kernel resident) by the kernel that follows."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lim = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
ends = adam[1::2]
seg = rows[ends[-iters - 1] + 1:ends[-1] + 1]
S = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in seg]
def short(n):
    n = n.split("(")[0].replace("void ", "").replace("at::native::", "")
    return n[:70]
alone = collections.defaultdict(lambda: [0, 0.0]); gaps = collections.defaultdict(lambda: [0, 0.0])
tot_alone = tot_gap = 0.0
maxend = S[0][0]
for i, (s, e, n) in enumerate(S):
    # overlapped with any other kernel?
    ov = False
    for j in range(max(0, i - 6), min(len(S), i + 7)):
        if j != i and S[j][0] < e and S[j][1] > s:
            ov = True; break
    d = (e - s) / 1e3
    if not ov and d <= lim:
        a = alone[short(n)]; a[0] += 1; a[1] += d; tot_alone += d
    if s > maxend:
        g = gaps[short(n)]; g[0] += 1; g[1] += (s - maxend) / 1e3; tot_gap += (s - maxend) / 1e3
    maxend = max(maxend, e)
print(f"kernels <= {lim} us running alone: {sum(a[0] for a in alone.values()) / iters:.0f} per iteration, {tot_alone / iters:.1f} us per iteration; idle gaps {tot_gap / iters:.1f} us per iteration")
for k, (c, d) in sorted(alone.items(), key=lambda kv: -kv[1][1])[:40]:
    g = gaps.get(k, [0, 0.0])
    print(f"  {d / iters:7.1f} us/it  x{c / iters:5.1f}  avg {d / c:5.1f}   + gap before {g[1] / iters:6.1f} us/it   {k}")
print("gaps by the kernel that follows (all kernels):")
for k, (c, d) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {d / iters:7.1f} us/it  x{c / iters:5.1f}  avg {d / c:5.2f}   {k}")
