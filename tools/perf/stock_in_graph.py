"""Stock (ATen / runtime) kernels inside the graph-replayed iteration, from a rocprofv3 kernel trace of bench.py:
    python stock_in_graph.py <kernel_trace.csv> [iters]
per (kernel, grid size): launches per iteration, average and total duration -- which fills / adds / copies are large."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
ends = adam[1::2]
seg = rows[ends[-iters - 1] + 1:ends[-1] + 1]
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = r["Kernel_Name"]
    if "at::native" not in n and "rocclr" not in n:
        continue
    short = n.split("(")[0].replace("void at::native::", "").replace("vectorized_elementwise_kernel", "vec")[:90]
    g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    a = agg[(short, g)]
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
print(f"stock kernels: {sum(a[0] for a in agg.values()) / iters:.0f} per iteration, {tot / 1e3 / iters:.1f} us per iteration")
for (k, g), (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  {d / 1e3 / iters:7.1f} us/it  x{c / iters:5.1f}  avg {d / 1e3 / c:6.1f} us  grid {g:>10d}  {k}")
