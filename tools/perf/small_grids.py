"""Kernels of a graph-replayed bench run whose launches are long for the number of workgroups they have:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/sg -o t -- python bench.py --steps 8 --no-cpu-baseline --no-g-forward --no-f32-mode --no-kernel-timer
    python tools/perf/small_grids.py <kernel_trace.csv> [max workgroups = 512] [min us = 6]
Per (kernel, grid): launches per iteration, workgroups, average us, total us per iteration -- the candidates for more parallelism."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
maxwg = int(sys.argv[2]) if len(sys.argv) > 2 else 512
minus = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
its = sum(1 for r in rows if "adam_kernel" in r["Kernel_Name"]) / 2 or 1
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    w = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    k = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:70], g // max(w, 1), w)
    a = agg[k]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = []
for (n, wgs, w), (c, t) in agg.items():
    avg = t / c / 1e3
    if wgs <= maxwg and avg >= minus:
        out.append((t / its / 1e3, c / its, wgs, w, avg, n))
for tot, c, wgs, w, avg, n in sorted(out, reverse=True)[:int(__import__("os").environ.get("SG_TOP", "40"))]:
    print(f"{tot:8.1f} us/it  x{c:5.1f}  {wgs:6d} WGs x {w:4d}  avg {avg:7.1f} us  {n}")
