"""usage: python tools/perf/kstats.py <kernel_stats.csv> [substr ...]: per-iteration time of the kernels whose names contain a substring"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
adam = [r for r in rows if "adam_kernel" in r["Name"]]
its = int(adam[0]["Calls"]) / 2 if adam else 1
keys = sys.argv[2:]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows:
    n = r["Name"]
    if not keys or any(k in n for k in keys):
        print("%8.1f us/it x%5.1f avg %6.1f  %s" % (float(r["TotalDurationNs"]) / its / 1e3, int(r["Calls"]) / its, float(r["AverageNs"]) / 1e3, n[:80]))
