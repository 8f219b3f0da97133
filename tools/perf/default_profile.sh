#!/bin/bash
# usage (GPU box): bash tools/perf/default_profile.sh [tag=r03] -- the EXACT default command of the contract under rocprofv3:
#   rocprofv3 --kernel-trace --stats -- python bench.py     -> gpurun_out/<tag>_default_kernel_stats.csv + the JSON line of that run
# and the check that the profiler's average conv-kernel duration agrees with the line's roofline.avg_launch_us (HIP events).
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp -o d -- python $R/bench.py > /tmp/dp.json 2> /tmp/dp.err
cp $(find /tmp/dp -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_default_kernel_stats.csv
tail -1 /tmp/dp.json > $R/gpurun_out/${TAG}_default_bench_line.json
python - <<PY
import csv, json
line = json.loads(open("$R/gpurun_out/${TAG}_default_bench_line.json").read())
tot = n = 0.0
for r in csv.DictReader(open("$R/gpurun_out/${TAG}_default_kernel_stats.csv")):
    if r["Name"].startswith(("void conv_halo", "void conv_igemm", "conv_halo", "conv_igemm")):
        tot += float(r["TotalDurationNs"]); n += int(r["Calls"])
print(f"rocprofv3: {n:.0f} conv kernels, average {tot / n / 1e3:.2f} us; bench line (under the profiler): avg_launch_us {line['roofline']['avg_launch_us']}, "
      f"frac {line['roofline']['frac']}, value {line['value']} images/s")
PY
