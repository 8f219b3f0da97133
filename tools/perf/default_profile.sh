#!/bin/bash
# usage (GPU box): bash tools/perf/default_profile.sh [tag=r03] -- the EXACT default command of the contract under rocprofv3:
#   rocprofv3 --kernel-trace --stats -- python bench.py     -> gpurun_out/<tag>_default_kernel_stats.csv + the JSON line of that run
# and the check that the profiler's average conv-kernel duration agrees with the line's roofline.avg_launch_us (HIP events).
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp -o d -- python $R/bench.py > /tmp/dp.json 2> /tmp/dp.err
cp $(find /tmp/dp -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_default_kernel_stats.csv
tail -1 /tmp/dp.json > $R/gpurun_out/${TAG}_default_bench_line.json
python - <<PY
import csv, json
line = json.loads(open("$R/gpurun_out/${TAG}_default_bench_line.json").read())
tot = n = tf = nf = 0.0
for r in csv.DictReader(open("$R/gpurun_out/${TAG}_default_kernel_stats.csv")):
    if r["Name"].startswith(("conv_wstat_reduce", "void conv_wstat_reduce", "conv_split_reduce", "void conv_split_reduce")):   # second kernel of a 4x4 weight-stationary launch: its time, not a launch
        tot += float(r["TotalDurationNs"])
    elif r["Name"].startswith(("void conv_halo", "void conv_igemm", "conv_halo", "conv_igemm", "conv_wstat_kernel", "void conv_wstat_kernel")):   # (a split launch = its halo kernel + conv_split_reduce: one launch)
        if "<float" in r["Name"]:   # the exact-f32 instantiations: only the line's secondary f32_mode / precision_modes legs launch them
            tf += float(r["TotalDurationNs"]); nf += int(r["Calls"])
        else:
            tot += float(r["TotalDurationNs"]); n += int(r["Calls"])
print(f"rocprofv3: {n:.0f} bf16 conv launches (conv_halo2/3, conv_igemm<unsigned short>, conv_wstat + its reduce), average {tot / n / 1e3:.2f} us "
      f"[+ {nf:.0f} conv_igemm<float> kernels of the secondary f32 legs, average {tf / max(nf, 1) / 1e3:.1f} us, not part of the roofline leg]; "
      f"bench line (under the profiler): avg_launch_us {line['roofline']['avg_launch_us']}, frac {line['roofline']['frac']}, value {line['value']} images/s")
PY
