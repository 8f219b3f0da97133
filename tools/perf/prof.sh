#!/bin/bash
# usage: tools/perf/prof.sh <tag>  -> gpurun_out/r01_kernel_stats_<tag>.csv
cd /tmp && export TMPDIR=/tmp
export L2I_OVERLAP=0   # one stream: per-kernel durations are those of a kernel that owns the GPU
rm -rf /tmp/prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-g-forward --no-f32-mode > /tmp/b.log 2>&1
tail -1 /tmp/b.log | cut -c1-160
f=$(ls /tmp/prof/*/*kernel_stats.csv | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/r01_kernel_stats_$1.csv
