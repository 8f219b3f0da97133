#!/bin/bash
# usage (GPU box): bash tools/perf/round_profiles.sh [tag=r03]  -> everything the round's DESIGN.md numbers come from, under gpurun_out/<tag>_*
# (copied into profiles/ afterwards): bench line, rocprofv3 kernel stats of the iteration and of the generator forward, HBM /
# MFMA counter passes (traffic2.sh), in-situ per-layer table, launch censuses, per-wave traces of representative launches.
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python bench.py > gpurun_out/${TAG}_bench_final.json 2> gpurun_out/${TAG}_bench_final.err
python bench.py --steps 5 --no-cpu-baseline --no-f32-mode --no-kernel-timer --g-batch-sweep 2> /dev/null | tail -1 | python -c "
import json, sys; g = json.loads(sys.stdin.read())['g_forward']; g.pop('precision_modes', None); print(json.dumps(g))" > gpurun_out/${TAG}_gfwd_batch_sweep.json
bash tools/perf/traffic2.sh $TAG > gpurun_out/${TAG}_traffic.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gf && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gf -o g -- python $R/tools/perf/gfwd_profile.py 10 > /tmp/gf.log 2>&1
cp $(find /tmp/gf -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_gfwd_kernel_stats.csv
cd $R
bash tools/perf/shape_profile.sh $TAG > /dev/null 2>&1 && mv gpurun_out/shapes_$TAG.txt gpurun_out/${TAG}_conv_shapes.txt
python tools/perf/gfwd_census.py 2>&1 | grep -v "amdgpu\|Warn\|warn" | cut -c1-160 > gpurun_out/${TAG}_gfwd_census.txt
python tools/perf/op_census3.py 2>&1 | grep -v "amdgpu\|Warn\|warn" > gpurun_out/${TAG}_op_census.txt
bash tools/perf/timeline.sh > gpurun_out/${TAG}_graph_timeline.txt 2>&1
python tools/perf/cpu_time.py 2>&1 | tail -1 > gpurun_out/${TAG}_host_time.txt
tail -c 400 gpurun_out/${TAG}_bench_final.json; cat gpurun_out/${TAG}_host_time.txt
