#!/bin/bash
# usage: tools/perf/shape_profile.sh <tag>   (on the GPU box; writes gpurun_out/shapes_<tag>.txt)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/sp_$1
rocprofv3 --kernel-trace -d /tmp/sp_$1 -o trace --output-format csv -- python $R/tools/perf/shape_profile.py run /tmp/spcalls_$1.json > /tmp/sp_$1.log 2>&1
T=$(find /tmp/sp_$1 -name '*kernel_trace.csv' | head -1)
python $R/tools/perf/shape_profile.py join /tmp/spcalls_$1.json $T $R/gpurun_out/shapes_$1.txt || tail -20 /tmp/sp_$1.log
