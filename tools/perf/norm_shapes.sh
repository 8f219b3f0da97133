#!/bin/bash
# usage (GPU box): tools/perf/norm_shapes.sh [lib.so ...]  -- per-shape norm kernel times for the in-tree library, then for each alternative library given
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/layout2img_amd/libl2i_hip.so /tmp/l2i_default.so
for v in default "$@"; do
  [ $v = default ] || cp $R/$v $R/layout2img_amd/libl2i_hip.so
  rm -rf /tmp/ns
  rocprofv3 --kernel-trace -d /tmp/ns -o t --output-format csv -- python $R/tools/perf/norm_shapes.py run > /tmp/ns.log 2>&1
  echo "######## $v"
  python $R/tools/perf/norm_shapes.py join $(find /tmp/ns -name "*kernel_trace.csv" | head -1) ${NS_FILTER:-norm_bwd_a8}
done
cp /tmp/l2i_default.so $R/layout2img_amd/libl2i_hip.so
