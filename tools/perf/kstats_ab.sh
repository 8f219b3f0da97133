#!/bin/bash
# usage (GPU box): tools/perf/kstats_ab.sh "ENV=0" "ENV=1" [substr ...] -- rocprofv3 --stats of a short bench run under each environment, per-iteration kernel times side by side
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
EA=$1; EB=$2; shift 2
for side in A B; do
  if [ $side = A ]; then E=$EA; else E=$EB; fi
  rm -rf /tmp/ks_$side
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$side -o ks -- python $R/bench.py --no-cpu-baseline --no-f32-mode --no-g-forward --no-kernel-timer --steps 20 > /tmp/ks_$side.log 2>&1
  echo "== $side: $E   $(tail -1 /tmp/ks_$side.log | cut -c1-90)"
  python $R/tools/perf/kstats.py $(find /tmp/ks_$side -name '*kernel_stats.csv' | head -1) "$@" | head -${KS_HEAD:-30}
done
