#!/bin/bash
# rocprofv3 kernel stats of prev (scratch/ab_prev) and new (working tree), per-iteration, all kernels
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for side in prev new; do
  if [ $side = prev ]; then D=$R/scratch/ab_prev; else D=$R; fi
  rm -rf /tmp/ks_$side
  (cd $D && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$side -o ks -- python bench.py --no-cpu-baseline --no-f32-mode --no-g-forward --no-kernel-timer --steps 20 > /tmp/ks_$side.log 2>&1)
  echo "== $side $(tail -1 /tmp/ks_$side.log | cut -c1-100)"
  python $R/tools/perf/kstats.py $(find /tmp/ks_$side -name '*kernel_stats.csv' | head -1) > $R/gpurun_out/ks_$side.txt
  wc -l $R/gpurun_out/ks_$side.txt
done
