# round 5, A/B: norm_mod8 with four row loads in flight (working tree) against the two-deep loop (_ab_prev = the revision before)
cd $GRAFT_REPO_ROOT
ROOT=$PWD
export TMPDIR=/tmp
A="--no-cpu-baseline --no-f32-mode --no-g-forward --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'])"; }
for i in 1 2; do (cd $ROOT/_ab_prev && run prev); (cd $ROOT && run new); done
for side in prev new; do
  if [ $side = prev ]; then cd $ROOT/_ab_prev; else cd $ROOT; fi
  rm -rf /tmp/ks_$side
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$side -o ks -- python bench.py --no-cpu-baseline --no-f32-mode --no-g-forward --no-kernel-timer --steps 20 > /dev/null 2>&1
  echo "== $side"; python $ROOT/tools/perf/kstats.py $(find /tmp/ks_$side -name '*kernel_stats.csv' | head -1) norm_ stats | head -14
done
