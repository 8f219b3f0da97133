# round 5: the working tree against the revision in _ab_prev (tools/perf/ab_prev.sh style, same box, ABAB)
cd $GRAFT_REPO_ROOT
ROOT=$PWD
A="--no-cpu-baseline --no-f32-mode --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], 'gfwd', d['g_forward']['ms'], d['g_forward'].get('sample_batch1_ms'))"; }
for i in 1 2; do (cd $ROOT/_ab_prev && run prev); (cd $ROOT && run new); done
