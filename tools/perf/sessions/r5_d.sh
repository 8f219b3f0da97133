# round 5, A/B: mask heads evaluated for the gathered classes only (L2I_CLASS_GATHER=1, default) against the dense 184-channel heads (=0)
cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-f32-mode --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], r['launches_per_step'], r['kernels_per_step'], 'gfwd', d['g_forward']['ms'], d['g_forward']['sample_batch1_ms'])"; }
for i in 1 2; do L2I_CLASS_GATHER=0 run gather_off; run gather_on; done
