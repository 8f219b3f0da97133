# round 5, A/B: padded-channel bias gradients summed by the weight-gradient launch into a slot of the accumulator (default) against a channel_stats pass over dY (L2I_BIAS_SLOTS=0)
cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-f32-mode --no-g-forward --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'])"; }
for i in 1 2; do L2I_BIAS_SLOTS=0 run slots_off; run slots_on; done
