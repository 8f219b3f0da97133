# round 5, A/B: ROI-align backward in gather form (default) against the scatter form (L2I_ROI_GATHER=0: atomics, cleared maps, cast of the maps)
cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-f32-mode --no-g-forward --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'])"; }
for i in 1 2; do L2I_ROI_GATHER=0 run gather_off; run gather_on; done
