# round 5, A/B: ISLA backward's projection gradients through partial rows + finish launch (default) against atomics (L2I_NORM_PART=0)
cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-f32-mode --no-g-forward --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], r['launches_per_step'], r['kernels_per_step'])"; }
for i in 1 2; do L2I_NORM_PART=0 run part_off; run part_on; done
