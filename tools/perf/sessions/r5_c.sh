# round 5, A/B: the D blocks' 1x1 shortcut data gradients folded into conv1's data-gradient launch (L2I_DGRAD_FOLD=1, default) against separate launches (=0)
cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-g-forward --no-f32-mode --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], r['launches_per_step'], r['kernels_per_step'])"; }
for i in 1 2; do L2I_DGRAD_FOLD=0 run fold_off; run fold_on; done
