# round 5, A/B: generator block results' two gradients (next block + mask head) joined in the head's data-gradient launch (default) against autograd add + cast (L2I_JOIN_HEADS=0)
cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-f32-mode --no-g-forward --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], r['launches_per_step'], r['kernels_per_step'])"; }
for i in 1 2; do L2I_JOIN_HEADS=0 run join_off; run join_on; done
