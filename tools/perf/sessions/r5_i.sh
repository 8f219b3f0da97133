# round 5, A/B: two-reader gradient joins of the discriminator incl. the ROI features (default) against autograd adds (L2I_JOIN_READERS=0)
cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-f32-mode --no-g-forward --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'])"; }
for i in 1 2; do L2I_JOIN_READERS=0 run join_off; run join_on; done
