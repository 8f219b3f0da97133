# round 5, A/B: split-K by stored partial tiles + reduce-with-epilogue (L2I_CONV_PART=1, default) against the round-4 rule (=0)
cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-g-forward --no-f32-mode --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], r['wgrad_frac'], r['kernels_per_step'])"; }
for i in 1 2; do L2I_CONV_PART=0 run part_off; run part_on; done
