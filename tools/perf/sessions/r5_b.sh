# round 5, A/B: under-filled launches on 128x128 tiles + K splits (L2I_PART_BIG=1) against 128x64 tiles (=0)
cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-g-forward --no-f32-mode --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], r['wgrad_frac'], r['kernels_per_step'])"; }
for i in 1 2; do L2I_PART_BIG=0 run big_off; run big_on; done
