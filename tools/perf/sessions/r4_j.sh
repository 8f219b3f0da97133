cd $GRAFT_REPO_ROOT
L2I_WGRAD_FUSE=1 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dual.py -q -k "wgrad or device_side" 2>&1 | tail -4
A="--no-cpu-baseline --no-g-forward --no-f32-mode --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['wgrad_frac'], r.get('wgrad_launches_per_step'))"; }
for i in 1 2; do run base; L2I_WGRAD_FUSE=1 run fuse; done
