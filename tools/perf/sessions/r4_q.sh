cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "small_map or conv_forward or split_k or conv_relu_mask" 2>&1 | tail -3
A="--no-cpu-baseline --no-g-forward --no-f32-mode --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], r['kernels_per_step'])"; }
for i in 1 2; do L2I_WSTAT=0 run base; run wstat; done
