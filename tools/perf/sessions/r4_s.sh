cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_dual.py tests/test_gpu_models.py -x -q -k "wgrad or device_side or fused_conv or train_loop or vs_reference or full_size" 2>&1 | tail -3
A="--no-cpu-baseline --no-g-forward --no-f32-mode --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], r['wgrad_frac'])"; }
for i in 1 2; do L2I_WGRAD_OVERWRITE=0 run accumulate; run overwrite; done
