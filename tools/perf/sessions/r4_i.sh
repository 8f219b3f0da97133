cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-g-forward --no-f32-mode --no-kernel-timer --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
L2I_DW_NAN=1 timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_dual.py tests/test_gpu_extra.py tests/test_gpu_ddp.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "wgrad or overwrite or linear or head" 2>&1 | tail -3
run new
L2I_WGRAD_OVERWRITE=0 run no_overwrite
run new
L2I_WGRAD_OVERWRITE=0 run no_overwrite
