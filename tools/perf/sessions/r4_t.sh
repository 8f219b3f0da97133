cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-g-forward --no-f32-mode --no-kernel-timer --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
L2I_DW_NAN=1 timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_dual.py -m gpu -x -q -k "loop or full_size or forward_dual or wgrad" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_extra.py -m gpu -x -q -k "wgrad or overwrite or linear or head or two_forwards" 2>&1 | tail -3
run new
run new
