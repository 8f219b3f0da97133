cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dual.py -q -k forward_dual 2>&1 | grep -E "AssertionError|passed|failed|^FAILED" | head -20
python -m pytest "tests/test_gpu_models.py" -x -q 2>&1 | tail -5
A="--no-cpu-baseline --no-g-forward --no-f32-mode --steps 30"
L2I_DUAL_D=0 python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('single', d['value'], d['ms_per_step'], d['eager'])"
python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('dual  ', d['value'], d['ms_per_step'], d['eager'])"
python tools/perf/cpu_time.py 2>&1 | tail -1
L2I_DUAL_D=0 python tools/perf/cpu_time.py 2>&1 | tail -1
