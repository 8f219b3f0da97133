cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dual.py -x -q 2>&1 | tail -15
python -m pytest tests/test_gpu_models.py tests/test_res64.py tests/test_gpu_ddp.py -x -q 2>&1 | tail -8
A="--no-cpu-baseline --no-g-forward --no-f32-mode --steps 40"
for i in 1 2; do
L2I_DUAL_D=0 python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('single', d['value'], d['ms_per_step'], r['frac'], r['wgrad_frac'], r['launches_per_step'])"
python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('dual  ', d['value'], d['ms_per_step'], r['frac'], r['wgrad_frac'], r['launches_per_step'])"
done
