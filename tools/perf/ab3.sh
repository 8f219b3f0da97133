cd $GRAFT_REPO_ROOT
ARGS="--steps 40 --no-cpu-baseline --no-g-forward --no-kernel-timer --no-f32-mode"
for i in 1 2; do
  (cd scratch/ab_prev && python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prev', d['value'], d['ms_per_step'])")
  L2I_SC_LAZY=0 python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new nofold', d['value'], d['ms_per_step'])"
  python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new fold', d['value'], d['ms_per_step'])"
done
