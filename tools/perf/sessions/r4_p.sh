cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-g-forward --no-f32-mode --no-kernel-timer --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['eager'])"; }
run base
L2I_MAIN_PRIO=-1 run main_high
L2I_MAIN_PRIO=-1 L2I_REAL_BWD_EARLY=1 run main_high_early
run base
