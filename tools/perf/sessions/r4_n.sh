cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -q -k "halo_tiles" 2>&1 | tail -2
A="--no-cpu-baseline --no-g-forward --no-f32-mode --no-kernel-timer --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
run base; L2I_HC6=64 run hc6_64; L2I_HC6=16 run hc6_16; L2I_HC6=8 run hc6_8; run base
