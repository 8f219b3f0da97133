cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_models.py tests/test_gpu_dual.py -x -q -k "discriminator or train_loop or forward_dual or full_size" 2>&1 | tail -4
A="--no-cpu-baseline --no-g-forward --no-f32-mode --no-kernel-timer --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do L2I_JOIN_READERS=0 run base; run join; done
