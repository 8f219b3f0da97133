cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_checkpoint.py tests/test_gpu_bench.py -m gpu -x -q 2>&1 | tail -3
A="--no-cpu-baseline --no-g-forward --no-f32-mode --no-kernel-timer --steps 40"
run() { python bench.py $A $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['config'].get('iterations_per_replay'))"; }
for i in 1 2; do run one "--graph-iters 1"; run four "--graph-iters 4"; run eight "--graph-iters 8"; done
python bench.py --steps 21 --warmup 3 --no-cpu-baseline --no-g-forward --no-f32-mode 2>/dev/null | tail -1 | cut -c1-300
timeout 300 python -m layout2img_amd.train --dataset coco --batch_size 32 --synthetic 1002 --total_epoch 1 --out_path /tmp/l2i_out 2>/dev/null | grep -v amdgpu
