cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_extra.py -q 2>&1 | tail -3
A="--no-cpu-baseline --no-g-forward --no-f32-mode --steps 30"
run() { python bench.py $A "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', '|', d['value'], d['ms_per_step'], r['frac'] if r else None, d['eager'])"; }
( run; run --layout vg; run --vgg; run --size 64 --batch 64 --dtype f32; run --batch 16; run --batch 64; L2I_DUAL_D=1 run --batch 32 ) > gpurun_out/r04_bench_other.txt 2>&1
cat gpurun_out/r04_bench_other.txt
