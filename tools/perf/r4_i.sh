cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_models.py -x -q -s -k "bf16x3" 2>&1 | grep "bf16x3 \|passed\|failed"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode > gpurun_out/r4i_bench.json 2> gpurun_out/r4i_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r4i_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(json.dumps(d['g_forward'], indent=1))"
tail -3 gpurun_out/r4i_bench.err
