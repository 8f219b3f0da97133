#!/bin/bash
# One gpurun call at the round's HEAD: the driver's test command (full log kept), the parity file's measured errors, one default bench line.
# usage: gpurun --timeout 1800 -- 'bash tools/perf/r06_final.sh'
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=25 -p no:cacheprovider > gpurun_out/r06_suite_head.txt 2>&1
echo "suite rc $?" | tee -a gpurun_out/r06_suite_head.txt
tail -1 gpurun_out/r06_suite_head.txt
python -m pytest tests/test_gpu_00_models.py -q -m gpu -rP -p no:cacheprovider 2>&1 | grep -E "^(loop|gradient norms|G coco|grad_|VGG|[0-9]+ passed)" > gpurun_out/r06_models_measured.txt
python bench.py > gpurun_out/r06_bench_head.json 2> gpurun_out/r06_bench_head.err
python tools/perf/cpu_time.py > gpurun_out/r06_host_time_head.txt 2>&1
tail -c 600 gpurun_out/r06_bench_head.json
# the generated binding on the GPU: the parity + determinism files through it, and the eager host time with / without it
L2I_FASTCALL=1 python -m pytest tests/test_gpu_00_models.py tests/test_gpu_06b_determinism.py tests/test_gpu_03_layout.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2 > gpurun_out/r06_fastcall_tests.txt
L2I_FASTCALL=1 python tools/perf/cpu_time.py >> gpurun_out/r06_fastcall_tests.txt 2>&1
cat gpurun_out/r06_host_time_head.txt gpurun_out/r06_fastcall_tests.txt
# the opt-in dead-f32-stream removal on the GPU: parity + determinism files with it, and an A/B pair of bench lines
L2I_F32_DEAD=1 python -m pytest tests/test_gpu_00_models.py tests/test_gpu_01_fullsize.py tests/test_gpu_06b_determinism.py tests/test_gpu_09_dual.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2 > gpurun_out/r06_f32dead_tests.txt
for k in 1 2; do
  python bench.py --no-kernel-timer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('default   ', d['value'], d['ms_per_step'])" >> gpurun_out/r06_f32dead_tests.txt
  L2I_F32_DEAD=1 python bench.py --no-kernel-timer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('F32_DEAD=1', d['value'], d['ms_per_step'])" >> gpurun_out/r06_f32dead_tests.txt
done
cat gpurun_out/r06_f32dead_tests.txt
