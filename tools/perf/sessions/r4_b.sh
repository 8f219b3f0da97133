cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_layout.py tests/test_gpu_ddp.py tests/test_gpu_checkpoint.py tests/test_gpu_bench.py -x -q 2>&1 | tail -15
python bench.py --steps 20 --warmup 5 > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err; tail -c 1500 gpurun_out/r4b_bench.json
L2I_FORCE_COLLECTIVES=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-g-forward --no-f32-mode > gpurun_out/r4b_force.json 2> gpurun_out/r4b_force.err; tail -c 600 gpurun_out/r4b_force.json; tail -3 gpurun_out/r4b_force.err
L2I_FORCE_COLLECTIVES=1 L2I_DDP_GRAPH=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-g-forward --no-f32-mode > gpurun_out/r4b_force_graph.json 2> gpurun_out/r4b_force_graph.err; tail -c 600 gpurun_out/r4b_force_graph.json; tail -3 gpurun_out/r4b_force_graph.err
