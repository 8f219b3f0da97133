cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dual.py -m gpu -x -q -k "wgrad" 2>&1 | tail -2
bash tools/perf/ab_run.sh
