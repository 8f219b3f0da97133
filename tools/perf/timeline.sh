cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-g-forward --no-f32-mode > /tmp/tl.log 2>&1
T=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/perf/timeline.py $T 4
tail -1 /tmp/tl.log | cut -c1-120
python $GRAFT_REPO_ROOT/tools/perf/stock_in_graph.py $T 4 > $GRAFT_REPO_ROOT/gpurun_out/stock_in_graph.txt 2>&1
python $GRAFT_REPO_ROOT/tools/perf/tiny_critical.py $T 4 > $GRAFT_REPO_ROOT/gpurun_out/tiny_critical.txt 2>&1
python $GRAFT_REPO_ROOT/tools/perf/blank_kernels.py $T > $GRAFT_REPO_ROOT/gpurun_out/blank_kernels.txt 2>&1
