# round 5: short bench + kernel stats of the fill / copy kernels (after lending pre-zeroed pack buffers to the capture)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do python $R/bench.py --no-cpu-baseline --no-f32-mode --no-g-forward --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python $R/bench.py --no-cpu-baseline --no-f32-mode --no-g-forward --no-kernel-timer --steps 20 > /tmp/ks.log 2>&1
python $R/tools/perf/kstats.py $(find /tmp/ks -name "*kernel_stats.csv" | head -1) FillFunctor copyBuffer CUDAFunctor cast_kernel
