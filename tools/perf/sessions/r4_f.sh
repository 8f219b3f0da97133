cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dual.py tests/test_gpu_models.py -x -q -k "dual" 2>&1 | tail -5
bash tools/perf/prof.sh r4f > /dev/null 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r01_kernel_stats_r4f.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms over 8 steps", tot/1e6)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:70]:
    print(f'{float(r["TotalDurationNs"])/1e6/8:8.3f} ms/it x{int(r["Calls"])/8:6.1f} avg {float(r["AverageNs"])/1e3:7.1f} us  {r["Name"][:110]}')
PY
