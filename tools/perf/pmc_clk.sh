#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for b in 4 8 32; do
rm -rf /tmp/pmc3
L2I_CFG=$1 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pmc3 -o c -- python $GRAFT_REPO_ROOT/tools/perf/conv_micro.py $b 32 32 512 512 3 0 10 2>&1 | grep shape
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc3/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "conv_" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v)/len(v)) for k, v in agg.items()})
kt = glob.glob("/tmp/pmc3/**/*kernel_trace.csv", recursive=True)[0]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "conv_" in r["Kernel_Name"]]
print("avg ns", sum(d)/len(d))
PY
done
