#!/bin/bash
cd /tmp && export TMPDIR=/tmp
shape="$1"
rm -rf /tmp/pmc1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1 -o a -- python $GRAFT_REPO_ROOT/tools/perf/wgrad_micro.py $shape 2>&1 | grep shape
python - <<PY
import csv, collections, glob
fs = glob.glob("/tmp/pmc1/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(fs[0])) if "wgrad" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
a = {k: sum(v)/len(v) for k, v in agg.items()}
print({k: round(v) for k, v in a.items()})
w = a["SQ_WAVE_CYCLES"]
print("wait_any %.2f wait_inst %.2f active %.2f  lds_conflict/idx %.3f" % (a["SQ_WAIT_ANY"]/w, a["SQ_WAIT_INST_ANY"]/w, a["SQ_ACTIVE_INST_ANY"]/w, a["SQ_LDS_BANK_CONFLICT"]/max(a["SQ_LDS_IDX_ACTIVE"],1)))
PY
