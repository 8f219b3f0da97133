#!/bin/bash
cd /tmp && export TMPDIR=/tmp
shape="$1"
rm -rf /tmp/pmc2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc2 -o b -- python $GRAFT_REPO_ROOT/tools/perf/wgrad_micro.py $shape 2>&1 | grep shape
python - <<PY
import csv, collections, glob
fs = glob.glob("/tmp/pmc2/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(fs[0])) if "wgrad" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
a = {k: sum(v)/len(v) for k, v in agg.items()}
print({k: round(v) for k, v in a.items()})
m = a["SQ_INSTS_MFMA"]
print("per MFMA: valu %.2f salu %.2f lds %.2f vmem %.2f" % (a["SQ_INSTS_VALU"]/m, a["SQ_INSTS_SALU"]/m, a["SQ_INSTS_LDS"]/m, a["SQ_INSTS_VMEM"]/m))
PY
