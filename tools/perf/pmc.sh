#!/bin/bash
# usage: pmc.sh "<shape args>"
cd /tmp && export TMPDIR=/tmp
shape="$1"
rm -rf /tmp/pmc1 /tmp/pmc2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1 -o a -- python $GRAFT_REPO_ROOT/tools/perf/conv_micro.py $shape 2>&1 | grep shape
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc2 -o b -- python $GRAFT_REPO_ROOT/tools/perf/conv_micro.py $shape 2>&1 | grep -c shape
python - <<PY
import csv, collections, glob
for d in ("/tmp/pmc1", "/tmp/pmc2"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no counter file in", d, glob.glob(d + "/**/*", recursive=True)[:5]); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "conv_" in r["Kernel_Name"]]
    agg = collections.defaultdict(list)
    for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print({k: round(sum(v)/len(v)) for k, v in agg.items()}, "dispatches", len(rows) // max(1, len(agg)))
    if rows: print({k: rows[0][k] for k in ("Grid_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count") if k in rows[0]})
PY
