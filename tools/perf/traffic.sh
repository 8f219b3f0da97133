#!/bin/bash
# HBM traffic of the conv kernels from PMC (separate passes, --kernel-trace only), averaged per launch.
cd /tmp && export TMPDIR=/tmp
export L2I_OVERLAP=0   # one stream: per-kernel durations are those of a kernel that owns the GPU
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-g-forward > /tmp/tr_$c.log 2>&1
done
python - <<'PY'
import csv, glob, json, collections, os
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/tr_{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0]
        k = "conv(fwd+dgrad)" if ("conv_halo2" in k or "conv_igemm" in k) else "wgrad" if "wgrad" in k else "other"
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    out[c] = {k: {"launches": v[0], "sum": v[1]} for k, v in agg.items()}
res = {}
for k in out["FETCH_SIZE"]:
    n = out["FETCH_SIZE"][k]["launches"]
    fetch_kb, write_kb = out["FETCH_SIZE"][k]["sum"], out["WRITE_SIZE"].get(k, {"sum": 0})["sum"]
    # guide: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads on gfx950 -> x2; unit KB
    res[k] = {"launches_profiled": n, "hbm_read_bytes_per_launch": 2 * fetch_kb * 1024 / n, "hbm_write_bytes_per_launch": write_kb * 1024 / n}
    res[k]["traffic_bytes_per_launch"] = res[k]["hbm_read_bytes_per_launch"] + res[k]["hbm_write_bytes_per_launch"]
res["_note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over python bench.py --steps 2 --warmup 1; FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section); WRITE_SIZE uncalibrated"
json.dump(res, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r01_conv_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
