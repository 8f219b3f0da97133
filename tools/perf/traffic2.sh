#!/bin/bash
# usage (GPU box): bash tools/perf/traffic2.sh [tag=r03]  -> gpurun_out/<tag>_conv_traffic.json, <tag>_hbm_kernels.json,
#                                                            <tag>_bench_kernel_stats.csv, <tag>_mfma_util.txt
# Four passes of the SAME command: (1) --kernel-trace --stats (durations), (2) --pmc FETCH_SIZE, (3) --pmc WRITE_SIZE,
# (4) --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE (MFMA utilisation per kernel, in situ)
# (PMC passes carry --kernel-trace only, as the guide / gpurun require). One stream (L2I_OVERLAP=0): a kernel's
# duration and counters are those of a kernel that owns the GPU.
TAG=${1:-r05}
export TAG
cd /tmp && export TMPDIR=/tmp
export L2I_OVERLAP=0
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-g-forward --no-f32-mode"
rm -rf /tmp/tr_stats /tmp/tr_FETCH_SIZE /tmp/tr_WRITE_SIZE
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_stats -o t -- $CMD > /tmp/tr_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- $CMD > /tmp/tr_$c.log 2>&1
done
rm -rf /tmp/tr_mfma
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/tr_mfma -o t -- $CMD > /tmp/tr_mfma.log 2>&1
cp $(find /tmp/tr_stats -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_bench_kernel_stats.csv
python - <<'PY'
import csv, glob, json, collections, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
TAG = os.environ.get("TAG", "r03")
def short(n):
    n = n.split("(")[0].replace("void ", "")
    return n.split("<")[0] if n.startswith(("conv_", "channel_stats", "cast_kernel", "norm_mod", "sn_")) else n.replace(", ", ",")
stats = {}
for r in csv.DictReader(open(glob.glob("/tmp/tr_stats/**/*kernel_stats.csv", recursive=True)[0])):
    d = stats.setdefault(short(r["Name"]), [0, 0.0])
    d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"])
cnt = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(glob.glob(f"/tmp/tr_{c}/**/*counter_collection.csv", recursive=True)[0])):
        if r["Counter_Name"] == c:
            a = agg[short(r["Kernel_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    cnt[c] = agg
def group(pred, is_launch=lambda k: True):   # is_launch: kernels that are the second half of a launch add bytes and time, not launches
    n = sum(v[0] for k, v in cnt["FETCH_SIZE"].items() if pred(k) and is_launch(k))
    rd = sum(v[1] for k, v in cnt["FETCH_SIZE"].items() if pred(k)) * 1024 * 2   # gfx950: FETCH_SIZE counts half of wide coalesced reads (guide, HBM section)
    wr = sum(v[1] for k, v in cnt["WRITE_SIZE"].items() if pred(k)) * 1024
    calls = sum(v[0] for k, v in stats.items() if pred(k) and is_launch(k)); ns = sum(v[1] for k, v in stats.items() if pred(k))
    return dict(launches_profiled=n, hbm_read_bytes_per_launch=rd / max(n, 1), hbm_write_bytes_per_launch=wr / max(n, 1),
                traffic_bytes_per_launch=(rd + wr) / max(n, 1), avg_launch_us=ns / max(calls, 1) / 1e3)
res = {"conv(fwd+dgrad)": group(lambda k: k.startswith(("conv_halo", "conv_igemm", "conv_wstat")), lambda k: not k.startswith(("conv_wstat_reduce", "conv_split_reduce"))), "wgrad": group(lambda k: k.startswith("conv_wgrad")),
       "_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over python bench.py --steps 4 --warmup 1, L2I_OVERLAP=0; "
                "FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section); WRITE_SIZE uncalibrated; durations from a third pass without counters"}
json.dump(res, open(R + "/gpurun_out/" + TAG + "_conv_traffic.json", "w"), indent=1)
hbm = {}
for k in ("norm_mod_kernel", "norm_mod8_kernel", "norm_bwd_a_kernel", "norm_bwd_a8_kernel<16>", "norm_bwd_a8_kernel<32>", "norm_bwd_b_kernel", "channel_stats_kernel", "cast_kernel", "adam_kernel", "sn_pack_kernel", "sn_wtu_kernel",
          "sn_wv_kernel", "sn_apply_kernel", "sn_dot_kernel", "roi_align_kernel<false>", "roi_align_bwd_sep_kernel", "wgrad_reduce_kernel", "ws_fold_kernel", "gram_head_fwd_kernel", "gram_head_bwd_kernel"):
    g = group(lambda n, k=k: n == k)
    if g["launches_profiled"] and g["avg_launch_us"] > 0:
        gbs = g["traffic_bytes_per_launch"] / (g["avg_launch_us"] * 1e-6) / 1e9
        hbm[k] = dict(launches_per_profile=g["launches_profiled"], avg_launch_us=round(g["avg_launch_us"], 2), hbm_bytes_per_launch=round(g["traffic_bytes_per_launch"]),
                      gb_per_s=round(gbs, 1), frac_of_6300=round(gbs / 6300.0, 3))
hbm["_note"] = "HBM bytes (PMC, as above) / average kernel duration (rocprofv3 --stats pass); 6.3 TB/s = achievable HBM3E bandwidth (MI355X_MICROARCH.md)"
json.dump(hbm, open(R + "/gpurun_out/" + TAG + "_hbm_kernels.json", "w"), indent=1)
print(json.dumps(res, indent=1)); print(json.dumps(hbm, indent=1))
# ---- MFMA utilisation per kernel (in situ): counters summed over the chip per dispatch.
#   busy  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): share of the busy CU-cycles in which a SIMD's matrix pipe was occupied
#   issue = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 FLOP-per-MOP ... cross-check: MOPS x 512 / duration = FLOP/s against 2.5 PFLOP/s
f = glob.glob("/tmp/tr_mfma/**/*counter_collection.csv", recursive=True)
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES": nd[k] += 1
    lines = ["kernel | dispatches | avg us | MFMA busy / (4 x busy CU cycles) | MOPS_BF16 x 512 / duration (TFLOP/s, of 2500) | raw per dispatch: MFMA_BUSY, BUSY_CU, MOPS_BF16, GUI_ACTIVE"]
    rows = []
    for k, c in agg.items():
        if not k.startswith(("conv_halo", "conv_igemm", "conv_wgrad")) or k not in stats or not nd[k]: continue
        calls, ns = stats[k]
        us = ns / calls / 1e3
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(4.0 * c["SQ_BUSY_CU_CYCLES"], 1.0)
        tf = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] / nd[k] * 512.0 / (us * 1e-6) / 1e12
        rows.append((ns, f"{k} | {nd[k]} | {us:.1f} | {busy:.3f} | {tf:.0f} ({tf / 2500:.3f}) | {c['SQ_VALU_MFMA_BUSY_CYCLES'] / nd[k]:.3g}, {c['SQ_BUSY_CU_CYCLES'] / nd[k]:.3g}, {c['SQ_INSTS_VALU_MFMA_MOPS_BF16'] / nd[k]:.3g}, {c['GRBM_GUI_ACTIVE'] / nd[k]:.3g}"))
    for _, l in sorted(rows, reverse=True): lines.append(l)
    open(R + "/gpurun_out/" + TAG + "_mfma_util.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
else:
    print("no MFMA counter file", open("/tmp/tr_mfma.log").read()[-800:])
PY
