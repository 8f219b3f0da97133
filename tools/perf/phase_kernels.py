"""Kernels of one graph replay grouped by the phases of GanTrainer._step: rocprofv3 --kernel-trace of tools/perf/phase_stamps.py, whose
stamp_kernel launches delimit the phases on the main stream.   python phase_kernels.py <kernel_trace.csv>
Per phase: span, kernels, summed durations by class, and the ten largest kernels (name, launches, us)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("stamp_kernel")]
NAMES = ["start", "G forward", "D(fake) forward", "D backward (+ D(real) backward on the side stream)", "join", "D flush + Adam", "G step: D forward",
         "G step: backward through D and G", "G flush + Adam", "end"]
n = len(NAMES)
last = st[-n:]
def short(x):
    x = x.split("(")[0].replace("void ", "")
    return x[:64]
def klass(nm):
    for k, v in (("conv_halo", "conv"), ("conv_igemm", "conv"), ("conv_wstat", "conv"), ("conv_wgrad", "wgrad"), ("wgrad_reduce", "wgrad_reduce"), ("sn_", "spectral norm"),
                 ("norm_", "norm"), ("channel_stats", "norm"), ("ws_fold", "norm"), ("adam", "adam"), ("cast_kernel", "cast"), ("at::native", "aten"), ("rocclr", "rocclr")):
        if k in nm: return v
    return "other own"
for k in range(n - 1):
    a, b = last[k], last[k + 1]
    t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["Start_Timestamp"])
    seg = [r for r in rows[a + 1:b]]
    cls = collections.Counter(); cnt = collections.Counter(); byk = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        c = klass(r["Kernel_Name"]); cls[c] += d; cnt[c] += 1
        e = byk[short(r["Kernel_Name"])]; e[0] += 1; e[1] += d
    print(f"== {NAMES[k]}: span {(t1 - t0) / 1e3:.0f} us, {len(seg)} kernels (both streams), summed {sum(cls.values()) / 1e3:.0f} us")
    print("   " + "; ".join(f"{c} {d / 1e3:.0f} x{cnt[c]}" for c, d in cls.most_common()))
    for nm, (c, d) in sorted(byk.items(), key=lambda kv: -kv[1][1])[:10]:
        print(f"      {d / 1e3:8.1f} us  x{c:3d}  {nm}")
