"""Every C-ABI call of one training iteration at the headline batch, in issue order, from a dry run (tests/dryrun.py: meta tensors, no GPU):
the convolution / weight-gradient launches with their dimensions, fused extras and dimensioned GFLOP, everything else by name.
usage: python tools/perf/launch_plan.py [coco|vg] [batch] > profiles/rNN_launch_plan.txt"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import dryrun
from layout2img_amd.synthetic import make_batch

kind = sys.argv[1] if len(sys.argv) > 1 else "coco"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
names = dryrun.header_parameters()
with dryrun.dry_run() as trace:
    tr, _ = dryrun.build(kind, torch.bfloat16)
    real, label, bbox, z, z_im = (t.to("meta") for t in make_batch(batch, 128, kind, seed=1234, device="cpu"))
    tr.overlap = False
    for _ in range(2):
        del trace[:]
        tr.step(real, label, bbox, None, z_im if kind == "vg" else None)
        tr.flush()
print(f"# {kind} layouts, batch {batch}, bf16 operands, one stream: {len(trace)} C-ABI calls per iteration (issue order)")
print(f"# {'#':>3s} {'entry point':26s} {'B':>4s} {'in HxWxC':>14s} {'out HxWxC':>14s} k  extras{'':34s} GFLOP (dimensioned, all ROI slots)")
tot = collections.Counter()
for i, (name, args) in enumerate(trace):
    a = dict(zip(names[name], args))
    if name in ("l2i_conv2d_fwd", "l2i_conv2d_fwd_dual", "l2i_conv2d_wgrad", "l2i_conv2d_wgrad_dual"):
        sc = a.get("sc_x") is not None
        f = 2.0 * a["B"] * a["Ho"] * a["Wo"] * a["Co"] * (a["KH"] ** 2 * a["Ci"] + (a["sc_Ci"] if sc else 0)) / 1e9
        ex = [k for k in ("up2", "pool2") if a[k]] + (["+1x1 shortcut %d" % a["sc_Ci"]] if sc else [])
        ex += ["two weight packs"] if a.get("w_b") is not None or a.get("dw_b") is not None else []
        ex += ["live ROIs only"] if a["nimg"] is not None else []
        ex += ["batch statistics"] if a.get("stats") is not None else []
        ex += ["bias gradient"] if a.get("dbias") is not None else []
        q = 2 if a["pool2"] else 1
        print(f"{i:5d} {name:26s} {a['B']:4d} {a['Hi']:4d}x{a['Wi']:<4d}x{a['Ci']:<4d} {a['Ho'] // q:4d}x{a['Wo'] // q:<4d}x{a['Co']:<4d} {a['KH']}  {', '.join(ex):40s} {f:9.2f}")
        tot["wgrad" if "wgrad" in name else "conv"] += f
    elif name == "l2i_conv2d_dgrad_sc":
        f = 2.0 * a["B"] * a["H"] * a["W"] * a["Co"] * (9 * a["Ci"] + a["sc_Ci"]) / 1e9
        print(f"{i:5d} {name:26s} {a['B']:4d} {a['H']:4d}x{a['W']:<4d}x{a['Ci']:<4d} {a['H']:4d}x{a['W']:<4d}x{a['Co']:<4d} 3  {'+1x1 shortcut data gradient %d, ReLU mask' % a['sc_Ci']:40s} {f:9.2f}")
        tot["conv"] += f
    else:
        print(f"{i:5d} {name}")
c = collections.Counter(n for n, _ in trace)
print(f"# dimensioned GFLOP per iteration: convolution forward + data gradient {tot['conv']:.0f}, weight gradient {tot['wgrad']:.0f}")
print("# calls by entry point: " + ", ".join(f"{n} x{k}" for n, k in c.most_common()))
