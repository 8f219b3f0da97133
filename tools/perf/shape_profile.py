"""In-situ per-layer kernel durations: run a few eager iterations (one stream) under
    rocprofv3 --kernel-trace -d <dir> -o trace --output-format csv -- python tools/perf/shape_profile.py run <calls.json>
then join the conv kernels of the trace with the logged calls, in launch order:
    python tools/perf/shape_profile.py join <calls.json> <trace_kernel_trace.csv> [steps]
Durations are the profiler's (kernel begin -> end), FLOPs are algorithmic (live ROI rows only)."""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run(path, steps=3):
    import torch
    os.environ["L2I_OVERLAP"] = "0"
    import layout2img_amd as L
    from layout2img_amd import ops, _lib
    from layout2img_amd.synthetic import make_batch
    dev = torch.device("cuda:0")
    if "L2I_CONV_CFG" in os.environ:   # force one tile configuration on every halo-eligible layer (tuning)
        for v in os.environ["L2I_CONV_CFG"].split(","):
            _lib.call("l2i_set_conv_config", int(v))
    if "L2I_WGRAD_BLOCKS" in os.environ:
        for v in os.environ["L2I_WGRAD_BLOCKS"].split(","):
            _lib.call("l2i_set_wgrad_blocks", int(v))
    torch.manual_seed(1234)
    netG = L.ResnetGenerator128_context(num_classes=184).finalize(dev, torch.bfloat16)
    netD = L.CombineDiscriminator128_app(num_classes=184).finalize(dev, torch.bfloat16)
    tr = L.GanTrainer(netG, netD)
    real, label, bbox, z, z_im = make_batch(32, 128, "coco", seed=1234, device=dev)
    live = float((label != 0).sum()) / label.numel()
    gfwd = os.environ.get("L2I_SHAPES_GFWD", "0") == "1"   # the generator forward alone (train-mode statistics, no autograd tape)
    def one():
        if gfwd:
            with torch.no_grad():
                netG(z, bbox, z_im=z_im, y=label)
        else:
            tr.step(real, label, bbox, z, None)
    for _ in range(2):
        one()
    torch.cuda.synchronize()
    calls = []
    orig = _lib.call

    def call(name, *a):
        if name == "l2i_conv2d_fwd":
            calls.append(("fwd",) + tuple(a[9:20]) + (a[22] is not None,))
        elif name in ("l2i_conv2d_fwd_sc", "l2i_conv2d_fwd_dual"):   # 3x3 conv with a block's 1x1 shortcut handed over (folded, or un-folded by the library)
            calls.append(("fwd",) + tuple(a[9:20]) + ((a[22] is not None, int(a[31])) if a[25] is not None else (a[22] is not None,)))
        elif name == "l2i_conv2d_dgrad_sc":   # conv1's data gradient of a D block with the shortcut's data gradient folded in (or un-folded by the library)
            calls.append(("fwd", a[6], a[7], a[8], a[9], a[7], a[8], a[10], 3, 0, 0, 0, a[13] is not None, int(a[19])))
        elif name == "l2i_conv2d_wgrad":
            calls.append(("wgr",) + tuple(a[4:14]) + (0, a[16] is not None))
        elif name in ("l2i_conv2d_wgrad_sc", "l2i_conv2d_wgrad_dual"):   # conv2's weight gradient carrying the shortcut's (one launch; assumed folded)
            calls.append(("wgr",) + tuple(a[4:14]) + ((0, a[16] is not None, -int(a[22])) if a[20] is not None else (0, a[16] is not None)))
        orig(name, *a)
    _lib.call = call
    ops._lib.call = call
    mark = torch.zeros(1, device=dev)
    mark.fill_(1.0)   # (the join skips everything before the LAST `steps` iterations by counting conv kernels from the end)
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    json.dump(dict(calls=calls, steps=steps, live=live), open(path, "w"))


def join(path, trace, out=None):
    meta = json.load(open(path))
    calls, steps, live = meta["calls"], meta["steps"], meta["live"]
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    is_red = lambda r: "conv_wstat_reduce" in r["Kernel_Name"] or "conv_split_reduce" in r["Kernel_Name"]
    conv = [r for r in rows if r["Kernel_Name"].startswith(("void conv_", "conv_")) and not is_red(r)]
    # (the reduce launch of the small-map kernel / of a split-K launch follows its main kernel: its time is added to it)
    red = {}
    allc = [r for r in rows if r["Kernel_Name"].startswith(("void conv_", "conv_"))]
    for a_, b_ in zip(allc, allc[1:]):
        if is_red(b_):
            red[id(a_)] = int(b_["End_Timestamp"]) - int(b_["Start_Timestamp"])
    # a hand-over call is ONE kernel when the shortcut was folded, TWO (generic 1x1, then the 3x3) when the library un-folded it
    def pair(i, k):
        return len(calls[i]) == 14 and "conv_igemm" in conv[k]["Kernel_Name"] and k + 1 < len(conv) and "conv_halo" in conv[k + 1]["Kernel_Name"]
    n_sc = sum(1 for c in calls if len(c) == 14 and c[0] == "fwd")
    for extra in range(n_sc + 1):   # find the alignment from the end: the trace has len(calls) + (number of un-folded) conv kernels
        cand = conv[-(len(calls) + extra):]
        k, ok = 0, True
        for i in range(len(calls)):
            k += 2 if (len(calls[i]) == 14 and calls[i][0] == "fwd" and "conv_igemm" in cand[k]["Kernel_Name"] and k + 1 < len(cand)
                       and "conv_halo" in cand[k + 1]["Kernel_Name"]) else 1
            if k > len(cand):
                ok = False
                break
        if ok and k == len(cand):
            conv = cand
            break
    else:
        raise AssertionError((len(conv), len(calls)))
    agg = collections.OrderedDict()
    k = 0
    for i, c in enumerate(calls):
        r = r0 = conv[k]
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) + red.get(id(r), 0)) / 1e3
        k += 1
        sc_ci = 0
        if len(c) == 14:
            sc_ci = abs(c[13])
            if c[0] == "fwd" and "conv_igemm" in r["Kernel_Name"] and k < len(conv) and "conv_halo" in conv[k]["Kernel_Name"]:
                r = conv[k]   # un-folded: the 1x1 launch, then the 3x3 -- both belong to this call
                us += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) + red.get(id(r), 0)) / 1e3
                k += 1
                sc_ci = -sc_ci
            c = c[:13]
        kind, B, Hi, Wi, Ci, Ho, Wo, Co, KH, up2, pool2, relu, limited = c
        kern = r["Kernel_Name"].split("(")[0].replace("void ", "") + (f" +sc{sc_ci}" if sc_ci else "") + (" +split-reduce" if ((id(r0) in red or id(r) in red) and "halo" in r["Kernel_Name"]) else "")
        d = agg.setdefault((tuple(c), kern), [0, 0.0])
        d[0] += 1
        d[1] += us
    lines = []
    tot = {"fwd": [0.0, 0.0], "wgr": [0.0, 0.0]}
    for (c, kern), (n, us) in agg.items():
        kind, B, Hi, Wi, Ci, Ho, Wo, Co, KH, up2, pool2, relu, limited = c
        ci = 3 if Ci == 8 and Hi >= 64 and kind == "fwd" and Co == 64 else Ci   # (the image is padded 3 -> 8 channels)
        fl = 2.0 * B * Ho * Wo * Co * ci * KH * KH * (live if limited else 1.0)
        if "+sc" in kern:
            fl += 2.0 * B * Ho * Wo * Co * abs(int(kern.split("+sc")[1].split()[0])) * (live if limited else 1.0)
        tot[kind][0] += fl * n / steps
        tot[kind][1] += us / steps
        lines.append((us / steps, n // steps, fl * n / us / 1e6, c, kern))
    lines.sort(key=lambda t: -t[0])
    o = open(out, "w") if out else sys.stdout
    for k in ("fwd", "wgr"):
        if tot[k][1] == 0:
            continue
        print(f"{k}: {tot[k][1] / 1e3:.3f} ms/iteration, {tot[k][0] / 1e9:.1f} GFLOP/iteration, {tot[k][0] / tot[k][1] / 1e6:.1f} TFLOP/s "
              f"= {tot[k][0] / tot[k][1] / 1e6 / 2500:.3f} of the dense bf16 MFMA peak", file=o)
    print("  us/iter  launches  TFLOP/s   us at 70 % of the MFMA peak   (kind, B, Hi, Wi, Ci, Ho, Wo, Co, KH, up2, pool2, relu, roi_limited)  kernel", file=o)
    t70 = 0.0
    for us, n, tf, c, kern in lines:
        at70 = us * tf / (0.7 * 2500.0)   # the same FLOPs at 1750 TFLOP/s
        t70 += at70
        print(f"{us:9.1f}  x{n:3d}  {tf:8.1f}  {at70:8.1f}   {c}  {kern}", file=o)
    print(f"sum of the conv launches: {sum(l[0] for l in lines):.1f} us per iteration measured, {t70:.1f} us at 70 % of the MFMA peak", file=o)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
    else:
        join(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
