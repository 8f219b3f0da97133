"""Per-shape tile-configuration sweep over the 3x3 layers of the iteration (shapes read from an in-situ table written by
tools/perf/shape_profile.py). Prints TFLOP/s (live rows) per configuration; tuning only."""
import ast, math, os, re, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layout2img_amd import ops, _lib
dev = torch.device('cuda:0')
table, cfgs = sys.argv[1], [int(c) for c in sys.argv[2].split(",")]
live_roi = 157
shapes = []
for line in open(table):
    m = re.search(r"\('fwd'.*?\)", line)
    if not m:
        continue
    c = ast.literal_eval(m.group(0))
    if c[8] == 3 and c[4] % 64 == 0 and c not in shapes:
        shapes.append(c)
g = torch.Generator().manual_seed(0)
for c in shapes:
    _, B, H, W, Ci, Ho, Wo, Co, KH, up2, pool2, relu, lim = c
    x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
    K = KH * KH * Ci; kpad = (K + 63) // 64 * 64; npad = (Co + 127) // 128 * 128
    w = (torch.randn(npad, kpad, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    nimg = torch.tensor([live_roi], dtype=torch.int32, device=dev) if lim else None
    fl = 2.0 * (live_roi if lim else B) * Ho * Wo * Co * K
    res = []
    for cfg in cfgs:
        _lib.call("l2i_set_conv_config", cfg)
        kw = dict(up2=bool(up2), pool2=bool(pool2), alpha=0.25 if pool2 else 1.0, nimg=nimg)
        try:
            for _ in range(2): ops.conv_raw(x, w, kpad, Co, KH, **kw)
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(2):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(8): ops.conv_raw(x, w, kpad, Co, KH, **kw)
                e.record(); torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) / 8)
            res.append(f"c{cfg}:{fl / best / 1e9:5.0f}/{best * 1e3:5.0f}us")
        except RuntimeError as ex:
            res.append(f"c{cfg}: ERR")
    print(c[1:11], "roi" if lim else "   ", " ".join(res), flush=True)
_lib.call("l2i_set_conv_config", -1)
