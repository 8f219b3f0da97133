"""Ablation / tile sweep of the halo conv kernel on a few layer shapes (tuning only; build with -DL2I_ABLATIONS)."""
import sys, math, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layout2img_amd import ops, _lib
dev = torch.device('cuda:0')
shapes = [(32, 32, 32, 512, 512, 3, 0, 0), (256, 8, 8, 512, 512, 3, 0, 0), (32, 64, 64, 128, 128, 3, 0, 0), (32, 16, 16, 512, 512, 3, 0, 0),
          (160, 8, 8, 1024, 1024, 3, 0, 1)]
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [14, 12, 10, 21, 22, 23, 24, 31, 32, 33]
g = torch.Generator().manual_seed(0)
for (B, H, W, Ci, Co, KH, up2, pool2) in shapes:
    x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
    K = KH * KH * Ci; kpad = (K + 63) // 64 * 64; npad = (Co + 127) // 128 * 128
    w = (torch.randn(npad, kpad, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    Ho = H * (2 if up2 else 1)
    res = []
    for cfg in cfgs:
        _lib.call("l2i_set_conv_config", cfg)
        kw = dict(up2=bool(up2), pool2=bool(pool2), alpha=0.25 if pool2 else 1.0)
        for _ in range(3): out, _, _ = ops.conv_raw(x, w, kpad, Co, KH, **kw)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): out, _, _ = ops.conv_raw(x, w, kpad, Co, KH, **kw)
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 10)
        res.append(f"c{cfg}:{2.0 * B * Ho * Ho * Co * K / best / 1e9:5.0f}")
    print((B, H, W, Ci, Co, KH, up2, pool2), " ".join(res), flush=True)
_lib.call("l2i_set_conv_config", -1)
