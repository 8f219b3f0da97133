"""Small-map layers (4x4 maps, 4 -> 8 upsampling): generic kernel vs conv_halo3 compact with forced split counts (tuning)."""
import sys, math, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layout2img_amd import ops, _lib
dev = torch.device('cuda:0')
shapes = [(32, 4, 4, 1024, 1024, 0, 0, None), (32, 4, 4, 1024, 1024, 1, 0, None), (256, 4, 4, 1024, 1024, 1, 0, 157), (256, 4, 4, 256, 256, 0, 0, None),
          (32, 8, 8, 1024, 1024, 0, 1, None), (32, 8, 8, 512, 1024, 0, 0, None), (32, 8, 8, 1024, 512, 0, 0, None)]
g = torch.Generator().manual_seed(0)
for (B, H, W, Ci, Co, up2, pool2, live) in shapes:
    x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
    K = 9 * Ci; kpad = K; npad = (Co + 127) // 128 * 128
    w = (torch.randn(npad, kpad, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    nimg = torch.tensor([live], dtype=torch.int32, device=dev) if live else None
    Ho = H * (2 if up2 else 1)
    fl = 2.0 * (live or B) * Ho * Ho * Co * K
    res = []
    ref = None
    for cfg, sp in [(-1, 0), (19, 1), (19, 2), (19, 4), (19, 8), (19, 16), (19, 32), (29, 4), (29, 8), (29, 16), (29, 32)]:
        _lib.call("l2i_set_conv_config", 2000 + sp)
        _lib.call("l2i_set_conv_config", cfg)
        kw = dict(up2=bool(up2), pool2=bool(pool2), alpha=0.25 if pool2 else 1.0, nimg=nimg)
        for _ in range(2): out, _, _ = ops.conv_raw(x, w, kpad, Co, 3, **kw)
        torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        err = float((out - ref).abs().max())
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(8): ops.conv_raw(x, w, kpad, Co, 3, **kw)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 8
        res.append(f"c{cfg}/s{sp}:{fl / ms / 1e9:4.0f}TF {ms * 1e3:4.0f}us" + (f" ERR{err:.1e}" if err > 1e-2 else ""))
    print((B, H, W, Ci, Co, up2, pool2, live), " | ".join(res), flush=True)
_lib.call("l2i_set_conv_config", 2000)
_lib.call("l2i_set_conv_config", -1)
