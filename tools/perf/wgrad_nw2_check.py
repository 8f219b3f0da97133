"""Two-wave weight-gradient kernel against the four-wave one: same results (up to f32 atomics order), and isolated timing."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layout2img_amd import ops, _lib
dev = torch.device('cuda:0')


def run(B, H, W, Ci, Co, KH, up2, pool2, nimg=None, bias=False, n=10, time=True):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
    Ho = 2 * H if up2 else H
    Hd = Ho // 2 if pool2 else Ho
    dy = torch.randn(B, Hd, Hd, Co, generator=g).to(dev, torch.bfloat16)
    kp = KH * KH * Ci
    ni = torch.tensor([nimg], dtype=torch.int32, device=dev) if nimg is not None else None
    res = []
    for mode in (-4, -2):
        _lib.call("l2i_set_wgrad_blocks", mode)
        dw = torch.zeros(Co, kp, device=dev)
        db = torch.zeros(Co, device=dev) if bias else None
        ops.wgrad_raw(x, dy, dw, kp, Co, KH, up2=bool(up2), pool2=bool(pool2), nimg=ni, dbias=db, alpha=0.25 if pool2 else 1.0)
        torch.cuda.synchronize()
        us = 0.0
        if time:
            d2 = torch.zeros(Co, kp, device=dev)
            for _ in range(3):
                ops.wgrad_raw(x, dy, d2, kp, Co, KH, up2=bool(up2), pool2=bool(pool2), nimg=ni)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n):
                ops.wgrad_raw(x, dy, d2, kp, Co, KH, up2=bool(up2), pool2=bool(pool2), nimg=ni)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / n * 1e3
        res.append((dw, db, us))
    (a, ab, ua), (b, bb, ub) = res
    err = float((a - b).abs().max()) / (float(a.abs().max()) + 1e-9)
    berr = float((ab - bb).abs().max()) / (float(ab.abs().max()) + 1e-9) if bias else 0.0
    fl = 2.0 * B * Ho * Ho * Co * kp * ((nimg / B) if nimg is not None else 1.0)
    print(f"{(B,H,W,Ci,Co,KH,up2,pool2,nimg,bias)}: rel diff {err:.2e} bias {berr:.2e}   nw4 {ua:7.1f}us {fl/ua/1e6 if ua else 0:5.0f}TF   nw2 {ub:7.1f}us {fl/ub/1e6 if ub else 0:5.0f}TF")
    assert err < 1e-5 and berr < 1e-5


for sh in [(2, 16, 16, 128, 128, 3, 0, 0, None, True), (3, 8, 8, 256, 136, 3, 0, 1, None, True), (2, 8, 8, 128, 256, 3, 1, 0, None, True),
           (20, 8, 8, 128, 128, 3, 0, 0, 7, True), (20, 8, 8, 136, 128, 1, 0, 1, 13, True), (1, 4, 4, 128, 200, 3, 0, 0, None, True),
           (3, 6, 6, 128, 128, 3, 0, 0, None, False)]:
    run(*sh, time=False)
shapes = [(32,64,64,128,128,3,0,0), (32,32,32,256,256,3,0,0), (32,32,32,128,256,3,0,0), (32,16,16,512,512,3,0,0), (32,16,16,256,512,3,0,0),
          (32,32,32,512,512,3,0,0), (256,8,8,512,512,3,0,0), (256,8,8,512,1024,3,0,0), (256,8,8,1024,1024,3,0,1), (32,8,8,1024,1024,3,0,1),
          (32,4,4,1024,1024,3,0,0), (32,32,32,256,512,3,0,0), (32,16,16,512,256,3,1,0)]
for sh in shapes:
    run(*sh)
