import sys, math, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layout2img_amd import ops, _lib
dev = torch.device('cuda:0')
def run(B, H, W, Ci, Co, KH, up2, pool2, n=10):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
    Ho = 2 * H if up2 else H
    Hd = Ho // 2 if pool2 else Ho
    dy = torch.randn(B, Hd, Hd, Co, generator=g).to(dev, torch.bfloat16)
    kp = KH * KH * Ci
    dw = torch.zeros(Co, kp, device=dev)
    for _ in range(3): ops.wgrad_raw(x, dy, dw, kp, Co, KH, up2=bool(up2), pool2=bool(pool2))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): ops.wgrad_raw(x, dy, dw, kp, Co, KH, up2=bool(up2), pool2=bool(pool2))
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / n * 1e3
    return us, 2.0 * B * Ho * Ho * Co * kp / us / 1e6
shapes = [(32,128,128,64,64,3,0,0), (32,64,64,128,128,3,0,0), (32,64,64,128,64,3,1,0), (32,32,32,256,256,3,0,0), (32,32,32,128,256,3,0,0), (32,16,16,512,512,3,0,0), (32,16,16,256,512,3,0,0),
          (32,32,32,512,512,3,0,0), (256,8,8,512,512,3,0,0), (256,8,8,1024,1024,3,0,1), (32,8,8,1024,1024,3,0,1), (32,4,4,1024,1024,3,0,0), (32,64,64,528,104,3,0,0), (32,128,128,64,8,3,0,0)]
for sh in shapes:
    out = []
    for tgt in (-128, -64):
        _lib.call("l2i_set_wgrad_blocks", tgt)
        us, tf = run(*sh)
        out.append(f"{tgt}:{us:6.1f}us/{tf:4.0f}TF")
    print(sh, " ".join(out))
