import sys, math, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layout2img_amd import ops, _lib
dev = torch.device('cuda:0')
B, H, W, Ci, Co, KH, up2, pool2 = [int(v) for v in sys.argv[1:9]]
n = int(sys.argv[9]) if len(sys.argv) > 9 else 10
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
print(f"shape {sys.argv[1:9]} {us:.1f} us  {2.0 * B * Ho * Ho * Co * kp / us / 1e6:.1f} TF/s")
