"""Per-wave timestamps of one weight-gradient launch (a -DL2I_TRACE build; see tools/perf/conv_trace.py).
usage: wgrad_trace.py B H W Ci Co KH up2 pool2"""
import ctypes, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from layout2img_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda:0')
B, H, W, Ci, Co, KH, up2, pool2 = [int(v) for v in sys.argv[1:9]]
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
Ho = H * (2 if up2 else 1)
Hd = Ho // 2 if pool2 else Ho
dy = torch.randn(B, Hd, Hd, Co, generator=g).to(dev, torch.bfloat16)
K = KH * KH * Ci
dw = torch.zeros(Co, K, device=dev)
for _ in range(5):
    ops.wgrad_raw(x, dy, dw, K, Co, KH, up2=bool(up2), pool2=bool(pool2))
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    ops.wgrad_raw(x, dy, dw, K, Co, KH, up2=bool(up2), pool2=bool(pool2))
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 100
nw = 8192 * 8
buf = np.zeros(nw * 4, dtype=np.int64)
lib.l2i_trace_read_wgrad.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.l2i_trace_read_wgrad(buf.ctypes.data, nw)
t = buf.reshape(nw, 4)
t = t[(t[:, 0] > 0) & (t[:, 3] > 0)]
t0 = t[:, 0].min()
r = (t - t0) / 100.0
print(f"wgrad {sys.argv[1:9]}: {us:.1f} us per launch (main + reduce), {2.0 * B * Ho * Ho * Co * K / us / 1e6:.0f} TF/s, {len(t)} waves traced")
for name, col in (("entry", 0), ("loop start", 1), ("loop end", 2), ("kernel end", 3)):
    c = r[:, col]
    print(f"  {name:11s} min {c.min():8.2f}  median {np.median(c):8.2f}  max {c.max():8.2f} us")
for name, a, b in (("prologue", 0, 1), ("loop", 1, 2), ("epilogue", 2, 3)):
    d = r[:, b] - r[:, a]
    print(f"  {name:11s} min {d.min():8.2f}  median {np.median(d):8.2f}  max {d.max():8.2f} us")
