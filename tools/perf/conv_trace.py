"""Where a conv launch spends its time, per wave: timestamps written by the kernels of a -DL2I_TRACE build
(L2I_EXTRA_FLAGS=-DL2I_TRACE python -m layout2img_amd.build --force): kernel entry, before / after the reduction loop, end.
usage: conv_trace.py B H W Ci Co KH up2 pool2 mode   (mode bits: 1 residual, 2 relu mask, 4 bf16 raw copy, 8 stats, 16 no f32 out)"""
import ctypes, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from layout2img_amd import ops, _lib
lib = _lib.load()
_lib.call('l2i_set_conv_config', int(os.environ.get('L2I_CFG', '-1')))
dev = torch.device('cuda:0')
B, H, W, Ci, Co, KH, up2, pool2, mode = [int(v) for v in sys.argv[1:10]]
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
K = KH * KH * Ci
kpad = (K + 63) // 64 * 64
npad = (Co + 127) // 128 * 128
w = (torch.randn(npad, kpad, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
Ho = H * (2 if up2 else 1)
Hq = Ho // 2 if pool2 else Ho
res = torch.randn(B, Hq, Hq, Co, device=dev) if mode & 1 else None
mask = torch.randn(B, Hq, Hq, Co, device=dev).to(torch.bfloat16) if mode & 2 else None
kw = dict(up2=bool(up2), pool2=bool(pool2), alpha=0.25 if pool2 else 1.0, res=res, relu_mask=mask, want_raw=bool(mode & 4), stats=bool(mode & 8),
          want_f32=not (mode & 16), want_op=bool(mode & 16), relu_op=True)
with ops.POOL.step(dev):
    for _ in range(5):
        ops.conv_raw(x, w, kpad, Co, KH, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.conv_raw(x, w, kpad, Co, KH, **kw)
    e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 100
nw = 8192 * 8
buf = np.zeros(nw * 4, dtype=np.int64)
lib.l2i_trace_read_conv.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.l2i_trace_read_conv(buf.ctypes.data, nw)
ids = np.zeros(nw, dtype=np.uint32)
lib.l2i_trace_ids_conv.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.l2i_trace_ids_conv(ids.ctypes.data, nw)
t = buf.reshape(nw, 4)
keep = t[:, 0] > 0
t, ids = t[keep], ids[keep]
t0 = t[:, 0].min()
r = (t - t0) / 100.0   # us
print(f"shape {sys.argv[1:10]}: {us:.1f} us per launch, {2.0 * B * Ho * Ho * Co * K / us / 1e6:.0f} TF/s, {len(t)} waves traced (last launch)")
for name, col in (("entry", 0), ("loop start", 1), ("loop end", 2), ("kernel end", 3)):
    c = r[:, col]
    print(f"  {name:11s} min {c.min():8.2f}  median {np.median(c):8.2f}  max {c.max():8.2f} us")
for name, a, b in (("prologue", 0, 1), ("loop", 1, 2), ("epilogue", 2, 3)):
    d = r[:, b] - r[:, a]
    print(f"  {name:11s} min {d.min():8.2f}  median {np.median(d):8.2f}  max {d.max():8.2f} us")
xcc = (ids >> 16) & 15
hw = ids & 0xffff
cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)   # CU_ID | SH_ID | SE_ID
simd = (hw >> 4) & 3
loop = r[:, 2] - r[:, 1]
end = r[:, 3]
print("  per XCC: waves, loop median, kernel-end max")
for x in sorted(set(xcc.tolist())):
    sel = xcc == x
    print(f"    xcc {x}: {int(sel.sum()):5d} waves  loop {np.median(loop[sel]):7.2f}  end max {end[sel].max():7.2f}  distinct CUs {len(set(cu[sel].tolist()))}")
# waves per (xcc, cu)
from collections import Counter
c = Counter(zip(xcc.tolist(), cu.tolist()))
print("  waves per CU histogram:", sorted(Counter(c.values()).items()))
# the two workgroups of a CU: first vs second finisher
key = xcc.astype(np.int64) * 1024 + cu
wg = (np.arange(len(t)) // (len(t) // max(1, len(t) // 4) if False else 1))
