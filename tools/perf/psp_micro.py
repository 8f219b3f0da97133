"""Isolated timing of the PSP stage kernels (csrc/psp.hip) at the generator's shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layout2img_amd import ops
from layout2img_amd.generator import psp_taps
dev = torch.device("cuda:0")
B, H, C, Fo = 32, 64, 128, 100
sizes = (1, 2, 3, 6)
taps = psp_taps(H, sizes, dev)
feats = torch.randn(B, H, H, C, device=dev)
y = torch.randn(B, 50, Fo, device=dev)
dt = torch.bfloat16


def timeit(name, fn, n=20, nbytes=0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / n * 1e3
    print(f"{name:24s} {us:8.1f} us  {nbytes / us / 1e3 if nbytes else 0:7.1f} GB/s")


fg = feats.clone().requires_grad_(True)
yg = y.clone().requires_grad_(True)
timeit("pool fwd", lambda: ops.psp_pool(feats, taps), nbytes=feats.numel() * 4)
timeit("expand fwd", lambda: ops.psp_expand(feats, y, taps, dt), nbytes=feats.numel() * 4 + B * H * H * 528 * 2)
pooled = ops.psp_pool(fg, taps)
gp = torch.randn_like(pooled)
timeit("pool fwd+bwd", lambda: torch.autograd.grad(ops.psp_pool(fg, taps), fg, gp), nbytes=feats.numel() * 8)
cat = ops.psp_expand(fg, yg, taps, dt)
gc = torch.randn(cat.shape, device=dev).to(dt)
timeit("expand fwd+bwd", lambda: torch.autograd.grad(ops.psp_expand(fg, yg, taps, dt), (fg, yg), gc),
       nbytes=feats.numel() * 8 + B * H * H * 528 * 4)
