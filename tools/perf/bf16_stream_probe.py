"""Round-5 measurement: what would bf16 result streams buy on the >= 64-px layers?  The conv side is measured directly:
the same launch writing its f32 result (+ statistics) against writing ONLY the bf16 copy (want_f32=False, want_raw=True), back to back."""
import sys, os, math, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from layout2img_amd import ops
dev = torch.device('cuda:0')
SHAPES = [(32, 64, 64, 128, 64, 1, "res5 conv1 64->128 px"), (32, 128, 128, 64, 64, 0, "res5 conv2 128 px"), (32, 32, 32, 256, 128, 1, "res4 conv1 32->64 px"),
          (32, 64, 64, 128, 128, 0, "res4 conv2 64 px"), (32, 128, 128, 64, 64, 0, "D block1 conv2 (pool)")]
for B, H, W, Ci, Co, up2, name in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
    K = 9 * Ci
    w = (torch.randn((Co + 127) // 128 * 128, K, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    bias = torch.randn(Co, generator=g).to(dev)
    variants = {"f32 + stats": dict(want_f32=True, stats=True), "f32 + raw copy": dict(want_f32=True, want_raw=True), "bf16 only": dict(want_f32=False, want_raw=True)}
    times = {k: [] for k in variants}
    for rnd in range(5):
        for k, kw in variants.items():
            for _ in range(3):
                ops.conv_raw(x, w, K, Co, 3, bias=bias, up2=bool(up2), **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                ops.conv_raw(x, w, K, Co, 3, bias=bias, up2=bool(up2), **kw)
            e.record(); torch.cuda.synchronize()
            times[k].append(s.elapsed_time(e) / 20 * 1e3)
    Ho = H * (2 if up2 else 1)
    print(f"{name:26s} out {B * Ho * Ho * Co * 4 / 1e6:6.1f} MB f32  " + "  ".join(f"{k}: {statistics.median(v):6.1f} us" for k, v in times.items()), flush=True)
