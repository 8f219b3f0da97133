"""Same layer launched back to back on ONE buffer set (warm L2 / Infinity Cache) vs rotating over enough buffer sets to
exceed the 256 MB Infinity Cache (cold), vs interleaved with an unrelated streaming kernel (tuning only)."""
import sys, math, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layout2img_amd import ops, _lib
dev = torch.device('cuda:0')
shapes = [(32, 32, 32, 512, 512, 3, 0, 0), (256, 8, 8, 512, 512, 3, 0, 0), (32, 64, 64, 128, 128, 3, 0, 0), (32, 16, 16, 512, 512, 3, 0, 0)]
g = torch.Generator().manual_seed(0)
for (B, H, W, Ci, Co, KH, up2, pool2) in shapes:
    K = KH * KH * Ci; kpad = (K + 63) // 64 * 64; npad = (Co + 127) // 128 * 128
    nset = 12
    xs = [torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16) for _ in range(nset)]
    ws = [(torch.randn(npad, kpad, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16) for _ in range(nset)]
    junk = torch.empty(128 << 20, device=dev)   # 512 MB
    kw = dict(up2=bool(up2), pool2=bool(pool2), alpha=0.25 if pool2 else 1.0)
    fl = 2.0 * B * H * W * Co * K
    def run(mode, n=24):
        for i in range(3): ops.conv_raw(xs[0], ws[0], kpad, Co, KH, **kw)
        torch.cuda.synchronize()
        tot = 0.0
        evs = []
        for i in range(n):
            j = 0 if mode == "warm" else i % nset
            if mode == "flush":
                junk.add_(1.0)    # 1 GB of traffic: evicts L2 and the Infinity Cache
                j = 0
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); ops.conv_raw(xs[j], ws[j], kpad, Co, KH, **kw); e.record()
            evs.append((s, e))
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in evs)
        return fl / ts[len(ts) // 2] / 1e9
    print((B, H, W, Ci, Co), " ".join(f"{m}:{run(m):5.0f}" for m in ("warm", "rotate", "flush")), flush=True)
