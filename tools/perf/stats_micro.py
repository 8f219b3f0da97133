import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layout2img_amd import ops, _lib
dev = torch.device('cuda:0')
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (B, H, C) in [(32, 128, 64), (32, 64, 128), (32, 64, 64), (32, 32, 256), (32, 16, 512)]:
    x = torch.randn(B * H * H, C, device=dev)
    sums = torch.zeros(2, C, device=dev)
    raw = torch.empty(B * H * H, C, device=dev, dtype=torch.bfloat16)
    out = []
    for ws in (ops._ws(dev), None):
        for r in (None, raw):
            f = lambda: _lib.call("l2i_channel_stats", x.data_ptr(), x.shape[0], C, x.shape[0], sums[0].data_ptr(), sums[1].data_ptr(), ops._p(r), 1, *_lib.wgrad_scratch(x.device), ops._stream())
            out.append(t(f))
    print(f"H{H} C{C}", [round(v) for v in out])
