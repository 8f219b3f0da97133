"""Round-5 experiment: conv_halo8_kernel (256 x 256 tiles, 8 waves, one workgroup per CU) against the production tiles, back to back,
same process, interleaved rounds, random normal operands. Prints TFLOP/s (median of the rounds) and the max relative difference
to the default configuration's result."""
import sys, os, math, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from layout2img_amd import ops, _lib
dev = torch.device('cuda:0')
SHAPES = [(32, 32, 32, 512, 512, 0), (32, 32, 32, 256, 512, 0), (32, 16, 16, 512, 512, 0), (157, 8, 8, 1024, 1024, 0), (157, 4, 4, 1024, 1024, 1),
          (32, 64, 64, 128, 256, 0), (64, 32, 32, 512, 512, 0)]
CFGS = [int(v) for v in os.environ.get("CFGS", "-1,19,40,41").split(",")]
for B, H, W, Ci, Co, up2 in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
    K = 9 * Ci
    w = (torch.randn((Co + 127) // 128 * 128, K, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    bias = torch.randn(Co, generator=g).to(dev)
    Ho = H * (2 if up2 else 1)
    flops = 2.0 * B * Ho * Ho * Co * K
    res, times = {}, {c: [] for c in CFGS}
    for c in CFGS:
        _lib.call('l2i_set_conv_config', c)
        res[c] = ops.conv_raw(x, w, K, Co, 3, bias=bias, up2=bool(up2))[0].clone()
    torch.cuda.synchronize()
    for rnd in range(5):
        for c in CFGS:
            _lib.call('l2i_set_conv_config', c)
            for _ in range(3):
                ops.conv_raw(x, w, K, Co, 3, bias=bias, up2=bool(up2))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                ops.conv_raw(x, w, K, Co, 3, bias=bias, up2=bool(up2))
            e.record(); torch.cuda.synchronize()
            times[c].append(s.elapsed_time(e) / 20)
    _lib.call('l2i_set_conv_config', -1)
    ref = res[CFGS[0]]
    print(f"({B},{H},{W},{Ci}->{Co},up{up2})  " + "  ".join(
        f"cfg{c}: {flops / statistics.median(times[c]) / 1e9:7.1f} TF/s ({statistics.median(times[c]) * 1e3:6.1f} us, diff {float((res[c] - ref).abs().max()) / float(ref.abs().max()):.1e})"
        for c in CFGS), flush=True)
