import sys, math, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layout2img_amd import ops, _lib
dev = torch.device('cuda:0')
def run(B, H, W, Ci, Co, KH, up2, n=20):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
    K = KH * KH * Ci; kpad = (K + 63) // 64 * 64; npad = (Co + 127) // 128 * 128
    w = (torch.randn(npad, kpad, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    for _ in range(3): ops.conv_raw(x, w, kpad, Co, KH, up2=bool(up2))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): ops.conv_raw(x, w, kpad, Co, KH, up2=bool(up2))
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
shapes = [(32,4,4,1024,1024,3,0), (32,4,4,1024,1024,3,1), (32,8,8,1024,1024,3,0), (32,8,8,512,512,3,1), (32,16,16,512,512,3,0), (32,16,16,512,256,3,0), (256,1,1,1024,1024,1,0), (32,8,8,1024,512,3,0)]
for sh in shapes:
    out = []
    for tgt in (128, 256, 384, 512, 768):
        _lib.call("l2i_set_conv_config", 1000 + tgt)
        out.append(f"{tgt}:{run(*sh):6.1f}us")
    print(sh, " ".join(out))
