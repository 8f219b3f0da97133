import sys, math, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layout2img_amd import ops, _lib
_lib.call('l2i_set_conv_config', int(os.environ.get('L2I_CFG', '-1')))
dev = torch.device('cuda:0')
B, H, W, Ci, Co, KH = [int(v) for v in sys.argv[1:7]]
up2 = int(sys.argv[7]) if len(sys.argv) > 7 else 0
n = int(sys.argv[8]) if len(sys.argv) > 8 else 5
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
K = KH * KH * Ci
kpad = (K + 63) // 64 * 64
npad = (Co + 127) // 128 * 128
w = (torch.randn(npad, kpad, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
for _ in range(n):
    out, _, _ = ops.conv_raw(x, w, kpad, Co, KH, up2=bool(up2))
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(n):
    out, _, _ = ops.conv_raw(x, w, kpad, Co, KH, up2=bool(up2))
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / n
Ho = H * (2 if up2 else 1)
print(f"shape {sys.argv[1:8]} {ms*1e3:.1f} us  {2.0*B*Ho*Ho*Co*K/ms/1e9:.1f} TF/s")
