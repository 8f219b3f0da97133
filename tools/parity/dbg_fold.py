import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from layout2img_amd import ops, _lib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_ops import _rt, _pack, _ref_conv
dev = "cuda:0"; dt = torch.bfloat16
for cfg in (-1, 14, 19, 29):
    B, H, W, Ci, Co, sCi, pooled = 2, 32, 32, 128, 64, 128, True
    g = torch.Generator().manual_seed(41)
    dh = _rt(torch.randn(B, H, W, Ci, generator=g), dt); w = _rt(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9), dt)
    dy = _rt(torch.randn(B, H // 2, W // 2, sCi, generator=g), dt); wsc = _rt(torch.randn(Co, sCi, 1, 1, generator=g) / math.sqrt(sCi), dt)
    mask = _rt(torch.randn(B, H, W, Co, generator=g), dt); res = torch.randn(B, H, W, Co, generator=g)
    pack, kpad = _pack(w, 64); pack_sc, kpad_sc = _pack(wsc, 64)
    ph = torch.full((B, H, W, Co), float("nan"), device=dev)
    sc = dict(x_op=dy.to(dev, dt), wpack=pack_sc.to(dev, dt), kpad=kpad_sc, bias=None, up2=pooled, alpha=0.25, out=ph, flops=0.0, mask_first=True)
    _lib.call("l2i_set_conv_config", cfg)
    out, _, raw = ops.conv_raw(dh.to(dev, dt), pack.to(dev, dt), kpad, Co, 3, relu_mask=mask.to(dev, dt), res=res.to(dev), sc=sc, want_raw=True)
    _lib.call("l2i_set_conv_config", -1)
    ref = _ref_conv(dh, w, None, False, False) * (mask > 0).float() + 0.25 * _ref_conv(dy, wsc, None, pooled, False) + res
    print(cfg, "err", float((out.cpu() - ref).abs().max()) / float(ref.abs().max()), "placeholder NaN fraction", float(torch.isnan(ph).float().mean()), "splits", _lib.load().l2i_debug_occupancy(100, 0))
