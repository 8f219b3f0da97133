"""GPU parity tests of every HIP kernel (through the C ABI) against the oracle / plain torch fp32 on CPU.

Operands are pre-rounded to the operand dtype before the reference is evaluated, so the bf16 cases
compare accumulation order only and can use tight tolerances; model-level bf16 tolerances live in
tests/test_gpu_00_models.py.
"""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import model as O

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _dev():
    return torch.device("cuda:0")


def _rt(t, dt):
    """round-trip through the operand dtype"""
    return t.to(dt).float()


def _pack(w, kpad_mult):
    """(Co,Ci,KH,KH) -> [Npad][Kpad] with k = (ky,kx,ci)"""
    co, ci, kh, _ = w.shape
    k = kh * kh * ci
    kpad, npad = (k + kpad_mult - 1) // kpad_mult * kpad_mult, (co + 127) // 128 * 128
    p = torch.zeros(npad, kpad)
    p[:co, :k] = w.permute(0, 2, 3, 1).reshape(co, k)
    return p, kpad


CONV_CASES = [
    # B, H, W, Ci, Co, KH, up2, pool2
    (2, 8, 8, 16, 24, 3, False, False),
    (2, 4, 4, 64, 64, 3, True, False),
    (1, 16, 16, 32, 136, 3, False, True),
    (3, 32, 32, 8, 64, 3, False, False),
    (2, 16, 16, 40, 104, 1, False, False),
    (5, 1, 1, 312, 72, 1, False, False),
    (130, 1, 1, 64, 8, 1, False, False),
    (1, 64, 64, 16, 16, 3, True, False),
    (2, 8, 8, 16, 16, 3, False, True),
    (3, 8, 8, 16, 32, 1, False, True),
    (2, 32, 32, 8, 8, 3, True, True),
    # 3x3 with >= 64 input channels on maps >= 8 wide: the halo-resident kernel (bf16)
    (2, 16, 16, 64, 72, 3, False, False),
    (1, 32, 32, 128, 136, 3, False, True),
    (2, 16, 16, 64, 64, 3, True, False),
    (3, 8, 8, 136, 128, 3, False, False),
    (1, 64, 64, 64, 8, 3, False, False),
    (9, 16, 16, 192, 128, 3, False, False),
    (2, 32, 32, 72, 64, 3, True, True),
    (5, 8, 8, 64, 264, 3, False, True),
]


@pytest.fixture(params=[0, 2], ids=["epi_direct", "epi_lds"])
def epi(request):
    """Both forms of the convolution epilogue (tuning hook 4000 + m: 0 direct stores from the accumulator layout, 2 the
    coalesced form through LDS, the default)."""
    from layout2img_amd import _lib
    _lib.call("l2i_set_conv_config", 4000 + request.param)
    yield request.param
    _lib.call("l2i_set_conv_config", 4009)


def _ref_conv(x_nhwc, w, bias, up2, pool2):
    x = x_nhwc.permute(0, 3, 1, 2)
    if up2:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    y = F.conv2d(x, w, bias, 1, w.shape[2] // 2)
    if pool2:
        y = F.avg_pool2d(y, 2)
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward(case, dt, epi):
    from layout2img_amd import ops
    B, H, W, Ci, Co, KH, up2, pool2 = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    w = _rt(torch.randn(Co, Ci, KH, KH, generator=g) / math.sqrt(Ci * KH * KH), dt)
    bias = torch.randn(Co, generator=g)
    ref = _ref_conv(x, w, bias, up2, pool2)
    res = torch.randn(ref.shape, generator=g)
    pack, kpad = _pack(w, 64 if dt == torch.bfloat16 else 32)
    out, out_op, out_raw = ops.conv_raw(x.to(_dev(), dt), pack.to(_dev(), dt), kpad, Co, KH, bias=bias.to(_dev()), res=res.to(_dev()),
                                        up2=up2, pool2=pool2, alpha=0.25 if pool2 else 1.0, want_op=True, relu_op=True, want_raw=True)
    # the kernel's pool2 sums a quad then scales by alpha; bias is added after -> same as avg_pool(conv)+bias
    expect = ref + res
    scale = float(expect.abs().max())
    assert float((out.cpu() - expect).abs().max()) < 2e-5 * scale + 1e-5
    assert float((out_raw.float().cpu() - _rt(out.cpu(), dt)).abs().max()) == 0.0
    assert float((out_op.float().cpu() - _rt(out.cpu().clamp_min(0), dt)).abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", [(2, 16, 16, 64, 72, 3, False, False), (1, 32, 32, 128, 136, 3, False, True),
                                  (9, 16, 16, 192, 128, 3, False, False), (2, 16, 16, 64, 64, 3, True, False),
                                  (3, 8, 8, 136, 256, 3, False, False), (2, 16, 16, 40, 104, 1, False, False)])
def test_conv_epilogue_statistics(case, dt):
    """`stats`: the epilogue gathers per-channel sum / sum of squares of the f32 result (the batch statistics of the
    normalisation that reads it, model/norm_module.py:163) -- equal to a pass over the result"""
    from layout2img_amd import ops
    B, H, W, Ci, Co, KH, up2, pool2 = case
    g = torch.Generator().manual_seed(23)
    x = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    w = _rt(torch.randn(Co, Ci, KH, KH, generator=g) / math.sqrt(Ci * KH * KH), dt)
    bias = torch.randn(Co, generator=g)
    ref = _ref_conv(x, w, bias, up2, pool2)
    res = torch.randn(ref.shape, generator=g)
    pack, kpad = _pack(w, 64 if dt == torch.bfloat16 else 32)
    out, _, _ = ops.conv_raw(x.to(_dev(), dt), pack.to(_dev(), dt), kpad, Co, KH, bias=bias.to(_dev()), res=res.to(_dev()),
                             up2=up2, pool2=pool2, alpha=0.25 if pool2 else 1.0, stats=True)
    sums, sq, ver = out._l2i_stats
    assert ver == out._version and sums.shape == (1, Co)
    o = out.double().cpu().view(-1, Co)
    assert float((out.cpu() - (ref + res)).abs().max()) < 2e-5 * float((ref + res).abs().max()) + 1e-5
    assert float((sums.cpu().double()[0] - o.sum(0)).abs().max()) < 1e-5 * float(o.abs().sum(0).max()) + 1e-4
    assert float((sq.cpu().double()[0] - (o * o).sum(0)).abs().max()) < 1e-5 * float((o * o).sum(0).max()) + 1e-4


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", [(2, 8, 8, 256, 64, 3, False, False), (4, 4, 4, 512, 136, 3, False, True),
                                  (3, 4, 4, 256, 128, 3, True, False), (8, 1, 1, 2048, 16, 1, False, False)])
def test_conv_split_k(case, dt):
    """small grids with a long reduction take the split-K path (f32 atomics into a zeroed output)"""
    from layout2img_amd import ops
    B, H, W, Ci, Co, KH, up2, pool2 = case
    g = torch.Generator().manual_seed(11)
    x = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    w = _rt(torch.randn(Co, Ci, KH, KH, generator=g) / math.sqrt(Ci * KH * KH), dt)
    bias = torch.randn(Co, generator=g)
    ref = _ref_conv(x, w, bias, up2, pool2)
    res = torch.randn(ref.shape, generator=g)
    mask = _rt(torch.randn(ref.shape, generator=g), dt)
    pack, kpad = _pack(w, 64 if dt == torch.bfloat16 else 32)
    out, _, _ = ops.conv_raw(x.to(_dev(), dt), pack.to(_dev(), dt), kpad, Co, KH, bias=bias.to(_dev()), res=res.to(_dev()),
                             relu_mask=mask.to(_dev(), dt), up2=up2, pool2=pool2, alpha=0.25 if pool2 else 1.0)
    expect = ref * (mask > 0).float() + res
    assert float((out.cpu() - expect).abs().max()) < 3e-5 * float(expect.abs().max()) + 1e-5


@pytest.mark.parametrize("cfg", [10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 29, 40, 41])   # (40 / 41: conv_halo8_kernel, the 256 x 256 eight-wave tile)
@pytest.mark.parametrize("case", [(2, 16, 16, 64, 72, 3, False, False), (1, 32, 32, 128, 136, 3, False, True),
                                  (2, 16, 16, 64, 64, 3, True, False), (9, 16, 16, 192, 128, 3, False, False),
                                  (5, 8, 8, 64, 264, 3, False, True), (1, 64, 64, 64, 8, 3, True, False),
                                  (3, 16, 16, 320, 136, 3, False, False), (2, 32, 32, 136, 72, 3, False, False),
                                  (7, 8, 8, 128, 136, 3, False, False), (6, 4, 4, 64, 64, 3, True, False),
                                  (1, 32, 32, 64, 200, 3, True, True), (2, 64, 64, 64, 64, 3, False, True),
                                  (3, 8, 8, 128, 128, 3, True, False), (1, 128, 128, 64, 64, 3, False, False)])
def test_conv_halo_tiles(case, cfg, epi):
    """every tile shape of the halo kernels (128x128, 128x64, 256x128; 17 / 18: the 256-pixel-tile kernel with its
    bordered and compact halos) on every geometry, forced through the tuning hook -- with both forms of the epilogue
    (epi 0: direct stores from the accumulator layout, 2: the coalesced form through LDS; hook 4000 + epi)"""
    from layout2img_amd import ops, _lib
    B, H, W, Ci, Co, KH, up2, pool2 = case
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    x = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    w = _rt(torch.randn(Co, Ci, KH, KH, generator=g) / math.sqrt(Ci * KH * KH), dt)
    bias = torch.randn(Co, generator=g)
    ref = _ref_conv(x, w, bias, up2, pool2)
    pack, kpad = _pack(w, 64)
    _lib.call("l2i_set_conv_config", cfg)
    try:
        out, _, _ = ops.conv_raw(x.to(_dev(), dt), pack.to(_dev(), dt), kpad, Co, KH, bias=bias.to(_dev()), up2=up2, pool2=pool2,
                                 alpha=0.25 if pool2 else 1.0)
    finally:
        _lib.call("l2i_set_conv_config", -1)
    # operands are pre-rounded to bf16, products are exact in f32: only the accumulation order differs from torch's
    assert float((out.cpu() - ref).abs().max()) < 3e-5 * float(ref.abs().max())


# (B, H, W, Ci, Co, sc_Ci, sc_up2, pool2, live images or None): the block shapes of the two networks in miniature
SC_CASES = [(2, 16, 16, 64, 72, 128, False, False, None), (1, 32, 32, 128, 136, 64, True, False, None),
            (3, 16, 16, 192, 128, 64, False, True, None), (5, 8, 8, 128, 264, 192, False, True, 3),
            (2, 32, 32, 64, 64, 128, True, False, None), (9, 8, 8, 64, 128, 64, True, False, None),
            (1, 64, 64, 64, 64, 128, False, True, None), (2, 16, 16, 64, 64, 72, False, False, None)]


@pytest.mark.parametrize("cfg", [-1, 14, 15, 19, 29, 40, 41])
@pytest.mark.parametrize("case", SC_CASES)
def test_conv_folded_shortcut(case, cfg, epi):
    """l2i_conv2d_fwd_sc: conv3x3(h) + conv1x1(x at (y >> up, x >> up)) + both biases (+ 2x2 pool of the sum) in ONE launch
    (reference model/resnet_generator_app_v2.py:664-678, model/rcnn_discriminator_app.py:317-344) equals the torch
    composition, on every halo tile shape (folded: conv_sc_tail) and where the library un-folds (the heuristic's generic
    / split-K choices for the small cases; sc_Ci = 72 is not a multiple of 64) -- with a live-image count as the ROI heads
    pass it, and with the epilogue's operand copy and batch statistics riding on the same launch."""
    from layout2img_amd import ops, _lib
    B, H, W, Ci, Co, sCi, sup, pool2, live = case
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(17)
    x = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    w = _rt(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9), dt)
    hs, ws_ = (H // 2, W // 2) if sup else (H, W)
    xs = _rt(torch.randn(B, hs, ws_, sCi, generator=g), dt)
    wsc = _rt(torch.randn(Co, sCi, 1, 1, generator=g) / math.sqrt(sCi), dt)
    bias, bias_sc = torch.randn(Co, generator=g), torch.randn(Co, generator=g)
    ref = _ref_conv(x, w, bias, False, pool2) + _ref_conv(xs, wsc, bias_sc, sup, pool2)
    pack, kpad = _pack(w, 64)
    pack_sc, kpad_sc = _pack(wsc, 64)
    dev = _dev()
    nimg = None
    if live is not None:
        nimg = torch.tensor([live], dtype=torch.int32, device=dev)
        ref[live:] = 0
    co_p = (Co + 7) // 8 * 8
    pad = lambda b: torch.nn.functional.pad(b, (0, co_p - Co)).to(dev)
    placeholder = torch.full(ref.shape[:3] + (co_p,), float("nan"), device=dev)
    sc = dict(x_op=xs.to(dev, dt), wpack=pack_sc.to(dev, dt), kpad=kpad_sc, bias=pad(bias_sc), up2=sup, out=placeholder,
              flops=0.0)
    _lib.call("l2i_set_conv_config", cfg)
    try:
        out, op, _ = ops.conv_raw(x.to(dev, dt), pack.to(dev, dt), kpad, co_p, 3, bias=pad(bias), pool2=pool2,
                                  alpha=0.25 if pool2 else 1.0, nimg=nimg, sc=sc, want_op=True, relu_op=True,
                                  stats=live is None)
    finally:
        _lib.call("l2i_set_conv_config", -1)
    tol = 3e-5 * float(ref.abs().max())
    assert float((out.cpu()[..., :Co] - ref).abs().max()) < tol
    assert float((op.float().cpu()[..., :Co] - torch.relu(ref)).abs().max()) < 1e-2 * float(ref.abs().max())
    if 10 <= cfg < 40:   # a forced halo tile with sc_Ci % 64 == 0 must FOLD: the placeholder is never written
        assert bool(torch.isnan(placeholder).all()) == (sCi % 64 == 0)
    if live is None:
        s1, s2, _ = out._l2i_stats
        assert float((s1.cpu().view(-1)[:Co] - ref.sum(dim=(0, 1, 2))).abs().max()) < 1e-3 * float(ref.abs().sum(dim=(0, 1, 2)).max())


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 72, 3, False, False), (1, 32, 32, 128, 136, 3, False, True), (2, 8, 8, 104, 64, 3, True, False),
                                  (5, 1, 1, 312, 72, 1, False, False), (2, 16, 16, 40, 104, 1, False, False), (2, 4, 4, 1024, 256, 3, False, False)])
def test_split_operand_conv_is_f32_accurate(case):
    """The "bf16x3" forward mode's arithmetic: operands [x_hi | x_lo | x_hi] (l2i_split_cast, with and without the ReLU) against a
    pack [w_hi | w_hi | w_lo] through the UNCHANGED bf16 convolution kernels equal the f32 convolution of the un-rounded operands
    to ~1e-5 relative -- plain bf16 operands are at 4e-3 on the same data."""
    from layout2img_amd import ops
    B, H, W, Ci, Co, KH, up2, pool2 = case
    g = torch.Generator().manual_seed(Ci + Co)
    x = torch.randn(B, H, W, Ci, generator=g)
    w = torch.randn(Co, Ci, KH, KH, generator=g) / math.sqrt(Ci * KH * KH)
    bias = torch.randn(Co, generator=g)
    hi = lambda t: t.to(torch.bfloat16).float()
    w_hi = hi(w)
    w_lo = hi(w - w_hi)
    k = KH * KH * 3 * Ci
    kpad, npad = (k + 63) // 64 * 64, (Co + 127) // 128 * 128
    pack = torch.zeros(npad, kpad)
    pack[:Co, :k] = torch.cat((w_hi, w_hi, w_lo), dim=1).permute(0, 2, 3, 1).reshape(Co, k)   # per tap: [hi | hi | lo] over the input channels
    for relu in (False, True):
        x3 = ops.split_cast(x.to(_dev()), relu=relu)
        xr = torch.relu(x) if relu else x
        assert x3.shape == (B, H, W, 3 * Ci) and torch.equal(x3[..., :Ci].float().cpu(), hi(xr)) and torch.equal(x3[..., 2 * Ci:], x3[..., :Ci])
        assert float((x3[..., :Ci].float().cpu() + x3[..., Ci:2 * Ci].float().cpu() - xr).abs().max()) < 2e-5 * float(xr.abs().max())
        out, _, _ = ops.conv_raw(x3, pack.to(_dev(), torch.bfloat16), kpad, Co, KH, bias=bias.to(_dev()), up2=up2, pool2=pool2,
                                 alpha=0.25 if pool2 else 1.0)
        ref = _ref_conv(xr, w, bias, up2, pool2)
        err = float((out.cpu() - ref).abs().max()) / float(ref.abs().max())
        plain = float((_ref_conv(hi(xr), w_hi, bias, up2, pool2) - ref).abs().max()) / float(ref.abs().max())
        assert err < 3e-5 and err < 0.05 * plain, (err, plain)


@pytest.mark.parametrize("case", [(32, 1024, 1024), (3, 256, 264), (40, 512, 128), (33, 320, 72), (64, 1024, 512)])
def test_conv_small_map_weight_stationary(case):
    """3x3 convolutions on 4x4 maps with >= 256 input channels (D block6, reference model/rcnn_discriminator_app.py:94-96) run on
    conv_wstat_kernel (one 64-channel chunk of the pack per workgroup, all pixels, partial tiles through the stream's scratch) +
    conv_wstat_reduce_kernel (splits, alpha, bias, ReLU mask, residual): forward form and data-gradient form (mask + residual)
    against the f32 convolution, incl. more than 32 images (two pixel tiles), a ragged last tile and Co that is not a multiple of 64."""
    from layout2img_amd import ops
    B, Ci, Co = case
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(B + Ci + Co)
    x = _rt(torch.randn(B, 4, 4, Ci, generator=g), dt)
    w = _rt(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9), dt)
    bias = torch.randn(Co, generator=g)
    ref = _ref_conv(x, w, bias, False, False)
    mask = _rt(torch.randn(ref.shape, generator=g), dt)
    res = torch.randn(ref.shape, generator=g)
    pack, kpad = _pack(w, 64)
    out, _, _ = ops.conv_raw(x.to(_dev(), dt), pack.to(_dev(), dt), kpad, Co, 3, bias=bias.to(_dev()))
    assert float((out.cpu() - ref).abs().max()) < 3e-5 * float(ref.abs().max())
    out2, _, _ = ops.conv_raw(x.to(_dev(), dt), pack.to(_dev(), dt), kpad, Co, 3, relu_mask=mask.to(_dev(), dt), res=res.to(_dev()), alpha=0.5)
    expect = 0.5 * _ref_conv(x, w, None, False, False) * (mask > 0).float() + res
    assert float((out2.cpu() - expect).abs().max()) < 3e-5 * float(expect.abs().max())


@pytest.mark.parametrize("dt", DTYPES)
def test_conv_relu_mask(dt, epi):
    from layout2img_amd import ops
    g = torch.Generator().manual_seed(3)
    x = _rt(torch.randn(2, 8, 8, 16, generator=g), dt)
    w = _rt(torch.randn(32, 16, 3, 3, generator=g) / 12, dt)
    mask = _rt(torch.randn(2, 8, 8, 32, generator=g), dt)
    res = torch.randn(2, 8, 8, 32, generator=g)
    pack, kpad = _pack(w, 64 if dt == torch.bfloat16 else 32)
    out, _, _ = ops.conv_raw(x.to(_dev(), dt), pack.to(_dev(), dt), kpad, 32, 3, res=res.to(_dev()), relu_mask=mask.to(_dev(), dt))
    expect = _ref_conv(x, w, None, False, False) * (mask > 0).float() + res
    assert float((out.cpu() - expect).abs().max()) < 1e-4


# shapes of the eight-wave / two-group weight-gradient kernel (128-pixel steps): 8x8 images inside a step, upsample, pool,
# an odd number of steps per split, a 1x1 layer
WGRAD_EXTRA = [(4, 8, 8, 64, 128, 3, False, False), (2, 16, 16, 64, 136, 3, True, False), (2, 16, 16, 64, 128, 3, False, True),
               (6, 16, 16, 72, 264, 3, False, False), (3, 32, 32, 64, 128, 1, False, False), (2, 64, 64, 16, 136, 3, False, True)]


@pytest.mark.parametrize("combine", ["scratch", "atomics"])
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES + WGRAD_EXTRA)
def test_conv_wgrad(case, dt, combine, monkeypatch):
    """combine: how the partial tiles of a split pixel reduction reach dw -- stored to the caller's scratch and added by
    a second kernel (the default), or f32 atomics on dw (no scratch given); dw starts non-zero (it is accumulated into)"""
    from layout2img_amd import ops, _lib
    if combine == "atomics":
        monkeypatch.setattr(_lib, "wgrad_scratch", lambda device: (None, 0))
    B, H, W, Ci, Co, KH, up2, pool2 = case
    if Co % 8:
        pytest.skip("operand channels are padded to 8 by the caller")
    g = torch.Generator().manual_seed(hash(case) % 1000 + 1)
    x = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    w = torch.zeros(Co, Ci, KH, KH, requires_grad=True)
    y = _ref_conv(x, w, None, up2, pool2)
    dy = _rt(torch.randn(y.shape, generator=g), dt)
    y.backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).reshape(Co, -1)
    K = KH * KH * Ci
    dw0 = torch.randn(Co, K, generator=g)
    dw = dw0.to(_dev())
    ops.wgrad_raw(x.to(_dev(), dt), dy.to(_dev(), dt), dw, K, Co, KH, up2=up2, pool2=pool2, alpha=0.25 if pool2 else 1.0)
    scale = float(ref.abs().max())
    assert float((dw.cpu() - dw0 - ref).abs().max()) < 5e-5 * scale + 1e-5


class _Net(nn.Module):
    def __init__(self, holders):
        super().__init__()
        self.h = nn.ModuleList(holders)


def _mk(holders, dt):
    from layout2img_amd.arena import FlatParams, WeightArena
    net = _Net(holders)
    flat = FlatParams(net, _dev())
    return net, flat, WeightArena(net, flat, _dev(), dt)


def _sd_of(h, name="c."):
    sd = {}
    if h.sn:
        sd[name + "weight_orig"] = h.weight_orig.detach().cpu().clone().requires_grad_(True)
        sd[name + "weight_u"] = h.weight_u.detach().cpu().clone()
        sd[name + "weight_v"] = h.weight_v.detach().cpu().clone()
    else:
        sd[name + "weight"] = h.weight.detach().cpu().clone().requires_grad_(True)
    if h.bias is not None:
        sd[name + "bias"] = h.bias.detach().cpu().clone().requires_grad_(True)
    return sd


@pytest.mark.parametrize("training", [True, False])
def test_weight_arena_spectral_norm(training):
    """power iteration, sigma, packs (f32) against torch's spectral-norm algebra; two passes."""
    from layout2img_amd.arena import GemmWeight
    torch.manual_seed(0)
    hs = [GemmWeight("conv", 24, 16, 3, sn=True, eps=1e-4), GemmWeight("linear", 300, 308, sn=True, eps=1e-12),
          GemmWeight("embedding", 184, 64, bias=False, sn=True), GemmWeight("conv", 100, 40, 1, sn=False),
          GemmWeight("linear", 4096, 24, sn=True)]
    sds = [_sd_of(h) for h in hs]
    net, flat, arena = _mk(hs, torch.float32)
    for _ in range(2):
        pc = arena.prepare(training=training)
        torch.cuda.synchronize()
        for h, sd in zip(hs, sds):
            wbar = O._w(sd, "c.", h.eps, training).detach()
            co, ci, kh = h.co, h.ci, h.kh
            fwd = pc.fwd_pack(h).view(h.npad, h.kpad).cpu()
            expect = torch.zeros(h.npad, h.kpad)
            w4 = wbar.reshape(co, ci, kh, kh)
            blk = torch.zeros(co, kh * kh, h.ci_p)
            blk[:, :, :ci] = w4.permute(0, 2, 3, 1).reshape(co, kh * kh, ci)
            expect[:co, :kh * kh * h.ci_p] = blk.reshape(co, -1)
            assert float((fwd - expect).abs().max()) < 2e-5 * float(expect.abs().max()), h.kind
            dg = pc.dgrad_pack(h).view(h.npad_d, h.kpad_d).cpu()
            expect = torch.zeros(h.npad_d, h.kpad_d)
            blk = torch.zeros(ci, kh * kh, h.co_p)
            blk[:, :, :co] = w4.flip(2, 3).permute(1, 2, 3, 0).reshape(ci, kh * kh, co)
            expect[:ci, :kh * kh * h.co_p] = blk.reshape(ci, -1)
            assert float((dg - expect).abs().max()) < 2e-5 * float(expect.abs().max()), h.kind
            if h.sn:
                assert float((h.weight_u.cpu() - sd["c.weight_u"]).abs().max()) < 1e-5
                assert float((h.weight_v.cpu() - sd["c.weight_v"]).abs().max()) < 1e-5


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("sup", [False, True], ids=["sc_same_res", "sc_upsampled"])
@pytest.mark.parametrize("case", [(4, 16, 16, 64, 128, 128, False, None), (2, 32, 32, 128, 136, 64, True, None),
                                  (6, 8, 8, 192, 264, 256, True, 4), (3, 16, 16, 72, 64, 136, False, None),
                                  (1, 64, 64, 64, 64, 128, True, None), (5, 8, 8, 64, 128, 64, False, 2)])
def test_conv_wgrad_folded_shortcut(case, dt, sup):
    """l2i_conv2d_wgrad_sc: the 3x3 weight gradient and the weight gradient of the block's 1x1 shortcut (same dY, its own
    input) from ONE launch, with both bias gradients, equal torch's -- on the scalar-step modes that fold (no pool; 2x2 pool),
    with a live-image count, and where the library runs the shortcut separately (f32 operands)."""
    from layout2img_amd import ops
    B, H, W, Ci, Co, sCi, pool2, live = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 7)
    x = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    xs = _rt(torch.randn(B, H >> int(sup), W >> int(sup), sCi, generator=g), dt)   # (sup: the generator's shortcut reads the block input at half resolution)
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    wsc = torch.zeros(Co, sCi, 1, 1, requires_grad=True)
    b1, b2 = torch.zeros(Co, requires_grad=True), torch.zeros(Co, requires_grad=True)
    y = _ref_conv(x, w, b1, False, pool2) + _ref_conv(xs, wsc, b2, sup, pool2)
    dy = _rt(torch.randn(y.shape, generator=g), dt)
    if live is not None:
        dy[live:] = 0   # (rows of dead images carry no gradient; the kernel does not read them at all)
    y.backward(dy)
    dev = _dev()
    kp, kps = 9 * Ci, sCi
    dw = torch.full((Co, kp), 0.5, device=dev)
    dws = torch.full((Co, kps), -0.25, device=dev)
    db, dbs = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    nimg = torch.tensor([live], dtype=torch.int32, device=dev) if live is not None else None
    sc = dict(x_op=xs.to(dev, dt), dw=dws, ldw=kps, dbias=dbs, flops=0.0, up2=sup)
    ops.wgrad_raw(x.to(dev, dt), dy.to(dev, dt), dw, kp, Co, 3, pool2=pool2, alpha=0.25 if pool2 else 1.0, nimg=nimg, dbias=db, sc=sc)
    ref = w.grad.permute(0, 2, 3, 1).reshape(Co, kp)
    refs = wsc.grad.reshape(Co, sCi)
    tol = lambda r: 2e-4 * float(r.abs().max()) + 1e-5
    assert float((dw.cpu() - 0.5 - ref).abs().max()) < tol(ref)
    assert float((dws.cpu() + 0.25 - refs).abs().max()) < tol(refs)
    assert float((db.cpu() - b1.grad).abs().max()) < tol(b1.grad) and float((dbs.cpu() - 1 - b2.grad).abs().max()) < tol(b2.grad)


@pytest.mark.parametrize("case", [(32, 4, 4, 1024, 1024, 3), (32, 8, 8, 512, 1024, 3), (4, 16, 16, 64, 128, 3), (2, 64, 64, 64, 64, 3), (6, 32, 32, 64, 128, 1),
                                  (8, 128, 128, 64, 64, 3)])
def test_conv_wgrad_overwrite_mode(case):
    """`overwrite` (l2i_conv2d_wgrad_dual): the launch is the only writer of its dW slice, so the result is STORED -- whatever the slice
    held before (NaN here) is gone: on single-split tiles (no atomics), split tiles (the reduce kernel writes) and split groups alike
    (few tiles x many splits, the 64-channel layers: the reduce's groups add with atomics into a slice the weight-gradient kernel has
    cleared on its way in, WgradArgs::zero_targets). The trainer's dW-bar accumulators are torch.empty (arena.PassCtx.dw)."""
    from layout2img_amd import ops
    B, H, W, Ci, Co, KH = case
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(B + Ci)
    x = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    wz = torch.zeros(Co, Ci, KH, KH, requires_grad=True)
    y = _ref_conv(x, wz, None, False, False)
    dy = _rt(torch.randn(y.shape, generator=g), dt)
    y.backward(dy)
    ref = wz.grad.permute(0, 2, 3, 1).reshape(Co, -1)
    K = KH * KH * Ci
    for fill in (float("nan"), 0.0):
        dw = torch.full((Co, K), fill, device=_dev())
        ops.wgrad_raw(x.to(_dev(), dt), dy.to(_dev(), dt), dw, K, Co, KH, overwrite=True)
        assert bool(torch.isfinite(dw).all())
        assert float((dw.cpu() - ref).abs().max()) < 2e-4 * float(ref.abs().max()) + 1e-5
    # without the caller's scratch the splits add with atomics straight from the main kernel: the library clears the slice itself
    from layout2img_amd import _lib
    xo, dyo = x.to(_dev(), dt), dy.to(_dev(), dt)
    dw = torch.full((Co, K), float("nan"), device=_dev())
    _lib.call("l2i_conv2d_wgrad_dual", xo.data_ptr(), dyo.data_ptr(), dw.data_ptr(), 1, B, H, W, Ci, H, W, Co, KH, 0, 0, K, 1.0, None, None,
              None, 0, None, None, 0, 0, 0, None, None, None, 1, _lib.raw_stream())
    assert float((dw.cpu() - ref).abs().max()) < 2e-4 * float(ref.abs().max()) + 1e-5


def test_spectral_norm_backward_two_passes_in_one_launch():
    """l2i_weights_backward2: the sigma-corrections of two passes over the same weights (D(real) + D(fake),
    train_context_app_v2.py:158,167 -- each with its own u, v, sigma) applied by one launch pair equal the two single-pass
    launches (whose formula the model-level gradient tests pin against the reference), incl. a non-SN layer, a 1x1 layer,
    a weight applied twice per forward, and an odd number of pending passes."""
    from layout2img_amd import _lib
    from layout2img_amd.arena import GemmWeight
    torch.manual_seed(3)
    hs = [GemmWeight("conv", 72, 40, 3, sn=True, eps=1e-4), GemmWeight("linear", 300, 308, sn=True, eps=1e-12),
          GemmWeight("conv", 100, 40, 1, sn=False), GemmWeight("conv", 64, 64, 3, sn=True, uses=2)]
    net, flat, arena = _mk(hs, torch.float32)
    g = torch.Generator(device="cpu").manual_seed(5)

    def passes(n):
        out = []
        for _ in range(n):
            pc = arena.prepare(training=True)
            pc.dw().copy_(torch.randn(pc.dw().shape, generator=g))
            out.append(pc)
        return out

    for n in (2, 3):
        pcs = passes(n)
        flat.zero_grad()   # (marks the buffer fresh: the first launch pair WRITES the weights' slices, later ones accumulate)
        arena.flush_grads()
        assert not flat.fresh
        fused = flat.grad.clone()
        flat.grad.zero_()
        for pc in pcs:
            _lib.call("l2i_weights_backward", arena.layers.data_ptr(), arena.n_layers, arena.t_dot.data_ptr(), arena.n_dot,
                      arena.t_apply.data_ptr(), arena.n_apply, flat.data.data_ptr(), pc.dwbar.data_ptr(), pc.pass_uv.data_ptr(),
                      pc.norms.data_ptr(), flat.grad.data_ptr(), _lib.workspace(arena.device), arena.t_dot_range.data_ptr(), 0, _lib.raw_stream())
        torch.cuda.synchronize()
        assert float(fused.abs().max()) > 0
        assert float((fused - flat.grad).abs().max()) < 1e-5 * float(flat.grad.abs().max())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("pro", ["cast", "relu", "isla", "affine", "instance", "isla_big"])
def test_fused_conv_fwd_bwd(pro, dt):
    """FusedConvFn (prologue + conv + bias + res [+ up2]) forward and all gradients, incl. the spectral-norm
    backward into the flat gradient buffer, against torch autograd on the oracle formulas."""
    from layout2img_amd import ops
    from layout2img_amd.arena import GemmWeight
    torch.manual_seed(1)
    B, H, W, Ci, Co, O_ = 3, 8, 8, 16, 24, 5
    if pro == "isla_big":   # 2 channel chunks (one partial), 2 object chunks, several 32-pixel sub-tiles per block
        B, H, W, Ci, Co, O_, pro = 40, 32, 32, 136, 8, 11, "isla"
        if dt == torch.float32:
            B = 8
    up2 = pro in ("isla", "cast") and H == 8
    h = GemmWeight("conv", Co, Ci, 3, sn=True, eps=1e-4)
    sd = _sd_of(h)
    net, flat, arena = _mk([h], dt)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, H, W, Ci, generator=g)
    mask = torch.rand(B, O_, H, W, generator=g) * (torch.rand(B, O_, H, W, generator=g) > 0.3)
    wproj, bproj = torch.randn(B, O_, Ci, generator=g) * 0.3, torch.randn(B, O_, Ci, generator=g) * 0.3
    aw, ab = 1 + 0.2 * torch.randn(Ci, generator=g), 0.2 * torch.randn(Ci, generator=g)
    Ho = 2 * H if up2 else H
    res = torch.randn(B, Ho, Ho, Co, generator=g)
    dy = torch.randn(B, Ho, Ho, Co, generator=g)

    # ---- reference (CPU, f32, operand rounded to dt where the kernel rounds)
    xr = x.clone().requires_grad_(True)
    mr, wr, br = mask.clone().requires_grad_(True), wproj.clone().requires_grad_(True), bproj.clone().requires_grad_(True)
    awr, abr = aw.clone().requires_grad_(True), ab.clone().requires_grad_(True)
    resr = res.clone().requires_grad_(True)
    xn = xr.permute(0, 3, 1, 2)
    if pro == "cast":
        a = xn
    elif pro == "relu":
        a = F.relu(xn)
    elif pro == "isla":
        xh = F.batch_norm(xn, None, None, None, None, True, 0.1, 1e-5)
        den = mr.sum(1, keepdim=True) + 1e-6
        a = F.relu((torch.einsum("bohw,boc->bchw", mr, wr) / den + 1) * xh + torch.einsum("bohw,boc->bchw", mr, br) / den)
    elif pro == "affine":
        a = F.relu(F.batch_norm(xn, None, None, awr, abr, True, 0.1, 1e-5))
    else:
        a = F.relu(F.instance_norm(xn, eps=1e-5))

    class _STE(torch.autograd.Function):  # operand rounding with straight-through gradient
        @staticmethod
        def forward(ctx, t):
            return _rt(t, dt)

        @staticmethod
        def backward(ctx, gg):
            return gg
    a = _STE.apply(a)
    if up2:
        a = F.interpolate(a, scale_factor=2, mode="nearest")
    wbar = O._w(sd, "c.", 1e-4, True)
    yref = F.conv2d(a, _STE.apply(wbar), sd["c.bias"], 1, 1).permute(0, 2, 3, 1) + resr
    yref.backward(dy)

    # ---- HIP
    dev = _dev()
    xg = x.to(dev).requires_grad_(True)
    mg, wg, bg = mask.to(dev).requires_grad_(True), wproj.to(dev).requires_grad_(True), bproj.to(dev).requires_grad_(True)
    awg, abg = aw.to(dev).requires_grad_(True), ab.to(dev).requires_grad_(True)
    resg = res.to(dev).requires_grad_(True)
    pc = arena.prepare(training=True)
    kw = {}
    if pro == "relu":
        kw["prologue"] = ops.RELU
    elif pro == "isla":
        kw.update(prologue=ops.NormSpec(0), mask=mg, wproj=wg, bproj=bg)
    elif pro == "affine":
        kw.update(prologue=ops.NormSpec(1), wproj=awg, bproj=abg)
    elif pro == "instance":
        kw.update(prologue=ops.NormSpec(2, instance=True))
    y = ops.fused_conv(xg, h, pc, res=resg, up2=up2, **kw)
    y.backward(dy.to(dev))
    arena.flush_grads()
    torch.cuda.synchronize()

    bf = dt == torch.bfloat16
    tol = 2e-2 if bf else 2e-4   # bf16: dy and the dgrad weights are rounded as operands too
    def close(a_, b_, name, t=tol):
        s = float(b_.abs().max()) + 1e-6
        d = float((a_.detach().cpu() - b_).abs().max())
        assert d < t * s, (name, d, s)
    close(y, yref, "y", 2e-3 if bf else 2e-5)
    close(xg.grad, xr.grad, "dx")
    close(resg.grad, resr.grad, "dres", 1e-6)
    close(h.bias.grad, sd["c.bias"].grad, "dbias", 2e-3 if bf else 2e-5)
    close(h.weight_orig.grad, sd["c.weight_orig"].grad, "dW")
    if pro == "isla":
        close(mg.grad, mr.grad, "dmask")
        close(wg.grad, wr.grad, "dwproj")
        close(bg.grad, br.grad, "dbproj")
    if pro == "affine":
        close(awg.grad, awr.grad, "daw")
        close(abg.grad, abr.grad, "dab")


@pytest.mark.parametrize("dt", DTYPES)
def test_fused_conv_pool_and_linear(dt):
    """D-style block piece (relu prologue, avg-pool epilogue, 1x1 shortcut with pool) and a linear layer."""
    from layout2img_amd import ops
    from layout2img_amd.arena import GemmWeight
    torch.manual_seed(2)
    c1, csc, lin = GemmWeight("conv", 32, 16, 3, sn=True, eps=1e-4), GemmWeight("conv", 32, 16, 1, sn=True, eps=1e-4), \
        GemmWeight("linear", 40, 308, sn=True)
    sds = [_sd_of(c1, "a."), _sd_of(csc, "b."), _sd_of(lin, "l.")]
    net, flat, arena = _mk([c1, csc, lin], dt)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 16, 16, 16, generator=g)
    v = torch.randn(6, 312, generator=g)
    v[:, 308:] = 0

    class _STE(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return _rt(t, dt)

        @staticmethod
        def backward(ctx, gg):
            return gg
    xr, vr = x.clone().requires_grad_(True), v.clone().requires_grad_(True)
    xn = xr.permute(0, 3, 1, 2)
    sc = F.avg_pool2d(F.conv2d(_STE.apply(xn), _STE.apply(O._w(sds[1], "b.", 1e-4, True)), sds[1]["b.bias"]), 2)
    yr = F.avg_pool2d(F.conv2d(_STE.apply(F.relu(xn)), _STE.apply(O._w(sds[0], "a.", 1e-4, True)), sds[0]["a.bias"], 1, 1), 2) + sc
    lr = F.linear(_STE.apply(vr[:, :308]), _STE.apply(O._w(sds[2], "l.", 1e-12, True)), sds[2]["l.bias"])
    gy, gl = torch.randn(yr.shape, generator=g), torch.randn(lr.shape, generator=g)
    (yr * gy).sum().backward()
    (lr * gl).sum().backward()

    dev = _dev()
    xg, vg = x.to(dev).requires_grad_(True), v.to(dev).requires_grad_(True)
    pc = arena.prepare(training=True)
    scg = ops.fused_conv(xg, csc, pc, pool2=True)
    yg = ops.fused_conv(xg, c1, pc, prologue=ops.RELU, res=scg, pool2=True)
    lg = ops.fused_conv(vg.view(6, 1, 1, 312), lin, pc).view(6, -1)[:, :40]
    (yg * gy.permute(0, 2, 3, 1).contiguous().to(dev)).sum().backward()
    (lg * gl.to(dev)).sum().backward()
    arena.flush_grads()
    bf = dt == torch.bfloat16
    tol = 2e-2 if bf else 2e-4

    def close(a_, b_, name, t=tol):
        s = float(b_.abs().max()) + 1e-6
        d = float((a_.detach().cpu() - b_).abs().max())
        assert d < t * s, (name, d, s)
    close(yg, yr.permute(0, 2, 3, 1), "y", 2e-3 if bf else 2e-5)
    close(lg, lr, "lin", 2e-3 if bf else 2e-5)
    close(xg.grad, xr.grad, "dx")
    close(vg.grad[:, :308], vr.grad[:, :308], "dv")
    for hh, sd, p in ((c1, sds[0], "a."), (csc, sds[1], "b."), (lin, sds[2], "l.")):
        close(hh.weight_orig.grad, sd[p + "weight_orig"].grad, p + "dW")
        close(hh.bias.grad, sd[p + "bias"].grad, p + "db", 2e-3 if bf else 2e-5)


def test_roi_align_fwd_bwd():
    from layout2img_amd import ops
    g = torch.Generator().manual_seed(4)
    fs, fl = torch.randn(2, 32, 32, 16, generator=g), torch.randn(2, 16, 16, 16, generator=g)
    rois = torch.tensor([[0, 10.0, 12.0, 100.0, 90.0], [1, 5.5, 3.25, 40.0, 60.0], [0, 0.0, 0.0, 128.0, 128.0],
                         [1, 100.0, 100.0, 127.0, 120.0], [0, 60.0, 10.0, 70.0, 18.0], [1, -76.8, -76.8, -12.8, -12.8],
                         [0, 20.0, 20.0, 83.0, 84.5]])
    valid = torch.tensor([1, 1, 1, 1, 1, 0, 1], dtype=torch.int32)
    fsr, flr = fs.clone().requires_grad_(True), fl.clone().requires_grad_(True)
    small = ((rois[:, 3] - rois[:, 1]) < 64) & ((rois[:, 4] - rois[:, 2]) < 64)
    ref = torch.zeros(7, 8, 8, 16)
    for i in range(7):
        if not valid[i]:
            continue
        f, sc = (fsr, 0.25) if small[i] else (flr, 0.125)
        ref[i] = O.roi_align(f.permute(0, 3, 1, 2), rois[i:i + 1], 8, sc, 0)[0].permute(1, 2, 0)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    dev = _dev()
    a, b = fs.to(dev).requires_grad_(True), fl.to(dev).requires_grad_(True)
    out = ops.roi_align(a, b, rois.to(dev), valid.to(dev))
    out.backward(gout.to(dev))
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 1e-5
    assert float((a.grad.cpu() - fsr.grad).abs().max()) < 1e-4
    assert float((b.grad.cpu() - flr.grad).abs().max()) < 1e-4


@pytest.mark.parametrize("O_,geo", [(8, True), (31, False), (31, True)])
def test_box_attention(O_, geo):
    from layout2img_amd import ops
    g = torch.Generator().manual_seed(5)
    B, D = 3, 308
    q, k, v = [torch.randn(B, O_, D, generator=g) for _ in range(3)]
    ge = torch.rand(B, O_, O_, generator=g) * (torch.rand(B, O_, O_, generator=g) > 0.2) if geo else None
    y = torch.randint(0, 3, (B, O_), generator=g)
    y[:, 0] = 1
    qr, kr, vr = [t.clone().requires_grad_(True) for t in (q, k, v)]
    gr = ge.clone().requires_grad_(True) if geo else None
    s = torch.matmul(qr, kr.transpose(-2, -1)) / math.sqrt(D)
    s = s.masked_fill(y.unsqueeze(1).expand(B, O_, O_) == 0, -1e9)
    if geo:
        s = torch.log(torch.clamp(gr, min=1e-6)) + s
    ref = torch.matmul(torch.softmax(s, -1), vr)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    dev = _dev()
    qg, kg, vg = [t.to(dev).requires_grad_(True) for t in (q, k, v)]
    gg = ge.to(dev).requires_grad_(True) if geo else None
    out = ops.box_attention(qg, kg, vg, gg, (y != 0).to(torch.int32).to(dev), 1 / math.sqrt(D))
    out.backward(go.to(dev))
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 2e-5
    for a_, b_ in ((qg, qr), (kg, kr), (vg, vr)):
        assert float((a_.grad.cpu() - b_.grad).abs().max()) < 5e-5
    if geo:
        assert float((gg.grad.cpu() - gr.grad).abs().max()) < 1e-3 * float(gr.grad.abs().max())


def test_hinge_l1_adam():
    from layout2img_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(37, 1, generator=g) * 2
    valid = (torch.rand(37, generator=g) > 0.3).to(torch.int32)
    for mode in (0, 1, 2):
        xr = x.clone().requires_grad_(True)
        sel = xr[valid.bool()]
        ref = (F.relu(1 - sel).mean() if mode == 0 else F.relu(1 + sel).mean() if mode == 1 else -sel.mean()) * 0.7
        ref.backward()
        xg = x.to(dev).requires_grad_(True)
        out = ops.hinge(xg, valid.to(dev), mode, 0.7)
        out.backward()
        assert abs(float(out) - float(ref)) < 1e-5
        assert float((xg.grad.cpu() - xr.grad).abs().max()) < 1e-6
    a, b = torch.randn(2, 3, 16, 16, generator=g), torch.randn(2, 3, 16, 16, generator=g)
    ar = a.clone().requires_grad_(True)
    F.l1_loss(ar, b).backward()
    ag = a.to(dev).requires_grad_(True)
    l = ops.l1_loss(ag, b.to(dev))
    l.backward()
    assert abs(float(l) - float(F.l1_loss(a, b))) < 1e-5 and float((ag.grad.cpu() - ar.grad).abs().max()) < 1e-7
    # Adam, betas (0, 0.999), three steps
    p = torch.randn(1024, generator=g)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-4, betas=(0.0, 0.999))

    class _F:
        pass
    flat = _F()
    flat.data, flat.grad, flat.numel = p.to(dev), torch.zeros(1024, device=dev), 1024
    m, v = torch.zeros(1024, device=dev), torch.zeros(1024, device=dev)
    for t in range(1, 4):
        gr = torch.randn(1024, generator=g)
        pr.grad = gr.clone()
        opt.step()
        flat.grad.copy_(gr)
        ops.adam_step(flat, m, v, 1e-4, 0.0, 0.999, 1e-8, t, lo=0, hi=512)   # (two ranges: the data-parallel trainer updates layer groups
        ops.adam_step(flat, m, v, 1e-4, 0.0, 0.999, 1e-8, t, lo=512)         #  one by one, trainer.FlatAdam)
    assert float((flat.data.cpu() - pr.detach()).abs().max()) < 1e-6


@pytest.mark.parametrize("case", [(32, 4, 1024, 8), (4, 16, 256, 8), (2, 32, 128, 7), (2, 32, 64, 8), (3, 8, 136, 5), (1, 64, 64, 8), (40, 8, 256, 3)])
def test_isla_backward_register_resident_shapes(case):
    """ISLA norm + ReLU with at most 8 objects (norm_bwd_a8_kernel): projection gradients through per-workgroup partial rows and
    the last-arriver sum (one pixel segment per image: plain read-modify-write; several: partials + arrival counter), mask
    gradient without atomics when one channel chunk covers C -- against torch autograd on the reference formula
    (model/norm_module.py:163-186). Run twice on the same scratch: the counters are left all-zero."""
    from layout2img_amd import ops
    B, H, C, O_ = case
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, H, H, C, generator=g) * 1.5 + 0.3
    mask = torch.rand(B, O_, H, H, generator=g) * (torch.rand(B, O_, H, H, generator=g) > 0.3)
    wproj, bproj = torch.randn(B, O_, C, generator=g) * 0.3, torch.randn(B, O_, C, generator=g) * 0.3
    go = torch.randn(B, H, H, C, generator=g)
    xr, mr, wr, br = [t.clone().requires_grad_(True) for t in (x, mask, wproj, bproj)]
    xh = F.batch_norm(xr.permute(0, 3, 1, 2), None, None, None, None, True, 0.1, 1e-5)
    den = mr.sum(1, keepdim=True) + 1e-6
    ref = F.relu((torch.einsum("bohw,boc->bchw", mr, wr) / den + 1) * xh + torch.einsum("bohw,boc->bchw", mr, br) / den).permute(0, 2, 3, 1)
    ref.backward(go)
    dev = _dev()
    for rep in range(2):
        xg, mg, wg, bg = [t.to(dev).requires_grad_(True) for t in (x, mask, wproj, bproj)]
        out = ops.norm_act(xg, ops.NormSpec(0), wg, bg, mask=mg)
        out.backward(go.to(dev))
        torch.cuda.synchronize()
        assert float((out.detach().cpu() - ref.detach()).abs().max()) < 5e-5
        for a_, b_, name in ((xg, xr, "dx"), (mg, mr, "dmask"), (wg, wr, "dwproj"), (bg, br, "dbproj")):
            d, sc = float((a_.grad.cpu() - b_.grad).abs().max()), float(b_.grad.abs().max())
            assert d < 2e-4 * sc, (name, rep, d, sc)


def test_norm_act_standalone():
    from layout2img_amd import ops
    g = torch.Generator().manual_seed(8)
    x = torch.randn(4, 8, 8, 104, generator=g) * 2 + 0.5
    aw, ab = 1 + 0.2 * torch.randn(104, generator=g), 0.2 * torch.randn(104, generator=g)
    xr, awr, abr = x.clone().requires_grad_(True), aw.clone().requires_grad_(True), ab.clone().requires_grad_(True)
    ref = F.relu(F.batch_norm(xr.permute(0, 3, 1, 2), None, None, awr, abr, True, 0.1, 1e-5)).permute(0, 2, 3, 1)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    dev = _dev()
    xg, awg, abg = x.to(dev).requires_grad_(True), aw.to(dev).requires_grad_(True), ab.to(dev).requires_grad_(True)
    rm, rv = torch.zeros(104, device=dev), torch.ones(104, device=dev)
    out = ops.norm_act(xg, ops.NormSpec(1, running=(rm, rv)), awg, abg)
    out.backward(go.to(dev))
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 2e-5
    assert float((xg.grad.cpu() - xr.grad).abs().max()) < 2e-4 * float(xr.grad.abs().max())
    assert float((awg.grad.cpu() - awr.grad).abs().max()) < 2e-4 * float(awr.grad.abs().max())
    assert float((abg.grad.cpu() - abr.grad).abs().max()) < 2e-4 * float(abr.grad.abs().max())
    xm = x.reshape(-1, 104)
    assert float((rm.cpu() - 0.1 * xm.mean(0)).abs().max()) < 1e-5
    assert float((rv.cpu() - (0.9 + 0.1 * xm.var(0, unbiased=True))).abs().max()) < 1e-4


@pytest.mark.parametrize("case", [(3, 5, 64, 64, 128, 128), (2, 8, 64, 64, 32, 32), (1, 2, 64, 64, 4, 4), (2, 3, 16, 16, 64, 64)])
def test_resize_bilinear(case):
    """l2i_resize_bilinear == F.interpolate(mode="bilinear", align_corners=False), forward and backward"""
    from layout2img_amd import ops
    b, o, h, w, H, W = case
    g = torch.Generator().manual_seed(4)
    x = torch.rand(b, o, h, w, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = F.interpolate(xr, size=(H, W), mode="bilinear")
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xg = x.to(_dev()).requires_grad_(True)
    out = ops.resize_bilinear(xg, H, W)
    out.backward(dy.to(_dev()))
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 1e-6
    assert float((xg.grad.cpu() - xr.grad).abs().max()) < 1e-5


def test_gram_head_matches_the_explicit_gram_matrices():
    """ops.gram_head == w1 . sum_rows(F F^T / C) / C with F = relu(x) viewed (C, hw) (reference
    model/rcnn_discriminator_app.py:148-157), forward and both gradients"""
    from layout2img_amd import ops
    g = torch.Generator().manual_seed(9)
    R, H, C = 5, 8, 72
    x = torch.randn(R, H, H, C, generator=g)
    w = torch.randn(C, generator=g)
    dy = torch.randn(R, 1, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    Fm = F.relu(xr).permute(0, 3, 1, 2).reshape(R, C, H * H)
    gram = torch.bmm(Fm, Fm.transpose(1, 2)) / C                 # (R, C, C)
    ref = (gram.sum(dim=1) / C) @ wr                               # rows summed, then the head's first-half weight
    ref.view(R, 1).backward(dy)
    xg, wg = x.to(_dev()).requires_grad_(True), w.to(_dev()).requires_grad_(True)
    out = ops.gram_head(xg, wg)
    out.backward(dy.to(_dev()))
    s = float(ref.abs().max())
    assert float((out.detach().cpu().view(-1) - ref.detach()).abs().max()) < 1e-5 * s
    assert float((xg.grad.cpu() - xr.grad).abs().max()) < 1e-5 * float(xr.grad.abs().max())
    assert float((wg.grad.cpu() - wr.grad).abs().max()) < 1e-5 * float(wr.grad.abs().max())


@pytest.mark.parametrize("dt", DTYPES)
def test_discriminator_heads_fused(dt):
    """ops.proj_head / ops.emb_dot against the reference's head arithmetic written with torch ops
    (model/rcnn_discriminator_app.py:127-129, 154-157, 160-166): outputs, dx and every weight / bias gradient
    (through the spectral-norm backward of the arena); repeated classes and an all-negative (dead) row included."""
    from layout2img_amd import ops
    from layout2img_amd.arena import GemmWeight
    torch.manual_seed(4)
    C, K, R, H = 72, 11, 9, 4
    lin, emb, app, emba = GemmWeight("linear", 1, C, sn=True), GemmWeight("embedding", K, C, bias=False, sn=True), \
        GemmWeight("linear", 1, 2 * C, sn=True), GemmWeight("embedding", K, C, bias=False, sn=True)
    sds = [_sd_of(lin, "l."), _sd_of(emb, "e."), _sd_of(app, "a."), _sd_of(emba, "f.")]
    net, flat, arena = _mk([lin, emb, app, emba], dt)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(R, H, H, C, generator=g)
    x[3] = -x[3].abs()
    y = torch.randint(0, K, (R,), generator=g)
    y[5] = y[1]
    gy, ga, gi = (torch.randn(R, 1, generator=g) for _ in range(3))

    class _STE(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return _rt(t, dt)

        @staticmethod
        def backward(ctx, gg):
            return gg
    xr = x.clone().requires_grad_(True)
    wl, we, wa, wf = (_STE.apply(O._w(sd, p, 1e-12, True)) for sd, p in zip(sds, ("l.", "e.", "a.", "f.")))
    f = F.relu(xr).sum(dim=(1, 2))
    obj_r = F.linear(f, wl, sds[0]["l.bias"]) + torch.sum(we.index_select(0, y) * f, dim=1, keepdim=True)
    img_r = F.linear(F.relu(xr).mean(dim=(1, 2)), wl, sds[0]["l.bias"])
    app_r = wf.index_select(0, y) @ wa[0, C:].unsqueeze(1) + sds[2]["a.bias"]
    ((obj_r * gy).sum() + (img_r * gi).sum() + (app_r * ga).sum()).backward()

    dev = _dev()
    xg = x.to(dev).requires_grad_(True)
    yd = y.to(dev)
    pc = arena.prepare(training=True)
    obj = ops.proj_head(xg, lin, pc, emb=emb, y=yd)
    img = ops.proj_head(xg, lin, pc, scale=1.0 / (H * H))
    apph = ops.emb_dot(emba, yd, app, C, pc)
    ((obj * gy.to(dev)).sum() + (img * gi.to(dev)).sum() + (apph * ga.to(dev)).sum()).backward()
    arena.flush_grads()
    bf = dt == torch.bfloat16
    tol = 2e-2 if bf else 2e-4

    def close(a_, b_, name, t=tol):
        sc = float(b_.abs().max()) + 1e-6
        d = float((a_.detach().cpu() - b_).abs().max())
        assert d < t * sc, (name, d, sc)
    close(obj, obj_r, "obj", 2e-3 if bf else 2e-5)
    close(img, img_r, "img", 2e-3 if bf else 2e-5)
    close(apph, app_r, "app", 2e-3 if bf else 2e-5)
    close(xg.grad, xr.grad, "dx")
    for hh, sd, p in ((lin, sds[0], "l."), (emb, sds[1], "e."), (app, sds[2], "a."), (emba, sds[3], "f.")):
        close(hh.weight_orig.grad, sd[p + "weight_orig"].grad, p + "dW")
        if hh.bias is not None:
            close(hh.bias.grad, sd[p + "bias"].grad, p + "db", 2e-3 if bf else 2e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", [(2, 16, 8, 12), (1, 32, 128, 100)])
def test_psp_pool_and_expand(case, dt):
    """ops.psp_pool / ops.psp_expand == the pyramid stages of reference model/resnet_generator_app_v2.py:741-751 written
    with nn.AdaptiveAvgPool2d and F.interpolate(bilinear, align_corners=True) on the CPU: pooled features, the concat
    [priors..., feats] and the gradients w.r.t. feats and the stage outputs (the two feats branches joined in one launch)."""
    from layout2img_amd import ops
    from layout2img_amd.generator import psp_taps
    B, H, C, Fo = case
    sizes = (1, 2, 3, 6)
    g = torch.Generator().manual_seed(H + C)
    feats = torch.randn(B, H, H, C, generator=g)
    mix = [torch.randn(C, Fo, generator=g) / C ** 0.5 for _ in sizes]   # stands in for the per-stage conv / BN / ReLU
    gcat = torch.randn(B, H, H, len(sizes) * Fo + C, generator=g)

    fr = feats.clone().requires_grad_(True)
    fn = fr.permute(0, 3, 1, 2)
    priors, ys_ref = [], []
    for s, m in zip(sizes, mix):
        pooled = F.adaptive_avg_pool2d(fn, (s, s))                                    # (B,C,s,s)
        y = torch.relu(torch.einsum("bcij,cf->bfij", pooled, m))
        ys_ref.append(y)
        priors.append(F.interpolate(y, size=(H, H), mode="bilinear", align_corners=True))
    ref = torch.cat(priors + [fn], dim=1).permute(0, 2, 3, 1)
    (ref * _rt(gcat, dt)).sum().backward()

    dev = _dev()
    taps = psp_taps(H, sizes, dev)
    fg = feats.to(dev).requires_grad_(True)
    j = ops.GradJoin()
    pooled = ops.psp_pool(fg, taps, j)
    ys = [torch.relu(part @ m.to(dev)) for part, m in zip(pooled.split([s * s for s in sizes], dim=1), mix)]
    cat = ops.psp_expand(fg, torch.cat(ys, dim=1), taps, dt, j)
    assert cat.dtype == dt and cat.shape == ref.shape
    cat.backward(gcat.to(dev).to(dt))
    bf = dt == torch.bfloat16
    assert float((cat.detach().float().cpu() - ref.detach()).abs().max()) < (3e-2 if bf else 1e-5) * float(ref.abs().max())
    assert float((fg.grad.cpu() - fr.grad).abs().max()) < (1e-4 if bf else 2e-5) * float(fr.grad.abs().max())


@pytest.mark.parametrize("case", [(3, 8, 8), (2, 8, 16), (2, 5, 32), (1, 8, 64), (2, 31, 16)])
def test_stage_mask_matches_the_composed_torch_ops(case):
    """ops.stage_mask == reference model/resnet_generator_app_v2.py:465-470 written with torch ops on the CPU (gather,
    sigmoid, nearest / bilinear F.interpolate, blend), forward and the three gradients; repeated classes inside one image
    (their logit gradients add up) and label-0 padding objects included."""
    from layout2img_amd import ops
    B, O, H = case
    g = torch.Generator().manual_seed(31 + H + O)
    Cp, S = 184, 64
    logits = torch.randn(B, H, H, Cp, generator=g)
    bmask = torch.rand(B, O, S, S, generator=g)
    boxm = (torch.rand(B, O, S, S, generator=g) > 0.4).float()
    alpha = torch.randn(1, Cp, 1, generator=g)
    y = torch.randint(0, Cp, (B, O), generator=g)
    y[:, -1] = 0
    if O > 2:
        y[:, 1] = y[:, 0]
    dy = torch.randn(B, O, H, H, generator=g)

    lr, br, ar = (t.clone().requires_grad_(True) for t in (logits, bmask, alpha))
    idx = y.view(B, 1, 1, O).expand(B, H, H, O)
    seman = torch.sigmoid(torch.gather(lr, 3, idx)).permute(0, 3, 1, 2) * F.interpolate(boxm, size=(H, H), mode="nearest")
    a = torch.gather(torch.sigmoid(ar).expand(B, -1, -1), dim=1, index=y.view(B, O, 1)).unsqueeze(-1)
    rb = br if H == S else F.interpolate(br, size=(H, H), mode="bilinear")
    ref = rb * (1 - a) + seman * a
    ref.backward(dy)

    dev = _dev()
    lg, bg, ag = (t.to(dev).requires_grad_(True) for t in (logits, bmask, alpha))
    out = ops.stage_mask(lg, bg, boxm.to(dev), ag, y.to(dev))
    out.backward(dy.to(dev))
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 2e-6
    for got, want in ((lg.grad, lr.grad), (bg.grad, br.grad), (ag.grad, ar.grad)):
        assert float((got.cpu() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("n_live", [0, 3, 9, 20])
@pytest.mark.parametrize("case", [(20, 8, 8, 64, 136, 3, False, False), (20, 8, 8, 128, 64, 3, False, True),
                                  (20, 8, 8, 72, 128, 1, False, True), (20, 8, 8, 16, 40, 3, False, False),
                                  # the ROI-head tile rule (Ci >= 256): 128x64 tiles, 256x64 tiles, and the 4 -> 8 data gradient
                                  (20, 8, 8, 256, 512, 3, False, False), (20, 8, 8, 256, 1024, 3, False, True),
                                  (20, 4, 4, 256, 1024, 3, True, False)])
def test_conv_device_side_image_count(case, n_live, epi):
    """`nimg`: the ROI heads run over the first *nimg images only (device-side count, fixed launch shape): live images
    equal the full convolution, every row of a dead image is exactly zero -- forward/dgrad kernel (halo and generic
    tiles, pooled and not) and weight gradient."""
    from layout2img_amd import ops
    B, H, W, Ci, Co, KH, up2, pool2 = case
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(17)
    x = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    w = _rt(torch.randn(Co, Ci, KH, KH, generator=g) / math.sqrt(Ci * KH * KH), dt)
    bias = torch.randn(Co, generator=g)
    ref = _ref_conv(x, w, bias, up2, pool2)
    res = torch.randn(ref.shape, generator=g)
    pack, kpad = _pack(w, 64)
    nimg = torch.tensor([n_live], dtype=torch.int32, device=_dev())
    out, out_op, _ = ops.conv_raw(x.to(_dev(), dt), pack.to(_dev(), dt), kpad, Co, KH, bias=bias.to(_dev()), res=res.to(_dev()),
                                  up2=up2, pool2=pool2, alpha=0.25 if pool2 else 1.0, want_op=True, relu_op=True, nimg=nimg)
    expect = ref + res
    expect[n_live:] = 0
    assert float((out.cpu() - expect).abs().max()) < 3e-5 * float(expect.abs().max()) + 1e-5
    assert float(out[n_live:].abs().max() if n_live < B else 0.0) == 0.0
    assert float(out_op[n_live:].float().abs().max() if n_live < B else 0.0) == 0.0
    if Co % 8 == 0:
        wz = torch.zeros(Co, Ci, KH, KH, requires_grad=True)
        y = _ref_conv(x[:n_live], wz, None, up2, pool2) if n_live else None
        dy = _rt(torch.randn(ref.shape, generator=g), dt)
        K = KH * KH * Ci
        refw = torch.zeros(Co, K)
        if n_live:
            y.backward(dy[:n_live])
            refw = wz.grad.permute(0, 2, 3, 1).reshape(Co, -1)
        dw = torch.zeros(Co, K, device=_dev())
        # rows of dead images hold garbage on purpose: they must not be read
        xg, dyg = x.clone(), dy.clone()
        xg[n_live:] = float("nan")
        dyg[n_live:] = float("nan")
        ops.wgrad_raw(xg.to(_dev(), dt), dyg.to(_dev(), dt), dw, K, Co, KH, up2=up2, pool2=pool2, alpha=0.25 if pool2 else 1.0, nimg=nimg)
        assert float((dw.cpu() - refw).abs().max()) < 5e-5 * float(refw.abs().max()) + 1e-5


def test_roi_rows_are_compacted_in_reference_order():
    """prepare_layout: valid rows first -- large ROIs, then small ones, original order within each
    (model/rcnn_discriminator_app.py:131-146, 413-417) -- padding behind, count on the device."""
    import layout2img_amd as L
    d = L.CombineDiscriminator128_app(num_classes=10)
    bbox = torch.tensor([[[0.1, 0.1, 0.2, 0.2], [-0.6, -0.6, 0.5, 0.5], [0.0, 0.0, 0.9, 0.9], [0.3, 0.3, 0.5, 0.1]],
                         [[0.2, 0.2, 0.1, 0.6], [0.5, 0.5, 0.3, 0.3], [-0.6, -0.6, 0.5, 0.5], [-0.6, -0.6, 0.5, 0.5]]])
    label = torch.tensor([[3, 0, 4, 5], [6, 7, 0, 0]])
    rois, y, valid, count = d.prepare_layout(bbox.to(_dev()), label.to(_dev()), 128, _dev())
    # large (w or h >= 64 px): (img0: 0.9x0.9), (img0: 0.5 wide = 64 px), (img1: 0.6 tall); small: img0 #0, img1 #1
    assert y.tolist()[:5] == [4, 5, 6, 3, 7] and valid.tolist() == [1, 1, 1, 1, 1, 0, 0, 0] and int(count) == 5
    assert rois[:5, 0].tolist() == [0.0, 0.0, 1.0, 0.0, 1.0]


# (B, H, W, Ci, Co, up2, pool2, sc_Ci | None, sc_up2, live | None): under-filled grids with a long reduction
SPLIT_CASES = [(2, 16, 16, 256, 72, False, False, None, False, None), (3, 8, 8, 512, 264, False, True, 128, False, 2),
               (2, 16, 16, 320, 128, False, False, 64, True, None), (4, 4, 4, 256, 128, True, False, None, False, None),
               (2, 8, 8, 1024, 136, False, False, None, False, None), (3, 16, 16, 256, 64, False, True, 128, False, None),
               (5, 8, 8, 384, 192, False, False, None, False, 3), (1, 32, 32, 256, 64, False, False, None, False, None)]


@pytest.mark.parametrize("cfg", [-1, 14, 15, 17, 18, 19, 29])
@pytest.mark.parametrize("case", SPLIT_CASES)
def test_conv_split_k_stored_partials_carry_the_whole_epilogue(case, cfg, epi):
    """Split-K by stores (conv_store_partial -> conv_split_reduce_kernel): a launch whose tiles fill less than 3/4 of the resident
    workgroup slots is split along K, the partial tiles go to the stream's scratch in register order and the reduce kernel applies
    the WHOLE epilogue -- alpha, both biases, ReLU mask, residual, 2x2 pool / upsampling, operand copies, batch statistics,
    live-row count, folded 1x1 shortcut (its K-steps ride on the last split) -- on every halo tile shape, both epilogue forms;
    against the torch f32 convolution (the layers of reference model/resnet_generator_app_v2.py:411-412 res1 / res2 and
    model/rcnn_discriminator_app.py:93-94 block5 / block6 in miniature)."""
    from layout2img_amd import ops, _lib
    B, H, W, Ci, Co, up2, pool2, sCi, sup, live = case
    if cfg in (17, 18, 19, 29) and Ci % 64:
        pytest.skip("the 256-pixel tiles need Ci % 64 == 0")
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(31)
    x = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    w = _rt(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9), dt)
    bias = torch.randn(Co, generator=g)
    ref = _ref_conv(x, w, bias, up2, pool2)
    dev = _dev()
    co_p = (Co + 7) // 8 * 8
    pad = lambda b: torch.nn.functional.pad(b, (0, co_p - Co)).to(dev)
    pack, kpad = _pack(w, 64)
    kw = {}
    Ho, Wo = (2 * H, 2 * W) if up2 else (H, W)
    if sCi is not None:
        hs, ws_ = (Ho // 2, Wo // 2) if sup else (Ho, Wo)
        xs = _rt(torch.randn(B, hs, ws_, sCi, generator=g), dt)
        wsc = _rt(torch.randn(Co, sCi, 1, 1, generator=g) / math.sqrt(sCi), dt)
        bias_sc = torch.randn(Co, generator=g)
        ref = ref + _ref_conv(xs, wsc, bias_sc, sup, pool2)
        pack_sc, kpad_sc = _pack(wsc, 64)
        placeholder = torch.full(ref.shape[:3] + (co_p,), float("nan"), device=dev)
        kw["sc"] = dict(x_op=xs.to(dev, dt), wpack=pack_sc.to(dev, dt), kpad=kpad_sc, bias=pad(bias_sc), up2=sup, out=placeholder, flops=0.0)
    else:
        mask = _rt(torch.randn(ref.shape, generator=g), dt)
        res = torch.randn(ref.shape, generator=g)
        ref = ref * (mask > 0).float() + res
        padc = lambda t: torch.nn.functional.pad(t, (0, co_p - Co)).contiguous()
        kw["relu_mask"], kw["res"] = padc(mask).to(dev, dt), padc(res).to(dev)
    nimg = None
    if live is not None:
        nimg = torch.tensor([live], dtype=torch.int32, device=dev)
        ref[live:] = 0
    _lib.call("l2i_set_conv_config", cfg)
    try:
        out, op, raw = ops.conv_raw(x.to(dev, dt), pack.to(dev, dt), kpad, co_p, 3, bias=pad(bias), up2=up2, pool2=pool2,
                                    alpha=0.25 if pool2 else 1.0, nimg=nimg, want_op=True, relu_op=True, want_raw=True, stats=live is None, **kw)
        splits = _lib.load().l2i_debug_occupancy(100, 0)
    finally:
        _lib.call("l2i_set_conv_config", -1)
    if cfg >= 10:   # a forced halo tile on these shapes is under-filled with >= 4 chunks: it must have taken the stored-partials path
        assert splits > 1, splits
    tol = 3e-5 * float(ref.abs().max())
    o = out.cpu()[..., :Co]
    assert float((o - ref).abs().max()) < tol
    assert float((raw.float().cpu() - _rt(out.cpu(), dt)).abs().max()) == 0.0
    assert float((op.float().cpu() - _rt(out.cpu().clamp_min(0), dt)).abs().max()) == 0.0
    if sCi is not None and cfg >= 10 and sCi % 64 == 0 and cfg not in (17, 18):   # (17 / 18: the tiles without a folding twin)
        assert bool(torch.isnan(kw["sc"]["out"]).all())   # folded on the last split: the placeholder is never written
    if live is None:
        s1, s2, _ = out._l2i_stats
        od = out.double().cpu().view(-1, co_p)
        assert float((s1.cpu().double().view(-1) - od.sum(0)).abs().max()) < 1e-5 * float(od.abs().sum(0).max()) + 1e-4
        assert float((s2.cpu().double().view(-1) - (od * od).sum(0)).abs().max()) < 1e-5 * float((od * od).sum(0).max()) + 1e-4


@pytest.mark.parametrize("cfg", [40, 41])
@pytest.mark.parametrize("case", [(4, 32, 32, 128, 264, False, False, None), (6, 16, 16, 192, 256, False, True, 128), (20, 8, 8, 256, 512, False, False, None),
                                  (12, 4, 4, 256, 256, True, False, None), (3, 32, 32, 64, 512, True, True, 64), (2, 64, 64, 64, 136, False, False, None)])
def test_conv_256x256_eight_wave_tile(case, cfg, epi):
    """conv_halo8_kernel (round-5 experiment: one workgroup of eight waves per CU, 256 pixels x 256 channels, double-buffered halo,
    weights in half-K-step tiles; cfg 41: three half-tiles in flight with counted vmcnt, two barriers per K-step) on bordered and
    compact halos, upsampling, pool, ragged channel counts, odd chunk counts and with a folded shortcut -- against torch f32."""
    from layout2img_amd import ops, _lib
    B, H, W, Ci, Co, up2, pool2, sCi = case
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(77)
    x = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    w = _rt(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9), dt)
    bias = torch.randn(Co, generator=g)
    ref = _ref_conv(x, w, bias, up2, pool2)
    dev = _dev()
    pack, kpad = _pack(w, 64)
    kw = {}
    if sCi is not None:
        Ho, Wo = (2 * H, 2 * W) if up2 else (H, W)
        xs = _rt(torch.randn(B, Ho, Wo, sCi, generator=g), dt)
        wsc = _rt(torch.randn(Co, sCi, 1, 1, generator=g) / math.sqrt(sCi), dt)
        bias_sc = torch.randn(Co, generator=g)
        ref = ref + _ref_conv(xs, wsc, bias_sc, False, pool2)
        pack_sc, kpad_sc = _pack(wsc, 64)
        kw["sc"] = dict(x_op=xs.to(dev, dt), wpack=pack_sc.to(dev, dt), kpad=kpad_sc, bias=bias_sc.to(dev), up2=False,
                        out=torch.full(ref.shape, float("nan"), device=dev), flops=0.0)
    else:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res
        kw["res"] = res.to(dev)
    _lib.call("l2i_set_conv_config", cfg)
    try:
        out, op, _ = ops.conv_raw(x.to(dev, dt), pack.to(dev, dt), kpad, Co, 3, bias=bias.to(dev), up2=up2, pool2=pool2,
                                  alpha=0.25 if pool2 else 1.0, want_op=True, relu_op=True, **kw)
    finally:
        _lib.call("l2i_set_conv_config", -1)
    assert float((out.cpu() - ref).abs().max()) < 3e-5 * float(ref.abs().max())
    assert float((op.float().cpu() - _rt(out.cpu().clamp_min(0), dt)).abs().max()) == 0.0
    if sCi is not None:
        assert bool(torch.isnan(kw["sc"]["out"]).all())   # folded


# (B, H, W, Ci = channels of dh, Co = channels of x, sc_Ci = channels of dy, shortcut pooled, live images | None)
DGRAD_SC_CASES = [(2, 32, 32, 128, 64, 128, True, None), (3, 16, 16, 256, 128, 256, True, None), (2, 32, 32, 256, 128, 256, False, None),
                  (5, 8, 8, 512, 256, 512, True, 3), (1, 64, 64, 64, 64, 128, True, None), (2, 16, 16, 128, 72, 192, False, None)]


@pytest.mark.parametrize("cfg", [-1, 14, 15, 19, 29])
@pytest.mark.parametrize("case", DGRAD_SC_CASES)
def test_conv_dgrad_with_the_shortcut_data_gradient_folded_in(case, cfg, epi):
    """l2i_conv2d_dgrad_sc: dx of a pre-activation discriminator block (reference model/rcnn_discriminator_app.py:326,336-341) from ONE launch:
    relu'(x) . conv3x3(dh, W1 flipped) + sc_alpha conv1x1(dy at (y >> 1, x >> 1), Wsc^T) + another reader's gradient -- the ReLU mask applied to
    the accumulators in front of the shortcut's K-steps on the 128-pixel tiles (folded: the placeholder stays NaN), un-folded by the library on the
    256-pixel tiles and for channel counts that are not multiples of 64; with a live-image count; against torch f32."""
    from layout2img_amd import ops, _lib
    B, H, W, Ci, Co, sCi, pooled, live = case
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(41)
    dh = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    w = _rt(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9), dt)
    hs, ws_ = (H // 2, W // 2) if pooled else (H, W)
    dy = _rt(torch.randn(B, hs, ws_, sCi, generator=g), dt)
    wsc = _rt(torch.randn(Co, sCi, 1, 1, generator=g) / math.sqrt(sCi), dt)
    mask = _rt(torch.randn(B, H, W, Co, generator=g), dt)
    res = torch.randn(B, H, W, Co, generator=g)
    sc_alpha = 0.25 if pooled else 1.0
    ref = _ref_conv(dh, w, None, False, False) * (mask > 0).float() + sc_alpha * _ref_conv(dy, wsc, None, pooled, False) + res
    dev = _dev()
    nimg = None
    if live is not None:
        nimg = torch.tensor([live], dtype=torch.int32, device=dev)
        ref[live:] = 0
    pack, kpad = _pack(w, 64)
    pack_sc, kpad_sc = _pack(wsc, 64)
    placeholder = torch.full((B, H, W, Co), float("nan"), device=dev)
    sc = dict(x_op=dy.to(dev, dt), wpack=pack_sc.to(dev, dt), kpad=kpad_sc, bias=None, up2=pooled, alpha=sc_alpha, out=placeholder, flops=0.0, mask_first=True)
    _lib.call("l2i_set_conv_config", cfg)
    try:
        out, _, raw = ops.conv_raw(dh.to(dev, dt), pack.to(dev, dt), kpad, Co, 3, relu_mask=mask.to(dev, dt), res=res.to(dev), nimg=nimg, sc=sc, want_raw=True)
    finally:
        _lib.call("l2i_set_conv_config", -1)
    assert float((out.cpu() - ref).abs().max()) < 3e-5 * float(ref.abs().max())
    assert float((raw.float().cpu() - _rt(out.cpu(), dt)).abs().max()) == 0.0
    if cfg in (14, 15):   # the 128-pixel tiles fold when the shortcut's channel count is a multiple of 64
        assert bool(torch.isnan(placeholder).all()) == (sCi % 64 == 0)
    # (19 / 29, where the heuristic honours them -- not for <= 64 output channels --: the 256-pixel tiles do not compile the mask-first tail
    #  and the library un-folds; same result either way)


@pytest.mark.parametrize("case", [(3, 8, 16, 100, 104, True), (2, 5, 64, 100, 104, True), (4, 8, 8, 100, 104, False), (1, 8, 32, 120, 124, True)])
def test_class_gathered_logits_equal_the_gather_of_the_dense_head(case):
    """ops.class_logits (l2i_class_logits_fwd / _bwd): the last 1x1 convolution of a generator mask head evaluated only for the classes its
    reader gathers (reference model/resnet_generator_app_v2.py:643-651, 465-466) equals gather(conv1x1(a), 1, y) forward and in all three
    gradients (activations incl. zeroed pad channels, weight, bias), with repeated classes and the padding class 0 among the objects."""
    from layout2img_amd import ops
    B, O, H, C, Cp, has_bias = case
    g = torch.Generator().manual_seed(B * 100 + O)
    a = torch.randn(B, H, H, Cp, generator=g)
    a[..., C:] = 0
    w = torch.randn(184, C, generator=g) / math.sqrt(C)
    bias = torch.randn(184, generator=g) if has_bias else None
    y = torch.randint(0, 184, (B, O), generator=g)
    y[:, -1] = y[:, 0]          # a repeated class
    y[0, 1] = 0                 # the padding class
    ar, wr = a.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True) if has_bias else None
    dense = torch.einsum("bhwc,kc->bkhw", ar[..., :C], wr) + (br.view(1, -1, 1, 1) if has_bias else 0)
    ref = torch.gather(dense, 1, y.view(B, O, 1, 1).expand(B, O, H, H))
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    dev = _dev()
    ad, wd = a.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    bd = bias.to(dev).requires_grad_(True) if has_bias else None
    lg = ops.class_logits(ad, wd, bd, y.to(dev))
    assert float((lg.cpu() - ref).abs().max()) < 1e-5 * float(ref.abs().max()) + 1e-6
    lg.backward(go.to(dev))
    tol = lambda r: 2e-5 * float(r.abs().max()) + 1e-6
    assert float((ad.grad.cpu()[..., :C] - ar.grad[..., :C]).abs().max()) < tol(ar.grad)
    assert float(ad.grad.cpu()[..., C:].abs().max()) == 0.0 if Cp > C else True
    assert float((wd.grad.cpu() - wr.grad).abs().max()) < tol(wr.grad)
    if has_bias:
        assert float((bd.grad.cpu() - br.grad).abs().max()) < tol(br.grad)
