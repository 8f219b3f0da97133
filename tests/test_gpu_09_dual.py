"""Dual launches (l2i_conv2d_fwd_dual / l2i_conv2d_wgrad_dual): the discriminator step's two passes -- D(real) and D(fake),
reference train_context_app_v2.py:158,167, each with its own spectral-norm iteration and therefore its own weight packs -- run as
ONE batch of 2b images. Every result must equal the two single-pass launches it replaces: forward / data gradient on every tile
family (incl. the folded 1x1 shortcut, the ReLU mask, residual, 2x2 pool, the live-image count of the ROI heads applied per
half, split-K grids) and the weight gradient into two accumulators (with the folded shortcut and the shared bias gradient);
batches whose halves do not fill whole tiles fall back to two launches inside ops.conv_raw / wgrad_raw and must agree as well.
The module-level test runs the discriminator's forward_dual against two forward_padded passes, outputs and all gradients."""
import math

import pytest
import torch

from tests.test_gpu_02_ops import _pack, _ref_conv, _rt

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# B (both halves), H, W, Ci, Co, KH, up2, pool2, live-per-half | None
DUAL_CASES = [
    (8, 16, 16, 64, 72, 3, False, False, None),     # 128x128 / 128x64 halo tiles
    (4, 32, 32, 128, 136, 3, False, True, None),    # pooled epilogue
    (64, 8, 8, 256, 512, 3, False, False, 20),      # ROI head, 128x64 tiles, live rows per half
    (64, 8, 8, 256, 1024, 3, False, True, 7),       # ROI head, 256x64 tiles, pooled
    (64, 4, 4, 256, 1024, 3, True, False, 11),      # 4 -> 8 data gradient of the pooled ROI conv (compact halo)
    (16, 16, 16, 40, 104, 1, False, False, None),   # generic kernel, 1x1
    (64, 4, 4, 512, 136, 3, False, False, None),    # 4x4 maps: generic kernel with split-K
    (32, 8, 8, 8, 64, 3, False, False, None),       # 8-channel input (generic)
    (2, 8, 8, 64, 64, 3, False, False, None),       # halves smaller than a tile: two launches inside conv_raw
    (6, 4, 4, 64, 64, 3, False, False, 2),          # the same with a live count
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("case", DUAL_CASES)
def test_dual_conv_equals_two_single_launches(case, dt):
    from layout2img_amd import ops
    B, H, W, Ci, Co, KH, up2, pool2, live = case
    hb = B // 2
    g = torch.Generator().manual_seed(B * 131 + Ci)
    x = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    wa = _rt(torch.randn(Co, Ci, KH, KH, generator=g) / math.sqrt(Ci * KH * KH), dt)
    wb = _rt(torch.randn(Co, Ci, KH, KH, generator=g) / math.sqrt(Ci * KH * KH), dt)
    bias = torch.randn(Co, generator=g)
    ref = torch.cat((_ref_conv(x[:hb], wa, bias, up2, pool2), _ref_conv(x[hb:], wb, bias, up2, pool2)))
    mask = _rt(torch.randn(ref.shape, generator=g), dt)
    res = torch.randn(ref.shape, generator=g)
    expect = ref * (mask > 0).float() + res
    nimg = None
    if live is not None:
        nimg = torch.tensor([live], dtype=torch.int32, device=DEV)
        expect[live:hb] = 0
        expect[hb + live:] = 0
    km = 64 if dt == torch.bfloat16 else 32
    (pa, kpad), (pb, _) = _pack(wa, km), _pack(wb, km)
    co_p = (Co + 7) // 8 * 8
    padc = lambda t: torch.nn.functional.pad(t, (0, co_p - Co))
    out, op, _ = ops.conv_raw(x.to(DEV, dt), pa.to(DEV, dt), kpad, co_p, KH, bias=padc(bias).to(DEV), res=padc(res).to(DEV),
                              relu_mask=padc(mask).to(DEV, dt), up2=up2, pool2=pool2, alpha=0.25 if pool2 else 1.0, nimg=nimg,
                              want_op=True, relu_op=True, wpack_b=pb.to(DEV, dt))
    tol = 3e-5 * float(expect.abs().max()) + 1e-5
    assert float((out.cpu()[..., :Co] - expect).abs().max()) < tol
    assert float((op.float().cpu()[..., :Co] - torch.relu(expect)).abs().max()) < (1e-2 if dt == torch.bfloat16 else 1e-4) * float(expect.abs().max())
    if live is not None:
        assert float(out[live:hb].abs().max()) == 0.0 and float(out[hb + live:].abs().max()) == 0.0


@pytest.mark.parametrize("cfg", [-1, 14, 15, 19, 29])
@pytest.mark.parametrize("case", [(8, 16, 16, 64, 72, 128, False, False, None), (4, 32, 32, 128, 136, 64, True, False, None),
                                  (64, 8, 8, 128, 264, 192, False, True, 9), (4, 32, 32, 64, 64, 128, True, False, None),
                                  (2, 64, 64, 64, 64, 128, False, True, None), (2, 8, 8, 64, 128, 64, False, True, None)])
def test_dual_conv_folded_shortcut(case, cfg):
    """conv3x3(h) + conv1x1(x) with BOTH layers' packs doubled: each half of the batch uses its own pair."""
    from layout2img_amd import ops, _lib
    B, H, W, Ci, Co, sCi, sup, pool2, live = case
    hb, dt = B // 2, torch.bfloat16
    g = torch.Generator().manual_seed(23 + B)
    x = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    hs, ws_ = (H // 2, W // 2) if sup else (H, W)
    xs = _rt(torch.randn(B, hs, ws_, sCi, generator=g), dt)
    w = [_rt(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9), dt) for _ in range(2)]
    wsc = [_rt(torch.randn(Co, sCi, 1, 1, generator=g) / math.sqrt(sCi), dt) for _ in range(2)]
    bias, bias_sc = torch.randn(Co, generator=g), torch.randn(Co, generator=g)
    ref = torch.cat([_ref_conv(x[k * hb:(k + 1) * hb], w[k], bias, False, pool2) + _ref_conv(xs[k * hb:(k + 1) * hb], wsc[k], bias_sc, sup, pool2)
                     for k in range(2)])
    nimg = None
    if live is not None:
        nimg = torch.tensor([live], dtype=torch.int32, device=DEV)
        ref[live:hb] = 0
        ref[hb + live:] = 0
    packs = [_pack(t, 64) for t in w]
    packs_sc = [_pack(t, 64) for t in wsc]
    co_p = (Co + 7) // 8 * 8
    pad = lambda b: torch.nn.functional.pad(b, (0, co_p - Co)).to(DEV)
    placeholder = torch.full(ref.shape[:3] + (co_p,), float("nan"), device=DEV)
    sc = dict(x_op=xs.to(DEV, dt), wpack=packs_sc[0][0].to(DEV, dt), wpack_b=packs_sc[1][0].to(DEV, dt), kpad=packs_sc[0][1], bias=pad(bias_sc),
              up2=sup, out=placeholder, flops=0.0)
    _lib.call("l2i_set_conv_config", cfg)
    try:
        out, op, _ = ops.conv_raw(x.to(DEV, dt), packs[0][0].to(DEV, dt), packs[0][1], co_p, 3, bias=pad(bias), pool2=pool2,
                                  alpha=0.25 if pool2 else 1.0, nimg=nimg, sc=sc, want_op=True, relu_op=True, wpack_b=packs[1][0].to(DEV, dt))
    finally:
        _lib.call("l2i_set_conv_config", -1)
    assert float((out.cpu()[..., :Co] - ref).abs().max()) < 3e-5 * float(ref.abs().max())
    assert float((op.float().cpu()[..., :Co] - torch.relu(ref)).abs().max()) < 1e-2 * float(ref.abs().max())
    if cfg >= 10 and ops.dual_conv_ok(B, H, W):   # a forced halo tile with sc_Ci % 64 == 0 folds: the placeholder is never written
        assert bool(torch.isnan(placeholder).all()) == (sCi % 64 == 0)


@pytest.mark.parametrize("combine", ["scratch", "atomics"])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("case", [(8, 16, 16, 64, 128, 3, False, False, None, 128), (4, 32, 32, 128, 136, 3, False, True, None, 64),
                                  (64, 8, 8, 192, 264, 3, False, True, 9, 256), (64, 8, 8, 64, 128, 3, False, False, 0, None),
                                  (6, 32, 32, 64, 128, 1, False, False, None, None), (4, 16, 16, 64, 136, 3, True, False, None, None),
                                  (2, 4, 4, 64, 64, 3, False, False, None, None), (10, 8, 8, 72, 64, 3, False, False, 3, None)])
def test_dual_wgrad_two_accumulators(case, dt, combine, monkeypatch):
    """dW of the first half into dw, of the second half into dw_b (+ the folded shortcut's two accumulators), ONE bias gradient."""
    from layout2img_amd import ops, _lib
    B, H, W, Ci, Co, KH, up2, pool2, live, sCi = case
    hb = B // 2
    g = torch.Generator().manual_seed(hash(case[:6]) % 1000 + 3)
    x = _rt(torch.randn(B, H, W, Ci, generator=g), dt)
    xs = _rt(torch.randn(B, H << int(up2), W << int(up2), sCi, generator=g), dt) if sCi else None
    ws_ = [torch.zeros(Co, Ci, KH, KH, requires_grad=True) for _ in range(2)]
    wsc = [torch.zeros(Co, sCi, 1, 1, requires_grad=True) for _ in range(2)] if sCi else None
    b1 = torch.zeros(Co, requires_grad=True)
    ys = []
    for k in range(2):
        sl = slice(k * hb, (k + 1) * hb)
        y = _ref_conv(x[sl], ws_[k], b1, up2, pool2)
        if sCi:
            y = y + _ref_conv(xs[sl], wsc[k], None, False, pool2)
        ys.append(y)
    y = torch.cat(ys)
    dy = _rt(torch.randn(y.shape, generator=g), dt)
    if live is not None:
        dy[live:hb] = 0
        dy[hb + live:] = 0
    y.backward(dy)
    K = KH * KH * Ci
    dw, dwb = torch.full((Co, K), 0.5, device=DEV), torch.full((Co, K), -1.0, device=DEV)
    db = torch.zeros(Co, device=DEV)
    nimg = torch.tensor([live], dtype=torch.int32, device=DEV) if live is not None else None
    sc = None
    if sCi:
        dws, dwsb = torch.zeros(Co, sCi, device=DEV), torch.full((Co, sCi), 2.0, device=DEV)
        sc = dict(x_op=xs.to(DEV, dt), dw=dws, dw_b=dwsb, ldw=sCi, dbias=None, flops=0.0, up2=False)
    if combine == "atomics":
        monkeypatch.setattr(_lib, "wgrad_scratch", lambda device: (None, 0))
    xg, dyg = x.clone(), dy.clone()
    if live is not None:   # rows of dead images hold garbage on purpose: they must not be read
        for t in (xg, dyg):
            t[live:hb] = float("nan")
            t[hb + live:] = float("nan")
    ops.wgrad_raw(xg.to(DEV, dt), dyg.to(DEV, dt), dw, K, Co, KH, up2=up2, pool2=pool2, alpha=0.25 if pool2 else 1.0, nimg=nimg, dbias=db,
                  sc=sc, dw_b=dwb)
    tol = lambda r: 2e-4 * float(r.abs().max()) + 1e-5
    flat = lambda w_: w_.grad.permute(0, 2, 3, 1).reshape(Co, -1)
    assert float((dw.cpu() - 0.5 - flat(ws_[0])).abs().max()) < tol(flat(ws_[0]))
    assert float((dwb.cpu() + 1.0 - flat(ws_[1])).abs().max()) < tol(flat(ws_[1]))
    assert float((db.cpu() - b1.grad).abs().max()) < tol(b1.grad)
    if sCi:
        assert float((dws.cpu() - wsc[0].grad.reshape(Co, sCi)).abs().max()) < tol(wsc[0].grad)
        assert float((dwsb.cpu() - 2.0 - wsc[1].grad.reshape(Co, sCi)).abs().max()) < tol(wsc[1].grad)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("size,b", [(128, 2), (128, 8), (64, 4)])
def test_forward_dual_equals_two_padded_passes(size, b, dt):
    """CombineDiscriminator*.forward_dual(real, fake) against forward_padded(real) + forward_padded(fake) from the same state:
    outputs of both passes, the flat parameter gradient (after the spectral-norm backward of both passes) and the image
    gradients. (b = 2: most layers' halves are smaller than a tile and run as two launches; b = 8: dual launches throughout.)"""
    import layout2img_amd as L
    from layout2img_amd import ops
    from layout2img_amd.synthetic import make_batch
    torch.manual_seed(5)
    net = (L.CombineDiscriminator128_app if size == 128 else L.CombineDiscriminator64)(num_classes=184).finalize(DEV, dt)
    net.train()
    real, label, bbox, _, _ = make_batch(b, size, "coco", seed=21, device=DEV)
    fake = torch.randn_like(real).clamp_(-1, 1)
    sn0 = net.arena.sn_flat.data.clone()
    n_out = 3 if size == 128 else 2
    wts = [torch.randn(1, device=DEV) for _ in range(2 * n_out)]

    def run(dual):
        net.arena.sn_flat.data.copy_(sn0)
        net.arena.drop_pending()
        net.zero_grad()
        ra, fa = real.clone().requires_grad_(True), fake.clone().requires_grad_(True)
        with ops.POOL.step(DEV):
            if dual:
                oa, ob, valid, _ = net.forward_dual(ra, fa, bbox, label)
            else:
                *oa, valid, _ = net.forward_padded(ra, bbox, label)
                *ob, _, _ = net.forward_padded(fa, bbox, label)
            vm = valid.float().view(-1, 1)
            loss = 0
            for k, o in enumerate(list(oa) + list(ob)):
                loss = loss + wts[k] * ((o * vm).sum() if o.shape[0] == vm.shape[0] else o.sum())
            loss.backward()
            net.arena.flush_grads()
        torch.cuda.synchronize()
        return [t.detach().clone() for t in list(oa) + list(ob)], net.flat.grad.clone(), ra.grad.clone(), fa.grad.clone(), net.arena.sn_flat.data.clone()

    o1, g1, ra1, fa1, sn1 = run(False)
    o2, g2, ra2, fa2, sn2 = run(True)
    assert float((sn1 - sn2).abs().max()) < 1e-5   # the same two power iterations, in the same order (their sums are atomics: not bit-identical)
    f32 = dt == torch.float32
    for a_, b_ in zip(o1, o2):   # forward: every output of both passes (a wrong pack would show as ~1e-3: the two sigmas differ by that much)
        assert a_.shape == b_.shape
        assert float((a_ - b_).abs().max()) <= (2e-5 if f32 else 2e-2) * max(1.0, float(a_.abs().max()))
    # backward: the forwards agree to ~1e-6, but a pre-activation within that distance of 0 lands on the other side of its ReLU
    # gate in one of the two runs and changes that IMAGE's gradient by ~1e-3 (tools/parity/dual_debug.py: the discrepancy moves
    # between images and configurations from run to run) -- hence per image: most images tight, every image loose
    assert float((g1 - g2).norm() / g1.norm()) < (1e-3 if f32 else 1e-2)
    for name, a_, b_ in (("first pass's images", ra1, ra2), ("second pass's images", fa1, fa2)):
        per = ((a_ - b_).flatten(1).norm(dim=1) / a_.flatten(1).norm(dim=1)).cpu()
        assert float(per.sort().values[(len(per) - 1) // 2]) < (1e-4 if f32 else 6e-2), (name, per.tolist())   # (lower median)
        assert float(per.max()) < (2e-2 if f32 else 3e-1), (name, per.tolist())
