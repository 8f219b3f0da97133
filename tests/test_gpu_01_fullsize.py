"""Parity AT THE HEADLINE SHAPES (BASELINE config 3: 128x128, batch 32, COCO layouts): what `bench.py` times is compared with
the oracle here, not only its batch-of-2 miniature.

(i)   generator forward, batch 32, recipe weights: exact-f32 operands and the split "bf16x3" mode against
      oracle.generator_forward on the CPU at the north star's bar (image L_inf < 1e-3, pre-tanh tap 1e-3 relative); the
      bf16-operand throughput mode at its own stated bar (L_inf 8.5e-2, rms 7.6e-3: 1.25 x the now run-to-run identical measured values).
(ii)  discriminator forward on the same batch (the 157-of-256 live ROI layout of synthetic.make_batch) against
      oracle.discriminator_forward, rows in the reference's output order.
(iii) EVERY distinct conv / data-gradient / weight-gradient launch of one full-size training iteration, captured live from
      GanTrainer.step (so a new tile heuristic, fold or epilogue option is covered the moment the trainer uses it), re-issued
      through ops.conv_raw / ops.wgrad_raw on pre-rounded random operands and compared with torch's f32 convolution on the CPU:
      accumulation order is the only difference, the bars are those of tests/test_gpu_02_ops.py (3e-5 / 2e-4 of the result's max).
      The captured signature list is written to gpurun_out/headline_launches.json.
"""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import model as O
from tests.helpers import fixture_state, load_fixture, maxdiff

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BATCH, SEED = 32, 1234   # bench.py's workload


def _batch():
    from layout2img_amd.synthetic import make_batch
    return make_batch(BATCH, 128, "coco", seed=SEED, device="cpu")


@pytest.fixture(scope="module")
def oracle_g():
    """oracle.generator_forward on the bench batch (train mode: batch statistics, one power iteration), once per module."""
    real, label, bbox, z, z_im = _batch()
    sd = fixture_state(load_fixture("g_coco.npz"), 11)
    taps = {}
    with torch.no_grad():
        img = O.generator_forward(sd, z, bbox, z_im, label, training=True, dropout_p=0.0, taps=taps)
    return img, taps["pre_tanh"]


@pytest.fixture(scope="module")
def oracle_d():
    real, label, bbox, z, z_im = _batch()
    sd = fixture_state(load_fixture("d_coco.npz"), 21)
    with torch.no_grad():
        return O.discriminator_forward(sd, real, bbox, label, training=True)


def _net_g(mode):
    import layout2img_amd as L
    torch.manual_seed(0)
    g = L.ResnetGenerator128_context(num_classes=184, output_dim=3)
    g.load_state_dict(fixture_state(load_fixture("g_coco.npz"), 11))
    g.finalize(DEV, mode)
    for m in g.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
    return g.train()


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16"])
def test_generator_forward_full_size_vs_oracle(mode, oracle_g):
    g = _net_g({"f32": torch.float32, "bf16": torch.bfloat16}.get(mode, mode))
    real, label, bbox, z, z_im = (t.to(DEV) for t in _batch())
    taps = {}
    with torch.no_grad():
        img = g(z, bbox, z_im, label, taps=taps)
    ref_img, ref_pre = oracle_g
    e_img = maxdiff(img, ref_img)
    e_pre = maxdiff(taps["pre_tanh"].permute(0, 3, 1, 2)[:, :3], ref_pre) / float(ref_pre.abs().max())
    rms = float((img.detach().cpu().float() - ref_img).pow(2).mean().sqrt())
    print(f"full-size G forward [{mode}]: image L_inf {e_img:.2e} (rms {rms:.2e}), pre-tanh rel {e_pre:.2e}")
    assert tuple(img.shape) == (BATCH, 3, 128, 128) and bool(torch.isfinite(img).all())
    if mode == "bf16":   # the throughput mode's own bar (2^-9 per operand pair, a random walk over ~25 layers).
        # Rounds 1-5: the maximum over 1.5 M pixels moved from run to run (6.2e-2 ... > 8e-2) with the order of the atomic batch statistics and
        # power-iteration sums, and the ceiling had to be 1.2e-1. Round 6: the forward is bit-identical from run to run (tests/test_gpu_06b_
        # determinism.py); measured on an MI355X: L_inf 6.59e-2, rms 6.07e-3, pre-tanh 1.70e-2 -- bars 1.25x that.
        assert e_img < 8.5e-2 and e_pre < 2.2e-2 and rms < 7.6e-3, (e_img, e_pre, rms)
    else:                # the north star's bar
        assert e_img < 1e-3 and e_pre < 1e-3, (e_img, e_pre)


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16"])
def test_discriminator_forward_full_size_vs_oracle(mode, oracle_d):
    import layout2img_amd as L
    torch.manual_seed(0)
    d = L.CombineDiscriminator128_app(num_classes=184)
    d.load_state_dict(fixture_state(load_fixture("d_coco.npz"), 21))
    d.finalize(DEV, {"f32": torch.float32, "bf16": torch.bfloat16}.get(mode, mode)).train()
    real, label, bbox, z, z_im = (t.to(DEV) for t in _batch())
    with torch.no_grad():
        outs = d(real, bbox, label)
    n_live = int((label != 0).sum())
    assert n_live == 157   # the live-row count the ROI-head launches of the bench batch see (of 256 slots)
    rel = 3e-2 if mode == "bf16" else 2e-4
    for t, ref, k in zip(outs, oracle_d, ("img", "obj", "app")):
        assert tuple(t.shape) == tuple(ref.shape), (k, t.shape, ref.shape)
        e, s = maxdiff(t, ref), max(1.0, float(ref.abs().max()))
        print(f"full-size D forward [{mode}] {k}: {e:.2e} (scale {s:.2e})")
        assert e < rel * s, (k, e, s)


# ----------------------------------------------------------------------------- (iii) every launch of the iteration
def _sig_conv(x_op, wpack, kpad, co, kh, *, bias=None, res=None, relu_mask=None, up2=False, pool2=False, alpha=1.0,
              want_f32=True, want_op=False, relu_op=False, want_raw=False, flops=None, nimg=None, stats=False, sc=None,
              wpack_b=None, _outs=None):
    scs = None if sc is None else (tuple(sc["x_op"].shape), int(sc["kpad"]), sc.get("bias") is not None, bool(sc["up2"]),
                                   bool(sc.get("mask_first")), float(sc.get("alpha", 0.0)))
    return ("conv", tuple(x_op.shape), str(x_op.dtype), int(kpad), int(co), int(kh), bias is not None, res is not None,
            relu_mask is not None, bool(up2), bool(pool2), float(alpha), bool(want_f32), bool(want_op), bool(relu_op),
            bool(want_raw), nimg is not None, bool(stats), scs, wpack_b is not None)


def _sig_wgrad(x_op, dy_op, dw, ldw, co, kh, *, up2=False, pool2=False, alpha=1.0, flops=None, nimg=None, dbias=None, sc=None,
               dw_b=None, overwrite=False):
    scs = None if sc is None else (tuple(sc["x_op"].shape), int(sc["ldw"]), sc["dbias"] is not None, bool(sc.get("up2", False)))
    return ("wgrad", tuple(x_op.shape), str(x_op.dtype), tuple(dy_op.shape), int(ldw), int(co), int(kh), bool(up2), bool(pool2),
            float(alpha), nimg is not None, dbias is not None, scs, dw_b is not None, bool(overwrite))


@pytest.fixture(scope="module")
def headline_launches():
    """The distinct conv_raw / wgrad_raw calls of ONE eager training iteration at the bench configuration."""
    import layout2img_amd as L
    from layout2img_amd import ops
    torch.manual_seed(SEED)
    netG = L.ResnetGenerator128_context(num_classes=184).finalize(DEV, torch.bfloat16)
    netD = L.CombineDiscriminator128_app(num_classes=184).finalize(DEV, torch.bfloat16)
    tr = L.GanTrainer(netG, netD)
    real, label, bbox, z, z_im = (t.to(DEV) for t in _batch())
    seen = {}
    conv0, wgrad0 = ops.conv_raw, ops.wgrad_raw

    def conv(*a, **k):
        s = _sig_conv(*a, **k)
        seen[s] = seen.get(s, 0) + 1
        return conv0(*a, **k)

    def wgrad(*a, **k):
        s = _sig_wgrad(*a, **k)
        seen[s] = seen.get(s, 0) + 1
        return wgrad0(*a, **k)
    ops.conv_raw, ops.wgrad_raw = conv, wgrad
    try:
        tr.step(real, label, bbox, z, None)
        torch.cuda.synchronize()
    finally:
        ops.conv_raw, ops.wgrad_raw = conv0, wgrad0
    live = int((label != 0).sum())
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        json.dump(dict(live_rois=live, launches=[dict(sig=list(map(str, s)), count=n) for s, n in seen.items()]),
                  open(os.path.join(out, "headline_launches.json"), "w"), indent=0)
    except OSError:
        pass
    del tr, netG, netD
    torch.cuda.empty_cache()
    return seen, live


def _rt(t):
    return t.to(torch.bfloat16).float()


def _pack(w, kpad):
    co, ci, kh, _ = w.shape
    k = kh * kh * ci
    p = torch.zeros((co + 127) // 128 * 128, kpad)
    p[:co, :k] = w.permute(0, 2, 3, 1).reshape(co, k)
    return p


def _conv_ref(x_nhwc, w, up2, pool2):
    x = x_nhwc.permute(0, 3, 1, 2)
    if up2:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    y = F.conv2d(x, w, None, 1, w.shape[2] // 2)
    if pool2:
        y = F.avg_pool2d(y, 2) * 4.0   # (the kernel sums the quad and scales by alpha)
    return y.permute(0, 2, 3, 1).contiguous()


def _replay_conv(sig, live_frac, seed):
    from layout2img_amd import ops
    (_, xs, dts, kpad, co, kh, has_bias, has_res, has_mask, up2, pool2, alpha, want_f32, want_op, relu_op, want_raw, has_nimg,
     stats, scs, dual) = sig
    assert dts == "torch.bfloat16" and not dual
    B, Hi, Wi, Ci = xs
    g = torch.Generator().manual_seed(seed)
    x = _rt(torch.randn(xs, generator=g))
    w = _rt(torch.randn(co, Ci, kh, kh, generator=g) / math.sqrt(Ci * kh * kh))
    ref = alpha * _conv_ref(x, w, up2, pool2)
    dev = DEV
    bf = torch.bfloat16
    kw = dict(up2=up2, pool2=pool2, alpha=alpha, want_f32=want_f32, want_op=want_op, relu_op=relu_op, want_raw=want_raw, stats=stats)
    mask_first = scs is not None and scs[4]
    if mask_first:   # the data gradient of a D block's conv1 with the shortcut's data gradient folded in: the mask belongs to the 3x3 part only
        mask = _rt(torch.randn(ref.shape, generator=g))
        ref = ref * (mask > 0).float()
        kw["relu_mask"] = mask.to(dev, bf)
    if scs is not None:
        sxs, skpad, sbias, sup, _, salpha = scs
        xsc = _rt(torch.randn(sxs, generator=g))
        wsc = _rt(torch.randn(co, sxs[3], 1, 1, generator=g) / math.sqrt(sxs[3]))
        ref = ref + (salpha if mask_first else alpha) * _conv_ref(xsc, wsc, sup, pool2)
        bsc = torch.randn(co, generator=g) if sbias else None
        if bsc is not None:
            ref = ref + bsc
        kw["sc"] = dict(x_op=xsc.to(dev, bf), wpack=_pack(wsc, skpad).to(dev, bf), kpad=skpad, bias=None if bsc is None else bsc.to(dev),
                        up2=sup, out=torch.full(ref.shape, float("nan"), device=dev), flops=0.0)
        if mask_first:
            kw["sc"].update(mask_first=True, alpha=salpha)
    if has_bias:
        bias = torch.randn(co, generator=g)
        ref = ref + bias
        kw["bias"] = bias.to(dev)
    if has_mask and not mask_first:
        mask = _rt(torch.randn(ref.shape, generator=g))
        ref = ref * (mask > 0).float()
        kw["relu_mask"] = mask.to(dev, bf)
    if has_res:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res
        kw["res"] = res.to(dev)
    if has_nimg:
        live = max(1, int(round(live_frac * B)))
        ref[live:] = 0
        kw["nimg"] = torch.tensor([live], dtype=torch.int32, device=dev)
    out, out_op, out_raw = ops.conv_raw(x.to(dev, bf), _pack(w, kpad).to(dev, bf), kpad, co, kh, **kw)
    scale = float(ref.abs().max())
    errs = {}
    if out is not None:
        errs["out"] = float((out.cpu() - ref).abs().max()) / scale
        assert errs["out"] < 3e-5, (sig, errs)
        if out_raw is not None:
            assert float((out_raw.float().cpu() - _rt(out.cpu())).abs().max()) == 0.0, sig
        if out_op is not None:
            o = out.cpu().clamp_min(0) if relu_op else out.cpu()
            assert float((out_op.float().cpu() - _rt(o)).abs().max()) == 0.0, sig
        if stats and hasattr(out, "_l2i_stats"):
            s1, s2, _ = out._l2i_stats
            o2 = out.double().cpu().view(-1, co)
            assert float((s1.cpu().double().view(-1) - o2.sum(0)).abs().max()) < 1e-5 * float(o2.abs().sum(0).max()) + 1e-3, sig
            assert float((s2.cpu().double().view(-1) - (o2 * o2).sum(0)).abs().max()) < 1e-5 * float((o2 * o2).sum(0).max()) + 1e-3, sig
    else:   # operand-only results (relu_op_out edges, operand-dtype gradients): one bf16 rounding of the f32 accumulator
        assert out_op is not None or out_raw is not None, sig
        if out_op is not None:
            o = ref.clamp_min(0) if relu_op else ref
            errs["op"] = float((out_op.float().cpu() - o).abs().max()) / scale
            assert errs["op"] < 4e-3 + 3e-5, (sig, errs)
        if out_raw is not None:
            errs["raw"] = float((out_raw.float().cpu() - ref).abs().max()) / scale
            assert errs["raw"] < 4e-3 + 3e-5, (sig, errs)
    return errs


def _replay_wgrad(sig, live_frac, seed):
    from layout2img_amd import ops
    _, xs, dts, dys, ldw, co, kh, up2, pool2, alpha, has_nimg, has_dbias, scs, dual, overwrite = sig
    assert dts == "torch.bfloat16" and not dual
    B, Hi, Wi, Ci = xs
    g = torch.Generator().manual_seed(seed)
    x = _rt(torch.randn(xs, generator=g))
    dy = _rt(torch.randn(dys, generator=g))
    dev, bf = DEV, torch.bfloat16
    nimg = None
    if has_nimg:
        live = max(1, int(round(live_frac * B)))
        dy[live:] = 0   # (rows of dead images carry no gradient; the kernel does not read them at all)
        nimg = torch.tensor([live], dtype=torch.int32, device=dev)

    def dw_ref(xin, up, k):
        xi = xin.permute(0, 3, 1, 2)
        if up:
            xi = F.interpolate(xi, scale_factor=2, mode="nearest")
        gy = dy.permute(0, 3, 1, 2)
        if pool2:
            gy = F.interpolate(gy, scale_factor=2, mode="nearest")
        gw = torch.nn.grad.conv2d_weight(xi, (co, xi.shape[1], k, k), gy.contiguous(), padding=k // 2)
        return alpha * gw.permute(0, 2, 3, 1).reshape(co, -1)
    ref = dw_ref(x, up2, kh)
    K = kh * kh * Ci
    assert ldw >= K
    fill = float("nan") if overwrite else 0.5
    dw = torch.full((co, ldw), fill, device=dev)
    db = torch.zeros(co, device=dev) if has_dbias else None
    kw = {}
    if scs is not None:
        sxs, sldw, sdb, sup = scs
        xsc = _rt(torch.randn(sxs, generator=g))
        refs = dw_ref(xsc, sup, 1)
        dws = torch.full((co, sldw), fill, device=dev)
        dbs = torch.ones(co, device=dev) if sdb else None
        kw["sc"] = dict(x_op=xsc.to(dev, bf), dw=dws, ldw=sldw, dbias=dbs, flops=0.0, up2=sup)
    ops.wgrad_raw(x.to(dev, bf), dy.to(dev, bf), dw, ldw, co, kh, up2=up2, pool2=pool2, alpha=alpha, nimg=nimg, dbias=db,
                  overwrite=overwrite, **kw)
    base = 0.0 if overwrite else fill
    tol = lambda r: 2e-4 * float(r.abs().max()) + 1e-5
    got = dw.cpu()[:, :K]
    assert bool(torch.isfinite(got).all()), sig
    errs = dict(dw=float((got - base - ref).abs().max()) / float(ref.abs().max()))
    assert float((got - base - ref).abs().max()) < tol(ref), (sig, errs)
    # (the bias is added behind the pool: its gradient is the plain sum of dY)
    if db is not None:
        rb = dy.sum(dim=(0, 1, 2))
        assert float((db.cpu() - rb).abs().max()) < tol(rb) + 1e-4 * float(dy.abs().sum(dim=(0, 1, 2)).max()), sig
    if scs is not None:
        gs = dws.cpu()[:, :sxs[3]]
        assert float((gs - base - refs).abs().max()) < tol(refs), (sig, "sc")
        if dbs is not None:
            rb = dy.sum(dim=(0, 1, 2))
            assert float((dbs.cpu() - 1 - rb).abs().max()) < tol(rb) + 1e-4 * float(dy.abs().sum(dim=(0, 1, 2)).max()), (sig, "sc bias")
    return errs


def _run_all(kind, headline_launches, replay):
    seen, live = headline_launches
    sigs = sorted((s for s in seen if s[0] == kind), key=str)
    assert len(sigs) >= 20, len(sigs)   # (an iteration has ~100 distinct conv and ~45 distinct weight-gradient launches)
    failures, worst = [], 0.0
    for i, s in enumerate(sigs):
        try:
            e = replay(s, live / (BATCH * 8.0), 1000 + i)
            worst = max(worst, max(e.values()))
        except Exception as ex:   # (collect every failing launch, then fail once with the list)
            failures.append(f"{type(ex).__name__}: {str(ex)[:400]} @ {s}")
        torch.cuda.empty_cache()
    print(f"{kind}: {len(sigs)} distinct launches of the full-size iteration replayed, worst relative error {worst:.2e}, {len(failures)} failed")
    assert not failures, failures[:5]


def test_every_conv_launch_of_the_full_size_iteration_vs_torch_f32(headline_launches):
    """forward and data-gradient launches (l2i_conv2d_fwd / _sc / _dual): every epilogue option the trainer uses at the bench
    shapes -- bias, residual, ReLU mask, pool / upsample, operand copies, statistics, live-row count, folded shortcut."""
    _run_all("conv", headline_launches, _replay_conv)


def test_every_weight_gradient_launch_of_the_full_size_iteration_vs_torch_f32(headline_launches):
    """weight-gradient launches (l2i_conv2d_wgrad_dual): splits + reduce, stores (overwrite) into NaN-filled slices, bias
    gradients, live-row count, the shortcut's gradient as extra column tiles."""
    _run_all("wgrad", headline_launches, _replay_wgrad)


# ----------------------------------------------------------------------------- BASELINE config 5: VG layouts (31 slots, 179 classes) at b = 32
def _vg_batch():
    from layout2img_amd.synthetic import make_batch
    return make_batch(BATCH, 128, "vg", seed=SEED, device="cpu")


@pytest.fixture(scope="module")
def oracle_vg():
    real, label, bbox, z, z_im = _vg_batch()
    sd_g = fixture_state(load_fixture("g_vg_img.npz"), 53)
    sd_d = fixture_state(load_fixture("d_vg.npz"), 54)
    with torch.no_grad():
        img = O.vg_generator_forward(sd_g, z, bbox, z_im, label, training=True)
        outs = O.discriminator_forward(sd_d, real, bbox, label, training=True)
    return img, outs


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16"])
def test_vg_models_full_size_vs_oracle(mode, oracle_vg):
    """context_aware_generator (reference model/resnet_generator_vg.py:639-727: attention and ISLA over 31 object slots) and the
    discriminator on VG layouts (992 ROI slots with the `__image__` box among them) at batch 32 against the oracle."""
    import layout2img_amd as L
    from layout2img_amd import generator as G
    dt = {"f32": torch.float32, "bf16": torch.bfloat16}.get(mode, mode)
    real, label, bbox, z, z_im = (t.to(DEV) for t in _vg_batch())
    ref_img, ref_outs = oracle_vg
    torch.manual_seed(0)
    g = G.context_aware_generator(num_classes=179, output_dim=3)
    g.load_state_dict(fixture_state(load_fixture("g_vg_img.npz"), 53))
    g.finalize(DEV, dt).train()
    with torch.no_grad():
        img = g(z, bbox, z_im, label)
    e_img = maxdiff(img, ref_img)
    print(f"full-size VG G forward [{mode}]: image L_inf {e_img:.2e}")
    assert e_img < (9e-2 if mode == "bf16" else 1e-3), e_img   # (bf16: 7.17e-2 measured, identical from run to run since round 6)
    del g
    d = L.CombineDiscriminator128_app(num_classes=179)
    d.load_state_dict(fixture_state(load_fixture("d_vg.npz"), 54))
    d.finalize(DEV, dt).train()
    with torch.no_grad():
        outs = d(real, bbox, label)
    rel = 3e-2 if mode == "bf16" else 2e-4
    for t, ref, k in zip(outs, ref_outs, ("img", "obj", "app")):
        assert tuple(t.shape) == tuple(ref.shape), (k, t.shape, ref.shape)
        e, s = maxdiff(t, ref), max(1.0, float(ref.abs().max()))
        print(f"full-size VG D forward [{mode}] {k}: {e:.2e} (scale {s:.2e}, {t.shape[0]} rows)")
        assert e < rel * s, (k, e, s)
