"""GPU parity of the HIP modules against the golden vectors captured from the reference (and against
the oracle where the golden does not hold a quantity). Stated tolerances:
  f32 operands: generator image L_inf < 1e-3 (north-star bar), intermediate taps 1e-3 relative;
  bf16 operands: image L_inf < 1e-1, pre-tanh 5e-2 relative (bf16 MFMA operands, f32 accumulate / streams).
"""
import numpy as np
import pytest
import torch

from tests.golden import recipe
from tests.helpers import fixture_inputs, fixture_shapes, fixture_state, load_fixture, maxdiff

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build_g(fx, seed, dt, kind="coco"):
    import layout2img_amd as L
    from layout2img_amd import generator as G
    torch.manual_seed(0)
    if kind == "coco":
        g = L.ResnetGenerator128_context(num_classes=184, output_dim=3)
    else:
        g = G.context_aware_generator(num_classes=179, output_dim=3)
    sd = fixture_state(fx, seed)
    assert set(g.state_dict().keys()) == set(sd.keys())
    g.load_state_dict(sd)
    g.finalize(DEV, dt)
    for m in g.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
    return g


def _build_d(fx, seed, dt, num_classes=184):
    import layout2img_amd as L
    torch.manual_seed(0)
    d = L.CombineDiscriminator128_app(num_classes=num_classes)
    sd = fixture_state(fx, seed)
    assert set(d.state_dict().keys()) == set(sd.keys())
    d.load_state_dict(sd)
    return d.finalize(DEV, dt)


def _check_grad_norms(named, fx, f32):
    """Per-parameter gradient L2 norms against the reference's.

    The gradients of this un-trained, batch-of-2 network are chaotic at the 1e-3 level even between two
    fp32 runs (ReLU gates sitting at 0 flip on 1e-7 forward differences; the atomically reduced sums are not
    order-deterministic), so the test bounds (i) the MEDIAN relative error tightly -- a systematic error
    in any backward kernel moves it -- and (ii) every individual norm loosely. Parameters whose true gradient
    is 0 by cancellation (conv biases in front of a batch norm) only see the absolute floor. With bf16
    operands the degenerate batch-norm of PSP stage 0 (a 1x1 map, N = 2 samples) is skipped.
    """
    names = [str(n) for n in fx["grad_names"]]
    gn = np.array([float(named[n].grad.norm()) for n in names])
    ref = fx["grad_norms"]
    keep = np.array([f32 or "stages.0." not in n for n in names])
    rel, floor, med_tol = (2e-2, 1e-3, 1e-3) if f32 else (2.5e-1, 5e-2, 3e-2)
    med = float(np.median(np.abs(ref)))
    err = np.abs(gn - ref)
    bad = err / (rel * np.abs(ref) + floor * med)
    order = np.argsort(-(bad * keep))[:6]
    big = np.abs(ref) > 1e-2 * med
    print(f"gradient norms [{'f32' if f32 else 'bf16'}]: worst (error / bar) {float((bad * keep).max()):.3f}, median relative error {float(np.median(err[big] / np.abs(ref[big]))):.2e} (bar {med_tol})")
    assert (bad * keep).max() < 1.0, [(names[i], float(gn[i]), float(ref[i])) for i in order]
    assert float(np.median(err[big] / np.abs(ref[big]))) < med_tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_generator_coco_vs_reference(dt):
    fx = load_fixture("g_coco.npz")
    g = _build_g(fx, 11, dt)
    inp = {k: v.to(DEV) for k, v in fixture_inputs(fx).items()}
    f32 = dt == torch.float32
    g.train()
    taps = {}
    out1 = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"], taps=taps)
    rel = 1e-3 if f32 else 5e-2

    def chk(a, b, name, r=rel):
        d, s = maxdiff(a, b), float(np.abs(b).max())
        assert d < r * s, (name, d, s)
    chk(taps["w"], fx["tap_w"], "w", 1e-4 if f32 else 2e-2)
    chk(taps["bmask"], fx["tap_bmask"], "bmask")
    chk(taps["stages"][0], fx["tap_stage_in2"], "stage2")
    chk(taps["stages"][3][:, :, ::4, ::4], fx["tap_stage_in5"], "stage5")
    chk(taps["pre_tanh"].permute(0, 3, 1, 2), fx["tap_pre_tanh"], "pre_tanh")
    # bf16 operands: 6.3e-2 measured (2^-9 per operand pair, a random walk over ~25 layers)
    print(f"G coco [{'f32' if f32 else 'bf16'}]: image L_inf train1 {maxdiff(out1, fx['out_train1']):.2e}")
    assert maxdiff(out1, fx["out_train1"]) < (1e-3 if f32 else 8e-2)
    proj = torch.randn(out1.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
    g.zero_grad()
    (out1 * proj).sum().backward()
    g.arena.flush_grads()
    torch.cuda.synchronize()
    named = dict(g.named_parameters())
    _check_grad_norms(named, fx, f32)
    def l2(a, b, name, r):
        """relative L2 error of a full gradient tensor (element-wise maxima are dominated by flipped ReLU gates)"""
        b = torch.as_tensor(b)
        e = float((a.detach().cpu() - b).norm() / b.norm())
        print(f"{name} [{'f32' if f32 else 'bf16'}]: relative L2 {e:.2e} (bar {r})")
        assert e < r, (name, e)
    l2(named["fc.bias"].grad, fx["grad_fc_bias"], "grad_fc_bias", 3e-2 if f32 else 3e-1)
    l2(named["label_embedding.weight"].grad, fx["grad_emb"], "grad_emb", 3e-2 if f32 else 3e-1)
    l2(named["alpha1"].grad, fx["grad_alpha1"], "grad_alpha1", 3e-2 if f32 else 3e-1)
    with torch.no_grad():
        out2 = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
        assert maxdiff(out2[:, :, ::2, ::2], fx["out_train2_sub"]) < (1e-3 if f32 else 1e-1)
        g.eval()
        oe = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
        assert maxdiff(oe, fx["out_eval"]) < (1e-3 if f32 else 1e-1)


@pytest.mark.parametrize("kind", ["coco", "vg"])
def test_generator_forward_bf16x3_meets_the_image_bar(kind):
    """The forward-only split-operand mode (finalize(dev, "bf16x3"): bf16 operands carried as hi + lo, three MFMA products per
    pair, f32 accumulation) against the reference's outputs: the north star's image bar L_inf < 1e-3 -- which plain bf16
    operands miss by 60x (6.3e-2) -- at MFMA speed, train-mode forward twice (batch statistics, running statistics) and eval."""
    coco = kind == "coco"
    fx = load_fixture("g_coco.npz" if coco else "g_vg.npz")
    g = _build_g(fx, 11 if coco else 12, "bf16x3", kind=kind)
    assert g.arena.split and g.arena.op_dtype == torch.bfloat16
    inp = {k: v.to(DEV) for k, v in fixture_inputs(fx).items()}
    g.train()
    with pytest.raises(RuntimeError, match="forward-only"):
        g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
    g.load_state_dict(fixture_state(fx, 11 if coco else 12))   # (whatever the refused call touched: start from the recipe state again)
    with torch.no_grad():
        taps = {}
        out1 = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"], **({"taps": taps} if coco else {}))
        e1 = maxdiff(out1, fx["out_train1"])
        if coco:
            for name, a, b in (("w", taps["w"], fx["tap_w"]), ("bmask", taps["bmask"], fx["tap_bmask"]),
                               ("pre_tanh", taps["pre_tanh"].permute(0, 3, 1, 2), fx["tap_pre_tanh"])):
                d, sc = maxdiff(a, b), float(np.abs(b).max())
                assert d < 2e-4 * sc, (name, d, sc)
        out2 = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
        e2 = maxdiff(out2[:, :, ::2, ::2], fx["out_train2_sub"])
        g.eval()
        e3 = maxdiff(g(inp["z"], inp["bbox"], inp["z_im"], inp["y"]), fx["out_eval"])
    print(f"bf16x3 {kind}: image L_inf train1 {e1:.2e} train2 {e2:.2e} eval {e3:.2e}")
    assert max(e1, e2, e3) < 1e-3, (e1, e2, e3)   # measured: see DESIGN.md section 2 (emulation on the CPU: 8e-5)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_generator_vg_vs_reference(dt):
    fx = load_fixture("g_vg.npz")
    g = _build_g(fx, 12, dt, kind="vg")
    inp = {k: v.to(DEV) for k, v in fixture_inputs(fx).items()}
    f32 = dt == torch.float32
    g.train()
    out1 = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
    assert maxdiff(out1, fx["out_train1"]) < (1e-3 if f32 else 1e-1)
    proj = torch.randn(out1.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
    g.zero_grad()
    (out1 * proj).sum().backward()
    g.arena.flush_grads()
    named = dict(g.named_parameters())
    _check_grad_norms(named, fx, f32)
    with torch.no_grad():
        out2 = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
        assert maxdiff(out2[:, :, ::2, ::2], fx["out_train2_sub"]) < (1e-3 if f32 else 1e-1)
        g.eval()
        oe = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
        assert maxdiff(oe, fx["out_eval"]) < (1e-3 if f32 else 1e-1)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_discriminator_vs_reference(dt):
    fx = load_fixture("d_coco.npz")
    d = _build_d(fx, 21, dt)
    inp = {k: v.to(DEV) for k, v in fixture_inputs(fx).items()}
    f32 = dt == torch.float32
    rel = 2e-4 if f32 else 3e-2
    d.train()
    real = inp["real"].clone().requires_grad_(True)
    bbox0 = inp["bbox"].clone()
    o1 = d(real, inp["bbox"], inp["y"].unsqueeze(-1))
    assert torch.equal(inp["bbox"], bbox0)
    for t, k in zip(o1, ("img", "obj", "app")):
        ref = fx[f"train1_{k}"]
        assert tuple(t.shape) == ref.shape, (k, t.shape, ref.shape)
        assert maxdiff(t, ref) < rel * max(1.0, float(np.abs(ref).max())), (k, maxdiff(t, ref), float(np.abs(ref).max()))
    gen = torch.Generator().manual_seed(6)
    d.zero_grad()
    sum((t * torch.randn(t.shape, generator=gen).to(DEV)).sum() for t in o1).backward()
    d.arena.flush_grads()
    named = dict(d.named_parameters())
    _check_grad_norms(named, fx, f32)
    gi = torch.from_numpy(fx["grad_input_sub"])
    assert float((real.grad[:, :, ::4, ::4].cpu() - gi).norm() / gi.norm()) < (1e-2 if f32 else 2e-1)
    gl = torch.from_numpy(fx["grad_l7_w"])
    assert float((named["obD.l7.weight_orig"].grad.cpu() - gl).norm() / gl.norm()) < (1e-2 if f32 else 2e-1)
    with torch.no_grad():
        o2 = d(inp["real"], inp["bbox"], inp["y"].unsqueeze(-1))
        d.eval()
        oe = d(inp["real"], inp["bbox"], inp["y"].unsqueeze(-1))
    for t, e, k in zip(o2, oe, ("img", "obj", "app")):
        assert maxdiff(t, fx[f"train2_{k}"]) < rel * max(1.0, float(np.abs(fx[f"train2_{k}"]).max())), k
        assert maxdiff(e, fx[f"eval_{k}"]) < rel * max(1.0, float(np.abs(fx[f"eval_{k}"]).max())), k


def test_discriminator_forward_bf16x3_at_the_f32_bar():
    """The forward-only split-operand mode on the discriminator (finalize(dev, "bf16x3")): train-mode forward twice and eval
    against the reference's outputs at the exact-f32 mode's bar (2e-4 relative; plain bf16 operands need 3e-2)."""
    fx = load_fixture("d_coco.npz")
    d = _build_d(fx, 21, "bf16x3")
    inp = {k: v.to(DEV) for k, v in fixture_inputs(fx).items()}
    d.train()
    with pytest.raises(RuntimeError, match="forward-only"):
        d(inp["real"], inp["bbox"], inp["y"].unsqueeze(-1))
    d.load_state_dict(fixture_state(fx, 21))
    with torch.no_grad():
        o1 = d(inp["real"], inp["bbox"], inp["y"].unsqueeze(-1))
        o2 = d(inp["real"], inp["bbox"], inp["y"].unsqueeze(-1))
        d.eval()
        oe = d(inp["real"], inp["bbox"], inp["y"].unsqueeze(-1))
    for outs, tag in ((o1, "train1"), (o2, "train2"), (oe, "eval")):
        for t, k in zip(outs, ("img", "obj", "app")):
            ref = fx[f"{tag}_{k}"]
            assert tuple(t.shape) == ref.shape
            assert maxdiff(t, ref) < 2e-4 * max(1.0, float(np.abs(ref).max())), (tag, k, maxdiff(t, ref), float(np.abs(ref).max()))


# Bars of the loop tests: (loss relative error, image L_inf) per iteration, and the parameter-sum slack in units of "every
# element of the tensor moved by 2 lr". f32: as in rounds 1-2. bf16: 1.5 x the LARGEST value of four runs of
# tools/parity/measure_bars.py on MI355X boxes -- the run-to-run spread is wide because already iteration 0's g_loss is
# taken after D's first Adam(beta1 = 0) step, i.e. +-lr per element by the SIGN of a bf16-noisy gradient:
#   COCO  d/g loss it 0: 4.0-4.4e-4 / 2.2e-4-1.07e-3, it 1: 0.5-5.0e-4 / 2.0e-4-1.5e-3; image 3.7-4.3e-2, 8.2-9.6e-2; sums 0.087-0.185
#   VG    d/g loss it 0: 7.5-8.0e-4 / 8.1e-4-9.9e-4,  it 1: 6.7-8.0e-4 / 5.1e-4-3.2e-3; image 3.7-4.9e-2, 1.15-1.57e-1; sums 0.069-0.165
_LOOP_BARS = {
    ("coco", True): ((5e-4, 1e-3), (3e-2, 2e-2), 0.05),
    ("coco", False): ((1.6e-3, 6.5e-2), (2.3e-3, 1.45e-1), 0.28),
    ("vg", True): ((5e-4, 1e-3), (3e-2, 2e-2), 0.05),
    ("vg", False): ((1.5e-3, 7.5e-2), (4.9e-3, 2.4e-1), 0.25),
}


def _loop_vs_reference(kind, dt, dual=False, early=False):
    import layout2img_amd as L
    vg = kind == "vg"
    f32 = dt == torch.float32
    fx = load_fixture("train_loop_vg.npz" if vg else "train_loop.npz")
    g = _build_g(load_fixture("g_vg_img.npz" if vg else "g_coco.npz"), 53 if vg else 31, dt, kind="vg" if vg else "coco")
    d = _build_d(load_fixture("d_vg.npz" if vg else "d_coco.npz"), 54 if vg else 32, dt, num_classes=179 if vg else 184)
    g.train(), d.train()
    tr = L.GanTrainer(g, d)
    tr.dual_d = dual   # D(real) + D(fake) of the D step as one batch (GanTrainer.dual_d, CombineDiscriminator.forward_dual)
    tr.real_bwd_early = early   # D(real)'s backward on the side stream right behind its forward (GanTrainer.real_bwd_early)
    bars = _LOOP_BARS[(kind, f32)]
    for it in range(2):
        mk = recipe.make_inputs_vg(2, 31, 179, 300 + it) if vg else recipe.make_inputs(2, 8, 184, 200 + it)
        inp = {k: v.to(DEV) for k, v in mk.items()}
        r = tr.step(inp["real"], inp["y"], inp["bbox"], inp["z"], inp["z_im"])
        rel, tol_img = bars[it]
        errs = {k: abs(float(r[k]) - float(fx[f"{k}{it}"])) / max(1.0, abs(float(fx[f"{k}{it}"]))) for k in ("d_loss", "g_loss")}
        e_img = maxdiff(r["fake"][:, :, ::4, ::4], fx[f"fake_sub{it}"])
        print(f"loop {kind} {'f32' if f32 else 'bf16'} dual={dual} early={early} it {it}: d_loss {errs['d_loss']:.2e} g_loss {errs['g_loss']:.2e} image {e_img:.2e}")
        for k in ("d_loss", "g_loss"):
            assert errs[k] < rel, (k, it, float(r[k]), float(fx[f"{k}{it}"]))
        assert e_img < tol_img
    for net, pre in ((g, "g"), (d, "d")):
        named = dict(net.named_parameters())
        names = [str(n) for n in fx[f"{pre}_param_names"]]
        sums = np.array([float(named[n].detach().double().sum()) for n in names])
        numel = np.array([named[n].numel() for n in names])
        tol = 1e-3 * (np.abs(fx[f"{pre}_param_sums"]) + 1.0) + 2 * 2e-4 * numel * bars[2]
        bad = np.abs(sums - fx[f"{pre}_param_sums"]) - tol
        slack = float(np.max((np.abs(sums - fx[f"{pre}_param_sums"]) - 1e-3 * (np.abs(fx[f"{pre}_param_sums"]) + 1.0)) / (2 * 2e-4 * numel)))
        print(f"loop {kind} {'f32' if f32 else 'bf16'} {pre}: largest parameter-sum slack used {slack:.3f} of the bar's {bars[2]}")
        assert np.all(bad < 0), (pre, names[int(bad.argmax())])


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_train_loop_vs_reference(dt):
    """Two iterations of the training loop (reference train_context_app_v2.py:148-189, VGG term omitted) against the losses,
    images and parameter sums captured from the reference loop -- f32 operands and the bf16 operands of the headline."""
    _loop_vs_reference("coco", dt)


@pytest.mark.parametrize("kind", ["coco", "vg"])
def test_train_loop_vs_reference_with_the_dual_discriminator_step(kind):
    """The same goldens with D(real) and D(fake) of the discriminator step run as ONE batch (two weight packs and two gradient
    accumulators per launch, l2i_conv2d_fwd_dual / l2i_conv2d_wgrad_dual): same losses, images and parameter sums, same bars."""
    _loop_vs_reference(kind, torch.float32, dual=True)


def test_train_loop_vs_reference_with_the_early_real_backward():
    """`GanTrainer.real_bwd_early` (L2I_REAL_BWD_EARLY=1: D(real)'s backward as its own backward call on the side stream, D(fake)'s
    afterwards on the main stream; measured slower, off by default) against the same goldens: the two separate backward passes
    accumulate into the same gradient buffers what one backward over the summed loss does."""
    _loop_vs_reference("coco", torch.float32, early=True)


def test_full_size_step_properties():
    """BASELINE config 3 (128x128, b=32, bf16): size-independent properties of one full training iteration:
    finite losses, image range, D outputs of the padded rows unaffected by boxes of padding slots,
    parameters actually move, spectral-norm sigma > 0."""
    import layout2img_amd as L
    from layout2img_amd.synthetic import make_batch
    torch.manual_seed(0)
    g = L.ResnetGenerator128_context(num_classes=184).finalize(DEV, torch.bfloat16)
    d = L.CombineDiscriminator128_app(num_classes=184).finalize(DEV, torch.bfloat16)
    tr = L.GanTrainer(g, d)
    real, label, bbox, z, z_im = make_batch(32, 128, "coco", seed=3, device=DEV)
    p0 = g.flat.data.clone()
    r = tr.step(real, label, bbox, z, z_im)
    torch.cuda.synchronize()
    assert torch.isfinite(r["d_loss"]) and torch.isfinite(r["g_loss"])
    assert r["fake"].shape == (32, 3, 128, 128) and float(r["fake"].abs().max()) <= 1.0
    assert float((g.flat.data - p0).abs().max()) > 0
    with torch.no_grad():
        a = d.forward_padded(real, bbox, label)
        bbox2 = bbox.clone()
        bbox2[label == 0] = torch.tensor([0.1, 0.1, 0.3, 0.3], device=DEV)
        b = d.forward_padded(real, bbox2, label)
    v = a[3].bool()
    assert torch.equal(a[3], b[3])
    assert maxdiff(a[0], b[0]) < 0.2 * float(a[0].abs().max()) + 1.0  # different SN iteration, same images
    assert torch.isfinite(a[1][v]).all() and torch.isfinite(a[2][v]).all()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_vg_models_with_image_slot_vs_reference(dt):
    """BASELINE config 5 models on layouts with the `__image__` slot (reference data/vg.py:120,135: label 0 and box
    [0,0,1,1] -- not an attention key, not an ROI, but its mask feeds ISLA): generator and discriminator (o = 31,
    179 classes, 45 valid of 62 ROI rows, one ROI exactly 64 px wide) against reference outputs."""
    f32 = dt == torch.float32
    fx = load_fixture("g_vg_img.npz")
    g = _build_g(fx, 51, dt, kind="vg")
    inp = {k: v.to(DEV) for k, v in fixture_inputs(fx).items()}
    g.train()
    out1 = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
    assert maxdiff(out1[:, :, ::2, ::2], fx["out_train1_sub"]) < (1e-3 if f32 else 1e-1)
    proj = torch.randn(out1.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
    g.zero_grad()
    (out1 * proj).sum().backward()
    g.arena.flush_grads()
    _check_grad_norms(dict(g.named_parameters()), fx, f32)
    with torch.no_grad():
        g.eval()
        oe = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
    assert maxdiff(oe[:, :, ::2, ::2], fx["out_eval_sub"]) < (1e-3 if f32 else 1e-1)

    fx = load_fixture("d_vg.npz")
    d = _build_d(fx, 52, dt, num_classes=179)
    rel = 2e-4 if f32 else 3e-2
    d.train()
    real = inp["real"].clone().requires_grad_(True)
    o1 = d(real, inp["bbox"], inp["y"].unsqueeze(-1))
    for t, k in zip(o1, ("img", "obj", "app")):
        ref = fx[f"train1_{k}"]
        assert tuple(t.shape) == ref.shape, (k, t.shape, ref.shape)
        assert maxdiff(t, ref) < rel * max(1.0, float(np.abs(ref).max())), (k, maxdiff(t, ref))
    gen = torch.Generator().manual_seed(6)
    d.zero_grad()
    sum((t * torch.randn(t.shape, generator=gen).to(DEV)).sum() for t in o1).backward()
    d.arena.flush_grads()
    _check_grad_norms(dict(d.named_parameters()), fx, f32)
    gi = torch.from_numpy(fx["grad_input_sub"])
    assert float((real.grad[:, :, ::4, ::4].cpu() - gi).norm() / gi.norm()) < (1e-2 if f32 else 2e-1)
    with torch.no_grad():
        d.eval()
        oe = d(inp["real"], inp["bbox"], inp["y"].unsqueeze(-1))
    for e, k in zip(oe, ("img", "obj", "app")):
        assert maxdiff(e, fx[f"eval_{k}"]) < rel * max(1.0, float(np.abs(fx[f"eval_{k}"]).max())), k


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_vg_train_loop_vs_reference(dt):
    """Two iterations of the loop with the VG models on `__image__`-slot layouts."""
    _loop_vs_reference("vg", dt)


def test_full_size_vg_step_properties():
    """BASELINE config 5 at full size (128x128, b = 32, o = 31, 179 classes, bf16): one training iteration -- finite
    losses, image range, parameters move, and the `__image__` slot is never an ROI while it does change the image."""
    import layout2img_amd as L
    from layout2img_amd import generator as G
    from layout2img_amd.synthetic import make_batch
    torch.manual_seed(0)
    g = G.context_aware_generator(num_classes=179).finalize(DEV, torch.bfloat16)
    d = L.CombineDiscriminator128_app(num_classes=179).finalize(DEV, torch.bfloat16)
    tr = L.GanTrainer(g, d)
    real, label, bbox, z, z_im = make_batch(32, 128, "vg", seed=3, device=DEV)
    n_real = (label != 0).sum(1)
    assert all(bbox[i, int(n_real[i])].tolist() == [0.0, 0.0, 1.0, 1.0] for i in range(32))
    p0g, p0d = g.flat.data.clone(), d.flat.data.clone()
    r = tr.step(real, label, bbox, z, z_im)
    torch.cuda.synchronize()
    assert torch.isfinite(r["d_loss"]) and torch.isfinite(r["g_loss"])
    assert r["fake"].shape == (32, 3, 128, 128) and float(r["fake"].abs().max()) <= 1.0
    assert float((g.flat.data - p0g).abs().max()) > 0 and float((d.flat.data - p0d).abs().max()) > 0
    with torch.no_grad():
        out = d.forward_padded(real, bbox, label)
        assert int(out[3].sum()) == int(n_real.sum())           # valid ROI rows == real objects: the image slot is none
        g.eval()
        a = g(z, bbox, z_im, label)
        bbox2 = bbox.clone()
        for i in range(32):
            bbox2[i, int(n_real[i])] = torch.tensor([-0.6, -0.6, 0.5, 0.5], device=DEV)   # image slot -> plain padding
        b = g(z, bbox2, z_im, label)
    assert maxdiff(a, b) > 1e-3   # its full-canvas mask contributes to ISLA


def _vgg(dt):
    import layout2img_amd as L
    from tests.helpers import vgg_state
    fx = load_fixture("vgg.npz")
    m = L.VGGLoss()
    sd = vgg_state(fx)
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd)
    return m.finalize(DEV, dt), fx


def test_vgg_loss_vs_reference_f32():
    """VGG19 perceptual loss (reference utils/util.py:49-94, train_context_app_v2.py:141,185) on the MFMA conv path with
    exact-f32 operands against the reference's own VGGLoss run on recipe weights (tests/golden/vgg.npz): loss value, the
    gradient with respect to the fake image, state_dict keys = the reference's.
    The launches are made DETERMINISTIC for this test (split-K off: l2i_set_conv_config(1001)): with f32 atomics combining
    split-K partials in a varying order, one run in twelve put a ~0 pre-activation on the other side of its ReLU gate and
    moved the gradient error from 9.3e-4 to 6.5e-3 (tools/perf/vgg_repeat.py); round 2 had widened the bar for that."""
    from layout2img_amd import _lib
    from tests.helpers import vgg_inputs
    m, fx = _vgg(torch.float32)
    x, y = vgg_inputs()
    x = x.to(DEV).requires_grad_(True)
    _lib.call("l2i_set_conv_config", 1001)
    try:
        loss = m(x, y.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
    finally:
        _lib.call("l2i_set_conv_config", 1512)
    ref = float(fx["loss"])
    assert abs(float(loss) - ref) < 1e-4 * abs(ref), (float(loss), ref)
    g = torch.from_numpy(fx["grad_x_sub"])
    err = float((x.grad[:, :, ::2, ::2].cpu() - g).norm() / g.norm())
    assert err < 1.5e-3, err
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in m.parameters())   # frozen


def test_vgg_loss_bf16_operands_tap_by_tap():
    """bf16 operands: the loss against the reference's (2 %), and -- instead of one loose bound on the image gradient -- every
    feature tap (relu1_1 ... relu5_1, the five terms of the loss) and the image gradient against the SAME kernels run with
    exact-f32 operands (which the test above pins to the reference). Bars = 1.5 x the values measured on an MI355X:
    per-tap relative L2 error of relu(tap) 2.2e-3, 3.7e-3, 4.2e-3, 5.9e-3, 7.2e-3 (13 layers of 2^-9 operand rounding,
    growing with depth); image gradient: relative L2 error 0.35, cosine 0.940 -- the gradient passes 13 ReLU gates of a
    RANDOM-weight network, where a rounding-sized change of a pre-activation near 0 switches its whole receptive field."""
    from tests.helpers import vgg_inputs
    x0, y0 = vgg_inputs()
    taps, grads, losses = {}, {}, {}
    for dt in (torch.float32, torch.bfloat16):
        m, fx = _vgg(dt)
        x = x0.to(DEV).requires_grad_(True)
        losses[dt] = m(x, y0.to(DEV))
        losses[dt].backward()
        with torch.no_grad():
            taps[dt] = [torch.relu(t).float().cpu() for t in m.vgg(m._nhwc8(x0.to(DEV)))]
        grads[dt] = x.grad.detach().float().cpu()
    ref = float(fx["loss"])
    assert abs(float(losses[torch.bfloat16]) - ref) < 2e-2 * abs(ref)
    errs = [float((a - b).norm() / b.norm()) for a, b in zip(taps[torch.bfloat16], taps[torch.float32])]
    for e, bar in zip(errs, (3.4e-3, 5.6e-3, 6.4e-3, 8.8e-3, 1.1e-2)):
        assert e < bar, errs
    ga, gb = grads[torch.bfloat16], grads[torch.float32]
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    rel = float((ga - gb).norm() / gb.norm())
    print(f"VGG bf16 vs f32 operands: taps {[round(e, 5) for e in errs]}, image gradient cosine {cos:.4f}, relative L2 {rel:.3f}")
    assert cos > 0.91 and rel < 0.52, (cos, rel, errs)


_EAGER_GRADS = {}


def _grads_after_one_iteration(dt, mode, b=32, seed=5):
    """Flat gradients of both networks after ONE training iteration from a fixed state, run eagerly or as the replayed HIP
    graph (GanTrainer.capture: its warm-up iterations are undone by restoring the state before the replay)."""
    if mode == "eager" and (dt, b, seed) in _EAGER_GRADS:   # (the eager iteration is the reference of two tests: run once)
        return _EAGER_GRADS[(dt, b, seed)]
    out = _grads_after_one_iteration_uncached(dt, mode, b, seed)
    if mode == "eager":
        _EAGER_GRADS[(dt, b, seed)] = out
    return out


def _grads_after_one_iteration_uncached(dt, mode, b, seed):
    import layout2img_amd as L
    from layout2img_amd.synthetic import make_batch
    from layout2img_amd.trainer import restore_state, snapshot_state
    torch.manual_seed(seed)
    g = L.ResnetGenerator128_context(num_classes=184).finalize(DEV, dt)
    d = L.CombineDiscriminator128_app(num_classes=184).finalize(DEV, dt)
    for m in g.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0   # (the graph replays its own Philox offsets: draws differ from an eager run's)
    tr = L.GanTrainer(g, d)
    tr.dual_d = mode == "dual"
    real, label, bbox, z, z_im = make_batch(b, 128, "coco", seed=3, device=DEV)
    if mode == "graph":
        st = snapshot_state(tr)
        assert tr.capture(real, label, bbox, z, z_im)
        restore_state(tr, st)
        tr.step_graphed(real, label, bbox, z, z_im)
    else:
        tr.step(real, label, bbox, z, z_im)
    torch.cuda.synchronize()
    out = {"g." + n: p.grad.detach().float().cpu().clone() for n, p in g.named_parameters()}
    out.update({"d." + n: p.grad.detach().float().cpu().clone() for n, p in d.named_parameters()})
    return out


def _grad_errors(a, b):
    """(whole-gradient relative L2, median per-parameter relative L2) of gradient dict a against b."""
    cat = lambda t: torch.cat([t[k].reshape(-1) for k in sorted(t)])
    whole = float((cat(a) - cat(b)).norm() / cat(b).norm())
    per = [float((a[k] - b[k]).norm() / b[k].norm()) for k in b if float(b[k].norm()) > 0]
    return whole, float(np.median(per))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_graph_replay_gradients_match_eager_at_full_size(dt):
    """What BENCH times -- the graph-replayed 128x128, b = 32 iteration -- computes the eager iteration's GRADIENTS: both
    networks' flat gradients after one iteration from the same state (before Adam's sign amplification).
    Exactly: see the comment at the assertion."""
    eager = _grads_after_one_iteration(dt, "eager")
    graph = _grads_after_one_iteration(dt, "graph")
    whole, med = _grad_errors(graph, eager)
    print(f"graph vs eager at b = 32 [{dt}]: whole-gradient relative L2 {whole:.2e}, median per parameter {med:.2e}")
    # Round 6: EVERY gradient tensor of both networks has the same bits -- no float atomics are left on the path, the replay issues the eager
    # iteration's launches in the eager iteration's order. (Rounds 2-5, when batch statistics, power-iteration sums, split-K results and bias
    # gradients were summed by float atomics: bars 2.5e-4 / 9e-4 (f32) and 8.7e-3 / 1.7e-2 (bf16), 1.5 x the eager-vs-eager floor.)
    bad = [k for k in eager if not torch.equal(graph[k], eager[k])]
    assert not bad and whole == 0.0, (len(bad), bad[:5], whole)


@pytest.mark.parametrize("dt", [torch.bfloat16])   # (the off-by-default one-batch form at the headline dtype; its f32 arithmetic is test_gpu_09_dual.py's)
def test_dual_discriminator_step_gradients_match_two_passes_at_full_size(dt):
    """128x128, b = 32: the discriminator step as ONE batch of 64 images (every D-step conv / data-gradient / weight-gradient
    launch dual: two packs, two accumulators, the ROI heads' live-row count per half) gives the two-pass iteration's gradients
    of both networks -- same floor and bars as graph vs eager above."""
    eager = _grads_after_one_iteration(dt, "eager")
    dual = _grads_after_one_iteration(dt, "dual")
    whole, med = _grad_errors(dual, eager)
    print(f"dual vs two-pass at b = 32 [{dt}]: whole-gradient relative L2 {whole:.2e}, median per parameter {med:.2e}")
    # (another launch structure -- one batch of 64 images, other tiles and splits -- so other summation orders: not bit-identical, and in bf16 a
    #  value on a rounding boundary goes the other way. Round 6, both sides reproducible: 6.6e-4 / 2.1e-3 measured; bars 3 x that. Rounds 4-5:
    #  8.7e-3 / 1.7e-2, 1.5 x the then run-to-run floor.)
    assert whole < 2e-3, whole
    assert med < 6.5e-3, med
