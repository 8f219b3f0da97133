"""CPU: the oracle restatement against golden vectors captured from the imported reference
(tools/capture_goldens.py). Tolerances are fp32 round-off of differently-ordered but equivalent sums."""
import numpy as np
import pytest
import torch

from oracle import model as O
from tests.helpers import fixture_inputs, fixture_state, load_fixture, maxdiff, vgg_inputs, vgg_state

TOL = 2e-5


def _grad_norms(sd, names):
    return np.array([0.0 if sd[n].grad is None else float(sd[n].grad.norm()) for n in names])


def _norms_close(gn, ref, rel=2e-3):
    """Per-parameter gradient norms; entries that are pure round-off (e.g. conv biases in front of a
    batch norm, whose true gradient is 0) are judged against the median scale instead of themselves."""
    floor = 1e-5 * float(np.median(np.abs(ref)) + 1e-12)
    return np.max(np.abs(gn - ref) / (np.abs(ref) + floor / rel)) < rel


def test_small_ops_known_answers():
    fx = load_fixture("small_ops.npz")
    boxes = torch.from_numpy(fx["boxes"])
    ones = O.masks_to_layout(boxes, torch.ones(1, 3, 16, 16), 64)
    assert maxdiff(ones, fx["ones_layout"]) < 1e-6
    # box [.25,.25,.5,.5] -> 1156 non-zero pixels summing to 991.75, pad box -> 0
    assert int((ones[0, 1] != 0).sum()) == 1156 and abs(float(ones[0, 1].sum()) - 991.75) < 1e-2
    assert float(ones[0, 2].abs().sum()) == 0.0
    assert maxdiff(O.masks_to_layout(boxes, torch.from_numpy(fx["rand_masks"]), 64), fx["rand_layout"]) < 1e-6
    bm = O.bbox_mask(boxes, 64, 64)
    assert maxdiff(bm, fx["bbox_mask"]) == 0.0
    assert bm.sum(dim=(2, 3)).tolist() == [[4096.0, 1024.0, 0.0]]


@pytest.mark.parametrize("kind,seed", [("coco", 11), ("vg", 12)])
def test_generator_matches_reference(kind, seed):
    fx = load_fixture(f"g_{kind}.npz")
    sd = O.make_trainable(fixture_state(fx, seed))
    inp = fixture_inputs(fx)
    fwd = O.generator_forward if kind == "coco" else O.vg_generator_forward
    kw = dict(dropout_p=0.0) if kind == "coco" else {}
    taps = {}
    if kind == "coco":
        kw["taps"] = taps
    out1 = fwd(sd, inp["z"], inp["bbox"], inp["z_im"], inp["y"], training=True, **kw)
    assert maxdiff(out1, fx["out_train1"]) < TOL
    if kind == "coco":
        assert maxdiff(taps["w"], fx["tap_w"]) < TOL
        assert maxdiff(taps["bmask"], fx["tap_bmask"]) < TOL
        assert maxdiff(taps["stages"][0], fx["tap_stage_in2"]) < TOL
        assert maxdiff(taps["stages"][3][:, :, ::4, ::4], fx["tap_stage_in5"]) < TOL
        assert maxdiff(taps["pre_tanh"], fx["tap_pre_tanh"]) < 2e-4
    proj = torch.randn(out1.shape, generator=torch.Generator().manual_seed(5))
    (out1 * proj).sum().backward()
    names = [str(n) for n in fx["grad_names"]]
    gn = _grad_norms(sd, names)
    ref = fx["grad_norms"]
    assert _norms_close(gn, ref)
    assert maxdiff(sd["fc.bias"].grad, fx["grad_fc_bias"]) < 1e-3 * max(1.0, float(np.abs(fx["grad_fc_bias"]).max()))
    assert maxdiff(sd["label_embedding.weight"].grad, fx["grad_emb"]) < 1e-3 * max(1.0, float(np.abs(fx["grad_emb"]).max()))
    with torch.no_grad():
        kw.pop("taps", None)
        out2 = fwd(sd, inp["z"], inp["bbox"], inp["z_im"], inp["y"], training=True, **kw)
        assert maxdiff(out2[:, :, ::2, ::2], fx["out_train2_sub"]) < TOL
        oe = fwd(sd, inp["z"], inp["bbox"], inp["z_im"], inp["y"], training=False, **kw)
        assert maxdiff(oe, fx["out_eval"]) < TOL


def test_discriminator_matches_reference():
    fx = load_fixture("d_coco.npz")
    sd = O.make_trainable(fixture_state(fx, 21))
    inp = fixture_inputs(fx)
    real = inp["real"].clone().requires_grad_(True)
    bbox0 = inp["bbox"].clone()
    o1 = O.discriminator_forward(sd, real, inp["bbox"], inp["y"], training=True)
    assert torch.equal(inp["bbox"], bbox0)  # never mutates the caller's boxes
    for t, k in zip(o1, ("img", "obj", "app")):
        ref = fx[f"train1_{k}"]
        assert t.shape == ref.shape
        assert maxdiff(t, ref) < 1e-4 * max(1.0, float(np.abs(ref).max())), k
    g = torch.Generator().manual_seed(6)
    sum((t * torch.randn(t.shape, generator=g)).sum() for t in o1).backward()
    names = [str(n) for n in fx["grad_names"]]
    gn, ref = _grad_norms(sd, names), fx["grad_norms"]
    assert _norms_close(gn, ref)
    assert maxdiff(real.grad[:, :, ::4, ::4], fx["grad_input_sub"]) < 1e-3 * max(1.0, float(np.abs(fx["grad_input_sub"]).max()))
    with torch.no_grad():
        o2 = O.discriminator_forward(sd, inp["real"], inp["bbox"], inp["y"], training=True)
        oe = O.discriminator_forward(sd, inp["real"], inp["bbox"], inp["y"], training=False)
    for t, e, k in zip(o2, oe, ("img", "obj", "app")):
        assert maxdiff(t, fx[f"train2_{k}"]) < 1e-4 * max(1.0, float(np.abs(fx[f"train2_{k}"]).max()))
        assert maxdiff(e, fx[f"eval_{k}"]) < 1e-4 * max(1.0, float(np.abs(fx[f"eval_{k}"]).max()))


def test_train_loop_matches_reference():
    from tests.golden import recipe
    fx = load_fixture("train_loop.npz")
    g_fx, d_fx = load_fixture("g_coco.npz"), load_fixture("d_coco.npz")
    sd_g = O.make_trainable(fixture_state(g_fx, 31))
    sd_d = O.make_trainable(fixture_state(d_fx, 32))
    tr = O.OracleTrainer(sd_g, sd_d, dropout_p=0.0)
    # Adam with beta1 = 0 moves every element by ~lr * sign(g): elements whose true gradient is 0 (round-off
    # sign) legitimately differ between implementations, so everything after the first update is compared
    # with tolerances that allow a few percent of sign flips.
    for it in range(2):
        inp = recipe.make_inputs(2, 8, 184, 200 + it)
        r = tr.step(inp["real"], inp["y"], inp["bbox"], inp["z"], inp["z_im"])
        rel, tol_img = (1e-4, 5e-5) if it == 0 else (2e-2, 1e-2)
        assert abs(float(r["d_loss"]) - float(fx[f"d_loss{it}"])) < rel * max(1.0, abs(float(fx[f"d_loss{it}"])))
        assert abs(float(r["g_loss"]) - float(fx[f"g_loss{it}"])) < rel * max(1.0, abs(float(fx[f"g_loss{it}"])))
        assert maxdiff(r["fake"][:, :, ::4, ::4], fx[f"fake_sub{it}"]) < tol_img
    for sd, pre in ((sd_g, "g"), (sd_d, "d")):
        names = [str(n) for n in fx[f"{pre}_param_names"]]
        sums = np.array([float(sd[n].detach().double().sum()) for n in names])
        numel = np.array([sd[n].numel() for n in names])
        tol = 1e-4 * (np.abs(fx[f"{pre}_param_sums"]) + 1.0) + 2 * 2e-4 * numel * 0.05
        assert np.all(np.abs(sums - fx[f"{pre}_param_sums"]) < tol)


def test_vg_models_with_image_slot_match_reference():
    """BASELINE config 5 (o = 31, 179 classes) on layouts with the `__image__` slot of data/vg.py:120,135 (label 0,
    box [0,0,1,1]): generator, discriminator (45 ROIs incl. one exactly 64 px wide) and two loop iterations."""
    from tests.golden import recipe
    fx = load_fixture("g_vg_img.npz")
    inp = fixture_inputs(fx)
    n_real = (inp["y"] != 0).sum(1)
    for i in range(inp["y"].shape[0]):   # the fixture really holds the slot
        assert inp["bbox"][i, int(n_real[i])].tolist() == [0.0, 0.0, 1.0, 1.0] and int(inp["y"][i, int(n_real[i])]) == 0
    sd = O.make_trainable(fixture_state(fx, 51))
    out1 = O.vg_generator_forward(sd, inp["z"], inp["bbox"], inp["z_im"], inp["y"], training=True)
    assert maxdiff(out1[:, :, ::2, ::2], fx["out_train1_sub"]) < TOL
    proj = torch.randn(out1.shape, generator=torch.Generator().manual_seed(5))
    (out1 * proj).sum().backward()
    assert _norms_close(_grad_norms(sd, [str(n) for n in fx["grad_names"]]), fx["grad_norms"])
    with torch.no_grad():
        oe = O.vg_generator_forward(sd, inp["z"], inp["bbox"], inp["z_im"], inp["y"], training=False)
    assert maxdiff(oe[:, :, ::2, ::2], fx["out_eval_sub"]) < TOL

    fx = load_fixture("d_vg.npz")
    sd = O.make_trainable(fixture_state(fx, 52))
    real = inp["real"].clone().requires_grad_(True)
    o1 = O.discriminator_forward(sd, real, inp["bbox"], inp["y"], training=True)
    for t, k in zip(o1, ("img", "obj", "app")):
        ref = fx[f"train1_{k}"]
        assert t.shape == ref.shape and maxdiff(t, ref) < 1e-4 * max(1.0, float(np.abs(ref).max())), k
    gen = torch.Generator().manual_seed(6)
    sum((t * torch.randn(t.shape, generator=gen)).sum() for t in o1).backward()
    assert _norms_close(_grad_norms(sd, [str(n) for n in fx["grad_names"]]), fx["grad_norms"])
    assert maxdiff(real.grad[:, :, ::4, ::4], fx["grad_input_sub"]) < 1e-3 * max(1.0, float(np.abs(fx["grad_input_sub"]).max()))
    with torch.no_grad():
        oe = O.discriminator_forward(sd, inp["real"], inp["bbox"], inp["y"], training=False)
    for e, k in zip(oe, ("img", "obj", "app")):
        assert maxdiff(e, fx[f"eval_{k}"]) < 1e-4 * max(1.0, float(np.abs(fx[f"eval_{k}"]).max()))

    fx = load_fixture("train_loop_vg.npz")
    sd_g = O.make_trainable(fixture_state(load_fixture("g_vg_img.npz"), 53))
    sd_d = O.make_trainable(fixture_state(load_fixture("d_vg.npz"), 54))
    tr = O.OracleTrainer(sd_g, sd_d, vg=True)
    for it in range(2):
        inp = recipe.make_inputs_vg(2, 31, 179, 300 + it)
        r = tr.step(inp["real"], inp["y"], inp["bbox"], inp["z"], inp["z_im"])
        rel, tol_img = (1e-4, 5e-5) if it == 0 else (2e-2, 1e-2)
        assert abs(float(r["d_loss"]) - float(fx[f"d_loss{it}"])) < rel * max(1.0, abs(float(fx[f"d_loss{it}"])))
        assert abs(float(r["g_loss"]) - float(fx[f"g_loss{it}"])) < rel * max(1.0, abs(float(fx[f"g_loss{it}"])))
        assert maxdiff(r["fake"][:, :, ::4, ::4], fx[f"fake_sub{it}"]) < tol_img


def test_vgg_loss_matches_reference():
    """utils/util.py:49-94 (VGGLoss over the torchvision VGG19 stack), recipe weights: the pretrained ones need network."""
    fx = load_fixture("vgg.npz")
    x, y = vgg_inputs()
    assert abs(float(x.double().sum()) - float(fx["x_sum"])) < 1e-6
    sd = vgg_state(fx)
    x = x.requires_grad_(True)
    loss = O.vgg_loss(sd, x, y)
    assert abs(float(loss) - float(fx["loss"])) < 1e-5 * max(1.0, abs(float(fx["loss"])))
    loss.backward()
    assert maxdiff(x.grad[:, :, ::2, ::2], fx["grad_x_sub"]) < 1e-4 * float(np.abs(fx["grad_x_sub"]).max())
    feats = O.vgg_features(sd, x.detach())
    assert np.abs(np.array([float(f.mean()) for f in feats]) - fx["tap_means"]).max() < 1e-5
    assert maxdiff(feats[4], fx["tap5"]) < 1e-4 * float(np.abs(fx["tap5"]).max())
