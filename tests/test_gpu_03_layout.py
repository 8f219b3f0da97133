"""The layout-side kernels (csrc/layout.hip) through the C ABI, forward and backward: box geometry, mask layout and box
indicator against the oracle's own functions (oracle/model.py: box_relational_embedding, masks_to_layout, bbox_mask -- run on the
CPU, they are the pinned restatement of the reference), the rest against the chains of torch ops they replace."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _boxes(b, o, g):
    """COCO-like boxes with padding slots [-0.6, -0.6, 0.5, 0.5] (data/cocostuff_loader.py:301-303)."""
    wh = torch.rand(b, o, 2, generator=g) * 0.75 + 0.15
    xy = torch.rand(b, o, 2, generator=g) * (1 - wh)
    box = torch.cat((xy, wh), dim=-1)
    box[:, -2:] = torch.tensor([-0.6, -0.6, 0.5, 0.5])
    return box


def test_box_geometry_matches_the_torch_chain():
    from layout2img_amd import ops
    from oracle import model as O
    g = torch.Generator().manual_seed(0)
    b, o = 5, 8
    bbox = _boxes(b, o, g)
    lin = torch.nn.Linear(64, 1)
    ref = F.relu(lin(O.box_relational_embedding(bbox).view(-1, 64))).view(b, o, o)   # the oracle's function, on the CPU
    gout = torch.randn(b, o, o, generator=g)
    rw, rb = torch.autograd.grad(ref, (lin.weight, lin.bias), gout)
    ref = ref.detach()
    dlin = torch.nn.Linear(64, 1).to(_dev())
    dlin.load_state_dict(lin.state_dict())
    out = ops.box_geometry(bbox.to(_dev()), dlin.weight, dlin.bias)
    dw, db = torch.autograd.grad(out, (dlin.weight, dlin.bias), gout.to(_dev()))
    # (sin / cos of arguments up to ~700 rad: the device's f32 argument reduction differs from the host's in the last bits)
    assert float((out.detach().cpu() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
    assert float((dw.cpu() - rw).abs().max()) < 5e-4 * max(1.0, float(rw.abs().max()))
    assert float((db.cpu() - rb).abs().max()) < 5e-4 * max(1.0, float(rb.abs().max()))


@pytest.mark.parametrize("o", [8, 31])
def test_layout_masks_match_grid_sample_and_bbox_mask(o):
    from layout2img_amd import ops
    from oracle import model as O
    g = torch.Generator().manual_seed(1)
    b, M, H = 3, 16, 64
    bbox = _boxes(b, o, g)
    m0 = torch.randn(b * o, M, M, 8, generator=g)
    mc = m0.clone().requires_grad_(True)
    ref = O.masks_to_layout(bbox, torch.sigmoid(mc[..., 0]).view(b, o, M, M), H)   # the oracle's functions, on the CPU
    gout = torch.randn(b, o, H, H, generator=g)
    (rm,) = torch.autograd.grad(ref, mc, gout)
    ref = ref.detach()
    m = m0.to(_dev()).requires_grad_(True)
    out, boxm = ops.layout_masks(m, bbox.to(_dev()), H, True)
    (dm,) = torch.autograd.grad(out, m, gout.to(_dev()))
    assert float((out.cpu() - ref).abs().max()) < 5e-6
    # the rectangle indicator is exact except where a pixel centre sits ON a box edge to within one rounding of (lin - x0) / w
    want = O.bbox_mask(bbox, H, H)
    diff = boxm.cpu() != want
    if bool(diff.any()):
        lin = torch.linspace(0, 1, steps=H)
        X = (lin.view(1, 1, 1, H) - bbox[..., 0].view(b, o, 1, 1)) / bbox[..., 2].view(b, o, 1, 1)
        Y = (lin.view(1, 1, H, 1) - bbox[..., 1].view(b, o, 1, 1)) / bbox[..., 3].view(b, o, 1, 1)
        edge = (torch.minimum(X.abs(), (X - 1).abs()) < 1e-6) | (torch.minimum(Y.abs(), (Y - 1).abs()) < 1e-6)
        assert bool(edge.expand_as(diff)[diff].all()) and int(diff.sum()) <= 4
    assert float((dm.cpu() - rm).abs().max()) < 1e-5 * max(1.0, float(rm.abs().max()))
    assert float(dm[..., 1:].abs().max()) == 0.0


@pytest.mark.parametrize("perm", [False, True])
def test_add_layernorm(perm):
    from layout2img_amd import ops
    g = torch.Generator().manual_seed(2)
    B, O, D, ld = 4, 8, 308, 312
    ln = torch.nn.LayerNorm(D).to(_dev())
    with torch.no_grad():
        ln.weight.copy_(torch.randn(D, generator=g))
        ln.bias.copy_(torch.randn(D, generator=g))
    a = torch.randn(B, O, D, generator=g).to(_dev()).requires_grad_(True)
    bp = F.pad(torch.randn(B * O, D, generator=g), (0, ld - D)).view(B * O, 1, 1, ld).to(_dev()).requires_grad_(True)
    a_ref = a.transpose(1, 2).contiguous().view(B, -1, D) if perm else a          # reference :197-198
    ref = ln(a_ref + bp.view(B, O, ld)[..., :D])
    gout = torch.randn(B, O, D, generator=g).to(_dev())
    ra, rb, rg, rbe = torch.autograd.grad(ref, (a, bp, ln.weight, ln.bias), gout)
    y = ops.add_layernorm(a, bp, ln, D, ld, torch.bfloat16, perm_O=O if perm else 0)
    gy = F.pad(gout.view(B * O, D), (0, ld - D)).view(B * O, 1, 1, ld)
    da, db, dg, dbe = torch.autograd.grad(y, (a, bp, ln.weight, ln.bias), gy)
    yv = y.view(B, O, ld)
    assert float((yv[..., :D] - ref).abs().max()) < 2e-5 and float(yv[..., D:].abs().max()) == 0.0
    assert ops._sibling(y, "raw", torch.bfloat16) is not None
    assert float((ops._sibling(y, "raw", torch.bfloat16).float() - y).abs().max()) < 2 ** -8 * float(y.abs().max())
    for got, want in ((da, ra), (db, rb), (dg, rg), (dbe, rbe)):
        assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))


def test_latent_and_key_mask():
    from layout2img_amd import ops
    g = torch.Generator().manual_seed(3)
    b, o, ld = 3, 8, 312
    emb = torch.nn.Embedding(184, 180).to(_dev())
    z = torch.randn(b, o, 128, generator=g).to(_dev())
    y = torch.randint(0, 184, (b, o), generator=g).to(_dev())
    y[:, -1] = 0
    ref = torch.cat((z.view(b * o, -1), emb(y).view(b * o, -1)), dim=1)
    gout = torch.randn(b * o, 1, 1, ld, generator=g).to(_dev())
    (re,) = torch.autograd.grad(ref, emb.weight, gout.view(b * o, ld)[:, :308])
    out, kv = ops.latent(z, emb.weight, y, ld, torch.bfloat16)
    (de,) = torch.autograd.grad(out, emb.weight, gout)
    assert torch.equal(out.view(b * o, ld)[:, :308], ref) and float(out.view(b * o, ld)[:, 308:].abs().max()) == 0.0
    assert torch.equal(kv, (y != 0).to(torch.int32))
    assert float((de - re).abs().max()) < 1e-5


def test_fc_to_nhwc_and_tanh_nchw():
    from layout2img_amd import ops
    g = torch.Generator().manual_seed(4)
    N, C = 6, 256
    x = torch.randn(N, 1, 1, C * 16, generator=g).to(_dev()).requires_grad_(True)
    ref = x.view(N, C, 4, 4).permute(0, 2, 3, 1)
    out = ops.fc_to_nhwc(x, C, torch.bfloat16)
    assert torch.equal(out, ref.contiguous())
    gout = torch.randn(N, 4, 4, C, generator=g).to(_dev())
    (rx,) = torch.autograd.grad(ref, x, gout)
    (dx,) = torch.autograd.grad(out, x, gout)
    assert torch.equal(dx, rx)
    pre = torch.randn(2, 16, 16, 8, generator=g).to(_dev()).requires_grad_(True)
    ref = torch.tanh(pre[..., :3]).permute(0, 3, 1, 2)
    img = ops.tanh_nchw(pre, 3, torch.bfloat16)
    gi = torch.randn(2, 3, 16, 16, generator=g).to(_dev())
    (rp,) = torch.autograd.grad(ref, pre, gi)
    (dp,) = torch.autograd.grad(img, pre, gi)
    assert float((img - ref).abs().max()) < 1e-6 and float((dp - rp).abs().max()) < 1e-6


@pytest.mark.parametrize("training", [True, False])
def test_psp_stages(training):
    from layout2img_amd import ops
    g = torch.Generator().manual_seed(5)
    B, C, Fo, sizes = 4, 128, 100, (1, 2, 3, 6)
    NB = sum(s * s for s in sizes)
    convs = [torch.nn.Conv2d(C, Fo, 1, bias=False).to(_dev()) for _ in sizes]
    bns = [torch.nn.BatchNorm2d(Fo).to(_dev()) for _ in sizes]
    with torch.no_grad():
        for bn in bns:
            bn.weight.copy_(torch.rand(Fo, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(Fo, generator=g) * 0.1)
            bn.running_mean.copy_(torch.randn(Fo, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(Fo, generator=g) + 0.5)
    pooled = torch.randn(B, NB, C, generator=g).to(_dev()).requires_grad_(True)
    rm0 = [bn.running_mean.clone() for bn in bns]
    rv0 = [bn.running_var.clone() for bn in bns]
    ys = []
    for conv, bn, part in zip(convs, bns, pooled.split([s * s for s in sizes], dim=1)):
        y = part.reshape(-1, C) @ conv.weight.view(Fo, C).t()
        y = F.relu(F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps))
        ys.append(y.view(B, -1, Fo))
    ref = torch.cat(ys, dim=1)
    gout = torch.randn(B, NB, Fo, generator=g).to(_dev())
    params = [c.weight for c in convs] + [b.weight for b in bns] + [b.bias for b in bns]
    rgrads = torch.autograd.grad(ref, [pooled] + params, gout)
    rm1 = [bn.running_mean.clone() for bn in bns]
    rv1 = [bn.running_var.clone() for bn in bns]
    with torch.no_grad():
        for bn, m, v in zip(bns, rm0, rv0):
            bn.running_mean.copy_(m), bn.running_var.copy_(v)
    out = ops.psp_stages(pooled, convs, bns, sizes, training)
    grads = torch.autograd.grad(out, [pooled] + params, gout)
    assert float((out - ref).abs().max()) < 3e-5 * max(1.0, float(ref.abs().max()))
    for bn, m, v in zip(bns, rm1, rv1):
        assert float((bn.running_mean - m).abs().max()) < 1e-5 and float((bn.running_var - v).abs().max()) < 1e-5
    for got, want in zip(grads, rgrads):
        assert got.shape == want.shape
        assert float((got - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("two_scale", [True, False])
@pytest.mark.parametrize("shape", [(32, 8), (32, 31), (3, 5), (40, 31), (129, 8)])   # the last two: more than 1024 rows
def test_roi_layout_is_the_reference_order(shape, two_scale):
    """the compacted ROI rows against the torch composition the host side used (stable argsort of the same key)"""
    from layout2img_amd import ops
    b, o = shape
    g = torch.Generator().manual_seed(6)
    bbox = _boxes(b, o, g).to(_dev())
    bbox[0, 0] = torch.tensor([0.1, 0.1, 0.5, 0.25], device=_dev())       # exactly 64 px wide at size 128: a large ROI
    label = torch.randint(1, 100, (b, o, 1), generator=g).to(_dev())
    label[:, -2:] = 0
    label[1, 0] = 0
    size = 128
    xyxy = torch.stack((bbox[..., 0], bbox[..., 1], bbox[..., 0] + bbox[..., 2], bbox[..., 1] + bbox[..., 3]), dim=-1) * size
    idx = torch.arange(b, device=_dev(), dtype=torch.float32).view(b, 1, 1).expand(b, o, 1)
    rois = torch.cat((idx, xyxy), dim=2).view(-1, 5)
    y = label.reshape(-1)
    valid = y != 0
    key = (~valid).to(torch.int64) * 2
    if two_scale:
        key = key + (((rois[:, 3] - rois[:, 1]) < 64) & ((rois[:, 4] - rois[:, 2]) < 64)).to(torch.int64)
    order = torch.argsort(key, stable=True)
    r2, y2, v2, c2 = ops.roi_layout(bbox, label, size, two_scale)
    assert torch.equal(r2, rois[order]) and torch.equal(y2, y[order])
    assert torch.equal(v2, valid[order].to(torch.int32)) and int(c2) == int(valid.sum())


def test_image_nhwc_resize_adjoint_and_dropout():
    from layout2img_amd import ops
    g = torch.Generator().manual_seed(7)
    img = torch.randn(3, 3, 16, 16, generator=g).to(_dev()).requires_grad_(True)
    x, xs = ops.image_nhwc(img, 8, torch.bfloat16, True)
    rx = F.pad(img.permute(0, 2, 3, 1), (0, 5))
    rxs = F.avg_pool2d(rx.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert torch.equal(x, rx) and float((xs - rxs).abs().max()) < 1e-6
    assert torch.equal(ops._sibling(x, "raw", torch.bfloat16), rx.to(torch.bfloat16))
    g1, g2 = torch.randn(x.shape, generator=g).to(_dev()), torch.randn(xs.shape, generator=g).to(_dev())
    (d,) = torch.autograd.grad((x, xs), img, (g1, g2))
    (r,) = torch.autograd.grad((rx, rxs), img, (g1, g2))
    assert float((d - r).abs().max()) < 1e-6
    # adjoint of the bilinear resize (down and up)
    for h, H in ((64, 16), (64, 128), (64, 32)):
        m = torch.randn(2, 3, h, h, generator=g).to(_dev()).requires_grad_(True)
        go = torch.randn(2, 3, H, H, generator=g).to(_dev())
        (d,) = torch.autograd.grad(ops.resize_bilinear(m, H, H), m, go)
        (r,) = torch.autograd.grad(F.interpolate(m, size=(H, H), mode="bilinear"), m, go)
        assert float((d - r).abs().max()) < 1e-5
    # Dropout2d scale
    y = torch.randn(2, 4, 4, 8, generator=g).to(_dev()).requires_grad_(True)
    u = torch.rand(2, 8, generator=g).to(_dev())
    out = ops.channel_dropout(y, u, 0.25)
    ref = y * ((u >= 0.25).float() / 0.75).view(2, 1, 1, 8)
    go = torch.randn(y.shape, generator=g).to(_dev())
    assert float((out - ref).abs().max()) < 1e-6
    assert float((torch.autograd.grad(out, y, go)[0] - torch.autograd.grad(ref, y, go)[0]).abs().max()) < 1e-6


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("S,C,N", [(4, 256, 7), (8, 256, 5), (8, 128, 3), (4, 320, 2)])
def test_in_relu_up2_matches_the_torch_chain(S, C, N, dt):
    """InstanceNorm2d -> ReLU -> F.interpolate(x2, bilinear, align_corners=False) (reference model/mask_regression.py:64-95)
    forward and backward in one launch each, incl. the operand copies both directions carry."""
    from layout2img_amd import ops
    g = torch.Generator().manual_seed(S * 1000 + C)
    x = torch.randn(N, S, S, C, generator=g)
    gout = torch.randn(N, 2 * S, 2 * S, C, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = F.interpolate(torch.relu(F.instance_norm(xr.permute(0, 3, 1, 2), eps=1e-5)), scale_factor=2, mode="bilinear", align_corners=False)
    ref = ref.permute(0, 2, 3, 1)
    ref.backward(gout)
    xd = x.to(_dev()).requires_grad_(True)
    out = ops.in_relu_up2(xd, 1e-5, dt)
    out.backward(gout.to(_dev()))
    err = (out.detach().cpu() - ref.detach()).abs()
    assert float(err.max()) < 2e-5
    op = ops._sibling(out, "raw", dt)
    assert op is not None and float((op.float().cpu() - ref.detach()).abs().max()) < (2e-5 if dt == torch.float32 else 2e-2)
    assert float((xd.grad.cpu() - xr.grad).abs().max()) < 2e-4 * max(1.0, float(xr.grad.abs().max()))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,S,C", [(62, 4, 128), (62, 8, 128), (3, 8, 64), (2, 16, 8)])
def test_up2_nhwc_matches_interpolate(N, S, C, dt):
    """ops.up2_nhwc == F.interpolate(scale 2, bilinear, align_corners=False) on NHWC maps, forward and backward (the VG generator's
    MaskRegressNet, reference model/mask_regression.py:20-33,42-58), with the operand copies of the result and of dx."""
    import torch.nn.functional as F
    from layout2img_amd import ops
    g = torch.Generator().manual_seed(N * 100 + S)
    x = torch.randn(N, S, S, C, generator=g)
    gy = torch.randn(N, 2 * S, 2 * S, C, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = F.interpolate(xr.permute(0, 3, 1, 2), size=(2 * S, 2 * S), mode="bilinear").permute(0, 2, 3, 1)
    ref.backward(gy)
    xd = x.to("cuda:0").requires_grad_(True)
    out = ops.up2_nhwc(xd, dt)
    out.backward(gy.to("cuda:0"))
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 1e-5
    assert float((xd.grad.cpu() - xr.grad).abs().max()) < 1e-5
    op = ops._sibling(out, "raw", dt)
    assert op is not None and op.dtype == dt and float((op.float().cpu() - ref.detach()).abs().max()) < (1e-5 if dt == torch.float32 else 4e-2)
