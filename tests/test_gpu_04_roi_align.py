"""GPU twin of tests/test_oracle_roi_align.py: the HIP ROIAlign (l2i_roi_align_fwd / _bwd through the C ABI) against
the same hand-derived known answers -- not against the oracle."""
import numpy as np
import pytest
import torch

from tests import roi_cases as RC

pytestmark = pytest.mark.gpu


def _run(feat, rois, scale, feat_l=None, scale_l=1.0, thr=1e30, need_grad=False):
    from layout2img_amd import ops
    dev = torch.device("cuda:0")
    f = torch.from_numpy(feat).permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(need_grad)     # NHWC
    fl = None if feat_l is None else torch.from_numpy(feat_l).permute(0, 2, 3, 1).contiguous().to(dev)
    out = ops.roi_align(f, fl, torch.from_numpy(rois).to(dev), None, RC.P, scale, scale_l, thr, 0)
    return out, f


def _pad4(feat):
    """the kernel wants C % 4 == 0: repeat the channels"""
    c = feat.shape[1]
    reps = (4 + c - 1) // c
    return np.concatenate([feat] * reps, axis=1)[:, :max(4, c)]


@pytest.mark.parametrize("case", RC.known_answer_cases(), ids=lambda c: c["name"])
def test_known_answers(case):
    c = case["feat"].shape[1]
    out, _ = _run(_pad4(case["feat"]), case["rois"], case["scale"])
    got = out.permute(0, 3, 1, 2)[:, :c].cpu().numpy().astype(np.float64)
    assert np.abs(got - case["expected"]).max() < 2e-5


@pytest.mark.parametrize("seed,scale,size", [(0, 0.25, 16), (1, 0.125, 16), (2, 0.25, 32), (3, 1.0, 12)])
def test_random_rois_against_the_tent_function_statement(seed, scale, size):
    case = RC.random_case(seed, H=size, W=size, scale=scale)
    out, _ = _run(case["feat"], case["rois"], scale)
    assert np.abs(out.permute(0, 3, 1, 2).cpu().numpy().astype(np.float64) - case["expected"]).max() < 5e-5


def test_backward_is_the_transpose_of_forward():
    case = RC.known_answer_cases()[1]
    feat = _pad4(case["feat"])
    r = case["rois"][0]
    Ay = RC.tent_weights(r[2], r[4], 32, case["scale"])
    Ax = RC.tent_weights(r[1], r[3], 32, case["scale"])
    for c, ph, pw in ((0, 0, 0), (1, 3, 5), (0, 7, 7)):
        out, f = _run(feat, case["rois"], case["scale"], need_grad=True)
        (g,) = torch.autograd.grad(out[0, ph, pw, c], f)
        expect = np.zeros((1, 32, 32, 4))
        expect[0, :, :, c] = np.outer(Ay[ph], Ax[pw])
        assert np.abs(g.cpu().numpy() - expect).max() < 1e-6


def test_two_scale_routing_and_validity():
    """fine map = 1, coarse map = 2: the output value says which map an ROI pooled from
    (model/rcnn_discriminator_app.py:131: fine iff width < 64 AND height < 64); invalid rows give zeros."""
    from layout2img_amd import ops
    dev = torch.device("cuda:0")
    fs, fl = torch.full((1, 32, 32, 4), 1.0, device=dev), torch.full((1, 16, 16, 4), 2.0, device=dev)
    rois = torch.tensor([[0.0, *box] for box, _ in RC.ROUTING] + [[0.0, 0.0, 0.0, 10.0, 10.0]], device=dev)
    valid = torch.tensor([1] * len(RC.ROUTING) + [0], dtype=torch.int32, device=dev)
    out = ops.roi_align(fs, fl, rois, valid, 8, 0.25, 0.125, 64.0, 0)
    vals = out.mean(dim=(1, 2, 3)).cpu().tolist()
    assert vals[:-1] == [1.0 if s else 2.0 for _, s in RC.ROUTING]
    assert vals[-1] == 0.0


@pytest.mark.parametrize("size,C", [(128, 64), (64, 32), (128, 512)])
def test_gather_backward_equals_scatter_backward(size, C):
    """l2i_roi_align_bwd with fresh = 1 (gather form: every pixel of both gradient maps written once by the workgroup that owns it,
    plus the bf16 copies) against fresh = 0 (the separable scatter with atomics into cleared maps) on the two-scale layout of
    model/rcnn_discriminator_app.py:131-145: boxes that straddle the borders, one-pixel boxes, large boxes on the coarse map,
    padding rows, an image without boxes."""
    from layout2img_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(size + C)
    B, o, P = 4, 6, 8
    Hs, Hl = size // 4, size // 8
    xy = torch.rand(B, o, 2, generator=g) * size * 0.9 - 0.05 * size
    wh = torch.rand(B, o, 2, generator=g) * size * 0.8 + 1.0
    wh[0, 0] = 1.0                                    # a one-pixel box
    wh[1, 1] = torch.tensor([size * 0.95, 70.0])      # wide: coarse map
    xy[1, 2], wh[1, 2] = torch.tensor([-9.0, -9.0]), torch.tensor([30.0, 30.0])             # straddles the top-left corner
    xy[2, 0], wh[2, 0] = torch.tensor([size - 20.0, size - 20.0]), torch.tensor([40.0, 40.0])   # ... the bottom-right one
    rois = torch.cat([torch.arange(B).view(B, 1, 1).expand(B, o, 1).float(), xy, xy + wh], dim=2).reshape(-1, 5)
    valid = torch.ones(B * o, dtype=torch.int32)
    valid[3 * o:] = 0                                 # image 3: no boxes at all
    valid[5] = 0
    R = B * o
    dout = torch.randn(R, P, P, C, generator=g)
    rois_d, valid_d, dout_d = rois.to(dev).contiguous(), valid.to(dev), dout.to(dev)

    def run(fresh):
        fill = float("nan") if fresh else 0.0
        ds = torch.full((B, Hs, Hs, C), fill, device=dev)
        dl = torch.full((B, Hl, Hl, C), fill, device=dev)
        ops_s = torch.empty((B, Hs, Hs, C), dtype=torch.bfloat16, device=dev) if fresh else None
        ops_l = torch.empty((B, Hl, Hl, C), dtype=torch.bfloat16, device=dev) if fresh else None
        _lib.call("l2i_roi_align_bwd", rois_d.data_ptr(), valid_d.data_ptr(), dout_d.data_ptr(), ds.data_ptr(), dl.data_ptr(), R, C, P,
                  Hs, Hs, 0.25, Hl, Hl, 0.125, 64.0, 0, B, fresh, ops_s.data_ptr() if fresh else None, ops_l.data_ptr() if fresh else None,
                  torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return ds, dl, ops_s, ops_l
    s0, l0, _, _ = run(0)
    s1, l1, o_s, o_l = run(1)
    assert bool(torch.isfinite(s1).all()) and bool(torch.isfinite(l1).all())   # every pixel written (the maps started as NaN)
    assert float(s0.abs().max()) > 0 and float(l0.abs().max()) > 0
    for a, b_ in ((s1, s0), (l1, l0)):
        assert float((a - b_).abs().max()) < 2e-5 * float(b_.abs().max())
    assert float(s1[3].abs().max()) == 0.0 and float(l1[3].abs().max()) == 0.0
    assert torch.equal(o_s, s1.to(torch.bfloat16)) and torch.equal(o_l, l1.to(torch.bfloat16))


def test_bf16_copies_are_refused_where_the_gather_form_cannot_run():
    """The bf16 copies of the gradient maps come out of the gather kernel's stores only: a geometry it does not take (C % 32 != 0 here)
    with the copies requested is an error, not a silently unwritten buffer."""
    from layout2img_amd import _lib
    dev = torch.device("cuda:0")
    B, R, C, P, Hs = 1, 2, 20, 8, 16
    rois = torch.tensor([[0, 2.0, 2.0, 30.0, 30.0], [0, 8.0, 8.0, 20.0, 20.0]], device=dev)
    dout = torch.randn(R, P, P, C, device=dev)
    ds = torch.empty(B, Hs, Hs, C, device=dev)
    cp = torch.empty(B, Hs, Hs, C, dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError):
        _lib.call("l2i_roi_align_bwd", rois.data_ptr(), None, dout.data_ptr(), ds.data_ptr(), None, R, C, P, Hs, Hs, 0.25, 0, 0, 0.0, 1e30, 0,
                  B, 1, cp.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    _lib.call("l2i_roi_align_bwd", rois.data_ptr(), None, dout.data_ptr(), ds.data_ptr(), None, R, C, P, Hs, Hs, 0.25, 0, 0, 0.0, 1e30, 0,
              B, 1, None, None, torch.cuda.current_stream().cuda_stream)   # (without the copies: the scatter form, maps cleared by the library)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ds).all()) and float(ds.abs().max()) > 0
