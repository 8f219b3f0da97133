"""CPU: pins oracle/model.py::roi_align (and the two-scale routing) to hand-derived known answers
(tests/roi_cases.py) -- torchvision is not installed here and the reference does not pin its version, so these
closed forms, not a reference output, are what anchors ROIAlign."""
import numpy as np
import pytest
import torch

from oracle import model as O
from tests import roi_cases as RC


@pytest.mark.parametrize("case", RC.known_answer_cases(), ids=lambda c: c["name"])
def test_known_answers(case):
    out = O.roi_align(torch.from_numpy(case["feat"]), torch.from_numpy(case["rois"]), RC.P, case["scale"], 0)
    assert out.shape == case["expected"].shape
    assert np.abs(out.numpy().astype(np.float64) - case["expected"]).max() < 2e-5


@pytest.mark.parametrize("seed,scale,size", [(0, 0.25, 16), (1, 0.125, 16), (2, 0.25, 32), (3, 1.0, 12)])
def test_random_rois_against_the_tent_function_statement(seed, scale, size):
    case = RC.random_case(seed, H=size, W=size, scale=scale)
    out = O.roi_align(torch.from_numpy(case["feat"]), torch.from_numpy(case["rois"]), RC.P, scale, 0)
    assert np.abs(out.numpy().astype(np.float64) - case["expected"]).max() < 5e-5


def test_backward_is_the_transpose_of_forward():
    """one-hot output gradient at (k, c, ph, pw) -> the feature gradient is that bin's sampling weights:
    outer product of the two tent-weight rows (box (10,20)-(73,99) px at 1/4: a 3 x 2 sample grid per bin)."""
    case = RC.known_answer_cases()[1]
    feat = torch.from_numpy(case["feat"]).clone().requires_grad_(True)
    out = O.roi_align(feat, torch.from_numpy(case["rois"]), RC.P, case["scale"], 0)
    r = case["rois"][0]
    Ay = RC.tent_weights(r[2], r[4], 32, case["scale"])
    Ax = RC.tent_weights(r[1], r[3], 32, case["scale"])
    for c, ph, pw in ((0, 0, 0), (1, 3, 5), (0, 7, 7)):
        (g,) = torch.autograd.grad(out[0, c, ph, pw], feat, retain_graph=True)
        expect = np.zeros(case["feat"].shape)
        expect[0, c] = np.outer(Ay[ph], Ax[pw])
        assert np.abs(g.numpy() - expect).max() < 1e-6
        assert abs(float(g.sum()) - 1.0) < 1e-5   # interior bin: the weights of a mean of interpolations sum to 1


def test_two_scale_routing():
    rois = torch.tensor([[0.0, *box] for box, _ in RC.ROUTING])
    assert O.route_small(rois).tolist() == [s for _, s in RC.ROUTING]
