import os

import numpy as np
import torch

from tests.golden import recipe

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def fixture_shapes(fx):
    return {str(k): tuple(int(x) for x in str(s).split(",") if x != "") for k, s in zip(fx["keys"], fx["shapes"])}


def fixture_state(fx, seed):
    return recipe.make_state_dict(fixture_shapes(fx), seed)


def fixture_inputs(fx):
    return {k: torch.from_numpy(fx[k]) for k in ("z", "z_im", "real", "y", "bbox")}


def maxdiff(a, b):
    a = a.detach().cpu().float() if torch.is_tensor(a) else torch.as_tensor(a).float()
    b = b.detach().cpu().float() if torch.is_tensor(b) else torch.as_tensor(b).float()
    return float((a - b).abs().max())


def vgg_inputs():
    """the (fake, real) pair tools/capture_goldens.py::capture_vgg fed to the reference's VGGLoss"""
    g = torch.Generator().manual_seed(62)
    x = torch.rand(2, 3, 128, 128, generator=g) * 2 - 1
    y = torch.rand(2, 3, 128, 128, generator=g) * 2 - 1
    return x, y


def vgg_state(fx):
    sd = fixture_state(fx, 61)
    return {k: (v * (2.0 ** 0.5) if k.endswith("weight") else v) for k, v in sd.items()}
