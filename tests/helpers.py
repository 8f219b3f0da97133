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
