"""Deterministic parameter recipe shared by tools/capture_goldens.py (which loads the result INTO the
imported reference) and the tests (which load it into the oracle / the HIP modules). Keeps the
committed fixtures small: they hold only key names + shapes, inputs and reference outputs."""
import math

import torch


def make_state_dict(shapes, seed):
    """shapes: {key: tuple} in the reference's state_dict layout. Values depend only on (key order, seed)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        if k.endswith("num_batches_tracked"):
            v = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_var"):
            v = torch.rand(shp, generator=g) * 0.5 + 0.75
        elif k.endswith("running_mean"):
            v = torch.randn(shp, generator=g) * 0.1
        elif k.endswith(("weight_u", "weight_v")):
            v = torch.nn.functional.normalize(torch.randn(shp, generator=g), dim=0)
        elif k.startswith("alpha"):
            v = torch.randn(shp, generator=g) * 0.5
        elif len(shp) == 1 and k.endswith("weight"):      # norm scales
            v = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:                                # biases
            v = 0.1 * torch.randn(shp, generator=g)
        elif "embedding" in k or ".l_y" in k:
            v = torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            v = torch.randn(shp, generator=g) / math.sqrt(fan_in)
        sd[k] = v
    return sd


def make_inputs(b, o, num_classes, seed, size=128):
    """A fixed batch: 2..o real boxes per image, the rest padding (label 0, box [-0.6,-0.6,0.5,0.5])."""
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(b, o, 128, generator=g)
    z_im = torch.randn(b, 128, generator=g)
    real = torch.rand(b, 3, size, size, generator=g) * 2 - 1
    y = torch.zeros(b, o, dtype=torch.long)
    bbox = torch.tensor([-0.6, -0.6, 0.5, 0.5]).repeat(b, o, 1)
    for i in range(b):
        n = max(2, o - 2 - i)
        for j in range(n):
            w, h = (0.15 + 0.75 * torch.rand(2, generator=g)).tolist()
            if j == 0:
                w, h = 0.8, 0.7      # one large ROI (>= 64 px) per image
            if j == 1:
                w, h = 0.2, 0.25     # one small ROI
            x0 = float(torch.rand(1, generator=g)) * (1 - w)
            y0 = float(torch.rand(1, generator=g)) * (1 - h)
            bbox[i, j] = torch.tensor([x0, y0, w, h])
            y[i, j] = int(torch.randint(1, num_classes, (1,), generator=g))
    return dict(z=z, z_im=z_im, real=real, y=y, bbox=bbox)


def make_inputs_vg(b, o, num_classes, seed, size=128):
    """A fixed Visual-Genome-style batch (data/vg.py:118-141): n real boxes, then the `__image__` slot
    (label 0, box [0,0,1,1] -- masked as an attention key, yet its mask covers the canvas and feeds ISLA), then padding
    (label 0, box [-0.6,-0.6,0.5,0.5])."""
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(b, o, 128, generator=g)
    z_im = torch.randn(b, 128, generator=g)
    real = torch.rand(b, 3, size, size, generator=g) * 2 - 1
    y = torch.zeros(b, o, dtype=torch.long)
    bbox = torch.tensor([-0.6, -0.6, 0.5, 0.5]).repeat(b, o, 1)
    for i in range(b):
        n = max(3, o - 4 - 9 * i)
        for j in range(n):
            w, h = (0.1 + 0.6 * torch.rand(2, generator=g)).tolist()
            if j == 0:
                w, h = 0.8, 0.7      # one large ROI (>= 64 px)
            if j == 1:
                w, h = 0.2, 0.25     # one small ROI
            if j == 2:
                w, h = 0.5, 0.3      # exactly 64 px wide: coarse map
            x0 = float(torch.rand(1, generator=g)) * (1 - w)
            y0 = float(torch.rand(1, generator=g)) * (1 - h)
            bbox[i, j] = torch.tensor([x0, y0, w, h])
            y[i, j] = int(torch.randint(1, num_classes, (1,), generator=g))
        bbox[i, n] = torch.tensor([0.0, 0.0, 1.0, 1.0])   # __image__
    return dict(z=z, z_im=z_im, real=real, y=y, bbox=bbox)
