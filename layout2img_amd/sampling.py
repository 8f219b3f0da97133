"""Checkpoint loading and the sampling path.

* `load_reference_checkpoint` -- what test_context_app_v2.py:44-59 does with a published `G_*.pth`: strip the
  `module.` prefix nn.DataParallel left on every key, keep the keys this model has, load. The state_dict layout of the
  HIP-path modules is the reference's (weight_orig / weight_u / weight_v, BN buffers), so no conversion is involved.
* `truncated_normal` -- the reference's `truncted_random` (utils/util.py:39-45) draws each of the o*128 latents by
  rejection in a Python loop on the host; the same distribution (N(0,1) restricted to [-thres, thres]) is sampled here
  on the device in one shot through the inverse CDF.
* `sample` -- the eval-mode generator call of test_context_app_v2.py:68-77.
"""
import math
from collections import OrderedDict

import torch


def load_reference_checkpoint(net, state, prefix="module."):
    """state: a state_dict (or a path to one saved with torch.save). Returns (loaded, ignored) key lists."""
    if isinstance(state, (str, bytes)):
        state = torch.load(state, map_location="cpu")
    own = net.state_dict()
    new = OrderedDict()
    ignored = []
    for k, v in state.items():
        name = k[len(prefix):] if k.startswith(prefix) else k
        if name in own and tuple(own[name].shape) == tuple(v.shape):
            new[name] = v
        else:
            ignored.append(k)
    merged = OrderedDict(own)
    merged.update(new)
    net.load_state_dict(merged)
    return list(new), ignored


def truncated_normal(shape, thres=1.0, device="cpu", generator=None, dtype=torch.float32):
    """N(0,1) conditioned on |z| <= thres, via z = sqrt(2) erfinv(u), u ~ U(-erf(t/sqrt2), erf(t/sqrt2))."""
    lim = math.erf(float(thres) / math.sqrt(2.0))
    u = (torch.rand(shape, device=device, generator=generator, dtype=torch.float64) * 2.0 - 1.0) * lim
    return (math.sqrt(2.0) * torch.erfinv(u)).clamp_(-thres, thres).to(dtype)


@torch.no_grad()
def sample(netG, label, bbox, thres=2.0, generator=None):
    """Eval-mode images for layouts (label (b,o) int64, bbox (b,o,4)); truncated latents as the reference draws them."""
    was_training = netG.training
    netG.eval()
    try:
        b, o = label.shape[0], label.shape[1]
        dev = bbox.device if bbox.is_cuda else next(netG.parameters()).device
        z = truncated_normal((b, o, 128), thres, dev, generator)
        z_im = truncated_normal((b, 128), thres, dev, generator)
        return netG(z, bbox.to(dev), z_im=z_im, y=label.to(dev).view(b, o))
    finally:
        netG.train(was_training)
