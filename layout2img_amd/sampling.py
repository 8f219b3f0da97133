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
import os
from collections import OrderedDict

import torch


def _flush(net):
    f = getattr(net, "_l2i_flush", None)
    if f is not None:
        f()


def _opt_state(o):
    """Adam moments per PARAMETER NAME (sliced out of the flat buffers by the network's own offsets): independent of the
    flat layout (alignment, parameter order), which has changed between revisions."""
    flat = o.net.flat
    out = {}
    for name, p in flat._params:
        off, k = flat.offsets[name], p.numel()
        out[name] = (o.m[off:off + k].detach().cpu().clone().view(p.shape), o.v[off:off + k].detach().cpu().clone().view(p.shape))
    return dict(state=out, t=int(o.t_dev.item()), lr=o.lr, betas=tuple(o.betas), eps=o.eps)   # (the device-side count: graph replays advance only that one)


def _load_opt_state(o, st, restore_hyper=False):
    """restore_hyper: also take lr / betas / eps from the file (default: keep the optimizer's own, i.e. the command line's
    --g_lr / --d_lr win on resume). Files of earlier revisions (flat moments 'm' / 'v' / 't') load when the flat layout still
    matches."""
    flat = o.net.flat
    if "state" not in st and "m" in st and "v" in st:   # legacy: the flat moment buffers themselves
        m, v = st["m"], st["v"]
        if m.numel() != o.m.numel() or v.numel() != o.v.numel():
            raise RuntimeError("optimizer checkpoint in the legacy flat format does not match this network's flat parameter "
                               f"layout ({m.numel()} vs {o.m.numel()} floats): re-save it with the current revision")
        with torch.no_grad():
            o.m.copy_(m.reshape(-1)), o.v.copy_(v.reshape(-1))
    else:
        if "state" not in st:
            raise RuntimeError(f"optimizer checkpoint has neither 'state' nor legacy 'm' / 'v' entries (keys: {sorted(st)})")
        names = {n for n, _ in flat._params}
        if set(st["state"]) != names:
            raise RuntimeError("optimizer checkpoint does not match this network: "
                               f"{len(names - set(st['state']))} parameters missing, {len(set(st['state']) - names)} unknown")
        with torch.no_grad():
            o.m.zero_(), o.v.zero_()
            for name, p in flat._params:
                off, k = flat.offsets[name], p.numel()
                m, v = st["state"][name]
                if tuple(m.shape) != tuple(p.shape):
                    raise RuntimeError(f"optimizer checkpoint: {name} has shape {tuple(m.shape)}, the model {tuple(p.shape)}")
                o.m[off:off + k].copy_(m.reshape(-1))
                o.v[off:off + k].copy_(v.reshape(-1))
    o.t = int(st["t"])
    o.t_dev.fill_(o.t)
    if restore_hyper and "lr" in st:
        o.lr, o.betas, o.eps = float(st["lr"]), tuple(st["betas"]), float(st["eps"])


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


def reference_state_dict(net, prefix="module."):
    """`net.state_dict()` as the reference writes it (train_context_app_v2.py:215-217: the nets are wrapped in
    nn.DataParallel there, hence the `module.` prefix): every entry cloned into its own CPU storage -- the parameters of
    a finalized network are views into ONE flat buffer, which must not be what ends up in the file.
    A deferred optimizer step of a data-parallel trainer (GanTrainer.defer_g) is completed first."""
    _flush(net)
    return OrderedDict((prefix + k, v.detach().cpu().clone()) for k, v in net.state_dict().items())


def save_checkpoint(out_path, epoch, netG, netD, g_opt=None, d_opt=None, prefix="module."):
    """`out_path/model/G_<epoch>.pth`, `D_<epoch>.pth` in the reference's layout (loadable by the reference's resume code,
    train_context_app_v2.py:71-105, and by its sampling script) plus -- which the reference omits -- the Adam state of
    both optimizers in `opt_<epoch>.pth`. Under data parallelism call it on rank 0 only."""
    d = os.path.join(out_path, "model")
    os.makedirs(d, exist_ok=True)
    paths = {}
    for name, net in (("G", netG), ("D", netD)):
        paths[name] = os.path.join(d, f"{name}_{epoch}.pth")
        torch.save(reference_state_dict(net, prefix), paths[name])
    if g_opt is not None and d_opt is not None:
        paths["opt"] = os.path.join(d, f"opt_{epoch}.pth")
        torch.save({k: _opt_state(o) for k, o in (("G", g_opt), ("D", d_opt))}, paths["opt"])
    return paths


def load_checkpoint(out_path, epoch, netG, netD, g_opt=None, d_opt=None, prefix="module.", restore_hyper=False):
    """Resume (train_context_app_v2.py:71-105): strip the prefix, keep the keys the models have, load; restore the Adam
    state (moments, step count) when `opt_<epoch>.pth` exists -- lr / betas / eps stay the optimizers' own (the command
    line's) unless restore_hyper. Returns the epoch to continue from."""
    d = os.path.join(out_path, "model")
    for name, net in (("G", netG), ("D", netD)):
        load_reference_checkpoint(net, os.path.join(d, f"{name}_{epoch}.pth"), prefix)
    p = os.path.join(d, f"opt_{epoch}.pth")
    if g_opt is not None and d_opt is not None and os.path.exists(p):
        st = torch.load(p, map_location="cpu")
        for k, o in (("G", g_opt), ("D", d_opt)):
            _load_opt_state(o, st[k], restore_hyper)   # per parameter name
    return epoch


def truncated_normal(shape, thres=1.0, device="cpu", generator=None, dtype=torch.float32):
    """N(0,1) conditioned on |z| <= thres, via z = sqrt(2) erfinv(u), u ~ U(-erf(t/sqrt2), erf(t/sqrt2))."""
    lim = math.erf(float(thres) / math.sqrt(2.0))
    u = (torch.rand(shape, device=device, generator=generator, dtype=torch.float64) * 2.0 - 1.0) * lim
    return (math.sqrt(2.0) * torch.erfinv(u)).clamp_(-thres, thres).to(dtype)


class GraphSampler:
    """`sample` as a replayed HIP graph, one per layout shape (b, o): the eval-mode generator call of the reference's sampling
    script (test_context_app_v2.py:68-77) is ~100 launches of which a batch of one fills a fraction of a CU each -- its
    latency is launch overhead, not work (2.7 ms eager against 2.3 ms for a batch of 32). The captured graph holds the draw of the
    truncated latents (torch's graph-safe Philox generator advances per replay, so every call returns a new sample) and the
    whole forward; label / bbox are copied into static inputs. The weight packs are NOT part of the graph: they are the
    arena's cached eval-mode pass (arena.WeightArena._eval_pass), re-packed in place -- eagerly, in front of the replay --
    when a parameter changed, so a training step between two calls is picked up without a new capture.

    s = GraphSampler(netG); img = s(label, bbox)     # img: a static tensor, overwritten by the next call of that shape
    """

    def __init__(self, netG, thres=2.0):
        self.net, self.thres, self._graphs = netG, float(thres), {}
        self._stream = None

    @torch.no_grad()
    def __call__(self, label, bbox, return_latents=False):
        net = self.net
        _flush(net)
        dev = next(net.parameters()).device
        b, o = label.shape[0], label.shape[1]
        key = (b, o)
        was_training = net.training
        net.eval()
        try:
            net.arena.prepare(training=False)   # the cached eval pass: created, or re-packed in place if a parameter changed
            ent = self._graphs.get(key)
            if ent is None:
                ent = self._capture(b, o, dev, label, bbox)
                self._graphs[key] = ent
            ent["label"].copy_(label.view(b, o), non_blocking=True)
            ent["bbox"].copy_(bbox, non_blocking=True)
            ent["graph"].replay()
            return (ent["img"], ent["z"], ent["z_im"]) if return_latents else ent["img"]
        finally:
            net.train(was_training)

    def _capture(self, b, o, dev, label, bbox):
        net = self.net
        ent = dict(label=label.to(dev).view(b, o).clone(), bbox=bbox.to(dev).float().clone())
        cur = torch.cuda.current_stream()
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        side = self._stream
        from . import _lib

        def run():
            z = truncated_normal((b, o, 128), self.thres, dev)
            z_im = truncated_normal((b, 128), self.thres, dev)
            return net(z, ent["bbox"], z_im=z_im, y=ent["label"]), z, z_im
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            _lib.workspace(dev)     # the stream's reduction workspace exists BEFORE the capture (never in the graph's pool)
            for _ in range(2):      # warm-up on the capture stream, as torch.cuda.graphs asks
                run()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            ent["img"], ent["z"], ent["z_im"] = run()
        ent["graph"] = graph
        return ent


@torch.no_grad()
def sample(netG, label, bbox, thres=2.0, generator=None, return_latents=False):
    """Eval-mode images for layouts (label (b,o) int64, bbox (b,o,4)); truncated latents as the reference draws them
    (test_context_app_v2.py:68-77: z_obj (b,o,128) and z_im (b,128), both truncated at `thres` = 2)."""
    _flush(netG)   # (a deferred generator step of a data-parallel trainer)
    was_training = netG.training
    netG.eval()
    try:
        b, o = label.shape[0], label.shape[1]
        dev = bbox.device if bbox.is_cuda else next(netG.parameters()).device
        z = truncated_normal((b, o, 128), thres, dev, generator)
        z_im = truncated_normal((b, 128), thres, dev, generator)
        img = netG(z, bbox.to(dev), z_im=z_im, y=label.to(dev).view(b, o))
        return (img, z, z_im) if return_latents else img
    finally:
        netG.train(was_training)
