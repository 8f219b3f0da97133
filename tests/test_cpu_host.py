"""CPU: host-side logic of the HIP package -- module/state_dict layout against the reference's keys,
flat parameter storage, arena tables, synthetic layouts, and that the product path refuses to run
without a GPU (no silent fallback)."""
import math

import numpy as np
import pytest
import torch

from tests.helpers import fixture_shapes, load_fixture


def _models():
    import layout2img_amd as L
    from layout2img_amd import generator as G
    return {"g_coco.npz": lambda: L.ResnetGenerator128_context(num_classes=184),
            "g_vg.npz": lambda: G.context_aware_generator(num_classes=179),
            "d_coco.npz": lambda: L.CombineDiscriminator128_app(num_classes=184)}


@pytest.mark.parametrize("fixture", ["g_coco.npz", "g_vg.npz", "d_coco.npz"])
def test_state_dict_layout_equals_reference(fixture):
    torch.manual_seed(0)
    m = _models()[fixture]()
    ref = fixture_shapes(load_fixture(fixture))
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == ref


def test_parameter_counts_match_survey():
    torch.manual_seed(0)
    ms = _models()
    assert sum(p.numel() for p in ms["g_coco.npz"]().parameters()) == 40853873
    assert sum(p.numel() for p in ms["d_coco.npz"]().parameters()) == 62696387


def test_flat_params_and_arena_tables_on_cpu():
    from layout2img_amd.arena import FlatParams, GemmWeight, WeightArena
    torch.manual_seed(0)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = GemmWeight("conv", 24, 3, 3, sn=True, eps=1e-4)
            self.b = GemmWeight("linear", 10, 308, sn=True, uses=2)
            self.c = GemmWeight("conv", 100, 16, 1, sn=False, bias=False)
            self.s = torch.nn.Parameter(torch.ones(5))
    net = Net()
    w0 = net.a.weight_orig.detach().clone()
    flat = FlatParams(net, "cpu")
    assert torch.equal(net.a.weight_orig.detach(), w0)
    assert net.a.weight_orig.data_ptr() == flat.data.data_ptr() + 4 * flat.offset_of(net.a.weight_orig)
    (net.s * 2).sum().backward()
    assert float(flat.grad.sum()) == 10.0          # autograd accumulated into the flat gradient view
    flat.zero_grad()
    # a parameter that is not a GemmWeight's ("loose") has its .grad detached while backward runs: autograd keeps the
    # incoming tensor (no add_ launch per parameter) and flush_loose() moves them all into the flat buffer at once
    assert float(flat.grad.abs().sum()) == 0.0 and net.s.grad is None
    assert net.a.weight_orig.grad.data_ptr() == flat.grad.data_ptr() + 4 * flat.offset_of(net.a.weight_orig)
    (net.s * 3).sum().backward()
    assert float(flat.grad.abs().sum()) == 0.0 and float(net.s.grad.sum()) == 15.0
    flat.flush_loose()
    assert float(flat.grad.sum()) == 15.0 and net.s.grad.data_ptr() == flat.grad.data_ptr() + 4 * flat.offset_of(net.s)
    flat.zero_grad()
    arena = WeightArena(net, flat, "cpu", torch.bfloat16)
    assert arena.n_layers == 4 and arena.rounds == 2          # b is applied twice -> two rows, two rounds
    tab = arena.layers.view(-1, 20).numpy()
    assert tab[0, 6] == 24 and tab[0, 7] == 8                  # Ci = 3 padded to 8
    assert tab[1, 7] == 312 and tab[1, 8] == 320               # 308 -> 312, Kpad multiple of 64
    assert tab[1, 0] == tab[2, 0] and tab[1, 1] == tab[2, 1] and tab[1, 16] != tab[2, 16] and tab[1, 10] != tab[2, 10]
    assert tab[3, 1] == -1                                     # no spectral norm
    assert net.b.use(1).fwd_off == tab[2, 10] and net.b.use(0) is net.b


def test_dw_accumulator_layout_and_written_bookkeeping_on_cpu():
    """The per-pass dW-bar accumulators are torch.empty (arena.PassCtx.dw): convolution slices are STORED by their first weight-gradient
    launch of the pass, everything that is accumulated into (Linear / Embedding layers, grouped projections) sits in ONE contiguous
    range at the front that is cleared per pass, slices never overlap, and the bookkeeping says when a launch must add instead
    (a second backward over the same forward) and which slices nothing reached."""
    from layout2img_amd.arena import FlatParams, GemmWeight, PassCtx, WeightArena, DualPass
    torch.manual_seed(0)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = GemmWeight("conv", 24, 16, 3, sn=True, eps=1e-4)
            self.l1 = GemmWeight("linear", 16, 312, sn=False)
            self.g1 = GemmWeight("linear", 8, 312, sn=True)
            self.c2 = GemmWeight("conv", 32, 24, 3, sn=True, uses=2)
            self.g2 = GemmWeight("linear", 16, 312, sn=True)
            self.e = GemmWeight("embedding", 10, 8, sn=True)
            self.g1.group = self.g2.group = "grp"
    net = Net()
    arena = WeightArena(net, FlatParams(net, "cpu"), "cpu", torch.bfloat16)
    (lo, hi), = arena.acc_ranges                                  # one range, at the front
    assert lo == 0 and hi < arena.dw_len
    uses = [u for h in (net.c1, net.c2) for u in h.use_rows]
    assert sorted(u.dw_off for u in arena.conv_uses) == sorted(u.dw_off for u in uses) and len(uses) == 3
    assert all(u.dw_off >= hi for u in uses)                      # convolutions behind the accumulated-into slices
    g = arena.groups["grp"]
    spans = [(u.dw_off, u.dw_off + u.co_p * u.kp) for u in uses] + [(net.l1.dw_off, net.l1.dw_off + net.l1.co_p * net.l1.kp),
                                                                    (net.e.dw_off, net.e.dw_off + net.e.co_p * net.e.kp), (g.dw_off, g.dw_off + g.n_total * g.kp)]
    spans.sort()
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] <= arena.dw_len   # no overlap
    assert net.g1.dw_off == g.dw_off and net.g2.dw_off == g.dw_off + net.g1.co_p * g.kp          # members: rows of the group's slice
    assert all(lo <= a and b <= hi for a, b in spans if a < hi)  # every non-conv slice inside the cleared range

    pc = PassCtx.__new__(PassCtx)                                 # (no device work: the bookkeeping alone)
    pc.arena, pc.dwbar = arena, None
    assert not pc.was_written(net.c1)                             # before the accumulator exists: first launch stores
    pc.dwbar = torch.full((arena.dw_len,), float("nan"))
    pc.dwbar[lo:hi] = 0
    pc.written = set()
    pc.mark_written(net.c1)
    assert pc.was_written(net.c1) and not pc.was_written(net.c2) and not pc.was_written(net.c2.use(1))
    s = pc.dw_acc(net.c2)                                         # a caller that ADDS: an unwritten conv slice starts from zero
    assert float(s.abs().sum()) == 0.0 and pc.was_written(net.c2)
    pc.clear_unwritten()                                          # what no launch reached (c2's second application) is zero for the spectral-norm backward
    assert bool(torch.isfinite(pc.dw_slice(net.c2.use(1))).all()) and float(pc.dw_slice(net.c2.use(1)).abs().sum()) == 0.0
    assert bool(torch.isnan(pc.dw_slice(net.c1)).all())           # (c1's was "stored" by its launch: left alone)
    a, b = PassCtx.__new__(PassCtx), PassCtx.__new__(PassCtx)
    for q in (a, b):
        q.arena, q.training, q.need_wgrad, q.dwbar, q.written = arena, True, True, torch.zeros(arena.dw_len), set()
    d = DualPass(a, b)
    d.mark_written(net.c1)
    assert a.was_written(net.c1) and b.was_written(net.c1) and d.was_written(net.c1) and not d.was_written(net.c2)


def test_no_silent_cpu_fallback():
    import layout2img_amd as L
    from layout2img_amd import ops
    torch.manual_seed(0)
    d = L.CombineDiscriminator128_app(num_classes=184)
    with pytest.raises((RuntimeError, AttributeError)):
        d(torch.zeros(1, 3, 128, 128), torch.zeros(1, 8, 4), torch.zeros(1, 8, 1, dtype=torch.long))
    with pytest.raises(RuntimeError):
        ops.cast_op(torch.zeros(8), torch.bfloat16)


def test_synthetic_layout_statistics():
    from layout2img_amd.synthetic import make_batch, make_layouts
    label, bbox = make_layouts(64, "coco", seed=1)
    n = (label != 0).sum(1)
    assert label.shape == (64, 8) and int(n.min()) >= 3 and int(n.max()) <= 8 and int(label.max()) <= 183
    real = bbox[label != 0]
    assert float((real[:, 2] * real[:, 3]).min()) > 0.02 and float((real[:, 0] + real[:, 2]).max()) <= 1.0 + 1e-6
    assert torch.allclose(bbox[label == 0], torch.tensor([-0.6, -0.6, 0.5, 0.5]))
    label, bbox = make_layouts(16, "vg", seed=2)
    assert label.shape == (16, 31)
    for b in range(16):
        k = int((label[b] != 0).sum())
        assert torch.allclose(bbox[b, k], torch.tensor([0.0, 0.0, 1.0, 1.0])) and int(label[b, k]) == 0
    real, label, bbox, z, z_im = make_batch(4, 128)
    assert real.shape == (4, 3, 128, 128) and z.shape == (4, 8, 128) and z_im.shape == (4, 128)
    assert float(real.min()) >= -1 and float(real.max()) <= 1


def test_truncated_normal_matches_the_rejection_sampler_distribution():
    """sampling.truncated_normal == N(0,1) restricted to [-t, t] (reference utils/util.py:39-45 draws it by rejection)"""
    from layout2img_amd.sampling import truncated_normal
    g = torch.Generator().manual_seed(0)
    for t in (1.0, 2.0):
        z = truncated_normal((200000,), t, "cpu", g)
        assert float(z.abs().max()) <= t
        # moments of the truncated normal: mean 0, var = 1 - 2 t phi(t) / (2 Phi(t) - 1)
        phi = math.exp(-t * t / 2) / math.sqrt(2 * math.pi)
        var = 1 - 2 * t * phi / math.erf(t / math.sqrt(2))
        assert abs(float(z.mean())) < 1e-2 and abs(float(z.var()) - var) < 1e-2
        ref = torch.randn(400000, generator=g)
        ref = ref[ref.abs() <= t][:200000]                 # the rejection sampler
        q = torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95])
        assert float((torch.quantile(z, q) - torch.quantile(ref, q)).abs().max()) < 2e-2


def test_reference_checkpoint_loading_strips_the_dataparallel_prefix():
    """load_reference_checkpoint == test_context_app_v2.py:44-59 (keys `module.<name>`, unknown / mis-shaped keys ignored)"""
    import layout2img_amd as L
    torch.manual_seed(0)
    src = L.ResnetGenerator64_context(num_classes=10)
    dst = L.ResnetGenerator64_context(num_classes=10)
    ckpt = {"module." + k: v.clone() + 1.0 for k, v in src.state_dict().items() if v.dtype.is_floating_point}
    ckpt["module.not_a_key"] = torch.zeros(3)
    loaded, ignored = L.load_reference_checkpoint(dst, ckpt)
    assert ignored == ["module.not_a_key"] and len(loaded) == len(ckpt) - 1
    sd, ss = dst.state_dict(), src.state_dict()
    for k in loaded:
        assert torch.equal(sd[k], ss[k] + 1.0), k
