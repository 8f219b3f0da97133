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


def test_power_iteration_partial_row_tables_on_cpu():
    """Round 6: the power iteration sums W^T u per block of 64 rows into stored partial rows that `sn_tfold_kernel` adds in order, and ||W v||^2 per
    block of 4 R rows into shares (csrc/weights.hip). The tables that place those rows in the caller's scratch are host logic (arena.py): here
    the kernels' address arithmetic is replayed in numpy from the tables alone -- every partial row is written exactly once, nothing overlaps, the
    fold of the rows is W^T u, the shares are disjoint -- and the <G, W> ranges of the backward name exactly a layer's dot-table entries."""
    import numpy as np
    from layout2img_amd import arena as A
    from layout2img_amd.arena import FlatParams, GemmWeight, WeightArena
    torch.manual_seed(0)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = GemmWeight("conv", 200, 24, 3, sn=True, eps=1e-4)        # 4 row blocks of 64, Kt = 216
            self.b = GemmWeight("linear", 1100, 128, sn=True)                # 18 row blocks: the narrow fold workgroups (cw = 64)
            self.c = GemmWeight("conv", 100, 16, 1, sn=False, bias=False)    # no spectral norm: no entries
            self.d = GemmWeight("conv", 72, 40, 3, sn=True, uses=2)          # applied twice: a second round with its own rows
            self.e = GemmWeight("linear", 4500, 16, sn=True)                 # 71 row blocks: cw = 16
    net = Net()
    flat = FlatParams(net, "cpu")
    ar = WeightArena(net, flat, "cpu", torch.bfloat16)
    tab = ar.layers.view(-1, 20).numpy()
    rows_per_block, wv_rows = 4 * A.WTU_RPW, 4 * A.WV_R
    for r in range(ar.rounds):
        wtu = ar.t_wtu[r][0].view(-1, 5).numpy()[:ar.t_wtu[r][1]]
        tf = ar.t_tfold[r][0].view(-1, 5).numpy()[:ar.t_tfold[r][1]]
        wv = ar.t_wv[r][0].view(-1, 2).numpy()[:ar.t_wv[r][1]]
        scratch = np.full(ar.sn_scratch_floats, np.nan)
        tpart = scratch[ar.np_len[r]:]
        layers = sorted(set(int(e[0]) for e in wtu))
        assert layers == sorted(set(int(e[0]) for e in tf)) == sorted(set(int(e[0]) for e in wv))
        W, U = {}, {}
        for li in layers:
            co, kt = int(tab[li, 3]), int(tab[li, 4] * tab[li, 5] * tab[li, 5])
            W[li], U[li] = np.random.RandomState(li).randn(co, kt), np.random.RandomState(100 + li).randn(co)
        for li, c0, r0, rpw, off in wtu:      # sn_wtu_kernel: 256 columns x (4 x rpw) rows -> columns [c0, c0 + 256) of one partial row
            co, kt = W[li].shape
            cols = slice(c0, min(c0 + 256, kt))
            assert np.all(np.isnan(tpart[off + cols.start:off + cols.stop])), "a partial row written twice"
            tpart[off + cols.start:off + cols.stop] = (U[li][r0:r0 + 4 * rpw, None] * W[li][r0:r0 + 4 * rpw, cols]).sum(0)
        seen = {li: np.zeros(W[li].shape[1], dtype=int) for li in layers}
        for li, c0, off, nrb, cw in tf:       # sn_tfold_kernel: cw columns, all row blocks, in order
            co, kt = W[li].shape
            assert nrb == -(-co // rows_per_block) and cw == (256 if nrb <= 8 else 64 if nrb <= 32 else 16) and 256 % cw == 0
            cols = slice(c0, min(c0 + cw, kt))
            t = sum(tpart[off + rb * kt + cols.start:off + rb * kt + cols.stop] for rb in range(nrb))
            assert np.allclose(t, (U[li][:, None] * W[li][:, cols]).sum(0)), (li, c0)
            seen[li][cols] += 1
        assert all(np.all(v == 1) for v in seen.values())          # every column of t folded exactly once
        share = scratch[:ar.np_len[r]]
        for li, r0 in wv:                      # sn_wv_kernel: one share per block of 4 R rows at the layer's row-19 offset
            idx = int(tab[li, 19]) + r0 // wv_rows
            assert np.isnan(share[idx]), "two sn_wv blocks share a slot"
            share[idx] = 1.0
        for li in layers:
            nb = -(-int(tab[li, 3]) // wv_rows)
            assert np.all(share[int(tab[li, 19]):int(tab[li, 19]) + nb] == 1.0)
    # the backward's <G, W>: (first entry, count) per layer row names exactly that row's dot-table entries
    dot = ar.t_dot.view(-1, 2).numpy()[:ar.n_dot]
    rng = ar.t_dot_range.view(-1, 2).numpy()
    for li in range(ar.n_layers):
        mine = np.nonzero(dot[:, 0] == li)[0]
        assert rng[li, 1] == len(mine) and (len(mine) == 0 or (rng[li, 0] == mine[0] and np.all(np.diff(mine) == 1)))
    assert rng[[i for i in range(ar.n_layers) if tab[i, 1] < 0], 1].sum() == 0      # layers without spectral norm: no entries


def test_bias_slots_and_pass_bookkeeping_on_cpu(monkeypatch):
    """Round 6 / ADVICE r05: (i) a convolution applied several times per forward gets a per-pass bias-gradient slot like the padded ones (its 2 uses x
    2 passes otherwise meet in stream order in the shared gradient); (ii) flush_grads adds the slots of EACH pass with its own multi-tensor add --
    one add over all passes names a destination twice, and the GPU kernel then keeps one addend or both (the 2.1e-3 flicker of round 5);
    (iii) a pass that holds gradients is never evicted from the pending list."""
    from layout2img_amd import arena as A
    from layout2img_amd.arena import FlatParams, GemmWeight, PassCtx, WeightArena
    torch.manual_seed(0)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.pad = GemmWeight("conv", 100, 16, 3, sn=False)            # 100 of 104 channels
            self.twice = GemmWeight("conv", 64, 16, 3, sn=True, uses=2)    # applied twice per forward
            self.plain = GemmWeight("conv", 64, 16, 3, sn=True)
    net = Net()
    flat = FlatParams(net, "cpu")
    ar = WeightArena(net, flat, "cpu", torch.bfloat16)
    assert getattr(net.pad, "bias_scr_off", None) is not None and getattr(net.twice, "bias_scr_off", None) is not None
    assert getattr(net.plain, "bias_scr_off", None) is None
    (lo, hi), = ar.acc_ranges
    for h in (net.pad, net.twice):
        assert lo <= h.bias_scr_off and h.bias_scr_off + h.co_p <= hi      # cleared with the accumulated-into slices of every pass
    flat.zero_grad()

    def fake_pass(value):
        p = PassCtx.__new__(PassCtx)
        p.arena, p.training, p.need_wgrad, p.written = ar, True, True, None
        p.dwbar = torch.zeros(ar.dw_len)
        p.norms, p.pass_uv = torch.zeros(4 * ar.n_layers), torch.zeros(ar.uv_len)
        p.bias_fix = {}
        for h in (net.pad, net.twice):
            p.bias_slot(h, h.bias.grad).fill_(value)
        return p
    ar.pending = [fake_pass(1.0), fake_pass(2.0)]
    calls = []
    real_add = torch._foreach_add_

    def recording(dsts, srcs):
        calls.append([d.data_ptr() for d in dsts])
        return real_add(dsts, srcs)
    monkeypatch.setattr(torch, "_foreach_add_", recording)
    monkeypatch.setattr(A._lib, "call", lambda *a, **k: None)              # (the spectral-norm backward launches: not on the CPU)
    monkeypatch.setattr(A._lib, "workspace", lambda dev: 0)
    monkeypatch.setattr(A._lib, "raw_stream", lambda: 0)
    from layout2img_amd import ops
    monkeypatch.setattr(ops.WgradSide, "join", classmethod(lambda cls: None))   # (side-stream bookkeeping: queries the current CUDA stream)
    ar.flush_grads()
    assert len(calls) == 2 and all(len(set(c)) == len(c) for c in calls)   # one add per pass, no destination twice in a call
    assert torch.all(net.pad.bias.grad == 3.0) and net.pad.bias.grad.numel() == 100 and torch.all(net.twice.bias.grad == 3.0)
    assert ar.pending == []
    # eviction: nine forwards without an optimiser step -- only passes WITHOUT gradients leave the list
    keep = fake_pass(5.0)
    ar.pending = [keep] + [PassCtx.__new__(PassCtx) for _ in range(8)]
    for q in ar.pending[1:]:
        q.dwbar, q.bias_fix = None, {}
    newest = PassCtx.__new__(PassCtx)
    newest.dwbar, newest.bias_fix = None, {}
    ar.pending.append(newest)
    if len(ar.pending) > 8:   # (WeightArena.prepare's rule)
        for k, old in enumerate(ar.pending[:-1]):
            if old.dwbar is None and not old.bias_fix:
                del ar.pending[k]
                break
    assert keep in ar.pending and newest in ar.pending and len(ar.pending) == 9
