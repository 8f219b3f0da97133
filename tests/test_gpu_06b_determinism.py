"""Run-to-run determinism (round 6). The reference's single-device arithmetic is order-deterministic on the CPU: F.batch_norm's batch
statistics (model/sync_batchnorm/batchnorm.py:51-53), torch.nn.utils.spectral_norm's power iteration, its
convolutions. Rounds 1-5 summed those quantities with float atomics -- batch statistics into a replicated workspace, W^T u and
||W v||^2 per block -- so two runs of the same launch sequence differed in the last bits, a pre-activation at ~0 landed on the other side
of its ReLU gate (or a value on a bf16 rounding boundary rounded the other way) and the flat gradient moved by ~1e-3 (f32) / ~1e-1 (bf16)
between two IDENTICAL runs. Now every wave / slab / row block STORES its partial sums and a fold kernel adds them in a fixed order
(csrc/common.h rows_fold, csrc/weights.hip sn_tfold_kernel / layer_sn2): the tests below require BIT-IDENTICAL results."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pack(w, kpad_mult):
    co, ci, kh, _ = w.shape
    k = kh * kh * ci
    kpad, npad = (k + kpad_mult - 1) // kpad_mult * kpad_mult, (co + 127) // 128 * 128
    p = torch.zeros(npad, kpad)
    p[:co, :k] = w.permute(0, 2, 3, 1).reshape(co, k)
    return p, kpad


STAT_CASES = [
    # B, H, W, Ci, Co, KH, up2, pool2, dtype     (row counts of the partial matrix on both sides of the one-launch / two-launch fold)
    (32, 64, 64, 128, 64, 3, False, False, torch.bfloat16),    # 128x64 halo tiles: 1024 x 2 rows -> chunked fold
    (32, 64, 64, 64, 64, 3, True, False, torch.bfloat16),      # 128^2 result: 8192 rows
    (32, 32, 32, 512, 256, 3, False, False, torch.bfloat16),   # 128x128 tiles
    (32, 16, 16, 512, 512, 3, False, False, torch.bfloat16),   # K-split by stored partials: the reduce kernel carries the statistics
    (8, 32, 32, 64, 104, 3, False, False, torch.bfloat16),     # padded channel count (mask heads)
    (4, 16, 16, 32, 136, 3, False, True, torch.float32),       # generic kernel, f32, pooled result
    (16, 8, 8, 1024, 1024, 3, False, False, torch.float32),    # f32 on a small grid: no atomic K split any more
    (6, 32, 32, 40, 104, 1, False, False, torch.float32),      # 1x1
]


@pytest.mark.parametrize("case", STAT_CASES)
def test_conv_epilogue_statistics_are_bit_identical_and_right(case):
    """fused statistics of a convolution's f32 result: three launches give the same bits (result AND statistics), and the sums are the
    sums of the result (f64 on the host)."""
    from layout2img_amd import ops
    B, H, W, Ci, Co, KH, up2, pool2, dt = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, H, W, Ci, generator=g).to(dt)
    w = (torch.randn(Co, Ci, KH, KH, generator=g) / math.sqrt(Ci * KH * KH)).to(dt).float()
    pack, kpad = _pack(w, 64 if dt == torch.bfloat16 else 32)
    xd, pd = x.to(DEV), pack.to(DEV, dt)
    runs = []
    for _ in range(3):
        out, _, _ = ops.conv_raw(xd, pd, kpad, Co, KH, up2=up2, pool2=pool2, alpha=0.25 if pool2 else 1.0, stats=True)
        s1, s2, _ = out._l2i_stats
        torch.cuda.synchronize()
        runs.append((out.clone(), s1.clone(), s2.clone()))
    for o, a, b in runs[1:]:
        assert torch.equal(o, runs[0][0]) and torch.equal(a, runs[0][1]) and torch.equal(b, runs[0][2])
    o64 = runs[0][0].double().view(-1, Co)
    ref1, ref2 = o64.sum(0), (o64 * o64).sum(0)
    n = o64.shape[0]
    tol1 = 1e-6 * float(o64.abs().sum(0).max()) + 1e-6
    assert float((runs[0][1].double().view(-1) - ref1).abs().max()) < 4 * tol1 * math.sqrt(n) / 50 + 1e-3 * float(ref1.abs().max()) * 1e-3 + 1e-4
    assert float(((runs[0][2].double().view(-1) - ref2) / ref2.clamp_min(1e-12)).abs().max()) < 1e-5


@pytest.mark.parametrize("rows,C,rpg,want_sq", [(32 * 64 * 64, 128, None, True), (32 * 4096, 64, 4096, True), (256, 19712, None, False),
                                                (8 * 16 * 16, 1024, None, True), (100, 8, None, True), (32 * 128 * 128, 64, None, True)])
def test_channel_stats_are_bit_identical_and_right(rows, C, rpg, want_sq):
    from layout2img_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(rows, C, generator=g).to(DEV)
    runs = []
    for _ in range(3):
        s, q = ops.channel_stats(x, rows_per_group=rpg, want_sq=want_sq)
        torch.cuda.synchronize()
        runs.append((s.clone(), None if q is None else q.clone()))
    for s, q in runs[1:]:
        assert torch.equal(s, runs[0][0]) and (q is None or torch.equal(q, runs[0][1]))
    G = 1 if rpg is None else rows // rpg
    x64 = x.double().view(G, -1, C)
    assert float((runs[0][0].double() - x64.sum(1)).abs().max()) < 1e-5 * float(x64.abs().sum(1).max())
    if want_sq:
        assert float(((runs[0][1].double() - (x64 * x64).sum(1)) / (x64 * x64).sum(1)).abs().max()) < 1e-5


def _nets(dt, size=128):
    import layout2img_amd as L
    torch.manual_seed(0)
    g = (L.ResnetGenerator128_context if size == 128 else L.ResnetGenerator64_context)(num_classes=184).finalize(DEV, dt).train()
    d = (L.CombineDiscriminator128_app if size == 128 else L.CombineDiscriminator64)(num_classes=184).finalize(DEV, dt).train()
    for m in g.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
    return g, d


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_power_iteration_and_packs_are_bit_identical(dt):
    """one train-mode power iteration + packing of every weight of both networks, twice from the same u / v: same u, v, sigma, same packs"""
    for net in _nets(dt):
        a = net.arena
        sn0 = a.sn_flat.data.clone()
        outs = []
        for _ in range(2):
            a.sn_flat.data.copy_(sn0)
            p = a.prepare(training=True, need_wgrad=False)
            torch.cuda.synchronize()
            outs.append((a.sn_flat.data.clone(), p.norms.clone(), p.packed.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
        assert bool(torch.isfinite(outs[0][1]).all()) and float(outs[0][1].view(-1, 4)[:, 2].min()) > 0   # (every sigma positive)


@pytest.mark.parametrize("dt,batch", [(torch.float32, 4), (torch.bfloat16, 4), (torch.bfloat16, 32)])
def test_train_mode_forwards_are_bit_identical(dt, batch):
    """the generator's and the discriminator's train-mode forward (batch statistics, power iteration) twice from the same state -- at the
    headline batch too, whose launches pick other tiles / K splits than a small batch's"""
    from layout2img_amd.synthetic import make_batch
    g, d = _nets(dt)
    real, label, bbox, z, z_im = make_batch(batch, 128, "coco", seed=3, device=torch.device(DEV))
    gsn, dsn = g.arena.sn_flat.data.clone(), d.arena.sn_flat.data.clone()
    gst = {k: v.clone() for k, v in g.state_dict().items()}
    outs = []
    with torch.no_grad():
        for _ in range(2):
            g.load_state_dict(gst)
            g.arena.sn_flat.data.copy_(gsn), d.arena.sn_flat.data.copy_(dsn)
            img = g(z, bbox, z_im, label)
            o = d(img, bbox, label.unsqueeze(-1))
            torch.cuda.synchronize()
            outs.append([img.clone()] + [t.clone() for t in o])
    for a, b in zip(*outs):
        assert torch.equal(a, b), float((a - b).abs().max())


_RUNS = {}


def _trainer_run_cached(dt, mode):
    if (dt, mode) not in _RUNS:   # (the eager run is one side of both tests below)
        _RUNS[(dt, mode)] = _trainer_run(dt, mode)
    return _RUNS[(dt, mode)]


def _trainer_run(dt, mode, iters=3, batch=8):
    """`iters` training iterations from a fixed state: eager (D(real) on the side stream, as layout2img_amd.train runs them when it cannot replay)
    or as replays of the captured iteration. Returns everything an iteration changes."""
    import layout2img_amd as L
    from layout2img_amd.synthetic import make_batch
    from layout2img_amd.trainer import restore_state, snapshot_state
    g, d = _nets(dt)
    tr = L.GanTrainer(g, d)
    real, label, bbox, z, z_im = make_batch(batch, 128, "coco", seed=3, device=torch.device(DEV))
    if mode == "graph":
        st = snapshot_state(tr)
        assert tr.capture(real, label, bbox, z, z_im)
        restore_state(tr, st)
        for _ in range(iters):
            r = tr.step_graphed(real, label, bbox, z, z_im)
    else:
        for _ in range(iters):
            r = tr.step(real, label, bbox, z, z_im)
    tr.flush()
    torch.cuda.synchronize()
    return [g.flat.data.clone(), d.flat.data.clone(), g.arena.sn_flat.data.clone(), d.arena.sn_flat.data.clone(),
            tr.g_opt.m.clone(), tr.g_opt.v.clone(), tr.d_opt.m.clone(), tr.d_opt.v.clone(), g.flat.grad.clone(), d.flat.grad.clone(),
            r["d_loss"].detach().clone().view(1), r["g_loss"].detach().clone().view(1), r["fake"].detach().clone()]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_training_iterations_are_bit_identical_from_run_to_run(dt):
    """Three whole iterations (D step on two streams, G step, both Adam steps) twice from the same state: parameters, power-iteration vectors,
    Adam moments and the last gradients are the SAME BITS -- as two runs of the reference's loop on one CPU are (train_context_app_v2.py:148-189).
    Rounds 1-5: two f32 runs were 2.8 % apart in parameter distance after ONE Adam step and decorrelated by iteration ~100."""
    a, b = _trainer_run_cached(dt, "eager"), _trainer_run(dt, "eager")
    for k, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), (k, float((x - y).abs().max()))
    assert bool(torch.isfinite(a[0]).all()) and float((a[8] != 0).float().mean()) > 0.5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_graph_replays_are_bit_identical_to_eager_iterations(dt):
    """What BENCH times -- replays of the captured iteration -- leaves the same bits as the eager iterations: same launches, same order,
    no float atomics anywhere on the path (the graph-vs-eager bars of rounds 2-5 were 2.5e-4 ... 8.7e-3 of the gradient)."""
    a, b = _trainer_run(dt, "graph"), _trainer_run_cached(dt, "eager")
    for k, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), (k, float((x - y).abs().max()), float((x - y).norm() / y.norm().clamp_min(1e-30)))


def test_vg_training_iteration_is_bit_identical_from_run_to_run():
    """BASELINE config 5's models (context_aware_generator, 31 object slots: the > 8-object ISLA backward, whose projection gradients are stored
    rows + an ordered finish launch since round 6): two iterations twice from the same state leave the same bits."""
    import layout2img_amd as L
    from layout2img_amd.synthetic import make_batch
    outs = []
    for _ in range(2):
        torch.manual_seed(0)
        g = L.context_aware_generator(num_classes=179).finalize(DEV, torch.bfloat16).train()
        d = L.CombineDiscriminator128_app(num_classes=179).finalize(DEV, torch.bfloat16).train()
        tr = L.GanTrainer(g, d)
        real, label, bbox, z, z_im = make_batch(2, 128, "vg", seed=3, device=torch.device(DEV))
        for _ in range(2):
            r = tr.step(real, label, bbox, z, z_im)
        tr.flush()
        torch.cuda.synchronize()
        outs.append([g.flat.data.clone(), d.flat.data.clone(), g.flat.grad.clone(), d.flat.grad.clone(), r["g_loss"].detach().clone().view(1)])
    for k, (x, y) in enumerate(zip(*outs)):
        assert torch.equal(x, y), (k, float((x - y).abs().max()))
    assert float((outs[0][2] != 0).float().mean()) > 0.5
