"""SURVEY section 8 rows f1 / f2 on the GPU: checkpoint save -> resume (reference layout, train_context_app_v2.py:71-105,
215-217) incl. the optimizer state, and the sampling path (test_context_app_v2.py:36-83) at batch 1 against the oracle."""
import numpy as np
import pytest
import torch

from oracle import model as O
from tests.helpers import maxdiff

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _nets(seed, dt=torch.float32, size=64):
    import layout2img_amd as L
    torch.manual_seed(seed)
    if size == 64:
        g, d = L.ResnetGenerator64_context(num_classes=184), L.CombineDiscriminator64(num_classes=184)
    else:
        g, d = L.ResnetGenerator128_context(num_classes=184), L.CombineDiscriminator128_app(num_classes=184)
    for m in g.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
    return g.finalize(DEV, dt), d.finalize(DEV, dt)


def test_save_resume_round_trip(tmp_path):
    """train 2 steps -> save (G, D in the reference's `module.`-prefixed layout + Adam state) -> load into freshly
    constructed, differently initialised networks -> identical eval output and an identical third step."""
    import layout2img_amd as L
    from layout2img_amd.synthetic import make_batch
    g, d = _nets(1)
    tr = L.GanTrainer(g, d)
    tr.overlap = False
    batch = make_batch(4, 64, "coco", seed=3, device=DEV)
    for _ in range(2):
        tr.step(*batch)
    paths = L.save_checkpoint(str(tmp_path), 7, g, d, tr.g_opt, tr.d_opt)
    sd = torch.load(paths["G"])
    assert all(k.startswith("module.") for k in sd)                              # what nn.DataParallel leaves (:215-217)
    assert len({v.untyped_storage().data_ptr() for v in sd.values()}) == len(sd)  # own storages, not views of the flat buffer
    assert sd["module.fc.weight_orig"].shape == (16384, 128) and "module.res2.conv1.weight_u" in sd

    g2, d2 = _nets(99)
    assert maxdiff(g2.flat.data, g.flat.data) > 1e-3
    tr2 = L.GanTrainer(g2, d2)
    tr2.overlap = False
    assert L.load_checkpoint(str(tmp_path), 7, g2, d2, tr2.g_opt, tr2.d_opt) == 7
    assert torch.equal(g2.flat.data, g.flat.data) and torch.equal(d2.flat.data, d.flat.data)
    assert torch.equal(g2.arena.sn_flat.data, g.arena.sn_flat.data)
    assert tr2.g_opt.t == tr.g_opt.t == 2 and int(tr2.d_opt.t_dev) == 2
    # resume keeps the command line's learning rate unless asked to restore the file's; files of earlier revisions (flat
    # moments m / v / t) still load; anything else is a clear error
    from layout2img_amd.sampling import _load_opt_state, _opt_state
    tr3 = L.GanTrainer(g2, d2, g_lr=3e-4)
    st = _opt_state(tr.g_opt)
    _load_opt_state(tr3.g_opt, st)
    assert tr3.g_opt.lr == 3e-4 and torch.equal(tr3.g_opt.m, tr.g_opt.m)
    _load_opt_state(tr3.g_opt, st, restore_hyper=True)
    assert tr3.g_opt.lr == tr.g_opt.lr
    tr3.g_opt.m.zero_()
    _load_opt_state(tr3.g_opt, dict(m=tr.g_opt.m.cpu(), v=tr.g_opt.v.cpu(), t=2))
    assert torch.equal(tr3.g_opt.m, tr.g_opt.m) and tr3.g_opt.t == 2
    with pytest.raises(RuntimeError, match="legacy flat format"):
        _load_opt_state(tr3.g_opt, dict(m=torch.zeros(8), v=torch.zeros(8), t=1))
    with pytest.raises(RuntimeError, match="neither"):
        _load_opt_state(tr3.g_opt, dict(t=1))
    real, label, bbox, z, z_im = batch
    g.eval(), g2.eval()
    with torch.no_grad():
        a, b = g(z, bbox, z_im, label), g2(z, bbox, z_im, label)
    assert maxdiff(a, b) < 5e-5   # (split-K partial sums are combined by atomics: not order-deterministic)
    g.train(), g2.train()
    ra, rb = tr.step(*batch), tr2.step(*batch)
    assert abs(float(ra["d_loss"]) - float(rb["d_loss"])) < 1e-4 * abs(float(ra["d_loss"])) + 1e-5
    # (atomically reduced sums are not order-deterministic: identical up to f32 round-off amplified by one Adam step)
    close = ((g.flat.data - g2.flat.data).abs() < 2.5e-4).float().mean()
    assert float(close) > 0.98, float(close)


def test_reference_checkpoint_loads_with_and_without_prefix(tmp_path):
    import layout2img_amd as L
    g, _ = _nets(2)
    sd = L.reference_state_dict(g, prefix="module.")
    sd["module.not_in_this_model"] = torch.zeros(3)
    torch.save(sd, tmp_path / "G_200.pth")
    g2, _ = _nets(5)
    loaded, ignored = L.load_reference_checkpoint(g2, str(tmp_path / "G_200.pth"))
    assert ignored == ["module.not_in_this_model"] and len(loaded) == len(g.state_dict())
    assert torch.equal(g2.flat.data, g.flat.data)
    g3, _ = _nets(6)
    L.load_reference_checkpoint(g3, L.reference_state_dict(g, prefix=""))
    assert torch.equal(g3.flat.data, g.flat.data)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_sample_batch1_matches_oracle(dt):
    """test_context_app_v2.py:68-77: eval mode, batch 1, latents truncated at 2 -- the HIP path against the oracle's eval
    forward on the very latents `sample` drew."""
    import layout2img_amd as L
    from layout2img_amd.synthetic import make_layouts
    g, _ = _nets(3, dt, size=128)
    real, label, bbox, z, z_im = __import__("layout2img_amd.synthetic", fromlist=["make_batch"]).make_batch(4, 128, "coco", seed=9, device=DEV)
    g.train()
    with torch.no_grad():
        for _ in range(3):   # fill BN running statistics / advance the power iteration (a cold eval saturates tanh)
            g(z, bbox, z_im, label)
    lab1, box1 = make_layouts(1, "coco", seed=21, device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(4)
    img, zs, zi = L.sample(g, lab1, box1, thres=2.0, generator=gen, return_latents=True)
    assert img.shape == (1, 3, 128, 128) and float(zs.abs().max()) <= 2.0 and float(zi.abs().max()) <= 2.0 and g.training
    sd = {k: v.detach().cpu().clone() for k, v in g.state_dict().items()}
    ref = O.generator_forward(sd, zs.cpu(), box1.cpu(), zi.cpu(), lab1.cpu(), training=False)
    assert maxdiff(img, ref) < (1e-3 if dt == torch.float32 else 1e-1)


def test_training_entry_runs_and_resumes(tmp_path):
    """python -m layout2img_amd.train on synthetic layouts: 2 epochs, checkpoint, resume for a third."""
    from layout2img_amd import train
    common = ["--dataset", "coco", "--batch_size", "4", "--out_path", str(tmp_path), "--synthetic", "2", "--img_size", "64", "--dtype", "f32"]
    tr = train.main(common + ["--total_epoch", "2"])
    assert tr.g_opt.t == 4 and (tmp_path / "coco" / "64" / "model" / "G_2.pth").exists()
    tr2 = train.main(common + ["--total_epoch", "3", "--checkpoint_epoch", "2"])
    assert tr2.g_opt.t == 6 and (tmp_path / "coco" / "64" / "model" / "D_3.pth").exists()


def test_multi_iteration_graph_equals_single_replays():
    """GanTrainer.capture_multi: ONE replay of a graph of three consecutive iterations (each with its own static batch) leaves the
    networks, the Adam moments and step counts where three replays of the one-iteration graph leave them -- what bench.py's timed
    steps and the training entry's batch groups run. (Round 6: exactly -- the same launches in the same order, no float atomics.)"""
    import layout2img_amd as L
    from layout2img_amd.synthetic import make_batch
    from layout2img_amd.trainer import restore_state, snapshot_state
    g, d = _nets(5)
    tr = L.GanTrainer(g, d)
    batches = [make_batch(4, 64, "coco", seed=11 + i, device=DEV) for i in range(3)]
    st = snapshot_state(tr)
    assert tr.capture(*batches[0])
    restore_state(tr, st)
    for b in batches:
        r1 = tr.step_graphed(*b)
    torch.cuda.synchronize()
    p1 = (g.flat.data.clone(), d.flat.data.clone(), tr.g_opt.m.clone(), tr.d_opt.v.clone(), int(tr.g_opt.t_dev), tr.g_opt.t, float(r1["d_loss"]))
    restore_state(tr, st)
    assert tr.capture_multi(batches)
    restore_state(tr, st)
    rs = tr.step_graphed_multi(batches)
    torch.cuda.synchronize()
    assert len(rs) == 3 and int(tr.g_opt.t_dev) == p1[4] == int(st["opt"][0][3]) + 3 and tr.g_opt.t == p1[5]
    # round 6: no float atomics on the path -- the same launches in the same order leave the SAME BITS (rounds 2-5: 98 % of the parameters
    # within 2.5e-4, Adam moments within 5e-2: the order of the atomically reduced sums changed from run to run)
    assert torch.equal(g.flat.data, p1[0]) and torch.equal(d.flat.data, p1[1])
    moved = float((g.flat.data - st["flat"][0]).abs().mean())
    assert moved > 1e-4   # (three Adam steps of lr 1e-4 did move the parameters: the comparison above is not vacuous)
    assert torch.equal(tr.g_opt.m, p1[2]) and torch.equal(tr.d_opt.v, p1[3])
    assert float(rs[-1]["d_loss"]) == p1[6]
    with pytest.raises(RuntimeError, match="holds 3 iterations"):
        tr.step_graphed_multi(batches[:2])


def test_eval_pass_cache_and_graph_sampler_follow_parameter_updates():
    """The eval-mode weight packs are cached by the arena (arena.WeightArena._eval_pass) and the sampling call is a replayed HIP
    graph over that cache (sampling.GraphSampler; reference test_context_app_v2.py:68-77): (i) a second eval forward launches no
    weight preparation; (ii) the graph's image equals the eager eval forward on the latents the graph drew, and two calls draw
    different latents; (iii) a parameter changed by torch (an in-place op), by the Adam KERNEL (raw pointers: a training step) or by
    load_state_dict is picked up by the next call -- without a new capture."""
    import layout2img_amd as L
    from layout2img_amd import _lib, arena as A
    from layout2img_amd.sampling import GraphSampler
    from layout2img_amd.synthetic import make_batch, make_layouts
    g, d = _nets(3, torch.float32, size=128)
    real, label, bbox, z, z_im = make_batch(4, 128, "coco", seed=9, device=DEV)
    g.train()
    with torch.no_grad():
        for _ in range(3):
            g(z, bbox, z_im, label)
    lab1, box1 = make_layouts(1, "coco", seed=21, device=DEV)
    calls = []
    orig = _lib.call

    def spy(name, *a):
        calls.append(name)
        return orig(name, *a)
    A._lib.call = spy
    try:
        g.eval()
        with torch.no_grad():
            g(z[:1], box1, z_im[:1], lab1)
            n1 = calls.count("l2i_weights_prepare")
            g(z[:1], box1, z_im[:1], lab1)
            assert n1 >= 1 and calls.count("l2i_weights_prepare") == n1   # (ii-th forward: the cached pass)
    finally:
        A._lib.call = orig
    s = GraphSampler(g, thres=2.0)

    def check(tag):
        img, zs, zi = s(lab1, box1, return_latents=True)
        img, zs, zi = img.clone(), zs.clone(), zi.clone()
        with torch.no_grad():
            ref = g(zs, box1, z_im=zi, y=lab1)
        assert float(zs.abs().max()) <= 2.0 and float((img - ref).abs().max()) < 1e-5, (tag, float((img - ref).abs().max()))
        return img, zs
    g.eval()
    img_a, z_a = check("first")
    img_b, z_b = check("second")
    assert len(s._graphs) == 1 and not torch.equal(z_a, z_b)
    with torch.no_grad():
        g.fc.weight_orig.mul_(1.25)          # torch-side write
    check("after an in-place parameter write")
    g.train()
    tr = L.GanTrainer(g, d)
    tr.step(real, label, bbox, z, z_im)     # the Adam kernel writes the flat buffer through raw pointers
    g.eval()
    check("after a training step")
    sd = {k: v.detach().cpu().clone() * (0.5 if k == "res5.conv2.weight_orig" else 1.0) for k, v in g.state_dict().items()}
    g.load_state_dict(sd)
    check("after load_state_dict")
    assert len(s._graphs) == 1   # never re-captured


def test_three_graph_replays_equal_three_eager_iterations():
    """The whole-iteration graph against the eager path over SEVERAL consecutive replays (round 5: replays after the first used to accumulate
    split-K results onto uncleared buffers -- `hipMemsetAsync` nodes in front of atomics -- which a one-replay comparison cannot see): from one
    snapshot, three replays and three eager iterations on the same batches and latents leave the parameters, the Adam moments and the last
    losses EQUAL (round 6: bit for bit)."""
    import layout2img_amd as L
    from layout2img_amd.synthetic import make_batch
    from layout2img_amd.trainer import restore_state, snapshot_state
    g, d = _nets(7)
    tr = L.GanTrainer(g, d)
    batches = [make_batch(4, 64, "coco", seed=31 + i, device=DEV) for i in range(3)]
    st = snapshot_state(tr)
    assert tr.capture(*batches[0])
    restore_state(tr, st)
    for b in batches:
        r_g = tr.step_graphed(*b)
    torch.cuda.synchronize()
    graph = (g.flat.data.clone(), d.flat.data.clone(), tr.g_opt.m.clone(), tr.d_opt.v.clone(), float(r_g["d_loss"]), float(r_g["g_loss"]))
    restore_state(tr, st)
    for b in batches:
        r_e = tr.step(*b)
    tr.flush()
    torch.cuda.synchronize()
    # round 6: bit-identical (no float atomics left on the path; tests/test_gpu_06b_determinism.py has the 128x128 pair in both operand modes)
    assert torch.equal(g.flat.data, graph[0]) and torch.equal(d.flat.data, graph[1])
    assert torch.equal(tr.g_opt.m, graph[2]) and torch.equal(tr.d_opt.v, graph[3])
    assert float(r_e["d_loss"]) == graph[4] and float(r_e["g_loss"]) == graph[5]
