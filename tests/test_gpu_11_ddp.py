"""GPU: the full data-parallel training iteration (GanTrainer under torch.distributed) with world_size 2 equals the
single-process iteration on the concatenated batch -- DataParallel semantics of the reference
(train_context_app_v2.py:50-57,108-110; model/sync_batchnorm/batchnorm.py:59-125).

Only one GPU is available to the tests, and RCCL refuses two ranks on one device, so the two ranks share cuda:0 and
use the gloo backend (which all-reduces CUDA tensors through the host). The code path is the one `bench.py --gpus N`
runs over RCCL: SyncBN statistic exchange in forward and backward, global-count loss normalisation, flat gradient
all-reduce, parameter broadcast.
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    import layout2img_amd as L
    torch.manual_seed(7)
    g = L.ResnetGenerator64_context(num_classes=184)
    d = L.CombineDiscriminator64(num_classes=184)
    g.finalize(DEV, torch.float32), d.finalize(DEV, torch.float32)
    for m in g.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()  # the PSP stage norms are per-replica (un-synchronised) in the reference too: freeze them
    return g, d, L.GanTrainer(g, d)


def _batch():
    from layout2img_amd.synthetic import make_batch
    return make_batch(4, 64, "coco", seed=11, device="cpu")


def _run(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from layout2img_amd import parallel
    if world > 1:
        parallel.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    g, d, tr = _build()
    real, label, bbox, z, z_im = _batch()
    n = 4 // world
    sl = slice(rank * n, (rank + 1) * n)
    args = [t[sl].to(DEV) for t in (real, label, bbox, z, z_im)]
    for _ in range(2):
        r = tr.step(*args)
    assert world == 1 or tr._pending_g   # data parallel: the generator step of the last iteration is still in flight ...
    tr.flush()                           # ... (its all-reduce overlaps the next iteration's D(real)) until flushed
    torch.cuda.synchronize()
    if rank == 0:
        out["g"] = g.flat.data.detach().cpu()
        out["d"] = d.flat.data.detach().cpu()
        out["d_loss"] = float(r["d_loss"])
        out["g_loss"] = float(r["g_loss"])
        out["bn_mean"] = g.res5.b2.batch_norm2d.running_mean.detach().cpu()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


_SINGLE = {}


def _single_process_reference():
    """Two iterations of the plain single-process trainer on the 4-image batch (one child process, shared by the tests of this file)."""
    if not _SINGLE:
        out = mp.Manager().dict()
        mp.spawn(_run, args=(1, _free_port(), out), nprocs=1, join=True)
        _SINGLE.update(out)
    return _SINGLE


def test_two_ranks_equal_single_process():
    mgr = mp.Manager()
    single, multi = _single_process_reference(), mgr.dict()
    mp.spawn(_run, args=(2, _free_port(), multi), nprocs=2, join=True)
    # global-count losses: rank 0 holds its share of the global mean; the sum over ranks is the global loss, so
    # compare parameters (which see the all-reduced gradients) and synchronised BN statistics instead.
    assert torch.allclose(multi["bn_mean"], single["bn_mean"], atol=2e-3)   # second step sees 1-step-old Adam sign noise
    for k in ("g", "d"):
        a, b = multi[k], single[k]
        # Adam (beta1 = 0) moves each element by ~lr*sign(g): identical gradients up to round-off except at
        # elements whose gradient is round-off noise (atomics reorder sums run to run); a broken exchange would move
        # MOST elements by ~lr per step. Require 97 % of the elements to agree to 1e-4 (= one lr step) after 2 steps.
        frac = float(((a - b).abs() < 1e-4).float().mean())
        assert frac > 0.97, (k, frac)
        assert float((a - b).abs().max()) <= 2e-3   # a few opposite-sign Adam steps of <= sqrt(2) * lr each


def _run_grads(rank, world, port, out, dt_name):
    """One iteration at 128x128 with bf16 / f32 operands; rank 0 returns both networks' flat gradients (after the SUM
    all-reduce under data parallelism = the gradient of the global batch)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import layout2img_amd as L
    from layout2img_amd import parallel
    from layout2img_amd.synthetic import make_batch
    if world > 1:
        parallel.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    dt = getattr(torch, dt_name)
    torch.manual_seed(7)
    g = L.ResnetGenerator128_context(num_classes=184).finalize(DEV, dt)
    d = L.CombineDiscriminator128_app(num_classes=184).finalize(DEV, dt)
    for m in g.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()  # the PSP stage norms are per-replica (un-synchronised) in the reference too: freeze them
    tr = L.GanTrainer(g, d)
    real, label, bbox, z, z_im = make_batch(16, 128, "coco", seed=11, device="cpu")
    n = 16 // world
    sl = slice(rank * n, (rank + 1) * n)
    tr.step(*[t[sl].to(DEV) for t in (real, label, bbox, z, z_im)])
    tr.flush()
    torch.cuda.synchronize()
    if rank == 0:
        out["g"] = g.flat.grad.detach().cpu()
        out["d"] = d.flat.grad.detach().cpu()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize("dt_name", ["bfloat16"])   # (config 4's operand dtype; the exact-f32 exchange is the 64x64 test above. r05 ran both: 112 s)
def test_two_rank_gradients_equal_single_process_at_128(dt_name):
    """BASELINE config 4's arithmetic at its resolution and operand dtype: two data-parallel ranks (8 images each) produce the
    single-process gradient of the 16-image batch -- SyncBN statistics in both directions, global-count losses, SUM
    all-reduce of the flat gradients (on its own process group), compared BEFORE Adam's sign amplification.
    The bar calibrates itself: two SINGLE-process runs of the same iteration already differ (f32 atomics reorder sums, a
    pre-activation at ~0 lands on the other side of its ReLU gate, a value on a bf16 rounding boundary rounds the other way;
    measured here at b = 16: generator gradient relative L2 ~4e-4 f32, ~1.5e-2 bf16), and the two-rank gradient must lie
    within 2.5 x that run-to-run distance (+ 1e-5). One pair of runs is a noisy estimate of that distance (a pair that happens
    to flip no gate reads several times lower than the typical pair: the f32 case failed once that way), so it is floored at half the
    typical value. A broken exchange -- unsynchronised batch statistics, local loss counts, a missing all-reduce -- moves
    the gradient by O(0.1 - 1)."""
    mgr = mp.Manager()
    single, multi = mgr.dict(), mgr.dict()
    mp.spawn(_run_grads, args=(1, _free_port(), single, dt_name), nprocs=1, join=True)
    mp.spawn(_run_grads, args=(2, _free_port(), multi, dt_name), nprocs=2, join=True)
    for k in ("g", "d"):
        b = single[k]
        # Round 6: a single-process run is bit-reproducible (tests/test_gpu_06b_determinism.py), so the second single-process run that used to
        # measure the run-to-run floor is gone. Two ranks still sum in another partition (per-rank statistics and gradients, then the all-reduce):
        # rounding differs, in bf16 a value on a rounding boundary goes the other way. The bar is the one of rounds 3-5: 2.5 x the typical
        # distance of two bf16 runs then (7e-3), far below what a broken exchange gives (O(0.1 - 1)).
        floor = 2e-4 if dt_name == "float32" else 7e-3
        err = float((multi[k] - b).norm() / b.norm())
        print(f"two ranks vs one process [{dt_name}] {k}: {err:.2e}")
        assert err < 2.5 * floor + 1e-5, (k, err, floor)


def _run_forced(rank, port, out, force, graph):
    """One rank on cuda:0. force: L2I_FORCE_COLLECTIVES=1 -> a one-rank RCCL ("nccl") process group whose collectives all run
    (SyncBN statistics, ROI count, flat-gradient all-reduces on their own group, the deferred generator step); graph: the
    iteration is additionally captured with the collectives inside (L2I_DDP_GRAPH=1) and replayed."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      L2I_FORCE_COLLECTIVES="1" if force else "0", L2I_DDP_GRAPH="1" if graph else "0")
    import torch.distributed as dist
    from layout2img_amd import parallel
    from layout2img_amd.trainer import restore_state, snapshot_state
    calls = {"n": 0}
    if force:
        parallel.init_from_env()
        assert parallel.FORCE and parallel.active() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
        real_all_reduce = dist.all_reduce

        def counting(*a, **k):
            calls["n"] += 1
            return real_all_reduce(*a, **k)
        dist.all_reduce = counting
    else:
        assert not parallel.active()
    torch.cuda.set_device(0)
    g, d, tr = _build()
    assert tr.dp == force and tr.defer_g == force
    args = [t.to(DEV) for t in _batch()]
    if graph:
        st = snapshot_state(tr)
        assert tr.capture(*args)          # two warm-up iterations + the capture, collectives recorded on the capture stream
        restore_state(tr, st)
        for _ in range(2):
            r = tr.step_graphed(*args)
    else:
        for _ in range(2):
            r = tr.step(*args)
    assert tr._pending_g == (force and not graph)   # (a captured iteration keeps its generator step inside the graph)
    tr.flush()
    torch.cuda.synchronize()
    out["g"] = g.flat.data.detach().cpu()
    out["d"] = d.flat.data.detach().cpu()
    out["d_loss"], out["g_loss"] = float(r["d_loss"]), float(r["g_loss"])
    out["bn_mean"] = g.res5.b2.batch_norm2d.running_mean.detach().cpu()
    out["collectives"] = calls["n"]
    if force:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("graph", [False, True])
def test_forced_one_rank_collectives_over_rccl(graph):
    """RCCL carries the data-parallel iteration on the one test GPU: with L2I_FORCE_COLLECTIVES=1 a ONE-rank "nccl" group stops
    short-circuiting (parallel.active), so the SyncBN / ROI-count / gradient all-reduces, the gradient group of its own, the
    deferred generator step and -- graph=True -- capture + replay with the collectives inside (L2I_DDP_GRAPH=1) all execute
    on RCCL communicator streams. A SUM over one rank is the identity: losses, parameters and synchronised statistics must
    equal the short-circuited single-process run (same bar as the two-rank test above)."""
    mgr = mp.Manager()
    plain, forced = _single_process_reference(), mgr.dict()   # (the same two plain iterations the two-rank test compares with)
    mp.spawn(_run_forced, args=(_free_port(), forced, True, graph), nprocs=1, join=True)
    assert forced["collectives"] >= 20, forced["collectives"]   # (per eager iteration: ~30 SyncBN pairs + count + gradients)
    assert abs(forced["d_loss"] - plain["d_loss"]) < 2e-4 * abs(plain["d_loss"]) + 1e-5
    assert abs(forced["g_loss"] - plain["g_loss"]) < 2e-4 * abs(plain["g_loss"]) + 1e-5
    assert torch.allclose(forced["bn_mean"], plain["bn_mean"], atol=2e-3)
    for k in ("g", "d"):
        a, b = forced[k], plain[k]
        frac = float(((a - b).abs() < 1e-4).float().mean())
        assert frac > 0.97, (k, frac)
        assert float((a - b).abs().max()) <= 2e-3
