"""CPU, gloo, world_size 2: the data-parallel helpers (layout2img_amd/parallel.py) reproduce single-process
results -- flat gradient all-reduce, SyncBN statistics exchange, global-count loss normalisation."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model_grads(x, y_valid, w, sync, n_roi, n_img, world):
    """A miniature of the hot path's data-parallel arithmetic: conv -> batch norm from (sum, sqsum) exchanged by
    `sync` -> ReLU -> per-image score and per-'ROI' score -> hinge terms divided by GLOBAL counts."""
    from layout2img_amd import parallel
    w = w.clone().requires_grad_(True)
    h = F.conv2d(x, w, padding=1)
    B, C, H, W = h.shape
    sums, sq = h.sum(dim=(0, 2, 3)), (h * h).sum(dim=(0, 2, 3))
    s2, q2 = sums.detach().clone(), sq.detach().clone()
    count = sync(s2, q2, float(B * H * W))
    # straight-through trick so autograd sees the global statistics: mean/var are functions of all ranks' data,
    # their gradient contribution is exchanged in the backward SyncBN all-reduce of the real kernels; here the
    # test only checks forward statistics + loss normalisation + gradient SUM, so statistics are constants.
    mean, var = s2 / count, q2 / count - (s2 / count) ** 2
    a = F.relu((h - mean.view(1, C, 1, 1)) / torch.sqrt(var.view(1, C, 1, 1) + 1e-5))
    img = a.mean(dim=(1, 2, 3))
    roi = a[:, :, :2, :2].reshape(B, -1).sum(1)
    loss = F.relu(1 - img).sum() / n_img + (F.relu(1 + roi) * y_valid).sum() / n_roi
    loss.backward()
    return w.grad, mean, var, loss.detach()


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from layout2img_amd import parallel
    r, w_, _ = parallel.init_from_env(backend="gloo")
    assert (r, w_) == (rank, world)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 3, 8, 8, generator=g)
    valid = torch.tensor([1.0, 0.0, 1.0, 1.0])
    w = torch.randn(5, 3, 3, 3, generator=g) * 0.3
    xs, vs = x[2 * rank:2 * rank + 2], valid[2 * rank:2 * rank + 2]
    n_roi = parallel.global_count(vs.sum().view(1).clone())
    grad, mean, var, loss = _model_grads(xs, vs, w, parallel.sync_bn_stats, float(n_roi), 2.0 * world, world)
    flat = torch.cat([grad.reshape(-1), torch.zeros(3)])
    parallel.allreduce_flat_(flat, chunk_bytes=64)            # many small chunks on purpose
    lt = loss.view(1).clone()
    torch.distributed.all_reduce(lt)
    p = torch.full((7,), float(rank))
    parallel.broadcast_flat_(p, src=0)
    # the host-side "did every rank succeed" decision of GanTrainer.capture (round 6: over a gloo group of its own, on CPU tensors --
    # a rank whose HIP runtime is unusable after an invalidated capture can still take part): MIN over the ranks
    host = parallel.host_group()
    votes = (parallel.all_ranks_ok(True, host), parallel.all_ranks_ok(rank == 0, host), parallel.all_ranks_ok(rank == 1, host))
    if rank == 0:
        ret.update(grad=flat[:-3].view_as(grad), mean=mean, var=var, loss=lt, n_roi=float(n_roi), bc=p, votes=votes)
    else:
        ret.update(votes1=votes)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_equals_single_process():
    from layout2img_amd import parallel
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 3, 8, 8, generator=g)
    valid = torch.tensor([1.0, 0.0, 1.0, 1.0])
    w = torch.randn(5, 3, 3, 3, generator=g) * 0.3
    assert parallel.world_size() == 1
    grad, mean, var, loss = _model_grads(x, valid, w, parallel.sync_bn_stats, float(valid.sum()), 4.0, 1)
    assert ret["n_roi"] == 3.0
    assert torch.allclose(ret["mean"], mean, atol=1e-6) and torch.allclose(ret["var"], var, atol=1e-6)
    assert torch.allclose(ret["loss"], loss.view(1), atol=1e-6)
    assert torch.allclose(ret["grad"], grad, atol=1e-6)
    assert torch.equal(ret["bc"], torch.zeros(7))
    assert tuple(ret["votes"]) == tuple(ret["votes1"]) == (True, False, False)   # every rank reaches the same decision
