"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Replaces the reference's single-process nn.DataParallel + thread-based SyncBN
(train_context_app_v2.py:108-110, model/sync_batchnorm/batchnorm.py:59-125, comm.py):
  * gradients: the flat gradient buffer is all-reduced (SUM) in a few large contiguous chunks --
    no per-step parameter broadcast, no bucket copies (xGMI rings are per-link bound, so chunks
    are large: default 64 MiB);
  * SyncBN: each BN layer all-reduces its [sum, sqsum] (forward) and [s1, s2] (backward) vectors;
  * losses divide by GLOBAL counts, so SUM-reduced gradients equal the single-process gradients
    of the global batch (DataParallel semantics).
Everything here works on any backend (the CPU tests run it over gloo).
"""
import os

import torch
import torch.distributed as dist


# L2I_FORCE_COLLECTIVES=1: a ONE-rank process group still runs every collective of the data-parallel iteration (SyncBN
# statistics, valid-ROI count, flat-gradient all-reduces on their own group, parameter broadcasts) instead of
# short-circuiting them -- on a one-GPU box this is the only way the RCCL communicator streams, the deferred generator
# step and graph capture with collectives inside (L2I_DDP_GRAPH=1) ever execute. Results equal the short-circuited path.
FORCE = os.environ.get("L2I_FORCE_COLLECTIVES", "0") == "1"


class CommStats:
    """Accounting of one iteration's collectives for bench.py's N > 1 line (`comm`): how many were issued, how many bytes the
    flat-gradient all-reduces carried, and -- HIP events on the issuing stream around every blocking collective and around every
    wait for an asynchronous one -- for how long that stream was held up by them (`exposed_ms`: communication that did NOT
    overlap compute; read after a device synchronisation). Off unless bench.py switches it on for one eager iteration."""
    on = False
    collectives = 0
    allreduce_bytes = 0
    _pairs = []

    @classmethod
    def start(cls):
        cls.on, cls.collectives, cls.allreduce_bytes, cls._pairs = True, 0, 0, []

    @classmethod
    def bracket(cls):
        """context manager: events on the current stream around a call that makes it wait for a communicator stream"""
        import contextlib

        @contextlib.contextmanager
        def _b():
            if not cls.on or not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
                yield
                return
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            yield
            b.record()
            cls._pairs.append((a, b))
        return _b()

    @classmethod
    def stop(cls):
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        ms = sum(a.elapsed_time(b) for a, b in cls._pairs)
        cls.on = False
        out = dict(collectives_per_step=cls.collectives, allreduce_bytes_per_step=cls.allreduce_bytes, comm_exposed_ms=round(ms, 3))
        cls._pairs = []
        return out


def active():
    """True when the iteration must issue its collectives: more than one rank, or the forced one-rank group."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE)


def init_from_env(backend=None):
    """Initialise the process group from torchrun's environment. Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks: L2I_DIST_BACKEND=gloo and L2I_FORCE_DEVICE=0 let several ranks share one GPU (RCCL refuses that), which
    # is how the multi-process path of bench.py is exercised on a single-GPU box
    backend = os.environ.get("L2I_DIST_BACKEND", backend)
    if "L2I_FORCE_DEVICE" in os.environ:
        local = int(os.environ["L2I_FORCE_DEVICE"])
    if (world > 1 or FORCE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


_GRAD_GROUP = None


def grad_group():
    """A process group of its own for the gradient all-reduces. A ProcessGroup has ONE communicator stream per device, and
    collectives of one group run in issue order: on the default group the small blocking collectives of the next iteration
    (valid-ROI count, SyncBN statistics) would queue behind an asynchronously launched 163 MB gradient all-reduce and the
    main stream would wait for all of it -- no overlap. Created collectively (every rank calls this at the same point:
    GanTrainer.__init__); None at world size 1."""
    global _GRAD_GROUP
    if not active():
        return None
    if _GRAD_GROUP is None:
        _GRAD_GROUP = dist.new_group()
    return _GRAD_GROUP


_HOST_GROUP = None


def host_group():
    """A gloo group for HOST-side decisions every rank must take alike (GanTrainer.capture: "did every rank's capture succeed?").
    A CPU tensor over gloo works whatever state the HIP runtime of a rank is in -- an invalidated capture leaves it in a sticky error
    state in which a device all-reduce on the failing rank would itself fail or hang. Created collectively (every rank calls this at
    the same point, before the capture); None at world size 1 without the forced group."""
    global _HOST_GROUP
    if not active():
        return None
    if _HOST_GROUP is None:
        _HOST_GROUP = dist.new_group(backend="gloo")
    return _HOST_GROUP


def all_ranks_ok(ok, group=None):
    """MIN over the ranks of a host-side boolean."""
    if not active():
        return bool(ok)
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group if group is not None else host_group())
    return float(t) >= 1.0


def allreduce_flat_(flat_grad, chunk_bytes=64 << 20, async_op=False, group=None):
    """SUM all-reduce of a flat buffer in large contiguous chunks (in place). Returns work handles.
    group: the process group to run on (grad_group() for gradient exchanges that should overlap other collectives)."""
    if not active():
        return []
    n = flat_grad.numel()
    step = max(1, chunk_bytes // flat_grad.element_size())
    works = []
    for s in range(0, n, step):
        if CommStats.on:
            CommStats.collectives += 1
            CommStats.allreduce_bytes += (min(n, s + step) - s) * flat_grad.element_size()
        if async_op:
            works.append(dist.all_reduce(flat_grad[s:min(n, s + step)], op=dist.ReduceOp.SUM, async_op=True, group=group))
        else:
            with CommStats.bracket():
                dist.all_reduce(flat_grad[s:min(n, s + step)], op=dist.ReduceOp.SUM, group=group)
    return works


def wait_all(works):
    """The current stream waits for asynchronous collectives (their exposed time is accounted when CommStats is on)."""
    if works:
        with CommStats.bracket():
            for w in works:
                w.wait()


def sync_bn_stats(a, b, count):
    """SyncBN exchange: all-reduce two stat tensors (one message) and scale the element count.

    Forward: (sum, sqsum, count) -> returns global count. Backward: (s1, s2, None) -> returns None.
    """
    ws = world_size()
    if not active():
        return count
    adjacent = (a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.device == b.device and
                a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and
                b.storage_offset() == a.storage_offset() + a.numel())
    if CommStats.on:
        CommStats.collectives += 1
    if adjacent:   # the usual case: both halves of one statistics buffer -> reduce it in place, no staging copies
        both = a.new_empty(0).set_(a.untyped_storage(), a.storage_offset(), (a.numel() + b.numel(),), (1,))
        with CommStats.bracket():
            dist.all_reduce(both, op=dist.ReduceOp.SUM)
    else:
        buf = torch.cat((a.reshape(-1), b.reshape(-1)))
        with CommStats.bracket():
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        a.copy_(buf[:a.numel()].view_as(a))
        b.copy_(buf[a.numel():].view_as(b))
    return None if count is None else count * ws


def global_count(local_count_tensor):
    """All-reduce a 1-element f32 device tensor holding a row count (stays on the device)."""
    if active():
        if CommStats.on:
            CommStats.collectives += 1
        with CommStats.bracket():
            dist.all_reduce(local_count_tensor, op=dist.ReduceOp.SUM)
    return local_count_tensor


def broadcast_flat_(flat, src=0):
    if active():
        dist.broadcast(flat, src=src)
