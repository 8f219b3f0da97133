"""Host-side dry run of the hot path without a GPU: the models are built on torch's `meta` device (shapes, strides, autograd -- no
memory, no arithmetic) and `_lib.call` is replaced by a recorder, so a whole training iteration runs through exactly the Python that
runs on the GPU box -- trainer, autograd Functions, arena bookkeeping, argument marshalling -- and leaves the list of C-ABI calls it would
have made. That the path CAN run this way is itself a property the tests pin: it never reads a device value on the host (no `.item()`,
no `.cpu()`), which is what makes the iteration capturable as one HIP graph (GanTrainer.capture).

Test infrastructure (tests/test_cpu_dryrun.py, tools/perf/host_dryrun.py); nothing here computes anything."""
import contextlib
import ctypes

import torch


class _FakeStream:
    cuda_stream = 0

    def wait_stream(self, s):
        pass

    def wait_event(self, e):
        pass

    def synchronize(self):
        pass


class _FakeEvent:
    def record(self, s=None):
        pass

    def wait(self, s=None):
        pass


class _FakeStreamCtx:
    def __init__(self, *a):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


SCRATCH_PTR, WORKSPACE_PTR = 0x7000 << 40, 0x7001 << 40


class _Aten(torch.utils._python_dispatch.TorchDispatchMode):
    """Records every stock torch operator the iteration issues between the C-ABI calls: ("aten::...", operands) with each tensor operand as
    (pointer, shape, strides, dtype) -- on the GPU these are launches (or views) too, so a refactoring that is to leave the GPU's work
    unchanged must leave them unchanged as well."""

    def __init__(self, trace):
        super().__init__()
        self.trace = trace

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))

        def d(x):
            if isinstance(x, torch.Tensor):
                return ("T", x.data_ptr(), tuple(x.shape), tuple(x.stride()), str(x.dtype))
            if isinstance(x, (list, tuple)):
                return tuple(d(y) for y in x)
            if isinstance(x, (int, float, bool, str, type(None))):
                return x
            return str(x)
        self.trace.append((str(func), (d(args), d(tuple(sorted((kwargs or {}).items()))), d(out))))
        return out


def canonical(trace):
    """The trace with every fake address replaced by (index of its buffer in order of first appearance, byte offset): two runs that hand
    the GPU the same work in the same buffers-by-role compare equal whatever the allocator did."""
    ids = {}

    def c(x):
        if isinstance(x, tuple):
            if len(x) == 5 and x[0] == "T":
                return ("T", c(x[1])) + x[2:]
            return tuple(c(y) for y in x)
        if type(x) is int and x >= (1 << 40):
            base = x >> 40
            return ("buf", ids.setdefault(base, len(ids)), x & ((1 << 40) - 1))
        return x
    return [(name, c(args)) for name, args in trace]


@contextlib.contextmanager
def dry_run(pointers=False, aten=False):
    """Inside the block: tensors report `is_cuda`, streams / events are inert, every C-ABI call is appended to the yielded list as
    (name, args) after being checked against the ctypes table the way ctypes itself would convert it.
    pointers: `Tensor.data_ptr()` returns a fake address, (storage number << 40) + byte offset, so that the trace shows which calls share
    which buffers (compare with `canonical`); every storage seen stays alive until the block ends, so no address is ever reused.
    aten: stock torch operators are recorded as well (`_Aten`)."""
    from layout2img_amd import _lib, ops
    trace = []
    bases, keep = {}, []

    def data_ptr(self):
        st = self.untyped_storage()
        base = bases.get(st._cdata)
        if base is None:
            base = bases[st._cdata] = (len(bases) + 1) << 40
            keep.append(st)
        return base + self.storage_offset() * self.element_size()
    kinds = {ctypes.c_void_p: "p", ctypes.c_int: "i", ctypes.c_longlong: "l", ctypes.c_float: "f"}

    def record(name, *args):
        sig = _lib.SIGNATURES[name]   # KeyError: an entry point include/l2i.h does not declare
        if len(args) != len(sig):
            raise TypeError(f"{name}: {len(args)} arguments for a {len(sig)}-parameter entry point")
        args = list(args)
        for k, (a, t) in enumerate(zip(args, sig)):
            kind = kinds[t]
            if kind == "p" and isinstance(a, ctypes.Array):   # a small table of pointers / ints passed by address
                args[k] = ("array", tuple(a))
                ok = True
            elif kind == "p":
                ok = a is None or (type(a) is int and a >= 0)
            elif kind == "i":
                ok = isinstance(a, int) and -2 ** 31 <= a < 2 ** 31   # (ctypes would truncate a wider value silently)
            elif kind == "l":
                ok = isinstance(a, int) and -2 ** 63 <= a < 2 ** 63
            else:
                ok = isinstance(a, (int, float)) and not isinstance(a, bool) and a == a
            if not ok:
                raise TypeError(f"{name}: argument {k} = {a!r} does not fit its parameter ({kind})")
        trace.append((name, tuple(args)))

    saved = [(_lib, "call", _lib.call), (_lib, "raw_stream", _lib.raw_stream), (_lib, "current_device", _lib.current_device),
             (_lib, "workspace", _lib.workspace), (_lib, "wgrad_scratch", _lib.wgrad_scratch), (ops, "_ws", ops._ws),
             (torch.cuda, "current_stream", torch.cuda.current_stream), (torch.cuda, "Stream", torch.cuda.Stream),
             (torch.cuda, "Event", torch.cuda.Event), (torch.cuda, "stream", torch.cuda.stream),
             (torch.cuda, "synchronize", torch.cuda.synchronize),
             (torch.cuda, "is_current_stream_capturing", torch.cuda.is_current_stream_capturing)]
    _lib.call = record
    _lib.raw_stream = lambda: 0
    _lib.current_device = lambda: None          # == torch.device("meta").index
    _lib.workspace = lambda device: WORKSPACE_PTR if pointers else 0
    _lib.wgrad_scratch = lambda device: (SCRATCH_PTR if pointers else 0, _lib.WGRAD_SCRATCH_FLOATS)
    ops._ws = _lib.workspace
    torch.cuda.current_stream = lambda *a: _FakeStream()
    torch.cuda.Stream = _FakeStream
    torch.cuda.Event = lambda *a, **k: _FakeEvent()
    torch.cuda.stream = _FakeStreamCtx
    torch.cuda.synchronize = lambda *a: None
    torch.cuda.is_current_stream_capturing = lambda: False
    torch.Tensor.is_cuda = property(lambda self: True)   # (shadows TensorBase's descriptor; deleted again below)
    if pointers:
        torch.Tensor.data_ptr = data_ptr
    pool = (ops.POOL.__dict__.pop("slabs", None), ops.POOL.buf)
    mode = _Aten(trace) if aten else contextlib.nullcontext()
    try:
        with mode:
            yield trace
    finally:
        del torch.Tensor.is_cuda
        if pointers:
            del torch.Tensor.data_ptr
        for obj, name, val in saved:
            setattr(obj, name, val)
        ops.POOL.__dict__.pop("slabs", None)
        if pool[0] is not None:
            ops.POOL.slabs = pool[0]
        ops.POOL.buf, ops.POOL.active = pool[1], False


def header_parameters():
    """{entry point: [parameter name, ...]} parsed from include/l2i.h, so a test can name the argument it looks at."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "l2i.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    hdr = re.sub(r"//[^\n]*", " ", hdr)
    out = {}
    for m in re.finditer(r"^int\s+(l2i_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.M | re.S):
        names = []
        for a in [x.strip() for x in m.group(2).replace("\n", " ").split(",") if x.strip() and x.strip() != "void"]:
            names.append(re.findall(r"\w+", a)[-1])
        out[m.group(1)] = names
    return out


def build(kind, dtype, size=128, vgg=False):
    """(trainer, inputs) of one of BASELINE.json's configurations on the meta device."""
    import layout2img_amd as L
    from layout2img_amd.synthetic import make_batch
    dev = torch.device("meta")
    if size == 64:
        g, d = L.ResnetGenerator64_context(num_classes=184), L.CombineDiscriminator64(num_classes=184)
    elif kind == "vg":
        g, d = L.context_aware_generator(num_classes=179), L.CombineDiscriminator128_app(num_classes=179)
    else:
        g, d = L.ResnetGenerator128_context(num_classes=184), L.CombineDiscriminator128_app(num_classes=184)
    g, d = g.finalize(dev, dtype), d.finalize(dev, dtype)
    v = L.VGGLoss().finalize(dev, dtype) if vgg else None
    tr = L.GanTrainer(g, d, vgg=v)
    real, label, bbox, z, z_im = make_batch(4, size, kind, seed=1234, device="cpu")
    return tr, tuple(None if t is None else t.to(dev) for t in (real, label, bbox, z, z_im))


@contextlib.contextmanager
def dry_ddp(world=2):
    """Inside the block (and inside dry_run): torch.distributed reports an initialised group of `world` ranks and records every collective as
    (kind, bytes, async_op, own_group) instead of running it -- the N > 1 iteration's communication schedule without a second process."""
    import torch.distributed as dist
    coll = []

    class _Work:
        def wait(self):
            pass

    def all_reduce(t, op=None, async_op=False, group=None):
        coll.append(("all_reduce", t.numel() * t.element_size(), bool(async_op), group is not None))
        return _Work()

    def broadcast(t, src=0, **kw):
        coll.append(("broadcast", t.numel() * t.element_size(), False, False))

    saved = [(dist, n, getattr(dist, n)) for n in ("is_initialized", "get_world_size", "all_reduce", "broadcast", "new_group", "get_backend")]
    dist.is_initialized = lambda: True
    dist.get_world_size = lambda group=None: world
    dist.all_reduce, dist.broadcast = all_reduce, broadcast
    dist.new_group = lambda *a, **k: object()
    dist.get_backend = lambda *a: "gloo"
    from layout2img_amd import parallel
    groups = (parallel._GRAD_GROUP, parallel._HOST_GROUP)
    try:
        yield coll
    finally:
        for obj, name, val in saved:
            setattr(obj, name, val)
        parallel._GRAD_GROUP, parallel._HOST_GROUP = groups


def header_pointer_kinds():
    """{entry point: [None (not a pointer) | "in" (const T*) | "out" (T*), ...]} from include/l2i.h."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "l2i.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    hdr = re.sub(r"//[^\n]*", " ", hdr)
    out = {}
    for m in re.finditer(r"^int\s+(l2i_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.M | re.S):
        ps = [x.strip() for x in m.group(2).replace("\n", " ").split(",") if x.strip() and x.strip() != "void"]
        out[m.group(1)] = [None if "*" not in a else ("in" if a.startswith("const") else "out") for a in ps]
    return out


_ALLOC = ("aten.empty.memory_format", "aten.empty_like.default", "aten.new_empty.default", "aten.empty_strided.default", "aten.new_empty_strided.default")
_VIEWS = ("aten.slice.Tensor", "aten.view.default", "aten.select.int", "aten.detach.default", "aten.as_strided.default", "aten.alias.default",
          "aten.expand.default", "aten.permute.default", "aten.transpose.int", "aten.t.default", "aten.unsqueeze.default", "aten._unsafe_view.default",
          "aten.squeeze.dim", "aten.narrow.default")


def uninitialised_reads(previous, trace, kinds=None):
    """Launches of `trace` (dry_run(pointers=True, aten=True)) that read a buffer nothing has written: a buffer that did not exist in the
    `previous` iteration's trace is a temporary; one made by torch.empty(_like) holds garbage until a C-ABI call names it through a
    non-const parameter or a stock operator produces / updates it; naming it through a `const` parameter (or as a stock operator's input)
    before that is a read of uninitialised memory. Whole buffers, not byte ranges: a partly written buffer counts as written.
    Returns [(position, entry point | operator, parameter, buffer number)]."""
    kinds = header_pointer_kinds() if kinds is None else kinds
    names = header_parameters()

    def buffers(x, out):
        if isinstance(x, tuple):
            if len(x) == 5 and x[0] == "T":
                out.append(x[1] >> 40)
            else:
                for y in x:
                    buffers(y, out)
        return out

    persistent = {SCRATCH_PTR >> 40, WORKSPACE_PTR >> 40}
    for name, args in previous:
        if name.startswith("l2i_"):
            persistent.update(a >> 40 for a in args if type(a) is int and a >= (1 << 40))
        else:
            persistent.update(buffers(args, []))
    written, found = set(), []
    for pos, (name, args) in enumerate(trace):
        if name.startswith("l2i_"):
            for kind, pname, a in zip(kinds[name], names[name], args):
                if kind is None or type(a) is not int or a < (1 << 40) or (a >> 40) in persistent:
                    continue
                if kind == "out":
                    written.add(a >> 40)
                elif (a >> 40) not in written:
                    found.append((pos, name, pname, a >> 40))
        elif name in _VIEWS:
            continue
        else:
            ins, outs = buffers(args[0], []), buffers(args[2], [])
            if name not in _ALLOC:
                found.extend((pos, name, "operand", b) for b in ins if b not in persistent and b not in written and b not in outs)
                written.update(outs)
                if name.split(".")[1].endswith("_"):
                    written.update(ins[:1])
    return found
