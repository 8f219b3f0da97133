"""Torch-facing wrappers of the C ABI (include/l2i.h) and the autograd Functions built on them.

Internal tensor conventions: activations are plain contiguous NHWC tensors (B, H, W, C) -- linear
layers are (rows, 1, 1, C); "streams" are f32, MFMA "operands" are `op_dtype` (bf16 | f32); masks
are planar (B, O, H, W) f32. Everything here requires a GPU and the built library: there is no
fallback of any kind.
"""
import ctypes

import torch
from torch.autograd import Function

from . import _lib
from .arena import GemmWeight, PassCtx


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return _lib.raw_stream()


def _code(dtype):
    return _lib.BF16 if dtype == torch.bfloat16 else _lib.F32


def _chk(t, dtype=None):
    if not t.is_cuda:
        raise RuntimeError("layout2img_amd ops run on the GPU only (no CPU fallback)")
    if t.device.index != _lib.current_device():
        raise RuntimeError(f"tensor on cuda:{t.device.index} but the current device is cuda:{_lib.current_device()}: "
                           "kernels are launched on the current device's stream (torch.cuda.set_device first)")
    if not t.is_contiguous():
        raise RuntimeError("layout2img_amd ops need contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"expected {dtype}, got {t.dtype}")
    return t


# ----------------------------------------------------------------------------- zero pool
class ZeroPool:
    """One pre-zeroed slab per training step for the many small atomically-accumulated outputs (channel sums,
    s1/s2, projection gradients): one memset instead of hundreds of tiny fill launches. Slices stay valid until the
    next begin(); outside a step (inactive) callers fall back to torch.zeros."""

    def __init__(self):
        self.buf, self.off, self.active = None, 0, False

    def begin(self, device, nfloats=8 << 20):
        key = (str(torch.device(device)), nfloats)   # one slab per (device, size): a training step and a stand-alone forward
        slabs = self.__dict__.setdefault("slabs", {})   # use different sizes and must not re-allocate each other's buffer
        buf = slabs.get(key)
        if buf is None:
            buf = slabs[key] = torch.empty(nfloats, dtype=torch.float32, device=device)
        self.buf = buf
        self.buf.zero_()
        self.off, self.active = 0, True

    def end(self):
        self.active = False

    def step(self, device, nfloats=8 << 20):
        """Context manager form: `with POOL.step(dev): ...` -- the pool is deactivated even when the step raises, so a
        failed iteration cannot leave later forwards / backwards taking never-re-zeroed slices."""
        pool = self

        class _Step:
            def __enter__(self_):
                pool.begin(device, nfloats)
                return pool

            def __exit__(self_, *exc):
                pool.end()
                return False
        return _Step()

    def take(self, shape, device):
        n = 1
        for d in shape:
            n *= d
        n4 = (n + 3) // 4 * 4
        if not self.active or n > (1 << 18) or self.off + n4 > self.buf.numel() or self.buf.device != torch.device(device):
            return torch.zeros(shape, dtype=torch.float32, device=device)
        v = self.buf[self.off:self.off + n].view(shape)
        self.off += n4
        return v


POOL = ZeroPool()


_ws = _lib.workspace


def _zeros(shape, device):
    return POOL.take(tuple(shape), device)


# ----------------------------------------------------------------------------- raw kernels
def cast_op(x, op_dtype, raw=True, act=False):
    """f32 stream -> operand copies. Returns (raw_copy | None, relu_copy | None)."""
    _chk(x, torch.float32)
    r = torch.empty(x.shape, dtype=op_dtype, device=x.device) if raw else None
    a = torch.empty(x.shape, dtype=op_dtype, device=x.device) if act else None
    _lib.call("l2i_cast_op", x.data_ptr(), _p(r), _p(a), x.numel(), _code(op_dtype), _stream())
    return r, a


def split_cast(x, relu=False):
    """f32 stream (..., C) -> the split bf16 operand (..., 3 C) = [hi | lo | hi] of the forward-only "bf16x3" mode (arena split)."""
    _chk(x, torch.float32)
    C = x.shape[-1]
    out = torch.empty(x.shape[:-1] + (3 * C,), dtype=torch.bfloat16, device=x.device)
    _lib.call("l2i_split_cast", x.data_ptr(), out.data_ptr(), x.numel() // C, C, int(relu), _stream())
    return out


class KernelTimer:
    """Timing of the individual conv / weight-gradient launches of an eager iteration (bench.py's roofline leg): the
    algorithmic work of every launch is accumulated here, the durations come from HIP events the library attaches to
    each DISPATCH (l2i_timing: hipExtLaunchKernelGGL start / stop events = the kernel's own begin / end on its stream,
    which is what rocprofv3's kernel trace reports)."""
    CLASSES = {"conv_igemm": 0, "conv_wgrad": 1}

    def __init__(self):
        self.acc = {k: [0, 0.0, 0.0] for k in self.CLASSES}   # launches, work, bytes
        _lib.call("l2i_timing", 1)

    def time(self, name, work, nbytes=0.0):
        a = self.acc[name]
        a[0] += 1
        a[1] += work
        a[2] += nbytes
        return None

    def summary(self):
        import ctypes
        torch.cuda.synchronize()
        out = {}
        for name, cls in self.CLASSES.items():
            ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
            _lib.call("l2i_timing_read", cls, ctypes.byref(ms), ctypes.byref(n))
            a = self.acc[name]
            # (a split weight-gradient launch is two kernels: tiles + reduce; a conv launch whose handed-over shortcut the
            #  library un-folds -- split-K grids at small batch -- is two kernels as well: their time belongs to the launch)
            if n.value < a[0]:
                raise RuntimeError(f"{name}: {a[0]} launches accounted, {n.value} timed")
            out[name] = dict(launches=a[0], kernels=n.value, ms=ms.value, work=a[1], bytes=a[2])
        return out

    def close(self):
        _lib.call("l2i_timing", 0)


TIMER = None  # set to a KernelTimer by bench.py
LIVE_IMAGE_FRACTION = 1.0   # bench.py: real ROIs / ROI slots of its batch, so that launches limited by a device-side image
                            # count (`nimg`) are credited with the work on live rows only


def dual_conv_ok(B, Ho, Wo):
    """A dual launch (two weight packs, images [0, B/2) and [B/2, B)) needs every tile inside one half: tiles are at most 256
    pixels = whole rows of min(Wo, 16) pixels. Otherwise conv_raw / wgrad_raw issue the halves as two launches."""
    return B % 2 == 0 and ((B // 2) * Ho * min(Wo, 16)) % 256 == 0


def conv_raw(x_op, wpack, kpad, co, kh, *, bias=None, res=None, relu_mask=None, up2=False, pool2=False, alpha=1.0,
             want_f32=True, want_op=False, relu_op=False, want_raw=False, flops=None, nimg=None, stats=False, sc=None,
             wpack_b=None, _outs=None):
    """out = alpha*pool?(conv(up?(x))) + bias, masked, + res.  x_op (B,Hi,Wi,Ci) operand dtype.
    wpack_b: DUAL launch (arena.DualPass) -- images [B/2, B) are multiplied with this pack (and sc["wpack_b"]), `nimg` counts
    the live images of each half (l2i_conv2d_fwd_dual).
    nimg: 1-element int32 device tensor = number of leading images that are live (the rest come out as zeros).
    sc: a residual block's 1x1 shortcut to fold into this launch (l2i_conv2d_fwd_sc) -- dict(x_op, wpack, kpad, bias, up2,
    out, flops): the launch adds conv1x1(sc.x_op) + sc.bias on its pre-pool grid instead of reading a residual.
    stats: the epilogue also gathers the per-channel sum / sum of squares of `out` (the batch statistics of a following
    normalisation); they ride on `out` (`_l2i_stats`) for ops._norm_stats."""
    _chk(x_op)
    B, Hi, Wi, Ci = x_op.shape
    Ho, Wo = (2 * Hi, 2 * Wi) if up2 else (Hi, Wi)
    Hq, Wq = (Ho // 2, Wo // 2) if pool2 else (Ho, Wo)
    dev = x_op.device
    if _outs is not None:   # (the halves of a dual call that cannot run as one launch: views of the caller's result tensors)
        out, out_op, out_raw = _outs
    else:
        out = torch.empty((B, Hq, Wq, co), dtype=torch.float32, device=dev) if want_f32 else None
        out_op = torch.empty((B, Hq, Wq, co), dtype=x_op.dtype, device=dev) if want_op else None
        out_raw = torch.empty((B, Hq, Wq, co), dtype=x_op.dtype, device=dev) if want_raw else None
    if wpack_b is not None and not dual_conv_ok(B, Ho, Wo):
        assert not stats and B % 2 == 0
        hb = B // 2
        for k, wp in enumerate((wpack, wpack_b)):
            sl = slice(k * hb, (k + 1) * hb)
            sck = None if sc is None else dict(sc, x_op=sc["x_op"][sl], out=sc["out"][sl], wpack=sc["wpack_b"] if k else sc["wpack"],
                                               wpack_b=None, flops=0.5 * sc["flops"])
            conv_raw(x_op[sl], wp, kpad, co, kh, bias=bias, res=None if res is None else res[sl],
                     relu_mask=None if relu_mask is None else relu_mask[sl], up2=up2, pool2=pool2, alpha=alpha, want_f32=want_f32,
                     want_op=want_op, relu_op=relu_op, want_raw=want_raw, flops=None if flops is None else 0.5 * flops, nimg=nimg, sc=sck,
                     _outs=tuple(None if t is None else t[sl] for t in (out, out_op, out_raw)))
        return out, out_op, out_raw
    if res is not None:
        _chk(res, torch.float32)
        assert res.shape == (B, Hq, Wq, co), (res.shape, (B, Hq, Wq, co))
    if relu_mask is not None:
        _chk(relu_mask, x_op.dtype)
        assert relu_mask.shape == (B, Hq, Wq, co), (relu_mask.shape, (B, Hq, Wq, co))
    end = None
    if TIMER is not None:
        esz = x_op.element_size()   # algorithmic bytes: every operand and result once
        nbytes = (x_op.numel() + wpack.numel()) * esz + B * Hq * Wq * co * (
            4 * (want_f32 + (res is not None)) + esz * (want_op + want_raw + (relu_mask is not None)))
        live = LIVE_IMAGE_FRACTION if nimg is not None else 1.0
        fl = flops if flops is not None else 2.0 * B * Ho * Wo * co * kh * kh * Ci
        if wpack_b is not None:
            nbytes += wpack_b.numel() * esz
        if sc is not None:   # the folded shortcut's work and operands belong to this launch
            fl += sc["flops"]
            nbytes += (sc["x_op"].numel() + sc["wpack"].numel() * (2 if wpack_b is not None else 1)) * esz
        end = TIMER.time("conv_igemm", live * fl, live * nbytes)
    st = _zeros((2, 1, co), dev) if (stats and out is not None and co % 4 == 0) else None
    # the stream's scratch (caller-owned, shared with the weight-gradient launches of the same stream): partial tiles of the
    # 4x4 weight-stationary kernel and of every split-K launch of the halo kernels (conv_store_partial + reduce with epilogue)
    # ... and the partial rows of the epilogue's batch statistics (one per wave, summed in a fixed order: no atomics)
    scr, nscr = _lib.wgrad_scratch(dev) if (x_op.dtype == torch.bfloat16 or st is not None) else (None, 0)
    if scr is not None and sc is None:
        _lib.call("l2i_conv2d_fwd_dual", x_op.data_ptr(), wpack.data_ptr(), _p(bias), _p(res), _p(relu_mask), _p(out), _p(out_op),
                  _p(out_raw), _code(x_op.dtype), B, Hi, Wi, Ci, Ho, Wo, co, kh, int(up2), int(pool2), int(relu_op), kpad,
                  float(alpha), _p(nimg), _p(st), _ws(dev) if st is not None else None, None, None, None, None, 0, 0, 0, 0, 0,
                  _p(wpack_b), None, scr, nscr, _stream())
    elif sc is None and wpack_b is None:
        _lib.call("l2i_conv2d_fwd", x_op.data_ptr(), wpack.data_ptr(), _p(bias), _p(res), _p(relu_mask), _p(out), _p(out_op),
                  _p(out_raw), _code(x_op.dtype), B, Hi, Wi, Ci, Ho, Wo, co, kh, int(up2), int(pool2), int(relu_op), kpad,
                  float(alpha), _p(nimg), _p(st), _ws(dev) if st is not None else None, _stream())
    elif sc is None:
        _lib.call("l2i_conv2d_fwd_dual", x_op.data_ptr(), wpack.data_ptr(), _p(bias), _p(res), _p(relu_mask), _p(out), _p(out_op),
                  _p(out_raw), _code(x_op.dtype), B, Hi, Wi, Ci, Ho, Wo, co, kh, int(up2), int(pool2), int(relu_op), kpad,
                  float(alpha), _p(nimg), None, None, None, None, None, None, 0, 0, 0, 0, 0, wpack_b.data_ptr(), None, None, 0, _stream())
    elif sc.get("mask_first"):
        # data gradient of a pre-activation block's conv1 with the data gradient of the block's 1x1 shortcut folded in (l2i_conv2d_dgrad_sc):
        # out = relu_mask > 0 ? alpha * conv3x3(x_op) : 0  +  sc.alpha * conv1x1(sc.x_op at (y >> up2, x >> up2))  + res
        sx = sc["x_op"]
        _chk(sx, torch.bfloat16), _chk(x_op, torch.bfloat16)
        assert kh == 3 and not up2 and not pool2 and bias is None and relu_mask is not None and not want_op and st is None and wpack_b is None
        assert sc["out"].shape == (B, Hq, Wq, co) and sx.shape[0] == B
        _lib.call("l2i_conv2d_dgrad_sc", x_op.data_ptr(), wpack.data_ptr(), _p(res), relu_mask.data_ptr(), _p(out), _p(out_raw), B, Hi, Wi, Ci, co, kpad,
                  float(alpha), _p(nimg), sx.data_ptr(), sc["wpack"].data_ptr(), sc["out"].data_ptr(), sx.shape[1], sx.shape[2], sx.shape[3],
                  int(sc["up2"]), sc["kpad"], float(sc["alpha"]), scr, nscr, _stream())
    else:
        sx = sc["x_op"]
        _chk(sx, x_op.dtype)
        assert res is None and relu_mask is None and sc["out"].shape == (B, Hq, Wq, co) and sx.shape[0] == B
        assert (wpack_b is None) == (sc.get("wpack_b") is None)
        _lib.call("l2i_conv2d_fwd_dual", x_op.data_ptr(), wpack.data_ptr(), _p(bias), None, None, _p(out), _p(out_op),
                  _p(out_raw), _code(x_op.dtype), B, Hi, Wi, Ci, Ho, Wo, co, kh, int(up2), int(pool2), int(relu_op), kpad,
                  float(alpha), _p(nimg), _p(st), _ws(dev) if st is not None else None,
                  sx.data_ptr(), sc["wpack"].data_ptr(), _p(sc["bias"]), sc["out"].data_ptr(), sx.shape[1], sx.shape[2], sx.shape[3],
                  int(sc["up2"]), sc["kpad"], _p(wpack_b), _p(sc.get("wpack_b")), scr, nscr, _stream())
    if st is not None:
        out._l2i_stats = (st[0], st[1], out._version)
    if end is not None:
        end.record()
    return out, out_op, out_raw


class WgradSide:
    """Weight-gradient launches on a side stream. A layer's weight gradient is needed only at the optimizer step,
    while its data gradient is on the critical path of the backward pass: with the weight gradients forked onto a second
    stream (one per launching stream) the chain of data-gradient launches no longer waits for them, and the two kinds of
    kernels fill each other's tails and under-occupied grids. Joined in WeightArena.flush_grads (before the
    spectral-norm backward / Adam read the accumulators). L2I_WGRAD_STREAM=1 turns it on."""
    enabled = __import__("os").environ.get("L2I_WGRAD_STREAM", "0") == "1"   # measured: 31.4 vs 30.0 ms per iteration with it ON (the two kernel
    # kinds thrash each other's L2 more than they fill tails): kept as an option, off by default
    _streams, _dirty = {}, {}

    @classmethod
    def fork(cls):
        cur = torch.cuda.current_stream()
        side = cls._streams.get(cur.cuda_stream)
        if side is None:
            side = cls._streams[cur.cuda_stream] = torch.cuda.Stream()
        side.wait_stream(cur)
        cls._dirty[side.cuda_stream] = side
        return side

    @classmethod
    def join(cls):
        cur = torch.cuda.current_stream()
        for side in cls._dirty.values():
            cur.wait_stream(side)
        cls._dirty.clear()


def wgrad_side(x_op, dy_op, dw, *args, **kw):
    """wgrad_raw on the weight-gradient side stream (or in place when that is off / while launches are being timed)."""
    if not WgradSide.enabled or TIMER is not None:
        return wgrad_raw(x_op, dy_op, dw, *args, **kw)
    side = WgradSide.fork()
    with torch.cuda.stream(side):
        wgrad_raw(x_op, dy_op, dw, *args, **kw)
    x_op.record_stream(side)
    dy_op.record_stream(side)
    sc = kw.get("sc")
    if sc is not None:   # the folded shortcut's operand and gradient targets are read / written by the same side-stream launch
        for t in (sc.get("x_op"), sc.get("dw"), sc.get("dw_b"), sc.get("dbias")):
            if t is not None:
                t.record_stream(side)


def wgrad_raw(x_op, dy_op, dw, ldw, co, kh, *, up2=False, pool2=False, alpha=1.0, flops=None, nimg=None, dbias=None, sc=None, dw_b=None,
              overwrite=False):
    """overwrite: this launch is the only writer of its dW slices since they were zeroed (FusedConvFn: one launch per layer application
    and pass) -- the library stores the result instead of accumulating (no atomics on single-split tiles, no read in the reduce).
    dbias: optional (co,) f32 tensor the bias gradient is atomically added to (summed from the staged dY tiles).
    dw_b: DUAL launch (arena.DualPass) -- the gradient of images [B/2, B) goes to this accumulator (and sc["dw_b"]); `nimg` counts
    the live images of each half (l2i_conv2d_wgrad_dual).
    sc: a block's 1x1 shortcut that received the same dY -- dict(x_op (B, Ho, Wo, Ci_sc), dw, ldw, dbias, flops): its weight (and
    bias) gradient become extra column tiles of this launch (l2i_conv2d_wgrad_sc)."""
    _chk(x_op)
    _chk(dy_op, x_op.dtype)
    B, Hi, Wi, Ci = x_op.shape
    Ho, Wo = (2 * Hi, 2 * Wi) if up2 else (Hi, Wi)
    if dw_b is not None and (B % 2 or ((B // 2) * Ho * Wo) % 64):   # halves that are not whole pixel steps: two launches
        hb = B // 2
        assert B % 2 == 0
        for k, d in enumerate((dw, dw_b)):
            sl = slice(k * hb, (k + 1) * hb)
            sck = None if sc is None else dict(sc, x_op=sc["x_op"][sl], dw=sc["dw_b"] if k else sc["dw"], dw_b=None, flops=0.5 * sc["flops"])
            wgrad_raw(x_op[sl], dy_op[sl], d, ldw, co, kh, up2=up2, pool2=pool2, alpha=alpha, flops=None if flops is None else 0.5 * flops,
                      nimg=nimg, dbias=dbias, sc=sck, overwrite=overwrite)
        return
    end = None
    if TIMER is not None:
        live = LIVE_IMAGE_FRACTION if nimg is not None else 1.0
        fl = flops if flops is not None else 2.0 * B * Ho * Wo * co * kh * kh * Ci
        end = TIMER.time("conv_wgrad", live * (fl + (sc["flops"] if sc is not None else 0.0)))
    scratch, nscratch = _lib.wgrad_scratch(x_op.device)
    sx = su = None
    if sc is not None:
        sx = sc["x_op"]
        _chk(sx, x_op.dtype)
        su = int(bool(sc.get("up2", False)))
        assert sx.shape[:3] == (B, Ho >> su, Wo >> su)
        assert (dw_b is None) == (sc.get("dw_b") is None)
    _lib.call("l2i_conv2d_wgrad_dual", x_op.data_ptr(), dy_op.data_ptr(), dw.data_ptr(), _code(x_op.dtype), B, Hi, Wi, Ci, Ho,
              Wo, co, kh, int(up2), int(pool2), ldw, float(alpha), _p(nimg), _p(dbias), scratch, nscratch,
              _p(sx), None if sc is None else sc["dw"].data_ptr(), 0 if sc is None else sx.shape[3], su or 0, 0 if sc is None else sc["ldw"],
              None if sc is None else _p(sc["dbias"]), _p(dw_b), None if sc is None else _p(sc.get("dw_b")), int(bool(overwrite)), _stream())
    if end is not None:
        end.record()


def channel_stats(x2d, rows_per_group=None, want_sq=True, cast_to=None, accumulate_into=None):
    """x2d: (rows, C) f32 view. Returns sums (G, C), sqsums (G, C)|None [, operand-dtype copy if cast_to].
    accumulate_into: an existing (C,) f32 tensor the sums are atomically ADDED to (e.g. a bias .grad view)."""
    _chk(x2d, torch.float32)
    rows, C = x2d.shape
    rpg = rows if rows_per_group is None else rows_per_group
    G = rows // rpg
    if accumulate_into is not None:
        assert G == 1 and not want_sq and accumulate_into.numel() == C and accumulate_into.is_contiguous()
        sums, sq = accumulate_into.view(1, C), None
    else:
        buf = _zeros((2 if want_sq else 1, G, C), x2d.device)
        sums, sq = buf[0], (buf[1] if want_sq else None)
    raw = torch.empty(x2d.shape, dtype=cast_to, device=x2d.device) if cast_to is not None else None
    scr, nscr = _lib.wgrad_scratch(x2d.device)   # partial rows of the slabs, summed in a fixed order (no atomics)
    _lib.call("l2i_channel_stats", x2d.data_ptr(), rows, C, rpg, sums.data_ptr(), _p(sq), _p(raw),
              _code(cast_to) if cast_to is not None else _lib.F32, scr, nscr, _stream())
    if cast_to is not None:
        return sums, sq, raw
    return sums, sq


# ----------------------------------------------------------------------------- prologues
class Prologue:
    """How a conv obtains its operand from an f32 stream: 'cast', 'relu', or a fused norm."""
    kind = "cast"


class NormSpec(Prologue):
    """Normalise (+modulate) (+ReLU) prologue. mode 0 ISLA, 1 affine, 2 none; instance=True groups
    statistics per image (InstanceNorm). `sync` is a callable all-reducing stat tensors in place."""
    kind = "norm"

    def __init__(self, mode, eps=1e-5, relu=True, instance=False, sync=None, running=None, momentum=0.1,
                 training=True):
        self.mode, self.eps, self.relu, self.instance = mode, eps, relu, instance
        self.sync, self.running, self.momentum, self.training = sync, running, momentum, training


def _norm_stats(x, spec: NormSpec):
    """Returns (sums, sqsums, count, stat_stride). Updates running stats in train mode."""
    B, H, W, C = x.shape
    x2 = x.view(B * H * W, C)
    if spec.instance:
        sums, sq = channel_stats(x2, H * W)
        return sums, sq, float(H * W), C
    if not spec.training and spec.running is not None:
        rm, rv = spec.running
        return rm.view(1, C).clone(), (rv + rm * rm).view(1, C), 1.0, 0
    st = getattr(x, "_l2i_stats", None)   # gathered by the epilogue of the convolution that produced x (fused_conv emit "stats")
    if st is not None and st[2] == x._version and st[0].shape == (1, C):
        sums, sq = st[0], st[1]
    else:
        sums, sq = channel_stats(x2)
    count = float(B * H * W)
    if spec.sync is not None:
        count = spec.sync(sums, sq, count)
    # (train-mode running statistics are updated by the normalisation launch itself: norm_fwd_raw(update_running=True))
    return sums, sq, count, 0


def _proj_strides(wproj, bproj, B, O, C):
    """ISLA projections (B, O, C): contiguous, or slices of a wider (B*O, N) matrix (GroupedLinearFn) -- the kernels
    address them as b*stride_b + o*stride_o + c."""
    for t in (wproj, bproj):
        if not t.is_cuda or t.dtype != torch.float32 or t.shape != (B, O, C) or t.stride(2) != 1:
            raise RuntimeError("ISLA projections must be f32 (B, O, C) GPU tensors with unit channel stride")
    if wproj.stride() != bproj.stride():
        raise RuntimeError("weight / bias projections must share their strides")
    return wproj.stride(0), wproj.stride(1)


def norm_fwd_raw(x, sums, sq, count, stat_stride, spec, mask, wproj, bproj, op_dtype, want_f32=False, update_running=True):
    B, H, W, C = x.shape
    O = mask.shape[1] if mask is not None else 0
    if spec.mode == 0:
        _chk(mask, torch.float32)
        psb, pso = _proj_strides(wproj, bproj, B, O, C)
        assert mask.shape == (B, O, H, W)
    else:
        psb = pso = 0
    out_op = torch.empty((B, H, W, C), dtype=op_dtype, device=x.device)
    out_f = torch.empty((B, H, W, C), dtype=torch.float32, device=x.device) if want_f32 else None
    rm = rv = None
    if update_running and spec.training and spec.running is not None and not spec.instance:
        rm, rv = spec.running   # nn.BatchNorm2d's momentum update, fused into this launch
        _chk(rm, torch.float32), _chk(rv, torch.float32)
    _lib.call("l2i_norm_mod_fwd", x.data_ptr(), B, H * W, C, sums.data_ptr(), sq.data_ptr(), float(count), float(spec.eps),
              stat_stride, _p(mask), O, _p(wproj), _p(bproj), psb, pso, spec.mode, int(spec.relu), out_op.data_ptr(),
              _p(out_f), _code(op_dtype), _p(rm), _p(rv), float(spec.momentum), _stream())
    return out_op, out_f


def norm_bwd_raw(x, dy, sums, sq, count, stat_stride, spec, mask, wproj, bproj, add_to=None, need_mask_grad=True, sink=None,
                 emit_op=None):
    """Returns (dx, dwproj, dbproj, dmask). dy is overwritten with dxhat.
    emit_op: torch.bfloat16 to have the second pass also write the operand copy of dx (attached to dx: ops._sibling).
    sink: (GradSink, weight-projection column, bias-projection column) when wproj / bproj are slices of a grouped
    projection: their gradients are accumulated straight into the group's dY matrix (same strides)."""
    B, H, W, C = x.shape
    O = mask.shape[1] if mask is not None else 0
    dev = x.device
    G = sums.shape[0]
    s12 = _zeros((2, G, C), dev)   # one buffer: the data-parallel exchange reduces both halves in one message, in place
    s1, s2 = s12[0], s12[1]
    dw = db = dm = None
    psb = pso = 0
    if spec.mode == 0:
        psb, pso = _proj_strides(wproj, bproj, B, O, C)
        if sink is not None:
            dw, db = sink[0].slice(sink[1], C, B, O), sink[0].slice(sink[2], C, B, O)
            assert dw.stride() == wproj.stride()
        else:
            if (psb, pso) != (O * C, C):
                wproj, bproj = wproj.contiguous(), bproj.contiguous()
                psb, pso = O * C, C
            dw, db = _zeros(wproj.shape, dev), _zeros(bproj.shape, dev)
        dm = torch.empty_like(mask) if need_mask_grad else None   # (dmask_fresh: written, or cleared first, by the launch)
    elif spec.mode == 1:
        dw, db = _zeros(wproj.shape, dev), _zeros(bproj.shape, dev)
    keep = None
    # the stream's scratch: partial dW / dB rows of the <= 8-object kernel, the workgroups' rows of the per-channel totals and the channel
    # chunks' dmask rows -- stored, then added in a fixed order (no float atomics: round 6)
    part, npart = _lib.wgrad_scratch(dev)
    _lib.call("l2i_norm_mod_bwd_a", x.data_ptr(), dy.data_ptr(), B, H * W, C, sums.data_ptr(), sq.data_ptr(), float(count),
              float(spec.eps), stat_stride, _p(mask), O, _p(wproj), _p(bproj), psb, pso, spec.mode, int(spec.relu),
              dy.data_ptr(), s1.data_ptr(), s2.data_ptr(), _p(dw), _p(db), _p(dm), _p(keep),
              _ws(dev) if C <= 1024 else None, part, npart, 1, _stream())
    frozen = (not spec.training) and spec.running is not None and not spec.instance
    if frozen:
        raise RuntimeError("backward through eval-mode batch norm is not part of the hot path")
    if spec.sync is not None and not spec.instance:
        spec.sync(s1, s2, None)
    rpg = H * W if spec.instance else B * H * W
    dx = add_to if add_to is not None else dy
    dx_op = torch.empty(dx.shape, dtype=emit_op, device=dev) if emit_op is torch.bfloat16 else None
    _lib.call("l2i_norm_bwd_b", x.data_ptr(), dy.data_ptr(), sums.data_ptr(), sq.data_ptr(), s1.data_ptr(), s2.data_ptr(),
              dx.data_ptr(), B * H * W, C, rpg, float(count), float(spec.eps), int(add_to is not None), _p(dx_op), _stream())
    if dx_op is not None:
        _attach(dx, raw=dx_op)
    return dx, dw, db, dm


# ----------------------------------------------------------------------------- fused conv Function
def _atomics_split(B, Ho, Wo, n_out, kh, k_in, op_dtype):
    """True when the library may run this launch as split-K combined by f32 ATOMICS into a plain f32 result (small grids with a
    long reduction on the generic kernel): such a launch cannot write operand copies or gather statistics in its epilogue. The
    halo kernels (bf16, 3x3, >= 64 input channels, maps >= 8 wide) split by STORED partial tiles instead, and their reduce kernel
    carries the whole epilogue (csrc conv_split_reduce_kernel): no restriction there."""
    if ((B * Ho * Wo + 127) // 128) * ((n_out + 127) // 128) >= 192:
        return False
    return not (kh == 3 and op_dtype == torch.bfloat16 and k_in >= 64 and Wo >= 8 and STORED_SPLITS)   # (4-wide maps: the weight-stationary kernel)


ROI_GATHER = __import__("os").environ.get("L2I_ROI_GATHER", "1") != "0"   # (the library's switch of the same name)
HEAD_DX_OP = __import__("os").environ.get("L2I_HEAD_DX_OP", "1") != "0"   # the discriminator heads' backward writes the bf16 copy of dx (A/B switch)
# A/B switch: padded-channel bias gradients (mask heads 100 of 104, to-RGB 3 of 8) from the weight-gradient launch. CONTRACT: such a gradient
# reaches `bias.grad` in WeightArena.flush_grads (before the per-group callbacks), not during backward() -- like every weight gradient of this
# package (the spectral-norm backward runs there). Code that reads .grad straight after backward() must call net.arena.flush_grads() first; a pass
# that holds such gradients is never evicted from arena.pending, arena.drop_pending() drops them with the pass's weight gradients.
BIAS_SLOTS = __import__("os").environ.get("L2I_BIAS_SLOTS", "1") != "0"
STORED_SPLITS = __import__("os").environ.get("L2I_CONV_PART", "1") != "0"   # (the library's switch of the same name)


class GradJoin:
    """Joins the two gradient branches of a residual block's input inside the data-gradient launch: the shortcut
    branch (whose backward runs first: autograd orders ready nodes by creation, latest first) GIVES its dx, the conv1
    branch TAKES it as the residual input of its own data-gradient epilogue (or of the norm backward's second pass) --
    instead of autograd materialising both and adding them with one more pass. If the order ever differs, both
    branches simply return their gradients the ordinary way.
    A join between two BLOCKS that read one tensor (fused_conv join_in / join_out) has a taker that is not always part of the
    backward pass: with a loss over d_obj / d_app only, the trunk block whose shortcut launch would take the object path's
    gradient never runs. The convolution that PRODUCED the tensor is the taker of last resort (fused_conv join_src): its backward
    runs after every reader's, and adds a gradient that is still parked there to its dY -- one extra pass in that rare case,
    never a silently dropped gradient."""

    fold_ok = False   # the taker is a ReLU-prologue 3x3 conv that can fold the giver's (a 1x1 shortcut's) data gradient into its own launch

    def __init__(self):
        self.t, self.state = None, "open"

    def give(self, dx):
        if self.state == "open" and dx is not None:
            self.t, self.state = dx, "filled"
            return None
        return dx

    def take(self):
        if self.state == "filled":
            t, self.t, self.state = self.t, None, "done"
            return t
        self.state = "closed"
        return None

    def leftover(self):
        """The parked gradient nobody took (see the class docstring), or None."""
        if self.state == "filled":
            t, self.t, self.state = self.t, None, "done"
            return t
        return None


class FusedConvFn(Function):
    """[prologue] -> implicit-GEMM conv/linear (+bias, +res, up2 / pool2) with f32 streams on both
    sides. The operand tensor produced by the prologue never becomes an autograd edge, so its
    gradient stays f32.

    forward(x, res, bias, mask, wproj, bproj, holder, passctx, prologue, up2, pool2, nimg)
    nimg: optional 1-element int32 device tensor, the number of leading images that are live (ROI heads).
    """

    @staticmethod
    def forward(ctx, x, res, bias, mask, wproj, bproj, holder: GemmWeight, pc: PassCtx, pro, up2, pool2, nimg=None,
                emit=(), dx_raw=False, join=None, op_out=False, lazy_sc=False, join_in=None, join_out=None, join_src=None, f32_dead=False):
        opd = pc.arena.op_dtype
        _chk(x, opd if (pro.kind in ("op", "opraw") and not pc.arena.split) else torch.float32)
        B, H, W, C = x.shape
        assert C == holder.ci_p, (C, holder.ci_p, holder.kind)
        stats = None
        split = pc.arena.split
        if split:   # forward-only "bf16x3" mode: every operand is made from an f32 tensor as [hi | lo | hi] (no epilogue copies)
            if pro.kind == "norm":
                sums, sq, count, sstride = _norm_stats(x, pro)
                xf, _ = norm_fwd_raw(x, sums, sq, count, sstride, pro, mask, wproj, bproj, torch.float32)
                x_op = split_cast(xf)
            else:
                x_op = split_cast(x if x.dtype == torch.float32 else x.float(), relu=pro.kind in ("relu", "op"))
            emit = tuple(e for e in emit if e == "stats")
            assert not op_out and not lazy_sc and (res is None or getattr(res, "_l2i_lazy_sc", None) is None)
        elif pro.kind in ("op", "opraw"):   # x IS the operand: the ReLU'd result of the producing conv's epilogue (`op_out`),
            x_op = x                      # or a tensor produced in the operand dtype (ops.psp_expand) -- no f32 stream
        elif pro.kind == "norm":
            sums, sq, count, sstride = _norm_stats(x, pro)
            x_op, _ = norm_fwd_raw(x, sums, sq, count, sstride, pro, mask, wproj, bproj, opd)
            stats = (sums, sq, count, sstride)
        elif pro.kind == "relu":
            x_op = _sibling(x, "relu", opd)
            if x_op is None:
                _, x_op = cast_op(x, opd, raw=False, act=True)
        else:
            x_op = _sibling(x, "raw", opd)
            if x_op is None:
                x_op, _ = cast_op(x, opd, raw=True, act=False)
        bias_p = None
        if bias is not None:
            if bias.numel() == holder.co_p:
                bias_p = bias
            elif bias.dim() == 1 and bias.is_contiguous() and getattr(bias, "_l2i_slot", 0) >= holder.co_p:
                # a view over the parameter's zero-padded slot of the flat buffer (arena.FlatParams): no pad launches
                bias_p = torch.as_strided(bias.detach(), (holder.co_p,), (1,), bias.storage_offset())
            else:
                bias_p = torch.nn.functional.pad(bias, (0, holder.co_p - bias.numel()))
        Ho, Wo = (2 * H, 2 * W) if up2 else (H, W)
        flops = 2.0 * B * Ho * Wo * holder.co * holder.ci * holder.kh * holder.kh  # algorithmic (unpadded) work
        # `emit`: operand copies of the RESULT written by this launch's epilogue for the layers that read it next
        # ("relu": ReLU'd copy for a pre-activation conv, "raw": plain copy for a shortcut conv) -- they ride on the
        # result tensor (`_sibling`) and replace separate cast launches over the f32 stream.
        if emit and _atomics_split(B, Ho, Wo, holder.co_p, holder.kh, holder.ci_p, opd):
            emit = ()   # small grids run split-K (partial sums combined by atomics): no epilogue copies there
        # `lazy_sc`: this is a residual block's 1x1 shortcut and its ONLY reader is the `res` input of the block's second 3x3
        # fused_conv: nothing is launched here; the operands ride on an unwritten placeholder of the result's shape and the
        # consumer folds the shortcut into its own launch (conv_raw sc=, csrc conv_sc_tail). Backward is unchanged: the
        # consumer hands dY through as the residual's gradient and this node computes the 1x1's gradients from it.
        sc = getattr(res, "_l2i_lazy_sc", None) if res is not None else None
        if sc is not None and (sc["ver"] != res._version or sc["pool2"] != bool(pool2) or sc["nimg"] is not nimg):
            raise RuntimeError("a lazy shortcut can only be the residual of the conv it was made for")
        ctx.sc_lazy = sc         # the shortcut node's own record (its backward looks at `wgrad_done`)
        if sc is not None:
            sc = dict(sc, out=res)   # (the placeholder itself: where the library puts the shortcut when it cannot fold it)
        if lazy_sc:
            assert holder.kh == 1 and res is None and not emit and not op_out and pro.kind not in ("norm",)
            Hq, Wq = (Ho // 2, Wo // 2) if pool2 else (Ho, Wo)
            out = torch.empty((B, Hq, Wq, holder.co_p), dtype=torch.float32, device=x.device)
            out._l2i_lazy_sc = dict(x_op=x_op, wpack=pc.fwd_span(holder), wpack_b=pc.fwd_span_b(holder), kpad=holder.kpad, bias=bias_p, up2=bool(up2),
                                    pool2=bool(pool2), nimg=nimg, flops=flops, ver=out._version, holder=holder, wgrad_done=False)
            ctx.lazy = out._l2i_lazy_sc   # (shared with the consumer: its backward may compute this node's weight gradient, see below)
        else:
        # `op_out`: the ONLY reader of the result is a pre-activation conv -- the epilogue writes relu(result) in the operand
        # dtype and nothing else; that tensor is the autograd edge (its gradient arrives, and is used, in the operand dtype)
            # `f32_dead` (L2I_F32_DEAD=1, opt-in): every reader of this result takes the emitted operand copies (a D block's result read by a
            # block with a learnable shortcut: conv1 reads "relu", the shortcut "raw") -- the f32 stream is then not written at all; the tensor
            # that stays the autograd edge is a placeholder nothing reads (tests/test_cpu_dryrun.py proves that for every traced configuration)
            phantom = (F32_DEAD and f32_dead and not op_out and opd == torch.bfloat16 and "relu" in emit and "raw" in emit and "stats" not in emit)
            out, o_relu, o_raw = conv_raw(x_op, pc.fwd_span(holder), holder.kpad, holder.co_p, holder.kh, bias=bias_p,
                                          res=None if sc is not None else res, sc=sc, wpack_b=pc.fwd_span_b(holder),
                                          up2=up2, pool2=pool2, alpha=0.25 if pool2 else 1.0, flops=flops, nimg=nimg,
                                          want_f32=not op_out and not phantom, want_op=op_out or "relu" in emit, relu_op=True,
                                          want_raw="raw" in emit, stats="stats" in emit and not op_out)
            if phantom:
                out = torch.empty(o_raw.shape, dtype=torch.float32, device=x.device)
            if op_out:
                out = o_relu
            elif emit and (o_raw is not None or o_relu is not None):
                _attach(out, raw=o_raw, relu=o_relu)
        ctx.op_out = op_out
        ctx.flops, ctx.nimg, ctx.dx_raw, ctx.join = flops, nimg, dx_raw, join
        ctx.join_in, ctx.join_out, ctx.join_src = join_in, join_out, join_src
        sw, sb = getattr(wproj, "_l2i_sink", None), getattr(bproj, "_l2i_sink", None)
        ctx.sink = (sw[0], sw[1], sb[1]) if sw is not None and sb is not None and sw[0] is sb[0] else None
        ctx.holder, ctx.pc, ctx.pro, ctx.up2, ctx.pool2, ctx.stats = holder, pc, pro, up2, pool2, stats
        ctx.has_bias, ctx.has_res = bias is not None, res is not None
        keep_x = x if pro.kind == "norm" else None
        ctx.save_for_backward(keep_x, x_op, mask, wproj, bproj)
        return out

    @staticmethod
    def backward(ctx, dy):
        h, pc, pro = ctx.holder, ctx.pc, ctx.pro
        x, x_op, mask, wproj, bproj = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.join_src is not None:   # a reader of this result parked its gradient for a launch that never ran (GradJoin)
            parked = ctx.join_src.leftover()
            if parked is not None:
                dy = dy + parked.to(dy.dtype)
        opd = pc.arena.op_dtype
        need_x = ctx.needs_input_grad[0]
        need_mod = pro.kind == "norm" and pro.mode in (0, 1)
        alpha = 0.25 if ctx.pool2 else 1.0
        d_bias = None
        # the operand copy of dY: written by the data-gradient launch that produced dY (`dx_raw`), or by another layer
        # that received the same dY (a block's conv2 and its shortcut), else cast here -- and left on dY for the others
        dy_op = dy if ctx.op_out else _sibling(dy, "raw", opd)
        dbias = None
        lazy_ = getattr(ctx, "lazy", None)
        # (a lazy shortcut whose consumer -- conv2 -- already computed its weight AND bias gradient with its own launch: nothing to do here)
        if pc.need_wgrad and ctx.has_bias and not (lazy_ is not None and lazy_["wgrad_done"]):
            bg = h.bias.grad
            direct = bg is not None and h.co == h.co_p and bg.is_contiguous() and bg.dtype == torch.float32 and getattr(h, "uses", 1) == 1
            if direct:   # the weight-gradient launch sums the bias gradient from the dY tiles it stages, straight into the
                dbias = bg   # flat gradient buffer: no pass over dY at all when its operand copy already exists
            elif (BIAS_SLOTS and bg is not None and getattr(h, "bias_scr_off", None) is not None and not pc.dual and dy_op is not None
                  and bg.is_contiguous() and bg.dtype == torch.float32 and not (lazy_ is not None and lazy_["wgrad_done"])):
                # padded channel count (the mask heads' 100 of 104, the to-RGB layer's 3 of 8): the same sum into a Co_p-wide slot
                # of the pass's accumulator; flush_grads adds its first Co values to the bias gradient -- instead of a pass over dY
                dbias = pc.bias_slot(h, bg)
            elif ctx.op_out:
                d_bias = dy.float().sum(dim=(0, 1, 2))[:h.co]
            else:        # (padded channel counts: bias gradient and the dY operand cast in one pass over dY)
                st = channel_stats(dy.view(-1, dy.shape[-1]), want_sq=False, cast_to=opd if dy_op is None else None)
                if dy_op is None:
                    dy_op = st[2].view(dy.shape)
                d_bias = st[0][0][:h.co]
        if dy_op is None:
            dy_op, _ = cast_op(dy, opd, raw=True, act=False)
        if not ctx.op_out and _sibling(dy, "raw", opd) is None:
            _attach(dy, raw=dy_op)
        lazy = getattr(ctx, "lazy", None)
        if pc.need_wgrad and lazy is not None and lazy["wgrad_done"]:
            pass   # a lazy shortcut whose consumer (conv2) already computed its weight and bias gradient with its own launch
        elif pc.need_wgrad:
            # conv2 of a block whose shortcut was handed over: the shortcut sees the same dY, so its weight gradient becomes
            # extra column tiles of this launch, its bias gradient the same sum
            scw = None
            sl = getattr(ctx, "sc_lazy", None)
            if sl is not None and SC_WGRAD and sl["holder"].kh == 1:
                hs = sl["holder"]
                bgs = hs.bias.grad if hs.bias is not None else None
                direct_s = hs.bias is None or (bgs is not None and hs.co == hs.co_p and bgs.is_contiguous() and bgs.dtype == torch.float32)
                if direct_s and hs.bias is not None and getattr(hs, "uses", 1) > 1:   # a weight applied several times per forward: through the pass's slot
                    bgs = pc.bias_slot(hs, bgs) if (BIAS_SLOTS and getattr(hs, "bias_scr_off", None) is not None and not pc.dual) else None
                    direct_s = bgs is not None
                if direct_s and hs.co_p == h.co_p:
                    scw = dict(x_op=sl["x_op"], dw=pc.dw_span(hs), dw_b=pc.dw_span_b(hs), ldw=hs.kp, dbias=bgs, flops=sl["flops"], up2=sl["up2"])
                    sl["wgrad_done"] = True
            # stored, not accumulated, when this is the slice's first launch of the pass (a second backward over the same forward adds)
            ow = WGRAD_OVERWRITE and h.kind == "conv" and not pc.was_written(h) and (scw is None or not pc.was_written(sl["holder"]))
            if WGRAD_OVERWRITE and h.kind == "conv" and not ow:   # (adding: a slice nothing has stored yet starts from zero)
                pc.dw_acc(h)
                if scw is not None:
                    pc.dw_acc(sl["holder"])
            wgrad_side(x_op, dy_op, pc.dw_span(h), h.kp, h.co_p, h.kh, up2=ctx.up2, pool2=ctx.pool2, alpha=alpha,
                       flops=ctx.flops, nimg=ctx.nimg, dbias=dbias, sc=scw, dw_b=pc.dw_span_b(h), overwrite=ow)
            pc.mark_written(h)
            if scw is not None:
                pc.mark_written(sl["holder"])
        dx = d_mask = d_w = d_b = None
        if need_x or need_mod:
            # data gradient: same kernel on the flipped pack; upsample <-> pool swap roles
            relu_mask = x_op if pro.kind in ("relu", "op") else None
            Bq, Hq, Wq = dy.shape[0], dy.shape[1] << int(ctx.pool2), dy.shape[2] << int(ctx.pool2)
            small = _atomics_split(Bq, Hq, Wq, h.ci_p, h.kh, h.co_p, opd)   # split-K grid combined by atomics: no epilogue copies
            op_in = pro.kind in ("op", "opraw")   # the input edge is an operand tensor: its gradient is the operand copy alone
            emit_raw = op_in or (ctx.dx_raw and pro.kind != "norm" and not small)   # (the norm backward rewrites dxo: its copy would be stale)
            joined = ctx.join[0].take() if ctx.join is not None and ctx.join[1] == "take" and need_x else None
            if joined is None and ctx.join_in is not None and need_x and pro.kind != "norm":
                joined = ctx.join_in.take()   # the complete gradient another reader of x left for this launch's free residual slot
            # (a) this is a block's 1x1 shortcut and the block's conv1 can fold its data gradient (GradJoin.fold_ok): hand over the
            #     OPERANDS instead of launching -- conv1's launch computes mask(alpha1 W1^T dh) + alpha W_sc^T dy (+ this node's residual)
            if (DGRAD_FOLD and ctx.join is not None and ctx.join[1] == "give" and ctx.join[0].fold_ok and ctx.join[0].state == "open" and need_x
                    and h.kh == 1 and pro.kind == "cast" and opd == torch.bfloat16 and not pc.arena.split and pc.dgrad_span_b(h) is None
                    and not ctx.up2 and h.co_p % 64 == 0 and not op_in and ctx.join_out is None):
                ctx.join[0].give(dict(x_op=dy_op, wpack=pc.dgrad_span(h), kpad=h.kpad_d, up2=bool(ctx.pool2), alpha=alpha, flops=ctx.flops,
                                      res=joined, mask_first=True, nimg=ctx.nimg))
                d_res = dy if ctx.has_res else None
                return None, d_res, d_bias, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None
            # (b) the taker's side of (a)
            sc_fold = None
            if isinstance(joined, dict):
                sc_fold, joined = joined, joined["res"]
                if relu_mask is None or h.kh != 3 or ctx.up2 or ctx.pool2 or op_in or sc_fold["nimg"] is not ctx.nimg or pro.kind == "norm":
                    raise RuntimeError("GradJoin.fold_ok was set on a join whose taker cannot fold the shortcut's data gradient")
                sc_fold = dict(sc_fold, bias=None, out=torch.empty((Bq, Hq, Wq, h.ci_p), dtype=torch.float32, device=dy.device))
            dxo, _, dx_op = conv_raw(dy_op, pc.dgrad_span(h), h.kpad_d, h.ci_p, h.kh, relu_mask=relu_mask, up2=ctx.pool2,
                                     pool2=ctx.up2, alpha=alpha, flops=ctx.flops, nimg=ctx.nimg, want_raw=emit_raw,
                                     want_f32=not op_in, res=joined if pro.kind != "norm" else None, wpack_b=pc.dgrad_span_b(h), sc=sc_fold)
            if op_in:
                dxo = dx_op
            elif emit_raw:
                _attach(dxo, raw=dx_op)
            if pro.kind == "norm":
                sums, sq, count, sstride = ctx.stats
                dx, d_w, d_b, d_mask = norm_bwd_raw(x, dxo, sums, sq, count, sstride, pro, mask, wproj, bproj, add_to=joined,
                                                    need_mask_grad=mask is not None and ctx.needs_input_grad[3], sink=ctx.sink,
                                                    emit_op=opd if ctx.dx_raw else None)
            else:
                dx = dxo
            if ctx.join is not None and ctx.join[1] == "give":
                dx = ctx.join[0].give(dx)
            if ctx.join_out is not None and dx is not None:
                dx = ctx.join_out.give(dx)
        d_res = dy if ctx.has_res else None
        if d_res is not None and ctx.join is not None and ctx.join[1] == "give_res":
            d_res = ctx.join[0].give(d_res)
        return dx, d_res, d_bias, d_mask, d_w, d_b, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None


def _attach(t, raw=None, relu=None):
    """Let operand-dtype copies of stream `t` ride on the tensor object. The tensor's version counter is recorded:
    autograd accumulates gradients IN PLACE when it can (a block input read by conv1 and by an identity shortcut), and a
    copy made before such an accumulation must not be used after it."""
    t._l2i_ops = {"raw": raw, "relu": relu, "ver": t._version}


def _sibling(t, kind, dtype):
    """Operand-dtype copy of stream `t` made by a producer's epilogue or by `precast`, or None."""
    d = getattr(t, "_l2i_ops", None)
    if not d or d.get("ver") != t._version:
        return None
    v = d.get(kind)
    return v if v is not None and v.dtype == dtype and v.shape == t.shape else None


def precast(x, op_dtype):
    """One launch producing BOTH operand copies of a stream (raw and ReLU'd) for a block whose two branches read it
    through different prologues (pre-activation conv + shortcut conv); the following fused_conv calls pick them up."""
    have_raw, have_act = _sibling(x, "raw", op_dtype), _sibling(x, "relu", op_dtype)
    if have_raw is None or have_act is None:
        raw, act = cast_op(x, op_dtype, raw=have_raw is None, act=have_act is None)
        _attach(x, raw=have_raw if have_raw is not None else raw, relu=have_act if have_act is not None else act)
    return x


def fused_conv(x, holder, pc, *, prologue=None, res=None, mask=None, wproj=None, bproj=None, up2=False, pool2=False, nimg=None,
               emit=(), dx_raw=False, join=None, relu_op_out=False, lazy_sc=False, join_in=None, join_out=None, join_src=None, f32_dead=False):
    """f32_dead: the caller guarantees that every reader of the result takes the emitted operand copies (see FusedConvFn.forward; opt-in).
    join_src: the GradJoin(s) of the two readers of THIS conv's result (a tuple is accepted): see GradJoin.leftover.
    join_in / join_out: a GradJoin shared with ANOTHER reader of x (a tensor read by two blocks): the reader whose backward runs
    first (the one created later) leaves its complete dx there (join_out), the other one's launch with a free residual slot (a
    block's 1x1 shortcut) adds it in its data-gradient epilogue (join_in) -- no autograd accumulation pass over the two gradients.
    emit: operand copies of the result to write in the epilogue ("relu", "raw") for the layers that read it next.
    dx_raw: x is read by this layer ONLY and was produced by another fused_conv -- the data-gradient launch then also
    writes the operand copy of dx that the producer's backward needs (no separate cast pass over dx).
    join: (GradJoin, "give" | "take" | "give_res") -- see GradJoin.
    relu_op_out: the result is read by ONE pre-activation (ReLU-prologue) fused_conv and by nothing else: where the grid
    allows epilogue copies the launch then writes only relu(result) in the operand dtype and returns that tensor (the f32
    stream of the result and of its gradient are never written); otherwise this is emit=("relu",)."""
    pro = prologue if prologue is not None else _CAST
    if pc.arena.split:
        if torch.is_grad_enabled():
            raise RuntimeError('the split-operand ("bf16x3") precision mode is forward-only: run it under torch.no_grad()')
        relu_op_out, lazy_sc = False, False
    if getattr(x, "_l2i_relu_op", False):
        if pro is not RELU:
            raise RuntimeError("a relu_op_out result can only feed a ReLU-prologue fused_conv")
        pro = _OP
    elif getattr(x, "_l2i_raw_op", False):
        if pro is not _CAST:
            raise RuntimeError("an operand-dtype tensor can only feed a plain (cast-prologue) fused_conv")
        pro = _OPRAW
    if relu_op_out:
        B, H, W, _ = x.shape
        Ho, Wo = (2 * H, 2 * W) if up2 else (H, W)
        if res is not None or pool2 or _atomics_split(B, Ho, Wo, holder.co_p, holder.kh, holder.ci_p, pc.arena.op_dtype) or not OP_EDGES:
            relu_op_out, emit = False, tuple(emit) + ("relu",)
    out = FusedConvFn.apply(x, res, holder.bias, mask, wproj, bproj, holder, pc, pro, up2, pool2, nimg, tuple(emit), dx_raw, join,
                            relu_op_out, bool(lazy_sc) and SC_FOLD, join_in, join_out, join_src, bool(f32_dead))
    if relu_op_out:
        out._l2i_relu_op = True
    return out


class _Simple(Prologue):
    def __init__(self, kind):
        self.kind = kind


_CAST, RELU, _OP, _OPRAW = _Simple("cast"), _Simple("relu"), _Simple("op"), _Simple("opraw")
WGRAD_OVERWRITE = __import__("os").environ.get("L2I_WGRAD_OVERWRITE", "1") != "0"   # weight-gradient launches store into their (freshly zeroed, per-pass) dW slices (A/B switch)
SC_WGRAD = __import__("os").environ.get("L2I_SC_WGRAD_PY", "1") != "0"   # conv2's weight-gradient launch also computes the handed-over shortcut's (A/B switch)
SC_FOLD = __import__("os").environ.get("L2I_SC_LAZY", "1") != "0"   # blocks hand their 1x1 shortcut to conv2's launch (A/B switch; L2I_SC_FOLD=0 keeps the hand-over but un-folds in the library)
DGRAD_FOLD = __import__("os").environ.get("L2I_DGRAD_FOLD", "1") != "0"   # a D block's 1x1 shortcut data gradient rides on conv1's data-gradient launch (A/B switch)
OP_EDGES = __import__("os").environ.get("L2I_OP_EDGES", "1") != "0"   # operand-dtype autograd edges inside D blocks (A/B switch)
F32_DEAD = __import__("os").environ.get("L2I_F32_DEAD", "0") == "1"   # opt-in: f32 result streams that no reader takes are not written (CPU-verified only)


class NormActFn(Function):
    """Stand-alone normalise (+modulate) (+ReLU) with an f32 result, for the places where something
    other than a convolution follows (bilinear upsampling in the mask regressor, Dropout2d in PSP)."""

    @staticmethod
    def forward(ctx, x, mask, wproj, bproj, spec, emit_op=None):
        _chk(x, torch.float32)
        sums, sq, count, sstride = _norm_stats(x, spec)
        out, _ = norm_fwd_raw(x, sums, sq, count, sstride, spec, mask, wproj, bproj, torch.float32)
        ctx.spec, ctx.stats, ctx.emit_op = spec, (sums, sq, count, sstride), emit_op
        ctx.save_for_backward(x, mask, wproj, bproj)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, mask, wproj, bproj = ctx.saved_tensors
        sums, sq, count, sstride = ctx.stats
        dy = dy.contiguous() if getattr(dy, "_l2i_owned", False) else dy.contiguous().clone()   # (norm_bwd_raw overwrites dy: a gradient nobody else holds needs no copy)
        dx, d_w, d_b, d_mask = norm_bwd_raw(x, dy, sums, sq, count, sstride, ctx.spec, mask, wproj, bproj,
                                            need_mask_grad=mask is not None and ctx.needs_input_grad[1], emit_op=ctx.emit_op)
        return dx, d_mask, d_w, d_b, None, None


def norm_act(x, spec, wproj=None, bproj=None, mask=None, emit_op=None):
    """emit_op: torch.bfloat16 when x is the result of a fused_conv read by this layer alone: the backward's second pass then also
    writes the operand copy of dx that conv's backward needs (as fused_conv(dx_raw=True) does for a norm prologue)."""
    return NormActFn.apply(x, mask, wproj, bproj, spec, emit_op)


class GradSink:
    """The dY matrix (rows, N) of one grouped projection, allocated (zero) when the first gradient slice is asked for."""

    def __init__(self, rows, n, device):
        self.rows, self.n, self.device, self.buf = rows, n, device, None
        self.delivered = set()   # column offsets whose gradient was accumulated here directly

    def slice(self, col, c, B, O):
        if self.buf is None:
            self.buf = torch.zeros((self.rows, self.n), dtype=torch.float32, device=self.device)
        self.delivered.add(col)
        return self.buf.view(B, O, self.n)[:, :, col:col + c]


class GroupedLinearFn(Function):
    """All layers of a GemmGroup (arena.py) applied to the same input in one GEMM: y = x Wcat^T + bcat, returned as one
    (rows, C_i) slice per member. Backward: one bias-gradient / cast pass, one weight-gradient launch (stacked dW),
    one data-gradient launch (K = sum C_i) -- instead of one of each per member and a chain of additions.
    The members' output gradients are expected in the group's GradSink (ISLA backward accumulates them there)."""

    @staticmethod
    def forward(ctx, x, group, pc, sink):
        _chk(x, torch.float32)
        ctx.set_materialize_grads(False)
        opd = pc.arena.op_dtype
        rows = x.shape[0]
        if pc.arena.split:
            x_op = split_cast(x)
        else:
            x_op = _sibling(x, "raw", opd)
            if x_op is None:
                x_op, _ = cast_op(x, opd, raw=True, act=False)
        flat = pc.arena.flat
        b0 = flat.offset_of(group.members[0].bias)
        bias = flat.data[b0:b0 + group.n_total]
        flops = 2.0 * rows * group.n_total * group.ci
        y, _, _ = conv_raw(x_op, pc.group_fwd_pack(group), group.kpad, group.n_total, 1, bias=bias, flops=flops)
        ctx.group, ctx.pc, ctx.sink, ctx.flops, ctx.b0 = group, pc, sink, flops, b0
        ctx.save_for_backward(x_op)
        y2 = y.view(rows, group.n_total)
        return tuple(y2[:, o:o + m.co_p] for o, m in zip(group.offsets, group.members))

    @staticmethod
    def backward(ctx, *grads):
        g, pc, sink = ctx.group, ctx.pc, ctx.sink
        (x_op,) = ctx.saved_tensors
        opd = pc.arena.op_dtype
        rows = x_op.shape[0]
        if sink.buf is None:
            sink.buf = torch.zeros((rows, g.n_total), dtype=torch.float32, device=x_op.device)
        dy = sink.buf
        for o, m, gr in zip(g.offsets, g.members, grads):   # anything that did not arrive through the sink
            if gr is not None and o not in sink.delivered:
                dy[:, o:o + m.co_p] += gr.reshape(rows, m.co_p)
        dx = None
        if pc.need_wgrad:
            bg = pc.arena.flat.grad[ctx.b0:ctx.b0 + g.n_total]
            _, _, dy_op = channel_stats(dy, want_sq=False, cast_to=opd, accumulate_into=bg)
            dy4 = dy_op.view(rows, 1, 1, g.n_total)
            wgrad_side(x_op, dy4, pc.group_dw_slice(g), g.kp, g.n_total, 1, flops=ctx.flops)   # (accumulates: the group's slice is cleared by PassCtx.dw)
        else:
            dy4, _ = cast_op(dy.view(rows, 1, 1, g.n_total), opd, raw=True, act=False)
        if ctx.needs_input_grad[0]:
            dx, _, _ = conv_raw(dy4, pc.group_dgrad_pack(g), g.kpad_d, g.ci_p, 1, flops=ctx.flops)
        sink.buf = None
        sink.delivered.clear()
        return dx, None, None, None


def grouped_linear(x, group, pc):
    """-> list of (rows, C_i) outputs, one per group member, each carrying its gradient-sink coordinates."""
    if pc.arena.split and torch.is_grad_enabled():
        raise RuntimeError('the split-operand ("bf16x3") precision mode is forward-only: run it under torch.no_grad()')
    sink = GradSink(x.shape[0], group.n_total, x.device)
    outs = GroupedLinearFn.apply(x, group, pc, sink)
    for o, t in zip(group.offsets, outs):
        t._l2i_sink = (sink, o)
    return list(outs)


class ArenaWeightFn(Function):
    """Expose the normalised weight Wbar[:Co,:Ci] of a linear/embedding holder to glue code; the
    gradient flowing back is added to the pass's dWbar accumulator (spectral-norm backward happens
    in WeightArena.flush_grads)."""

    @staticmethod
    def forward(ctx, w_param, holder: GemmWeight, pc: PassCtx):
        ctx.holder, ctx.pc = holder, pc
        assert holder.kh == 1
        return pc.fwd_pack(holder).view(holder.npad, holder.kpad)[:holder.co, :holder.ci].float()

    @staticmethod
    def backward(ctx, g):
        h, pc = ctx.holder, ctx.pc
        if pc.need_wgrad:
            pc.dw_acc(h).view(h.co_p, h.kp)[:h.co, :h.ci].add_(g)
        return None, None, None


def _wbar_f32(holder, pc):
    """W / sigma of a layer in f32 straight from the parameter and the pass's sigma: what the heads read in the forward-only
    split-operand mode, whose packs hold hi | hi | lo blocks (no gradient: that mode runs under no_grad)."""
    w = holder.w.detach().reshape(holder.co, -1)
    return w / pc.sigma(holder) if holder.sn else w


def arena_weight(holder, pc):
    if pc.arena.split:
        return _wbar_f32(holder, pc)
    return ArenaWeightFn.apply(holder.w, holder, pc)


class SplitHalvesFn(Function):
    """(2n, ...) -> (x[:n], x[n:]) as two autograd edges whose gradients come back as ONE concatenation (plain slicing
    would materialise two zero-filled full-size gradients and add them). The per-pass heads of a dual discriminator pass
    (arena.DualPass) read their half of the trunk's rows through this."""

    @staticmethod
    def forward(ctx, x):
        n = x.shape[0] // 2
        ctx.meta = (x.shape, x.dtype, x.device)
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        shape, dtype, dev = ctx.meta
        n = shape[0] // 2
        if ga is None and gb is None:
            return None
        half = (n,) + tuple(shape[1:])
        ga = torch.zeros(half, dtype=dtype, device=dev) if ga is None else ga
        gb = torch.zeros(half, dtype=dtype, device=dev) if gb is None else gb
        return torch.cat((ga, gb))


def split_halves(x):
    assert x.shape[0] % 2 == 0
    return SplitHalvesFn.apply(x)


# ----------------------------------------------------------------------------- ROIAlign
class RoiAlignFn(Function):
    @staticmethod
    def forward(ctx, feat_s, feat_l, rois, valid, P, scale_s, scale_l, thr, sampling, join_src=None, emit_op=None):
        _chk(feat_s, torch.float32)
        ctx.join_src, ctx.emit_op = join_src, emit_op
        B, Hs, Ws, C = feat_s.shape
        Hl = Wl = 0
        if feat_l is not None:
            _chk(feat_l, torch.float32)
            _, Hl, Wl, _ = feat_l.shape
        rois = _chk(rois.contiguous(), torch.float32)
        R = rois.shape[0]
        out = torch.empty((R, P, P, C), dtype=torch.float32, device=feat_s.device)
        raw = relu = None
        if emit_op is torch.bfloat16:   # both operand copies of the ROI features from this launch (ops.precast's job otherwise)
            raw, relu = torch.empty_like(out, dtype=emit_op), torch.empty_like(out, dtype=emit_op)
        _lib.call("l2i_roi_align_fwd", feat_s.data_ptr(), _p(feat_l), rois.data_ptr(), _p(valid), out.data_ptr(), R, C, P,
                  Hs, Ws, float(scale_s), Hl, Wl, float(scale_l), float(thr), sampling, _p(raw), _p(relu), _stream())
        if raw is not None:
            _attach(out, raw=raw, relu=relu)
        ctx.save_for_backward(rois, valid)
        ctx.cfg = (P, scale_s, scale_l, thr, sampling, feat_s.shape, None if feat_l is None else feat_l.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        rois, valid = ctx.saved_tensors
        P, scale_s, scale_l, thr, sampling, shp_s, shp_l = ctx.cfg
        g = g.contiguous()
        if ctx.join_src is not None:   # a reader of the ROI features parked its gradient for a launch that never ran (GradJoin.leftover)
            parked = ctx.join_src.leftover()
            if parked is not None:
                g = g + parked
        # fresh maps: the library writes every pixel (gather form) or clears them itself before scattering
        ds = torch.empty(shp_s, dtype=torch.float32, device=g.device)
        dl = torch.empty(shp_l, dtype=torch.float32, device=g.device) if shp_l is not None else None
        Hl, Wl = (shp_l[1], shp_l[2]) if shp_l is not None else (0, 0)
        # the bf16 copies of both maps from the same stores (what block_obj4's conv2 backward reads), where the gather form runs
        emit = (ctx.emit_op is torch.bfloat16 and P == 8 and shp_s[3] % 32 == 0 and shp_s[2] <= 32 and Wl <= 32 and rois.shape[0] <= 1024
                and ROI_GATHER)   # (the library's conditions for the gather form; it refuses the copies otherwise)
        ds_op = torch.empty(shp_s, dtype=torch.bfloat16, device=g.device) if emit else None
        dl_op = torch.empty(shp_l, dtype=torch.bfloat16, device=g.device) if (emit and shp_l is not None) else None
        _lib.call("l2i_roi_align_bwd", rois.data_ptr(), _p(valid), g.data_ptr(), ds.data_ptr(), _p(dl), rois.shape[0],
                  shp_s[3], P, shp_s[1], shp_s[2], float(scale_s), Hl, Wl, float(scale_l), float(thr), sampling, shp_s[0], 1,
                  _p(ds_op), _p(dl_op), _stream())
        if ds_op is not None:
            _attach(ds, raw=ds_op)
        if dl_op is not None:
            _attach(dl, raw=dl_op)
        return ds, dl, None, None, None, None, None, None, None, None, None


def roi_align(feat_s, feat_l, rois, valid, P=8, scale_s=0.25, scale_l=0.125, thr=64.0, sampling=0, join_src=None, emit_op=None):
    """join_src: the GradJoin of the result's two readers (see GradJoin: the producer is the taker of last resort).
    emit_op: torch.bfloat16 to have the launch also write the raw and ReLU'd operand copies of the result (attached to it)."""
    return RoiAlignFn.apply(feat_s, feat_l, rois, valid, P, scale_s, scale_l, thr, sampling, join_src, emit_op)


# ----------------------------------------------------------------------------- attention core
class BoxAttentionFn(Function):
    """q, k, v: (B, O, Dp) f32 views with unit column stride whose first D columns are read -- contiguous tensors or column
    slices of ONE grouped projection result (ops.grouped_linear: same row stride); in the latter case the backward writes
    dq / dk / dv straight into the group's gradient sink."""

    @staticmethod
    def forward(ctx, q, k, v, geo, keyvalid, scale, D):
        for t in (q, k, v):
            if not t.is_cuda or t.dtype != torch.float32 or t.stride(-1) != 1 or t.stride() != q.stride() or t.shape != q.shape:
                raise RuntimeError("box_attention: q, k, v must be f32 GPU tensors of one shape and stride with unit column stride")
        B, O, _ = q.shape
        ld = q.stride(1)
        assert q.stride(0) == O * ld
        geo_c = None if geo is None else _chk(geo.contiguous(), torch.float32)
        out = torch.empty((B, O, D), dtype=torch.float32, device=q.device)
        prob = torch.empty((B, O, O), dtype=torch.float32, device=q.device)
        _lib.call("l2i_box_attention_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), _p(geo_c), _p(keyvalid), out.data_ptr(),
                  prob.data_ptr(), B, O, D, ld, float(scale), _stream())
        ctx.save_for_backward(q, k, v, geo_c, prob)
        ctx.scale, ctx.D = scale, D
        ctx.sinks = tuple(getattr(t, "_l2i_sink", None) for t in (q, k, v))
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, geo, prob = ctx.saved_tensors
        B, O, Dp = q.shape
        D = ctx.D
        g = g.contiguous()
        sk = ctx.sinks
        if all(s_ is not None for s_ in sk) and sk[0][0] is sk[1][0] is sk[2][0]:
            dq, dk, dv = (s_[0].slice(s_[1], Dp, B, O) for s_ in sk)   # (pad columns D..Dp stay at the sink's zeros)
        else:
            dq, dk, dv = (torch.zeros((B, O, Dp), dtype=torch.float32, device=g.device) if Dp != D else
                          torch.empty((B, O, D), dtype=torch.float32, device=g.device) for _ in range(3))
        dgeo = torch.empty_like(geo) if geo is not None else None
        _lib.call("l2i_box_attention_bwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), _p(geo), prob.data_ptr(), g.data_ptr(),
                  dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), _p(dgeo), B, O, D, q.stride(1), dq.stride(1), float(ctx.scale), _stream())
        return dq, dk, dv, dgeo, None, None, None


def box_attention(q, k, v, geo, keyvalid, scale, D=None):
    return BoxAttentionFn.apply(q, k, v, geo, keyvalid, scale, q.shape[-1] if D is None else D)


# ----------------------------------------------------------------------------- losses
class HingeFn(Function):
    """mode 0: mean relu(1-x) | 1: mean relu(1+x) | 2: -mean x, over rows with valid != 0."""

    @staticmethod
    def forward(ctx, x, valid, mode, weight, count):
        xc = _chk(x.contiguous().view(-1), torch.float32)
        loss = torch.zeros((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(xc)
        _lib.call("l2i_hinge_fwd_bwd", xc.data_ptr(), _p(valid), xc.numel(), mode, float(weight), _p(count),
                  loss.data_ptr(), grad.data_ptr(), _stream())
        ctx.save_for_backward(grad)
        ctx.shape = x.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).view(ctx.shape), None, None, None, None


def hinge(x, valid, mode, weight=1.0, count=None):
    """count: optional 1-element f32 device tensor = global number of rows (data parallel)."""
    return HingeFn.apply(x, valid, mode, weight, count)


class PadParamFn(Function):
    """A 1-D parameter seen with its channel count padded to n: a VIEW of the parameter's slot in the flat buffer
    (arena.FlatParams keeps the pad behind every parameter at zero), not a copy -- no pad launch forward, a slice view
    backward."""

    @staticmethod
    def forward(ctx, p, n):
        ctx.c = p.numel()
        return torch.as_strided(p.detach(), (n,), (1,), p.storage_offset())

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.c], None


def pad_param(p, n):
    if p.numel() == n:
        return p
    if p.dim() == 1 and p.is_contiguous() and getattr(p, "_l2i_slot", 0) >= n:
        return PadParamFn.apply(p, n)
    return torch.nn.functional.pad(p, (0, n - p.numel()))


class HingeSumFn(Function):
    """Sum of hinge terms over several discriminator outputs (ops.hinge semantics per term) accumulated into ONE scalar:
    no per-term zero fills, additions or gradient scalings."""

    @staticmethod
    def forward(ctx, mode, n, *args):
        xs, valids, weights, counts = args[:n], args[n:2 * n], args[2 * n:3 * n], args[3 * n:4 * n]
        loss = _zeros((1,), xs[0].device)
        grads = []
        for x, valid, wgt, cnt in zip(xs, valids, weights, counts):
            xc = _chk(x.contiguous().view(-1), torch.float32)
            g = torch.empty_like(xc)
            _lib.call("l2i_hinge_fwd_bwd", xc.data_ptr(), _p(valid), xc.numel(), mode, float(wgt), _p(cnt), loss.data_ptr(),
                      g.data_ptr(), _stream())
            grads.append(g.view(x.shape))
        ctx.save_for_backward(*grads)
        ctx.n = n
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        grads = torch._foreach_mul(list(ctx.saved_tensors), g)
        return (None, None, *grads, *([None] * (3 * ctx.n)))


def hinge_sum(terms, mode):
    """terms: [(x, valid | None, weight, count | None), ...] -> scalar sum of the hinge terms."""
    n = len(terms)
    return HingeSumFn.apply(mode, n, *[t[0] for t in terms], *[t[1] for t in terms], *[t[2] for t in terms], *[t[3] for t in terms])


class GramHeadFn(Function):
    """gram_term[r] = w1 . (Gram_r 1) / C with Gram_r = F F^T / C, F = relu(x_r) -- the only part of the Gram matrix the
    appearance head reads (reference model/rcnn_discriminator_app.py:148-157) -- in one pass over x, both ways."""

    @staticmethod
    def forward(ctx, x, w):
        _chk(x, torch.float32), _chk(w, torch.float32)
        R, H, W, C = x.shape
        out = _zeros((R, 1), x.device)
        keep = torch.empty((2, R, H * W), dtype=torch.float32, device=x.device)
        _lib.call("l2i_gram_head_fwd", x.data_ptr(), w.data_ptr(), out.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(),
                  R, H * W, C, _stream())
        ctx.save_for_backward(x, w, keep)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, keep = ctx.saved_tensors
        R, H, W, C = x.shape
        g = g.contiguous()
        dx = torch.empty_like(x)
        dw = _zeros((C,), x.device)
        dop = torch.empty_like(x, dtype=torch.bfloat16) if HEAD_DX_OP else None   # (x is the result of a convolution read by this head alone)
        _lib.call("l2i_gram_head_bwd", x.data_ptr(), w.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(), g.data_ptr(),
                  dx.data_ptr(), dw.data_ptr(), *_lib.wgrad_scratch(x.device), R, H * W, C, _p(dop), _stream())
        if dop is not None:
            _attach(dx, raw=dop)
        return dx, dw


def gram_head(x, w):
    return GramHeadFn.apply(x, w)


class StageMaskFn(Function):
    """The generator's stage-mask blend (reference model/resnet_generator_app_v2.py:465-470) as one launch forward and two
    backward (csrc/misc.hip): out = bilinear(bmask, H) * (1 - a) + sigmoid(logits[..., y]) * nearest(boxm, H) * a with
    a = sigmoid(alpha[y])."""

    @staticmethod
    def forward(ctx, logits, bmask, boxm, alpha, y, planar=False):
        """planar: `logits` are the class-GATHERED logits (B, O, H, H) of ops.class_logits, not the dense (B, H, H, Cp) tensor."""
        for t in (logits, bmask, boxm, alpha):
            _chk(t, torch.float32)
        b, o, S, _ = bmask.shape
        if planar:
            B, o_, H, W = logits.shape
            Cp = 0
            assert o_ == o
        else:
            B, H, W, Cp = logits.shape
            assert alpha.numel() == Cp
        assert H == W and b == B and boxm.shape == bmask.shape and y.dtype == torch.int64
        out = torch.empty((B, o, H, H), dtype=torch.float32, device=logits.device)
        keep = torch.empty((2, B, o, H, H), dtype=torch.float32, device=logits.device)
        _lib.call("l2i_stage_mask_fwd", logits.data_ptr(), bmask.data_ptr(), boxm.data_ptr(), alpha.data_ptr(), y.data_ptr(),
                  out.data_ptr(), keep.data_ptr(), B, o, H, Cp, S, _stream())
        ctx.save_for_backward(keep, boxm, alpha, y)
        ctx.geom = (B, o, H, Cp, S)
        return out

    @staticmethod
    def backward(ctx, g):
        keep, boxm, alpha, y = ctx.saved_tensors
        B, o, H, Cp, S = ctx.geom
        g = g.contiguous()
        dev = g.device
        gl = torch.empty((B, o, H, H), dtype=torch.float32, device=dev)
        dlogits = torch.empty((B, H, H, Cp), dtype=torch.float32, device=dev) if Cp else None
        dbmask = torch.empty((B, o, S, S), dtype=torch.float32, device=dev)
        dalpha = _zeros(tuple(alpha.shape), dev)
        share = torch.empty((B * o,), dtype=torch.float32, device=dev)   # the slots' shares of dalpha, added per class in slot order (no atomics)
        _lib.call("l2i_stage_mask_bwd", g.data_ptr(), keep.data_ptr(), boxm.data_ptr(), alpha.data_ptr(), y.data_ptr(),
                  gl.data_ptr(), _p(dlogits), dbmask.data_ptr(), dalpha.data_ptr(), B, o, H, Cp, S, share.data_ptr(), alpha.numel(), _stream())
        return (dlogits if Cp else gl), dbmask, None, dalpha, None, None


def stage_mask(logits, bmask, boxm, alpha, y, planar=False):
    return StageMaskFn.apply(logits.contiguous(), bmask.contiguous(), boxm.contiguous(), alpha.contiguous(), y.contiguous(), planar)


class ClassLogitsFn(Function):
    """The last layer of a generator mask head, Conv2d(100, 184, 1), evaluated ONLY for the classes its single reader gathers
    (reference model/resnet_generator_app_v2.py:643-651, 465-466: `seman = gather(m, 1, y)`): lg[b,o,p] = bias[y[b,o]] + a[b,p,:] . W[y[b,o],:]
    -- 8 x 100 MACs per pixel instead of 184 x 100, and neither the 184-channel tensor (96 MB at 64 x 64) nor its dense, mostly-zero
    gradient ever exists (csrc/misc.hip class_logits_*_kernel). a (B, H, W, Cp) f32; w (classes, C) f32 (ops.arena_weight); y (B, O)."""

    @staticmethod
    def forward(ctx, a, w, bias, y):
        _chk(a, torch.float32), _chk(w, torch.float32)
        B, H, W, Cp = a.shape
        O, C = y.shape[1], w.shape[1]
        lg = torch.empty((B, O, H, W), dtype=torch.float32, device=a.device)
        _lib.call("l2i_class_logits_fwd", a.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), lg.data_ptr(), B, O, H * W, Cp, C, w.stride(0), _stream())
        ctx.save_for_backward(a, w, y)
        ctx.has_bias = bias is not None
        ctx.nb = 0 if bias is None else bias.numel()
        return lg

    @staticmethod
    def backward(ctx, g):
        a, w, y = ctx.saved_tensors
        B, H, W, Cp = a.shape
        O, C = y.shape[1], w.shape[1]
        g = _chk(g.contiguous(), torch.float32)
        da = torch.empty_like(a)
        dw = _zeros(tuple(w.shape), a.device)   # (slices of the step's pre-zeroed slab: consumed by the accumulations right behind this node)
        db = _zeros((ctx.nb,), a.device) if ctx.has_bias else None
        parts = -(-(H * W) // (128 if H * W >= 2048 else 256))   # (l2i_class_logits_bwd_parts: pixel parts of an image, one stored row each)
        tmp = torch.empty(((parts + 1) * B * O, 128), dtype=torch.float32, device=a.device)   # (+ one slab: the parts' sums)
        _lib.call("l2i_class_logits_bwd", a.data_ptr(), w.data_ptr(), y.data_ptr(), g.data_ptr(), da.data_ptr(), dw.data_ptr(), _p(db), tmp.data_ptr(),
                  w.shape[0], B, O, H * W, Cp, C, w.stride(0), _stream())
        da._l2i_owned = True   # (fresh, handed to exactly one consumer: NormActFn.backward may overwrite it)
        return da, dw, db, None


def class_logits(a, w, bias, y):
    return ClassLogitsFn.apply(a.contiguous(), w.contiguous(), bias, y.contiguous())


class ProjHeadFn(Function):
    """Projection head of the discriminator in one launch each way (reference model/rcnn_discriminator_app.py:127-129,
    160-166): out[r] = sum_c f[r,c] (wl[c] + E[y[r],c]) + bias, f = scale * sum_hw relu(x). The normalised weights are
    read from the pass's packed operands and their gradients are added straight to the pass's dW accumulators."""

    @staticmethod
    def forward(ctx, x, bias, y, hl, he, pc, scale):
        _chk(x, torch.float32)
        R, H, W, C = x.shape
        assert hl.kh == 1 and hl.ci == C and (he is None or he.ci == C)
        wl = pc.fwd_pack(hl)
        emb = pc.fwd_pack(he) if he is not None else None
        out = torch.empty((R, 1), dtype=torch.float32, device=x.device)
        feat = torch.empty((R, C), dtype=torch.float32, device=x.device)
        _lib.call("l2i_proj_head_fwd", x.data_ptr(), wl.data_ptr(), _p(emb), he.kpad if he is not None else 0, _p(y), _p(bias),
                  float(scale), out.data_ptr(), feat.data_ptr(), R, H * W, C, _code(wl.dtype), _stream())
        ctx.save_for_backward(x, feat, y, bias)
        ctx.meta = (hl, he, pc, float(scale))
        return out

    @staticmethod
    def backward(ctx, g):
        x, feat, y, bias = ctx.saved_tensors
        hl, he, pc, scale = ctx.meta
        R, H, W, C = x.shape
        g = g.contiguous()
        wl = pc.fwd_pack(hl)
        emb = pc.fwd_pack(he) if he is not None else None
        dx = torch.empty_like(x)
        wg = pc.need_wgrad
        dbias = _zeros((1,), x.device) if (wg and bias is not None) else None
        dwl = pc.dw_slice(hl) if wg else None
        demb = pc.dw_slice(he) if (wg and he is not None) else None
        dop = torch.empty_like(x, dtype=torch.bfloat16) if (HEAD_DX_OP and wl.dtype == torch.bfloat16) else None
        _lib.call("l2i_proj_head_bwd", x.data_ptr(), wl.data_ptr(), _p(emb), he.kpad if he is not None else 0, _p(y), g.data_ptr(),
                  feat.data_ptr(), scale, dx.data_ptr(), _p(dwl), _p(demb), he.kp if he is not None else 0, _p(dbias),
                  R, H * W, C, _code(wl.dtype), _p(dop), _stream())
        if dop is not None:
            _attach(dx, raw=dop)
        return dx, dbias, None, None, None, None, None


def proj_head(x, linear, pc, emb=None, y=None, scale=1.0):
    """x (R,H,W,C) pre-ReLU block output; linear: GemmWeight Linear(C -> 1) (+ bias); emb: GemmWeight embedding (K, C), y (R,)."""
    if pc.arena.split:   # forward-only precision mode: the head in f32 torch ops on W / sigma (reference :127-129, 160-166)
        f = torch.relu(x).sum(dim=(1, 2)) * scale
        out = f @ _wbar_f32(linear, pc)[0]
        if linear.bias is not None:
            out = out + linear.bias.detach()
        if emb is not None:
            out = out + (f * _wbar_f32(emb, pc)[y]).sum(dim=1)
        return out.view(-1, 1)
    return ProjHeadFn.apply(x.contiguous(), linear.bias, y, linear, emb, pc, scale)


class EmbDotFn(Function):
    """out[r] = E[y[r]] . w2 + bias: the class-embedding term of the appearance head (reference
    model/rcnn_discriminator_app.py:154-157), w2 = columns [off, off + C) of the head's Linear(2C -> 1) weight."""

    @staticmethod
    def forward(ctx, bias, y, he, hl, off, pc):
        emb, wl = pc.fwd_pack(he), pc.fwd_pack(hl)
        R, C = y.numel(), he.ci
        out = torch.empty((R, 1), dtype=torch.float32, device=y.device)
        _lib.call("l2i_emb_dot_fwd", emb.data_ptr(), he.kpad, y.data_ptr(), wl.data_ptr() + off * wl.element_size(), _p(bias),
                  out.data_ptr(), R, C, _code(wl.dtype), _stream())
        ctx.save_for_backward(y, bias)
        ctx.meta = (he, hl, off, pc)
        return out

    @staticmethod
    def backward(ctx, g):
        y, bias = ctx.saved_tensors
        he, hl, off, pc = ctx.meta
        if not pc.need_wgrad:
            return None, None, None, None, None, None
        g = g.contiguous()
        emb, wl = pc.fwd_pack(he), pc.fwd_pack(hl)
        dbias = _zeros((1,), g.device) if bias is not None else None
        _lib.call("l2i_emb_dot_bwd", emb.data_ptr(), he.kpad, y.data_ptr(), wl.data_ptr() + off * wl.element_size(), g.data_ptr(),
                  pc.dw_slice(he).data_ptr(), he.kp, pc.dw_slice(hl).data_ptr() + 4 * off, _p(dbias), y.numel(), he.ci,
                  _code(wl.dtype), _stream())
        return dbias, None, None, None, None, None


def emb_dot(emb, y, linear, off, pc):
    if pc.arena.split:
        out = (_wbar_f32(emb, pc)[y] * _wbar_f32(linear, pc)[0, off:off + emb.ci]).sum(dim=1)
        return (out + linear.bias.detach() if linear.bias is not None else out).view(-1, 1)
    return EmbDotFn.apply(linear.bias, y, emb, linear, off, pc)


class PspPoolFn(Function):
    """pooled[b,k,:] = sum_p A[k,p] feats[b,p,:]: every adaptive-average-pool stage of the PSP head in one pass over the
    feature map (csrc/psp.hip; taps = generator.psp_taps). Backward adds the gradient the concat branch left in `join`."""

    @staticmethod
    def forward(ctx, feats, taps, join, join_x=None, emit_op=None):
        _chk(feats, torch.float32)
        B, H, W, C = feats.shape
        NB, NQ = taps["nb"], taps["nq"]
        assert H == W
        pooled = torch.empty((B, NB, C), dtype=torch.float32, device=feats.device)
        rows = torch.empty((B, H, NQ, C), dtype=torch.float32, device=feats.device)
        _lib.call("l2i_psp_pool_fwd", feats.data_ptr(), taps["pwx"].data_ptr(), taps["pwy"].data_ptr(), taps["xq"].data_ptr(),
                  pooled.data_ptr(), rows.data_ptr(), B, H, C, NB, NQ, _stream())
        ctx.shape, ctx.join, ctx.taps, ctx.join_x, ctx.emit_op = (B, H, W, C), join, taps, join_x, emit_op
        return pooled

    @staticmethod
    def backward(ctx, g):
        taps = ctx.taps
        B, H, W, C = ctx.shape
        g = g.contiguous()
        add = ctx.join.take() if ctx.join is not None else None
        cat, cat_w, cat_off, cat_dt = None, 0, 0, 0
        if isinstance(add, dict):   # the concat branch handed over the gradient of the concat tensor itself (PspExpandFn.backward)
            cat, cat_w, cat_off, cat_dt, add = add["g"], add["width"], add["off"], _code(add["g"].dtype), None
        add2 = ctx.join_x.take() if ctx.join_x is not None else None   # another reader of feats (the next block) left its gradient
        dfeats = torch.empty((B, H, W, C), dtype=torch.float32, device=g.device)
        # the operand copy of the sum, when this is the complete gradient of feats (what the producing conv's backward reads)
        dop = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=g.device) if (ctx.emit_op is torch.bfloat16 and add2 is not None) else None
        _lib.call("l2i_psp_pool_bwd", g.data_ptr(), taps["aidx"].data_ptr(), taps["aw"].data_ptr(), taps["aidx"].shape[1], _p(add), _p(add2),
                  _p(cat), cat_w, cat_off, cat_dt, dfeats.data_ptr(), _p(dop), B, H * W, C, taps["nb"], _stream())
        if dop is not None:
            _attach(dfeats, raw=dop)
        return dfeats, None, None, None, None


def psp_pool(feats, taps, join=None, join_x=None, emit_op=None):
    """join: shared with psp_expand (the concat branch's part of d feats enters this launch); join_x: a GradJoin another reader of
    feats (created later) leaves its complete gradient in; emit_op: torch.bfloat16 to attach the operand copy of the joined sum."""
    return PspPoolFn.apply(feats.contiguous(), taps, join, join_x, emit_op)


PSP_HANDOVER = __import__("os").environ.get("L2I_PSP_HANDOVER", "1") != "0"   # A/B switch (see PspExpandFn.backward)


class PspExpandFn(Function):
    """cat = [bilinear(align_corners) upsample of each stage's y] ++ feats, written once in the operand dtype of the 3x3
    bottleneck that reads it (its gradient arrives in that dtype too): no per-stage bmm, no f32 concat, no cast pass."""

    @staticmethod
    def forward(ctx, feats, y, taps, op_dtype, join):
        _chk(feats, torch.float32), _chk(y, torch.float32)
        B, H, W, C = feats.shape
        NB, F_ = y.shape[1], y.shape[2]
        ns = taps["uidx"].shape[1]
        assert NB == taps["nb"]
        cat = torch.empty((B, H, W, ns * F_ + C), dtype=op_dtype, device=feats.device)
        _lib.call("l2i_psp_expand_fwd", feats.data_ptr(), y.data_ptr(), taps["uidx"].data_ptr(), taps["uw"].data_ptr(),
                  cat.data_ptr(), B, H * W, C, F_, NB, ns, _code(op_dtype), _stream())
        ctx.meta = (B, H, W, C, NB, F_, ns, op_dtype, join, taps)
        return cat

    @staticmethod
    def backward(ctx, g):
        B, H, W, C, NB, F_, ns, op_dtype, join, taps = ctx.meta
        g = _chk(g.contiguous(), op_dtype)
        dy = torch.empty((B, NB, F_), dtype=torch.float32, device=g.device)
        # d feats = the last C columns of g: when the pooling branch's backward is going to run (it takes from `join`; it runs
        # whenever this one did, the stage outputs y descend from its result) it reads them from g in place -- no f32 copy here
        hand = join is not None and join.state == "open" and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and PSP_HANDOVER
        dfeats = None if hand else torch.empty((B, H, W, C), dtype=torch.float32, device=g.device)
        rows = torch.empty((B, H, taps["nq"], F_), dtype=torch.float32, device=g.device)
        _lib.call("l2i_psp_expand_bwd", g.data_ptr(), taps["uwx"].data_ptr(), taps["uwy"].data_ptr(), taps["xq"].data_ptr(),
                  taps["qoff"].data_ptr(), dy.data_ptr(), _p(dfeats), rows.data_ptr(), B, H, C, F_, NB, taps["nq"], ns,
                  _code(op_dtype), _stream())
        if hand:
            join.give(dict(g=g, width=ns * F_ + C, off=ns * F_))
        elif join is not None:
            dfeats = join.give(dfeats)
        return dfeats, dy, None, None, None


def psp_expand(feats, y, taps, op_dtype, join=None):
    """-> (B, H, W, n_stages * F + C) tensor of op_dtype that a following plain fused_conv consumes as its operand."""
    out = PspExpandFn.apply(feats.contiguous(), y.contiguous(), taps, op_dtype, join)
    out._l2i_raw_op = True
    return out


class ResizeBilinearFn(Function):
    """F.interpolate(x, size=(H, W), mode="bilinear") for planar (b, o, h, w) f32 maps, and its adjoint."""

    @staticmethod
    def forward(ctx, x, H, W):
        _chk(x, torch.float32)
        b, o, h, w = x.shape
        out = torch.empty((b, o, H, W), dtype=torch.float32, device=x.device)
        _lib.call("l2i_resize_bilinear", x.data_ptr(), out.data_ptr(), b * o, h, w, H, W, _stream())
        ctx.in_shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        b, o, h, w = ctx.in_shape
        g = _chk(g.contiguous(), torch.float32)
        dx = torch.empty((b, o, h, w), dtype=torch.float32, device=g.device)
        _lib.call("l2i_resize_bilinear_bwd", g.data_ptr(), dx.data_ptr(), b * o, h, w, g.shape[2], g.shape[3], _stream())
        return dx, None, None


def resize_bilinear(x, H, W):
    return ResizeBilinearFn.apply(x.contiguous(), H, W)


class L1Fn(Function):
    @staticmethod
    def forward(ctx, a, b, weight):
        ac, bc = _chk(a.contiguous(), torch.float32), _chk(b.contiguous(), torch.float32)
        loss = torch.zeros((), dtype=torch.float32, device=a.device)
        grad = torch.empty_like(ac)
        _lib.call("l2i_l1_fwd_bwd", ac.data_ptr(), bc.data_ptr(), ac.numel(), float(weight), loss.data_ptr(), grad.data_ptr(),
                  _lib.wgrad_scratch(a.device)[0], _stream())   # (the workgroups' shares of the loss: stored in the scratch, added in order)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def l1_loss(a, b, weight=1.0):
    return L1Fn.apply(a, b, weight)


def adam_step(flat, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, step_dev=None, lo=0, hi=None):
    """step_dev: 1-element int32 device tensor with the step count (read on the device; graph-capturable).
    lo, hi: the range of the flat buffer to update (multiples of 4 floats) -- the data-parallel trainer updates a layer group as
    soon as its gradient all-reduce has landed (trainer.FlatAdam); default: everything."""
    hi = flat.numel if hi is None else hi
    assert lo % 4 == 0 and (hi - lo) % 4 == 0 and 0 <= lo < hi <= flat.numel
    _lib.call("l2i_adam_step", flat.data.data_ptr() + 4 * lo, flat.grad.data_ptr() + 4 * lo, m.data_ptr() + 4 * lo, v.data_ptr() + 4 * lo,
              hi - lo, float(lr), float(beta1), float(beta2), float(eps), int(step), float(grad_scale), _p(step_dev), _stream())
    touch = getattr(flat, "touch", None)   # (the kernel writes the parameters through raw pointers: cached eval-mode packs are stale,
    if touch is not None:                  #  arena.WeightArena._stamp)
        touch()


# ----------------------------------------------------------------------------- layout-side glue (csrc/layout.hip)
_CONST = {}


def _const(key, make):
    """Small device constants computed ONCE by the torch op the reference uses (so their bits are the reference's)."""
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = make()
    return t


def geometry_dim_mat(device):
    """1 / 1000^(k/8), k = 0..7 (reference model/resnet_generator_app_v2.py:66-68)."""
    return _const(("dim_mat", str(device)), lambda: (1.0 / torch.pow(1000.0, torch.arange(8.0, device=device) / 8.0)).contiguous())


def unit_linspace(n, device):
    return _const(("lin", n, str(device)), lambda: torch.linspace(0, 1, steps=n, device=device).contiguous())


class BoxGeometryFn(Function):
    """relu(WGs(BoxRelationalEmbedding(bbox))) -> (B, O, O): reference model/resnet_generator_app_v2.py:17-76,175-180."""

    @staticmethod
    def forward(ctx, bbox, wg_w, wg_b):
        bbox = _chk(bbox.contiguous(), torch.float32)
        B, O, _ = bbox.shape
        dm = geometry_dim_mat(bbox.device)
        geo = torch.empty((B, O, O), dtype=torch.float32, device=bbox.device)
        _lib.call("l2i_box_geometry_fwd", bbox.data_ptr(), dm.data_ptr(), wg_w.data_ptr(), wg_b.data_ptr(), geo.data_ptr(), B, O, _stream())
        ctx.save_for_backward(bbox, geo)
        ctx.shapes = (wg_w.shape, wg_b.shape)
        return geo

    @staticmethod
    def backward(ctx, g):
        bbox, geo = ctx.saved_tensors
        B, O, _ = bbox.shape
        g = g.contiguous()
        dm = geometry_dim_mat(bbox.device)
        d = _zeros((65,), bbox.device)
        _lib.call("l2i_box_geometry_bwd", bbox.data_ptr(), dm.data_ptr(), geo.data_ptr(), g.data_ptr(), d.data_ptr(), d.data_ptr() + 256,
                  B, O, _stream())
        return None, d[:64].view(ctx.shapes[0]), d[64:65].view(ctx.shapes[1])


def box_geometry(bbox, wg_w, wg_b):
    return BoxGeometryFn.apply(bbox, wg_w, wg_b)


class LayoutMasksFn(Function):
    """sigmoid + masks_to_layout (utils/bilinear.py:137-192) + bbox_mask (app_v2.py:697-721) in one launch.
    m (N, M, M, Cp) f32: channel 0 holds the mask logits (the padded result of the 1-channel conv)."""

    @staticmethod
    def forward(ctx, m, bbox, H, want_boxm):
        m = _chk(m, torch.float32)
        b, o, _ = bbox.shape
        N, M, _, Cp = m.shape
        assert N == b * o
        bb = _chk(bbox.contiguous(), torch.float32)
        lin = unit_linspace(H, m.device)
        bmask = torch.empty((b, o, H, H), dtype=torch.float32, device=m.device)
        boxm = torch.empty((b, o, H, H), dtype=torch.float32, device=m.device) if want_boxm else None
        _lib.call("l2i_layout_masks_fwd", m.data_ptr(), Cp, bb.data_ptr(), lin.data_ptr(), bmask.data_ptr(), _p(boxm), N, M, H, _stream())
        ctx.save_for_backward(m, bb)
        ctx.H = H
        if boxm is not None:
            ctx.mark_non_differentiable(boxm)
        return bmask, boxm

    @staticmethod
    def backward(ctx, g, _g2):
        m, bb = ctx.saved_tensors
        N, M, _, Cp = m.shape
        H = ctx.H
        g = g.contiguous()
        dm = torch.empty_like(m)
        _lib.call("l2i_layout_masks_bwd", m.data_ptr(), Cp, bb.data_ptr(), unit_linspace(H, m.device).data_ptr(), g.data_ptr(),
                  dm.data_ptr(), Cp, N, M, H, _stream())
        return dm, None, None, None


def layout_masks(m, bbox, H, want_boxm=True):
    return LayoutMasksFn.apply(m, bbox, H, want_boxm)


class AddLayerNormFn(Function):
    """y = LayerNorm(a' + b) gamma + beta as a (rows, 1, 1, ldy) stream (pad columns zero) with its operand copy attached;
    a' = a or, perm_O > 0, the reference's h = 1 "concat heads" shuffle of a (app_v2.py:197-198). a: (rows, lda)-like,
    b: (rows, ldb)-like (leading dimensions = last dimension of the contiguous tensors)."""

    @staticmethod
    def forward(ctx, a, b, gamma, beta, eps, D, ldy, perm_O, op_dtype):
        a, b = _chk(a, torch.float32), _chk(b, torch.float32)
        lda, ldb = a.shape[-1], b.shape[-1]
        rows = a.numel() // lda
        assert b.numel() // ldb == rows
        dev = a.device
        y = torch.empty((rows, 1, 1, ldy), dtype=torch.float32, device=dev)
        y_op = torch.empty((rows, 1, 1, ldy), dtype=op_dtype, device=dev)
        st = torch.empty((2, rows), dtype=torch.float32, device=dev)
        _lib.call("l2i_add_layernorm_fwd", a.data_ptr(), lda, b.data_ptr(), ldb, gamma.data_ptr(), beta.data_ptr(), float(eps), y.data_ptr(),
                  ldy, y_op.data_ptr(), _code(op_dtype), st[0].data_ptr(), st[1].data_ptr(), rows, D, perm_O, _stream())
        ctx.save_for_backward(a, b, gamma, st)
        ctx.meta = (D, ldy, perm_O, a.shape, b.shape)
        ctx.mark_non_differentiable(y_op)
        return y, y_op

    @staticmethod
    def backward(ctx, g, _g2):
        a, b, gamma, st = ctx.saved_tensors
        D, ldy, perm_O, a_shape, b_shape = ctx.meta
        lda, ldb = a_shape[-1], b_shape[-1]
        rows = a.numel() // lda
        g = g.contiguous()
        dev = g.device
        da = torch.empty(a_shape, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        db = torch.empty(b_shape, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        dgb = _zeros((2, D), dev)
        _lib.call("l2i_add_layernorm_bwd", a.data_ptr(), lda, b.data_ptr(), ldb, gamma.data_ptr(), st[0].data_ptr(), st[1].data_ptr(),
                  g.data_ptr(), ldy, _p(da), _p(db), dgb[0].data_ptr(), dgb[1].data_ptr(), rows, D, perm_O, *_lib.wgrad_scratch(dev), _stream())
        return da, db, dgb[0], dgb[1], None, None, None, None, None


def add_layernorm(a, b, ln, D, ldy, op_dtype, perm_O=0):
    """ln: an nn.LayerNorm(D). Returns the (rows, 1, 1, ldy) f32 stream; its operand-dtype copy rides on it (ops._sibling)."""
    y, y_op = AddLayerNormFn.apply(a, b, ln.weight, ln.bias, ln.eps, D, ldy, perm_O, op_dtype)
    _attach(y, raw=y_op)
    return y


class LatentFn(Function):
    """[z | label_embedding(y)] padded to ld columns, as a (rows, 1, 1, ld) stream + operand copy, plus the attention's key mask
    (y != 0) -- reference model/resnet_generator_app_v2.py:437-441."""

    @staticmethod
    def forward(ctx, z, emb, y, ld, op_dtype):
        z, emb = _chk(z.contiguous(), torch.float32), _chk(emb, torch.float32)
        y = y.contiguous()
        if y.dtype != torch.int64:   # (the kernel reads long long labels)
            y = y.to(torch.int64)
        rows, Z, E = y.numel(), z.shape[-1], emb.shape[1]
        dev = z.device
        out = torch.empty((rows, 1, 1, ld), dtype=torch.float32, device=dev)
        out_op = torch.empty((rows, 1, 1, ld), dtype=op_dtype, device=dev)
        kv = torch.empty(y.shape, dtype=torch.int32, device=dev)
        _lib.call("l2i_latent_fwd", z.data_ptr(), emb.data_ptr(), y.data_ptr(), out.data_ptr(), out_op.data_ptr(), _code(op_dtype),
                  kv.data_ptr(), rows, Z, E, ld, _stream())
        ctx.save_for_backward(y)
        ctx.meta = (rows, Z, E, ld, emb.shape)
        ctx.mark_non_differentiable(out_op, kv)
        return out, out_op, kv

    @staticmethod
    def backward(ctx, g, _g2, _g3):
        (y,) = ctx.saved_tensors
        rows, Z, E, ld, eshape = ctx.meta
        g = g.contiguous()
        demb = _zeros(tuple(eshape), g.device)
        _lib.call("l2i_latent_bwd", g.data_ptr(), y.data_ptr(), demb.data_ptr(), rows, Z, E, ld, _stream())
        return None, demb, None, None, None


def latent(z, emb_weight, y, ld, op_dtype):
    out, out_op, kv = LatentFn.apply(z, emb_weight, y, ld, op_dtype)
    _attach(out, raw=out_op)
    return out, kv


class FcToNhwcFn(Function):
    """A Linear's (N, 1, 1, C*P) result seen as .view(N, C, 4, 4) (reference :453, mask_regression.py:87), delivered as the
    NHWC (N, 4, 4, C) f32 stream + its operand copy in one launch; the backward returns the (N, 1, 1, C*P) gradient with ITS
    operand copy (what the Linear's weight gradient reads)."""

    @staticmethod
    def forward(ctx, x, C, op_dtype):
        x = _chk(x, torch.float32)
        N = x.shape[0]
        P = x.numel() // (N * C)
        side = int(round(P ** 0.5))
        out = torch.empty((N, side, side, C), dtype=torch.float32, device=x.device)
        out_op = torch.empty((N, side, side, C), dtype=op_dtype, device=x.device)
        _lib.call("l2i_fc_to_nhwc", x.data_ptr(), out.data_ptr(), out_op.data_ptr(), _code(op_dtype), N, C, P, 0, _stream())
        ctx.meta = (N, C, P, op_dtype, tuple(x.shape))
        ctx.mark_non_differentiable(out_op)
        return out, out_op

    @staticmethod
    def backward(ctx, g, _g2):
        N, C, P, op_dtype, shape = ctx.meta
        g = _chk(g.contiguous(), torch.float32)
        d = torch.empty(shape, dtype=torch.float32, device=g.device)
        d_op = torch.empty(shape, dtype=op_dtype, device=g.device)
        _lib.call("l2i_fc_to_nhwc", g.data_ptr(), d.data_ptr(), d_op.data_ptr(), _code(op_dtype), N, C, P, 1, _stream())
        _attach(d, raw=d_op)
        return d, None, None


def fc_to_nhwc(x, C, op_dtype):
    out, out_op = FcToNhwcFn.apply(x, C, op_dtype)
    _attach(out, raw=out_op)
    return out


class TanhNchwFn(Function):
    """img (B, C, H, W) = tanh(pre[..., :C]) from the NHWC to-RGB result (reference :497-499); the backward writes the padded
    NHWC gradient and its operand copy (the to-RGB convolution's dY) in one launch."""

    @staticmethod
    def forward(ctx, pre, C, op_dtype):
        pre = _chk(pre, torch.float32)
        B, H, W, Cp = pre.shape
        img = torch.empty((B, C, H, W), dtype=torch.float32, device=pre.device)
        _lib.call("l2i_tanh_nchw_fwd", pre.data_ptr(), img.data_ptr(), B, C, Cp, H * W, _stream())
        ctx.save_for_backward(img)
        ctx.meta = (B, H, W, Cp, C, op_dtype)
        return img

    @staticmethod
    def backward(ctx, g):
        (img,) = ctx.saved_tensors
        B, H, W, Cp, C, op_dtype = ctx.meta
        g = _chk(g.contiguous(), torch.float32)
        d = torch.empty((B, H, W, Cp), dtype=torch.float32, device=g.device)
        d_op = torch.empty((B, H, W, Cp), dtype=op_dtype, device=g.device)
        _lib.call("l2i_tanh_nchw_bwd", img.data_ptr(), g.data_ptr(), d.data_ptr(), d_op.data_ptr(), _code(op_dtype), B, C, Cp, H * W, _stream())
        _attach(d, raw=d_op)
        return d, None, None


def tanh_nchw(pre, C, op_dtype):
    return TanhNchwFn.apply(pre, C, op_dtype)


class InReluUp2Fn(Function):
    """InstanceNorm2d (no affine) -> ReLU -> bilinear x2 (align_corners=False) of per-object maps x (N, S, S, C), S in (4, 8):
    the step between two convolutions of the mask regressor (reference model/mask_regression.py:64-95) as one launch each
    way; the result carries its operand copy for the convolution that reads it next."""

    @staticmethod
    def forward(ctx, x, eps, op_dtype):
        x = _chk(x, torch.float32)
        N, S, S2, C = x.shape
        assert S == S2 and S in (4, 8)
        out = torch.empty((N, 2 * S, 2 * S, C), dtype=torch.float32, device=x.device)
        op = torch.empty((N, 2 * S, 2 * S, C), dtype=op_dtype, device=x.device)
        _lib.call("l2i_in_relu_up2_fwd", x.data_ptr(), out.data_ptr(), op.data_ptr(), _code(op_dtype), N, S, C, float(eps), _stream())
        _attach(out, raw=op)
        ctx.save_for_backward(x)
        ctx.eps, ctx.op_dtype = float(eps), op_dtype
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        N, S, _, C = x.shape
        g = _chk(g.contiguous(), torch.float32)
        dx = torch.empty_like(x)
        dx_op = torch.empty(x.shape, dtype=ctx.op_dtype, device=x.device)
        _lib.call("l2i_in_relu_up2_bwd", x.data_ptr(), g.data_ptr(), dx.data_ptr(), dx_op.data_ptr(), _code(ctx.op_dtype), N, S, C,
                  ctx.eps, _stream())
        _attach(dx, raw=dx_op)
        return dx, None, None


def in_relu_up2(x, eps, op_dtype):
    return InReluUp2Fn.apply(x, eps, op_dtype)


class Up2NhwcFn(Function):
    """Bilinear x2 (align_corners=False) of NHWC maps (N, S, S, C) f32 -> (N, 2S, 2S, C) + the operand copy the next convolution reads:
    F.interpolate between the convolutions of the VG generator's MaskRegressNet (reference model/mask_regression.py:20-33,42-58).
    One launch each way (csrc/layout.hip) -- a batched torch.matmul with the dense resampling matrix (rocBLAS) through round 5."""

    @staticmethod
    def forward(ctx, x, op_dtype):
        x = _chk(x.contiguous(), torch.float32)
        N, S, S2, C = x.shape
        assert S == S2 and C % 4 == 0
        out = torch.empty((N, 2 * S, 2 * S, C), dtype=torch.float32, device=x.device)
        op = torch.empty((N, 2 * S, 2 * S, C), dtype=op_dtype, device=x.device)
        _lib.call("l2i_up2_nhwc_fwd", x.data_ptr(), out.data_ptr(), op.data_ptr(), _code(op_dtype), N, S, C, _stream())
        _attach(out, raw=op)
        ctx.meta = (N, S, C, op_dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        N, S, C, op_dtype = ctx.meta
        g = _chk(g.contiguous(), torch.float32)
        dx = torch.empty((N, S, S, C), dtype=torch.float32, device=g.device)
        dx_op = torch.empty((N, S, S, C), dtype=op_dtype, device=g.device)
        _lib.call("l2i_up2_nhwc_bwd", g.data_ptr(), dx.data_ptr(), dx_op.data_ptr(), _code(op_dtype), N, S, C, _stream())
        _attach(dx, raw=dx_op)
        return dx, None


def up2_nhwc(x, op_dtype):
    return Up2NhwcFn.apply(x, op_dtype)


class PspStagesFn(Function):
    """The four pyramid stages of the PSP head between the pooling and the expansion kernels (reference
    model/resnet_generator_app_v2.py:741-746: Conv2d(C, F, 1, bias=False) -> BatchNorm2d -> ReLU on the s x s pooled maps):
    pooled (B, NB, C) -> (B, NB, F), one launch (two backward). apply(pooled, meta, *W, *gamma, *beta) with the stage modules'
    own parameters (S each) and meta = (sizes, training, eps, momentum, running_means, running_vars): running statistics are
    updated in place in train mode (momentum, unbiased variance), read in eval mode."""

    @staticmethod
    def _ptrs(ts):
        return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])

    @staticmethod
    def forward(ctx, pooled, meta, *params):
        sizes, training, eps, momentum, rms, rvs = meta
        S = len(sizes)
        Ws, gs, bs = params[:S], params[S:2 * S], params[2 * S:]
        pooled = _chk(pooled, torch.float32)
        for t in (*Ws, *gs, *bs, *rms, *rvs):
            _chk(t, torch.float32)
        B, NB, C = pooled.shape
        F_ = Ws[0].shape[0]
        dev = pooled.device
        raw = torch.empty((B, NB, F_), dtype=torch.float32, device=dev)
        y = torch.empty((B, NB, F_), dtype=torch.float32, device=dev)
        stat = torch.empty((S, 2, F_), dtype=torch.float32, device=dev)
        sz = (ctypes.c_int * S)(*sizes)
        P = PspStagesFn._ptrs
        _lib.call("l2i_psp_stages_fwd", pooled.data_ptr(), P(Ws), P(gs), P(bs), P(rms), P(rvs), raw.data_ptr(), y.data_ptr(), stat.data_ptr(),
                  B, NB, C, F_, S, sz, int(training), float(eps), float(momentum), _stream())
        ctx.save_for_backward(pooled, raw, stat, *params)
        ctx.meta = (tuple(sizes), bool(training), tuple(Ws[0].shape))
        return y

    @staticmethod
    def backward(ctx, g):
        pooled, raw, stat, *params = ctx.saved_tensors
        sizes, training, wshape = ctx.meta
        S = len(sizes)
        Ws, gs, bs = params[:S], params[S:2 * S], params[2 * S:]
        B, NB, C = pooled.shape
        F_ = wshape[0]
        dev = g.device
        g = g.contiguous()
        draw = torch.empty_like(raw)
        dpooled = torch.empty_like(pooled)
        dW = torch.empty((S, F_, C), dtype=torch.float32, device=dev)
        dgb = torch.empty((2, S, F_), dtype=torch.float32, device=dev)
        sz = (ctypes.c_int * S)(*sizes)
        P = PspStagesFn._ptrs
        _lib.call("l2i_psp_stages_bwd", pooled.data_ptr(), P(Ws), P(gs), P(bs), raw.data_ptr(), stat.data_ptr(), g.data_ptr(), draw.data_ptr(),
                  dpooled.data_ptr(), dW.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(), B, NB, C, F_, S, sz, int(training), _stream())
        return (dpooled, None, *[dW[i].view(wshape) for i in range(S)], *[dgb[0, i] for i in range(S)], *[dgb[1, i] for i in range(S)])


def psp_stages(pooled, convs, bns, sizes, training):
    """convs / bns: the stage modules (nn.Conv2d(C, F, 1, bias=False), nn.BatchNorm2d(F)) in stage order."""
    meta = (tuple(sizes), training, bns[0].eps, bns[0].momentum, [b.running_mean for b in bns], [b.running_var for b in bns])
    return PspStagesFn.apply(pooled, meta, *[c.weight for c in convs], *[b.weight for b in bns], *[b.bias for b in bns])


def roi_layout(bbox, label, size, two_scale):
    """(rois (R, 5), y (R,), valid (R,) int32, count (1,) int32) with the R = b*o rows compacted in the reference's output
    order (csrc/layout.hip roi_layout_kernel; reference model/rcnn_discriminator_app.py:131-146,402-417). One launch."""
    bb = _chk(bbox.contiguous(), torch.float32)
    b, o, _ = bb.shape
    lab = label.reshape(b, o).contiguous()
    if lab.dtype != torch.int64:
        lab = lab.to(torch.int64)
    R, dev = b * o, bb.device
    rois = torch.empty((R, 5), dtype=torch.float32, device=dev)
    y = torch.empty((R,), dtype=torch.int64, device=dev)
    valid = torch.empty((R,), dtype=torch.int32, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    _lib.call("l2i_roi_layout", bb.data_ptr(), lab.data_ptr(), float(size), int(two_scale), o, R, rois.data_ptr(), y.data_ptr(),
              valid.data_ptr(), count.data_ptr(), _stream())
    return rois, y, valid, count


class ImageNhwcFn(Function):
    """(b, 3, H, W) image -> the padded NHWC stream (b, H, W, 8) the discriminator's first block reads and its 2x2 average
    (the OptimizedBlock shortcut pools BEFORE its 1x1 conv), each with its operand copy: one launch instead of permute + pad +
    contiguous + avg_pool2d + two casts; one launch backward."""

    @staticmethod
    def forward(ctx, img, cp, op_dtype, want_half):
        img = _chk(img.contiguous(), torch.float32)
        B, C, H, W = img.shape
        dev = img.device
        x = torch.empty((B, H, W, cp), dtype=torch.float32, device=dev)
        x_op = torch.empty((B, H, W, cp), dtype=op_dtype, device=dev)
        xs = torch.empty((B, H // 2, W // 2, cp), dtype=torch.float32, device=dev) if want_half else None
        xs_op = torch.empty((B, H // 2, W // 2, cp), dtype=op_dtype, device=dev) if want_half else None
        _lib.call("l2i_image_nhwc_fwd", img.data_ptr(), x.data_ptr(), x_op.data_ptr(), _p(xs), _p(xs_op), _code(op_dtype), B, C, cp, H, W,
                  _stream())
        ctx.meta = (B, C, cp, H, W)
        ctx.mark_non_differentiable(x_op)
        if want_half:
            ctx.mark_non_differentiable(xs_op)
        return x, x_op, xs, xs_op

    @staticmethod
    def backward(ctx, dx, _a, dxs, _b):
        B, C, cp, H, W = ctx.meta
        dx = None if dx is None else _chk(dx.contiguous(), torch.float32)
        dxs = None if dxs is None else _chk(dxs.contiguous(), torch.float32)
        ref = dx if dx is not None else dxs
        dimg = torch.empty((B, C, H, W), dtype=torch.float32, device=ref.device)
        _lib.call("l2i_image_nhwc_bwd", _p(dx), _p(dxs), dimg.data_ptr(), B, C, cp, H, W, _stream())
        return dimg, None, None, None


def image_nhwc(img, cp, op_dtype, want_half):
    """-> (x, xs | None): f32 NHWC streams with their operand copies attached (ops._sibling "raw")."""
    x, x_op, xs, xs_op = ImageNhwcFn.apply(img, cp, op_dtype, want_half)
    _attach(x, raw=x_op)
    if xs is not None:
        _attach(xs, raw=xs_op)
    return x, xs


class ChannelDropoutFn(Function):
    """nn.Dropout2d on an NHWC stream given the uniform draws u (B, C): y = x * (u >= p) / (1 - p)."""

    @staticmethod
    def forward(ctx, x, u, prob):
        x, u = _chk(x, torch.float32), _chk(u, torch.float32)
        B, H, W, C = x.shape
        out = torch.empty_like(x)
        _lib.call("l2i_channel_dropout", x.data_ptr(), u.data_ptr(), out.data_ptr(), B, H * W, C, float(prob), _stream())
        ctx.save_for_backward(u)
        ctx.prob = prob
        return out

    @staticmethod
    def backward(ctx, g):
        (u,) = ctx.saved_tensors
        g = _chk(g.contiguous(), torch.float32)
        B, H, W, C = g.shape
        d = torch.empty_like(g)
        _lib.call("l2i_channel_dropout", g.data_ptr(), u.data_ptr(), d.data_ptr(), B, H * W, C, float(ctx.prob), _stream())
        return d, None, None


def channel_dropout(x, u, prob):
    return ChannelDropoutFn.apply(x, u, prob)

