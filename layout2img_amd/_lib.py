"""ctypes binding of libl2i_hip.so (the C ABI declared in include/l2i.h).

There is no fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libl2i_hip.so")

F32, BF16 = 0, 1
BF16X3 = 3   # l2i_weights_prepare only: bf16 with split (hi + lo) forward packs, the forward-only "bf16x3" precision mode

_p, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong

SIGNATURES = {
    "l2i_version": [],
    "l2i_conv2d_fwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p],
    "l2i_conv2d_fwd_sc": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p,
                          _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "l2i_conv2d_fwd_dual": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p,
                            _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _ll, _p],
    "l2i_conv2d_dgrad_sc": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _ll, _p],
    "l2i_in_relu_up2_fwd": [_p, _p, _p, _i, _ll, _i, _i, _f, _p],
    "l2i_in_relu_up2_bwd": [_p, _p, _p, _p, _i, _ll, _i, _i, _f, _p],
    "l2i_up2_nhwc_fwd": [_p, _p, _p, _i, _ll, _i, _i, _p],
    "l2i_up2_nhwc_bwd": [_p, _p, _p, _i, _ll, _i, _i, _p],
    "l2i_set_conv_config": [_i],
    "l2i_timing": [_i],
    "l2i_timing_read": [_i, _p, _p],
    "l2i_conv2d_wgrad": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _ll, _p],
    "l2i_conv2d_wgrad_sc": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _ll, _p, _p, _i, _i, _i, _p, _p],
    "l2i_conv2d_wgrad_dual": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _ll, _p, _p, _i, _i, _i, _p, _p, _p, _i, _p],
    "l2i_weights_prepare": [_p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _p, _p, _ll, _p, _p, _i, _i, _i, _p, _i, _p, _ll, _ll, _p],
    "l2i_weights_backward": [_p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    "l2i_weights_backward2": [_p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _p],
    "l2i_channel_stats": [_p, _ll, _i, _ll, _p, _p, _p, _i, _p, _ll, _p],
    "l2i_norm_mod_fwd": [_p, _i, _i, _i, _p, _p, _f, _f, _i, _p, _i, _p, _p, _ll, _ll, _i, _i, _p, _p, _i, _p, _p, _f, _p],
    "l2i_norm_mod_bwd_a": [_p, _p, _i, _i, _i, _p, _p, _f, _f, _i, _p, _i, _p, _p, _ll, _ll, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _ll, _i, _p],
    "l2i_norm_bwd_b": [_p, _p, _p, _p, _p, _p, _p, _ll, _i, _ll, _f, _f, _i, _p, _p],
    "l2i_roi_align_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _f, _f, _i, _p, _p, _p],
    "l2i_roi_align_bwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _f, _f, _i, _i, _i, _p, _p, _p],
    "l2i_box_attention_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "l2i_box_attention_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "l2i_hinge_fwd_bwd": [_p, _p, _i, _i, _f, _p, _p, _p, _p],
    "l2i_l1_fwd_bwd": [_p, _p, _ll, _f, _p, _p, _p, _p],
    "l2i_adam_step": [_p, _p, _p, _p, _ll, _f, _f, _f, _f, _i, _f, _p, _p],
    "l2i_cast_op": [_p, _p, _p, _ll, _i, _p],
    "l2i_split_cast": [_p, _p, _ll, _i, _i, _p],
    "l2i_debug_stamp": [_p, _p],
    "l2i_set_wgrad_blocks": [_i],
    "l2i_debug_occupancy": [_i, _i],
    "l2i_resize_bilinear": [_p, _p, _ll, _i, _i, _i, _i, _p],
    "l2i_gram_head_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "l2i_gram_head_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _ll, _i, _i, _i, _p, _p],
    "l2i_proj_head_fwd": [_p, _p, _p, _i, _p, _p, _f, _p, _p, _i, _i, _i, _i, _p],
    "l2i_proj_head_bwd": [_p, _p, _p, _i, _p, _p, _p, _f, _p, _p, _p, _i, _p, _i, _i, _i, _i, _p, _p],
    "l2i_emb_dot_fwd": [_p, _i, _p, _p, _p, _p, _i, _i, _i, _p],
    "l2i_emb_dot_bwd": [_p, _i, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _p],
    "l2i_psp_pool_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "l2i_psp_pool_bwd": [_p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _p, _p, _i, _i, _i, _i, _p],
    "l2i_psp_expand_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "l2i_psp_expand_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "l2i_class_logits_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "l2i_class_logits_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "l2i_class_logits_bwd_parts": [_i],
    "l2i_stage_mask_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "l2i_stage_mask_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p],
    "l2i_relu_bwd": [_p, _p, _p, _p, _ll, _p],
    "l2i_box_geometry_fwd": [_p, _p, _p, _p, _p, _i, _i, _p],
    "l2i_box_geometry_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _p],
    "l2i_layout_masks_fwd": [_p, _i, _p, _p, _p, _p, _i, _i, _i, _p],
    "l2i_layout_masks_bwd": [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "l2i_add_layernorm_fwd": [_p, _i, _p, _i, _p, _p, _f, _p, _i, _p, _i, _p, _p, _i, _i, _i, _p],
    "l2i_add_layernorm_bwd": [_p, _i, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p, _ll, _p],
    "l2i_latent_fwd": [_p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _p],
    "l2i_latent_bwd": [_p, _p, _p, _i, _i, _i, _i, _p],
    "l2i_fc_to_nhwc": [_p, _p, _p, _i, _ll, _i, _i, _i, _p],
    "l2i_tanh_nchw_fwd": [_p, _p, _ll, _i, _i, _i, _p],
    "l2i_tanh_nchw_bwd": [_p, _p, _p, _p, _i, _ll, _i, _i, _i, _p],
    "l2i_roi_layout": [_p, _p, _f, _i, _i, _i, _p, _p, _p, _p, _p],
    "l2i_image_nhwc_fwd": [_p, _p, _p, _p, _p, _i, _ll, _i, _i, _i, _i, _p],
    "l2i_image_nhwc_bwd": [_p, _p, _p, _ll, _i, _i, _i, _i, _p],
    "l2i_resize_bilinear_bwd": [_p, _p, _ll, _i, _i, _i, _i, _p],
    "l2i_channel_dropout": [_p, _p, _p, _ll, _i, _i, _f, _p],
    "l2i_psp_stages_fwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _f, _f, _p],
    "l2i_psp_stages_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p],
}

_lib = None
_FN = {}     # name -> bound foreign function; filled by load() so call() is one dict lookup per launch


def load():
    """Load the shared library (once). Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -m layout2img_amd.build` (or __graft_entry__.build()). "
            "There is no CPU or PyTorch fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _i
        _FN[name] = fn
    if os.environ.get("L2I_FASTCALL", "0") == "1":
        # the generated CPython binding instead of ctypes' per-argument conversion (fastcall.py: same symbols, same values; opt-in until it
        # has run on a GPU box). Calls whose arguments it does not take (ctypes.byref objects: l2i_timing_read) stay on ctypes.
        from . import fastcall
        mod = fastcall.load(lib=lib)
        for name in SIGNATURES:
            if name != "l2i_timing_read":
                _FN[name] = getattr(mod, name)
    _lib = lib
    return lib


def call(name, *args):
    fn = _FN.get(name)
    if fn is None:
        load()
        fn = _FN[name]   # KeyError: a name that include/l2i.h does not declare
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc} ({'bad argument' if rc == -1 else 'HIP launch error'})")


WS_FLOATS = 32 * 4 * 1024   # L2I_WS_FLOATS of include/l2i.h
_WS = {}


def current_device():
    import torch
    return torch._C._cuda_getDevice()


def raw_stream():
    """hipStream_t of torch's current stream on the CURRENT device (queried per call: two plain C calls). The public
    torch.cuda.current_stream() costs ~10 us of Python per call, which at ~1000 launches per iteration is a tenth of the
    host time. ops._chk refuses tensors that live on another device than the current one, so a kernel is never
    enqueued on a stream of the wrong GPU."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def workspace(device):
    """Pointer to the (device, current stream)'s all-zero reduction workspace (self-cleaning; csrc/common.h).
    Created on first use; GanTrainer.capture creates the ones of its capture streams BEFORE capturing, so none is
    ever allocated inside a graph's private pool."""
    import torch
    idx = torch.device(device).index
    key = (current_device() if idx is None else idx, raw_stream())
    w = _WS.get(key)
    if w is None:
        w = _WS[key] = torch.zeros(WS_FLOATS, dtype=torch.float32, device=device)
    return w.data_ptr()


WGRAD_SCRATCH_FLOATS = 32 << 20   # L2I_WGRAD_SCRATCH_FLOATS of include/l2i.h
_WGS = {}


def wgrad_scratch(device):
    """(pointer, floats) of the (device, current stream)'s scratch: partial tiles of split weight-gradient / convolution launches and,
    since round 6, every launch's stored partial rows (batch statistics, power iteration, bias gradients, channel totals ...), which an
    ordered fold adds up behind the launch. Created on first use; GanTrainer.capture creates the ones of its streams before capturing.
    (Called ~400 times per iteration: the fast path is two C calls and a dict lookup.)"""
    import torch
    idx = device.index if type(device) is torch.device else torch.device(device).index
    if idx is None:
        idx = torch._C._cuda_getDevice()
    key = (idx, torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    w = _WGS.get(key)
    if w is None:
        t = torch.empty(WGRAD_SCRATCH_FLOATS, dtype=torch.float32, device=device)
        w = _WGS[key] = (t, t.data_ptr())
    return w[1], WGRAD_SCRATCH_FLOATS
