"""Context-aware layout-to-image generators on the HIP path.

Module boundary mirrors the reference (same class names, constructor arguments, forward signature,
state_dict keys): `ResnetGenerator128_context` (model/resnet_generator_app_v2.py:400-506) and the
VG `context_aware_generator` (model/resnet_generator_vg.py:639-727). Internally activations are
NHWC f32 streams, every conv / linear runs through the MFMA implicit-GEMM kernel with a fused
prologue (cast | ISLA norm+ReLU | BN+ReLU | IN+ReLU), and all weights come from the per-pass
WeightArena. Small layout-sized tensors (masks, embeddings, box geometry) are handled with
device-side torch ops.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .arena import FlatParams, GemmWeight, WeightArena
from .ops import NormSpec, fused_conv


def _pad_last(t, n):
    return t if t.shape[-1] == n else F.pad(t, (0, n - t.shape[-1]))


class BNState(nn.Module):
    """Buffers (and optional affine parameters) of a batch-norm layer under the reference's key names."""

    def __init__(self, c, affine=True, eps=1e-5, momentum=0.1):
        super().__init__()
        self.c, self.eps, self.momentum, self.affine = c, eps, momentum, affine
        if affine:
            self.weight = nn.Parameter(torch.ones(c))
            self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def _padded_running(self, cp):
        """running_mean / running_var as the first c entries of PADDED storage (cp channels: mean 0, var 1 behind them),
        so that a layer with a padded channel count reads and updates them in place -- no pad / copy-back launches per
        forward. Re-linked if the buffers were moved or replaced (.to(), a fresh load)."""
        rm, rv = self.running_mean, self.running_var
        st = getattr(self, "_store", None)
        if (st is None or st.shape[1] != cp or st.device != rm.device or rm.data_ptr() != st[0].data_ptr()
                or rv.data_ptr() != st[1].data_ptr()):
            with torch.no_grad():
                st = torch.zeros(2, cp, dtype=torch.float32, device=rm.device)
                st[1].fill_(1.0)
                st[0, :self.c].copy_(rm)
                st[1, :self.c].copy_(rv)
                rm.data, rv.data = st[0, :self.c], st[1, :self.c]
            self._store = st
        return st[0], st[1]

    def spec(self, training, sync, cp=None):
        """NormSpec + (weight, bias) padded to cp channels."""
        cp = cp or self.c
        running = self._padded_running(cp) if cp != self.c else (self.running_mean, self.running_var)
        s = NormSpec(1 if self.affine else 2, eps=self.eps, relu=True, sync=sync, running=running, momentum=self.momentum,
                     training=training)
        w = b = None
        if self.affine:
            w, b = ops.pad_param(self.weight, cp), ops.pad_param(self.bias, cp)
        return s, w, b

    _nbt_shared = False   # True once the network keeps every num_batches_tracked in one tensor and bumps them together

    def commit(self, cp=None):
        """After a train-mode forward (the running statistics were updated in place by the normalisation launch)."""
        if self.training and not self._nbt_shared:
            self.num_batches_tracked += 1


class ISLANorm(nn.Module):
    """SpatialAdaptiveSynBatchNorm2d (reference model/norm_module.py:152-186): parameters only; the
    computation is the fused norm prologue of the following convolution."""

    def __init__(self, c, num_w):
        super().__init__()
        self.c = c
        self.weight_proj = GemmWeight("linear", c, num_w, sn=True, eps=1e-12)
        self.bias_proj = GemmWeight("linear", c, num_w, sn=True, eps=1e-12)
        self.batch_norm2d = BNState(c, affine=False)

    _proj = None   # (gw, gb) of the current forward pass when the network runs all projections as one grouped GEMM

    def project(self, w, pc, B, O):
        if self._proj is not None:
            return self._proj
        gw = fused_conv(w, self.weight_proj, pc).view(B, O, -1)
        gb = fused_conv(w, self.bias_proj, pc).view(B, O, -1)
        return gw, gb

    def spec(self, training, sync):
        bn = self.batch_norm2d
        return NormSpec(0, eps=bn.eps, relu=True, sync=sync, running=(bn.running_mean, bn.running_var),
                        momentum=bn.momentum, training=training)


_RESAMPLE = {}
_PSP_TAPS = {}


def resample_matrix(kind, n_in, n_out, device):
    """(n_out^2, n_in^2) matrix of a fixed spatial resampling (square maps), built once by pushing the identity
    basis through the torch op it replaces: 'bilinear' (align_corners=False), 'bilinear_ac' (align_corners=True,
    the PSP upsample, reference :750), 'adaptive_avg' (nn.AdaptiveAvgPool2d, reference :743)."""
    key = (kind, n_in, n_out, str(device))
    if key not in _RESAMPLE:
        eye = torch.eye(n_in * n_in).view(n_in * n_in, 1, n_in, n_in)
        if kind == "adaptive_avg":
            out = F.adaptive_avg_pool2d(eye, (n_out, n_out))
        else:
            out = F.interpolate(eye, size=(n_out, n_out), mode="bilinear", align_corners=(kind == "bilinear_ac"))
        _RESAMPLE[key] = out.reshape(n_in * n_in, n_out * n_out).t().contiguous().to(device)
    return _RESAMPLE[key]


def psp_taps(H, sizes, device):
    """Per-pixel tap tables of the PSP resamplings (csrc/psp.hip), extracted from the dense maps torch's own ops give for
    the identity basis (resample_matrix: nn.AdaptiveAvgPool2d, reference :743; bilinear align_corners=True, :750):
    dict(aidx, aw (HW, 12): bins a pixel is pooled into; uidx, uw (HW, n_stages, 4): bins a pixel's bilinear sample reads;
    nb = total number of bins; pwx / uwx (nq, H), pwy / uwy (nb, H), xq (nb), qoff: the separable form, see below).
    Bins are numbered stage after stage, row-major inside a stage."""
    key = (H, tuple(sizes), str(device))
    if key not in _PSP_TAPS:
        assert max(sizes) <= 8, "csrc/psp.hip psp_expand_rows_kernel keeps at most 8 x-bins of a stage in registers"
        At = torch.cat([resample_matrix("adaptive_avg", H, s, "cpu") for s in sizes], dim=0).t().contiguous()   # (HW, NB)
        assert int((At != 0).sum(dim=1).max()) <= 12
        _, ai = torch.topk(At.abs(), 12, dim=1)
        aw = torch.gather(At, 1, ai)
        ai = torch.where(aw != 0, ai, torch.zeros_like(ai))
        uis, uws, off = [], [], 0
        for s in sizes:
            Us = resample_matrix("bilinear_ac", s, H, "cpu")                                                       # (HW, s*s)
            assert int((Us != 0).sum(dim=1).max()) <= 4
            k = min(4, s * s)
            _, ui = torch.topk(Us.abs(), k, dim=1)
            uw = torch.gather(Us, 1, ui)
            ui, uw = F.pad(ui, (0, 4 - k)), F.pad(uw, (0, 4 - k))
            uis.append(ui + off), uws.append(uw)
            off += s * s
        # the same two maps factored along x and y (both are separable) for the reductions over pixels: x-bins q are
        # numbered stage after stage (qoff), bin k = (stage, ky, kx) reads x-bin xq[k] with the y-weights wy[k]
        eye = torch.eye(H).view(H, 1, 1, H)
        pwx, uwx, pwy, uwy, xq, qoff = [], [], [], [], [], [0]
        for s in sizes:
            p1 = F.adaptive_avg_pool2d(eye, (1, s)).view(H, s).t()                                               # (s, H): bin <- position
            u1 = F.interpolate(torch.eye(s).view(s, 1, 1, s), size=(1, H), mode="bilinear", align_corners=True).view(s, H)
            pwx.append(p1), uwx.append(u1)
            for ky in range(s):
                for kx in range(s):
                    pwy.append(p1[ky]), uwy.append(u1[ky]), xq.append(qoff[-1] + kx)
            qoff.append(qoff[-1] + s)
        dv = lambda t: t.contiguous().to(device)
        _PSP_TAPS[key] = dict(aidx=dv(ai.to(torch.int32)), aw=dv(aw), uidx=dv(torch.stack(uis, 1).to(torch.int32)),
                              uw=dv(torch.stack(uws, 1)), nb=off, sizes=tuple(sizes), nq=qoff[-1],
                              pwx=dv(torch.cat(pwx, 0)), pwy=dv(torch.stack(pwy, 0)), uwx=dv(torch.cat(uwx, 0)),
                              uwy=dv(torch.stack(uwy, 0)), xq=dv(torch.tensor(xq, dtype=torch.int32)),
                              qoff=dv(torch.tensor(qoff, dtype=torch.int32)))
    return _PSP_TAPS[key]


def _resize_mask(mask, H, W):
    return mask if mask.shape[-2:] == (H, W) else ops.resize_bilinear(mask, H, W)


class PSPModule(nn.Module):
    """reference model/resnet_generator_app_v2.py:724-752. Pyramid stages are tiny (1..6 px) and stay in
    torch ops; the 528->100 3x3 bottleneck (15 % of generator FLOPs) runs on the MFMA kernel."""

    def __init__(self, features, out_features=100, sizes=(1, 2, 3, 6)):
        super().__init__()
        self.stages = nn.ModuleList([nn.Sequential(nn.AdaptiveAvgPool2d((s, s)), nn.Conv2d(features, out_features, 1, bias=False),
                                                   nn.BatchNorm2d(out_features), nn.ReLU()) for s in sizes])
        self.bottleneck = nn.ModuleList([GemmWeight("conv", out_features, features + len(sizes) * out_features, 3, bias=False, sn=False),
                                         BNState(out_features)])
        self.dropout_p = 0.1

    SIZES = (1, 2, 3, 6)

    def forward(self, feats, pc, sync, join_in=None):
        """join_in: see ConvMaskHead.forward (taken by the pooling branch's backward launch). feats (B,H,W,C) f32. Each stage is AdaptiveAvgPool(s) -> 1x1 conv -> BatchNorm2d -> ReLU -> bilinear
        (align_corners=True) back to HxW; the two resamplings are fixed sparse linear maps over the pixels (psp_taps) applied
        by csrc/psp.hip for all stages at once, the per-stage conv / BN / ReLU between them is one launch (csrc/layout.hip)."""
        B, H, W, C = feats.shape
        taps = psp_taps(H, self.SIZES, feats.device)
        j = ops.GradJoin()   # d feats: the concat branch's part enters the pooling branch's backward launch
        pooled = ops.psp_pool(feats, taps, j, join_in, pc.arena.op_dtype if (join_in is not None and not pc.arena.split) else None)   # (B, 50, C)
        bn_training = self.stages[0][2].training   # (the stage norms' own mode: they are plain nn.BatchNorm2d and may be frozen separately)
        ys = ops.psp_stages(pooled, [st[1] for st in self.stages], [st[2] for st in self.stages], self.SIZES, bn_training)   # (B, 50, F)
        if bn_training and not getattr(self.stages[0][2], "_nbt_shared", False):
            for st in self.stages:
                st[2].num_batches_tracked += 1
        cat = ops.psp_expand(feats, ys, taps, torch.float32 if pc.arena.split else pc.arena.op_dtype, j)                           # (B,H,W,4*100+C), operand dtype
        conv, bn = self.bottleneck
        h = fused_conv(cat, conv, pc, emit=("stats",) if self.training else ())
        spec, w, b = bn.spec(self.training, sync, conv.co_p)
        y = ops.norm_act(h, spec, w, b, emit_op=pc.arena.op_dtype if (pc.arena.op_dtype == torch.bfloat16 and not pc.arena.split) else None)   # (h: read by this layer alone)
        bn.commit(conv.co_p)
        if self.training and self.dropout_p > 0:  # nn.Dropout2d: whole channels per sample (the draws are torch's: same RNG stream as before)
            y = ops.channel_dropout(y, torch.rand(B, 1, 1, y.shape[3], device=y.device).view(B, -1), self.dropout_p)
        return y


JOIN_HEADS = os.environ.get("L2I_JOIN_HEADS", "1") != "0"   # A/B switch: a block result's two gradients (next block, mask head) summed in the head's data-gradient launch
CLASS_GATHER = os.environ.get("L2I_CLASS_GATHER", "1") != "0"   # mask heads compute only the gathered classes (A/B switch)


class ConvMaskHead(nn.ModuleList):
    """conv_mask of the generator ResBlock (reference :643-651). Keys 0,1,3 (plain) or 0,1 (PSP)."""

    def __init__(self, c, psp):
        if psp:
            super().__init__([PSPModule(c, 100), GemmWeight("conv", 184, 100, 1, sn=False)])
        else:
            super().__init__([GemmWeight("conv", 100, c, 3, sn=False), BNState(100), nn.Identity(),
                              GemmWeight("conv", 184, 100, 1, sn=False)])
        self.psp = psp

    def forward(self, x, pc, sync, y=None, join_in=None):
        """join_in (ops.GradJoin): x is also read by the NEXT block, whose backward runs first and leaves its complete
        gradient there; this head's 3x3 data-gradient launch (PSP head: the pooling branch's backward launch) adds it and writes the
        operand copy of the sum.
        y (b, o) int64, o <= 8: the caller will only ever gather the channels of these classes from the head's 184-channel result
        (`seman = gather(m, 1, y)`, reference :465-466) -- the last 1x1 convolution is then evaluated for those classes alone
        (ops.class_logits) and the result is the planar (b, o, H, W) tensor of gathered logits, tagged `_l2i_planar`. None: the dense
        (b, H, W, 184) logits."""
        # (the library's limits -- l2i_class_logits_bwd: C <= 126 input channels, padded <= 128 -- are part of the gate: a head of another hidden
        #  width takes the dense 1x1 convolution instead of failing at backward)
        gather = y is not None and CLASS_GATHER and y.shape[1] <= 8 and self[-1].ci <= 126 and self[-1].ci_p <= 128
        if self.psp:
            a = self[0](x, pc, sync, join_in)
            if not gather:
                return fused_conv(a, self[1], pc)
            out = self[1]
        else:
            conv, bn, _, out = self
            h = fused_conv(x, conv, pc, emit=("stats",) if self.training else (), join_in=join_in, dx_raw=join_in is not None)
            spec, w, b = bn.spec(self.training, sync, conv.co_p)
            if not gather:
                m = fused_conv(h, out, pc, prologue=spec, wproj=w, bproj=b, dx_raw=True)   # (h has this one reader: its gradient's operand copy comes out of the norm backward)
                bn.commit(conv.co_p)
                return m
            a = ops.norm_act(h, spec, w, b, emit_op=pc.arena.op_dtype if pc.arena.op_dtype == torch.bfloat16 and not pc.arena.split else None)
            bn.commit(conv.co_p)
        lg = ops.class_logits(a, ops.arena_weight(out, pc), out.bias, y)
        lg._l2i_planar = True
        return lg


class ResBlock(nn.Module):
    """Generator up-block (reference model/resnet_generator_app_v2.py:628-678)."""

    def __init__(self, in_ch, out_ch, upsample=True, num_w=308, predict_mask=True, psp_module=False):
        super().__init__()
        self.upsample = upsample
        self.conv1 = GemmWeight("conv", out_ch, in_ch, 3, sn=True, eps=1e-4)
        self.conv2 = GemmWeight("conv", out_ch, out_ch, 3, sn=True, eps=1e-4)
        self.b1 = ISLANorm(in_ch, num_w)
        self.b2 = ISLANorm(out_ch, num_w)
        self.learnable_sc = in_ch != out_ch or upsample
        if self.learnable_sc:
            self.c_sc = GemmWeight("conv", out_ch, in_ch, 1, sn=True, eps=1e-4)
        self.predict_mask = predict_mask
        if predict_mask:
            self.conv_mask = ConvMaskHead(out_ch, psp_module)

    def forward(self, x, w, mask, pc, sync, emit=(), y=None, join_x=None, join_next=False):
        """emit: operand copies of the block's result written by conv2's epilogue ("raw": what the next block's shortcut
        conv and this block's mask head read). y: the object classes (see ConvMaskHead.forward).
        join_x: the GradJoin of x's two readers (this block and the previous block's mask head): conv1's backward leaves the block's
        complete dx there for the mask head's data-gradient launch. join_next: create that join for THIS block's result (returned as
        the third value): given to the next block, taken by this block's mask head, conv2 the taker of last resort."""
        B, H, W, C = x.shape
        O = mask.shape[1]
        up = self.upsample
        gw1, gb1 = self.b1.project(w, pc, B, O)
        j = ops.GradJoin()   # the shortcut's dx is accumulated by the second pass of b1's backward instead of a separate add
        h = fused_conv(x, self.conv1, pc, prologue=self.b1.spec(self.training, sync), mask=_resize_mask(mask, H, W).contiguous(),
                       wproj=gw1, bproj=gb1, up2=up, join=(j, "take"), join_out=join_x,
                       emit=("stats",) if self.training else ())   # b2's batch statistics come out of this epilogue
        H2, W2 = h.shape[1], h.shape[2]
        sc = fused_conv(x, self.c_sc, pc, up2=up, join=(j, "give"), lazy_sc=True) if self.learnable_sc else x
        gw2, gb2 = self.b2.project(w, pc, B, O)
        jn = ops.GradJoin() if (join_next and JOIN_HEADS and self.predict_mask and torch.is_grad_enabled()) else None
        out = fused_conv(h, self.conv2, pc, prologue=self.b2.spec(self.training, sync),
                         mask=_resize_mask(mask, H2, W2).contiguous(), wproj=gw2, bproj=gb2, res=sc, emit=tuple(emit) + (("stats",) if self.training else ()),
                         dx_raw=True,   # (the block's result is normalised next: by the following block's b1 or by the final BN)
                         join_src=jn)
        self.b1.batch_norm2d.commit()
        self.b2.batch_norm2d.commit()
        m = self.conv_mask(out, pc, sync, y, join_in=jn) if self.predict_mask else None
        return (out, m, jn) if join_next else (out, m)


class BoxMultiHeadedAttention(nn.Module):
    """reference model/resnet_generator_app_v2.py:123-214 (h = 1, dropout 0). `geometry=False` gives the
    VG variant whose logits ignore the box geometry (model/resnet_generator_vg.py:115)."""

    def __init__(self, h, d_model, geometry=True):
        super().__init__()
        assert h == 1
        self.d = d_model
        self.geometry = geometry
        self.linears = nn.ModuleList([GemmWeight("linear", d_model, d_model, sn=False) for _ in range(4)])
        self.WGs = nn.ModuleList([nn.Linear(64, 1, bias=True)])
        self.layer_norm = nn.LayerNorm(d_model)
        self.layer_norm0 = nn.LayerNorm(d_model)

    def forward(self, w0, bbox, keyvalid, pc, B, O):
        """w0: the latents as a (B*O, 1, 1, Dp) stream (columns D..Dp zero) with its operand copy attached (ops.latent);
        keyvalid (B, O) int32 = label != 0. Returns the context-aware latents in the same form.
        Launches: one grouped GEMM for q / k / v (they read the same rows), the geometry kernel, the attention kernel,
        shuffle + add + LayerNorm, the output projection, add + LayerNorm (csrc/layout.hip) -- the reference's ~65 small ops."""
        D, dp = self.d, self.linears[0].ci_p
        opd = pc.arena.op_dtype
        q, k, v = [_view_keep_sink(t, B, O) for t in ops.grouped_linear(w0, pc.arena.groups["qkv"], pc)]   # column slices of ONE (B*O, 3 Dp) result
        geo = ops.box_geometry(bbox, self.WGs[0].weight, self.WGs[0].bias) if self.geometry else None
        x = ops.box_attention(q, k, v, geo, keyvalid, 1.0 / math.sqrt(D), D)                    # (B, O, D)
        # reference :197-198: with h = 1 the "concat heads" view shuffles each image's (o, d) matrix; kept as is (perm_O)
        out0 = ops.add_layernorm(x, w0, self.layer_norm0, D, dp, opd, perm_O=O)
        o2 = fused_conv(out0, self.linears[3], pc)
        return ops.add_layernorm(o2, out0, self.layer_norm, D, dp, opd)


def _view_keep_sink(t, B, O):
    """(rows, C) slice of a grouped projection -> (B, O, C) view that keeps the gradient-sink coordinates (ops.GradSink)."""
    v = t.view(B, O, -1)
    v._l2i_sink = t._l2i_sink
    return v


class MaskRegressNetv2(nn.Module):
    """reference model/mask_regression.py:58-102 (256 ch, InstanceNorm). v1 (:11-55): 128 ch, SyncBN."""

    def __init__(self, obj_feat=308, mask_size=16, map_size=64, ch=256, instance=True):
        super().__init__()
        self.mask_size, self.map_size, self.ch, self.instance = mask_size, map_size, ch, instance
        self.fc = GemmWeight("linear", ch * 16, obj_feat, sn=True, eps=1e-12)

        def block(extra):
            mods = [GemmWeight("conv", ch, ch, 3, sn=True, eps=1e-12), nn.Identity() if instance else BNState(ch), nn.Identity()]
            return nn.ModuleList(mods + extra)
        self.conv1, self.conv2 = block([]), block([])
        self.conv3 = block([GemmWeight("conv", 1, ch, 1, sn=True, eps=1e-12), nn.Identity()])

    def _spec(self, blk, sync):
        if self.instance:
            return NormSpec(2, eps=1e-5, relu=True, instance=True), None, None
        return blk[1].spec(self.training, sync)

    def forward(self, w, bbox, pc, sync, want_boxm=False):
        """-> (regressed masks (b, o, map, map), hard box masks (b, o, map, map) | None)."""
        b, o, _ = bbox.shape
        N = b * o
        x = ops.fc_to_nhwc(fused_conv(w, self.fc, pc), self.ch, pc.arena.op_dtype)            # (N, 4, 4, ch)
        h = fused_conv(x, self.conv1[0], pc)
        for blk_prev, blk, size in ((self.conv1, self.conv2, 8), (self.conv2, self.conv3, 16)):
            if self.instance:   # InstanceNorm -> ReLU -> bilinear x2 + the operand copy: one launch (csrc/layout.hip)
                h = fused_conv(ops.in_relu_up2(h, 1e-5, pc.arena.op_dtype), blk[0], pc)
                continue
            spec, wa, ba = self._spec(blk_prev, sync)
            a = ops.norm_act(h, spec, wa, ba)
            if not self.instance:
                blk_prev[1].commit()
            a = ops.up2_nhwc(a.view(N, size // 2, size // 2, self.ch), pc.arena.op_dtype)   # bilinear x2 + the operand copy: one launch (csrc/layout.hip)
            h = fused_conv(a, blk[0], pc)
        spec, wa, ba = self._spec(self.conv3, sync)
        m = fused_conv(h, self.conv3[3], pc, prologue=spec, wproj=wa, bproj=ba, dx_raw=True)   # (N, 16, 16, 8): channel 0 = the logits
        if not self.instance:
            self.conv3[1].commit()
        # sigmoid + masks_to_layout + bbox_mask: one launch (csrc/layout.hip)
        return ops.layout_masks(m, bbox, self.map_size, want_boxm)


def _with_zero_pool(fwd):
    """Run a forward inside a zero-pool step of its own when the caller (GanTrainer.step) has not opened one: the forward's
    ~30 small accumulation targets (batch statistics, ...) then come out of ONE pre-zeroed slab instead of one fill launch
    each (sampling, the generator-forward benchmark). Only without an autograd tape: slices of the slab (batch statistics)
    are saved for backward, and the NEXT stand-alone forward re-zeroes the slab -- two forwards alive before one backward
    (f1 = G(z1); f2 = G(z2); loss(f1, f2).backward()) would hand the first graph zeroed statistics. With gradients enabled
    a stand-alone forward therefore takes its buffers from torch.zeros as before."""
    import functools

    @functools.wraps(fwd)
    def run(self, z, *a, **k):
        if ops.POOL.active or not z.is_cuda or torch.is_grad_enabled():
            return fwd(self, z, *a, **k)
        with ops.POOL.step(z.device, nfloats=2 << 20):
            return fwd(self, z, *a, **k)
    return run


class _GeneratorBase(nn.Module):
    """Shared plumbing: flat parameters, weight arena, SyncBN hook, state_dict layout."""

    def finalize(self, device, op_dtype=torch.bfloat16):
        """op_dtype: torch.bfloat16 (MFMA operands in bf16: the throughput mode), torch.float32 (exact-f32 MFMA), or "bf16x3" --
        bf16 operands carried as hi + lo with three MFMA products per pair (arena split; forward-only: sampling and the parity of
        the forward against the reference to the 1e-3 image bar at MFMA speed)."""
        split = op_dtype == "bf16x3"
        if split:
            op_dtype = torch.bfloat16
        self.op_dtype = op_dtype
        # the 2 x (number of ISLA layers) projections Linear(num_w -> C) (model/norm_module.py:158-159) all read the same
        # object latents: they run as ONE grouped GEMM per pass (arena.GemmGroup / ops.GroupedLinearFn)
        self._isla = [m for m in self.modules() if isinstance(m, ISLANorm)]
        for n in self._isla:
            n.weight_proj.group = n.bias_proj.group = "isla"
        for l in self.context.linears[:3]:   # q, k, v projections of the context attention read the same rows: one GEMM
            l.group = "qkv"
        self.flat = FlatParams(self, device)
        self.arena = WeightArena(self, self.flat, device, op_dtype, split=split)
        self.sync = None  # set by parallel.attach_sync_bn for world_size > 1
        # every batch-norm layer of the network runs in every forward: their num_batches_tracked counters live in one
        # tensor (each buffer a 0-dim view of it) and a training forward bumps them with ONE launch (_bump_nbt)
        bns = [m for m in self.modules() if isinstance(m, (BNState, nn.BatchNorm2d))]
        self._nbt = torch.zeros(len(bns), dtype=torch.long, device=device)
        with torch.no_grad():
            for i, m in enumerate(bns):
                self._nbt[i] = m.num_batches_tracked
                m.num_batches_tracked.data = self._nbt[i]
                m._nbt_shared = True
        return self

    def _bump_nbt(self):
        if self.training:
            self._nbt += 1

    def _project_isla(self, wp, pc, b, o):
        """One GEMM for every ISLA projection of the pass; each norm layer picks up its (b, o, C) slices."""
        outs = ops.grouped_linear(wp if wp.dim() == 4 else wp.view(b * o, 1, 1, -1), self.arena.groups["isla"], pc)   # (keeps wp's operand copy)
        for i, n in enumerate(self._isla):
            gw, gb = outs[2 * i], outs[2 * i + 1]
            sw, sb = gw._l2i_sink, gb._l2i_sink
            gw, gb = gw.view(b, o, -1), gb.view(b, o, -1)
            gw._l2i_sink, gb._l2i_sink = sw, sb
            n._proj = (gw, gb)

    def _release_isla(self):
        for n in self._isla:
            n._proj = None

    def init_parameter(self):
        """reference :501-506: orthogonal_ on every parameter with dim > 1, zeros on '*bias'."""
        for name, p in self.named_parameters():
            if p.dim() > 1:
                torch.nn.init.orthogonal_(p)
            if name[-4:] == "bias":
                torch.nn.init.constant_(p, 0)

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()  # pending pass contexts stay: they are consumed by FlatAdam.step()

    def _latent(self, z, y, ld):
        """[z | label_embedding(y)] as a (b*o, 1, 1, ld) stream with its operand copy, and the attention key mask (y != 0)."""
        return ops.latent(z, self.label_embedding.weight, y, ld, self.op_dtype)


class ResnetGenerator128_context(_GeneratorBase):
    def __init__(self, ch=64, z_dim=128, num_classes=10, output_dim=3):
        super().__init__()
        self.num_classes, self.ch, self.output_dim = num_classes, ch, output_dim
        self.label_embedding = nn.Embedding(num_classes, 180)
        num_w = 128 + 180
        self.context = BoxMultiHeadedAttention(1, num_w)
        self.fc = GemmWeight("linear", 4 * 4 * 16 * ch, z_dim, sn=True, eps=1e-12)
        self.res1 = ResBlock(ch * 16, ch * 16, num_w=num_w)
        self.res2 = ResBlock(ch * 16, ch * 8, num_w=num_w)
        self.res3 = ResBlock(ch * 8, ch * 4, num_w=num_w)
        self.res4 = ResBlock(ch * 4, ch * 2, num_w=num_w, psp_module=True)
        self.res5 = ResBlock(ch * 2, ch * 1, num_w=num_w, predict_mask=False)
        self.final = nn.ModuleList([BNState(ch), nn.Identity(), GemmWeight("conv", output_dim, ch, 3, sn=True, eps=1e-4), nn.Identity()])
        self.alpha1 = nn.Parameter(torch.zeros(1, 184, 1))
        self.alpha2 = nn.Parameter(torch.zeros(1, 184, 1))
        self.alpha3 = nn.Parameter(torch.zeros(1, 184, 1))
        self.alpha4 = nn.Parameter(torch.zeros(1, 184, 1))
        self.mask_regress = MaskRegressNetv2(num_w)
        self.init_parameter()

    def _stage_mask(self, stage_logits, bmask, bbox_mask_, alpha, y):
        """reference :465-470: blend the regressed mask with the predicted semantic mask."""
        b, o = y.shape
        S = bmask.shape[-1]
        if getattr(stage_logits, "_l2i_planar", False):   # class-gathered logits (b, o, H, W) of ConvMaskHead
            H, W = stage_logits.shape[2], stage_logits.shape[3]
            if H == W and S % H == 0 and (S == H or (S // H) % 2 == 0):
                return ops.stage_mask(stage_logits, bmask, bbox_mask_, alpha, y, planar=True)
            seman = torch.sigmoid(stage_logits) * F.interpolate(bbox_mask_, size=(H, W), mode="nearest")
            a = torch.gather(torch.sigmoid(alpha).expand(b, -1, -1), dim=1, index=y.view(b, o, 1)).unsqueeze(-1)
            return (_resize_mask(bmask, H, W) * (1 - a) + seman * a).contiguous()
        B, H, W, Cp = stage_logits.shape
        if H == W and Cp % 4 == 0 and alpha.numel() == Cp and S % H == 0 and (S == H or (S // H) % 2 == 0):
            return ops.stage_mask(stage_logits, bmask, bbox_mask_, alpha, y)   # the fused form of the lines below
        idx = y.view(b, 1, 1, o).expand(b, H, W, o)
        seman = torch.sigmoid(torch.gather(stage_logits, 3, idx)).permute(0, 3, 1, 2)
        seman = seman * F.interpolate(bbox_mask_, size=(H, W), mode="nearest")
        a = torch.gather(torch.sigmoid(alpha).expand(b, -1, -1), dim=1, index=y.view(b, o, 1)).unsqueeze(-1)
        return (_resize_mask(bmask, H, W) * (1 - a) + seman * a).contiguous()

    @_with_zero_pool
    def forward(self, z, bbox, z_im=None, y=None, taps=None):
        if not z.is_cuda:
            raise RuntimeError("layout2img_amd generators run on the GPU HIP path only")
        b, o = z.size(0), z.size(1)
        bbox = bbox.to(z.device).float()
        pc = self.arena.prepare(training=self.training)
        w0, keyvalid = self._latent(z, y, self.context.linears[0].ci_p)
        wp = self.context(w0, bbox, keyvalid, pc, b, o)                                          # (b*o, 1, 1, 312) + operand copy
        w = wp.view(b, o, -1)[..., :self.context.d]
        self._project_isla(wp, pc, b, o)
        bmask, bbox_mask_ = self.mask_regress(wp, bbox, pc, self.sync, want_boxm=True)
        if z_im is None:
            z_im = torch.randn((b, 128), device=z.device)
        x = ops.fc_to_nhwc(fused_conv(z_im.reshape(b, 1, 1, -1).contiguous(), self.fc, pc), 16 * self.ch, self.op_dtype)
        x, m, jx = self.res1(x, wp, bmask, pc, self.sync, emit=("raw",), y=y, join_next=True)
        stage = bmask
        stages = []
        res_out = [x]
        for blk, alpha in ((self.res2, self.alpha1), (self.res3, self.alpha2), (self.res4, self.alpha3), (self.res5, self.alpha4)):
            stage = self._stage_mask(m, bmask, bbox_mask_, alpha, y)
            stages.append(stage)
            x, m, jx = blk(x, wp, stage, pc, self.sync, emit=() if blk is self.res5 else ("raw",), y=y, join_x=jx, join_next=True)
            res_out.append(x)
        bn, _, conv, _ = self.final
        spec, wa, ba = bn.spec(self.training, self.sync)
        pre = fused_conv(x, conv, pc, prologue=spec, wproj=wa, bproj=ba, dx_raw=True)   # (x: the last block's result, read by this layer alone)
        bn.commit()
        self._release_isla()
        self._bump_nbt()
        img = ops.tanh_nchw(pre, self.output_dim, self.op_dtype)
        if taps is not None:
            taps.update(w=w, bmask=bmask, stages=stages, pre_tanh=pre[..., :self.output_dim], res=res_out)
        return img


class context_aware_generator(_GeneratorBase):
    """VG generator (reference model/resnet_generator_vg.py:639-727): context attention without the geometry
    bias, MaskRegressNet (128 ch, SyncBN), the same regressed mask for every block, no mask heads."""

    def __init__(self, ch=64, z_dim=128, num_classes=10, output_dim=3):
        super().__init__()
        self.num_classes, self.ch, self.output_dim = num_classes, ch, output_dim
        self.label_embedding = nn.Embedding(num_classes, 180)
        num_w = 128 + 180
        self.context = BoxMultiHeadedAttention(1, num_w, geometry=False)
        self.fc = GemmWeight("linear", 4 * 4 * 16 * ch, z_dim, sn=True, eps=1e-12)
        chans = [(16, 16), (16, 8), (8, 4), (4, 2), (2, 1)]
        for i, (a, b_) in enumerate(chans, 1):
            setattr(self, f"res{i}", ResBlock(ch * a, ch * b_, num_w=num_w, predict_mask=False))
        self.final = nn.ModuleList([BNState(ch), nn.Identity(), GemmWeight("conv", output_dim, ch, 3, sn=True, eps=1e-4), nn.Identity()])
        self.mask_regress = MaskRegressNetv2(num_w, ch=128, instance=False)
        self.init_parameter()

    @_with_zero_pool
    def forward(self, z, bbox, z_im=None, y=None):
        if not z.is_cuda:
            raise RuntimeError("layout2img_amd generators run on the GPU HIP path only")
        b, o = z.size(0), z.size(1)
        bbox = bbox.to(z.device).float()
        pc = self.arena.prepare(training=self.training)
        w0, keyvalid = self._latent(z, y, self.context.linears[0].ci_p)
        wp = self.context(w0, bbox, keyvalid, pc, b, o)
        self._project_isla(wp, pc, b, o)
        mask, _ = self.mask_regress(wp, bbox, pc, self.sync)
        if z_im is None:
            z_im = torch.randn((b, 128), device=z.device)
        x = ops.fc_to_nhwc(fused_conv(z_im.reshape(b, 1, 1, -1).contiguous(), self.fc, pc), 16 * self.ch, self.op_dtype)
        for i in range(1, 6):
            x, _ = getattr(self, f"res{i}")(x, wp, mask, pc, self.sync, emit=("raw",) if i < 5 else ())
        bn, _, conv, _ = self.final
        spec, wa, ba = bn.spec(self.training, self.sync)
        pre = fused_conv(x, conv, pc, prologue=spec, wproj=wa, bproj=ba, dx_raw=True)   # (x: the last block's result, read by this layer alone)
        bn.commit()
        self._release_isla()
        self._bump_nbt()
        return ops.tanh_nchw(pre, self.output_dim, self.op_dtype)


class ResnetGenerator64_context(ResnetGenerator128_context):
    """64x64 generator for BASELINE configs 1-2. The reference has no working 64x64 `app_v2` generator (SURVEY.md
    section 0, fact 10); this follows its own pattern for 64x64 models (model/resnet_generator_vg.py:301-355: drop
    `res1`, blocks named res2..res5) with the app_v2 blocks: fc -> 4x4x16ch -> four up-blocks at 8/16/32/64 px, mask
    heads on the first three (PSP on the second-to-last, as :411-415 do for their last two blocks), none on the last;
    the 64x64 regressed masks and rectangle indicators are unchanged. Parity for this class is block-level (every
    block is the 128x128 model's, checked against the reference goldens) plus an end-to-end test (tests/test_gpu_05_res64.py)."""

    def __init__(self, ch=64, z_dim=128, num_classes=10, output_dim=3):
        _GeneratorBase.__init__(self)
        self.num_classes, self.ch, self.output_dim = num_classes, ch, output_dim
        self.label_embedding = nn.Embedding(num_classes, 180)
        num_w = 128 + 180
        self.context = BoxMultiHeadedAttention(1, num_w)
        self.fc = GemmWeight("linear", 4 * 4 * 16 * ch, z_dim, sn=True, eps=1e-12)
        self.res2 = ResBlock(ch * 16, ch * 8, num_w=num_w)
        self.res3 = ResBlock(ch * 8, ch * 4, num_w=num_w)
        self.res4 = ResBlock(ch * 4, ch * 2, num_w=num_w, psp_module=True)
        self.res5 = ResBlock(ch * 2, ch * 1, num_w=num_w, predict_mask=False)
        self.final = nn.ModuleList([BNState(ch), nn.Identity(), GemmWeight("conv", output_dim, ch, 3, sn=True, eps=1e-4), nn.Identity()])
        self.alpha1 = nn.Parameter(torch.zeros(1, 184, 1))
        self.alpha2 = nn.Parameter(torch.zeros(1, 184, 1))
        self.alpha3 = nn.Parameter(torch.zeros(1, 184, 1))
        self.mask_regress = MaskRegressNetv2(num_w)
        self.init_parameter()

    @_with_zero_pool
    def forward(self, z, bbox, z_im=None, y=None, taps=None):
        if not z.is_cuda:
            raise RuntimeError("layout2img_amd generators run on the GPU HIP path only")
        b, o = z.size(0), z.size(1)
        bbox = bbox.to(z.device).float()
        pc = self.arena.prepare(training=self.training)
        w0, keyvalid = self._latent(z, y, self.context.linears[0].ci_p)
        wp = self.context(w0, bbox, keyvalid, pc, b, o)
        self._project_isla(wp, pc, b, o)
        bmask, bbox_mask_ = self.mask_regress(wp, bbox, pc, self.sync, want_boxm=True)
        if z_im is None:
            z_im = torch.randn((b, 128), device=z.device)
        x = ops.fc_to_nhwc(fused_conv(z_im.reshape(b, 1, 1, -1).contiguous(), self.fc, pc), 16 * self.ch, self.op_dtype)
        x, m, jx = self.res2(x, wp, bmask, pc, self.sync, emit=("raw",), y=y, join_next=True)
        for blk, alpha in ((self.res3, self.alpha1), (self.res4, self.alpha2), (self.res5, self.alpha3)):
            stage = self._stage_mask(m, bmask, bbox_mask_, alpha, y)
            x, m, jx = blk(x, wp, stage, pc, self.sync, emit=() if blk is self.res5 else ("raw",), y=y, join_x=jx, join_next=True)
        bn, _, conv, _ = self.final
        spec, wa, ba = bn.spec(self.training, self.sync)
        pre = fused_conv(x, conv, pc, prologue=spec, wproj=wa, bproj=ba, dx_raw=True)   # (x: the last block's result, read by this layer alone)
        bn.commit()
        self._release_isla()
        self._bump_nbt()
        return ops.tanh_nchw(pre, self.output_dim, self.op_dtype)
