"""RCNN-style discriminator with appearance head on the HIP path.

Mirrors `CombineDiscriminator128_app` / `ResnetDiscriminator128_app`
(reference model/rcnn_discriminator_app.py:84-168, 294-344, 396-421; rcnn_discriminator_vg.py is
byte-identical): same class names, constructor arguments, call signature, outputs and state_dict
keys. All 3x3/1x1 convolutions run on the MFMA implicit-GEMM kernel (ReLU prologue, avg-pool
epilogue), ROIAlign is one HIP launch over a FIXED R = b*o rows with a validity mask (no
host-synchronising nonzero()), the small heads use device-side torch ops on arena weights.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .arena import DualPass, FlatParams, GemmWeight, WeightArena
from .ops import RELU, arena_weight, fused_conv


def _passes(pc):
    """The pass contexts the heads run under: one, or the two of a dual pass (rows [0, n) belong to the first)."""
    return (pc.a, pc.b) if pc.dual else (pc,)


def _rows(t, pc):
    """t's rows per pass: the tensor itself, or its two halves (one concatenation backward, ops.split_halves)."""
    if not pc.dual:
        return (t,)
    return ops.split_halves(t) if t.requires_grad else (t[:t.shape[0] // 2], t[t.shape[0] // 2:])


JOIN_READERS = __import__("os").environ.get("L2I_JOIN_READERS", "1") != "0"   # A/B switch: two-reader tensors summed inside a data-gradient launch


def _conv(ci, co, k, uses=1):
    return GemmWeight("conv", co, ci, k, sn=True, eps=1e-4, uses=uses)


class OptimizedBlock(nn.Module):
    """reference :294-314 -- the shortcut pools BEFORE its 1x1 conv."""

    def __init__(self, in_ch, out_ch, downsample=False):
        super().__init__()
        self.conv1, self.conv2, self.c_sc = _conv(in_ch, out_ch, 3), _conv(out_ch, out_ch, 3), _conv(in_ch, out_ch, 1)
        self.downsample = downsample

    def forward(self, x, pc, emit=(), f32_dead=False):
        h = fused_conv(x, self.conv1, pc, relu_op_out=True)   # conv2's ReLU'd operand comes out of conv1's epilogue
        xs = x
        if self.downsample:
            xs = getattr(x, "_l2i_half", None)   # the 2x2 average made by the input-image kernel (ops.image_nhwc)
            if xs is None:
                xs = F.avg_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).contiguous()
        sc = fused_conv(xs, self.c_sc, pc)
        return fused_conv(h, self.conv2, pc, prologue=RELU, res=sc, pool2=self.downsample, emit=emit, dx_raw=True, f32_dead=f32_dead)


class ResBlock(nn.Module):
    """reference :317-344 -- pre-activation block; the shortcut pools AFTER its 1x1 conv."""

    def __init__(self, in_ch, out_ch, downsample=False, uses=1):
        super().__init__()
        self.conv1, self.conv2 = _conv(in_ch, out_ch, 3, uses), _conv(out_ch, out_ch, 3, uses)
        self.downsample = downsample
        self.learnable_sc = (in_ch != out_ch) or downsample
        if self.learnable_sc:
            self.c_sc = _conv(in_ch, out_ch, 1, uses)

    def forward(self, x, pc, use=0, nimg=None, emit=(), sole_reader=False, join_in=None, join_out=None, join_src=None, f32_dead=False):
        """`f32_dead`: every reader of this block's result is a block with a learnable shortcut (reads the emitted copies only): ops.fused_conv.
        `sole_reader`: x is the result of another block's conv2 and this block is its ONLY reader -- the data-gradient launch
        of conv1 (which also takes the shortcut branch's gradient as its residual: the complete dx) then writes the operand
        copy of dx that the producing conv2's backward needs, instead of a separate cast pass over the f32 gradient.
        `join_in` / `join_out` (ops.fused_conv): x is read by TWO blocks -- the one created later (its backward runs first) leaves
        its complete dx in the shared GradJoin (join_out, conv1's launch), this block's shortcut launch adds it (join_in), so the sum
        over both readers comes out of conv1's data-gradient launch (with `sole_reader` its operand copy as well).
        `join_src`: the GradJoin of the two readers of THIS block's result: conv2's backward adds a gradient that was parked
        there for a launch that never ran (a loss without the image head: ops.GradJoin.leftover).
        `use`: index of this application within the forward pass (each application of a spectral-normed
        module runs its own power iteration in the reference). `nimg`: device count of live leading images (ROI heads).
        `emit`: operand copies of the block's result its consumers will read ("relu" / "raw"), written by conv2's
        epilogue. Inside the block conv1's epilogue writes conv2's ReLU'd operand and conv2's data-gradient launch
        writes dh in the operand dtype -- on all but the smallest maps that operand tensor IS the autograd edge between
        the two convs, so neither h nor dh exists as an f32 stream (ops.fused_conv relu_op_out)."""
        if self.learnable_sc:
            ops.precast(x, pc.arena.op_dtype)   # conv1 reads relu(x), the shortcut reads x: one cast launch for both (if not emitted upstream)
        j = ops.GradJoin()   # dx of the shortcut branch enters conv1's data-gradient epilogue instead of a separate add
        # ... and, round 5, is not even a launch of its own: conv1's data-gradient launch multiplies the shortcut's operands into its own
        # accumulators behind the ReLU mask (ops.FusedConvFn.backward (a) / (b), csrc conv_mask_first + conv_sc_tail)
        j.fold_ok = self.learnable_sc
        h = fused_conv(x, self.conv1.use(use), pc, prologue=RELU, nimg=nimg, relu_op_out=True, join=(j, "take"), dx_raw=sole_reader,
                       join_out=join_out)
        sc = (fused_conv(x, self.c_sc.use(use), pc, pool2=self.downsample, nimg=nimg, join=(j, "give"), lazy_sc=True, join_in=join_in)
              if self.learnable_sc else x)
        return fused_conv(h, self.conv2.use(use), pc, prologue=RELU, res=sc, pool2=self.downsample, nimg=nimg, emit=emit,
                          dx_raw=True, join=None if self.learnable_sc else (j, "give_res"), join_src=join_src, f32_dead=f32_dead)


class ResnetDiscriminator128_app(nn.Module):
    def __init__(self, num_classes=0, input_dim=3, ch=64):
        super().__init__()
        self.num_classes, self.ch = num_classes, ch
        self.block1 = OptimizedBlock(input_dim, ch, downsample=True)
        self.block2 = ResBlock(ch, ch * 2, downsample=True)
        self.block3 = ResBlock(ch * 2, ch * 4, downsample=True)
        self.block4 = ResBlock(ch * 4, ch * 8, downsample=True)
        self.block5 = ResBlock(ch * 8, ch * 16, downsample=True)
        self.block6 = ResBlock(ch * 16, ch * 16, downsample=False)
        self.l7 = GemmWeight("linear", 1, ch * 16, sn=True)
        self.block_obj3 = ResBlock(ch * 2, ch * 4, downsample=False)
        self.block_obj4 = ResBlock(ch * 4, ch * 8, downsample=False, uses=2)  # applied to x1-path and x2
        self.block_obj5 = ResBlock(ch * 8, ch * 16, downsample=True)
        self.l_obj = GemmWeight("linear", 1, ch * 16, sn=True)
        self.l_y = GemmWeight("embedding", num_classes, ch * 16, bias=False, sn=True)
        self.app_conv = ResBlock(ch * 8, ch * 8, downsample=False)
        self.l_y_app = GemmWeight("embedding", num_classes, ch * 8, bias=False, sn=True)
        self.app = GemmWeight("linear", 1, ch * 16, sn=True)

    def forward(self, x, y, rois, valid, pc, nimg=None):
        """x (b,H,W,8) f32 NHWC (3 real channels); y (R,) labels; rois (R,5); valid (R,) int32; nimg: device int32 =
        number of valid rows when they are compacted to the front (the ROI heads then skip the rest)."""
        both = ("relu", "raw")   # what a following block with a learnable shortcut reads
        # (f32_dead: each of these results is read by blocks with a learnable shortcut only -- conv1 takes "relu", the shortcut "raw")
        x = self.block1(x, pc, emit=both, f32_dead=True)
        jx1, jx2 = ops.GradJoin(), ops.GradJoin()   # x1 and x2 are each read by a trunk block and by an object-path block
        x1 = self.block2(x, pc, emit=both, sole_reader=True, join_src=jx1, f32_dead=True)
        x2 = self.block3(x1, pc, emit=both, join_in=jx1, sole_reader=JOIN_READERS, join_src=jx2, f32_dead=True)
        x = self.block4(x2, pc, emit=both, join_in=jx2, sole_reader=JOIN_READERS, f32_dead=True)
        x = self.block5(x, pc, emit=("relu",), sole_reader=True)
        x = self.block6(x, pc, sole_reader=True)
        P = _passes(pc)
        out_im = [ops.proj_head(xk, self.l7, p) for xk, p in zip(_rows(x, pc), P)]            # l7(sum_hw relu(x))

        feat_s = self.block_obj4(self.block_obj3(x1, pc, emit=both, join_out=jx1 if JOIN_READERS else None, f32_dead=True), pc, use=0, sole_reader=True)   # reference order :136-141
        feat_l = self.block_obj4(x2, pc, use=1, join_out=jx2 if JOIN_READERS else None)
        # the ROI features are read by block_obj5 and by app_conv: the one created LATER (its backward runs first) leaves its complete
        # gradient in jobj, block_obj5's shortcut launch adds it; ROI-align's backward is the taker of last resort (a loss over d_app alone)
        jobj = ops.GradJoin() if (JOIN_READERS and torch.is_grad_enabled()) else None
        obj = ops.roi_align(feat_s, feat_l, rois, valid, 8, 1.0 / 4.0, 1.0 / 8.0, 64.0, 0, join_src=jobj,
                            emit_op=pc.arena.op_dtype if not pc.arena.split else None)  # (R,8,8,C) + both operand copies

        # appearance head (reference :148-157): Gram of the ROI features + class embedding
        if not pc.arena.split:   # both operand copies of the ROI features in ONE launch: app_conv's conv1 reads relu(obj), block_obj5 relu(obj) and obj
            ops.precast(obj, pc.arena.op_dtype)
        # projection head (reference :160-166): l_obj(f) + sum(l_y(y) * f), f = sum_hw relu(block_obj5(obj))
        # (called before app_conv, reference :148-166 the other way round: the results do not depend on the order, the join above does)
        f5 = self.block_obj5(obj, pc, nimg=nimg, join_in=jobj)
        a = self.app_conv(obj, pc, nimg=nimg, join_out=jobj)              # (R, 8, 8, C) pre-ReLU
        s2 = a.shape[3]
        out_app, out_obj = [], []
        for ak, fk, yk, p in zip(_rows(a, pc), _rows(f5, pc), _rows(y, pc), P):
            wa = arena_weight(self.app, p)                                # (1, 2C)
            # w1 . sum_rows(Gram) / C with Gram = F F^T / C, F = relu(a): only this contraction of the (R,C,C) Gram
            # matrices reaches the output, so they are never formed (ops.GramHeadFn / csrc/misc.hip); the class-embedding
            # half of the head, l_y_app(y) . w2 + bias, is ops.emb_dot
            gram_term = ops.gram_head(ak.contiguous(), wa[0, :s2].contiguous())
            out_app.append(gram_term + ops.emb_dot(self.l_y_app, yk, self.app, s2, p))
            out_obj.append(ops.proj_head(fk, self.l_obj, p, emb=self.l_y, y=yk))
        if pc.dual:   # ((d_img, d_obj, d_app) of the first pass, the same of the second)
            return tuple((out_im[k], out_obj[k], out_app[k]) for k in range(2))
        return out_im[0], out_obj[0], out_app[0]


class CombineDiscriminator128_app(nn.Module):
    """netD(images (b,3,H,W), bbox (b,o,4) xywh in [0,1], label (b,o,1)|(b,o)) ->
    (d_img (b,1), d_obj (R,1), d_app (R,1)) with R = number of label != 0 rows, ordered large ROIs
    first then small, original order within each (reference :145-146, 401-421). `bbox` is never
    modified. `forward_padded` is the sync-free form used by the training loop."""

    def __init__(self, num_classes=81):
        super().__init__()
        self.obD = ResnetDiscriminator128_app(num_classes=num_classes, input_dim=3)

    def finalize(self, device, op_dtype=torch.bfloat16):
        """op_dtype: torch.bfloat16 | torch.float32 | "bf16x3" (forward-only split operands, see the generators' finalize)."""
        split = op_dtype == "bf16x3"
        if split:
            op_dtype = torch.bfloat16
        self.op_dtype = op_dtype
        self.flat = FlatParams(self, device)
        self.arena = WeightArena(self, self.flat, device, op_dtype, split=split)
        return self

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()  # pending pass contexts stay: they are consumed by FlatAdam.step()

    two_scale = True   # ROIs >= 64 px in width or height pool from the coarse map and come first (reference :131-146)

    def prepare_layout(self, bbox, label, size, device):
        """xywh in [0,1] -> (R,5) pixel ROIs with batch index, flat labels, validity and the device-side count, with the
        R = b*o rows COMPACTED: real ROIs first, in the reference's output order (large ROIs, then small ones, original
        order within each, :145-146), padding rows (label 0) behind them. Reference :402-417 without the
        host-synchronising nonzero(): the count stays on the device, the ROI heads read it there (`nimg`).
        The layout depends on (bbox, label) only: GanTrainer computes it once per iteration for the three D passes."""
        return ops.roi_layout(bbox.to(device).float(), label.to(device), float(size), self.two_scale)   # one launch (csrc/layout.hip)

    def _prepare(self, images, bbox, label, layout=None):
        """-> (rois, y, valid, count) as prepare_layout gives them, and the NHWC image padded to 8 channels."""
        if layout is None:
            layout = self.prepare_layout(bbox, label, images.size(2), images.device)
        # padded NHWC stream + operand copy (+ the 2x2 average the first block's shortcut reads): one launch
        x, xs = ops.image_nhwc(images, 8, self.op_dtype, bool(getattr(self.obD.block1, "downsample", False)))
        if xs is not None:
            x._l2i_half = xs
        return (*layout, x)

    def forward_padded(self, images, bbox, label, need_wgrad=True, pc=None, layout=None):
        """The sync-free form: (d_img (b,1), d_obj (R,1), d_app (R,1), valid (R,), rois (R,5)) over the fixed R = b*o rows
        in compacted order (prepare_layout); rows with valid == 0 are padding.
        pc: a pass context already prepared for this pass (GanTrainer prepares the fake pass's weights on the side
        stream while the generator is still running), else one is prepared here."""
        if not images.is_cuda:
            raise RuntimeError("layout2img_amd discriminators run on the GPU HIP path only")
        rois, y, valid, count, x = self._prepare(images, bbox, label, layout)
        if pc is None:
            pc = self.arena.prepare(training=self.training, need_wgrad=need_wgrad)
        d_img, d_obj, d_app = self.obD(x, y, rois, valid, pc, nimg=count)
        return d_img, d_obj, d_app, valid, rois

    def forward_dual(self, images_a, images_b, bbox, label, pcs=None, layout=None, need_wgrad=True):
        """TWO passes over the same layout as one batch of 2b images -- the discriminator step's D(real) and D(fake)
        (reference train_context_app_v2.py:158,167): -> (outs_a, outs_b, valid, rois), outs_k = forward_padded's outputs of pass
        k. Each pass keeps what the reference gives it: its own power iteration, W / sigma packs and weight-gradient
        accumulator (arena.DualPass; pcs = the two prepared pass contexts, first pass first, else prepared here in that
        order); every convolution, data-gradient and weight-gradient launch processes both passes' tiles at once."""
        if not (images_a.is_cuda and images_b.is_cuda):
            raise RuntimeError("layout2img_amd discriminators run on the GPU HIP path only")
        if images_a.shape != images_b.shape:
            raise RuntimeError("forward_dual: the two passes must have the same batch shape")
        b = images_a.shape[0]
        rois, y, valid, count, x = self._prepare(torch.cat((images_a, images_b)), bbox, label, layout)
        R = rois.shape[0]
        rois2, y2, valid2 = torch.cat((rois, rois)), torch.cat((y, y)), torch.cat((valid, valid))
        rois2[R:, 0] += b                                                # the second pass's images follow the first's
        if pcs is None:
            pcs = (self.arena.prepare(training=self.training, need_wgrad=need_wgrad),
                   self.arena.prepare(training=self.training, need_wgrad=need_wgrad))
        outs_a, outs_b = self.obD(x, y2, rois2, valid2, DualPass(*pcs), nimg=count)
        return outs_a, outs_b, valid, rois

    def obD_arena(self):
        return self.arena

    def forward(self, images, bbox, label, mask=None):
        d_img, d_obj, d_app, valid, rois = self.forward_padded(images, bbox, label)
        n = int(valid.sum())   # (host sync: the module boundary returns the reference's dynamic shape)
        return d_img, d_obj[:n], d_app[:n]


class ResnetDiscriminator64(nn.Module):
    """reference model/rcnn_discriminator_orig.py:83-135 (the only working 64x64 discriminator):
    single-scale RoIAlign((8,8), 1/2, 0) on the 32x32 map, MEAN pooling on the image path (:117), orthogonal init."""

    def __init__(self, num_classes=0, input_dim=3, ch=64):
        super().__init__()
        self.block1 = OptimizedBlock(input_dim, ch, downsample=False)
        self.block2 = ResBlock(ch, ch * 2, downsample=False)
        self.block3 = ResBlock(ch * 2, ch * 4, downsample=True)
        self.block4 = ResBlock(ch * 4, ch * 8, downsample=True)
        self.block5 = ResBlock(ch * 8, ch * 16, downsample=True)
        self.l_im = GemmWeight("linear", 1, ch * 16, sn=True)
        self.block_obj4 = ResBlock(ch * 4, ch * 8, downsample=True)
        self.l_obj = GemmWeight("linear", 1, ch * 8, sn=True)
        self.l_y = GemmWeight("embedding", num_classes, ch * 8, bias=False, sn=True)
        for name, p in self.named_parameters():   # :130-135
            if p.dim() > 1:
                nn.init.orthogonal_(p)
            if name[-4:] == "bias":
                nn.init.constant_(p, 0)

    def forward(self, x, y, rois, valid, pc, nimg=None):
        both = ("relu", "raw")
        x = self.block1(x, pc, emit=both)
        x = self.block2(x, pc, emit=both)
        x1 = self.block3(x, pc, emit=both)
        x = self.block4(x1, pc, emit=both)
        x = self.block5(x, pc)
        P = _passes(pc)
        out_im = [ops.proj_head(xk, self.l_im, p, scale=1.0 / (x.shape[1] * x.shape[2])) for xk, p in zip(_rows(x, pc), P)]    # MEAN pooling (:117)
        obj = ops.roi_align(x1, None, rois, valid, 8, 1.0 / 2.0, 1.0, 1e30, 0)
        f4 = self.block_obj4(obj, pc, nimg=nimg)
        out_obj = [ops.proj_head(fk, self.l_obj, p, emb=self.l_y, y=yk) for fk, yk, p in zip(_rows(f4, pc), _rows(y, pc), P)]
        if pc.dual:
            return tuple((out_im[k], out_obj[k]) for k in range(2))
        return out_im[0], out_obj[0]


class CombineDiscriminator64(CombineDiscriminator128_app):
    """reference model/rcnn_discriminator_orig.py:305-325; returns (d_img, d_obj). `bbox` is not modified."""

    two_scale = False   # single-scale ROIAlign: valid rows keep their original order

    def __init__(self, num_classes=81):
        nn.Module.__init__(self)
        self.obD = ResnetDiscriminator64(num_classes=num_classes, input_dim=3)

    def forward_padded(self, images, bbox, label, need_wgrad=True, pc=None, layout=None):
        """pc: a pass context already prepared for this pass (GanTrainer prepares the fake pass's weights on the side
        stream while the generator is still running), else one is prepared here."""
        if not images.is_cuda:
            raise RuntimeError("layout2img_amd discriminators run on the GPU HIP path only")
        rois, y, valid, count, x = self._prepare(images, bbox, label, layout)
        if pc is None:
            pc = self.arena.prepare(training=self.training, need_wgrad=need_wgrad)
        d_img, d_obj = self.obD(x, y, rois, valid, pc, nimg=count)
        return d_img, d_obj, valid, rois

    def forward(self, images, bbox, label, mask=None):
        d_img, d_obj, valid, _ = self.forward_padded(images, bbox, label)
        return d_img, d_obj[:int(valid.sum())]
