"""Flat parameter storage and the weight arena (spectral norm + MFMA weight packs).

Design (MI355X-first, 288 GB HBM): all parameters of a network live in ONE flat f32 buffer and all
gradients in another, so Adam is one elementwise launch and the data-parallel gradient exchange is
an all-reduce over contiguous slices with no bucketing copies. Every forward pass packs all
GEMM-shaped weights (after spectral normalisation) into a per-pass arena in the layouts the MFMA
kernels read; per-pass arenas keep W/sigma, u, v alive for that pass's backward, which is what the
reference's autograd does implicitly when netD is called twice before backward
(train_context_app_v2.py:158,167,173).
"""
import os
import struct

import numpy as np
import torch
import torch.nn as nn

from . import _lib

ALIGN = 8  # elements: every parameter owns a zero-padded slot of a multiple of 8 floats (ops.pad_param reads the pad)


WTU_RPW = int(os.environ.get("L2I_SN_WTU_RPW", "16"))   # rows per wave of a W^T u block
WV_R = int(os.environ.get("L2I_SN_WV_R", "4"))          # rows per wave of a W v block (the library reads the same variable)


def _round_up(v, m):
    return (v + m - 1) // m * m


class FlatParams:
    """Re-homes every parameter (and its .grad) of `module` into two flat f32 buffers on `device`."""

    def __init__(self, module: nn.Module, device):
        module.to(device)
        params = [(n, p) for n, p in module.named_parameters()]
        # biases of grouped GEMM weights (GemmWeight.group: layers that run as ONE concatenated GEMM) are laid out
        # back to back, in member order, so that the group sees one contiguous bias / bias-gradient vector
        grouped = [m for m in module.modules() if isinstance(m, GemmWeight) and m.group is not None and m.bias is not None]
        gid = {id(m.bias): i for i, m in enumerate(grouped)}
        params = [t for t in params if id(t[1]) not in gid] + sorted((t for t in params if id(t[1]) in gid), key=lambda t: gid[id(t[1])])
        offs, total = {}, 0
        for n, p in params:
            offs[n] = total
            total += _round_up(p.numel(), ALIGN)
        total = _round_up(max(total, ALIGN), ALIGN)
        self.data = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.offsets = offs
        self.numel = total
        with torch.no_grad():
            for n, p in params:
                o, k = offs[n], p.numel()
                self.data[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.data[o:o + k].view(p.shape)
                p.grad = self.grad[o:o + k].view(p.shape)
                p._l2i_slot = _round_up(k, ALIGN)   # floats of flat storage this parameter owns (pad stays zero: its gradient is never written)
        self._params = params
        self._by_id = {id(p): offs[n] for n, p in params}
        # "loose" parameters: everything that is not a GemmWeight's weight / bias (norm affines, layer norms, alphas, the
        # label embedding, PSP stage convs ...). Their gradients come through autograd's AccumulateGrad, one small add_
        # launch each (~45 per generator backward). With .grad detached (None) while the backward runs, AccumulateGrad
        # keeps the incoming tensor instead, and flush_loose() moves all of them into the flat buffer in one
        # multi-tensor launch before the optimizer / the gradient all-reduce read it.
        owned = {id(q) for m in module.modules() if isinstance(m, GemmWeight) for q in m.parameters(recurse=False)}
        self._loose = [(p, offs[n]) for n, p in params if id(p) not in owned]
        self.defer_loose = True
        self.epoch = 0   # see touch()

    def touch(self):
        """The parameters were changed by something torch's version counters cannot see (the Adam kernel writes through raw
        pointers; a graph replay does so without any host call): WeightArena's cached eval-mode packs are stale."""
        self.epoch += 1

    def offset_of(self, param):
        return self._by_id[id(param)]

    def zero_grad(self):
        self.grad.zero_()
        self.fresh = True   # nothing has been added to the weights' slices yet: the first flush may write instead of accumulate
        base = self.grad.data_ptr()
        for _, p in self._params:  # re-attach a view if someone set .grad to None
            o = self._by_id[id(p)]
            if p.grad is None or p.grad.data_ptr() != base + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
        if self.defer_loose:
            for p, _ in self._loose:
                p.grad = None

    def flush_loose(self):
        """Gradients autograd left on detached loose parameters -> the flat buffer (one multi-tensor add), views re-attached."""
        src, dst = [], []
        base = self.grad.data_ptr()
        for p, o in self._loose:
            g = p.grad
            view = self.grad[o:o + p.numel()].view(p.shape)
            if g is not None and g.data_ptr() != base + 4 * o:
                src.append(g.to(torch.float32).reshape(p.shape))
                dst.append(view)
            p.grad = view
        if src:
            torch._foreach_add_(dst, src)


class FlatBuffers:
    """Same idea for the spectral-norm u / v buffers (registered module buffers named weight_u/_v)."""

    def __init__(self, tensors, device):
        offs, total = [], 0
        for t in tensors:
            offs.append(total)
            total += _round_up(t.numel(), ALIGN)
        self.data = torch.zeros(max(total, ALIGN), dtype=torch.float32, device=device)
        self.offsets = offs
        with torch.no_grad():
            for t, o in zip(tensors, offs):
                self.data[o:o + t.numel()].copy_(t.reshape(-1))
                t.data = self.data[o:o + t.numel()].view(t.shape)


class GemmWeight(nn.Module):
    """Holder of one GEMM-shaped weight in the reference's state_dict layout.

    With spectral norm the keys are `weight_orig`, `weight_u`, `weight_v` (+ `bias`), exactly what
    torch.nn.utils.spectral_norm leaves on nn.Conv2d / nn.Linear / nn.Embedding; without it
    `weight` (+ `bias`). kind: 'conv' (Co,Ci,KH,KH), 'linear' (Co,Ci), 'embedding' (rows, dim).
    """

    def __init__(self, kind, co, ci, kh=1, bias=True, sn=True, eps=1e-12, uses=1):
        super().__init__()
        self.kind, self.co, self.ci, self.kh, self.sn, self.eps = kind, co, ci, kh, sn, eps
        # `uses`: how many times the reference applies this module per forward. torch's spectral-norm
        # pre-forward hook runs one power iteration PER APPLICATION, so a module applied twice
        # (block_obj4, rcnn_discriminator_app.py:137,141) iterates twice and each application sees its
        # own sigma. Each use gets its own row / packs in the arena.
        self.uses = uses
        shape = (co, ci, kh, kh) if kind == "conv" else (co, ci)
        w = torch.empty(shape)
        if kind == "embedding":
            nn.init.normal_(w)
        else:
            nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        if sn:
            self.weight_orig = nn.Parameter(w)
            # torch draws u, v ~ N(0,1) normalised at wrap time
            u = torch.nn.functional.normalize(torch.randn(co), dim=0, eps=eps)
            v = torch.nn.functional.normalize(torch.randn(ci * kh * kh), dim=0, eps=eps)
            self.register_buffer("weight_u", u)
            self.register_buffer("weight_v", v)
        else:
            self.weight = nn.Parameter(w)
        if bias:
            fan_in = ci * kh * kh
            bound = 1.0 / fan_in ** 0.5
            self.bias = nn.Parameter(torch.empty(co).uniform_(-bound, bound))
        else:
            self.bias = None
        self.layer_id = -1  # set by WeightArena
        self.use_rows = []   # LayerUse per application, set by WeightArena
        self.group = None    # name of a GemmGroup (set by the owning network before finalize): see WeightArena

    def use(self, k):
        return self.use_rows[k] if k else self

    @property
    def w(self):
        return self.weight_orig if self.sn else self.weight

    @property
    def co_p(self):
        return _round_up(self.co, 8)

    @property
    def ci_p(self):
        return _round_up(self.ci, 8)


def _f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


class LayerUse:
    """Arena row of the k-th application of a GemmWeight (k >= 1); quacks like the holder for ops.py."""

    def __init__(self, h, **kw):
        self.holder = h
        self.kind, self.co, self.ci, self.kh, self.sn, self.eps = h.kind, h.co, h.ci, h.kh, h.sn, h.eps
        self.co_p, self.ci_p = h.co_p, h.ci_p
        self.__dict__.update(kw)

    @property
    def bias(self):
        return self.holder.bias

    @property
    def w(self):
        return self.holder.w


LSTRIDE = 20


class GemmGroup:
    """Linear layers that read the SAME input (the 40 ISLA projections of the generator all read the object latents,
    model/norm_module.py:158-159,178) laid out as ONE GEMM: forward packs stacked along N, data-gradient packs side by side
    along K (so dX = dYcat . Wcat is one launch instead of 40 launches + 39 additions), weight-gradient slices stacked,
    biases contiguous. Spectral normalisation stays per layer (each row has its own sigma, u, v)."""

    def __init__(self, name, members):
        self.name, self.members = name, members
        h0 = members[0]
        # (a member's bias owns a zero-padded slot of _round_up(co, ALIGN) = co_p floats in the flat buffer: the group's biases
        #  are one contiguous vector of n_total floats whatever the members' channel counts)
        assert all(m.kh == 1 and m.ci == h0.ci and m.uses == 1 and _round_up(m.co, ALIGN) == m.co_p and m.bias is not None for m in members)
        self.ci, self.ci_p = h0.ci, h0.ci_p
        self.offsets, n = [], 0
        for m in members:
            self.offsets.append(n)
            n += m.co_p
        self.n_total = n


class WeightArena:
    def __init__(self, net: nn.Module, flat: FlatParams, device, op_dtype, split=False):
        """split: the forward-only "bf16x3" precision mode -- bf16 operands carried as hi + lo: every forward pack is laid out
        [w_hi | w_hi | w_lo] per tap (three Ci_p-wide blocks, csrc/weights.hip sn_pack_body SPLIT) for operands
        [x_hi | x_lo | x_hi] (ops.split_cast); the convolution kernels are unchanged and see 3 Ci_p input channels."""
        self.flat = flat
        self.device = device
        self.op_dtype = op_dtype
        self.split = bool(split)
        if self.split and op_dtype != torch.bfloat16:
            raise ValueError("split operands are a bf16 mode")
        self.dtype_code = (_lib.BF16X3 if self.split else _lib.BF16) if op_dtype == torch.bfloat16 else _lib.F32
        bk = 64 if op_dtype == torch.bfloat16 else 32
        holders = [m for m in net.modules() if isinstance(m, GemmWeight)]
        self.holders = holders
        sn_tensors = []
        for h in holders:
            if h.sn:
                sn_tensors += [h.weight_u, h.weight_v]
        for t in sn_tensors:
            t.data = t.data.to(device)
        self.sn_flat = FlatBuffers(sn_tensors, device)
        self._sn_tensors = sn_tensors
        state_off = {}
        k = 0
        for h in holders:
            if h.sn:
                state_off[id(h)] = (self.sn_flat.offsets[k], self.sn_flat.offsets[k + 1])
                k += 2
        rows = [(h, u) for h in holders for u in range(h.uses)]
        self.rounds = max(h.uses for h in holders)
        self.groups = {}
        for h in holders:
            if h.group is not None:
                self.groups.setdefault(h.group, []).append(h)
        self.groups = {k: GemmGroup(k, v) for k, v in self.groups.items()}
        member_of = {id(m): (g, i) for g in self.groups.values() for i, m in enumerate(g.members)}
        L = len(rows)
        tab = np.zeros((L, LSTRIDE), dtype=np.int64)
        packed_len = dw_len = uv_len = 0
        t_wtu = [[] for _ in range(self.rounds)]
        t_tfold = [[] for _ in range(self.rounds)]
        np_len = [0] * self.rounds    # floats of sn_wv block shares per round (csrc/weights.hip: layer row 19)
        tp_len = [0] * self.rounds    # floats of W^T u partial rows per round (behind the shares in the scratch)
        t_wv = [[] for _ in range(self.rounds)]
        t_pack = [[] for _ in range(self.rounds)]
        t_fin = [[] for _ in range(self.rounds)]
        t_dot, t_apply = [], []
        # dW-bar offsets: the slices that are ACCUMULATED into (Linear / Embedding layers, grouped projections) first and
        # contiguous -- one clear per pass (PassCtx.dw) -- then the convolutions, whose slices are stored by their launches
        def _kp(h):
            return h.kh * h.kh * h.ci_p * (3 if self.split else 1)
        dw_of = {}
        self.bias_slots = []
        for phase in (0, 1):
            if phase == 1:
                # between the two: one slot of Co_p floats per convolution whose bias is shorter than its padded channel count --
                # the weight-gradient launch sums the bias gradient over all Co_p columns there (cleared with the accumulated
                # slices), flush_grads adds the first Co values to the bias gradient (ops.FusedConvFn.backward)
                for h, use in rows:
                    # (also: a convolution applied several times per forward -- block_obj4. Its bias gradient has 2 uses x 2 passes = 4 contributions;
                    #  added straight into the shared gradient from two streams they would meet in an order that changes with the interleaving of
                    #  the streams -- eager and replayed iterations differed in their last bits. Per pass the uses add in stream order into the slot,
                    #  flush_grads adds the passes' slots in a fixed order: round 6)
                    if use == 0 and h.kind == "conv" and h.bias is not None and (h.co != h.co_p or h.uses > 1) and id(h) not in member_of:
                        h.bias_scr_off = dw_len
                        self.bias_slots.append((dw_len, dw_len + _round_up(h.co_p, ALIGN)))
                        dw_len += _round_up(h.co_p, ALIGN)
            for i, (h, use) in enumerate(rows):
                if (h.kind != "conv" or id(h) in member_of) != (phase == 0):
                    continue
                if id(h) in member_of:
                    g, gi = member_of[id(h)]
                    if gi == 0:   # first member: reserve the group's region
                        g.dw_off, dw_len = dw_len, dw_len + _round_up(g.n_total * _kp(h), ALIGN)
                    dw_of[i] = g.dw_off + g.offsets[gi] * _kp(h)
                else:
                    dw_of[i] = dw_len
                    dw_len += _round_up(h.co_p * _kp(h), ALIGN)
        for i, (h, use) in enumerate(rows):
            taps = h.kh * h.kh
            kt = h.ci * taps
            kp = taps * h.ci_p * (3 if self.split else 1)   # (split: forward K = taps x [hi | hi | lo] blocks)
            kpad, npad = _round_up(kp, bk), _round_up(h.co_p, 128)
            kp_d = taps * h.co_p
            kpad_d, npad_d = _round_up(kp_d, bk), _round_up(h.ci_p, 128)
            row = tab[i]
            row[0] = flat.offset_of(h.w)
            if h.sn:
                row[1], row[2] = state_off[id(h)]
                row[16], row[17] = uv_len, uv_len + _round_up(h.co, ALIGN)
                uv_len += _round_up(h.co, ALIGN) + _round_up(kt, ALIGN)
            else:
                row[1] = row[2] = -1
            row[3], row[4], row[5], row[6], row[7] = h.co, h.ci, h.kh, h.co_p, h.ci_p
            if id(h) in member_of:
                g, gi = member_of[id(h)]
                if gi == 0:   # first member: reserve the group's regions
                    g.kpad, g.kp, g.npad = kpad, kp, _round_up(g.n_total, 128)
                    g.kpad_d, g.npad_d = _round_up(g.n_total, bk), npad_d
                    g.fwd_off, packed_len = packed_len, packed_len + g.npad * kpad
                    g.dg_off, packed_len = packed_len, packed_len + g.npad_d * g.kpad_d
                off = g.offsets[gi]
                row[8], row[9], row[10] = kpad, npad, g.fwd_off + off * kpad          # its rows of the stacked forward pack
                row[11], row[12], row[13] = g.kpad_d, npad_d, g.dg_off + off          # its columns of the data-gradient pack (row stride = the group's K)
                kpad_d = g.kpad_d
            else:
                row[8], row[9], row[10] = kpad, npad, packed_len
                packed_len += npad * kpad
                row[11], row[12], row[13] = kpad_d, npad_d, packed_len
                packed_len += npad_d * kpad_d
            row[14] = dw_of[i]
            row[15] = _f32_bits(h.eps)
            attrs = dict(layer_id=i, kpad=kpad, npad=npad, fwd_off=int(row[10]), kpad_d=kpad_d, npad_d=npad_d,
                         dg_off=int(row[13]), dw_off=int(row[14]), kp=kp)
            if use == 0:
                for a, v in attrs.items():
                    setattr(h, a, v)
                h.use_rows = [h]
            else:
                h.use_rows.append(LayerUse(h, **attrs))
            row[18] = 1 if h.uses > 1 else 0
            bw_dot = 2304 if taps == 1 else 256                                           # csrc/weights.hip: bw_chunk_dot()
            bw = 2304 if taps == 1 else (1024 if (taps == 9 and h.ci % 4 == 0) else 256)   # csrc/weights.hip: bw_chunk() (sn_apply)
            pair_chunks = (h.co * h.ci + bw - 1) // bw
            if h.sn:
                # csrc/weights.hip sn_wtu_kernel: 256 columns x 4 waves x rpw rows; a block STORES its column sums in row (r0 / rows per block) of the
                # layer's partial matrix [row blocks][kt] in the scratch, sn_tfold_kernel adds the row blocks in order (no atomics: round 6)
                nrb = (h.co + 4 * WTU_RPW - 1) // (4 * WTU_RPW)
                for c0 in range(0, kt, 256):
                    for rb, r0 in enumerate(range(0, h.co, 4 * WTU_RPW)):
                        t_wtu[use].append((i, c0, r0, WTU_RPW, tp_len[use] + rb * kt))
                cw = 256 if nrb <= 8 else (64 if nrb <= 32 else 16)   # columns per fold workgroup: 256 / cw row lanes walk the row blocks
                for c0 in range(0, kt, cw):
                    t_tfold[use].append((i, c0, tp_len[use], nrb, cw))
                tp_len[use] += _round_up(nrb * kt, 4)
                row[19] = np_len[use]                     # sn_wv_kernel<R>: 4 R rows per block, one stored share of ||W v||^2 each
                for r0 in range(0, h.co, 4 * WV_R):
                    t_wv[use].append((i, r0))
                    np_len[use] += 1
                t_dot += [(i, c) for c in range((h.co * h.ci + bw_dot - 1) // bw_dot)]
            tci = 256 if taps == 1 else 32
            for ct in range((h.co_p + 63) // 64):         # PK_TCO
                for cc in range((h.ci_p + tci - 1) // tci):
                    t_pack[use].append((i, ct, cc))
            t_fin[use].append((i,))
            t_apply += [(i, c) for c in range(pair_chunks)]
        self.n_layers = L
        self.packed_len, self.dw_len = packed_len, max(dw_len, ALIGN)
        self.uv_len = max(uv_len, ALIGN)

        def dev(a, width):
            arr = np.array(a if a else [(0,) * width], dtype=np.int32).reshape(-1)
            return torch.from_numpy(np.ascontiguousarray(arr)).to(device), len(a)
        self.layers = torch.from_numpy(tab.reshape(-1)).to(device)
        self.t_wtu = [dev(t, 5) for t in t_wtu]
        self.t_tfold = [dev(t, 5) for t in t_tfold]
        self.np_len = [_round_up(n, 4) for n in np_len]
        self.sn_scratch_floats = max(a + b for a, b in zip(self.np_len, tp_len)) if self.rounds else 0
        self.t_wv = [dev(t, 2) for t in t_wv]
        self.t_pack = [dev(t, 3) for t in t_pack]
        self.t_fin = [dev(t, 1) for t in t_fin]
        # Layer GROUPS for the data-parallel gradient exchange (flush_grads(on_group=)): the rows of both backward tables are laid out
        # group-major, a group being a contiguous range of the flat parameter buffer of ~1/NG of the weights; the spectral-norm
        # backward then runs group by group and every finished range of the flat gradient buffer can be handed to its all-reduce
        # while the following groups are still being corrected (trainer.FlatAdam).
        NG = max(1, int(os.environ.get("L2I_GRAD_GROUPS", "4")))
        by_off = sorted(range(L), key=lambda i: int(tab[i][0]))
        total_w = sum(rows[i][0].co * rows[i][0].ci * rows[i][0].kh * rows[i][0].kh for i in by_off) or 1
        group_of, acc_w, starts, prev_h = {}, 0, [0], None
        for i in by_off:
            h = rows[i][0]
            if h is not prev_h:   # (the uses of one weight share its flat slot: they stay in one group)
                if min(NG - 1, acc_w * NG // total_w) > len(starts) - 1:
                    starts.append(int(tab[i][0]))
                acc_w += h.co * h.ci * h.kh * h.kh
                prev_h = h
            group_of[i] = len(starts) - 1
        ng = len(starts)
        t_dot.sort(key=lambda r: group_of[r[0]])     # (stable: layer order inside a group is kept)
        t_apply.sort(key=lambda r: group_of[r[0]])
        def _bounds(t):
            b = [0] * (ng + 1)
            for r in t:
                b[group_of[r[0]] + 1] += 1
            for k in range(ng):
                b[k + 1] += b[k]
            return b
        self.grad_groups = dict(n=ng, dot=_bounds(t_dot), apply=_bounds(t_apply), lo=starts, hi=starts[1:] + [flat.numel])
        self.grad_groups["lo"][0] = 0
        # (first dot-table entry, number of entries) per layer row: sn_dotfold_kernel adds a layer's stored shares of <G, W> in that order
        rng = np.zeros((L, 2), dtype=np.int32)
        for k, (li, _) in enumerate(t_dot):
            if rng[li, 1] == 0:
                rng[li, 0] = k
            rng[li, 1] += 1
        self.t_dot_range = torch.from_numpy(rng.reshape(-1).copy()).to(device)
        self.t_dot, self.n_dot = dev(t_dot, 2)
        self.t_apply, self.n_apply = dev(t_apply, 2)
        self.pending = []
        self.free_packs = []   # zero-padded pack buffers of finished passes (see PassCtx)
        # dW-bar accumulators (PassCtx.dw): a convolution's slice is STORED by its one weight-gradient launch per pass
        # (ops.WGRAD_OVERWRITE), so only the slices that are accumulated into (Linear / Embedding layers: heads add with atomics,
        # grouped projections, ArenaWeightFn) need clearing -- a few MB instead of the whole 163 / 251 MB buffer per pass.
        self.conv_uses, acc = [], []
        for h in holders:
            if h.group is not None:
                continue   # (its rows live in the group's slice)
            for u in h.use_rows:
                if h.kind == "conv":
                    self.conv_uses.append(u)
                else:
                    acc.append((u.dw_off, u.dw_off + _round_up(h.co_p * u.kp, ALIGN)))
        for g in self.groups.values():
            acc.append((g.dw_off, g.dw_off + _round_up(g.n_total * g.kp, ALIGN)))
        acc += self.bias_slots
        acc.sort()
        self.acc_ranges = []
        for a, b in acc:
            if self.acc_ranges and a <= self.acc_ranges[-1][1]:
                self.acc_ranges[-1][1] = max(self.acc_ranges[-1][1], b)
            else:
                self.acc_ranges.append([a, b])

    # ---- eval-mode pass cache (sampling, reference test_context_app_v2.py:68-77): in eval mode the spectral-norm hook does not
    # iterate, so W / sigma and both packs depend on the parameters and u / v only -- normalising and packing all of a
    # generator's 41 M parameters again on every forward was ~0.2 ms of each sampling call. The pass context of the last
    # no-grad eval forward is kept and handed out again while nothing has changed: the version counters of every parameter
    # and u / v buffer (torch-side writes: load_state_dict, .copy_, ...) plus FlatParams.epoch / sn_epoch (raw-pointer writes:
    # the Adam kernel, graph replays, the train-mode power iteration). A stale cache is re-packed IN PLACE (same buffers), so
    # that a HIP graph captured over it (sampling.GraphSampler) stays valid.
    EVAL_CACHE = os.environ.get("L2I_EVAL_CACHE", "1") != "0"
    _eval_pc = None
    _eval_stamp = None
    _eval_event = None
    sn_epoch = 0

    def _stamp(self):
        v = 0
        for _, p in self.flat._params:
            v += p._version
        for t in self._sn_tensors:
            v += t._version
        return (self.flat.epoch, self.sn_epoch, v, self.flat.data._version, self.sn_flat.data._version)

    def _eval_pass(self):
        st = self._stamp()
        pc = self._eval_pc
        cur = torch.cuda.current_stream()
        if pc is not None and st == self._eval_stamp:
            if self._eval_event is not None and self._eval_event[0] != cur.cuda_stream:
                cur.wait_event(self._eval_event[1])   # (packed on another stream)
            return pc
        if torch.cuda.is_current_stream_capturing():
            return None   # never create / refresh the cache inside a capture (its buffers would live in the graph's pool)
        if pc is None:
            pc = PassCtx(self, False, False)
        self._pack(pc, False)
        ev = torch.cuda.Event()
        ev.record(cur)
        self._eval_pc, self._eval_stamp, self._eval_event = pc, st, (cur.cuda_stream, ev)
        return pc

    def prepare(self, training=True, need_wgrad=True):
        """Run the power iteration(s) (train mode) and pack all weights; returns the pass context."""
        if not training and self.EVAL_CACHE and not torch.is_grad_enabled():
            p = self._eval_pass()
            if p is not None:
                return p
        p = PassCtx(self, training, need_wgrad)
        self._pack(p, training)
        if training:
            self.sn_epoch += 1   # (u / v advanced through raw pointers)
        if need_wgrad and torch.is_grad_enabled():
            self.pending.append(p)
            if len(self.pending) > 8:  # forwards that were never followed by an optimiser step: the oldest one WITHOUT gradients goes
                # (a pass whose backward has run holds weight gradients in its accumulator and padded-channel bias gradients in its
                #  bias_fix slots until flush_grads: evicting it would silently lose them -- ADVICE r05 -- so such passes stay)
                for k, old in enumerate(self.pending[:-1]):
                    if old.dwbar is None and not old.bias_fix:
                        del self.pending[k]
                        break
        return p

    def _pack(self, p, training):
        scr, nscr = _lib.wgrad_scratch(self.device)   # the power iteration's partial sums (stored, added in a fixed order): transient, this stream's scratch
        assert self.sn_scratch_floats <= nscr
        for r in range(self.rounds):
            (wtu, n_wtu), (wv, n_wv), (pk, n_pk), (fin, n_fin) = self.t_wtu[r], self.t_wv[r], self.t_pack[r], self.t_fin[r]
            tf, n_tf = self.t_tfold[r]
            if n_pk == 0:
                continue
            _lib.call("l2i_weights_prepare", self.layers.data_ptr(), self.n_layers, wtu.data_ptr(), n_wtu, wv.data_ptr(), n_wv,
                      pk.data_ptr(), n_pk, fin.data_ptr(), n_fin, self.flat.data.data_ptr(), self.sn_flat.data.data_ptr(), p.pass_uv.data_ptr(),
                      self.uv_len, p.norms.data_ptr(), p.packed.data_ptr(), self.dtype_code, 1 if training else 0,
                      1 if r == 0 else 0, tf.data_ptr(), n_tf, scr, nscr, self.np_len[r], _lib.raw_stream())

    def flush_grads(self, on_group=None):
        """Apply the spectral-norm backward of every pending pass into the flat gradient buffer.
        on_group(lo, hi): called after each layer group (self.grad_groups) with the range of the flat gradient buffer that is
        FINAL from then on -- the data-parallel trainer launches that range's all-reduce there, so that the exchange of group k
        runs beside the correction of groups k + 1 .. (reference train_context_app_v2.py:108-110: nn.DataParallel reduces the
        replicas' gradients inside backward). None: one launch pair over all layers."""
        from .ops import WgradSide
        WgradSide.join()   # weight-gradient launches run on side streams (ops.WgradSide)
        self.flat.flush_loose()
        # padded-channel bias gradients summed by the weight-gradient launches: one multi-tensor add PER PASS. (Round 5 issued one add over all
        # pending passes: two passes of one network name the SAME bias gradient twice, and a multi-tensor kernel that reads and writes one
        # destination from two list entries at once keeps only one of the two addends -- or both, from run to run: final.2.bias differed by 6 %
        # between identical runs of "two forwards, then backward", which is what made tests/test_gpu_12_extra.py flicker at 2.1e-3 in round 5.)
        for p in self.pending:
            if p.bias_fix:
                fix = list(p.bias_fix.values())
                torch._foreach_add_([d for d, _ in fix], [s_ for _, s_ in fix])
                p.bias_fix = {}
        live = [p for p in self.pending if p.dwbar is not None]
        for p in live:
            p.clear_unwritten()
        gg = self.grad_groups
        spans = [(0, self.n_dot, 0, self.n_apply, 0, self.flat.numel)] if on_group is None else \
            [(gg["dot"][k], gg["dot"][k + 1], gg["apply"][k], gg["apply"][k + 1], gg["lo"][k], gg["hi"][k]) for k in range(gg["n"])]
        fresh = bool(getattr(self.flat, "fresh", False))
        for d0, d1, a0, a1, lo, hi in spans:
            for i in range(0, len(live), 2):   # two passes per launch pair (D(real) + D(fake)): W and the gradient buffer are walked once
                p, q = live[i], (live[i + 1] if i + 1 < len(live) else None)
                if d1 > d0 or a1 > a0:
                    _lib.call("l2i_weights_backward2", self.layers.data_ptr(), self.n_layers, self.t_dot.data_ptr() + 8 * d0, d1 - d0,
                              self.t_apply.data_ptr() + 8 * a0, a1 - a0, self.flat.data.data_ptr(), p.dwbar.data_ptr(),
                              p.pass_uv.data_ptr(), p.norms.data_ptr(), q.dwbar.data_ptr() if q else None,
                              q.pass_uv.data_ptr() if q else None, q.norms.data_ptr() if q else None, self.flat.grad.data_ptr(),
                              _lib.workspace(self.device), 1 if (fresh and i == 0) else 0, self.t_dot_range.data_ptr(), d0, _lib.raw_stream())
            if on_group is not None:
                on_group(lo, hi)
        if live:
            self.flat.fresh = False
        self.pending = []

    def drop_pending(self):
        self.pending = []


class Span:
    """(address, element count) of a slice of one of a pass's long-lived buffers (weight packs, dWbar accumulator): all a launch needs of
    it. The eager iteration names ~375 such slices; as tensor views each costs a trip through torch's dispatcher (~1.5 us) for the sake
    of one `data_ptr()`. Quacks like the view where ops.conv_raw / wgrad_raw look at it: data_ptr(), numel(), record_stream()."""
    __slots__ = ("ptr", "n", "base")

    def __init__(self, base, ptr, n):
        self.base, self.ptr, self.n = base, ptr, n

    def data_ptr(self):
        return self.ptr

    def numel(self):
        return self.n

    def record_stream(self, stream):
        self.base.record_stream(stream)


class PassCtx:
    """Packed weights, u/v snapshots, sigma and the dWbar accumulator of ONE forward pass."""

    def __init__(self, arena: WeightArena, training, need_wgrad):
        self.arena = arena
        self.training = training
        self.need_wgrad = need_wgrad
        dev = arena.device
        # The pack kernel writes only the true (co < Co_p, ci < Ci_p) elements; padding rows and K tails must be
        # zero, so buffers are zero-filled once and recycled through the arena when their pass dies.
        self.packed = arena.free_packs.pop() if arena.free_packs else torch.zeros(arena.packed_len, dtype=arena.op_dtype, device=dev)
        self._packed_ptr, self._esz = self.packed.data_ptr(), self.packed.element_size()   # (for the spans below: the buffer is this pass's for life)
        # (one allocation: the library clears norms and pass_uv with a single memset when they are adjacent)
        nn4 = _round_up(4 * arena.n_layers, ALIGN)
        buf = torch.empty(nn4 + arena.uv_len, dtype=torch.float32, device=dev)
        self.norms, self.pass_uv = buf[:4 * arena.n_layers], buf[nn4:]
        self.dwbar = None
        self.bias_fix = {}   # slot offset -> (bias gradient view, slot[:Co]): what flush_grads adds up

    def __del__(self):
        try:
            # (a buffer lent to a graph capture -- GanTrainer._lend_packs -- is rewritten by every replay: never back to the eager free list)
            if len(self.arena.free_packs) < 4 and not getattr(self.packed, "_l2i_lent", False):
                self.arena.free_packs.append(self.packed)
        except Exception:
            pass

    def dw(self):
        if self.dwbar is None:
            from . import ops
            a = self.arena
            if not ops.WGRAD_OVERWRITE:
                self.dwbar = torch.zeros(a.dw_len, dtype=torch.float32, device=a.device)
            else:   # conv slices are stored by their weight-gradient launch; the accumulated-into slices are cleared here
                self.dwbar = torch.empty(a.dw_len, dtype=torch.float32, device=a.device)
                if os.environ.get("L2I_DW_NAN", "0") == "1":   # (tests: any slice that is neither stored nor cleared shows up as NaN gradients)
                    self.dwbar.fill_(float("nan"))
                for lo, hi in a.acc_ranges:
                    self.dwbar[lo:hi].zero_()
                self.written = set()
        return self.dwbar

    def bias_slot(self, h, bg):
        """h's Co_p-float bias-gradient slot in the accumulator (WeightArena.bias_slots); its first Co values are added to `bg` by flush_grads."""
        s = self.dw()[h.bias_scr_off:h.bias_scr_off + h.co_p]
        self.bias_fix[h.bias_scr_off] = (bg, s[:h.co])   # (keyed: a second backward over the same forward adds into the same slot)
        return s

    def mark_written(self, h):
        """h's dW slice has been stored by a weight-gradient launch of this pass."""
        if self.written is not None:
            self.written.add(h.dw_off)

    written = None

    def was_written(self, h):
        """A weight-gradient launch of this pass has already stored h's slice (a second backward over the same forward, e.g.
        retain_graph): the next launch must ADD to it."""
        return self.written is not None and h.dw_off in self.written

    def dw_acc(self, h):
        """h's dW slice for a caller that ADDS to it (ArenaWeightFn): a convolution's slice is cleared on first use."""
        s = self.dw_slice(h)
        if self.written is not None and h.kind == "conv" and getattr(h, "group", None) is None and h.dw_off not in self.written:
            s.zero_()
            self.written.add(h.dw_off)
        return s

    def clear_unwritten(self):
        """Before the spectral-norm backward reads the accumulator: convolutions whose weight gradient was never launched in this
        pass (a layer off the loss's path) have an undefined slice -- their gradient is zero."""
        if self.written is None or self.dwbar is None:
            return
        for u in self.arena.conv_uses:
            if u.dw_off not in self.written:
                self.dw_slice(u).zero_()
        self.written = None

    def fwd_pack(self, h):
        return self.packed[h.fwd_off:h.fwd_off + h.npad * h.kpad]

    def dgrad_pack(self, h):
        return self.packed[h.dg_off:h.dg_off + h.npad_d * h.kpad_d]

    def dw_slice(self, h):
        return self.dw()[h.dw_off:h.dw_off + h.co_p * h.kp]

    # the same three as `Span`s, for callers that only launch on them (ops.FusedConvFn)
    def fwd_span(self, h):
        return Span(self.packed, self._packed_ptr + self._esz * h.fwd_off, h.npad * h.kpad)

    def dgrad_span(self, h):
        return Span(self.packed, self._packed_ptr + self._esz * h.dg_off, h.npad_d * h.kpad_d)

    def dw_span(self, h):
        d = self.dw()
        return Span(d, d.data_ptr() + 4 * h.dw_off, h.co_p * h.kp)

    def fwd_span_b(self, h):
        return None

    def dgrad_span_b(self, h):
        return None

    def dw_span_b(self, h):
        return None

    def sigma(self, h):
        return self.norms[4 * h.layer_id + 2]

    # (single pass: no second set of packs / accumulators -- see DualPass)
    dual = False

    def fwd_pack_b(self, h):
        return None

    def dgrad_pack_b(self, h):
        return None

    def dw_slice_b(self, h):
        return None

    def group_fwd_pack(self, g):
        return self.packed[g.fwd_off:g.fwd_off + g.npad * g.kpad]

    def group_dgrad_pack(self, g):
        return self.packed[g.dg_off:g.dg_off + g.npad_d * g.kpad_d]

    def group_dw_slice(self, g):
        return self.dw()[g.dw_off:g.dw_off + g.n_total * g.kp]


class DualPass:
    """Two passes of one network run as ONE batch: images [0, b) belong to pass `a`, images [b, 2b) to pass `b` (the
    discriminator step's D(real) and D(fake), reference train_context_app_v2.py:158,167). The reference's spectral_norm hook
    iterates once per pass, so each pass has its own W / sigma packs and its own dWbar accumulator (the sigma-correction of
    the spectral-norm backward differs per pass); every conv / data-gradient / weight-gradient launch takes both
    (l2i_conv2d_fwd_dual, l2i_conv2d_wgrad_dual) and processes twice the tiles. Quacks like a PassCtx for ops.fused_conv;
    the heads, which are not convolutions, take `.a` / `.b` for their half of the rows."""
    dual = True

    def __init__(self, a: PassCtx, b: PassCtx):
        assert a.arena is b.arena and a.training == b.training and a.need_wgrad == b.need_wgrad
        self.a, self.b = a, b
        self.arena, self.training, self.need_wgrad = a.arena, a.training, a.need_wgrad

    def fwd_pack(self, h):
        return self.a.fwd_pack(h)

    def dgrad_pack(self, h):
        return self.a.dgrad_pack(h)

    def dw_slice(self, h):
        return self.a.dw_slice(h)

    def fwd_pack_b(self, h):
        return self.b.fwd_pack(h)

    def dgrad_pack_b(self, h):
        return self.b.dgrad_pack(h)

    def dw_slice_b(self, h):
        return self.b.dw_slice(h)

    def fwd_span(self, h):
        return self.a.fwd_span(h)

    def dgrad_span(self, h):
        return self.a.dgrad_span(h)

    def dw_span(self, h):
        return self.a.dw_span(h)

    def fwd_span_b(self, h):
        return self.b.fwd_span(h)

    def dgrad_span_b(self, h):
        return self.b.dgrad_span(h)

    def dw_span_b(self, h):
        return self.b.dw_span(h)

    def mark_written(self, h):
        self.a.mark_written(h)
        self.b.mark_written(h)

    def was_written(self, h):
        return self.a.was_written(h) or self.b.was_written(h)

    def dw_acc(self, h):
        self.b.dw_acc(h)
        return self.a.dw_acc(h)
