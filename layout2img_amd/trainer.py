"""The GAN training iteration of reference train_context_app_v2.py:148-189 on the HIP path.

D step: hinge(real) + hinge(fake) over (image x0.1, object x1, appearance x1), Adam.
G step: -D(fake) terms + L1(fake, real), Adam. (The VGG perceptual term needs downloaded
weights and is excluded, as stated in BASELINE.md / DESIGN.md.) Adam betas (0, 0.999), lr 1e-4.

Differences from the reference that do not change the mathematics:
  * the third D forward (G step) skips D's weight gradients -- the reference computes and then
    discards them at the next netD.zero_grad() (:156,178-188);
  * ROI rows are a fixed b*o with a validity mask, losses average over valid rows;
  * under data parallelism losses divide by global counts and gradients are SUM all-reduced.
"""
import os

import torch

from . import ops, parallel


class FlatAdam:
    """torch.optim.Adam(betas=(0, 0.999)) semantics over a network's flat parameter buffer
    (train_context_app_v2.py:112-127). step(): spectral-norm backward flush -> gradient all-reduce
    -> one fused Adam launch."""

    def __init__(self, net, lr=1e-4, betas=(0.0, 0.999), eps=1e-8):
        self.net, self.lr, self.betas, self.eps = net, lr, betas, eps
        self.m = torch.zeros_like(net.flat.data)
        self.v = torch.zeros_like(net.flat.data)
        self.t = 0
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=net.flat.data.device)   # the count the kernel reads
        self.group = None   # process group of the gradient all-reduce (GanTrainer: parallel.grad_group(), its own comm stream)

    _works = None
    # Data parallel: the spectral-norm backward runs in layer groups (arena.grad_groups) and each group's range of the flat
    # gradient buffer is handed to its all-reduce the moment it is final; the Adam update of a range is launched as soon as ITS
    # all-reduce has landed. The exchange of group k thus runs beside the correction of groups k + 1 .. and beside the Adam
    # launches of groups .. k - 1, instead of one 251 MB all-reduce (D) fully exposed between flush and Adam. L2I_GRAD_CHUNKED=0:
    # one flush, one all-reduce over the whole buffer, one Adam launch (the round-4 form).
    chunked = os.environ.get("L2I_GRAD_CHUNKED", "1") != "0"

    def begin_step(self):
        """First half of step(): spectral-norm backward flush, then the gradient all-reduce LAUNCHED (asynchronously under
        data parallelism: the collective runs on the communicator's stream while this stream goes on)."""
        flat = self.net.flat
        if parallel.active() and self.chunked:
            self._works = []
            self.net.arena.flush_grads(on_group=lambda lo, hi: self._works.append(
                (lo, hi, parallel.allreduce_flat_(flat.grad[lo:hi], async_op=True, group=self.group))))
        else:
            self.net.arena.flush_grads()
            self._works = [(0, flat.numel, parallel.allreduce_flat_(flat.grad, async_op=True, group=self.group))]

    def finish_step(self):
        """Second half: per range, the current stream waits for the range's all-reduce, then its Adam launch."""
        self.t += 1
        self.t_dev += 1   # (device-side: a captured graph of the iteration replays with the right bias corrections)
        for lo, hi, works in self._works or [(0, self.net.flat.numel, [])]:
            parallel.wait_all(works)
            ops.adam_step(self.net.flat, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.t, step_dev=self.t_dev, lo=lo, hi=hi)
        self._works = None

    def step(self):
        self.begin_step()
        self.finish_step()


_WGRAD_SIDE_G = os.environ.get("L2I_WGRAD_STREAM", "0") == "2"


class PhaseStamps:
    """Measurement aid (tools/perf/phase_stamps.py): GanTrainer.stamps = PhaseStamps(dev) makes _step launch a one-lane kernel that stores
    the device wall clock at each phase boundary, in stream order -- also inside a captured graph, so the slots hold the timeline
    of the last REPLAY, with no profiler attached. None (default): nothing is launched."""

    def __init__(self, device, n=32):
        self.buf = torch.zeros(n, dtype=torch.int64, device=device)
        self.names = []

    def mark(self, name):
        from . import _lib
        if name not in self.names:
            self.names.append(name)
        _lib.call("l2i_debug_stamp", self.buf[self.names.index(name)].data_ptr(), _lib.raw_stream())

    def read_us(self):
        t = self.buf[:len(self.names)].cpu().tolist()
        return [(n, (v - t[0]) / 100.0) for n, v in zip(self.names, t)]   # 100 MHz ticks -> us since the first mark


class GanTrainer:
    stamps = None

    def _mark(self, name):
        if self.stamps is not None:
            self.stamps.mark(name)

    def __init__(self, netG, netD, g_lr=1e-4, d_lr=1e-4, lamb_obj=1.0, lamb_app=1.0, lamb_img=0.1, z_dim=128, vgg=None):
        """vgg: an optional layout2img_amd.VGGLoss (finalized) -- the perceptual term of the G loss
        (train_context_app_v2.py:141,185); None leaves it out (the headline benchmark's configuration)."""
        self.netG, self.netD, self.vgg = netG, netD, vgg
        self.g_opt, self.d_opt = FlatAdam(netG, g_lr), FlatAdam(netD, d_lr)
        self.l_obj, self.l_app, self.l_img, self.z_dim = lamb_obj, lamb_app, lamb_img, z_dim
        self.world = parallel.world_size()
        self.dp = parallel.active()   # collectives are issued (world > 1, or the forced one-rank group: parallel.FORCE)
        # D(real) on a side stream next to G's forward (L2I_OVERLAP=0 turns it off)
        self.overlap = os.environ.get("L2I_OVERLAP", "1") != "0"
        self._side = None
        # D(real) and D(fake) of the discriminator step as ONE batch of 2b images (CombineDiscriminator.forward_dual): every
        # conv / data-gradient / weight-gradient launch of the D step then has twice the tiles (at b = 32 a single pass is ONE
        # round of workgroups per launch, so prologue, K loop and epilogue add up instead of overlapping), with each pass's own
        # spectral-norm iteration, packs and gradient accumulator as in the reference. Measured (round 4, 128x128, b = 32, same
        # box): 52 conv launches fewer per iteration, conv fraction of the MFMA peak 0.300 -> 0.324, kernel time of the iteration
        # 22.9 -> 21.7 ms under the profiler -- but the two-pass form runs D(real) on a side stream next to G's forward and next to
        # D(fake)'s backward, which hides ~2 ms of kernel time that the one-batch form (it needs G's output before it can start)
        # cannot hide: 20.5 ms against 21.4 ms per iteration, replayed or eager. OFF by default; L2I_DUAL_D=1 / `dual_d = True`.
        self.dual_d = os.environ.get("L2I_DUAL_D", "0") != "0"
        self.real_bwd_early = os.environ.get("L2I_REAL_BWD_EARLY", "0") != "0"   # (two-pass form: see _step)
        # Data parallel: G's gradient all-reduce (163 MB) is launched at the end of an iteration and waited for -- together
        # with G's Adam step -- only when the NEXT iteration needs G's weights, i.e. after that iteration's D(real) pass
        # has been enqueued on the side stream: the collective overlaps D(real). (D's all-reduce has nothing independent
        # next to it: the G step reads D's updated weights at once.) flush() completes a pending step; L2I_DEFER_G=0: off.
        # (one GPU: off by default -- L2I_DEFER_G=1 turns it on there too: G's spectral-norm backward + Adam then run next to the
        #  following iteration's D(real) pass instead of at the end of their own iteration)
        self.defer_g = self.overlap and os.environ.get("L2I_DEFER_G", "1" if self.dp else "0") != "0"
        self._pending_g = False
        # whoever reads the generator's parameters from outside the loop (sampling.sample, checkpoints) completes a deferred
        # step first: the hook rides on the network
        netG._l2i_flush = self.flush
        if self.dp:
            self.g_opt.group = self.d_opt.group = parallel.grad_group()
            netG.sync = parallel.sync_bn_stats
            parallel.broadcast_flat_(netG.flat.data)
            parallel.broadcast_flat_(netD.flat.data)
            parallel.broadcast_flat_(netG.arena.sn_flat.data)
            parallel.broadcast_flat_(netD.arena.sn_flat.data)
            # every other buffer too (BN running statistics, num_batches_tracked): a checkpoint loaded on rank 0 only
            # must give every rank the same eval-mode statistics. (The PSP stages' plain nn.BatchNorm2d are NOT
            # synchronised in the reference either, resnet_generator_app_v2.py:745, so their running statistics drift
            # per rank afterwards; checkpoints are saved from rank 0.)
            sn_ptrs = {netG.arena.sn_flat.data.untyped_storage().data_ptr(), netD.arena.sn_flat.data.untyped_storage().data_ptr()}
            for net in (netG, netD):
                for buf in net.buffers():
                    if buf.untyped_storage().data_ptr() not in sn_ptrs:
                        parallel.broadcast_flat_(buf)

    def flush(self):
        """Complete a deferred generator step (see defer_g). Call before reading / saving the generator's parameters."""
        if self._pending_g:
            self.g_opt.finish_step()
            self._pending_g = False

    def _counts(self, valid, b):
        if not self.dp:
            return None, None
        n_roi = parallel.global_count(valid.sum().float().view(1))
        n_img = torch.full((1,), float(b * self.world), device=valid.device)
        return n_roi, n_img

    def _d_terms(self, outs, valid, mode, n_roi, n_img):
        """outs = (d_img, d_obj[, d_app]) -- the 64x64 discriminator has no appearance head."""
        terms = [(outs[1], valid, self.l_obj, n_roi), (outs[0], None, self.l_img, n_img)]
        if len(outs) > 2:
            terms.append((outs[2], valid, self.l_app, n_roi))
        return ops.hinge_sum(terms, mode)

    def step(self, real, label, bbox, z=None, z_im=None):
        """One iteration. real (b,3,H,W) in [-1,1]; label (b,o) int64; bbox (b,o,4). Returns loss tensors
        (device scalars, no host sync)."""
        b, o = label.shape[0], label.shape[1]
        y = label.view(b, o)
        if z is None:
            z = torch.randn(b, o, self.z_dim, device=real.device)
        with ops.POOL.step(real.device):
            return self._step(real, y, bbox, z, z_im, b)

    def _step(self, real, y, bbox, z, z_im, b):
        netG, netD = self.netG, self.netD
        # ROI rows compacted to the front in the reference's order + their device-side count: once for the three D passes
        self._mark("start")
        layout = netD.prepare_layout(bbox, y, real.size(2), real.device)
        valid = layout[2]
        # ---- D step (reference :156-174)
        netD.zero_grad()
        if self.dual_d:
            # both passes' power iterations + weight packs need only D's weights: on the side stream, next to G's forward
            cur = torch.cuda.current_stream()
            if self.overlap:
                if self._side is None:
                    self._side = torch.cuda.Stream()
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    pcs = (netD.arena.prepare(training=netD.training, need_wgrad=True),    # reference order: real first (:158), then fake (:167)
                           netD.arena.prepare(training=netD.training, need_wgrad=True))
            else:
                pcs = None
            n_roi, n_img = self._counts(valid, b)
            self.flush()
            fake = netG(z, bbox, z_im=z_im, y=y)
            if self.overlap:
                cur.wait_stream(self._side)
            outs_r, outs_f, _, _ = netD.forward_dual(real, fake.detach(), bbox, y, pcs=pcs, layout=layout)
            d_loss_real = self._d_terms(outs_r, valid, 0, n_roi, n_img)
            d_loss_fake = self._d_terms(outs_f, valid, 1, n_roi, n_img)
        elif self.overlap:
            # D(real) does not depend on the generator: it runs on a side stream next to G's forward (and, through
            # autograd's stream bookkeeping, its backward runs next to D(fake)'s). D has no batch norm, so the side
            # stream carries no collective.
            # The loss terms of D(real) are formed on the side stream as well: the backward of that branch then hangs
            # off the very first node of the graph and starts at once (formed on the main stream it would queue behind
            # the whole backward of D(fake): measured 3 % slower than no overlap at all).
            n_roi, n_img = self._counts(valid, b)   # (collective, if any, on the main stream)
            cur = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                *outs_r, _, _ = netD.forward_padded(real, bbox, y, layout=layout)
                d_loss_real = self._d_terms(outs_r, valid, 0, n_roi, n_img)
                # the fake pass's power iteration + weight packs need only D's weights: done here, off the main stream
                pc_fake = netD.arena.prepare(training=netD.training, need_wgrad=True)
                if self.real_bwd_early:
                    # D(real)'s backward needs nothing but its own forward: it starts here, on the side stream, next to G's forward
                    # and D(fake)'s forward (small, latency-bound launches) instead of next to D(fake)'s backward (the same
                    # chip-filling weight-gradient launches competing for the L2s). The main stream waits for the packs only.
                    packs_ready = torch.cuda.Event()
                    packs_ready.record(self._side)
                    d_loss_real.backward()
                    d_loss_real = d_loss_real.detach()
            self.flush()   # the previous iteration's G all-reduce + Adam: behind D(real)'s launches, in front of G's forward
            self._mark("G forward")
            fake = netG(z, bbox, z_im=z_im, y=y)
            self._mark("D(fake) forward")
            if self.real_bwd_early:
                cur.wait_event(packs_ready)
            else:
                cur.wait_stream(self._side)
            *outs_f, _, _ = netD.forward_padded(fake.detach(), bbox, y, pc=pc_fake, layout=layout)
            d_loss_fake = self._d_terms(outs_f, valid, 1, n_roi, n_img)
        else:
            self.flush()
            *outs_r, _, _ = netD.forward_padded(real, bbox, y, layout=layout)
            n_roi, n_img = self._counts(valid, b)
            d_loss_real = self._d_terms(outs_r, valid, 0, n_roi, n_img)
            fake = netG(z, bbox, z_im=z_im, y=y)
            *outs_f, _, _ = netD.forward_padded(fake.detach(), bbox, y, layout=layout)
            d_loss_fake = self._d_terms(outs_f, valid, 1, n_roi, n_img)
        if self.overlap and not self.dual_d and self.real_bwd_early:
            d_loss_fake.backward()
            d_loss = d_loss_real + d_loss_fake.detach()
        else:
            d_loss = d_loss_real + d_loss_fake
            self._mark("D backward (real on the side stream)")
            d_loss.backward()
            self._mark("join side stream")
        if self.overlap and not self.dual_d:
            # D(real)'s backward ran on the side stream; what it wrote outside autograd's view (weight-gradient
            # accumulators, direct bias-gradient atomics) and the pack buffers it read must be ordered before the
            # optimizer step on this stream explicitly (capture-safe: one event).
            torch.cuda.current_stream().wait_stream(self._side)
        self._mark("D flush + all-reduce + Adam")
        self.d_opt.step()
        # ---- G step (reference :178-189)
        netG.zero_grad()
        self._mark("G step: D forward")
        *outs_g, _, _ = netD.forward_padded(fake, bbox, y, need_wgrad=False, layout=layout)
        g_adv = self._d_terms(outs_g, valid, 2, n_roi, n_img)
        pixel = ops.l1_loss(fake, real, 1.0 / self.world)
        g_loss = g_adv + pixel
        if self.vgg is not None:
            feat = self.vgg(fake, real)
            g_loss = g_loss + (feat if self.world == 1 else feat / self.world)
        if _WGRAD_SIDE_G:   # (tuning, L2I_WGRAD_STREAM=2: the generator's weight gradients on a side stream during the G step only)
            ops.WgradSide.enabled = True
        self._mark("G step: backward through D and G")
        g_loss.backward()
        self._mark("G flush + Adam")
        if _WGRAD_SIDE_G:
            ops.WgradSide.enabled = False
            ops.WgradSide.join()
        if self.defer_g:
            self.g_opt.begin_step()
            self._pending_g = True
        else:
            self.g_opt.step()
        self._mark("end")
        return {"d_loss": d_loss.detach(), "g_loss": g_loss.detach(), "pixel": pixel.detach(), "fake": fake.detach()}

    # ---- whole-iteration HIP graph: the iteration is ~1700 launches and the Python / autograd side costs about as much
    # host time as the GPU takes (measured 37 ms vs 38 ms), so a captured graph removes the host from the loop.
    def capture(self, real, label, bbox, z, z_im=None):
        """Capture one iteration on static copies of the inputs (shapes are fixed: the padded-ROI form has no host
        syncs). Returns True on success; afterwards `step_graphed` copies new inputs in and replays.
        z = None: the latents are drawn INSIDE the captured iteration (torch's graph-safe generator advances its Philox
        offset per replay), as the reference draws them per iteration (train_context_app_v2.py:165)."""
        if self.dp:
            # Data parallel: the iteration's collectives (SyncBN statistics, ROI count, flat-gradient all-reduces) are RCCL
            # calls on communicator streams that fork from / join the capture stream, which torch can capture into the graph
            # like any other stream dependency -- but this path has never run on a multi-GPU node (none was available to
            # rounds 1-3) and gloo (the backend of the tests' shared-GPU ranks) cannot be captured at all, so it is OPT-IN:
            # L2I_DDP_GRAPH=1 on an RCCL process group. Every rank must take the same decision. Default: eager
            # (host enqueue 15.8 ms against 21 ms of GPU time per iteration: GPU-bound with ~25 % headroom, tools/perf/cpu_time.py).
            import torch.distributed as dist
            if os.environ.get("L2I_DDP_GRAPH", "0") != "1" or dist.get_backend() != "nccl":
                return False
            # A replayed iteration must be self-contained: the deferred generator step (all-reduce launched in one iteration,
            # waited for in the next, through work handles that would belong to the capture) is an eager-mode overlap. In the
            # graph the generator's all-reduce, its wait and its Adam step stay inside their own iteration.
            self.flush()
            self._defer_g_eager, self.defer_g = self.defer_g, False
            host = parallel.host_group()   # (collective: created before the capture, used for the "every capture succeeded" decision)
        if ops.TIMER is not None:
            raise RuntimeError("capture with the kernel timer on")
        self._static = [None if t is None else t.detach().clone() for t in (real, label, bbox, z, z_im)]
        from . import _lib
        cur = torch.cuda.current_stream()
        side = self._cap_stream = torch.cuda.Stream()   # the capture stream; also used for the warm-up, as torch.cuda.graphs asks
        if self.overlap and self._side is None:
            self._side = torch.cuda.Stream()
        for s_ in (side, self._side):   # reduction workspaces of both streams exist BEFORE capture (never in the graph's pool)
            if s_ is not None:
                with torch.cuda.stream(s_):
                    _lib.workspace(real.device)
                    _lib.wgrad_scratch(real.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(2):
                self.step(*self._static)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        if self.dp:
            # The warm-up's collectives are complete, but their work handles sit in the process groups' watchdog lists until the
            # watchdog threads' next poll (every 100 ms). A handle polled AFTER its communicator stream has joined the capture
            # fails the event query ("operation not permitted on an event last recorded in a capturing stream") and the
            # watchdog takes the process down -- measured: 4 of 11 runs of the one-rank RCCL test. Let the lists drain first.
            import time
            time.sleep(float(os.environ.get("L2I_CAPTURE_DRAIN_S", "0.5")))
        self._graph_packs = []   # (a new capture replaces the single-iteration graph and the multi-iteration one built on it)
        self._graph_multi = None
        self._lend_packs()
        graph = torch.cuda.CUDAGraph()
        err = None
        t_host = (self.g_opt.t, self.d_opt.t)   # host-side step counts: a capture that fails has advanced them without running anything
        try:
            # (data parallel: RCCL's watchdog thread polls its own events while this thread captures; in the default "global"
            #  capture mode any other thread's event query invalidates the capture -- hipErrorStreamCaptureInvalidated, measured)
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local" if self.dp else "global"):
                self._graph_out = self.step(*self._static)
                if os.environ.get("L2I_TEST_CAPTURE_FAIL", "0") == "1":   # (tests: an illegal call invalidates the capture)
                    torch.cuda.synchronize()
        except Exception as e:
            err = e
            self._graph = None
            if not self.dp:
                self._undo_failed_capture(t_host)
                raise
        for net in (self.netG, self.netD):
            net.arena.free_packs = []   # buffers from the graph's private pool must not be handed to eager iterations
        if self.dp:
            # The decision to replay is COLLECTIVE: a rank that replays while another runs eagerly would issue its collectives
            # in a different order and dead-lock the job. Every rank contributes "my capture succeeded"; MIN over the ranks -- of a
            # HOST tensor over gloo (parallel.host_group): the failing rank's HIP runtime may be unusable.
            if not parallel.all_ranks_ok(err is None, host):
                self._graph = None
                self.defer_g = self._defer_g_eager   # (back to the eager-mode overlap of the generator's all-reduce)
                if err is not None:
                    print(f"[layout2img_amd] graph capture failed on this rank ({type(err).__name__}: {str(err)[:160]}); every rank runs eagerly", flush=True)
                self._undo_failed_capture(t_host)
                return False
        self._graph = graph
        return True

    def _undo_failed_capture(self, t_host):
        """Nothing of a failed (or abandoned) capture ran, but its host-side traces exist: pass contexts in arena.pending whose dW-bar
        accumulators and packs live in the discarded graph pool and were never computed (the next eager flush_grads() would fold that
        garbage into flat.grad), captured work handles of the optimizers' all-reduces, a "generator step pending" flag, advanced step
        counts, and the pack buffers lent to the capture. Put the trainer back where it was before the capture."""
        try:
            torch.cuda.synchronize()
        except RuntimeError:
            pass   # (an invalidated capture can leave the runtime in a sticky error state: the caller re-launches eagerly / aborts)
        for net in (self.netG, self.netD):
            net.arena.drop_pending()
            net.arena.free_packs = []
        self._graph_packs = []
        self.g_opt._works = self.d_opt._works = None
        self._pending_g = False
        self.g_opt.t, self.d_opt.t = t_host

    _graph = None
    _graph_multi = None
    _cap_stream = None

    def capture_multi(self, batches):
        """ONE graph of len(batches) consecutive iterations (after a successful `capture`, which warmed the allocator and made the
        per-stream workspaces). A replay costs ~0.3 ms of launch on top of the 19.4 ms of kernels; n iterations per
        replay divide that by n (measured: 19.36 -> 19.27 ms per iteration at n = 2). batches: n tuples (real, label, bbox, z, z_im)
        -- each iteration has its own static inputs (`step_graphed_multi` copies n new batches in); z = None draws the latents inside."""
        if self._graph is None or self.dp:
            return False
        self._static_multi = [[None if t is None else t.detach().clone() for t in (tuple(b) + (None,) * 5)[:5]] for b in batches]
        cur = torch.cuda.current_stream()
        # capture()'s stream again: its reduction workspace and weight-gradient scratch (keyed by the raw stream, _lib.workspace)
        # exist already -- a fresh stream would allocate them INSIDE the capture, in the graph's private pool
        side = self._cap_stream
        side.wait_stream(cur)
        torch.cuda.synchronize()
        self._lend_packs()
        t_host = (self.g_opt.t, self.d_opt.t)
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, stream=side):
                self._multi_out = [self.step(*b) for b in self._static_multi]
        except Exception:
            self._graph_multi = None
            self._undo_failed_capture(t_host)
            self._graph = None   # (the lent buffers of the single-iteration graph went with the others: capture() again before replaying)
            raise
        self.g_opt.t, self.d_opt.t = t_host   # (nothing ran: the host-side mirrors of the step counts stay where they were)
        for net in (self.netG, self.netD):
            net.arena.free_packs = []
        self._graph_multi = graph
        return True

    def _lend_packs(self):
        """Before a capture: the pass contexts created inside it take their W / sigma pack buffers from here. A pack buffer must be
        zero in its padding rows and K tails (the pack kernel writes only true elements): allocated INSIDE the capture, every replay
        would clear 2 x 80 ... 130 M bf16 again (a memset node per buffer: 94 us per iteration, profiles/r05). These are cleared once,
        here, and kept alive by the trainer for as long as the graphs exist -- a graph writes to their addresses on every replay, so
        they must never go back to the allocator (the eager iterations' free list is emptied again right after the capture)."""
        keep = self.__dict__.setdefault("_graph_packs", [])
        for net, n in ((self.netG, 2), (self.netD, 3)):
            a = net.arena
            a.free_packs = [torch.zeros(a.packed_len, dtype=a.op_dtype, device=a.device) for _ in range(n)]
            for t in a.free_packs:
                t._l2i_lent = True   # (PassCtx.__del__ never hands a graph-owned buffer to an eager pass)
            keep += a.free_packs

    def step_graphed_multi(self, batches):
        """Copies len(batches) new batches into the static inputs of the multi-iteration graph and replays it; returns the list of
        the iterations' result dicts (static tensors, overwritten by the next replay)."""
        if len(batches) != len(self._static_multi):
            raise RuntimeError(f"step_graphed_multi: the graph holds {len(self._static_multi)} iterations, got {len(batches)} batches")
        for st, b in zip(self._static_multi, batches):
            for name, dst, src in zip(("real", "label", "bbox", "z", "z_im"), st, (tuple(b) + (None,) * 5)[:5]):
                if (dst is None) != (src is None):
                    raise RuntimeError(f"step_graphed_multi: `{name}` was {'drawn inside' if dst is None else 'an input of'} the captured iterations")
                if dst is not None and dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
        self._graph_multi.replay()
        self._touch()
        self.g_opt.t += len(batches)
        self.d_opt.t += len(batches)
        return self._multi_out

    def _touch(self):
        """A replay changed parameters and u / v with no host-side call: cached eval-mode packs are stale (arena._stamp)."""
        for net in (self.netG, self.netD):
            net.flat.touch()
            net.arena.sn_epoch += 1

    def step_graphed(self, real, label, bbox, z, z_im=None):
        for name, dst, src in zip(("real", "label", "bbox", "z", "z_im"), self._static, (real, label, bbox, z, z_im)):
            if (dst is None) != (src is None):
                raise RuntimeError(f"step_graphed: `{name}` was {'drawn inside' if dst is None else 'an input of'} the captured iteration")
            if dst is not None and dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self._graph.replay()
        self._touch()
        self.g_opt.t += 1   # (host-side mirrors of the device-side step counts the replayed Adam launches advance)
        self.d_opt.t += 1
        return self._graph_out


def snapshot_state(tr):
    """Everything one training iteration changes (parameters, spectral-norm u / v, batch-norm buffers, Adam moments and
    step counts) as clones: `restore_state` puts a trainer back, e.g. to compare two ways of running the SAME iteration."""
    nets = (tr.netG, tr.netD)
    return dict(flat=[n.flat.data.clone() for n in nets], sn=[n.arena.sn_flat.data.clone() for n in nets],
                bufs=[[b.detach().clone() for b in n.buffers()] for n in nets],
                opt=[(o.m.clone(), o.v.clone(), o.t, o.t_dev.clone()) for o in (tr.g_opt, tr.d_opt)])


def restore_state(tr, st):
    nets = (tr.netG, tr.netD)
    with torch.no_grad():
        for n, f, s_, bufs in zip(nets, st["flat"], st["sn"], st["bufs"]):
            n.flat.data.copy_(f)
            n.arena.sn_flat.data.copy_(s_)
            for b, v in zip(n.buffers(), bufs):
                b.copy_(v)
            n.arena.drop_pending()
            n.flat.fresh = False   # (whatever the undone iterations left in the gradient buffers: the next zero_grad() resets it)
        tr._pending_g = False      # a deferred generator step of the undone iterations is dropped with them
        tr.g_opt._works = tr.d_opt._works = None
        for o, (m, v, t, td) in zip((tr.g_opt, tr.d_opt), st["opt"]):
            o.m.copy_(m), o.v.copy_(v), o.t_dev.copy_(td)
            o.t = t

