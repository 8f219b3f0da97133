"""CPU: the host side of the whole training iteration, dry (tests/dryrun.py): meta-device tensors, the C ABI replaced by a recorder that checks
every call against the ctypes table. What a GPU run would only show as a crash or as garbage on the stack -- a call site that still passes
last round's argument list, an int that no longer fits 32 bits, a host read in the middle of the iteration -- fails here, for every
configuration of BASELINE.json, in seconds. The launch census is tied to the measured profile: the 186 convolution launches per iteration
that `roofline` in bench.py's line averages over, the 70 weight-gradient launches of `wgrad_frac`."""
import collections

import pytest
import torch

from tests import dryrun


def _iterations(tr, inputs, n=3, z_from_host=True):
    real, label, bbox, z, z_im = inputs
    traces = []
    with dryrun.dry_run() as trace:
        for _ in range(n):
            del trace[:]
            tr.step(real, label, bbox, z if z_from_host else None, None)
            traces.append(list(trace))
    return traces


def _static(trace):
    """the trace without the one argument that legitimately changes from iteration to iteration on the host: Adam's step count (eager calls pass
    it by value for the bias correction; a captured iteration passes `step_ptr`, a device counter, instead -- include/l2i.h)"""
    names = dryrun.header_parameters()["l2i_adam_step"]
    k = names.index("step")
    return [(n, a[:k] + (None,) + a[k + 1:]) if n == "l2i_adam_step" else (n, a) for n, a in trace]


def _census(trace):
    return collections.Counter(name for name, _ in trace)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_coco_iteration_dry_run_launch_census_and_steady_state(dtype):
    from layout2img_amd import _lib
    tr, inputs = dryrun.build("coco", dtype)
    t = _iterations(tr, inputs)
    # steady state from the first iteration on: the same calls with the same integer / float arguments (pointers are all 0 on the meta
    # device) -- the iteration is one static launch sequence, which is what GanTrainer.capture relies on
    assert _static(t[0]) == _static(t[1]) == _static(t[2])
    assert [a[9] for n, a in t[2] if n == "l2i_adam_step"] == [3, 3]   # D's and G's third step
    c = _census(t[1])
    conv = c["l2i_conv2d_fwd"] + c["l2i_conv2d_fwd_dual"] + c["l2i_conv2d_dgrad_sc"]
    # bench.py's roofline averages over exactly these (profiles/r06_bench_final.json: "launches": 186 per iteration; 70 weight-gradient launches)
    # (exact-f32: + 24, the 1x1 shortcuts' data gradients are launches of their own there -- l2i_conv2d_dgrad_sc is a bf16 kernel)
    assert conv == (186 if dtype == torch.bfloat16 else 210) and c["l2i_conv2d_wgrad"] + c["l2i_conv2d_wgrad_dual"] == 70, c
    assert c["l2i_conv2d_dgrad_sc"] == (24 if dtype == torch.bfloat16 else 0)
    assert c["l2i_adam_step"] == 2 and c["l2i_weights_prepare"] == 7 and c["l2i_weights_backward2"] == 2
    assert c["l2i_norm_mod_fwd"] == c["l2i_norm_mod_bwd_a"] == c["l2i_norm_bwd_b"] == 16   # G: 5 blocks x 2 + mask-head / final norms, x 1 backward
    assert c["l2i_hinge_fwd_bwd"] == 9     # 3 D passes x (image, object, appearance) heads: loss and its gradient in one launch
    # round 6's contract "no float atomics": every bf16 convolution launch and every launch that gathers batch statistics is handed the
    # stream's scratch for its partial rows (a launch without it falls back to the atomic epilogue inside the library)
    names = dryrun.header_parameters()
    for name, args in t[1]:
        a = dict(zip(names[name], args))
        if name == "l2i_conv2d_fwd_dual" and not (a["dtype"] == _lib.BF16 or a["stats"] is not None):
            continue                                  # (an exact-f32 launch without statistics has nothing to store)
        for ptr, size in (("scratch", "scratch_floats"), ("part", "part_floats")):
            if size in a:
                assert a[ptr] is not None and a[size] == _lib.WGRAD_SCRATCH_FLOATS, (name, ptr, a)
    if dtype == torch.float32:
        assert c["l2i_conv2d_fwd"] > 0      # exact-f32 launches without statistics take the plain entry point
    else:
        assert c["l2i_conv2d_fwd"] == 0


def test_latents_drawn_inside_the_iteration_and_the_other_d_step_forms():
    """z = None (the captured form draws the latents on the device), the dual D step (D(real) and D(fake) as one batch) and the early
    D(real) backward issue the same set of entry points with valid arguments; the dual form merges D's two forward passes."""
    tr, inputs = dryrun.build("coco", torch.bfloat16)
    base = _census(_iterations(tr, inputs, n=2)[1])
    assert _census(_iterations(tr, inputs, n=2, z_from_host=False)[1]) == base
    tr.real_bwd_early = True
    early = _census(_iterations(tr, inputs, n=2)[1])
    assert early == base          # the same launches, in another order
    tr.real_bwd_early = False
    tr.dual_d = True
    dual = _census(_iterations(tr, inputs, n=2)[1])
    conv = lambda c: c["l2i_conv2d_fwd"] + c["l2i_conv2d_fwd_dual"] + c["l2i_conv2d_dgrad_sc"]
    assert conv(dual) < conv(base) and dual["l2i_weights_prepare"] == base["l2i_weights_prepare"]
    tr.overlap = False
    tr.dual_d = False
    assert _census(_iterations(tr, inputs, n=2)[1]) == base   # one stream: the same launches


def test_vg_iteration_and_perceptual_loss_dry_run():
    """BASELINE config 5 (o = 31, 179 classes, z_im) and the VGG feature loss in the G step."""
    tr, inputs = dryrun.build("vg", torch.bfloat16, vgg=True)
    real, label, bbox, z, z_im = inputs
    with dryrun.dry_run() as trace:
        for _ in range(2):
            del trace[:]
            tr.step(real, label, bbox, z, z_im)
    c = _census(trace)
    assert c["l2i_adam_step"] == 2 and c["l2i_up2_nhwc_fwd"] == c["l2i_up2_nhwc_bwd"] > 0   # the mask regressor's bilinear x2: a HIP kernel (round 6)
    assert c["l2i_conv2d_fwd_dual"] + c["l2i_conv2d_fwd"] > 186   # + 13 VGG layers x (fake, real) and their data gradients


def test_res64_iteration_dry_run():
    """BASELINE config 0 (64 x 64)."""
    tr, inputs = dryrun.build("coco", torch.float32, size=64)
    t = _iterations(tr, inputs)
    assert _static(t[0]) == _static(t[1]) == _static(t[2])
    assert _census(t[1])["l2i_adam_step"] == 2


def test_inference_forward_dry_run():
    """generator.eval() forward (the sampler's path): no gradient bookkeeping, no weight-gradient or backward entry point."""
    tr, inputs = dryrun.build("coco", torch.bfloat16)
    real, label, bbox, z, z_im = inputs
    g = tr.netG.eval()
    with dryrun.dry_run() as trace, torch.no_grad():
        img = g(z, bbox, y=label.view(4, 8))
    assert tuple(img.shape) == (4, 3, 128, 128)
    names = [n for n, _ in trace]
    assert not [n for n in names if "bwd" in n or "wgrad" in n or "dgrad" in n], names
    g.train()


def test_dry_run_leaves_no_trace_behind():
    from layout2img_amd import _lib, ops
    call = _lib.call
    with dryrun.dry_run():
        assert _lib.call is not call and torch.empty(1, device="meta").is_cuda
    assert _lib.call is call and not torch.empty(1).is_cuda and not ops.POOL.active
    assert torch.cuda.Stream.__module__.startswith("torch")


def test_two_rank_iteration_communication_schedule():
    """The N > 1 iteration (eager, one process per GPU) at world size 2, dry: which collectives one iteration issues. bench.py's `comm` object
    measured exactly this schedule on the GPU (42 collectives, 414 MB of gradient all-reduce per iteration: DESIGN.md section 5); here it is
    pinned without a second process: the SyncBN statistics and the ROI count block on the default group, the flat gradients go out
    asynchronously in chunks on the group of their own, every gradient byte exactly once, and the kernel launches are those of N = 1."""
    with dryrun.dry_run() as trace, dryrun.dry_ddp(2) as coll:
        tr, inputs = dryrun.build("coco", torch.bfloat16)
        assert tr.dp and tr.world == 2
        assert [c[0] for c in coll] == ["broadcast"] * len(coll) and len(coll) >= 4   # parameters + power-iteration state, from rank 0
        real, label, bbox, z, z_im = inputs
        per_it = []
        for _ in range(3):
            del coll[:], trace[:]
            tr.step(real, label, bbox, z, None)
            per_it.append((list(coll), list(trace)))
        tr.flush()
    # (the generator's exchange + Adam are deferred into the next iteration: iteration 0 lacks them, 1 and 2 are the steady state)
    assert per_it[1][0] == per_it[2][0] and _static(per_it[1][1]) == _static(per_it[2][1])
    coll, census = per_it[1][0], _census(per_it[1][1])
    assert len(coll) == 42
    grads = [c for c in coll if c[2]]
    blocking = [c for c in coll if not c[2]]
    assert all(c[3] for c in grads) and not any(c[3] for c in blocking)   # own group <=> asynchronous gradient chunk
    n_params = tr.netG.flat.numel + tr.netD.flat.numel
    assert sum(c[1] for c in grads) == 4 * n_params                      # every gradient exactly once (414.3 MB)
    assert abs(sum(c[1] for c in grads) / 1e6 - 414.3) < 0.1
    assert len(blocking) == 31 and max(c[1] for c in blocking) <= 2 * 4 * 1024   # 30 x [sum | sqsum] / [s1 | s2] of a norm layer, 1 x the ROI count
    conv = census["l2i_conv2d_fwd"] + census["l2i_conv2d_fwd_dual"] + census["l2i_conv2d_dgrad_sc"]
    assert conv == 186
    # Adam runs chunk by chunk behind the chunks of the exchange: every parameter stepped exactly once
    assert sum(a[4] for n, a in per_it[1][1] if n == "l2i_adam_step") == n_params and census["l2i_adam_step"] > 2


def test_roofline_numerators_of_the_committed_bench_line_reproduce_without_a_gpu():
    """bench.py's `roofline.achieved` = accounted algorithmic FLOPs / HIP-event time. The numerator is host arithmetic (ops.KernelTimer: every
    conv / weight-gradient launch adds its 2 M N K and its operand / result bytes, ROI launches scaled by the live-ROI fraction): the dry run
    repeats bench.py's instrumented iteration at the headline batch (same seed, same batch, one stream) and must arrive at the launches,
    GFLOP per launch and algorithmic bytes per launch that the committed line of the round carries."""
    import json
    import os
    from layout2img_amd import ops
    from layout2img_amd.synthetic import make_batch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = json.load(open(os.path.join(root, "profiles", "r06_bench_final_b.json")))
    roof = line["roofline"]
    assert line["config"]["global_batch"] == 32 and line["n_gpus"] == 1
    real, label, bbox, z, z_im = make_batch(32, 128, "coco", seed=1234, device="cpu")
    live = float((label != 0).sum()) / label.numel()
    assert round(live, 4) == roof["live_roi_fraction"]
    saved = ops.LIVE_IMAGE_FRACTION, ops.TIMER
    try:
        with dryrun.dry_run() as trace:
            tr, _ = dryrun.build("coco", torch.bfloat16)
            real, label, bbox = (t.to("meta") for t in (real, label, bbox))
            ops.LIVE_IMAGE_FRACTION = live
            tr.overlap = False
            tr.step(real, label, bbox, None, None)
            tr.flush()
            del trace[:]
            ops.TIMER = ops.KernelTimer()
            tr.step(real, label, bbox, None, None)
            tr.flush()
            acc = {k: list(v) for k, v in ops.TIMER.acc.items()}
    finally:
        ops.LIVE_IMAGE_FRACTION, ops.TIMER = saved
    n, work, nbytes = acc["conv_igemm"]
    assert n == roof["launches_per_step"] == 186
    assert round(work / n / 1e9, 3) == roof["gflop_per_launch"]
    assert round(nbytes / n) == roof["algorithmic_bytes_per_launch"]
    assert acc["conv_wgrad"][0] == roof["wgrad_launches_per_step"] == 70
    # and the line is consistent with itself: achieved = GFLOP per launch / average launch time; frac = achieved / peak
    assert abs(roof["gflop_per_launch"] / roof["avg_launch_us"] * 1e3 - roof["achieved"]) < 1e-3 * roof["achieved"]   # (GFLOP / us = 1000 TFLOP/s)
    assert abs(roof["achieved"] / roof["peak"] - roof["frac"]) < 1e-3
    # 26.35 GFLOP per generated image, forward
    assert line["g_forward"]["gflop_per_image"] == 26.35
    # the accounting never credits more than the launches are dimensioned for: 2 B Ho Wo Co KH^2 Ci (+ the folded 1x1 shortcut) of every
    # launch's ARGUMENTS (channel counts padded to the kernels' multiples there) bounds the accounted work from above, closely
    names = dryrun.header_parameters()
    dim = {"conv": 0.0, "wgrad": 0.0}
    for name, args in trace:
        a = dict(zip(names[name], args))
        if name in ("l2i_conv2d_fwd_dual", "l2i_conv2d_wgrad_dual"):
            f = 2.0 * a["B"] * a["Ho"] * a["Wo"] * a["Co"] * (a["KH"] ** 2 * a["Ci"] + (a["sc_Ci"] if a["sc_x"] is not None else 0))
        elif name == "l2i_conv2d_dgrad_sc":
            f = 2.0 * a["B"] * a["H"] * a["W"] * a["Co"] * (9 * a["Ci"] + a["sc_Ci"])
        else:
            continue
        dim["wgrad" if "wgrad" in name else "conv"] += f * (live if a["nimg"] is not None else 1.0)
    assert 0.99 * dim["conv"] < work <= dim["conv"] and 0.99 * dim["wgrad"] < acc["conv_wgrad"][1] <= dim["wgrad"], (dim, acc)


def test_steady_state_holds_buffer_by_buffer_and_for_the_stock_operators():
    """The stronger form of the steady-state check: with fake addresses (dry_run(pointers=True): storage number << 40 + byte offset) and the
    stock torch operators recorded between the C-ABI calls, iterations 2 and 3 hand the GPU the same work on the same buffers-by-role --
    every launch reads and writes the same slices of the same arena buffers, every temporary is created and consumed at the same place.
    And within one iteration no call is handed the null address where the header expects a tensor it was given."""
    with dryrun.dry_run(pointers=True, aten=True) as trace:
        tr, (real, label, bbox, z, z_im) = dryrun.build("coco", torch.bfloat16)
        its = []
        for _ in range(3):
            del trace[:]
            tr.step(real, label, bbox, z, None)
            its.append(list(trace))
    a, b = (dryrun.canonical(_static(t)) for t in its[1:])
    assert a == b and len(a) > 1500
    calls = [(n, args) for n, args in its[2] if n.startswith("l2i_")]
    assert len(calls) == 433
    # every conv launch's weight pack lies inside ONE buffer per pass (the pass's pack arena) and inside its bounds
    names = dryrun.header_parameters()
    packs = {}
    for n, args in calls:
        if n == "l2i_conv2d_fwd_dual":
            w = dict(zip(names[n], args))["w"]
            packs.setdefault(w >> 40, []).append(w & ((1 << 40) - 1))
    assert 3 <= len(packs) <= 8          # D(real), D(fake), D(G-step), G (+ the eval-mode VGG has none here)
    for offs in packs.values():
        assert max(offs) < 2 * max(tr.netD.arena.packed_len, tr.netG.arena.packed_len)


@pytest.mark.parametrize("cfg", [dict(kind="coco", dtype=torch.bfloat16), dict(kind="coco", dtype=torch.float32), dict(kind="vg", dtype=torch.bfloat16, vgg=True),
                                 dict(kind="coco", dtype=torch.float32, size=64)], ids=["coco-bf16", "coco-f32", "vg-vgg", "res64"])
def test_no_launch_reads_a_temporary_that_nothing_has_written(cfg):
    """include/l2i.h says which pointer parameters are inputs (`const T*`); the traced iteration says which buffers are fresh `torch.empty`
    temporaries. No call of the iteration may name such a buffer as an input before some call has produced it (the weight-gradient
    accumulators and the lazily folded shortcut's placeholder are made by torch.empty on purpose: their first user must be their writer).
    The detector is shown to see: declaring the convolution's `out` an input makes every consumer chain light up."""
    with dryrun.dry_run(pointers=True, aten=True) as trace:
        tr, (real, label, bbox, z, z_im) = dryrun.build(**cfg)
        its = []
        for _ in range(3):
            del trace[:]
            tr.step(real, label, bbox, z, z_im if cfg["kind"] == "vg" else None)
            its.append(list(trace))
    assert dryrun.uninitialised_reads(its[1], its[2]) == []
    kinds = dryrun.header_pointer_kinds()
    entry = "l2i_conv2d_fwd_dual"
    k = dryrun.header_parameters()[entry].index("out")
    assert kinds[entry][k] == "out"
    kinds[entry] = kinds[entry][:k] + ["in"] + kinds[entry][k + 1:]
    assert len(dryrun.uninitialised_reads(its[1], its[2], kinds)) > 50


@pytest.mark.parametrize("dual", [False, True], ids=["two-passes", "dual"])
def test_opt_in_dead_f32_streams_are_read_by_nothing(monkeypatch, dual):
    """L2I_F32_DEAD=1 (ops.F32_DEAD, opt-in until it has run on a GPU): the discriminator blocks whose results are read through the emitted
    operand copies only hand the convolution no f32 result pointer; the tensor that remains the autograd edge is never written -- so nothing
    may read it: the detector of unwritten reads stays empty, at the test batch and at the headline batch."""
    from layout2img_amd import ops
    from layout2img_amd.synthetic import make_batch
    monkeypatch.setattr(ops, "F32_DEAD", True)
    names = dryrun.header_parameters()["l2i_conv2d_fwd_dual"]
    for batch in (4, 32):
        with dryrun.dry_run(pointers=True, aten=True) as trace:
            tr, _ = dryrun.build("coco", torch.bfloat16)
            tr.dual_d = dual
            real, label, bbox, z, z_im = (t.to("meta") for t in make_batch(batch, 128, "coco", seed=1234, device="cpu"))
            its = []
            for _ in range(3):
                del trace[:]
                tr.step(real, label, bbox, z, None)
                its.append(list(trace))
        assert dryrun.uninitialised_reads(its[1], its[2]) == []
        copies_only = [a for n, a in its[2] if n == "l2i_conv2d_fwd_dual" and (lambda d: d["out"] is None and d["out_op"] is not None and d["out_op_raw"] is not None)(dict(zip(names, a)))]
        assert len(copies_only) == (10 if dual else 15)   # block1-4 and block_obj3 of every D pass (the dual step runs two passes as one)


def test_inference_forward_reads_no_unwritten_temporary():
    """The sampler's path (eval mode, no gradients; cached eval-mode weight packs) under the same detector."""
    with dryrun.dry_run(pointers=True, aten=True) as trace, torch.no_grad():
        tr, (real, label, bbox, z, z_im) = dryrun.build("coco", torch.bfloat16)
        g = tr.netG.eval()
        runs = []
        for _ in range(3):
            del trace[:]
            g(z, bbox, y=label.view(4, 8))
            runs.append(list(trace))
        g.train()
    assert dryrun.uninitialised_reads(runs[1], runs[2]) == []
    assert dryrun.canonical(runs[1]) == dryrun.canonical(runs[2])
    assert not any(n == "l2i_weights_prepare" for n, _ in runs[2])   # the eval-mode packs are cached: no power iteration, no repack per call
