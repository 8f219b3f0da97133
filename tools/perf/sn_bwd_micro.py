"""l2i_weights_backward2 (spectral-norm backward: sn_dot + sn_apply) in isolation, on D's and G's layer tables with two / one pending passes:
time per flush and the bytes it has to move (W once, each pass's dWbar twice, the gradient buffer once).
    python tools/perf/sn_bwd_micro.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L
DEV = "cuda:0"


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, net, npass in (("D x2", L.CombineDiscriminator128_app(num_classes=184), 2), ("D x1", L.CombineDiscriminator128_app(num_classes=184), 1), ("G x1", L.ResnetGenerator128_context(num_classes=184), 1)):
    torch.manual_seed(0)
    net.finalize(DEV, torch.bfloat16).train()
    a = net.arena
    pcs = [a.prepare(training=True, need_wgrad=True) for _ in range(npass)]
    for p in pcs:
        p.dw().normal_()
        p.written = None
    net.zero_grad()

    def flush():
        a.pending = list(pcs)
        a.flat.fresh = True
        a.flush_grads()
    us = t(flush)
    wbytes = 4 * a.flat.numel
    dwb = 4 * a.dw_len
    need = wbytes + 2 * npass * dwb + wbytes
    print(f"{name}: flush {us:.1f} us; params {wbytes / 1e6:.0f} MB, dWbar {dwb / 1e6:.0f} MB per pass; bytes to move {need / 1e6:.0f} MB -> {need / us * 1e-6:.2f} TB/s")
