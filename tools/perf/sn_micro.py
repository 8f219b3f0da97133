"""Spectral-norm / weight-arena kernels of the discriminator (62.7 M parameters) and the generator (40.9 M) in isolation:
per-kernel average duration and the HBM rate of the bytes each one has to move (W read once per phase; the pack phase
also writes both operand packs; backward: G and W read, then G read + gradient read-modify-write).
usage: python tools/perf/sn_micro.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for name, net in (("D", L.CombineDiscriminator128_app(num_classes=184)), ("G", L.ResnetGenerator128_context(num_classes=184))):
    net = net.finalize(dev, torch.bfloat16)
    ar = net.arena
    nparam = ar.flat.data.numel()
    for _ in range(2):
        p = ar.prepare(training=True)
        p.dw().normal_()
        ar.flush_grads()
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        for _ in range(reps):
            p = ar.prepare(training=True)
            _ = p.dw()
            ar.flush_grads()
        torch.cuda.synchronize()
    mb = nparam * 4 / 1e6
    moved = {"sn_wtu": mb, "sn_wv": mb, "sn_pack": mb * 2, "sn_dot": 2 * mb, "sn_apply": 3 * mb}
    print(f"{name}: {nparam / 1e6:.1f} M parameters = {mb:.0f} MB f32")
    for k in prof.key_averages():
        if k.device_time_total <= 0:
            continue
        us = k.device_time_total / k.count
        tag = next((m for m in moved if k.key.startswith(m) or (" " + m) in k.key), None)
        rate = f"  {moved[tag] / us:5.2f} TB/s of the nominal {moved[tag]:.0f} MB" if tag else ""
        print(f"   {k.key[:60]:60s} x{k.count:3d}  avg {us:8.1f} us{rate}")
