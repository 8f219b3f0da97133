"""Where do the small torch kernels of one iteration come from? (~500 launches of ~5 us each in the r02 kernel stats.)
Runs one eager iteration under torch.profiler with python stacks and prints the aten ops by device time, with the
innermost layout2img_amd frame that issued them."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["L2I_OVERLAP"] = "0"
import layout2img_amd as L  # noqa: E402
from layout2img_amd.synthetic import make_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    netG = L.ResnetGenerator128_context(num_classes=184).finalize(dev, torch.bfloat16)
    netD = L.CombineDiscriminator128_app(num_classes=184).finalize(dev, torch.bfloat16)
    tr = L.GanTrainer(netG, netD)
    batch = make_batch(32, 128, "coco", seed=1234, device=dev)[:4]
    for _ in range(2):
        tr.step(*batch)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        tr.step(*batch)
        torch.cuda.synchronize()
    agg, tim = collections.Counter(), collections.Counter()
    for ev in prof.events():
        if not ev.kernels:
            continue
        if not (ev.name.startswith("aten::") or "Backward" in ev.name):
            continue
        frame, par = "", ev.cpu_parent
        while par is not None:
            if "evaluate_function" in par.name or par.name.endswith("Fn") or par.name.endswith("FnBackward"):
                frame = par.name.replace("autograd::engine::evaluate_function: ", "bwd:")
            par = par.cpu_parent
        if not frame:
            frs = [s.split("layout2img_amd/")[-1].split("(")[0] + ":" + s.split("(")[-1].split(")")[0] for s in (ev.stack or [])
                   if "layout2img_amd/" in s]
            frame = " < ".join(frs[:3]) or "<py>"
        shp = str(ev.input_shapes[:2])[:50] if ev.input_shapes else ""
        k = (ev.name, frame, shp)
        agg[k] += 1
        tim[k] += sum(kk.duration for kk in ev.kernels)
    print(f"aten ops with kernels: {sum(agg.values())} launches, {sum(tim.values()) / 1e3:.2f} ms")
    for k, t in tim.most_common(400):
        print(f"{agg[k]:4d} {t:8.1f}us {k[0]:26s} {k[1][:80]:80s} {k[2]}")


if __name__ == "__main__":
    main()
