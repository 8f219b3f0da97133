"""Measured values behind the bf16 / graph / data-parallel test bars (run on the GPU box; the tests hold <= 1.5 x these):
 1. two iterations of the training loop with bf16 operands against the reference-captured losses / images (COCO, VG);
 2. graph replay vs eager at 128x128, b = 32, bf16: per-parameter relative L2 error of the flat GRADIENTS after one
    iteration from the same state (and eager vs eager: the run-to-run noise floor of atomics + ReLU gates)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import layout2img_amd as L
from tests.golden import recipe
from tests.helpers import load_fixture, maxdiff
from tests import test_gpu_00_models as T
DEV = "cuda:0"


def loop(kind, dt):
    vg = kind == "vg"
    fx = load_fixture("train_loop_vg.npz" if vg else "train_loop.npz")
    g = T._build_g(load_fixture("g_vg_img.npz" if vg else "g_coco.npz"), 53 if vg else 31, dt, kind="vg" if vg else "coco")
    d = T._build_d(load_fixture("d_vg.npz" if vg else "d_coco.npz"), 54 if vg else 32, dt, num_classes=179 if vg else 184)
    g.train(), d.train()
    tr = L.GanTrainer(g, d)
    out = {}
    for it in range(2):
        mk = recipe.make_inputs_vg(2, 31, 179, 300 + it) if vg else recipe.make_inputs(2, 8, 184, 200 + it)
        inp = {k: v.to(DEV) for k, v in mk.items()}
        r = tr.step(inp["real"], inp["y"], inp["bbox"], inp["z"], inp["z_im"])
        for k in ("d_loss", "g_loss"):
            ref = float(fx[f"{k}{it}"])
            out[f"{k}{it}_rel"] = abs(float(r[k]) - ref) / max(1.0, abs(ref))
        out[f"fake{it}_linf"] = maxdiff(r["fake"][:, :, ::4, ::4], fx[f"fake_sub{it}"])
    for net, pre in ((g, "g"), (d, "d")):
        named = dict(net.named_parameters())
        names = [str(n) for n in fx[f"{pre}_param_names"]]
        sums = np.array([float(named[n].detach().double().sum()) for n in names])
        numel = np.array([named[n].numel() for n in names])
        err = np.abs(sums - fx[f"{pre}_param_sums"])
        out[f"{pre}_param_sum_worst_over_lr_numel"] = float((err / (2e-4 * numel)).max())   # in units of "every element moved by 2 lr"
    return out


def grads_after_one_step(dt, mode, seed=5, size=128, b=32):
    from layout2img_amd.synthetic import make_batch
    torch.manual_seed(seed)
    g = L.ResnetGenerator128_context(num_classes=184).finalize(DEV, dt)
    d = L.CombineDiscriminator128_app(num_classes=184).finalize(DEV, dt)
    for m in g.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
    tr = L.GanTrainer(g, d)
    real, label, bbox, z, z_im = make_batch(b, size, "coco", seed=3, device=DEV)
    if mode == "graph":
        from layout2img_amd.trainer import snapshot_state, restore_state
        st = snapshot_state(tr)
        assert tr.capture(real, label, bbox, z, z_im)
        restore_state(tr, st)
        tr.step_graphed(real, label, bbox, z, z_im)
    else:
        tr.step(real, label, bbox, z, z_im)
    torch.cuda.synchronize()
    return {("g." + n): p.grad.detach().float().cpu().clone() for n, p in g.named_parameters()} | \
           {("d." + n): p.grad.detach().float().cpu().clone() for n, p in d.named_parameters()}


def rel_l2(a, b):
    out = {}
    for k in a:
        nb = float(b[k].norm())
        if nb > 0:
            out[k] = float((a[k] - b[k]).norm()) / nb
    return out


if __name__ == "__main__":
    what = sys.argv[1:] or ["loop", "graph"]
    if "loop" in what:
        for kind in ("coco", "vg"):
            for dt in (torch.float32, torch.bfloat16):
                print("loop", kind, str(dt).split(".")[-1], {k: f"{v:.3g}" for k, v in loop(kind, dt).items()}, flush=True)
    if "graph" in what:
        for dt in (torch.bfloat16, torch.float32):
            e1 = grads_after_one_step(dt, "eager")
            e2 = grads_after_one_step(dt, "eager")
            gr = grads_after_one_step(dt, "graph")
            for name, r in (("eager vs eager", rel_l2(e2, e1)), ("graph vs eager", rel_l2(gr, e1))):
                v = np.array(sorted(r.values()))
                worst = sorted(r.items(), key=lambda kv: -kv[1])[:4]
                print(str(dt).split(".")[-1], name, f"per-parameter rel L2: median {np.median(v):.3g}  p90 {v[int(0.9 * len(v))]:.3g}  max {v[-1]:.3g}",
                      [(k, f"{x:.3g}") for k, x in worst], flush=True)
            tot = lambda t: torch.cat([x.reshape(-1) for x in t.values()])
            for name, a in (("eager vs eager", e2), ("graph vs eager", gr)):
                print(str(dt).split(".")[-1], name, "whole-gradient rel L2:", f"{float((tot(a) - tot(e1)).norm() / tot(e1).norm()):.3g}", flush=True)
