"""Run-to-run determinism of the exact-f32 mode (and, for comparison, the bf16-operand mode): the same launch sequence twice from the same
state -- generator forward (train mode, every tap), discriminator forward, their flat gradients, and N trainer iterations.
Prints max |a - b| (forward) / relative L2 distance (gradients, parameters); 0 = bit-identical.   python tools/parity/determinism_probe.py [--dtype f32|bf16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L   # noqa: E402
from layout2img_amd.synthetic import make_batch   # noqa: E402

DEV = torch.device("cuda:0")


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--layout", default="coco")   # vg: context_aware_generator, 31 object slots (BASELINE config 5)
    a = ap.parse_args()
    dt = torch.float32 if a.dtype == "f32" else torch.bfloat16
    torch.manual_seed(0)
    G = L.ResnetGenerator128_context if a.size == 128 else L.ResnetGenerator64_context
    D = L.CombineDiscriminator128_app if a.size == 128 else L.CombineDiscriminator64
    ncls = 184
    if a.layout == "vg":
        G, ncls = L.context_aware_generator, 179
    g = G(num_classes=ncls).finalize(DEV, dt).train()
    d = D(num_classes=ncls).finalize(DEV, dt).train()
    for m in g.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
    real, label, bbox, z, z_im = make_batch(a.batch, a.size, a.layout, seed=3, device=DEV)
    gs = {k: v.clone() for k, v in g.state_dict().items()}
    gsn = g.arena.sn_flat.data.clone()
    ds = {k: v.clone() for k, v in d.state_dict().items()}
    dsn = d.arena.sn_flat.data.clone()

    def reset():
        g.load_state_dict(gs); g.arena.sn_flat.data.copy_(gsn); g.arena.drop_pending(); g.zero_grad()
        d.load_state_dict(ds); d.arena.sn_flat.data.copy_(dsn); d.arena.drop_pending(); d.zero_grad()

    def g_pass():
        reset()
        taps = {}
        img = g(z, bbox, z_im, label, taps=taps) if (a.size == 128 and a.layout == "coco") else g(z, bbox, z_im, label)
        (img * real).sum().backward()
        g.arena.flush_grads()
        torch.cuda.synchronize()
        flat = [img.detach().clone()]
        for k, v in sorted(taps.items()):
            for t in (v if isinstance(v, (list, tuple)) else [v]):
                if torch.is_tensor(t):
                    flat.append(t.detach().clone())
        return flat, g.flat.grad.clone(), g.arena.sn_flat.data.clone()

    def d_pass():
        reset()
        o = d(real, bbox, label.unsqueeze(-1) if label.dim() == 2 else label)
        (o[0].sum() + o[1].sum() + 0.5 * o[2].sum()).backward()
        d.arena.flush_grads()
        torch.cuda.synchronize()
        return [t.detach().clone() for t in o], d.flat.grad.clone(), d.arena.sn_flat.data.clone()

    for name, f, net in (("G", g_pass, g), ("D", d_pass, d)):
        (o1, g1, s1), (o2, g2, s2) = f(), f()
        print(f"{name} {a.dtype} b={a.batch}: forward max|diff| per output/tap {[float((x - y).abs().max()) for x, y in zip(o1, o2)]}")
        print(f"{name}   power-iteration state rel {rel(s1, s2):.3e}   flat gradient rel {rel(g1, g2):.3e}")
        # per parameter: which gradients differ between the two runs (views of the flat gradient buffer, in registration order)
        base = net.flat.grad.data_ptr()
        rows = []
        for n, p_ in net.named_parameters():
            off = (p_.grad.data_ptr() - base) // 4
            x, y = g1[off:off + p_.numel()], g2[off:off + p_.numel()]
            rows.append((rel(x, y) if float(y.norm()) > 0 else float((x - y).abs().max()), n, tuple(p_.shape)))
        same = sum(1 for r in rows if r[0] == 0.0)
        print(f"{name}   {same} of {len(rows)} parameter gradients bit-identical; the others:")
        for r in rows:
            if r[0] != 0.0:
                print(f"        {r[0]:.2e}  {r[1]}  {r[2]}")

    def train(n):
        reset()
        torch.manual_seed(5)
        tr = L.GanTrainer(g, d)
        for _ in range(n):
            r = tr.step(real, label, bbox, z, z_im)
        tr.flush()
        torch.cuda.synchronize()
        return g.flat.data.clone(), d.flat.data.clone(), float(r["d_loss"]), float(r["g_loss"])
    t1, t2 = train(a.iters), train(a.iters)
    print(f"trainer {a.iters} iterations twice: G params rel {rel(t1[0], t2[0]):.3e}  D params rel {rel(t1[1], t2[1]):.3e}  losses {t1[2:]} {t2[2:]}")


if __name__ == "__main__":
    main()
