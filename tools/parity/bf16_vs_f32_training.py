"""Does training with bf16 MFMA operands track training with exact-f32 operands?  (reference loop: train_context_app_v2.py:148-189)

Three runs of N iterations of GanTrainer.step from the SAME initial state, the SAME pool of device-resident synthetic batches and
the SAME latent draws (torch seed reset per run), Dropout2d off:
    f32 A, f32 B  -- exact-f32 MFMA operands twice: their difference is the FLOOR (atomically reduced sums are not
                     order-deterministic; a GAN iteration amplifies 1e-7 differences through ReLU gates and Adam(beta1 = 0)'s
                     sign-like first steps)
    bf16          -- the throughput mode (BASELINE config 3)
Reported per checkpoint iteration t: the losses of each run, their means over the window since the last checkpoint, and the
parameter distance  d(run, f32 A) / d(f32 A, start)  for G and D -- for bf16 against f32 A and, as the floor, for f32 B against f32 A.

    python tools/parity/bf16_vs_f32_training.py [--iters 500] [--batch 32] [--out profiles/r05_bf16_vs_f32_training.txt]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
DEV = "cuda:0"


def run(dt, iters, batch, size, pool, marks, init=None, seed=77):
    import layout2img_amd as L
    torch.manual_seed(1234)
    g = L.ResnetGenerator128_context(num_classes=184) if size == 128 else L.ResnetGenerator64_context(num_classes=184)
    d = L.CombineDiscriminator128_app(num_classes=184) if size == 128 else L.CombineDiscriminator64(num_classes=184)
    if init is not None:
        g.load_state_dict(init[0]), d.load_state_dict(init[1])
    g.finalize(DEV, dt), d.finalize(DEV, dt)
    for m in g.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
    g.train(), d.train()
    tr = L.GanTrainer(g, d)
    start = (g.flat.data.clone(), d.flat.data.clone())
    torch.manual_seed(seed)   # the latent draws of the iterations
    losses, snaps = [], {}
    for it in range(1, iters + 1):
        real, label, bbox = pool[(it - 1) % len(pool)]
        r = tr.step(real, label, bbox, None, None)
        losses.append(torch.stack([r["d_loss"], r["g_loss"], r["pixel"]]))
        if it in marks:
            tr.flush()
            snaps[it] = (g.flat.data.clone(), d.flat.data.clone())
    tr.flush()
    torch.cuda.synchronize()
    return dict(losses=torch.stack(losses).cpu(), snaps=snaps, start=start,
                state=({k: v.detach().cpu().clone() for k, v in g.state_dict().items()}, {k: v.detach().cpu().clone() for k, v in d.state_dict().items()}))


def compare(iters=500, batch=32, size=128, npool=16):
    from layout2img_amd.synthetic import make_batch
    pool = [make_batch(batch, size, "coco", seed=500 + i, device=DEV)[:3] for i in range(npool)]
    marks = sorted({m for m in (1, 2, 5, 10, 20, 50, 100, 200, 300, 400, 500, 750, 1000, 2000) if m <= iters} | {iters})
    a = run(torch.float32, iters, batch, size, pool, marks)
    b = run(torch.float32, iters, batch, size, pool, marks)
    c = run(torch.bfloat16, iters, batch, size, pool, marks)
    assert torch.equal(a["start"][0], c["start"][0]) and torch.equal(a["start"][1], b["start"][1])
    rows = []
    prev = 0
    for t in marks:
        row = dict(it=t)
        for name, r in (("f32A", a), ("f32B", b), ("bf16", c)):
            row[name] = [float(v) for v in r["losses"][t - 1]]
            row[name + "_mean"] = [float(v) for v in r["losses"][prev:t].mean(0)]
        for k, net in ((0, "G"), (1, "D")):
            moved = float((a["snaps"][t][k] - a["start"][k]).norm())
            row[f"{net}_moved"] = moved
            row[f"{net}_floor"] = float((b["snaps"][t][k] - a["snaps"][t][k]).norm()) / moved
            row[f"{net}_bf16"] = float((c["snaps"][t][k] - a["snaps"][t][k]).norm()) / moved
        rows.append(row)
        prev = t
    return rows


def fmt(rows, iters, batch, size):
    out = [f"bf16-operand training against exact-f32 training: {iters} iterations, {size}x{size}, batch {batch}, COCO layouts, pool of 16 synthetic batches,",
           "same initial state, same latent draws, Dropout2d off (tools/parity/bf16_vs_f32_training.py). f32 A / f32 B: two exact-f32 runs (the floor).",
           "",
           "losses AT iteration t (d_loss, g_loss, pixel) and their MEAN over the window since the previous row:",
           f"{'t':>5}  {'f32 A':^26}  {'f32 B':^26}  {'bf16':^26}"]
    f3 = lambda v: " ".join(f"{x:8.4f}" for x in v)
    for r in rows:
        out.append(f"{r['it']:5d}  {f3(r['f32A'])}  {f3(r['f32B'])}  {f3(r['bf16'])}")
    out.append("window means:")
    for r in rows:
        out.append(f"{r['it']:5d}  {f3(r['f32A_mean'])}  {f3(r['f32B_mean'])}  {f3(r['bf16_mean'])}")
    out += ["", "parameter distance to run f32 A, in units of the distance f32 A has moved from the start (||theta_run - theta_A|| / ||theta_A - theta_0||):",
            f"{'t':>5}  {'G moved':>10} {'G: f32 B (floor)':>18} {'G: bf16':>10}   {'D moved':>10} {'D: f32 B (floor)':>18} {'D: bf16':>10}"]
    for r in rows:
        out.append(f"{r['it']:5d}  {r['G_moved']:10.4f} {r['G_floor']:18.4f} {r['G_bf16']:10.4f}   {r['D_moved']:10.4f} {r['D_floor']:18.4f} {r['D_bf16']:10.4f}")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rows = compare(args.iters, args.batch, args.size)
    text = fmt(rows, args.iters, args.batch, args.size)
    print(text)
    if args.out:
        open(args.out, "w").write(text)
