"""Does training with bf16 MFMA operands track training with exact-f32 operands?  (reference loop: train_context_app_v2.py:148-189)

Several runs of N iterations of GanTrainer.step from the SAME initial state, the SAME pool of device-resident synthetic batches and
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


def compare(iters=500, batch=32, size=128, npool=16, n_f32=2, n_bf16=2):
    """n_f32 exact-f32 runs (the first is the reference trajectory, the others the floor) and n_bf16 bf16-operand runs."""
    from layout2img_amd.synthetic import make_batch
    pool = [make_batch(batch, size, "coco", seed=500 + i, device=DEV)[:3] for i in range(npool)]
    marks = sorted({m for m in (1, 2, 5, 10, 20, 50, 100, 200, 300, 400, 500, 750, 1000, 2000) if m <= iters} | {iters})
    runs = [(f"f32 {chr(65 + k)}", run(torch.float32, iters, batch, size, pool, marks)) for k in range(n_f32)]
    runs += [(f"bf16 {chr(65 + k)}", run(torch.bfloat16, iters, batch, size, pool, marks)) for k in range(n_bf16)]
    a = runs[0][1]
    for _, r in runs[1:]:
        assert torch.equal(a["start"][0], r["start"][0]) and torch.equal(a["start"][1], r["start"][1])
    rows = []
    prev = 0
    for t in marks:
        row = dict(it=t, runs={})
        for name, r in runs:
            e = dict(at=[float(v) for v in r["losses"][t - 1]], mean=[float(v) for v in r["losses"][prev:t].mean(0)])
            for k, net in ((0, "G"), (1, "D")):
                moved = float((a["snaps"][t][k] - a["start"][k]).norm())
                e[net] = float((r["snaps"][t][k] - a["snaps"][t][k]).norm()) / moved
                row[f"{net}_moved"] = moved
            row["runs"][name] = e
        rows.append(row)
        prev = t
    return rows


def fmt(rows, iters, batch, size):
    names = list(rows[0]["runs"])
    out = [f"bf16-operand training against exact-f32 training: {iters} iterations, {size}x{size}, batch {batch}, COCO layouts, pool of 16 synthetic batches,",
           "same initial state, same latent draws, Dropout2d off (tools/parity/bf16_vs_f32_training.py). Runs: " + ", ".join(names) +
           " -- every f32 run after the first is a measurement of the FLOOR (two exact-f32 runs differ by accumulation order only).",
           "",
           "MEAN of (d_loss, g_loss, pixel) over the window since the previous row:",
           f"{'t':>5}  " + "  ".join(f"{n:^26}" for n in names)]
    f3 = lambda v: " ".join(f"{x:8.4f}" for x in v)
    for r in rows:
        out.append(f"{r['it']:5d}  " + "  ".join(f3(r["runs"][n]["mean"]) for n in names))
    out += ["", "losses AT iteration t:", f"{'t':>5}  " + "  ".join(f"{n:^26}" for n in names)]
    for r in rows:
        out.append(f"{r['it']:5d}  " + "  ".join(f3(r["runs"][n]["at"]) for n in names))
    out += ["", f"parameter distance to run {names[0]}, in units of the distance {names[0]} has moved from the start (||theta_run - theta_A|| / ||theta_A - theta_0||):",
            f"{'t':>5}  {'G moved':>9} " + " ".join(f"{'G: ' + n:>10}" for n in names[1:]) + f"   {'D moved':>9} " + " ".join(f"{'D: ' + n:>10}" for n in names[1:])]
    for r in rows:
        out.append(f"{r['it']:5d}  {r['G_moved']:9.4f} " + " ".join(f"{r['runs'][n]['G']:10.4f}" for n in names[1:]) +
                   f"   {r['D_moved']:9.4f} " + " ".join(f"{r['runs'][n]['D']:10.4f}" for n in names[1:]))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--out", default=None)
    ap.add_argument("--n-f32", type=int, default=2)
    ap.add_argument("--n-bf16", type=int, default=2)
    args = ap.parse_args()
    rows = compare(args.iters, args.batch, args.size, n_f32=args.n_f32, n_bf16=args.n_bf16)
    text = fmt(rows, args.iters, args.batch, args.size)
    print(text)
    if args.out:
        open(args.out, "w").write(text)
