"""Would promoting a few layers to exact arithmetic (split-bf16: hi + lo operands, 3 MFMAs = f32-accurate products)
bring the bf16-operand generator image inside the north-star bar? CPU experiment on the oracle (tests-side code, run
here, results quoted in DESIGN.md section 2): every F.conv2d / F.linear of the generator forward gets its input and
weight rounded to bf16 (the MFMA path: bf16 operands, exact products, f32 accumulation) EXCEPT the layers named in
`keep`; the image is compared with the all-f32 oracle on the reference golden inputs (train-mode forward, b = 2)."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model as O
from tests.helpers import fixture_inputs, fixture_state, load_fixture

fx = load_fixture("g_coco.npz")
sd = fixture_state(fx, 11)
inp = fixture_inputs(fx)
torch.set_num_threads(8)


def run(keep=(), mode="bf16"):
    """keep: set of weight-tensor shapes / call indices whose operands stay f32.
    mode: how the other layers' MFMA operands are rounded -- "bf16" (the throughput mode), "fp16" (v_mfma_f32_32x32x16_f16: same
    rate on gfx950, 11 instead of 8 significand bits), "bf16x2" (split operands hi + lo, both bf16: x = hi + lo to 16 bits, three
    MFMA passes hi*hi + hi*lo + lo*hi), "fp16_act" (fp16 activations, bf16 weights: is it the weights or the activations?)."""
    calls = []
    real_conv, real_lin = F.conv2d, F.linear

    def r16(t, dt):
        return t.to(dt).to(torch.float32)

    def rb(t, is_w=False):
        if mode == "fp16" or (mode == "fp16_act" and not is_w):
            return r16(t, torch.float16)
        if mode == "bf16x2":
            hi = r16(t, torch.bfloat16)
            return hi + r16(t - hi, torch.bfloat16)
        return r16(t, torch.bfloat16)

    def conv2d(x, w, *a, **k):
        i = len(calls); calls.append(("conv", tuple(w.shape)))
        if i in keep or "all" in keep:
            return real_conv(x, w, *a, **k)
        return real_conv(rb(x), rb(w, True), *a, **k)

    def linear(x, w, *a, **k):
        i = len(calls); calls.append(("lin", tuple(w.shape)))
        if i in keep or "all" in keep:
            return real_lin(x, w, *a, **k)
        return real_lin(rb(x), rb(w, True), *a, **k)
    O.F.conv2d, O.F.linear = conv2d, linear
    try:
        sdc = {k: v.clone() for k, v in sd.items()}
        with torch.no_grad():
            img = O.generator_forward(sdc, inp["z"], inp["bbox"], inp["z_im"], inp["y"], training=True, dropout_p=0.0)
    finally:
        O.F.conv2d, O.F.linear = real_conv, real_lin
    return img, calls


ref, calls = run(keep={"all"})
n = len(calls)
print(f"{n} conv / linear calls in one generator forward; last ten:", calls[-10:])
base, _ = run()
print(f"all layers bf16 operands:                         image L_inf {float((base - ref).abs().max()):.4f}")
# the to-RGB conv is the last call; res5.conv2 / res5.c_sc / res5.conv1 precede it (+ the ISLA projections in between)
last = [i for i, c in enumerate(calls) if c[0] == "conv"][-1]
convs = [i for i, c in enumerate(calls) if c[0] == "conv"]
fc = [i for i, c in enumerate(calls) if c == ("lin", (16384, 128))]
for name, keep in (("to-RGB conv exact", {last}), ("to-RGB + last 3 convs (res5) exact", set(convs[-4:])),
                   ("to-RGB + res5 + fc exact", set(convs[-4:]) | set(fc)), ("last 8 convs exact", set(convs[-8:])),
                   ("every conv of res4, res5, to-RGB exact (last 12 convs)", set(convs[-12:])),
                   ("every SECOND layer exact", set(range(0, n, 2)))):
    img, _ = run(keep)
    print(f"{name:55s} image L_inf {float((img - ref).abs().max()):.4f}")

# round 4: other operand formats at the same (fp16) or three times (split bf16) the MFMA cost -- forward / sampling only
# (gradients are not range-safe in fp16)
print()
for mode in ("bf16", "fp16", "fp16_act", "bf16x2"):
    img, _ = run((), mode)
    d = (img - ref).abs()
    print(f"all layers, operands {mode:9s}: image L_inf {float(d.max()):.5f}  rms {float(d.pow(2).mean().sqrt()):.6f}   max |operand| seen: n/a")
