"""Where does the bf16-operand error of the generator come from? Per-tap error against the reference golden
(tests/golden/g_coco.npz), f32 vs bf16 operands, train-mode forward (tuning / DESIGN.md table)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import fixture_inputs, fixture_state, load_fixture
import layout2img_amd as L

fx = load_fixture("g_coco.npz")
DEV = "cuda:0"
for dt in (torch.float32, torch.bfloat16):
    torch.manual_seed(0)
    g = L.ResnetGenerator128_context(num_classes=184, output_dim=3)
    g.load_state_dict(fixture_state(fx, 11))
    g.finalize(DEV, dt)
    for m in g.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
    g.train()
    inp = {k: v.to(DEV) for k, v in fixture_inputs(fx).items()}
    taps = {}
    with torch.no_grad():
        out = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"], taps=taps)

    def rel(a, b):
        a = a.detach().float().cpu().numpy()
        return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12)), float(np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-12))
    rows = [("w (context attention)", taps["w"], fx["tap_w"]), ("bmask (mask regression)", taps["bmask"], fx["tap_bmask"]),
            ("res1 out", taps["res"][0].permute(0, 3, 1, 2), fx["tap_res1"]),
            ("stage mask 2", taps["stages"][0], fx["tap_stage_in2"]),
            ("res3 out channel means", taps["res"][2].mean(dim=(0, 1, 2)), fx["tap_res3_mean"]),
            ("stage mask 5", taps["stages"][3][:, :, ::4, ::4], fx["tap_stage_in5"]),
            ("pre-tanh", taps["pre_tanh"].permute(0, 3, 1, 2), fx["tap_pre_tanh"]), ("image", out, fx["out_train1"])]
    print("operands", dt)
    for name, a, b in rows:
        mx, rms = rel(a, b)
        print(f"   {name:28s} max-rel {mx:9.2e}   rms-rel {rms:9.2e}")
    print("   image Linf", float(np.abs(out.cpu().numpy() - fx["out_train1"]).max()))
