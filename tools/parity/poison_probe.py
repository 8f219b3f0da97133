"""Debug aid: does any kernel of the generator / discriminator forward READ memory it (or its producer) never wrote?
Every torch.empty on the GPU is filled with a huge finite value (a zero weight times it is still 0, so legitimately untouched
padding stays harmless); the taps of a poisoned eager forward are compared with those of a clean one."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import layout2img_amd as L
from layout2img_amd.synthetic import make_batch
DEV = "cuda:0"
POISON = float(sys.argv[2]) if len(sys.argv) > 2 else 3e30
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(3)
g = L.ResnetGenerator128_context(num_classes=184).finalize(DEV, torch.bfloat16)
d = L.CombineDiscriminator128_app(num_classes=184).finalize(DEV, torch.bfloat16)
real, label, bbox, z, z_im = make_batch(max(B, 4), 128, "coco", seed=9, device=DEV)
g.train()
with torch.no_grad():
    for _ in range(3):
        g(z, bbox, z_im, label)
real, label, bbox, z, z_im = real[:B], label[:B], bbox[:B], z[:B], z_im[:B]
orig_empty, orig_empty_like = torch.empty, torch.empty_like
state = {"on": False}
def empty(*a, **k):
    t = orig_empty(*a, **k)
    if state["on"] and t.is_cuda and t.is_floating_point():
        t.fill_(POISON)
    return t
def empty_like(*a, **k):
    t = orig_empty_like(*a, **k)
    if state["on"] and t.is_cuda and t.is_floating_point():
        t.fill_(POISON)
    return t
torch.empty, torch.empty_like = empty, empty_like

def flat_taps(t):
    return [("w", t["w"]), ("bmask", t["bmask"])] + [(f"stage{i}", s) for i, s in enumerate(t["stages"])] + [(f"res{i}", r) for i, r in enumerate(t["res"])] + [("pre", t["pre_tanh"])]

for mode in ("eval", "train"):
    g.train(mode == "train")
    outs = []
    for poison in (False, True, True):
        state["on"] = poison
        torch.manual_seed(11)
        t = {}
        with torch.no_grad():
            img = g(z, bbox, z_im=z_im, y=label, taps=t)
        torch.cuda.synchronize()
        outs.append((img.clone(), [(n, x.clone()) for n, x in flat_taps(t)]))
        state["on"] = False
    for k in (1, 2):
        msg = [f"{n}:{float((a.float() - b.float()).abs().max()):.1e}" for (n, a), (_, b) in zip(outs[k][1], outs[0][1])]
        print(f"G {mode} b={B} poisoned#{k} vs clean: img {float((outs[k][0] - outs[0][0]).abs().max()):.2e}", " ".join(msg), flush=True)
d.train()
res = []
for poison in (False, True):
    state["on"] = poison
    with torch.no_grad():
        o = d(real, bbox, label)
    torch.cuda.synchronize()
    res.append([x.clone() for x in o])
    state["on"] = False
print(f"D train b={B} poisoned vs clean:", [f"{float((a - b).abs().max()):.2e}" for a, b in zip(res[1], res[0])], flush=True)
