import sys, torch, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L
from layout2img_amd import ops, _lib
from layout2img_amd.synthetic import make_batch
dev = torch.device('cuda:0')
torch.manual_seed(1234)
netG = L.ResnetGenerator128_context(num_classes=184).finalize(dev, torch.bfloat16)
netD = L.CombineDiscriminator128_app(num_classes=184).finalize(dev, torch.bfloat16)
tr = L.GanTrainer(netG, netD)
real, label, bbox, z, z_im = make_batch(32, 128, "coco", seed=1234, device=dev)
for _ in range(2): tr.step(real, label, bbox, z, None)
recs = []
orig_call = _lib.call
def call(name, *a):
    if name in ("l2i_conv2d_fwd", "l2i_conv2d_wgrad"):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); orig_call(name, *a); e.record()
        if name == "l2i_conv2d_fwd":
            key = ("fwd",) + tuple(a[9:20])   # B,Hi,Wi,Ci,Ho,Wo,Co,KH,up2,pool2,relu
        else:
            key = ("wgr",) + tuple(a[4:14])
        recs.append((key, s, e))
    else:
        orig_call(name, *a)
_lib.call = call; ops._lib.call = call
for _ in range(2): tr.step(real, label, bbox, z, None)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, s, e in recs:
    ms = s.elapsed_time(e)
    d = agg.setdefault(key, [0, 0.0]); d[0] += 1; d[1] += ms
rows = []
for key, (n, ms) in agg.items():
    if key[0] == "fwd":
        _, B, Hi, Wi, Ci, Ho, Wo, Co, KH, up2, pool2, relu = key
    else:
        _, B, Hi, Wi, Ci, Ho, Wo, Co, KH, up2, pool2 = key
    fl = 2.0 * B * Ho * Wo * Co * Ci * KH * KH
    rows.append((ms / 2, n // 2, key, fl * n / (ms * 1e-3) / 1e12))
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print("total conv ms/step", tot)
for ms, n, key, tf in rows[:120]:
    print(f"{ms:7.3f} ms/step x{n:3d}  {tf:7.1f} TF/s  {key}")
