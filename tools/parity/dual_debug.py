"""Debug aid: forward_dual vs two forward_padded passes on the 64x64 discriminator (f32), with the dual conv launches replaced by
the two-launch fallback to localise a discrepancy."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import layout2img_amd as L
from layout2img_amd import ops
from layout2img_amd.synthetic import make_batch
DEV = "cuda:0"
size, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 4
dt = torch.bfloat16 if os.environ.get("BF16") else torch.float32
torch.manual_seed(5)
net = (L.CombineDiscriminator128_app if size == 128 else L.CombineDiscriminator64)(num_classes=184).finalize(DEV, dt)
net.train()
real, label, bbox, _, _ = make_batch(b, size, "coco", seed=21, device=DEV)
fake = torch.randn_like(real).clamp_(-1, 1)
sn0 = net.arena.sn_flat.data.clone()
n_out = 3 if size == 128 else 2
wts = [torch.randn(1, device=DEV) for _ in range(2 * n_out)]
def run(dual):
    net.arena.sn_flat.data.copy_(sn0); net.arena.drop_pending(); net.zero_grad()
    ra, fa = real.clone().requires_grad_(True), fake.clone().requires_grad_(True)
    if dual:
        oa, ob, valid, _ = net.forward_dual(ra, fa, bbox, label)
    else:
        *oa, valid, _ = net.forward_padded(ra, bbox, label)
        *ob, _, _ = net.forward_padded(fa, bbox, label)
    vm = valid.float().view(-1, 1)
    loss = 0
    for k, o in enumerate(list(oa) + list(ob)):
        loss = loss + wts[k] * ((o * vm).sum() if o.shape[0] == vm.shape[0] else o.sum())
    loss.backward(); net.arena.flush_grads(); torch.cuda.synchronize()
    return [t.detach().clone() for t in list(oa) + list(ob)], net.flat.grad.clone(), ra.grad.clone(), fa.grad.clone()
def rel(a, b): return float((a - b).norm() / b.norm())
ref = run(False)
def report(tag):
    r = run(True)
    if os.environ.get("TOP"):
        errs = []
        for n, p_ in net.named_parameters():
            o = net.flat.offset_of(p_); k = p_.numel()
            a, b_ = r[1][o:o + k], ref[1][o:o + k]
            if float(b_.norm()) > 0: errs.append((rel(a, b_), n, float(b_.norm())))
        print("   worst parameters:", [(f"{e:.1e}", n) for e, n, _ in sorted(errs, reverse=True)[:6]])
        d = (r[3] - ref[3]).abs().amax(dim=(1, 2, 3)); print("   fake-img max abs err per image:", [f"{float(v):.1e}" for v in d], "scale %.1e" % float(ref[3].abs().max()))
    print(tag, "outs", [f"{rel(x, y):.1e}" for x, y in zip(r[0], ref[0])], "params %.1e real-img %.1e fake-img %.1e" % (rel(r[1], ref[1]), rel(r[2], ref[2]), rel(r[3], ref[3])))
report("dual launches      ")
ok = ops.dual_conv_ok
ops.dual_conv_ok = lambda *a: False
report("conv fallback      ")
ops.dual_conv_ok = ok
orig = ops.wgrad_raw
def wg(x_op, dy_op, dw, ldw, co, kh, **kw):
    dwb = kw.pop("dw_b", None)
    if dwb is None: return orig(x_op, dy_op, dw, ldw, co, kh, **kw)
    hb = x_op.shape[0] // 2
    sc = kw.pop("sc", None)
    for k, d in enumerate((dw, dwb)):
        sl = slice(k * hb, (k + 1) * hb)
        sck = None if sc is None else dict(sc, x_op=sc["x_op"][sl], dw=sc["dw_b"] if k else sc["dw"], dw_b=None)
        orig(x_op[sl], dy_op[sl], d, ldw, co, kh, sc=sck, **kw)
ops.wgrad_raw = wg
report("wgrad fallback     ")
