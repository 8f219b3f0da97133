import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layout2img_amd import ops
dev = torch.device('cuda:0')
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (B, H, C, O) in [(32, 128, 64, 8), (32, 64, 128, 8), (32, 64, 64, 8), (32, 32, 256, 8), (32, 16, 512, 8), (32, 8, 1024, 8), (32,128,64,0)]:
    x = torch.randn(B, H, H, C, device=dev)
    dy = torch.randn(B, H, H, C, device=dev)
    mode = 0 if O else 2
    spec = ops.NormSpec(mode, relu=True)
    mask = torch.rand(B, O, H, H, device=dev) if O else None
    wp = torch.randn(B, O, C, device=dev) * 0.1 if O else None
    bp = torch.randn(B, O, C, device=dev) * 0.1 if O else None
    sums, sq = ops.channel_stats(x.view(-1, C))
    cnt = float(B * H * H)
    st = t(lambda: ops.channel_stats(x.view(-1, C)))
    stc = t(lambda: ops.channel_stats(x.view(-1, C), cast_to=torch.bfloat16))
    fw = t(lambda: ops.norm_fwd_raw(x, sums, sq, cnt, 0, spec, mask, wp, bp, torch.bfloat16))
    bw = t(lambda: ops.norm_bwd_raw(x, dy.clone(), sums, sq, cnt, 0, spec, mask, wp, bp))
    bwnm = t(lambda: ops.norm_bwd_raw(x, dy.clone(), sums, sq, cnt, 0, spec, mask, wp, bp, need_mask_grad=False))
    cl = t(lambda: dy.clone())
    mb = x.numel() * 4 / 1e6
    print(f"B{B} H{H} C{C} O{O}: {mb:.0f} MB  stats {st:.0f}us ({mb/st*1e-3:.2f} TB/s)  stats+cast {stc:.0f}us  fwd {fw:.0f}us ({mb*1.5/fw*1e-3:.2f} TB/s) bwd(a+b+clone) {bw:.0f}us  nomask {bwnm:.0f}us clone {cl:.0f}us")
