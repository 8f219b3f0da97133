"""Debug aid: is a captured launch idempotent under replay?  conv_raw cases (split-K by atomics = hipMemsetAsync + atomics; stored
partial tiles; the 4x4 weight-stationary kernel; a plain launch) captured alone, replayed three times, each replay compared with
the eager result."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from layout2img_amd import ops, _lib
DEV = "cuda:0"
bf = torch.bfloat16

def pack(w, mult=64):
    co, ci, kh, _ = w.shape
    k = kh * kh * ci
    kpad, npad = (k + mult - 1) // mult * mult, (co + 127) // 128 * 128
    p = torch.zeros(npad, kpad)
    p[:co, :k] = w.permute(0, 2, 3, 1).reshape(co, k)
    return p, kpad

PRE = os.environ.get("PRE", "0") == "1"     # a torch kernel in front of the launch, so that the memset is not the graph's root node
cases = {"b32 4->8 up 1024->1024": (32, 4, 4, 1024, 1024, 3, True), "b32 8x8 1024->512": (32, 8, 8, 1024, 512, 3, False), "linear splitK atomics": (8, 1, 1, 2048, 16, 1, False), "3x3 8x8 256->64": (2, 8, 8, 256, 64, 3, False), "4x4 wstat 512->128": (8, 4, 4, 512, 128, 3, False),
         "3x3 4->8 up 256->128": (3, 4, 4, 256, 128, 3, True), "plain 16x16 64->72": (2, 16, 16, 64, 72, 3, False), "1x1 32x32 64->128": (2, 32, 32, 64, 128, 1, False)}
side = torch.cuda.Stream()
for name, (B, H, W, Ci, Co, KH, up2) in cases.items():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, W, Ci, generator=g).to(DEV, bf)
    w = torch.randn(Co, Ci, KH, KH, generator=g) / math.sqrt(Ci * KH * KH)
    p, kpad = pack(w)
    p = p.to(DEV, bf)
    bias = torch.randn(Co, generator=g).to(DEV)
    run = lambda: ops.conv_raw(x, p, kpad, Co, KH, bias=bias, up2=up2)[0]
    ref = run().clone()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _lib.workspace(DEV)
        for _ in range(2):
            run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    dummy = torch.zeros(1024, device=DEV)
    with torch.cuda.graph(graph, stream=side):
        if PRE:
            dummy.add_(1.0)
        out = run()
        if PRE:
            dummy.add_(1.0)
    errs = []
    for r in range(3):
        graph.replay()
        torch.cuda.synchronize()
        errs.append(float((out - ref).abs().max()) / float(ref.abs().max()))
    print(f"{name:28s} splits {_lib.load().l2i_debug_occupancy(100, 0):3d}  replay errors vs eager: " + " ".join(f"{e:.2e}" for e in errs), flush=True)
