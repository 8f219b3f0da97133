import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ["GRAFT_REPO_ROOT"])
dev = torch.device("cuda:0")
x = torch.randn(4, device=dev)
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=side):
        y = torch.randn(4, device=dev)
        torch.cuda.synchronize()   # illegal during capture -> invalidates it
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:80])
torch.cuda.synchronize()
for attempt in ("plain", "manual_seed", "new graph"):
    try:
        if attempt == "manual_seed":
            torch.cuda.manual_seed(1234)
        if attempt == "new graph":
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=side):
                z = torch.randn(4, device=dev)
            g2.replay()
        print(attempt, "-> randn ok", torch.randn(2, device=dev).tolist())
    except Exception as e:
        print(attempt, "-> FAILED", type(e).__name__, str(e)[:100])
