"""Per-shape time of the ISLA / batch-norm kernels at the generator's ten norm sites (batch 32, 8 objects), from a kernel trace:
    rocprofv3 --kernel-trace -d /tmp/ns -o t --output-format csv -- python tools/perf/norm_shapes.py run
    python tools/perf/norm_shapes.py join $(find /tmp/ns -name '*kernel_trace.csv')
Every shape issues each kernel exactly N times, in order, so the k-th run of N launches of a kernel name belongs to shape k.
Bandwidth = algorithmic bytes (x, dy read, dxhat / y written) / time."""
import csv, collections, os, sys
N = 10
SHAPES = [(32, 4, 1024), (32, 8, 1024), (32, 8, 1024), (32, 16, 512), (32, 16, 512), (32, 32, 256), (32, 32, 256), (32, 64, 128), (32, 64, 128), (32, 128, 64)]


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from layout2img_amd import ops
    dev = torch.device("cuda:0")
    O = 8
    for (B, H, C) in SHAPES:
        x = torch.randn(B, H, H, C, device=dev)
        dy = torch.randn(B, H, H, C, device=dev)
        spec = ops.NormSpec(0, relu=True)
        mask = torch.rand(B, O, H, H, device=dev)
        wp = torch.randn(B, O, C, device=dev) * 0.1
        bp = torch.randn(B, O, C, device=dev) * 0.1
        sums, sq = ops.channel_stats(x.view(-1, C))
        cnt = float(B * H * H)
        for _ in range(N - 1):
            ops.channel_stats(x.view(-1, C))
        for _ in range(N):
            ops.norm_fwd_raw(x, sums, sq, cnt, 0, spec, mask, wp, bp, torch.bfloat16)
        for _ in range(N):
            ops.norm_bwd_raw(x, dy, sums, sq, cnt, 0, spec, mask, wp, bp)
        torch.cuda.synchronize()


def join(path, only=None):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seq = collections.defaultdict(list)
    for r in rows:
        nm = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if any(k in nm for k in ("norm_", "channel_stats", "ws_fold")):
            seq[nm.split("<")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), nm, r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
    traffic = {"channel_stats_kernel": 1, "norm_mod8_kernel": 1.5, "norm_bwd_a8_kernel": 3, "norm_bwd_b_kernel": 3}
    for base, lst in seq.items():
        if only and only not in base:
            continue
        print(f"== {base}: {len(lst)} launches")
        for k, (B, H, C) in enumerate(SHAPES):
            grp = lst[k * N:(k + 1) * N]
            if not grp:
                break
            us = sorted(d for d, _, _ in grp)[len(grp) // 2] / 1e3
            mb = B * H * H * C * 4 / 1e6 * traffic.get(base, 1)
            print(f"   b{B} {H:3d}x{H:<3d} C{C:<5d} grid {grp[0][2]:>8}  median {us:7.1f} us   {mb:7.1f} MB  {mb / us:5.2f} TB/s   {grp[0][1][:60]}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else join(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
