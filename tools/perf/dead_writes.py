"""Result streams a launch writes and nothing ever reads, from a dry run of one b = 32 iteration (tests/dryrun.py: fake addresses, stock torch
operators recorded): a buffer that a C-ABI call names through a non-const pointer and that no later call or operator of the iteration names
again. No GPU. `sc_out` (the lazily folded shortcut's placeholder: written only when the library cannot fold) is left out; a non-const
parameter may be read-modify-write, so a buffer is reported only when its LAST naming is a write.
usage: python tools/perf/dead_writes.py [coco|vg] [batch] > profiles/rNN_dead_writes.txt"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import dryrun
from layout2img_amd.synthetic import make_batch

kind = sys.argv[1] if len(sys.argv) > 1 else "coco"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
kinds, names = dryrun.header_pointer_kinds(), dryrun.header_parameters()
ESZ = {"torch.float32": 4, "torch.bfloat16": 2, "torch.int32": 4, "torch.int64": 8, "torch.float16": 2, "torch.uint8": 1, "torch.bool": 1}


def tensors(x, out):
    if isinstance(x, tuple):
        if len(x) == 5 and x[0] == "T":
            out.append((x[1] >> 40, x[2], x[4]))
        else:
            for y in x:
                tensors(y, out)
    return out


with dryrun.dry_run(pointers=True, aten=True) as trace:
    tr, _ = dryrun.build(kind, torch.bfloat16)
    real, label, bbox, z, z_im = (t.to("meta") for t in make_batch(batch, 128, kind, seed=1234, device="cpu"))
    for _ in range(3):
        del trace[:]
        tr.step(real, label, bbox, z, z_im if kind == "vg" else None)
size, last, dtype_of = {}, {}, {}
for pos, (n, a) in enumerate(trace):
    if n.startswith("l2i_"):
        for k, pn, v in zip(kinds[n], names[n], a):
            if k and type(v) is int and v >= (1 << 40):
                last[v >> 40] = (pos, n, pn, k)
    else:
        for b, shape, dt in tensors(a[2], []):
            nb = ESZ.get(dt, 4)
            for d in shape:
                nb *= d
            size[b] = max(size.get(b, 0), nb)
            dtype_of[b] = dt
        if n not in dryrun._VIEWS and n not in dryrun._ALLOC:
            for b, _, _ in tensors(a[0], []):
                last[b] = (pos, n, "operand", "in")
dead = sorted(((size[b], b) + last[b] for b in last if last[b][3] == "out" and last[b][2] != "sc_out" and b in size and size[b] >= (1 << 20)), reverse=True)
by = collections.defaultdict(lambda: [0, 0])
for s, b, pos, n, pn, _ in dead:
    by[(n, pn, dtype_of[b])][0] += 1
    by[(n, pn, dtype_of[b])][1] += s
print(f"# {kind} layouts, batch {batch}, bf16 operands: result streams (>= 1 MB) written by a launch and never named again in the iteration")
print(f"# total {sum(d[0] for d in dead) / 1e6:.0f} MB per iteration in {len(dead)} buffers; by writer:")
for (n, pn, dt), (k, s) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"{s / 1e6:9.1f} MB  x{k:3d}  {n}.{pn}  ({dt})")
print("# buffer by buffer (MB, call #, writer):")
for s, b, pos, n, pn, _ in dead:
    print(f"{s / 1e6:9.2f}  #{pos:<5d} {n}.{pn}")
