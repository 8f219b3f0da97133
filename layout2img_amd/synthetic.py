"""Synthetic layouts with the statistics of the reference data loaders.

COCO-layout (data/cocostuff_loader.py:19-20,123-125,301-303): o = 8 slots, 3..8 real objects, box
w,h ~ U(0.15, 0.9) with w*h > 0.02, labels 1..183, padding slots label 0 and box
[-0.6,-0.6,0.5,0.5]. VG-layout (data/vg.py:118-141): o = 31, 3..30 real objects, then one
`__image__` slot (label 0, box [0,0,1,1]), remaining pads as above.
"""
import torch

PAD_BOX = (-0.6, -0.6, 0.5, 0.5)


def make_layouts(batch, kind="coco", seed=0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    o, ncls, lo, hi = (8, 184, 3, 8) if kind == "coco" else (31, 179, 3, 30)
    label = torch.zeros(batch, o, dtype=torch.long)
    bbox = torch.tensor(PAD_BOX).repeat(batch, o, 1)
    for b in range(batch):
        n = int(torch.randint(lo, hi + 1, (1,), generator=g))
        for i in range(n):
            while True:
                w, h = (0.15 + 0.75 * torch.rand(2, generator=g)).tolist()
                if w * h > 0.02:
                    break
            x = float(torch.rand(1, generator=g)) * (1 - w)
            y = float(torch.rand(1, generator=g)) * (1 - h)
            bbox[b, i] = torch.tensor([x, y, w, h])
            label[b, i] = int(torch.randint(1, ncls, (1,), generator=g))
        if kind != "coco":
            bbox[b, n] = torch.tensor([0.0, 0.0, 1.0, 1.0])
    return label.to(device), bbox.to(device)


def make_batch(batch, size=128, kind="coco", seed=0, device="cpu", z_dim=128):
    """real images ~ U(-1,1), labels, boxes, z ~ N(0,1) (b,o,128), z_im ~ N(0,1) (b,128)."""
    label, bbox = make_layouts(batch, kind, seed, "cpu")
    g = torch.Generator().manual_seed(seed + 7919)
    real = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    z = torch.randn(batch, label.shape[1], z_dim, generator=g)
    z_im = torch.randn(batch, z_dim, generator=g)
    return real.to(device), label.to(device), bbox.to(device), z.to(device), z_im.to(device)
