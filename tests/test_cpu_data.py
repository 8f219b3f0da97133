"""CPU: layout2img_amd.data against the outputs of the reference's own dataset classes (data/cocostuff_loader.py,
data/vg.py) on the miniature datasets of tests/golden/tiny_datasets.py (captured by tools/capture_goldens.py data)."""
import os
import random

import numpy as np
import torch

from layout2img_amd import data as D
from tests.golden import tiny_datasets
from tests.helpers import load_fixture


def test_coco_layouts_match_the_reference_loader(tmp_path):
    fx = load_fixture("datasets.npz")
    root = tiny_datasets.write(str(tmp_path))
    ds = D.CocoLayoutDataset(os.path.join(root, "images"), os.path.join(root, "instances.json"), os.path.join(root, "stuff.json"),
                             stuff_only=True, image_size=(32, 32), left_right_flip=True)
    assert len(ds) == int(fx["coco_len"]) == 6 and ds.image_ids == fx["coco_ids"].tolist() == [100, 102, 105]
    assert ds.vocab["object_name_to_idx"]["__image__"] == 0
    for i in range(len(ds)):
        im, objs, boxes = ds[i]
        assert im.dtype == torch.float32 and im.shape == (3, 32, 32) and objs.dtype == torch.long and boxes.shape == (8, 4)
        assert np.abs(im.numpy() - fx[f"coco_img{i}"]).max() == 0.0
        assert objs.tolist() == fx[f"coco_objs{i}"].tolist()
        assert np.abs(boxes.numpy().astype(np.float64) - fx[f"coco_boxes{i}"]).max() < 1e-6
    im, objs, boxes = ds[0]
    n = int((objs != 0).sum())
    assert 3 <= n <= 8 and boxes[n:].tolist() == [list(map(lambda v: float(np.float32(v)), D.PAD_BOX))] * (8 - n)
    # mirrored half: x0 -> 1 - (x0 + w), same labels
    _, objs_f, boxes_f = ds[3]
    assert objs_f.tolist() == objs.tolist()
    assert torch.allclose(boxes_f[:n, 0], 1 - (boxes[:n, 0] + boxes[:n, 2]), atol=1e-6) and torch.equal(boxes_f[:n, 1:], boxes[:n, 1:])


def test_vg_layouts_match_the_reference_loader(tmp_path):
    fx = load_fixture("datasets.npz")
    root = tiny_datasets.write(str(tmp_path))
    ds = D.VgLayoutDataset(os.path.join(root, "vocab.json"), os.path.join(root, "vg.npz"), os.path.join(root, "images"),
                           image_size=(32, 32), max_objects=10, left_right_flip=True)
    assert len(ds) == int(fx["vg_len"]) == 6
    for i in range(len(ds)):
        random.seed(1000 + i)   # the object subset is drawn with random.sample, as in the reference
        im, objs, boxes = ds[i]
        assert objs.shape == (11,) and boxes.shape == (11, 4)
        assert np.abs(im.numpy() - fx[f"vg_img{i}"]).max() == 0.0
        assert objs.tolist() == fx[f"vg_objs{i}"].tolist()
        assert np.abs(boxes.numpy() - fx[f"vg_boxes{i}"]).max() < 1e-6
        n = int((objs != 0).sum())
        assert boxes[n].tolist() == [0.0, 0.0, 1.0, 1.0] and int(objs[n]) == 0    # the __image__ slot (data/vg.py:120,135)


def test_loader_shards_and_device_batcher(tmp_path):
    root = tiny_datasets.write(str(tmp_path))
    args = (os.path.join(root, "images"), os.path.join(root, "instances.json"), os.path.join(root, "stuff.json"))
    ds = D.CocoLayoutDataset(*args, image_size=(32, 32), left_right_flip=True)
    seen = []
    for rank in range(2):
        ld = D.make_loader(ds, batch_size=1, num_workers=0, shuffle=False, rank=rank, world=2)
        seen.append([int(b[1][0, 0]) for b in ld])
        assert len(seen[-1]) == 3
    assert sorted(seen[0] + seen[1]) == sorted(int(ds[i][1][0]) for i in range(6))
    # raw uint8 images of mixed sizes, resized + normalised by the batcher (here on the CPU; on the GPU in training)
    raw = D.CocoLayoutDataset(*args, image_size=(32, 32), raw_images=True)
    imgs, objs, boxes = next(iter(D.make_loader(raw, batch_size=3, num_workers=0, shuffle=False)))
    assert isinstance(imgs, list) and imgs[0].dtype == torch.uint8 and imgs[0].shape[2] == 3 and objs.shape == (3, 8)
    out, o2, b2 = D.DeviceBatcher("cpu", (32, 32))((imgs, objs, boxes))
    ref = torch.stack([ds[i][0] for i in range(3)])
    assert out.shape == ref.shape and float((out - ref).abs().max()) < 3.0 / 255 * 2   # antialiased bilinear ~ PIL BILINEAR
