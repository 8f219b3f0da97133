"""GPU leg of the dataset front-end (reference data/cocostuff_loader.py:222-380, data/vg.py:71-161,
train_context_app_v2.py:25-35,62-64): miniature datasets on disk -> dataset classes with raw uint8 images -> loader ->
DeviceBatcher on cuda (GPU-side resize + normalise) -> one GanTrainer.step; and the training entry on a dataset on disk."""
import os
import random
import shutil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _trainer64():
    import layout2img_amd as L
    torch.manual_seed(0)
    g = L.ResnetGenerator64_context(num_classes=184).finalize(DEV, torch.float32)
    d = L.CombineDiscriminator64(num_classes=184).finalize(DEV, torch.float32)
    return L.GanTrainer(g, d)


def test_coco_dataset_to_trainer_through_the_device_batcher(tmp_path):
    from layout2img_amd import data as D
    from tests.golden import tiny_datasets
    from tests.helpers import load_fixture
    fx = load_fixture("datasets.npz")
    root = tiny_datasets.write(str(tmp_path))
    args = (os.path.join(root, "images"), os.path.join(root, "instances.json"), os.path.join(root, "stuff.json"))
    raw = D.CocoLayoutDataset(*args, image_size=(64, 64), left_right_flip=True, raw_images=True)
    host = D.CocoLayoutDataset(*args, image_size=(64, 64), left_right_flip=True)
    batcher = D.DeviceBatcher(DEV, (64, 64))
    seen = 0
    tr = _trainer64()
    for bi, batch in enumerate(D.make_loader(raw, batch_size=3, num_workers=0, shuffle=False)):
        imgs, objs, boxes = batcher(batch)
        assert imgs.is_cuda and imgs.shape == (3, 3, 64, 64) and objs.is_cuda and boxes.is_cuda
        for k in range(3):
            i = bi * 3 + k
            # labels and boxes: bit-equal to what the REFERENCE loader produced on this dataset (tests/golden/datasets.npz)
            assert objs[k].tolist() == fx[f"coco_objs{i}"].tolist()
            assert np.abs(boxes[k].cpu().numpy().astype(np.float64) - fx[f"coco_boxes{i}"]).max() < 1e-6
            # images: the GPU resize (float, antialiased bilinear) against the host path (PIL BILINEAR: 8-bit fixed point,
            # rounded to uint8 after each of its two passes) -- not bit-equal by construction; within 2 grey levels of 255
            # (2 * 2/255 on the [-1, 1] scale), typically below 1
            ref = host[i][0]
            err = float((imgs[k].cpu() - ref).abs().max()) * 255.0 / 2.0
            assert err <= 2.0, (i, err)
            seen += 1
        r = tr.step(imgs, objs, boxes)
        assert torch.isfinite(r["d_loss"]) and torch.isfinite(r["g_loss"]) and r["fake"].shape == (3, 3, 64, 64)
    assert seen == 6


def test_vg_dataset_to_trainer_through_the_device_batcher(tmp_path):
    import layout2img_amd as L
    from layout2img_amd import data as D, generator as G
    from tests.golden import tiny_datasets
    from tests.helpers import load_fixture
    fx = load_fixture("datasets.npz")
    root = tiny_datasets.write(str(tmp_path))
    raw = D.VgLayoutDataset(os.path.join(root, "vocab.json"), os.path.join(root, "vg.npz"), os.path.join(root, "images"),
                            image_size=(128, 128), max_objects=10, left_right_flip=True, raw_images=True)
    batcher = D.DeviceBatcher(DEV, (128, 128))
    samples = []
    for i in range(len(raw)):
        random.seed(1000 + i)   # the object subset is drawn with random.sample, as in the reference
        samples.append(raw[i])
    imgs, objs, boxes = batcher(D._collate(samples[:2]))
    for k in range(2):
        assert objs[k].tolist() == fx[f"vg_objs{k}"].tolist()
        assert np.abs(boxes[k].cpu().numpy() - fx[f"vg_boxes{k}"]).max() < 1e-6
    torch.manual_seed(0)
    g = G.context_aware_generator(num_classes=179).finalize(DEV, torch.bfloat16)
    d = L.CombineDiscriminator128_app(num_classes=179).finalize(DEV, torch.bfloat16)
    r = L.GanTrainer(g, d).step(imgs, objs, boxes)
    assert torch.isfinite(r["d_loss"]) and torch.isfinite(r["g_loss"]) and r["fake"].shape == (2, 3, 128, 128)


@pytest.mark.parametrize("graph", [True, False])
def test_training_entry_on_a_dataset_on_disk(tmp_path, graph):
    """python -m layout2img_amd.train without --synthetic: the reference's directory layout (train_context_app_v2.py:25-35),
    loader workers, GPU-side resize, one epoch -- as the replayed HIP graph (the default at one GPU) and eagerly."""
    from layout2img_amd import train
    from tests.golden import tiny_datasets
    src = tiny_datasets.write(str(tmp_path / "src"))
    root = tmp_path / "root"
    (root / "datasets/coco/annotations").mkdir(parents=True)
    shutil.copytree(os.path.join(src, "images"), root / "datasets/coco/images/train2017")
    shutil.copy(os.path.join(src, "instances.json"), root / "datasets/coco/annotations/instances_train2017.json")
    shutil.copy(os.path.join(src, "stuff.json"), root / "datasets/coco/annotations/stuff_train2017.json")
    argv = ["--dataset", "coco", "--batch_size", "2", "--total_epoch", "1", "--out_path", str(tmp_path / "out"), "--data_root", str(root),
            "--img_size", "64", "--dtype", "f32", "--num_workers", "1"] + ([] if graph else ["--no_graph"])
    tr = train.main(argv)
    assert int(tr.g_opt.t_dev) == 3   # 6 samples / batch 2 = 3 iterations (the capture's warm-up iterations are undone)
    assert (tmp_path / "out" / "coco" / "64" / "model" / "G_1.pth").exists()
