"""Dataset front-end: real COCO-Stuff / Visual Genome layouts with the contract of the
reference loaders -- `CocoSceneGraphDataset.__getitem__` (data/cocostuff_loader.py:222-380) and
`VgSceneGraphDataset.__getitem__` (data/vg.py:71-161):

    image  (3, H, W) float32 in [-1, 1]   (bilinear resize, /255, (x - 0.5) / 0.5)
    objs   (O,) int64      O = 8 (COCO) / max_objects + 1 = 31 (VG); padding slots carry label 0 (`__image__`)
    boxes  (O, 4) float32  (x0, y0, w, h) in [0, 1]; padding box [-0.6, -0.6, 0.5, 0.5]; VG: one `__image__` slot
                           (label 0, box [0, 0, 1, 1]) directly after the real objects

Only the standard library, numpy, PIL and torch are needed (no torchvision / pycocotools / skimage: the reference
imports them but its __getitem__ uses none of their arithmetic beyond ToTensor / Normalize). The VG arrays are read
from the sg2im-style `.h5` when h5py is importable, or from an `.npz` holding the same arrays.

MI355X side: `DeviceBatcher` keeps the decode on the host workers but moves resize + normalise to the GPU (one
antialiased bilinear `interpolate` per batch on uint8 data uploaded once), and shards by rank for one-process-per-GPU
data parallelism (the reference scales the batch by the GPU count instead, train_context_app_v2.py:50-57).
"""
import json
import os
import random
from collections import defaultdict

import numpy as np
import torch
from torch.utils.data import Dataset

PAD_BOX = (-0.6, -0.6, 0.5, 0.5)


def _load_image(path, size, flip, normalize=True, raw=False):
    """PIL decode -> optional mirror -> RGB -> bilinear resize to (H, W) -> float CHW in [-1, 1]. Returns (tensor, (WW, HH))."""
    import PIL.Image
    import PIL.ImageOps
    H, W = size
    with open(path, "rb") as f:
        with PIL.Image.open(f) as image:
            if flip:
                image = PIL.ImageOps.mirror(image)
            WW, HH = image.size
            image = image.convert("RGB")
            if raw:   # uint8 HWC at the source resolution: the GPU resizes (DeviceBatcher)
                return torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()), (WW, HH)
            image = image.resize((W, H), PIL.Image.BILINEAR)
            arr = np.asarray(image, dtype=np.float32) / 255.0
    t = torch.from_numpy(arr).permute(2, 0, 1).contiguous()
    if normalize:
        t = (t - 0.5) / 0.5
    return t, (WW, HH)


class CocoLayoutDataset(Dataset):
    """COCO-Stuff layouts as the reference filters them (data/cocostuff_loader.py:15-192)."""

    def __init__(self, image_dir, instances_json, stuff_json=None, stuff_only=True, image_size=(128, 128),
                 normalize_images=True, max_samples=None, min_object_size=0.02, min_objects_per_image=3,
                 max_objects_per_image=8, left_right_flip=False, include_other=False, instance_whitelist=None,
                 stuff_whitelist=None, raw_images=False):
        super().__init__()
        self.image_dir, self.image_size = image_dir, tuple(image_size)
        self.normalize_images, self.max_samples = normalize_images, max_samples
        self.max_objects_per_image, self.left_right_flip, self.raw_images = max_objects_per_image, left_right_flip, raw_images
        with open(instances_json) as f:
            instances = json.load(f)
        stuff = None
        if stuff_json:
            with open(stuff_json) as f:
                stuff = json.load(f)
        self.image_ids, self.image_id_to_filename, self.image_id_to_size = [], {}, {}
        for im in instances["images"]:
            self.image_ids.append(im["id"])
            self.image_id_to_filename[im["id"]] = im["file_name"]
            self.image_id_to_size[im["id"]] = (im["width"], im["height"])
        name_to_idx, idx_to_name = {}, {}
        inst_names, stuff_names = [], []
        for cat in instances["categories"]:
            inst_names.append(cat["name"])
            idx_to_name[cat["id"]] = cat["name"]
            name_to_idx[cat["name"]] = cat["id"]
        for cat in (stuff["categories"] if stuff else []):
            stuff_names.append(cat["name"])
            idx_to_name[cat["id"]] = cat["name"]
            name_to_idx[cat["name"]] = cat["id"]
        whitelist = set(inst_names if instance_whitelist is None else instance_whitelist) | \
            set(stuff_names if stuff_whitelist is None else stuff_whitelist)

        def keep(obj):
            _, _, w, h = obj["bbox"]
            W, H = self.image_id_to_size[obj["image_id"]]
            name = idx_to_name[obj["category_id"]]
            return ((w * h) / (W * H) > min_object_size and name in whitelist and (name != "other" or include_other)
                    and obj["iscrowd"] != 1)
        self.image_id_to_objects = defaultdict(list)
        for obj in instances["annotations"]:
            if keep(obj):
                self.image_id_to_objects[obj["image_id"]].append(obj)
        if stuff:
            with_stuff = set()
            for obj in stuff["annotations"]:
                with_stuff.add(obj["image_id"])
                if keep(obj):
                    self.image_id_to_objects[obj["image_id"]].append(obj)
            if stuff_only:
                self.image_ids = [i for i in self.image_ids if i in with_stuff]
        name_to_idx["__image__"] = 0   # COCO category ids start at 1
        names = ["NONE"] * (1 + max(name_to_idx.values()))
        for n, i in name_to_idx.items():
            names[i] = n
        self.vocab = {"object_name_to_idx": name_to_idx, "object_idx_to_name": names}
        self.image_ids = [i for i in self.image_ids
                          if min_objects_per_image <= len(self.image_id_to_objects[i]) <= max_objects_per_image]

    def __len__(self):
        if self.max_samples is None:
            return len(self.image_ids) * (2 if self.left_right_flip else 1)
        return min(len(self.image_ids), self.max_samples)

    def __getitem__(self, index):
        flip = False
        if index >= len(self.image_ids):   # second half of a left_right_flip dataset: mirrored images
            index -= len(self.image_ids)
            flip = True
        image_id = self.image_ids[index]
        image, (WW, HH) = _load_image(os.path.join(self.image_dir, self.image_id_to_filename[image_id]), self.image_size,
                                      flip, self.normalize_images, self.raw_images)
        objs, boxes = [], []
        for obj in self.image_id_to_objects[image_id]:
            x, y, w, h = obj["bbox"]
            x0, y0, bw, bh = x / WW, y / HH, w / WW, h / HH
            if flip:
                x0 = 1 - (x0 + bw)
            objs.append(obj["category_id"])
            boxes.append((x0, y0, bw, bh))
        while len(objs) < self.max_objects_per_image:
            objs.append(0)
            boxes.append(PAD_BOX)
        return image, torch.tensor(objs, dtype=torch.long), torch.tensor(boxes, dtype=torch.float32)


def _read_vg_arrays(path):
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    try:
        import h5py
    except ImportError as e:
        raise RuntimeError(f"{path}: reading the sg2im .h5 needs h5py (not installed here); convert it to .npz "
                           "(same array names) or install h5py") from e
    with h5py.File(path, "r") as f:
        return {k: np.asarray(v) for k, v in f.items()}


class VgLayoutDataset(Dataset):
    """Visual Genome layouts (data/vg.py:19-161): up to `max_objects` objects sampled per image (those taking part in
    relationships first, then orphans), the `__image__` slot, padding up to max_objects + 1 slots."""

    def __init__(self, vocab_json, data_path, image_dir, image_size=(128, 128), normalize_images=True, max_objects=30,
                 max_samples=None, use_orphaned_objects=True, left_right_flip=False, raw_images=False):
        super().__init__()
        self.image_dir, self.image_size, self.normalize_images = image_dir, tuple(image_size), normalize_images
        self.max_objects, self.max_samples = max_objects, max_samples
        self.use_orphaned_objects, self.left_right_flip, self.raw_images = use_orphaned_objects, left_right_flip, raw_images
        with open(vocab_json) as f:
            self.vocab = json.load(f)
        arrays = _read_vg_arrays(data_path)
        self.image_paths = [p.decode() if isinstance(p, bytes) else str(p) for p in arrays.pop("image_paths")]
        self.data = {k: torch.from_numpy(np.asarray(v).astype(np.int64)) for k, v in arrays.items()}

    def _num(self):
        return self.data["object_names"].size(0)

    def __len__(self):
        if self.max_samples is not None:
            return min(self.max_samples, self._num())
        return self._num() * (2 if self.left_right_flip else 1)

    def __getitem__(self, index):
        flip = False
        if index >= self._num():
            index -= self._num()
            flip = True
        image, (WW, HH) = _load_image(os.path.join(self.image_dir, self.image_paths[index]), self.image_size, flip,
                                      self.normalize_images, self.raw_images)
        d = self.data
        with_rels, without = set(), set(range(int(d["objects_per_image"][index])))
        for r in range(int(d["relationships_per_image"][index])):
            for key in ("relationship_subjects", "relationship_objects"):
                i = int(d[key][index, r])
                with_rels.add(i)
                without.discard(i)
        obj_idxs, without = list(with_rels), list(without)
        if len(obj_idxs) > self.max_objects - 1:
            obj_idxs = random.sample(obj_idxs, self.max_objects)
        if len(obj_idxs) < self.max_objects - 1 and self.use_orphaned_objects:
            obj_idxs += random.sample(without, min(self.max_objects - 1 - len(obj_idxs), len(without)))
        O = len(obj_idxs) + 1
        objs = torch.zeros(self.max_objects + 1, dtype=torch.long)
        boxes = torch.tensor(PAD_BOX).repeat(self.max_objects + 1, 1)
        for i, oi in enumerate(obj_idxs):
            objs[i] = int(d["object_names"][index, oi])
            x, y, w, h = [float(v) for v in d["object_boxes"][index, oi]]
            x0, y0, bw, bh = x / WW, y / HH, w / WW, h / HH
            if flip:
                x0 = 1 - (x0 + bw)
            boxes[i] = torch.tensor([x0, y0, bw, bh])
        objs[O - 1] = self.vocab["object_name_to_idx"]["__image__"]   # the special slot: label 0, the whole canvas
        boxes[O - 1] = torch.tensor([0.0, 0.0, 1.0, 1.0])
        return image, objs, boxes


def get_dataset(dataset, img_size, root=".", raw_images=False):
    """The two configurations of train_context_app_v2.py:25-35. raw_images: the samples carry the decoded uint8 image at its
    source resolution and DeviceBatcher resizes + normalises on the GPU (the training entry's default)."""
    if dataset == "coco":
        return CocoLayoutDataset(os.path.join(root, "datasets/coco/images/train2017/"),
                                 os.path.join(root, "datasets/coco/annotations/instances_train2017.json"),
                                 os.path.join(root, "datasets/coco/annotations/stuff_train2017.json"),
                                 stuff_only=True, image_size=(img_size, img_size), left_right_flip=True, raw_images=raw_images)
    if dataset == "vg":
        h5 = os.path.join(root, "data/tmp/preprocess_vg/train.h5")
        return VgLayoutDataset(os.path.join(root, "data/tmp/vocab.json"), h5 if os.path.exists(h5) else h5[:-3] + ".npz",
                               os.path.join(root, "datasets/vg/"), image_size=(img_size, img_size), max_objects=30,
                               left_right_flip=True, raw_images=raw_images)
    raise ValueError(dataset)


def make_loader(dataset, batch_size, num_workers=2, shuffle=True, rank=0, world=1, seed=0):
    """DataLoader(drop_last, shuffle) as train_context_app_v2.py:62-64; under data parallelism each rank draws its own
    shard (DistributedSampler) with a per-rank batch of `batch_size`."""
    sampler = None
    if world > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=shuffle,
                                                                  seed=seed, drop_last=True)
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, drop_last=True, shuffle=shuffle and sampler is None,
                                       sampler=sampler, num_workers=num_workers, collate_fn=_collate if getattr(dataset, "raw_images", False) else None,
                                       pin_memory=torch.cuda.is_available())   # (pinned batches: DeviceBatcher's non_blocking copies do not stall the host)


def _collate(batch):
    """raw_images datasets: images stay a list of uint8 HWC tensors of different sizes"""
    return [b[0] for b in batch], torch.stack([b[1] for b in batch]), torch.stack([b[2] for b in batch])


class DeviceBatcher:
    """Host -> device hand-over of one batch: labels / boxes in one copy each; images either already resized on the host
    (float CHW) or raw uint8 HWC of mixed sizes, which are uploaded as they are and resized + normalised on the GPU
    (antialiased bilinear: what PIL's BILINEAR filter does when shrinking; agrees with the host path to ~1/255)."""

    def __init__(self, device, image_size=(128, 128)):
        self.device, self.image_size = torch.device(device), tuple(image_size)

    def __call__(self, batch):
        images, objs, boxes = batch
        if isinstance(images, (list, tuple)):
            out = []
            for im in images:
                x = im.to(self.device, non_blocking=True).permute(2, 0, 1).unsqueeze(0).float()
                x = torch.nn.functional.interpolate(x, size=self.image_size, mode="bilinear", antialias=True, align_corners=False)
                out.append(x)
            images = (torch.cat(out).clamp_(0, 255) / 255.0 - 0.5) / 0.5
        else:
            images = images.to(self.device, non_blocking=True)
        return images, objs.to(self.device, non_blocking=True), boxes.to(self.device, non_blocking=True)
