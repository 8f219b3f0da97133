"""Deterministic miniature COCO-Stuff / Visual Genome datasets on disk (PNG images + annotation files), shared by
tools/capture_goldens.py (which runs the REFERENCE loaders on them) and the tests (which run layout2img_amd.data)."""
import json
import os

import numpy as np


def _image(rng, w, h):
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 255 // max(w + h - 2, 1))], -1)
    noise = rng.integers(0, 64, size=(h, w, 3))
    return np.clip(base * 3 // 4 + noise, 0, 255).astype(np.uint8)


def write(root, seed=7):
    import PIL.Image
    rng = np.random.default_rng(seed)
    img_dir = os.path.join(root, "images")
    os.makedirs(img_dir, exist_ok=True)
    sizes = [(64, 48), (40, 56), (50, 50), (72, 36), (33, 47), (60, 60)]
    images = []
    for i, (w, h) in enumerate(sizes):
        name = f"im{i}.png"
        PIL.Image.fromarray(_image(rng, w, h)).save(os.path.join(img_dir, name))
        images.append(dict(id=100 + i, file_name=name, width=w, height=h))
    inst_cats = [dict(id=1, name="person"), dict(id=2, name="dog"), dict(id=5, name="cup")]
    stuff_cats = [dict(id=92, name="sky"), dict(id=100, name="grass"), dict(id=183, name="other")]
    inst_ann, stuff_ann, aid = [], [], 0

    def box(w, h, fx, fy, fw, fh):
        return [round(fx * w, 2), round(fy * h, 2), round(fw * w, 2), round(fh * h, 2)]
    plan = {   # image -> [(category, fractional box, iscrowd)]: enough / too few / too many objects, tiny boxes, crowds, 'other'
        100: [(1, (.1, .1, .5, .6), 0), (2, (.5, .4, .4, .5), 0), (92, (0, 0, 1, .4), 0), (100, (0, .6, 1, .4), 0), (5, (.2, .2, .05, .05), 0)],
        101: [(1, (.2, .1, .3, .8), 0), (5, (.6, .5, .3, .3), 1), (92, (0, 0, 1, .5), 0), (183, (0, .5, 1, .5), 0)],       # 2 kept -> pruned
        102: [(2, (.1, .2, .4, .4), 0), (1, (.55, .1, .4, .8), 0), (5, (.3, .7, .2, .25), 0), (100, (0, .5, 1, .5), 0)],
        103: [(1, (.05 + .1 * k, .1, .3, .7), 0) for k in range(6)] + [(92, (0, 0, 1, .3), 0), (100, (0, .7, 1, .3), 0), (2, (.5, .5, .3, .4), 0)],  # 9 -> pruned
        104: [(1, (.1, .1, .6, .6), 0), (2, (.3, .3, .5, .5), 0), (5, (.1, .6, .3, .3), 0)],                                 # no stuff -> dropped (stuff_only)
        105: [(2, (.2, .2, .5, .5), 0), (1, (.0, .0, .3, .9), 0), (92, (0, 0, 1, .35), 0), (5, (.7, .6, .25, .3), 0), (100, (0, .8, 1, .2), 0)],
    }
    for im in images:
        for cat, fb, crowd in plan[im["id"]]:
            ann = dict(id=aid, image_id=im["id"], category_id=cat, bbox=box(im["width"], im["height"], *fb), iscrowd=crowd, area=1.0)
            aid += 1
            (stuff_ann if cat >= 92 else inst_ann).append(ann)
    with open(os.path.join(root, "instances.json"), "w") as f:
        json.dump(dict(images=images, categories=inst_cats, annotations=inst_ann), f)
    with open(os.path.join(root, "stuff.json"), "w") as f:
        json.dump(dict(images=images, categories=stuff_cats, annotations=stuff_ann), f)
    # ---- Visual Genome (sg2im preprocessing layout): 3 images, up to 12 objects, up to 6 relationships
    names = ["__image__", "man", "tree", "sky", "car", "dog", "road"]
    vocab = dict(object_name_to_idx={n: i for i, n in enumerate(names)}, object_idx_to_name=names)
    with open(os.path.join(root, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    N, MO, MR = 3, 12, 6
    per = [5, 12, 3]
    obj_names = np.full((N, MO), -1, np.int64)
    obj_boxes = np.full((N, MO, 4), -1, np.int64)
    for n in range(N):
        w, h = sizes[n]
        for k in range(per[n]):
            obj_names[n, k] = 1 + (n + k) % 6
            bw, bh = int(rng.integers(w // 5, w // 2)), int(rng.integers(h // 5, h // 2))
            obj_boxes[n, k] = [int(rng.integers(0, w - bw)), int(rng.integers(0, h - bh)), bw, bh]
    rels = [3, 6, 0]
    subj, obj = np.full((N, MR), -1, np.int64), np.full((N, MR), -1, np.int64)
    for n in range(N):
        for r in range(rels[n]):
            subj[n, r], obj[n, r] = r % per[n], (r + 2) % per[n]
    np.savez(os.path.join(root, "vg.npz"), image_paths=np.array([f"im{i}.png" for i in range(N)]), object_names=obj_names,
             object_boxes=obj_boxes, objects_per_image=np.array(per), relationships_per_image=np.array(rels),
             relationship_subjects=subj, relationship_objects=obj)
    return root
