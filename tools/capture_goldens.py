"""Generate tests/golden/*.npz by importing the REAL reference (build container only).

Patches applied to make the reference importable/runnable on CPU:
 (1) stub `torchvision` (`ops.RoIAlign` = the oracle's restated roi_align, `ops.RoIPool`, `models`);
 (2) `torch.Tensor.cuda` -> identity; (3) always pass z_im; (4) clone bbox per D call;
 (5) Adam betas as floats; (6) Dropout2d p = 0.
Parameters come from tests/golden/recipe.py (loaded into the reference with load_state_dict), so the
fixtures hold only key names/shapes, inputs and the reference's outputs. Nothing of the reference's
source travels. Run:  python tools/capture_goldens.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import model as O  # noqa: E402
from tests.golden import recipe  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def patch_env():
    class RoIAlign(nn.Module):
        def __init__(self, output_size, spatial_scale, sampling_ratio):
            super().__init__()
            self.o, self.s, self.r = output_size, spatial_scale, sampling_ratio

        def forward(self, x, rois):
            return O.roi_align(x, rois, self.o[0], self.s, self.r)

    tv = types.ModuleType("torchvision")
    tv.ops = types.ModuleType("torchvision.ops")
    tv.ops.RoIAlign = RoIAlign
    tv.ops.RoIPool = RoIAlign
    tv.models = types.ModuleType("torchvision.models")

    def vgg19(pretrained=True):
        """torchvision.models.vgg19's `features` stack (configuration "E", no batch norm), random-initialised: the
        reference's utils/util.py::Vgg19 only slices `.features[0:30]`. (The pretrained weights cannot be fetched here.)"""
        layers, c = [], 3
        for v in [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                c = v
        m = nn.Module()
        m.features = nn.Sequential(*layers)
        return m
    tv.models.vgg19 = vgg19
    sys.modules.update({"torchvision": tv, "torchvision.ops": tv.ops, "torchvision.models": tv.models})
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


def shapes_of(m):
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def load(m, seed):
    sd = recipe.make_state_dict(shapes_of(m), seed)
    m.load_state_dict(sd)
    for mod in m.modules():
        if isinstance(mod, nn.Dropout2d):
            mod.p = 0.0
    return m


def keys_blob(shapes):
    ks = sorted(shapes)
    return np.array(ks), np.array([",".join(map(str, shapes[k])) for k in ks])


def grad_norms(m):
    names = sorted(n for n, p in m.named_parameters())
    d = dict(m.named_parameters())
    return np.array(names), np.array([0.0 if d[n].grad is None else float(d[n].grad.norm()) for n in names], dtype=np.float64)


def capture_generator(kind):
    if kind == "coco":
        from model.resnet_generator_app_v2 import ResnetGenerator128_context as G
        o, ncls, seed = 8, 184, 11
    else:
        from model.resnet_generator_vg import context_aware_generator as G
        o, ncls, seed = 31, 179, 12
    torch.manual_seed(0)
    g = load(G(num_classes=ncls, output_dim=3), seed)
    inp = recipe.make_inputs(2, o, ncls, seed + 100)
    ks, shp = keys_blob(shapes_of(g))
    rec = dict(keys=ks, shapes=shp, **{k: v.numpy() for k, v in inp.items()})
    g.train()
    taps = {}
    hooks = []
    if kind == "coco":
        def tap(name):
            def hook(m, i, out):
                taps[name] = out.detach().numpy().copy()
            return hook

        def tap_res(k):
            def hook(m, i, out):
                taps[f"res{k}"] = out[0].detach().numpy().copy()
                taps[f"stage_in{k}"] = i[2].detach().numpy().copy()
            return hook
        hooks.append(g.mask_regress.register_forward_hook(tap("bmask")))
        hooks.append(g.context.register_forward_hook(tap("w")))
        hooks.append(g.final[2].register_forward_hook(tap("pre_tanh")))
        for k in (1, 2, 3, 4, 5):
            hooks.append(getattr(g, f"res{k}").register_forward_hook(tap_res(k)))
    out1 = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
    # scalar loss for the backward golden: fixed random projection of the image
    proj = torch.randn(out1.shape, generator=torch.Generator().manual_seed(5))
    g.zero_grad()
    (out1 * proj).sum().backward()
    gn_names, gn = grad_norms(g)
    rec.update(out_train1=out1.detach().numpy(), grad_names=gn_names, grad_norms=gn,
               grad_alpha1=g.alpha1.grad.numpy().copy() if kind == "coco" else np.zeros(1),
               grad_fc_bias=g.fc.bias.grad.numpy().copy(),
               grad_emb=g.label_embedding.weight.grad.numpy().copy())
    if kind == "coco":
        rec.update(tap_w=taps["w"], tap_bmask=taps["bmask"], tap_pre_tanh=taps["pre_tanh"],
                   tap_res1=taps["res1"], tap_res3_mean=taps["res3"].mean(axis=(0, 2, 3)),
                   tap_stage_in2=taps["stage_in2"], tap_stage_in5=taps["stage_in5"][:, :, ::4, ::4])
    for h in hooks:
        h.remove()
    out2 = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
    g.eval()
    out_eval = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
    rec.update(out_train2_sub=out2.detach().numpy()[:, :, ::2, ::2], out_eval=out_eval.detach().numpy())
    np.savez_compressed(os.path.join(OUT, f"g_{kind}.npz"), **rec)
    print(kind, "G captured; |out|max", float(out1.abs().max()), "eval", float(out_eval.abs().max()))


def capture_discriminator():
    from model.rcnn_discriminator_app import CombineDiscriminator128_app as D
    torch.manual_seed(0)
    d = load(D(num_classes=184), 21)
    inp = recipe.make_inputs(2, 8, 184, 121)
    ks, shp = keys_blob(shapes_of(d))
    rec = dict(keys=ks, shapes=shp, **{k: v.numpy() for k, v in inp.items()})
    d.train()
    label = inp["y"].unsqueeze(-1)
    real = inp["real"].clone().requires_grad_(True)
    o1 = d(real, inp["bbox"].clone(), label)
    d.zero_grad()
    g = torch.Generator().manual_seed(6)
    loss = sum((t * torch.randn(t.shape, generator=g)).sum() for t in o1)
    loss.backward()
    gn_names, gn = grad_norms(d)
    rec.update(train1_img=o1[0].detach().numpy(), train1_obj=o1[1].detach().numpy(), train1_app=o1[2].detach().numpy(),
               grad_names=gn_names, grad_norms=gn, grad_input_sub=real.grad.numpy()[:, :, ::4, ::4].copy(),
               grad_l7_w=d.obD.l7.weight_orig.grad.numpy().copy())
    o2 = d(inp["real"], inp["bbox"].clone(), label)
    d.eval()
    oe = d(inp["real"], inp["bbox"].clone(), label)
    rec.update(train2_img=o2[0].detach().numpy(), train2_obj=o2[1].detach().numpy(), train2_app=o2[2].detach().numpy(),
               eval_img=oe[0].detach().numpy(), eval_obj=oe[1].detach().numpy(), eval_app=oe[2].detach().numpy())
    np.savez_compressed(os.path.join(OUT, "d_coco.npz"), **rec)
    print("D captured", [tuple(t.shape) for t in o1])


def capture_discriminator64():
    """model/rcnn_discriminator_orig.CombineDiscriminator64 at 64x64 (it mutates bbox in place: pass a clone)."""
    from model.rcnn_discriminator_orig import CombineDiscriminator64 as D
    torch.manual_seed(0)
    d = load(D(num_classes=184), 41)
    inp = recipe.make_inputs(2, 8, 184, 141, size=64)
    ks, shp = keys_blob(shapes_of(d))
    rec = dict(keys=ks, shapes=shp, **{k: v.numpy() for k, v in inp.items()})
    d.train()
    label = inp["y"].unsqueeze(-1)
    real = inp["real"].clone().requires_grad_(True)
    o1 = d(real, inp["bbox"].clone(), label)
    d.zero_grad()
    g = torch.Generator().manual_seed(6)
    sum((t * torch.randn(t.shape, generator=g)).sum() for t in o1).backward()
    gn_names, gn = grad_norms(d)
    rec.update(train1_img=o1[0].detach().numpy(), train1_obj=o1[1].detach().numpy(), grad_names=gn_names, grad_norms=gn,
               grad_input_sub=real.grad.numpy()[:, :, ::2, ::2].copy())
    o2 = d(inp["real"], inp["bbox"].clone(), label)
    d.eval()
    oe = d(inp["real"], inp["bbox"].clone(), label)
    rec.update(train2_img=o2[0].detach().numpy(), train2_obj=o2[1].detach().numpy(), eval_img=oe[0].detach().numpy(),
               eval_obj=oe[1].detach().numpy())
    np.savez_compressed(os.path.join(OUT, "d64.npz"), **rec)
    print("D64 captured", [tuple(t.shape) for t in o1])


def capture_train_loop():
    """Two iterations of train_context_app_v2.py:148-189 (VGG term omitted) on reference modules."""
    from model.rcnn_discriminator_app import CombineDiscriminator128_app as D
    from model.resnet_generator_app_v2 import ResnetGenerator128_context as G
    torch.manual_seed(0)
    netG, netD = load(G(num_classes=184, output_dim=3), 31), load(D(num_classes=184), 32)
    netG.train(), netD.train()
    g_opt = torch.optim.Adam([{"params": [p], "lr": 1e-4} for p in netG.parameters()], betas=(0.0, 0.999))
    d_opt = torch.optim.Adam([{"params": [p], "lr": 1e-4} for p in netD.parameters()], betas=(0.0, 0.999))
    relu = torch.nn.ReLU()
    rec = {}
    for it in range(2):
        inp = recipe.make_inputs(2, 8, 184, 200 + it)
        real, label, bbox = inp["real"], inp["y"].unsqueeze(-1), inp["bbox"]
        netD.zero_grad()
        r_im, r_obj, r_app = netD(real, bbox.clone(), label)
        fake = netG(inp["z"], bbox, inp["z_im"], label.squeeze(-1))
        f_im, f_obj, f_app = netD(fake.detach(), bbox.clone(), label)
        d_loss = (1.0 * (relu(1 - r_obj).mean() + relu(1 + f_obj).mean()) + 0.1 * (relu(1 - r_im).mean() + relu(1 + f_im).mean()) +
                  1.0 * (relu(1 - r_app).mean() + relu(1 + f_app).mean()))
        d_loss.backward()
        d_opt.step()
        netG.zero_grad()
        g_im, g_obj, g_app = netD(fake, bbox.clone(), label)
        pixel = torch.nn.L1Loss()(fake, real).mean()
        g_loss = -g_obj.mean() * 1.0 - g_im.mean() * 0.1 + pixel - 1.0 * g_app.mean()
        g_loss.backward()
        g_opt.step()
        rec[f"d_loss{it}"], rec[f"g_loss{it}"], rec[f"pixel{it}"] = float(d_loss), float(g_loss), float(pixel)
        rec[f"fake_sub{it}"] = fake.detach().numpy()[:, :, ::4, ::4]
    names = sorted(n for n, _ in netG.named_parameters())
    dg, dd = dict(netG.named_parameters()), dict(netD.named_parameters())
    rec["g_param_names"] = np.array(names)
    rec["g_param_sums"] = np.array([float(dg[n].detach().double().sum()) for n in names])
    dn = sorted(n for n, _ in netD.named_parameters())
    rec["d_param_names"] = np.array(dn)
    rec["d_param_sums"] = np.array([float(dd[n].detach().double().sum()) for n in dn])
    np.savez_compressed(os.path.join(OUT, "train_loop.npz"), **rec)
    print("loop captured", {k: v for k, v in rec.items() if isinstance(v, float)})


def capture_vg():
    """BASELINE config 5 models (context_aware_generator + rcnn_discriminator_vg, 179 classes, o = 31) on layouts WITH
    the `__image__` slot (data/vg.py:120,135): a generator golden, a discriminator golden and two loop iterations."""
    from model.rcnn_discriminator_vg import CombineDiscriminator128_app as D
    from model.resnet_generator_vg import context_aware_generator as G
    torch.manual_seed(0)
    # --- generator
    g = load(G(num_classes=179, output_dim=3), 51)
    inp = recipe.make_inputs_vg(2, 31, 179, 151)
    ks, shp = keys_blob(shapes_of(g))
    rec = dict(keys=ks, shapes=shp, **{k: v.numpy() for k, v in inp.items()})
    g.train()
    out1 = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
    proj = torch.randn(out1.shape, generator=torch.Generator().manual_seed(5))
    g.zero_grad()
    (out1 * proj).sum().backward()
    gn_names, gn = grad_norms(g)
    g.eval()
    oe = g(inp["z"], inp["bbox"], inp["z_im"], inp["y"])
    rec.update(out_train1_sub=out1.detach().numpy()[:, :, ::2, ::2], grad_names=gn_names, grad_norms=gn,
               out_eval_sub=oe.detach().numpy()[:, :, ::2, ::2])
    np.savez_compressed(os.path.join(OUT, "g_vg_img.npz"), **rec)
    print("VG G (image slot) captured", float(out1.abs().max()))
    # --- discriminator at o = 31 / 179 classes
    d = load(D(num_classes=179), 52)
    ks, shp = keys_blob(shapes_of(d))
    rec = dict(keys=ks, shapes=shp, **{k: v.numpy() for k, v in inp.items()})
    d.train()
    label = inp["y"].unsqueeze(-1)
    real = inp["real"].clone().requires_grad_(True)
    o1 = d(real, inp["bbox"].clone(), label)
    d.zero_grad()
    gen = torch.Generator().manual_seed(6)
    sum((t * torch.randn(t.shape, generator=gen)).sum() for t in o1).backward()
    gn_names, gn = grad_norms(d)
    d.eval()
    oe = d(inp["real"], inp["bbox"].clone(), label)
    rec.update(train1_img=o1[0].detach().numpy(), train1_obj=o1[1].detach().numpy(), train1_app=o1[2].detach().numpy(),
               grad_names=gn_names, grad_norms=gn, grad_input_sub=real.grad.numpy()[:, :, ::4, ::4].copy(),
               eval_img=oe[0].detach().numpy(), eval_obj=oe[1].detach().numpy(), eval_app=oe[2].detach().numpy())
    np.savez_compressed(os.path.join(OUT, "d_vg.npz"), **rec)
    print("VG D captured", [tuple(t.shape) for t in o1])
    # --- two loop iterations (train_context_app_v2.py:148-189, VGG term omitted)
    netG, netD = load(G(num_classes=179, output_dim=3), 53), load(D(num_classes=179), 54)
    netG.train(), netD.train()
    g_opt = torch.optim.Adam([{"params": [p], "lr": 1e-4} for p in netG.parameters()], betas=(0.0, 0.999))
    d_opt = torch.optim.Adam([{"params": [p], "lr": 1e-4} for p in netD.parameters()], betas=(0.0, 0.999))
    relu = torch.nn.ReLU()
    rec = {}
    for it in range(2):
        inp = recipe.make_inputs_vg(2, 31, 179, 300 + it)
        real, label, bbox = inp["real"], inp["y"].unsqueeze(-1), inp["bbox"]
        netD.zero_grad()
        r_im, r_obj, r_app = netD(real, bbox.clone(), label)
        fake = netG(inp["z"], bbox, inp["z_im"], label.squeeze(-1))
        f_im, f_obj, f_app = netD(fake.detach(), bbox.clone(), label)
        d_loss = (1.0 * (relu(1 - r_obj).mean() + relu(1 + f_obj).mean()) + 0.1 * (relu(1 - r_im).mean() + relu(1 + f_im).mean()) +
                  1.0 * (relu(1 - r_app).mean() + relu(1 + f_app).mean()))
        d_loss.backward()
        d_opt.step()
        netG.zero_grad()
        g_im, g_obj, g_app = netD(fake, bbox.clone(), label)
        pixel = torch.nn.L1Loss()(fake, real).mean()
        g_loss = -g_obj.mean() * 1.0 - g_im.mean() * 0.1 + pixel - 1.0 * g_app.mean()
        g_loss.backward()
        g_opt.step()
        rec[f"d_loss{it}"], rec[f"g_loss{it}"], rec[f"pixel{it}"] = float(d_loss), float(g_loss), float(pixel)
        rec[f"fake_sub{it}"] = fake.detach().numpy()[:, :, ::4, ::4]
    for net, pre in ((netG, "g"), (netD, "d")):
        names = sorted(n for n, _ in net.named_parameters())
        dd = dict(net.named_parameters())
        rec[f"{pre}_param_names"] = np.array(names)
        rec[f"{pre}_param_sums"] = np.array([float(dd[n].detach().double().sum()) for n in names])
    np.savez_compressed(os.path.join(OUT, "train_loop_vg.npz"), **rec)
    print("VG loop captured", {k: v for k, v in rec.items() if isinstance(v, float)})


def capture_vgg():
    """The reference's own VGGLoss (utils/util.py:49-94) on the stubbed torchvision stack with recipe weights."""
    from utils.util import VGGLoss
    torch.manual_seed(0)
    m = VGGLoss()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = recipe.make_state_dict(shapes, 61)
    for k in sd:   # He-style scale so that activations neither vanish nor explode through 13 ReLU convs
        if k.endswith("weight"):
            sd[k] = sd[k] * (2.0 ** 0.5)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(62)
    x = (torch.rand(2, 3, 128, 128, generator=g) * 2 - 1).requires_grad_(True)
    y = torch.rand(2, 3, 128, 128, generator=g) * 2 - 1
    loss = m(x, y)
    loss.backward()
    feats = m.vgg(x.detach())
    ks, shp = keys_blob(shapes)
    # (inputs are regenerated by the tests from the same seed: tests/helpers.py::vgg_inputs)
    np.savez_compressed(os.path.join(OUT, "vgg.npz"), keys=ks, shapes=shp, x_sum=float(x.detach().double().sum()), loss=float(loss.detach()),
                        grad_x_sub=x.grad.numpy()[:, :, ::2, ::2].copy(),
                        tap_means=np.array([float(f.mean()) for f in feats]), tap5=feats[4].detach().numpy())
    print("vgg captured: loss", float(loss), [float(f.mean()) for f in feats])


def capture_datasets():
    """The reference's dataset classes (data/cocostuff_loader.py, data/vg.py) on the miniature datasets of
    tests/golden/tiny_datasets.py. Stubs for what is not installed: torchvision.transforms (Compose / ToTensor /
    Normalize -- scale to [0,1], CHW, (x - mean) / std), skimage, pycocotools (imported, never called), h5py (a
    read-only File over the .npz holding the same arrays)."""
    import random
    import tempfile
    from tests.golden import tiny_datasets
    T = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToTensor:
        def __call__(self, im):
            return torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()

    class Normalize:
        def __init__(self, mean, std):
            self.m, self.s = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)

        def __call__(self, x):
            return (x - self.m) / self.s
    T.Compose, T.ToTensor, T.Normalize = Compose, ToTensor, Normalize
    sys.modules["torchvision"].transforms = T
    sys.modules["torchvision.transforms"] = T
    sk, skt = types.ModuleType("skimage"), types.ModuleType("skimage.transform")
    skt.resize = None
    sys.modules.update({"skimage": sk, "skimage.transform": skt})
    pc, pcm = types.ModuleType("pycocotools"), types.ModuleType("pycocotools.mask")
    sys.modules.update({"pycocotools": pc, "pycocotools.mask": pcm})
    h5 = types.ModuleType("h5py")

    class File:
        def __init__(self, path, mode):
            self.z = np.load(path, allow_pickle=False)

        def __enter__(self):
            return {k: self.z[k] for k in self.z.files}

        def __exit__(self, *a):
            self.z.close()
    h5.File = File
    sys.modules["h5py"] = h5
    import PIL.ImageOps  # noqa: F401  (the reference calls PIL.ImageOps.mirror without importing the submodule)
    from data.cocostuff_loader import CocoSceneGraphDataset
    from data.vg import VgSceneGraphDataset
    rec = {}
    with tempfile.TemporaryDirectory() as root:
        tiny_datasets.write(root)
        ds = CocoSceneGraphDataset(os.path.join(root, "images"), os.path.join(root, "instances.json"), os.path.join(root, "stuff.json"),
                                   stuff_only=True, image_size=(32, 32), left_right_flip=True)
        rec["coco_len"], rec["coco_ids"] = len(ds), np.array(ds.image_ids)
        for i in range(len(ds)):
            im, objs, boxes = ds[i]
            rec[f"coco_img{i}"], rec[f"coco_objs{i}"], rec[f"coco_boxes{i}"] = im.numpy(), objs.numpy(), np.asarray(boxes, np.float64)
        vg = VgSceneGraphDataset(os.path.join(root, "vocab.json"), os.path.join(root, "vg.npz"), os.path.join(root, "images"),
                                 image_size=(32, 32), max_objects=10, left_right_flip=True)
        rec["vg_len"] = len(vg)
        for i in range(len(vg)):
            random.seed(1000 + i)
            im, objs, boxes = vg[i]
            rec[f"vg_img{i}"], rec[f"vg_objs{i}"], rec[f"vg_boxes{i}"] = im.numpy(), objs.numpy(), boxes.numpy()
    np.savez_compressed(os.path.join(OUT, "datasets.npz"), **rec)
    print("datasets captured: coco", rec["coco_len"], rec["coco_ids"], "vg", rec["vg_len"])


def capture_small_ops():
    """Known answers the survey lists for masks_to_layout / bbox_mask."""
    from model.resnet_generator_app_v2 import bbox_mask
    from utils.bilinear import masks_to_layout
    boxes = torch.tensor([[[0.0, 0.0, 1.0, 1.0], [0.25, 0.25, 0.5, 0.5], [-0.6, -0.6, 0.5, 0.5]]])
    m = masks_to_layout(boxes, torch.ones(1, 3, 16, 16), 64)
    g = torch.Generator().manual_seed(3)
    rm = torch.rand(1, 3, 16, 16, generator=g)
    bm = bbox_mask(torch.zeros(1), boxes, 64, 64)
    np.savez_compressed(os.path.join(OUT, "small_ops.npz"), boxes=boxes.numpy(), ones_layout=m.numpy(), rand_masks=rm.numpy(),
                        rand_layout=masks_to_layout(boxes, rm, 64).numpy(), bbox_mask=bm.numpy())
    print("small ops: sums", m.sum(dim=(2, 3)), bm.sum(dim=(2, 3)))


if __name__ == "__main__":
    patch_env()
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["small", "g_coco", "g_vg", "d", "d64", "loop", "vg", "vgg", "data"]
    with torch.random.fork_rng():
        if "small" in which:
            capture_small_ops()
        if "g_coco" in which:
            capture_generator("coco")
        if "g_vg" in which:
            capture_generator("vg")
        if "d" in which:
            capture_discriminator()
        if "d64" in which:
            capture_discriminator64()
        if "loop" in which:
            capture_train_loop()
        if "vg" in which:
            capture_vg()
        if "vggloss" in which or "vgg" in which:
            capture_vgg()
        if "data" in which:
            capture_datasets()
