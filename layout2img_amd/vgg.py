"""VGG19 perceptual ("feature matching") loss of the generator step on the HIP path.

Mirrors `VGGLoss` / `Vgg19` of the reference (utils/util.py:49-94, used at train_context_app_v2.py:141,185): the
torchvision VGG19 `features` stack up to relu5_1 cut into five slices, L1 distance between the slice outputs of the
fake and the real image with weights 1/32, 1/16, 1/8, 1/4, 1; the real branch is detached; the VGG weights are frozen.
Images go in as they are ([-1, 1], no ImageNet normalisation -- as in the reference). state_dict keys are the
reference's (`vgg.slice1.0.weight`, `vgg.slice2.2.bias`, ...); `load_torchvision_state_dict` maps a torchvision
`vgg19().state_dict()` (`features.N.weight`) onto them. The pretrained weights themselves cannot be fetched in this
environment: parity is checked with recipe weights (tests/golden/vgg.npz).

Every 3x3 convolution runs on the MFMA implicit-GEMM kernel (ReLU fused as the next layer's prologue / this layer's
epilogue copy); frozen weights are packed ONCE; the backward pass is data gradients only.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .arena import FlatParams, GemmWeight, WeightArena
from .ops import RELU, fused_conv

# (index in torchvision's vgg19().features, in channels, out channels); "M" = max-pool 2x2
_SLICES = {
    "slice1": [(0, 3, 64)],
    "slice2": [(2, 64, 64), "M", (5, 64, 128)],
    "slice3": [(7, 128, 128), "M", (10, 128, 256)],
    "slice4": [(12, 256, 256), (14, 256, 256), (16, 256, 256), "M", (19, 256, 512)],
    "slice5": [(21, 512, 512), (23, 512, 512), (25, 512, 512), "M", (28, 512, 512)],
}


class _Slice(nn.Module):
    def __init__(self, spec):
        super().__init__()
        self.spec = spec
        for item in spec:
            if item != "M":
                idx, ci, co = item
                self.add_module(str(idx), GemmWeight("conv", co, ci, 3, bias=True, sn=False))


def _max_pool(x):
    """2x2 max-pool of an NHWC stream (torch's channels-last kernel on a permuted view: no layout copies)."""
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)


class Vgg19(nn.Module):
    def __init__(self, requires_grad=False):
        super().__init__()
        for name, spec in _SLICES.items():
            setattr(self, name, _Slice(spec))
        for p in self.parameters():
            p.requires_grad = requires_grad

    def finalize(self, device, op_dtype=torch.bfloat16):
        self.op_dtype = op_dtype
        self.flat = FlatParams(self, device)
        self.arena = WeightArena(self, self.flat, device, op_dtype)
        self._pc = None
        return self

    def repack(self):
        """Pack the (frozen) weights for the MFMA kernels; call again after loading other weights."""
        with torch.no_grad():
            self._pc = self.arena.prepare(training=False, need_wgrad=False)
        return self._pc

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._pc = None
        return r

    def forward(self, x):
        """x: (b, H, W, 8) NHWC f32 (3 real channels). Returns the five PRE-activation taps (relu is applied by the loss)."""
        pc = self._pc or self.repack()
        taps = []
        first = True
        for name in _SLICES:
            sl = getattr(self, name)
            for item in sl.spec:
                if item == "M":
                    x = _max_pool(x).contiguous()
                    continue
                conv = getattr(sl, str(item[0]))
                x = fused_conv(x, conv, pc, prologue=None if first else RELU)   # conv(relu(previous stream)); max and relu commute
                first = False
            taps.append(x)
        return taps


class VGGLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.vgg = Vgg19()
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def finalize(self, device, op_dtype=torch.bfloat16):
        self.vgg.finalize(device, op_dtype)
        return self

    def load_torchvision_state_dict(self, sd):
        """sd: `torchvision.models.vgg19().state_dict()` (keys `features.N.weight|bias`)."""
        own = {}
        for name, spec in _SLICES.items():
            for item in spec:
                if item != "M":
                    for leaf in ("weight", "bias"):
                        own[f"vgg.{name}.{item[0]}.{leaf}"] = sd[f"features.{item[0]}.{leaf}"]
        self.load_state_dict(own)
        self.vgg.repack()

    @staticmethod
    def _nhwc8(img):
        return F.pad(img.permute(0, 2, 3, 1), (0, 8 - img.size(1))).contiguous()

    def forward(self, x, y):
        """x: fake images (b,3,H,W) (gradient flows), y: real images (detached, as the reference does)."""
        if not x.is_cuda:
            raise RuntimeError("layout2img_amd.VGGLoss runs on the GPU HIP path only")
        fx = self.vgg(self._nhwc8(x))
        with torch.no_grad():
            fy = self.vgg(self._nhwc8(y))
        loss = 0
        for w, a, b in zip(self.weights, fx, fy):
            c = a.shape[-1]   # (taps have their true channel count: 64..512, no padding)
            loss = loss + ops.l1_loss(F.relu(a), F.relu(b), w)
        return loss
