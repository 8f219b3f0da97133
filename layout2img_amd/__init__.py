"""layout2img_amd -- MI355X (gfx950) native hot path of wtliao/layout2img.

Public surface mirrors the reference's module boundary:
  ResnetGenerator128_context, CombineDiscriminator128_app  (+ GanTrainer for the training iteration).
The compute path is the C-ABI library libl2i_hip.so (include/l2i.h); it must be built first
(`python -m layout2img_amd.build`) and there is no CPU / PyTorch fallback.
"""
from .discriminator import CombineDiscriminator64, CombineDiscriminator128_app, ResnetDiscriminator128_app  # noqa: F401
from .generator import ResnetGenerator64_context, ResnetGenerator128_context, context_aware_generator  # noqa: F401
from .trainer import FlatAdam, GanTrainer  # noqa: F401
from .sampling import (GraphSampler, load_checkpoint, load_reference_checkpoint, reference_state_dict, sample, save_checkpoint,  # noqa: F401
                       truncated_normal)
from .vgg import VGGLoss, Vgg19  # noqa: F401
