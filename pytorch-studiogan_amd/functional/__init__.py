"""autograd.Function wrappers: every forward/backward below is one or more launches of libsgamd.so kernels.

Internal activation layout is NHWC ([N,H,W,C] contiguous, fp32 or bf16). Weight operands come from the network's
WeightBank slot of the current forward (see bank.py). Gradients w.r.t. parameters are written by the kernels straight
into the gradient arena (``p.grad`` views) -- the Functions return None for parameter inputs on purpose.

One module per operator family (round 6; `studiogan_amd.functional` keeps exporting every name): _base -- shared state of the autograd layer, layout -- layout conversions at the reference's NCHW fp32 boundary, conv -- convolution / transposed convolution / linear / embedding autograd functions over the weight bank, norm -- (conditional) batch norm, synchronised across ranks, pointwise -- small elementwise operators of the residual blocks, attention -- self-attention core, dhead -- discriminator head, adversarial losses, gradient penalties, top-k, cond_heads -- class-conditioning heads and losses, augment -- differentiable augmentations in front of the discriminator and the consistency losses, ada_ops -- image-side operators of adaptive discriminator augmentation, info -- InfoGAN's Q-head operators."""
from ._base import *  # noqa: F401,F403
from .layout import *  # noqa: F401,F403
from .conv import *  # noqa: F401,F403
from .norm import *  # noqa: F401,F403
from .pointwise import *  # noqa: F401,F403
from .attention import *  # noqa: F401,F403
from .dhead import *  # noqa: F401,F403
from .cond_heads import *  # noqa: F401,F403
from .augment import *  # noqa: F401,F403
from .ada_ops import *  # noqa: F401,F403
from .info import *  # noqa: F401,F403
from . import _base, layout, conv, norm, pointwise, attention, dhead, cond_heads, augment, ada_ops, info  # noqa: F401
