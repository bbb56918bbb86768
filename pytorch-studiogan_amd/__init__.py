"""studiogan_amd -- MI355X (gfx950) native implementation of StudioGAN's GAN training step and FID/IS
feature-extraction hot path. Host side mirrors the reference's operator / backbone / loss / optimizer interfaces
(reference src/utils/ops.py, src/models/*.py, src/utils/losses.py, src/config.py:497-565, src/utils/ema.py); all
device work is done by hand-written HIP kernels in libsgamd.so (C ABI: include/sgamd.h)."""
from . import _lib
from ._lib import build, lib, LIB_PATH

__all__ = ["build", "lib", "LIB_PATH"]
