"""Host mirrors of the reference's StyleGAN2 / StyleGAN3 native operators (reference src/utils/style_ops/{bias_act,upfirdn2d,filtered_lrelu}.py,
SURVEY.md 8(f4)): same function names, argument meaning and error behaviour, every launch a kernel of libsgamd.so (csrc/style.hip).
`impl='cuda'` (the default, kept for call-site compatibility -- on PyTorch-ROCm the device type is still 'cuda') runs the HIP kernels;
there is no `impl='ref'` here: the reference path lives in the reference and, restated for the tests, in oracle/style_ref.py."""
from . import bias_act, upfirdn2d, filtered_lrelu  # noqa: F401
