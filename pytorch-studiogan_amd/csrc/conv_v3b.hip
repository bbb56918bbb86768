// conv_v3b.hip -- the halo kernel's instantiations for layers whose last 64-channel slice is half full (C % 64 == 32: the 96-channel
// layers of BigGAN); see conv_v3.h (NKL) and conv_v3.hip (dispatcher). A translation unit of its own so that the two sets build in parallel.
#include "conv_common.h"
#include "conv_v3.h"
template int sg_conv_v3_dispatch<2>(int, int, const ConvV3Params&, const Epilogue<bf16_t>&, hipStream_t);
