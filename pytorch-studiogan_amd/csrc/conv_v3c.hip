// conv_v3c.hip -- the halo kernel's four-wave instantiations (one wave per SIMD; see conv_v3.h sg_conv_v3_dispatch_nw4). A translation unit of
// its own so that the sets build in parallel.
#include "conv_common.h"
#include "conv_v3.h"
template int sg_conv_v3_dispatch_nw4<4>(int, int, const ConvV3Params&, const Epilogue<bf16_t>&, hipStream_t);
