// conv_common.h -- helpers shared by the convolution translation units (conv.hip: forward API + tile kernels, conv_v3.hip: halo
// kernel, conv_sk.hip: streaming kernel, conv_wgrad.hip: weight gradient). One translation unit per kernel family keeps
// `make -j` at the longest single family (~1.5 min) instead of their sum.
#pragma once
#include <stdlib.h>
#include "gemm_core.h"
#include "../../include/sgamd.h"

static inline int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int s = 0;
  while ((1 << s) < v) s++;
  return s;
}
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <typename T>
static void fill_geom(PixGeom<T>& g, const void* x, int N, int Hs, int Ws, int C, int ldx, int Ho, int Wo, int R, int S,
                      int stride, int pad_h, int pad_w, int flags) {
  g.x = (const T*)x; g.N = N; g.Hs = Hs; g.Ws = Ws;
  const int up = (flags & SG_PIX_UPSAMPLE) ? 2 : 1;
  g.Hin = Hs * up; g.Win = Ws * up; g.C = C; g.ldx = ldx; g.Ho = Ho; g.Wo = Wo;
  g.R = R; g.S = S; g.stride = stride; g.pad_h = pad_h; g.pad_w = pad_w; g.flags = flags;
  g.vec_ok = (C % ET<T>::VEC == 0) && (ldx % ET<T>::VEC == 0) && aligned16(x);
  g.wshift = ilog2_exact(Wo); g.hshift = ilog2_exact(Ho);
}


// the bf16 fast paths living in their own translation units; false = not eligible (the caller falls through to the next engine)
bool sg_conv_fwd_v3_try(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st);
bool sg_conv_fwd_sk_try(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st);
bool sg_conv_fwd_v4_try(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st);
bool sg_conv_fwd_rs_try(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st);   // conv_rs.hip
bool sg_conv_fwd_v4_skip_try(const sg_conv_fwd_desc* d, const sg_conv_skip_desc* sk, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st, bool dry);
