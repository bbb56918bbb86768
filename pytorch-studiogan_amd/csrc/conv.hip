// conv.hip -- convolution forward / data-gradient / weight-gradient as implicit GEMM on the MFMA engine
// (gemm_core.h). Replaces nn.Conv2d + autograd's convolution_backward on the StudioGAN hot path
// (reference src/utils/ops.py:165-173,195-204; call sites models/big_resnet.py:28-42,177-242).
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

template <typename T> static int conv_fwd_t(const sg_conv_fwd_desc* d, hipStream_t st) {
  const int K = d->R * d->S * d->C;
  const int I = d->Cout;
  const long long Jll = (long long)d->N * d->Ho * d->Wo;
  SG_CHECK(Jll < (1ll << 31), "sg_conv2d_fwd: too many output pixels");
  const int J = (int)Jll;
  int pflags = d->pix_flags;
  if (d->epi_flags & SG_EPI_POOL) {
    SG_CHECK((d->Ho % 2 == 0) && (d->Wo % 2 == 0), "sg_conv2d_fwd: pooled output needs even Ho, Wo");
    pflags |= SG_PIX_QUAD;
  } else {
    pflags &= ~SG_PIX_QUAD;
  }
  StridedKC<T> lp;
  lp.base = (const T*)d->w; lp.bstride = 0; lp.ld = K; lp.rows = I; lp.K = K;
  lp.vec_ok = (K % ET<T>::VEC == 0) && aligned16(d->w);
  ConvPixKC<T> lq;
  fill_geom<T>(lq.g, d->x, d->N, d->Hs, d->Ws, d->C, d->ldx, d->Ho, d->Wo, d->R, d->S, d->stride, d->pad_h, d->pad_w, pflags);
  lq.rows = J; lq.K = K;
  Epilogue<T> e;
  e.out = d->out; e.out_bstride = 0; e.ldo = d->ldo; e.bias = d->bias;
  e.res = d->res; e.res_bstride = 0; e.ldr = d->ldr; e.beta = d->beta;
  e.mask = (const T*)d->mask; e.mask_bstride = 0; e.ldm = d->ldm;
  e.alpha = d->alpha; e.alpha_ptr = d->alpha_ptr; e.flags = d->epi_flags; e.I = I; e.J = J;
  const int prof = sg_prof_begin(st, 2.0 * (double)I * (double)J * (double)K, 0);
  if (I <= 32) sg_launch_gemm<T, StridedKC<T>, ConvPixKC<T>, 32, 256, 1, 4>(lp, lq, e, I, J, K, 1, 1, st);
  else if (I % 128 != 0 && (I % 96 == 0 || (I < 128 && I > 64))) sg_launch_gemm<T, StridedKC<T>, ConvPixKC<T>, 96, 256, 1, 4>(lp, lq, e, I, J, K, 1, 1, st);
  else sg_launch_gemm<T, StridedKC<T>, ConvPixKC<T>, 128, 128, 2, 2>(lp, lq, e, I, J, K, 1, 1, st);
  sg_prof_end(st, prof);
  SG_LAUNCH_CHECK();
  return 0;
}

extern "C" int sg_conv2d_fwd(const sg_conv_fwd_desc* d, sg_stream_t stream) {
  SG_CHECK(d && d->x && d->w && d->out, "sg_conv2d_fwd: null pointer");
  SG_CHECK(d->N > 0 && d->C > 0 && d->Cout > 0 && d->R > 0 && d->S > 0 && d->stride > 0, "sg_conv2d_fwd: bad shape");
  SG_CHECK(!(d->epi_flags & SG_EPI_ATOMIC), "sg_conv2d_fwd: atomic epilogue not supported");
  if (d->dtype == SG_DTYPE_F32) return conv_fwd_t<float>(d, (hipStream_t)stream);
  if (d->dtype == SG_DTYPE_BF16) return conv_fwd_t<bf16_t>(d, (hipStream_t)stream);
  sg_set_error("sg_conv2d_fwd: bad dtype");
  return -1;
}

template <typename T, bool TR> static int conv_wgrad_t(const sg_conv_wgrad_desc* d, hipStream_t st) {
  const int I = d->R * d->S * d->C;
  const int J = d->Cout;
  const long long Kll = (long long)d->N * d->Ho * d->Wo;
  SG_CHECK(Kll < (1ll << 31), "sg_conv2d_wgrad: too many pixels");
  const int K = (int)Kll;
  ConvPixMC<T> lp;
  fill_geom<T>(lp.g, d->x, d->N, d->xHs, d->xWs, d->C, d->ldx, d->Ho, d->Wo, d->R, d->S, d->stride, d->pad_h, d->pad_w,
               d->x_flags & ~SG_PIX_QUAD);
  lp.rows = I; lp.K = K;
  ConvPixMC<T> lq;
  fill_geom<T>(lq.g, d->dy, d->N, d->gHs, d->gWs, d->Cout, d->ldg, d->Ho, d->Wo, 1, 1, 1, 0, 0, d->g_flags & ~SG_PIX_QUAD);
  lq.rows = J; lq.K = K;
  Epilogue<T> e;
  e.out = d->dw; e.out_bstride = 0; e.ldo = I; e.bias = nullptr; e.res = nullptr; e.res_bstride = 0; e.ldr = 0; e.beta = 0.f;
  e.mask = nullptr; e.mask_bstride = 0; e.ldm = 0; e.alpha = d->alpha; e.alpha_ptr = d->alpha_ptr;
  e.flags = SG_EPI_ATOMIC | SG_EPI_OUT_F32; e.I = I; e.J = J;
  // tile config by output-channel count, then enough k-splits to fill 256 CUs a few times over
  int BI, BJ;
  if (I <= 32) { BI = 32; BJ = 256; }
  else if (J <= 32) { BI = 256; BJ = 32; }
  else if (J % 128 != 0 && (J % 96 == 0 || (J < 128 && J > 64))) { BI = 256; BJ = 96; }
  else { BI = 128; BJ = 128; }
  const int tiles = ((I + BI - 1) / BI) * ((J + BJ - 1) / BJ);
  int splits = d->splits;
  if (splits <= 0) {
    splits = (1024 + tiles - 1) / tiles;
    int maxs = K / (ET<T>::BK * 8);
    if (maxs < 1) maxs = 1;
    if (splits > maxs) splits = maxs;
    if (splits > 1024) splits = 1024;
  }
  const int prof = sg_prof_begin(st, 2.0 * (double)I * (double)J * (double)K, 1);
  if (BI == 32) sg_launch_gemm<T, ConvPixMC<T>, ConvPixMC<T>, 32, 256, 1, 4, TR>(lp, lq, e, I, J, K, splits, 1, st);
  else if (BJ == 32) sg_launch_gemm<T, ConvPixMC<T>, ConvPixMC<T>, 256, 32, 4, 1, TR>(lp, lq, e, I, J, K, splits, 1, st);
  else if (BJ == 96) sg_launch_gemm<T, ConvPixMC<T>, ConvPixMC<T>, 256, 96, 4, 1, TR>(lp, lq, e, I, J, K, splits, 1, st);
  else sg_launch_gemm<T, ConvPixMC<T>, ConvPixMC<T>, 128, 128, 2, 2, TR>(lp, lq, e, I, J, K, splits, 1, st);
  sg_prof_end(st, prof);
  SG_LAUNCH_CHECK();
  return 0;
}

extern "C" int sg_conv2d_wgrad(const sg_conv_wgrad_desc* d, sg_stream_t stream) {
  SG_CHECK(d && d->x && d->dy && d->dw, "sg_conv2d_wgrad: null pointer");
  SG_CHECK(d->N > 0 && d->C > 0 && d->Cout > 0 && d->R > 0 && d->S > 0 && d->stride > 0, "sg_conv2d_wgrad: bad shape");
  if (d->dtype == SG_DTYPE_F32) return conv_wgrad_t<float, true>(d, (hipStream_t)stream);
  if (d->dtype == SG_DTYPE_BF16) {
    if (d->no_tr) return conv_wgrad_t<bf16_t, false>(d, (hipStream_t)stream);
    return conv_wgrad_t<bf16_t, true>(d, (hipStream_t)stream);
  }
  sg_set_error("sg_conv2d_wgrad: bad dtype");
  return -1;
}
