// conv.hip -- convolution forward / data-gradient / weight-gradient as implicit GEMM on the MFMA engine
// (gemm_core.h). Replaces nn.Conv2d + autograd's convolution_backward on the StudioGAN hot path
// (reference src/utils/ops.py:165-173,195-204; call sites models/big_resnet.py:28-42,177-242).
#include <stdlib.h>
#include "gemm_core.h"
#include "conv_v2.h"
#include "wgrad_v2.h"
#include "conv_v3.h"
#include "conv_sk.h"
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

template <typename T, bool FAST>
static void conv_fwd_launch(const sg_conv_fwd_desc* d, const Epilogue<T>& e, int I, int J, int K, int pflags, hipStream_t st) {
  typedef StridedKC<T, FAST> LP;
  typedef ConvPixKC<T, FAST> LQ;
  LP lp;
  lp.base = (const T*)d->w; lp.bstride = 0; lp.ld = K; lp.rows = I; lp.K = K;
  lp.vec_ok = (K % ET<T>::VEC == 0) && aligned16(d->w);
  LQ lq;
  fill_geom<T>(lq.g, d->x, d->N, d->Hs, d->Ws, d->C, d->ldx, d->Ho, d->Wo, d->R, d->S, d->stride, d->pad_h, d->pad_w, pflags);
  lq.rows = J; lq.K = K;
  if (I <= 32) sg_launch_gemm<T, LP, LQ, 32, 256, 1, 4>(lp, lq, e, I, J, K, 1, 1, st);
  else if (I % 128 != 0 && (I % 96 == 0 || (I < 128 && I > 64))) sg_launch_gemm<T, LP, LQ, 96, 256, 1, 4>(lp, lq, e, I, J, K, 1, 1, st);
  else sg_launch_gemm<T, LP, LQ, 128, 128, 2, 2>(lp, lq, e, I, J, K, 1, 1, st);
}

// second-generation kernel (conv_v2.h) for the hot bf16 shapes; returns false when the problem is not eligible
template <typename T> static bool conv_fwd_v2_try(const sg_conv_fwd_desc*, const Epilogue<T>&, int, int, int, int, hipStream_t) { return false; }
template <> bool conv_fwd_v2_try<bf16_t>(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st) {
  // SG_CONV_V2=0 disables the kernel, =force skips the "enough tiles to fill the chip" heuristic (used by the tests)
  const char* mode = getenv("SG_CONV_V2");
  const bool disabled = mode && mode[0] == '0';
  const bool force = mode && mode[0] == 'f';
  if (disabled || d->stride != 1 || (pflags & SG_PIX_TRANSPOSED)) return false;
  if (d->C % 8 || d->ldx % 8 || d->R * d->S > 25 || J < 256) return false;
  // descriptor extents: offsets with bit 31 (activations) / bit 30 (weights) set must be out of range
  const long long xbytes = (((long long)d->N * d->Hs * d->Ws - 1) * d->ldx + d->C) * 2, wbytes = (long long)I * K * 2;
  if (xbytes >= (1ll << 31) || wbytes >= (1ll << 30)) return false;
  if (!aligned16(d->x) || !aligned16(d->w)) return false;
  // the kernel's only epilogue: bf16 rows, 16-byte stores; its ReLU-mask OR residual tile (bf16) is pre-staged with 16-byte loads
  if ((e.flags & (SG_EPI_ATOMIC | SG_EPI_OUT_F32)) || (e.ldo & 7) || !aligned16(e.out)) return false;
  if (e.mask && e.res) return false;
  if (e.mask && ((e.ldm & 7) || !aligned16(e.mask))) return false;
  if (e.res && ((e.flags & SG_EPI_RES_F32) || (e.ldr & 7) || !aligned16(e.res))) return false;
  const int tj = (J + 255) / 256;
  const int cands[3] = {192, 128, 96};   // (a 256-wide cout tile puts part of its 128 accumulator registers in scratch with hipcc 7.2: left out)
  int best = 0, best_tiles = 0;
  for (int c = 0; c < 3; c++) {
    if (I % cands[c]) continue;
    const int tiles = (I / cands[c]) * tj;
    if (tiles >= 512) { best = cands[c]; best_tiles = tiles; break; }
    if (tiles > best_tiles) { best = cands[c]; best_tiles = tiles; }
  }
  if (!best || (best_tiles < 160 && !force)) return false;
  ConvV2Params p;
  p.x = (const bf16_t*)d->x; p.w = (const bf16_t*)d->w;
  p.N = d->N; p.Hs = d->Hs; p.Ws = d->Ws; p.C = d->C; p.ldx = d->ldx;
  const int up = (pflags & SG_PIX_UPSAMPLE) ? 2 : 1;
  p.Hin = d->Hs * up; p.Win = d->Ws * up; p.Ho = d->Ho; p.Wo = d->Wo;
  p.R = d->R; p.S = d->S; p.pad_h = d->pad_h; p.pad_w = d->pad_w; p.flags = pflags;
  p.I = I; p.J = J; p.K = K; p.cpt = d->C / 8; p.ntap = d->R * d->S;
  p.wshift = ilog2_exact(d->Wo); p.hshift = ilog2_exact(d->Ho);
  p.xbytes = (unsigned)xbytes; p.wbytes = (unsigned)wbytes;
  int rc = 0;
  // piece placement (conv_v2.h SCHED): the 96-wide tiles spread their DMA pieces over the MFMA sub-steps (SCHED 1), the 192/128-wide
  // ones issue them in front and prefetch fragments across sub-steps (SCHED 7); SCHED 2..6 are the ablation variants of a
  // -DSG_ABLATION build (no DMA / no fragment reads / neither / one k-tile / one k-tile without epilogue).
  if (best == 192) rc = sg_launch_conv_v2<192, 4, 2, 256, 7>(p, e, st);
  else if (best == 128) rc = sg_launch_conv_v2<128, 4, 2, 256, 7>(p, e, st);
  else {
    // 96 output channels: a 512-pixel tile gives every wave a 64 x 96 block (24 MFMAs per 20 fragment reads instead of 12 per 16)
    if (J >= 512 * 256) rc = sg_launch_conv_v2<96, 8, 1, 512, 1>(p, e, st);
    else rc = sg_launch_conv_v2<96, 8, 1, 256, 1>(p, e, st);
  }
  return rc == 0;
}

// halo kernel (conv_v3.h) for 3x3 / stride 1 / pad 1 with >= 64 input channels; returns false when the problem is not eligible.
// SG_CONV_V3=0 disables it, =force skips the tile-count heuristic (tests), =all also takes the shapes the default table leaves to v2.
template <typename T> static bool conv_fwd_v3_try(const sg_conv_fwd_desc*, const Epilogue<T>&, int, int, int, int, hipStream_t) { return false; }
template <> bool conv_fwd_v3_try<bf16_t>(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st) {
  const char* mode = getenv("SG_CONV_V3");
  if (mode && mode[0] == '0') return false;
  const bool force = mode && mode[0] == 'f';
  if (d->stride != 1 || (pflags & SG_PIX_TRANSPOSED) || d->R != 3 || d->S != 3 || d->pad_h != 1 || d->pad_w != 1) return false;
  if (d->C < 64 || d->C % 8 || d->ldx % 8 || !aligned16(d->x) || !aligned16(d->w)) return false;
  const bool up = (pflags & SG_PIX_UPSAMPLE) != 0, quad = (pflags & SG_PIX_QUAD) != 0;
  if (d->Ho != d->Hs * (up ? 2 : 1) || d->Wo != d->Ws * (up ? 2 : 1)) return false;
  const int wshift = ilog2_exact(d->Wo), hshift = ilog2_exact(d->Ho);
  if (wshift < 0 || hshift < 0 || d->Ws < 4 || d->Hs < 2) return false;
  const long long xbytes = (((long long)d->N * d->Hs * d->Ws - 1) * d->ldx + d->C) * 2, wbytes = (long long)I * K * 2;
  if (xbytes >= (1ll << 31) || wbytes >= (1ll << 30)) return false;
  if ((e.flags & (SG_EPI_ATOMIC | SG_EPI_OUT_F32)) || (e.ldo & 7) || !aligned16(e.out)) return false;
  if (e.mask && e.res) return false;
  if (e.mask && ((e.ldm & 7) || !aligned16(e.mask))) return false;
  if (e.res && ((e.flags & SG_EPI_RES_F32) || (e.ldr & 7) || !aligned16(e.res))) return false;
  const int tj = (J + 255) / 256;
  const int cands[3] = {192, 128, 96};
  int best = 0, best_tiles = 0;
  for (int c = 0; c < 3; c++) {
    if (I % cands[c]) continue;
    const int tiles = (I / cands[c]) * tj;
    if (tiles >= 512) { best = cands[c]; best_tiles = tiles; break; }
    if (tiles > best_tiles) { best = cands[c]; best_tiles = tiles; }
  }
  if (I <= 32 && I % 8 == 0 && (J >= 512 * 256 || force)) { best = 32; best_tiles = (J + 511) / 512; }   // narrow outputs (G's RGB layer, 8 padded couts): HBM-bound, one cout tile
  if (!best || (best_tiles < 160 && !force)) return false;
  int BJ = (best == 32 || (best == 96 && J >= 512 * 256)) ? 512 : 256;
  if (best == 96) { const char* bj = getenv("SG_V3_BJ96"); if (bj && bj[0] == '2') BJ = 256; }   // A/B: 256-pixel tiles (double patch buffer) for the 96-wide layers
  if ((quad || up) && (BJ % (2 * d->Wo))) return false;     // the tile must cover whole (pairs of) image rows
  if (J % d->Wo) return false;
  ConvV3Params p;
  p.x = (const bf16_t*)d->x; p.w = (const bf16_t*)d->w;
  p.W = d->Ws; p.wlog = ilog2_exact(d->Ws); p.C = d->C; p.ldx = d->ldx;
  p.Ho = d->Ho; p.Wo = d->Wo; p.wshift = wshift; p.hshift = hshift; p.flags = pflags;
  p.I = I; p.J = J; p.K = K; p.nslice = (d->C + 63) / 64;
  p.npix_src = d->N * d->Hs * d->Ws;
  p.npx = (up ? BJ / 4 : BJ) + 2 * d->Ws + 16;
  p.xbytes = (unsigned)xbytes; p.wbytes = (unsigned)wbytes;
  p.zero_off = 0; p.bias_off = 0;
  {   // DMA piece placement (conv_v3.h): SG_V3_SCHED=0 / 1 overrides the default
    static int sched = -1;
    if (sched < 0) { const char* e2 = getenv("SG_V3_SCHED"); sched = e2 ? (e2[0] - '0') : 1; }
    p.sched = sched;
  }
  if (((p.npx >> 3) + 7) / 8 >= 19) return false;           // would need more than 2 patch pieces per tap and wave (never with <= 160 KB of LDS)
  int rc;
  if (best == 192) rc = sg_launch_conv_v3<192, 4, 2, 256>(p, e, st);
  else if (best == 128) rc = sg_launch_conv_v3<128, 4, 2, 256>(p, e, st);
  else if (best == 32) rc = sg_launch_conv_v3<32, 8, 1, 512>(p, e, st);
  else if (BJ == 512) rc = sg_launch_conv_v3<96, 8, 1, 512>(p, e, st);
  else rc = sg_launch_conv_v3<96, 8, 1, 256>(p, e, st);
  return rc == 0;
}

// small-K streaming kernel (conv_sk.h): 1x1 convolutions with <= 192 input channels and the 3x3 stem over 8 padded channels.
// SG_CONV_SK=0 disables it, =force skips the problem-size heuristic (tests).
template <typename T> static bool conv_fwd_sk_try(const sg_conv_fwd_desc*, const Epilogue<T>&, int, int, int, int, hipStream_t) { return false; }
template <> bool conv_fwd_sk_try<bf16_t>(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st) {
  const char* mode = getenv("SG_CONV_SK");
  if (mode && mode[0] == '0') return false;
  const bool force = mode && mode[0] == 'f';
  if (d->stride != 1 || (pflags & SG_PIX_TRANSPOSED)) return false;
  const bool up = (pflags & SG_PIX_UPSAMPLE) != 0;
  const bool one = d->R == 1 && d->S == 1 && d->pad_h == 0 && d->pad_w == 0 && d->C <= 192;
  const bool stem = d->R == 3 && d->S == 3 && d->pad_h == 1 && d->pad_w == 1 && d->C == 8 && !up;
  if (!one && !stem) return false;
  if (d->C % 8 || d->ldx % 8 || I % 8 || I > 384 || !aligned16(d->x) || !aligned16(d->w)) return false;
  if (d->Ho != d->Hs * (up ? 2 : 1) || d->Wo != d->Ws * (up ? 2 : 1)) return false;
  const int wshift = ilog2_exact(d->Wo), hshift = ilog2_exact(d->Ho);
  if (wshift < 1 || hshift < 1) return false;
  const long long xbytes = (((long long)d->N * d->Hs * d->Ws - 1) * d->ldx + d->C) * 2;
  if (xbytes >= (1ll << 31) || J >= (1 << 30)) return false;
  if ((e.flags & (SG_EPI_ATOMIC | SG_EPI_OUT_F32)) || (e.ldo & 7) || !aligned16(e.out)) return false;
  if (e.mask && e.res) return false;
  if (e.mask && ((e.ldm & 7) || !aligned16(e.mask))) return false;
  if (e.res && ((e.flags & SG_EPI_RES_F32) || (e.ldr & 7) || !aligned16(e.res))) return false;
  if (J < 16384 && !force) return false;
  // output / mask / residual go through buffer descriptors too (32-bit offsets, bit 31 = "no access")
  const bool pooled = (e.flags & SG_EPI_POOL) != 0;
  const long long jout = pooled ? (J >> 2) : J;
  const long long obytes = ((jout - 1) * e.ldo + I) * 2;
  const long long sbytes = e.mask ? ((jout - 1) * e.ldm + I) * 2 : (e.res ? ((jout - 1) * e.ldr + I) * 2 : 16);
  if (obytes >= (1ll << 31) || sbytes >= (1ll << 31)) return false;
  ConvSkParams p;
  p.obytes = (unsigned)obytes; p.sbytes = (unsigned)sbytes;
  p.x = (const bf16_t*)d->x; p.w = (const bf16_t*)d->w;
  p.C = d->C; p.ldx = d->ldx; p.Hs = d->Hs; p.Ws = d->Ws;
  p.Ho = d->Ho; p.Wo = d->Wo; p.wshift = wshift; p.hshift = hshift;
  p.mode3 = stem ? 1 : 0; p.flags = pflags;
  p.I = I; p.J = J; p.K = K; p.xbytes = (unsigned)xbytes; p.nrb = 0;
  return sg_launch_conv_sk(p, e, st) == 0;
}

template <typename T> static int conv_fwd_t(const sg_conv_fwd_desc* d, hipStream_t st) {
  const int K = d->R * d->S * d->C;
  const int I = d->Cout;
  const long long Jll = (long long)d->N * d->Ho * d->Wo;
  SG_CHECK(Jll < (1ll << 31), "sg_conv2d_fwd: too many output pixels");
  SG_CHECK((long long)d->N * d->Hs * d->Ws * d->ldx < (1ll << 31), "sg_conv2d_fwd: input tensor too large for 32-bit element offsets");
  const int J = (int)Jll;
  int pflags = d->pix_flags;
  if (d->epi_flags & SG_EPI_POOL) {
    SG_CHECK((d->Ho % 2 == 0) && (d->Wo % 2 == 0), "sg_conv2d_fwd: pooled output needs even Ho, Wo");
    pflags |= SG_PIX_QUAD;
  } else {
    pflags &= ~SG_PIX_QUAD;
  }
  const bool w_vec = (K % ET<T>::VEC == 0) && aligned16(d->w);
  const bool x_vec = (d->C % ET<T>::VEC == 0) && (d->ldx % ET<T>::VEC == 0) && aligned16(d->x);
  Epilogue<T> e;
  e.out = d->out; e.out_bstride = 0; e.ldo = d->ldo; e.bias = d->bias;
  e.res = d->res; e.res_bstride = 0; e.ldr = d->ldr; e.beta = d->beta;
  e.mask = (const T*)d->mask; e.mask_bstride = 0; e.ldm = d->ldm; e.split_stride = 0;
  e.alpha = d->alpha; e.alpha_ptr = d->alpha_ptr; e.flags = d->epi_flags; e.I = I; e.J = J;
  const int prof = sg_prof_begin(st, 2.0 * (double)I * (double)J * (double)K, 0);
  if (w_vec && x_vec && conv_fwd_sk_try<T>(d, e, I, J, K, pflags, st)) {}
  else if (w_vec && x_vec && conv_fwd_v3_try<T>(d, e, I, J, K, pflags, st)) {}
  else if (w_vec && x_vec && conv_fwd_v2_try<T>(d, e, I, J, K, pflags, st)) {}
  else if (w_vec && x_vec) conv_fwd_launch<T, true>(d, e, I, J, K, pflags, st);   // all-vector kernels: no gather code in the k-loop
  else conv_fwd_launch<T, false>(d, e, I, J, K, pflags, st);
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

// tile configuration and split-K plan of the weight gradient (shared by the launcher and sg_conv2d_wgrad_plan)
static bool wgrad_v2_ok(const sg_conv_wgrad_desc* d) {
  const char* mode = getenv("SG_CONV_V2");
  if (mode && mode[0] == '0') return false;
  const bool force = mode && mode[0] == 'f';
  if (d->dtype != SG_DTYPE_BF16 || d->stride != 1 || d->no_tr) return false;
  if ((d->x_flags | d->g_flags) & SG_PIX_TRANSPOSED) return false;
  if (ilog2_exact(d->Ho) < 0 || ilog2_exact(d->Wo) < 0) return false;
  if (d->C % 8 || d->ldx % 8 || d->Cout % 8 || d->ldg % 8 || !aligned16(d->x) || !aligned16(d->dy)) return false;
  const long long K = (long long)d->N * d->Ho * d->Wo;
  const int I = d->R * d->S * d->C;
  // buffer-descriptor DMA: bit 31 of a byte offset must be out of range
  if ((long long)d->N * d->xHs * d->xWs * d->ldx * 2 >= (1ll << 31) || (long long)d->N * d->gHs * d->gWs * d->ldg * 2 >= (1ll << 31)) return false;
  if (!force && (I < 64 || d->Cout < 64 || K < 4096)) return false;   // narrow I (1x1 convs, the 8-channel RGB stem) wastes part of the
                                                                       // 256-row tile but still beats the generic kernel 3-4x
  return true;
}

static void wgrad_plan(int I, int J, int K, int bk, int want_splits, int& BI, int& BJ, int& splits, bool v2 = false) {
  if (v2) { BI = 256; BJ = sg_wgrad_v2_bj(I, J, K); bk = 64; }
  else if (I <= 32) { BI = 32; BJ = 256; }
  else if (J <= 32) { BI = 256; BJ = 32; }
  else if (J % 128 != 0 && (J % 96 == 0 || (J < 128 && J > 64))) { BI = 256; BJ = 96; }
  else { BI = 128; BJ = 128; }
  const int tiles = ((I + BI - 1) / BI) * ((J + BJ - 1) / BJ);
  splits = want_splits;
  if (splits <= 0) {
    splits = (768 + tiles - 1) / tiles;          // ~3 workgroups per CU
    int maxs = K / (bk * 16); if (maxs < 1) maxs = 1;
    if (splits > maxs) splits = maxs;
    if (splits > 512) splits = 512;
  }
  if (splits > 1) {  // what sg_launch_gemm will really use after rounding klen up to a multiple of bk
    int klen = (K + splits - 1) / splits; klen = ((klen + bk - 1) / bk) * bk;
    splits = (K + klen - 1) / klen;
  }
}

extern "C" int sg_conv2d_wgrad_plan(const sg_conv_wgrad_desc* d, int* splits, long long* work_floats) {
  SG_CHECK(d && splits && work_floats, "sg_conv2d_wgrad_plan: null");
  const int I = d->R * d->S * d->C, J = d->Cout;
  const long long K = (long long)d->N * d->Ho * d->Wo;
  int BI, BJ, sp;
  wgrad_plan(I, J, (int)K, d->dtype == SG_DTYPE_BF16 ? 32 : 16, d->splits, BI, BJ, sp, wgrad_v2_ok(d));
  *splits = sp;
  *work_floats = sp > 1 ? (long long)sp * I * J : 0;
  return 0;
}

// out[i] += sum_s partial[s][i]   (fixed summation order: deterministic weight gradients)
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* partial, float* out, int splits, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float acc = 0.f;
    for (int s = 0; s < splits; s++) acc += partial[(long long)s * n + i];
    out[i] += acc;
  }
}

template <typename T, bool TR, bool FAST>
static void conv_wgrad_launch(const sg_conv_wgrad_desc* d, const Epilogue<T>& e, int I, int J, int K, int BI, int BJ, int splits, hipStream_t st) {
  typedef ConvPixMC<T, FAST> LM;
  LM lp;
  fill_geom<T>(lp.g, d->x, d->N, d->xHs, d->xWs, d->C, d->ldx, d->Ho, d->Wo, d->R, d->S, d->stride, d->pad_h, d->pad_w,
               d->x_flags & ~SG_PIX_QUAD);
  lp.rows = I; lp.K = K;
  LM lq;
  fill_geom<T>(lq.g, d->dy, d->N, d->gHs, d->gWs, d->Cout, d->ldg, d->Ho, d->Wo, 1, 1, 1, 0, 0, d->g_flags & ~SG_PIX_QUAD);
  lq.rows = J; lq.K = K;
  if (BI == 32) sg_launch_gemm<T, LM, LM, 32, 256, 1, 4, TR>(lp, lq, e, I, J, K, splits, 1, st);
  else if (BJ == 32) sg_launch_gemm<T, LM, LM, 256, 32, 4, 1, TR>(lp, lq, e, I, J, K, splits, 1, st);
  else if (BJ == 96) sg_launch_gemm<T, LM, LM, 256, 96, 4, 1, TR>(lp, lq, e, I, J, K, splits, 1, st);
  else sg_launch_gemm<T, LM, LM, 128, 128, 2, 2, TR>(lp, lq, e, I, J, K, splits, 1, st);
}

template <typename T> static bool wgrad_v2_launch(const sg_conv_wgrad_desc*, const Epilogue<T>&, int, int, int, int, hipStream_t) { return false; }
template <> bool wgrad_v2_launch<bf16_t>(const sg_conv_wgrad_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int splits, hipStream_t st) {
  WgradV2Params p;
  p.x = (const bf16_t*)d->x; p.dy = (const bf16_t*)d->dy;
  p.xHs = d->xHs; p.xWs = d->xWs; p.C = d->C; p.ldx = d->ldx;
  p.x_up = (d->x_flags & SG_PIX_UPSAMPLE) ? 1 : 0; p.x_relu = (d->x_flags & SG_PIX_RELU) ? 1 : 0;
  p.Hin = d->xHs * (p.x_up ? 2 : 1); p.Win = d->xWs * (p.x_up ? 2 : 1);
  p.gHs = d->gHs; p.gWs = d->gWs; p.Cout = d->Cout; p.ldg = d->ldg; p.g_up = (d->g_flags & SG_PIX_UPSAMPLE) ? 1 : 0;
  p.Ho = d->Ho; p.Wo = d->Wo; p.wshift = ilog2_exact(d->Wo); p.hshift = ilog2_exact(d->Ho);
  p.R = d->R; p.S = d->S; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
  p.I = I; p.J = J; p.K = K;
  p.xbytes = (unsigned)((((long long)d->N * d->xHs * d->xWs - 1) * d->ldx + d->C) * 2);
  p.gbytes = (unsigned)((((long long)d->N * d->gHs * d->gWs - 1) * d->ldg + d->Cout) * 2);
  int klen = K;
  if (splits > 1) { klen = (K + splits - 1) / splits; klen = ((klen + 63) / 64) * 64; splits = (K + klen - 1) / klen; }
  p.klen = klen;
  return sg_launch_wgrad_v2(p, e, splits, st) == 0;
}

template <typename T, bool TR> static int conv_wgrad_t(const sg_conv_wgrad_desc* d, hipStream_t st) {
  const int I = d->R * d->S * d->C;
  const int J = d->Cout;
  const long long Kll = (long long)d->N * d->Ho * d->Wo;
  SG_CHECK(Kll < (1ll << 31), "sg_conv2d_wgrad: too many pixels");
  SG_CHECK((long long)d->N * d->xHs * d->xWs * d->ldx < (1ll << 31) && (long long)d->N * d->gHs * d->gWs * d->ldg < (1ll << 31),
           "sg_conv2d_wgrad: tensor too large for 32-bit element offsets");
  const int K = (int)Kll;
  int BI, BJ, splits;
  const bool v2 = wgrad_v2_ok(d);
  wgrad_plan(I, J, K, ET<T>::BK, d->splits, BI, BJ, splits, v2);
  const long long n = (long long)I * J;
  const bool two_stage = splits > 1 && d->work && d->work_floats >= (long long)splits * n;
  Epilogue<T> e;
  e.out = d->dw; e.out_bstride = 0; e.ldo = I; e.bias = nullptr; e.res = nullptr; e.res_bstride = 0; e.ldr = 0; e.beta = 0.f;
  e.mask = nullptr; e.mask_bstride = 0; e.ldm = 0; e.alpha = d->alpha; e.alpha_ptr = d->alpha_ptr; e.split_stride = 0;
  e.flags = SG_EPI_OUT_F32; e.I = I; e.J = J;
  if (splits == 1) { e.res = d->dw; e.ldr = I; e.beta = 1.f; e.flags |= SG_EPI_RES_F32; }   // single writer: dw += tile, no atomics
  else if (two_stage) { e.out = d->work; e.split_stride = n; }                                 // partial tiles, reduced below
  else e.flags |= SG_EPI_ATOMIC;                                                               // no workspace: fp32 atomics
  const int prof = sg_prof_begin(st, 2.0 * (double)I * (double)J * (double)K, 1);
  const bool fast = (d->C % ET<T>::VEC == 0) && (d->ldx % ET<T>::VEC == 0) && aligned16(d->x) &&
                    (d->Cout % ET<T>::VEC == 0) && (d->ldg % ET<T>::VEC == 0) && aligned16(d->dy);
  if (v2 && wgrad_v2_launch<T>(d, e, I, J, K, splits, st)) {}
  else if (fast) conv_wgrad_launch<T, TR, true>(d, e, I, J, K, BI, BJ, splits, st);
  else conv_wgrad_launch<T, TR, false>(d, e, I, J, K, BI, BJ, splits, st);
  if (two_stage) {
    long long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((int)blocks), dim3(256), 0, st, (const float*)d->work, d->dw, splits, n);
  }
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
