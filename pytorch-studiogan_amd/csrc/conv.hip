// conv.hip -- convolution forward / data-gradient / weight-gradient as implicit GEMM on the MFMA engine
// (gemm_core.h). Replaces nn.Conv2d + autograd's convolution_backward on the StudioGAN hot path
// (reference src/utils/ops.py:165-173,195-204; call sites models/big_resnet.py:28-42,177-242).
#include "conv_common.h"
#include "conv_v2.h"

// fp32 arithmetic mode of the generic engine's convolutions -- forward, data gradient (here) and weight gradient (conv_wgrad.hip) -- (sg_set_f32_mode): 0 = exact fp32 MFMA, 3 = bf16x3 split (gemm_core.h SPLIT)
int g_sg_f32_mode = 0;
extern "C" int sg_set_f32_mode(int mode) {
  SG_CHECK(mode == 0 || mode == 3, "sg_set_f32_mode: 0 (exact fp32 MFMA) or 3 (bf16x3 split)");
  g_sg_f32_mode = mode;
  return 0;
}
extern "C" int sg_get_f32_mode() { return g_sg_f32_mode; }

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
  if constexpr (sizeof(T) == 4 && FAST) {
    if (g_sg_f32_mode == 3) {      // fp32 in / fp32 out, three bf16 MFMAs per k-tile (all-vector operands only: the Inception stack, the fp32 backbones' aligned layers)
      if (I <= 32) sg_launch_gemm<T, LP, LQ, 32, 256, 1, 4, true, 3>(lp, lq, e, I, J, K, 1, 1, st);
      else if (I % 128 != 0 && (I % 96 == 0 || (I < 128 && I > 64))) sg_launch_gemm<T, LP, LQ, 96, 256, 1, 4, true, 3>(lp, lq, e, I, J, K, 1, 1, st);
      else sg_launch_gemm<T, LP, LQ, 128, 128, 2, 2, true, 3>(lp, lq, e, I, J, K, 1, 1, st);
      return;
    }
  }
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
  if (disabled || (pflags & SG_PIX_TRANSPOSED)) return false;
  // stride 2 (round 3: InceptionV3's reduction layers ran on the generic engine): plain row order, no upsample-on-load
  if (d->stride != 1 && (d->stride != 2 || (pflags & (SG_PIX_UPSAMPLE | SG_PIX_QUAD)) || getenv("SG_CONV_V2_STRIDE2_OFF"))) return false;
  if (d->C % 8 || d->ldx % 8 || d->R * d->S > 25 || J < 256) return false;
  // descriptor extents: offsets with bit 31 (activations) / bit 30 (weights) set must be out of range
  const long long xbytes = (((long long)d->N * d->Hs * d->Ws - 1) * d->ldx + d->C) * 2, wbytes = (long long)I * K * 2;
  if (xbytes >= (1ll << 31) || wbytes >= (1ll << 30)) return false;
  if (!aligned16(d->x) || !aligned16(d->w)) return false;
  // the kernel's only epilogue: bf16 rows, 16-byte stores; its ReLU-mask OR residual tile (bf16) is pre-staged with 16-byte loads
  if ((e.flags & (SG_EPI_ATOMIC | SG_EPI_OUT_F32)) || (e.ldo & 7) || !aligned16(e.out)) return false;
  // (mask AND residual together: sg_conv_epilogue condenses the mask tile to register bits, then stages the residual tile)
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
  if (!best && I % 8 == 0 && I >= 32) {
    // no candidate divides the cout count (InceptionV3: 32 / 48 / 64 / 160 / 224 / 320 / 448 couts): take the one that pads least -- the kernel
    // zero-fills weight rows >= I and its epilogue stores only the 16-byte chunks that exist. Up to 50 % padding still beats the generic
    // engine these layers ran on (17 % of the FID leg at ~150 us per launch, profiles/r02_fid_leg_kerneltrace.txt).
    int best_pad = 0;
    for (int c = 0; c < 3; c++) {
      const int padded = ((I + cands[c] - 1) / cands[c]) * cands[c];
      if (2 * padded > 3 * I) continue;
      const int tiles = (padded / cands[c]) * tj;
      if (!best || padded < best_pad) { best = cands[c]; best_pad = padded; best_tiles = tiles; }
    }
  }
  // tile-count floor: below it the chip is too empty for this kernel's one workgroup per CU. SG_CONV_V2_MIN_TILES=<n> moves it (A/B switch: 128-cout
  // layers at 17 x 17 have 145 tiles and run on the generic engine at ~59 us per launch in the FID leg, profiles/r04_fid_leg_kerneltrace.txt)
  int min_tiles = 160;
  if (const char* mt = getenv("SG_CONV_V2_MIN_TILES")) { const int v = atoi(mt); if (v > 0) min_tiles = v; }
  if (!best || (best_tiles < min_tiles && !force)) return false;
  ConvV2Params p;
  p.x = (const bf16_t*)d->x; p.w = (const bf16_t*)d->w;
  p.N = d->N; p.Hs = d->Hs; p.Ws = d->Ws; p.C = d->C; p.ldx = d->ldx;
  const int up = (pflags & SG_PIX_UPSAMPLE) ? 2 : 1;
  p.Hin = d->Hs * up; p.Win = d->Ws * up; p.Ho = d->Ho; p.Wo = d->Wo;
  p.R = d->R; p.S = d->S; p.pad_h = d->pad_h; p.pad_w = d->pad_w; p.flags = pflags; p.stride = d->stride;
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

template <typename T> static bool conv_fwd_v3_try(const sg_conv_fwd_desc*, const Epilogue<T>&, int, int, int, int, hipStream_t) { return false; }
template <> bool conv_fwd_v3_try<bf16_t>(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st) {
  return sg_conv_fwd_v3_try(d, e, I, J, K, pflags, st);      // conv_v3.hip
}
template <typename T> static bool conv_fwd_v4_try(const sg_conv_fwd_desc*, const Epilogue<T>&, int, int, int, int, hipStream_t) { return false; }
template <> bool conv_fwd_v4_try<bf16_t>(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st) {
  return sg_conv_fwd_v4_try(d, e, I, J, K, pflags, st);      // conv_v4.hip
}
template <typename T> static bool conv_fwd_rs_try(const sg_conv_fwd_desc*, const Epilogue<T>&, int, int, int, int, hipStream_t) { return false; }
template <> bool conv_fwd_rs_try<bf16_t>(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st) {
  return sg_conv_fwd_rs_try(d, e, I, J, K, pflags, st);      // conv_rs.hip
}
template <typename T> static bool conv_fwd_sk_try(const sg_conv_fwd_desc*, const Epilogue<T>&, int, int, int, int, hipStream_t) { return false; }
template <> bool conv_fwd_sk_try<bf16_t>(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st) {
  return sg_conv_fwd_sk_try(d, e, I, J, K, pflags, st);      // conv_sk.hip
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
  int eng = SG_ENG_CONV_GEMM;
  if (w_vec && x_vec && conv_fwd_sk_try<T>(d, e, I, J, K, pflags, st)) eng = SG_ENG_CONV_SK;
  else if (w_vec && x_vec && conv_fwd_rs_try<T>(d, e, I, J, K, pflags, st)) eng = SG_ENG_CONV_RS;
  else if (w_vec && x_vec && conv_fwd_v4_try<T>(d, e, I, J, K, pflags, st)) eng = SG_ENG_CONV_V4;
  else if (w_vec && x_vec && conv_fwd_v3_try<T>(d, e, I, J, K, pflags, st)) eng = SG_ENG_CONV_V3;
  else if (w_vec && x_vec && conv_fwd_v2_try<T>(d, e, I, J, K, pflags, st)) eng = SG_ENG_CONV_V2;
  else if (w_vec && x_vec) conv_fwd_launch<T, true>(d, e, I, J, K, pflags, st);   // all-vector kernels: no gather code in the k-loop
  else conv_fwd_launch<T, false>(d, e, I, J, K, pflags, st);
  {
    // algorithmic HBM bytes: input, filter, result (pooled size when pooling), ReLU-mask / residual operands once each
    const double es = sizeof(T), jout = (d->epi_flags & SG_EPI_POOL) ? (double)J / 4.0 : (double)J;
    const double b = es * ((double)d->N * d->Hs * d->Ws * d->C + (double)I * K + jout * I * (1.0 + (d->mask ? 1.0 : 0.0) + (d->res ? 1.0 : 0.0)));
    sg_prof_tag(prof, eng, b);
  }
  sg_prof_end(st, prof);
  SG_LAUNCH_CHECK();
  return 0;
}

// fused 3x3 + 1x1-skip launch (conv_v4.h SKIP). dry: eligibility only.
static int conv_fwd_skip(const sg_conv_skip_desc* sk, hipStream_t st, bool dry) {
  const sg_conv_fwd_desc* d = &sk->main;
  if (d->dtype != SG_DTYPE_BF16 || d->R != 3 || d->S != 3 || d->stride != 1 || (d->pix_flags & (SG_PIX_UPSAMPLE | SG_PIX_TRANSPOSED))) return 0;
  const int K = 9 * d->C, I = d->Cout;
  const long long Jll = (long long)d->N * d->Ho * d->Wo;
  if (Jll >= (1ll << 31) || (long long)d->N * d->Hs * d->Ws * d->ldx >= (1ll << 31)) return 0;
  const int J = (int)Jll;
  int pflags = d->pix_flags;
  if (d->epi_flags & SG_EPI_POOL) { if ((d->Ho & 1) || (d->Wo & 1)) return 0; pflags |= SG_PIX_QUAD; } else pflags &= ~SG_PIX_QUAD;
  Epilogue<bf16_t> e;
  e.out = d->out; e.out_bstride = 0; e.ldo = d->ldo; e.bias = d->bias;
  e.res = d->res; e.res_bstride = 0; e.ldr = d->ldr; e.beta = d->beta;
  e.mask = (const bf16_t*)d->mask; e.mask_bstride = 0; e.ldm = d->ldm; e.split_stride = 0;
  e.alpha = d->alpha; e.alpha_ptr = d->alpha_ptr; e.flags = d->epi_flags; e.I = I; e.J = J;
  if (dry) return sg_conv_fwd_v4_skip_try(d, sk, e, I, J, K, pflags, st, true) ? 1 : 0;
  const int prof = sg_prof_begin(st, 2.0 * (double)I * (double)J * ((double)K + (double)sk->C2), 0);
  const bool ok = sg_conv_fwd_v4_skip_try(d, sk, e, I, J, K, pflags, st, false);
  {
    const double jout = (d->epi_flags & SG_EPI_POOL) ? (double)J / 4.0 : (double)J;
    const double x2 = (double)d->N * (sk->x2_up ? (d->Ho / 2) * (d->Wo / 2) : d->Ho * d->Wo) * sk->C2;
    sg_prof_tag(prof, SG_ENG_CONV_V4_SKIP, 2.0 * ((double)d->N * d->Hs * d->Ws * d->C + x2 + (double)I * (K + sk->C2) + jout * I));
  }
  sg_prof_end(st, prof);
  return ok ? 1 : 0;
}
// rows of the per-tile statistics buffer a fused launch with d->stats writes ([rows][Cout][2] floats): 256-pixel tiles
extern "C" int sg_conv2d_fwd_skip_stat_rows(const sg_conv_skip_desc* d) {
  if (!d) return 0;
  const long long J = (long long)d->main.N * d->main.Ho * d->main.Wo;
  return (int)((J + 255) / 256);
}
extern "C" int sg_conv2d_fwd_skip_ok(const sg_conv_skip_desc* d) {
  if (!d || !d->main.x || !d->main.w || !d->main.out || !d->x2 || !d->w2) return 0;
  return conv_fwd_skip(d, nullptr, true);
}
extern "C" int sg_conv2d_fwd_skip(const sg_conv_skip_desc* d, sg_stream_t stream) {
  SG_CHECK(d && d->main.x && d->main.w && d->main.out && d->x2 && d->w2, "sg_conv2d_fwd_skip: null pointer");
  SG_CHECK(conv_fwd_skip(d, (hipStream_t)stream, false) == 1, "sg_conv2d_fwd_skip: problem not eligible for the fused kernel (ask sg_conv2d_fwd_skip_ok first)");
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

