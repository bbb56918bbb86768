// conv_wgrad.hip -- convolution weight gradient (sg_conv2d_wgrad, sg_conv2d_wgrad_plan): wgrad_v2.h for the hot bf16 shapes,
// gemm_core.h's ConvPixMC x ConvPixMC contraction otherwise; deterministic two-stage split-K.
// Replaces autograd's convolution_backward (weight part) on the StudioGAN hot path (reference src/utils/ops.py:165-173,195-204).
#include "conv_common.h"
#include "wgrad_v2.h"
#include "wgrad_sk.h"
#include "wgrad_v3.h"
#include "wgrad_v3l.h"

// ---- thin layers: streaming kernel (wgrad_sk.h). SG_WGRAD_SK=0 disables it. -----------------------------------------------------
struct SkPlan { bool ok, taps; int NI, NJ, swap, nw; long long n; };
static SkPlan wgrad_sk_plan(const sg_conv_wgrad_desc* d) {
  SkPlan s; s.ok = false; s.taps = false; s.NI = s.NJ = s.swap = s.nw = 0; s.n = 0;
  const char* mode = getenv("SG_WGRAD_SK");
  if (mode && mode[0] == '0') return s;
  if (d->dtype != SG_DTYPE_BF16 || d->stride != 1 || d->no_tr) return s;
  if ((d->x_flags | d->g_flags) & SG_PIX_TRANSPOSED) return s;
  const int wsh = ilog2_exact(d->Wo), hsh = ilog2_exact(d->Ho);
  if (wsh < 0 || hsh < 0) return s;
  if (d->C % 8 || d->ldx % 8 || d->Cout % 8 || d->ldg % 8 || !aligned16(d->x) || !aligned16(d->dy)) return s;
  const long long K = (long long)d->N * d->Ho * d->Wo;
  if (K % 32 || K / 32 >= (1ll << 30)) return s;
  if ((long long)d->N * d->xHs * d->xWs * d->ldx * 2 >= (1ll << 31) || (long long)d->N * d->gHs * d->gWs * d->ldg * 2 >= (1ll << 31)) return s;
  const bool xup = d->x_flags & SG_PIX_UPSAMPLE, gup = d->g_flags & SG_PIX_UPSAMPLE, xrelu = d->x_flags & SG_PIX_RELU;
  if (d->R == 1 && d->S == 1 && d->pad_h == 0 && d->pad_w == 0) {
    if (d->C > 192 || d->Cout > 96) return s;
    const int ni = (d->C + 31) / 32, nj = (d->Cout + 31) / 32;
    if (ni == 4 || ni == 5) return s;                       // (no layer of the networks has 97..160 input channels on a thin 1x1)
    s.ok = true; s.NI = ni; s.NJ = nj; s.n = (long long)d->C * d->Cout;
  } else if (d->R == 3 && d->S == 3 && d->pad_h == 1 && d->pad_w == 1 && !xup && !gup && !xrelu && d->Wo >= 32) {
    if (d->C == 8 && d->Cout <= 96) { s.ok = true; s.taps = true; s.NI = 3; s.NJ = (d->Cout + 31) / 32; s.n = 72ll * d->Cout; }
    else if (d->Cout == 8 && d->C <= 96) { s.ok = true; s.taps = true; s.swap = 1; s.NI = 3; s.NJ = (d->C + 31) / 32; s.n = 72ll * d->C; }
  }
  if (s.ok) s.nw = sg_wgrad_sk_waves((int)(K / 32));
  return s;
}
static int wgrad_sk_launch(const sg_conv_wgrad_desc* d, const SkPlan& s, hipStream_t st) {
  WgradSkParams p;
  const long long K = (long long)d->N * d->Ho * d->Wo;
  const unsigned xbytes = (unsigned)((((long long)d->N * d->xHs * d->xWs - 1) * d->ldx + d->C) * 2);
  const unsigned gbytes = (unsigned)((((long long)d->N * d->gHs * d->gWs - 1) * d->ldg + d->Cout) * 2);
  p.H = d->Ho; p.W = d->Wo; p.wshift = ilog2_exact(d->Wo); p.hshift = ilog2_exact(d->Ho);
  p.nchunk = (int)(K / 32);
  p.arelu = (d->x_flags & SG_PIX_RELU) ? 1 : 0;
  p.work = (float*)d->work; p.n = s.n;
  p.alpha = d->alpha; p.alpha_ptr = d->alpha_ptr;
  const int xup = (d->x_flags & SG_PIX_UPSAMPLE) ? 1 : 0, gup = (d->g_flags & SG_PIX_UPSAMPLE) ? 1 : 0;
  if (!s.swap) {       // A = x, B = dy
    p.a = (const bf16_t*)d->x; p.lda = d->ldx; p.CA = d->C; p.upA = xup; p.HsA = d->xHs; p.WsA = d->xWs; p.abytes = xbytes;
    p.b = (const bf16_t*)d->dy; p.ldb = d->ldg; p.CB = d->Cout; p.upB = gup; p.HsB = d->gHs; p.WsB = d->gWs; p.bbytes = gbytes;
    p.sgn = 1; p.swap = 0; p.I = s.taps ? 72 : d->C; p.J = d->Cout;
  } else {             // generator RGB layer: A = dy (8 padded couts, shifted by -tap), B = x
    p.a = (const bf16_t*)d->dy; p.lda = d->ldg; p.CA = 8; p.upA = 0; p.HsA = d->gHs; p.WsA = d->gWs; p.abytes = gbytes;
    p.b = (const bf16_t*)d->x; p.ldb = d->ldx; p.CB = d->C; p.upB = 0; p.HsB = d->xHs; p.WsB = d->xWs; p.bbytes = xbytes;
    p.sgn = -1; p.swap = 1; p.I = 72; p.J = d->C;
  }
  return sg_launch_wgrad_sk(p, s.taps, s.NI, s.NJ, st);
}

// ---- wide-image 3x3 layers: halo kernel (wgrad_v3.h). SG_WGRAD_V3=0 disables it. -----------------------------------------------------
struct V3Plan { bool ok; int NB, nci, nco, splits, nchunk; long long n, stride; };   // stride: floats per partial = n (+ Cout when the bias gradient rides along)
static V3Plan wgrad_v3_plan(const sg_conv_wgrad_desc* d) {
  V3Plan s; s.ok = false; s.NB = s.nci = s.nco = s.splits = s.nchunk = 0; s.n = 0; s.stride = 0;
  const char* mode = getenv("SG_WGRAD_V3");
  if (mode && mode[0] == '0') return s;
  const bool force = mode && mode[0] == 'f';                            // test hook: no lower bound on the problem size
  if (d->dtype != SG_DTYPE_BF16 || d->stride != 1 || d->no_tr || d->R != 3 || d->S != 3 || d->pad_h != 1 || d->pad_w != 1) return s;
  if ((d->x_flags | d->g_flags) & SG_PIX_TRANSPOSED) return s;
  if (d->Wo != 4 && d->Wo != 8 && d->Wo != 16 && d->Wo != 32 && d->Wo % 64) return s;
  if (d->Wo == 4) {     // (round 4) 4 x 4 images: a chunk is four whole images, plain operands only
    if (d->Ho != 4 || d->N % 4 || ((d->x_flags | d->g_flags) & SG_PIX_UPSAMPLE)) return s;
  } else if (d->Wo < 64 && d->Ho % (64 / d->Wo)) return s;              // a chunk = 64 / W whole image rows
  if (d->C % 32 || d->ldx % 8 || d->ldg % 8 || !aligned16(d->x) || !aligned16(d->dy)) return s;
  if (d->Cout % 96 == 0) s.NB = 3; else if (d->Cout % 64 == 0) s.NB = 2; else return s;
  const long long K = (long long)d->N * d->Ho * d->Wo;
  if (K % 64 || (!force && K < (d->Wo == 4 ? 4096 : 16384)) || K >= (1ll << 31)) return s;      // (4 x 4 layers: 16 pixels per image, batch 256 = 4096)
  if ((long long)d->N * d->xHs * d->xWs * d->ldx * 2 >= (1ll << 31) || (long long)d->N * d->gHs * d->gWs * d->ldg * 2 >= (1ll << 31)) return s;
  s.nci = d->C / 32; s.nco = d->Cout / (32 * s.NB); s.nchunk = (int)(K / 64);
  s.n = 9ll * d->C * d->Cout;
  s.stride = s.n + (d->dbias ? d->Cout : 0);
  const int tiles = s.nci * s.nco;
  int sp = d->splits > 0 ? d->splits : (768 >= tiles ? 768 / tiles : 1);   // three workgroups per CU fit (registers, LDS): one full wave of 768
  const int maxs = s.nchunk / 8 > 0 ? s.nchunk / 8 : 1;                // at least eight chunks per workgroup
  if (sp > maxs) sp = maxs;
  if (sp > 512) sp = 512;
  s.splits = sp;
  s.ok = true;
  return s;
}
static int wgrad_v3_launch(const sg_conv_wgrad_desc* d, const V3Plan& s, hipStream_t st) {
  WgradV3Params p;
  p.x = (const bf16_t*)d->x; p.dy = (const bf16_t*)d->dy;
  p.xHs = d->xHs; p.xWs = d->xWs; p.ldx = d->ldx; p.x_up = (d->x_flags & SG_PIX_UPSAMPLE) ? 1 : 0; p.x_relu = (d->x_flags & SG_PIX_RELU) ? 1 : 0;
  p.gHs = d->gHs; p.gWs = d->gWs; p.ldg = d->ldg; p.g_up = (d->g_flags & SG_PIX_UPSAMPLE) ? 1 : 0;
  p.N = d->N; p.H = d->Ho; p.W = d->Wo; p.C = d->C; p.Cout = d->Cout;
  p.nci = s.nci; p.nco = s.nco; p.nchunk = s.nchunk; p.splits = s.splits;
  p.xbytes = (unsigned)((((long long)d->N * d->xHs * d->xWs - 1) * d->ldx + d->C) * 2);
  p.gbytes = (unsigned)((((long long)d->N * d->gHs * d->gWs - 1) * d->ldg + d->Cout) * 2);
  p.out = d->work; p.split_stride = s.stride;
  p.bias_off = d->dbias ? s.n : -1; p.bias_scale = p.g_up ? 0.25f : 1.f;
  p.alpha = d->alpha; p.alpha_ptr = d->alpha_ptr;
  // the lean kernel (wgrad_v3l.h; round 5, same box: the wgrad_v3 layers of C3 -8..-12 %, profiles/r05_variant_ab_layer_tables_b.txt); it hands problems whose
  // operands are read through a 2x upsampling to wgrad_v3.h's kernel. SG_WGRAD_V3_LEAN=0 (read per call): that kernel everywhere (the bit-identity reference of
  // tests/test_conv_v2_gpu.py)
  if (const char* m = getenv("SG_WGRAD_V3_LEAN")) { if (m[0] == '0') return sg_launch_wgrad_v3(p, s.NB, st); }
  return sg_launch_wgrad_v3l(p, s.NB, st);
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

extern "C" int sg_conv2d_wgrad_fuses_bias(const sg_conv_wgrad_desc* d) {
  if (!d || !d->dbias) return 0;
  if (wgrad_sk_plan(d).ok) return 0;
  return wgrad_v3_plan(d).ok ? 1 : 0;            // (the launcher also needs the workspace sg_conv2d_wgrad_plan asks for: the caller provides it)
}

extern "C" int sg_conv2d_wgrad_plan(const sg_conv_wgrad_desc* d, int* splits, long long* work_floats) {
  SG_CHECK(d && splits && work_floats, "sg_conv2d_wgrad_plan: null");
  const int I = d->R * d->S * d->C, J = d->Cout;
  const long long K = (long long)d->N * d->Ho * d->Wo;
  const SkPlan sk = wgrad_sk_plan(d);
  if (sk.ok) { *splits = sk.nw; *work_floats = (long long)sk.nw * sk.n; return 0; }
  const V3Plan v3 = wgrad_v3_plan(d);
  if (v3.ok) { *splits = v3.splits; *work_floats = (long long)v3.splits * v3.stride; return 0; }
  int BI, BJ, sp;
  wgrad_plan(I, J, (int)K, d->dtype == SG_DTYPE_BF16 ? 32 : 16, d->splits, BI, BJ, sp, wgrad_v2_ok(d));
  *splits = sp;
  *work_floats = sp > 1 ? (long long)sp * I * J : 0;
  return 0;
}

// out[i] += sum_s partial[s][i]   (fixed summation order: deterministic weight gradients)
// 256 threads = 64 outputs x 4 split phases: phase q sums the splits s = q, q + 4, ... with four loads in flight, the four phase sums are
// combined in a fixed order through LDS. (One thread per output walking all splits -- the first version -- was a chain of up to 1024
// dependent loads on 27-324 workgroups: 106 us average, 11 ms per training step in the r02 trace for a few hundred MB.)
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* partial, float* out, int splits, long long n, long long stride = 0, float* out2 = nullptr, long long n2 = 0) {
  __shared__ float sm[4][64];
  if (stride == 0) stride = n;                    // partial s at partial + s * stride; elements n .. n + n2 - 1 of a partial go to out2 (bias gradient)
  const long long nt = n + n2;
  const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
  for (long long i0 = blockIdx.x * 64ll; i0 < nt; i0 += (long long)gridDim.x * 64) {
    const long long i = i0 + li;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < nt) {
      int s = q;
      for (; s + 12 < splits; s += 16) {
        a0 += partial[(long long)s * stride + i];
        a1 += partial[(long long)(s + 4) * stride + i];
        a2 += partial[(long long)(s + 8) * stride + i];
        a3 += partial[(long long)(s + 12) * stride + i];
      }
      for (; s < splits; s += 4) a0 += partial[(long long)s * stride + i];
    }
    sm[q][li] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (q == 0 && i < nt) {
      const float t = (sm[0][li] + sm[1][li]) + (sm[2][li] + sm[3][li]);
      if (i < n) out[i] += t; else out2[i - n] += t;
    }
    __syncthreads();
  }
}

// 16-byte variant (round 3): a lane owns FOUR consecutive outputs (one 16-byte load per split and lane, 1 KB per wave instruction instead of
// 256 B), 256 threads = 64 lanes x 4 split phases, eight loads in flight per lane. The partial tiles of one layer are 85 MB however the layer
// is shaped (768 workgroups x a 288 x 96 fp32 tile), read once: the scalar kernel above moved them at 1.8 TB/s (48 us per launch, 104 launches,
// 5 ms per C3 step in profiles/r03_bench_biggan128_bs256_kerneltrace_a.txt). Same fixed summation order per output -> deterministic.
__global__ __launch_bounds__(256) void k_splitk_reduce_v4(const float* partial, float* out, int splits, long long n, long long stride, float* out2, long long n2) {
  __shared__ f32x4 sm[4][64];
  const long long nt4 = (n + n2) >> 2, n4 = n >> 2;
  const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
  for (long long i0 = blockIdx.x * 64ll; i0 < nt4; i0 += (long long)gridDim.x * 64) {
    const long long i = i0 + li;
    f32x4 a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (i < nt4) {
      const float* src = partial + 4 * i;
      int s = q;
      for (; s + 28 < splits; s += 32) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = *(const f32x4*)(src + (long long)(s + 4 * k) * stride);
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] += v[k];
      }
      for (; s < splits; s += 4) a[0] += *(const f32x4*)(src + (long long)s * stride);
    }
    sm[q][li] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (q == 0 && i < nt4) {
      const f32x4 t = (sm[0][li] + sm[1][li]) + (sm[2][li] + sm[3][li]);
      float* dst = i < n4 ? out + 4 * i : out2 + 4 * (i - n4);
      f32x4 o = *(f32x4*)dst;
      *(f32x4*)dst = o + t;
    }
    __syncthreads();
  }
}
static void splitk_reduce_launch(const float* partial, float* out, int splits, long long n, long long stride, float* out2, long long n2, hipStream_t st) {
  if (stride == 0) stride = n;
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("SG_REDUCE_V4"); mode = (e && e[0] == '0') ? 0 : 1; }
  const bool v4 = mode && (n % 4 == 0) && (n2 % 4 == 0) && (stride % 4 == 0) && ((((uintptr_t)partial | (uintptr_t)out | (uintptr_t)out2) & 15) == 0);
  if (v4) {
    long long blocks = ((n + n2) / 4 + 63) / 64; if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_splitk_reduce_v4, dim3((int)blocks), dim3(256), 0, st, partial, out, splits, n, stride, out2, n2);
  } else {
    long long blocks = (n + n2 + 63) / 64; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((int)blocks), dim3(256), 0, st, partial, out, splits, n, stride, out2, n2);
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
  if constexpr (sizeof(T) == 4 && FAST) {
    if (g_sg_f32_mode == 3) {      // bf16x3 split (gemm_core.h SPLIT): fp32 operands, three bf16 MFMAs per 16 pixels of the reduction
      if (BI == 32) sg_launch_gemm<T, LM, LM, 32, 256, 1, 4, TR, 3>(lp, lq, e, I, J, K, splits, 1, st);
      else if (BJ == 32) sg_launch_gemm<T, LM, LM, 256, 32, 4, 1, TR, 3>(lp, lq, e, I, J, K, splits, 1, st);
      else if (BJ == 96) sg_launch_gemm<T, LM, LM, 256, 96, 4, 1, TR, 3>(lp, lq, e, I, J, K, splits, 1, st);
      else sg_launch_gemm<T, LM, LM, 128, 128, 2, 2, TR, 3>(lp, lq, e, I, J, K, splits, 1, st);
      return;
    }
  }
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

// algorithmic HBM bytes of a weight gradient: x and dy once (stored sizes), the fp32 gradient read and written
template <typename T> static double wgrad_alg_bytes(const sg_conv_wgrad_desc* d) {
  return sizeof(T) * ((double)d->N * d->xHs * d->xWs * d->C + (double)d->N * d->gHs * d->gWs * d->Cout) + 8.0 * (double)d->R * d->S * d->C * d->Cout;
}
template <typename T, bool TR> static int conv_wgrad_t(const sg_conv_wgrad_desc* d, hipStream_t st) {
  const int I = d->R * d->S * d->C;
  const int J = d->Cout;
  const long long Kll = (long long)d->N * d->Ho * d->Wo;
  SG_CHECK(Kll < (1ll << 31), "sg_conv2d_wgrad: too many pixels");
  SG_CHECK((long long)d->N * d->xHs * d->xWs * d->ldx < (1ll << 31) && (long long)d->N * d->gHs * d->gWs * d->ldg < (1ll << 31),
           "sg_conv2d_wgrad: tensor too large for 32-bit element offsets");
  const int K = (int)Kll;
  {
    const SkPlan sk = wgrad_sk_plan(d);
    if (sk.ok && d->work && d->work_floats >= (long long)sk.nw * sk.n) {
      const int prof = sg_prof_begin(st, 2.0 * (double)I * (double)J * (double)K, 1);
      if (wgrad_sk_launch(d, sk, st) != 0) { sg_set_error("sg_conv2d_wgrad: streaming kernel launch failed"); return -2; }
      splitk_reduce_launch((const float*)d->work, d->dw, sk.nw, sk.n, 0, nullptr, 0, st);
      sg_prof_tag(prof, SG_ENG_WGRAD_SK, wgrad_alg_bytes<T>(d));
      sg_prof_end(st, prof);
      SG_LAUNCH_CHECK();
      return 0;
    }
  }
  {
    const V3Plan v3 = wgrad_v3_plan(d);
    if (v3.ok && d->work && d->work_floats >= (long long)v3.splits * v3.stride) {
      const int prof = sg_prof_begin(st, 2.0 * (double)I * (double)J * (double)K, 1);
      if (wgrad_v3_launch(d, v3, st) != 0) { sg_set_error("sg_conv2d_wgrad: halo kernel launch failed"); return -2; }
      splitk_reduce_launch((const float*)d->work, d->dw, v3.splits, v3.n, v3.stride, d->dbias, (long long)(d->dbias ? d->Cout : 0), st);
      sg_prof_tag(prof, SG_ENG_WGRAD_V3, wgrad_alg_bytes<T>(d));
      sg_prof_end(st, prof);
      SG_LAUNCH_CHECK();
      return 0;
    }
  }
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
  int eng = SG_ENG_WGRAD_GEMM;
  if (v2 && wgrad_v2_launch<T>(d, e, I, J, K, splits, st)) eng = SG_ENG_WGRAD_V2;
  else if (fast) conv_wgrad_launch<T, TR, true>(d, e, I, J, K, BI, BJ, splits, st);
  else conv_wgrad_launch<T, TR, false>(d, e, I, J, K, BI, BJ, splits, st);
  if (two_stage) {
    splitk_reduce_launch((const float*)d->work, d->dw, splits, n, 0, nullptr, 0, st);
  }
  sg_prof_tag(prof, eng, wgrad_alg_bytes<T>(d));
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
