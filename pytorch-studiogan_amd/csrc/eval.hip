// eval.hip -- FID/IS feature-extraction side kernels: the on-device replacement of the reference's
// quantize -> D2H -> per-image CPU resize -> H2D -> normalise loop (reference src/utils/ops.py:251-263,
// src/utils/resize.py:72-93 "legacy" resizer, src/metrics/preparation.py:103-108), the Inception pooling layers
// (reference src/metrics/inception_net.py:135-249) and the fp64 moment accumulation that replaces gathering
// 50k x 2048 features (reference src/metrics/fid.py:65-98).
#include "common.h"
#include "../../include/sgamd.h"

#define DISPATCH_T(dtype, ...)                                          \
  if ((dtype) == SG_DTYPE_F32) { typedef float T; __VA_ARGS__; }        \
  else if ((dtype) == SG_DTYPE_BF16) { typedef bf16_t T; __VA_ARGS__; } \
  else { sg_set_error("bad dtype"); return -1; }

static inline int grid1d(long long total) { long long b = (total + 255) / 256; if (b > 256 * 32) b = 256 * 32; if (b < 1) b = 1; return (int)b; }

// uint8 quantisation exactly as ops.quantize_images: x=(x+1)/2 ; (255*x+0.5).clamp(0,255) ; astype(uint8) (truncation).
// Individually rounded fp32 ops (no fma contraction) so the integer result is bit-identical to the reference.
__device__ __forceinline__ float quant_u8(float x) {
  float t = __fdiv_rn(__fadd_rn(x, 1.0f), 2.0f);
  t = __fadd_rn(__fmul_rn(255.0f, t), 0.5f);
  t = fminf(fmaxf(t, 0.0f), 255.0f);
  return truncf(t);
}
__global__ void k_quantize_u8(const float* x, uint8_t* q, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) q[i] = (uint8_t)quant_u8(x[i]);
}
// bilinear, align_corners=False (torch area_pixel_compute_source_index): src = scale*(dst+0.5)-0.5 clamped at 0
template <typename T> __global__ void k_qrn(const float* x, T* out, int N, int C, int H, int W, int OH, int OW, int quantize, float sh, float sw) {
  const long long total = (long long)N * OH * OW * C;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C; const int ow = (int)(t % OW); t /= OW; const int oh = (int)(t % OH); const int n = (int)(t / OH);
    float fy = __fadd_rn(__fmul_rn(sh, (float)oh + 0.5f), -0.5f); if (fy < 0.f) fy = 0.f;
    float fx = __fadd_rn(__fmul_rn(sw, (float)ow + 0.5f), -0.5f); if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* p = x + ((long long)n * C + c) * H * W;
    float v00 = p[(long long)y0 * W + x0], v01 = p[(long long)y0 * W + x1], v10 = p[(long long)y1 * W + x0], v11 = p[(long long)y1 * W + x1];
    if (quantize) { v00 = quant_u8(v00); v01 = quant_u8(v01); v10 = quant_u8(v10); v11 = quant_u8(v11); }
    else { v00 = truncf(v00); v01 = truncf(v01); v10 = truncf(v10); v11 = truncf(v11); }
    float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    v = fminf(fmaxf(v, 0.f), 255.f);
    v = (v / 255.0f - 0.5f) / 0.5f;
    out[i] = from_f<T>(v);
  }
}
extern "C" int sg_quantize_resize_normalize(int dtype, const float* x, void* out, uint8_t* quant_out, int N, int C, int H, int W, int OH, int OW, int quantize, sg_stream_t s) {
  SG_CHECK(x && out && N > 0 && C > 0, "sg_quantize_resize_normalize: bad args");
  hipStream_t st = (hipStream_t)s;
  if (quant_out) hipLaunchKernelGGL(k_quantize_u8, dim3(grid1d((long long)N * C * H * W)), dim3(256), 0, st, x, quant_out, (long long)N * C * H * W);
  const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_qrn<T>, dim3(grid1d((long long)N * OH * OW * C)), dim3(256), 0, st, x, (T*)out, N, C, H, W, OH, OW, quantize, sh, sw));
  SG_LAUNCH_CHECK();
  return 0;
}

// generic NHWC pooling; output may be a channel slice [c_off, c_off+C) of a wider (ldy) concat tensor
template <typename T> __global__ void k_pool2d(const T* x, T* y, int N, int H, int W, int C, int k, int stride, int pad, int mode, int OH, int OW, int ldy, int c_off) {
  const long long total = (long long)N * OH * OW * C;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C; const int ow = (int)(t % OW); t /= OW; const int oh = (int)(t % OH); const int n = (int)(t / OH);
    float acc = (mode == 0) ? -INFINITY : 0.f; int cnt = 0;
    for (int r = 0; r < k; r++) {
      const int h = oh * stride - pad + r; if (h < 0 || h >= H) continue;
      for (int q = 0; q < k; q++) {
        const int w = ow * stride - pad + q; if (w < 0 || w >= W) continue;
        const float v = to_f<T>(x[(((long long)n * H + h) * W + w) * C + c]);
        if (mode == 0) acc = fmaxf(acc, v); else acc += v;
        cnt++;
      }
    }
    if (mode == 1) acc /= (float)(k * k); else if (mode == 2) acc /= (float)cnt;
    y[(((long long)n * OH + oh) * OW + ow) * ldy + c_off + c] = from_f<T>(acc);
  }
}
// bf16, C / ldy / c_off multiples of 8: a thread owns 8 channels of one output pixel (16-byte loads and store, 32-bit index arithmetic,
// same accumulation order as k_pool2d -> identical results). The scalar kernel was 20 % of the FID leg's kernel time
// (profiles/r02_fid_leg_kerneltrace.txt: 286 launches, 480 us average).
__global__ __launch_bounds__(256) void k_pool2d_v8(const bf16_t* x, bf16_t* y, int N, int H, int W, int C, int k, int stride, int pad, int mode, int OH, int OW, int ldy, int c_off) {
  const int CV = C / 8;
  const unsigned total = (unsigned)N * OH * OW * CV;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned q0 = i / CV, cv = i - q0 * CV;
    const unsigned ow = q0 % OW, t = q0 / OW, oh = t % OH, n = t / OH;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = (mode == 0) ? -INFINITY : 0.f;
    int cnt = 0;
    for (int r = 0; r < k; r++) {
      const int h = (int)oh * stride - pad + r; if (h < 0 || h >= H) continue;
      for (int q = 0; q < k; q++) {
        const int w = (int)ow * stride - pad + q; if (w < 0 || w >= W) continue;
        float v[8];
        unpack16<bf16_t>(*(const u32x4*)(x + (((long long)n * H + h) * W + w) * C + cv * 8), v);
#pragma unroll
        for (int e = 0; e < 8; e++) { if (mode == 0) acc[e] = fmaxf(acc[e], v[e]); else acc[e] += v[e]; }
        cnt++;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) { if (mode == 1) acc[e] /= (float)(k * k); else if (mode == 2) acc[e] /= (float)cnt; }
    *(u32x4*)(y + (long long)q0 * ldy + c_off + cv * 8) = pack16<bf16_t>(acc);
  }
}
// fp32, C / ldy / c_off multiples of 4: the same walk with 4 channels (one 16-byte vector) per thread -- same accumulation order as k_pool2d, identical results.
// (Round 6: with the convolutions of the fp32 InceptionV3 on the bf16x3 path the scalar kernel was 16 % of the FID leg, 513 us per launch.)
__global__ __launch_bounds__(256) void k_pool2d_f4(const float* x, float* y, int N, int H, int W, int C, int k, int stride, int pad, int mode, int OH, int OW, int ldy, int c_off) {
  const int CV = C / 4;
  const unsigned total = (unsigned)N * OH * OW * CV;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned q0 = i / CV, cv = i - q0 * CV;
    const unsigned ow = q0 % OW, t = q0 / OW, oh = t % OH, n = t / OH;
    f32x4 acc;
#pragma unroll
    for (int e = 0; e < 4; e++) acc[e] = (mode == 0) ? -INFINITY : 0.f;
    int cnt = 0;
    for (int r = 0; r < k; r++) {
      const int h = (int)oh * stride - pad + r; if (h < 0 || h >= H) continue;
      for (int q = 0; q < k; q++) {
        const int w = (int)ow * stride - pad + q; if (w < 0 || w >= W) continue;
        const f32x4 v = *(const f32x4*)(x + (((long long)n * H + h) * W + w) * C + cv * 4);
#pragma unroll
        for (int e = 0; e < 4; e++) { if (mode == 0) acc[e] = fmaxf(acc[e], v[e]); else acc[e] += v[e]; }
        cnt++;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) { if (mode == 1) acc[e] /= (float)(k * k); else if (mode == 2) acc[e] /= (float)cnt; }
    *(f32x4*)(y + (long long)q0 * ldy + c_off + cv * 4) = acc;
  }
}
extern "C" int sg_pool2d(int dtype, const void* x, void* y, int N, int H, int W, int C, int k, int stride, int pad, int mode, int ldy, int c_off, sg_stream_t s) {
  SG_CHECK(x && y && k > 0 && stride > 0, "sg_pool2d: bad args");
  const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
  if (dtype == SG_DTYPE_F32 && C % 4 == 0 && ldy % 4 == 0 && c_off % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && (long long)N * OH * OW * (C / 4) < (1ll << 31) &&
      !getenv("SG_POOL_SCALAR")) {
    hipLaunchKernelGGL(k_pool2d_f4, dim3(grid1d((long long)N * OH * OW * (C / 4))), dim3(256), 0, (hipStream_t)s, (const float*)x, (float*)y, N, H, W, C, k, stride, pad, mode, OH, OW, ldy, c_off);
    SG_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == SG_DTYPE_BF16 && C % 8 == 0 && ldy % 8 == 0 && c_off % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && (long long)N * OH * OW * (C / 8) < (1ll << 31)) {
    hipLaunchKernelGGL(k_pool2d_v8, dim3(grid1d((long long)N * OH * OW * (C / 8))), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, k, stride, pad, mode, OH, OW, ldy, c_off);
    SG_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_pool2d<T>, dim3(grid1d((long long)N * OH * OW * C)), dim3(256), 0, (hipStream_t)s, (const T*)x, (T*)y, N, H, W, C, k, stride, pad, mode, OH, OW, ldy, c_off));
  SG_LAUNCH_CHECK();
  return 0;
}
template <typename T> __global__ void k_global_avgpool(const T* x, float* y, int N, int HW, int C) {
  const long long total = (long long)N * C;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / C), c = (int)(i % C);
    const T* p = x + (long long)n * HW * C + c;
    float acc = 0.f;
    for (int k = 0; k < HW; k++) acc += to_f<T>(p[(long long)k * C]);
    y[i] = acc / (float)HW;
  }
}
extern "C" int sg_global_avgpool(int dtype, const void* x, float* y, int N, int HW, int C, sg_stream_t s) {
  SG_CHECK(x && y, "sg_global_avgpool: null");
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_global_avgpool<T>, dim3(grid1d((long long)N * C)), dim3(256), 0, (hipStream_t)s, (const T*)x, y, N, HW, C));
  SG_LAUNCH_CHECK();
  return 0;
}

// FID moments: sum_f[c] += sum_n f[n][c]; sum_ff[c1][c2] += sum_n f[n][c1]*f[n][c2] in fp64.
// 16x16 output tile per workgroup, features staged through LDS in chunks of 64 samples.
__global__ __launch_bounds__(256) void k_feat_moments(const float* f, int n, int C, double* sum_f, double* sum_ff) {
  __shared__ float a[64][17], b[64][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c1 = blockIdx.y * 16 + ty, c2 = blockIdx.x * 16 + tx;
  double acc = 0.0, accf = 0.0;
  for (int n0 = 0; n0 < n; n0 += 64) {
    for (int e = threadIdx.x; e < 64 * 16; e += 256) {
      const int r = e >> 4, cc = e & 15;
      const int nn = n0 + r;
      a[r][cc] = (nn < n && blockIdx.y * 16 + cc < C) ? f[(long long)nn * C + blockIdx.y * 16 + cc] : 0.f;
      b[r][cc] = (nn < n && blockIdx.x * 16 + cc < C) ? f[(long long)nn * C + blockIdx.x * 16 + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < 64; r++) { acc += (double)a[r][ty] * (double)b[r][tx]; if (blockIdx.x == 0 && tx == 0) accf += (double)a[r][ty]; }
    __syncthreads();
  }
  if (c1 < C && c2 < C) sum_ff[(long long)c1 * C + c2] += acc;
  if (blockIdx.x == 0 && tx == 0 && c1 < C) sum_f[c1] += accf;
}
extern "C" int sg_feat_moments_accumulate(const float* f, int n, int C, double* sum_f, double* sum_ff, sg_stream_t s) {
  SG_CHECK(f && sum_f && sum_ff && n > 0 && C > 0, "sg_feat_moments_accumulate: bad args");
  hipLaunchKernelGGL(k_feat_moments, dim3((C + 15) / 16, (C + 15) / 16), dim3(256), 0, (hipStream_t)s, f, n, C, sum_f, sum_ff);
  SG_LAUNCH_CHECK();
  return 0;
}

// top-k hit test with sklearn.metrics.top_k_accuracy_score's tie rule (stable ascending argsort, reversed: among equal
// scores the HIGHER class index ranks first). One wave per sample. hits[n] = 1 iff fewer than k classes beat the true one.
// (reference src/metrics/ins.py:62-76 runs this on the host through sklearn)
__global__ __launch_bounds__(256) void k_topk_hits(const float* scores, int ld, int ncls, const int64_t* labels, int k, int N, uint8_t* hits) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  const float* s = scores + (long long)n * ld;
  const int t = (int)labels[n];
  const float st = (t >= 0 && t < ncls) ? s[t] : INFINITY;
  int beat = 0;
  for (int c = lane; c < ncls; c += 64) {
    const float v = s[c];
    beat += (v > st) || (v == st && c > t);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) beat += __shfl_xor(beat, o, 64);
  if (lane == 0) hits[n] = (t >= 0 && t < ncls && beat < k) ? 1 : 0;
}
extern "C" int sg_topk_hits(const float* scores, int ld, int ncls, const int64_t* labels, int k, int N, uint8_t* hits, sg_stream_t s) {
  SG_CHECK(scores && labels && hits && N > 0 && ncls > 0 && k > 0, "sg_topk_hits: bad args");
  hipLaunchKernelGGL(k_topk_hits, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)s, scores, ld, ncls, labels, k, N, hits);
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- top-k training of the generator (reference src/worker.py:565-566: torch.topk(adv_output, k).values; losses.py:364-366) ------
// vals[r] / idx[r] = the r-th largest of x[0..n) (descending; among equal values the lower index first). Rank by counting:
// n <= a few thousand logits, one workgroup. The set of selected logits is what the loss depends on (a mean over them).
__global__ __launch_bounds__(256) void k_topk_select(const float* x, int n, int k, float* vals, int* idx) {
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = x[i];
    int rank = 0;
    for (int j = 0; j < n; j++) {
      const float u = x[j];
      rank += (u > v) || (u == v && j < i);
    }
    if (rank < k) { vals[rank] = v; idx[rank] = i; }
  }
}
extern "C" int sg_topk_select(const float* x, int n, int k, float* vals, int* idx, sg_stream_t s) {
  SG_CHECK(x && vals && idx && n > 0 && k > 0 && k <= n, "sg_topk_select: bad args");
  hipLaunchKernelGGL(k_topk_select, dim3(1), dim3(256), 0, (hipStream_t)s, x, n, k, vals, idx);
  SG_LAUNCH_CHECK();
  return 0;
}
// dx[i] = g[r] if i == idx[r] else 0
__global__ __launch_bounds__(256) void k_topk_scatter(const float* g, const int* idx, int k, float* dx, int n) {
  for (int i = threadIdx.x; i < n; i += 256) dx[i] = 0.f;
  __syncthreads();
  for (int r = threadIdx.x; r < k; r += 256) dx[idx[r]] = g[r];
}
extern "C" int sg_topk_scatter(const float* g, const int* idx, int k, float* dx, int n, sg_stream_t s) {
  SG_CHECK(g && idx && dx && n > 0 && k > 0 && k <= n, "sg_topk_scatter: bad args");
  hipLaunchKernelGGL(k_topk_scatter, dim3(1), dim3(256), 0, (hipStream_t)s, g, idx, k, dx, n);
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- precision / recall / density / coverage on the device (reference src/metrics/prdc.py:87-168: sklearn pairwise_distances with
// n_jobs = 8 + numpy argpartition over 50k x 50k float64 matrices on the host) -------------------------------------------------------
// The pairwise term -2 x.y + |y|^2 comes from the exact-fp32 MFMA GEMM (sg_gemm, alpha = -2, bias = |y|^2) in row blocks; the kernels
// below add |x|^2 per row, clamp at 0 (sklearn clips its own cancellation noise the same way) and reduce every row on the fly, so no
// distance matrix larger than one row block ever exists. All comparisons are on SQUARED distances (monotone in the distance).

// sq[n] = sum_c f[n][c]^2 (fp32 features, fp64 accumulation)
__global__ __launch_bounds__(256) void k_row_sqnorm(const float* f, int n, int C, float* sq) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (int c = lane; c < C; c += 64) { const double v = f[(long long)r * C + c]; acc += v * v; }
  acc = wave_sum_d(acc);
  if (lane == 0) sq[r] = (float)acc;
}
extern "C" int sg_row_sqnorm(const float* f, int n, int C, float* sq, sg_stream_t s) {
  SG_CHECK(f && sq && n > 0 && C > 0, "sg_row_sqnorm: bad args");
  hipLaunchKernelGGL(k_row_sqnorm, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)s, f, n, C, sq);
  SG_LAUNCH_CHECK();
  return 0;
}

#define SG_KTH_MAX 16
// out[r] = k-th smallest (1-based) of max(0, D[r][c] + row_add[r]) over c < cols. One wave per row: every lane keeps its own k
// smallest in a sorted register list, then k rounds of "global minimum, owner pops" merge the 64 lists.
__global__ __launch_bounds__(256) void k_kth_smallest_rows(const float* D, long long ld, int rows, int cols, int k, const float* row_add, float* out) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float add = row_add ? row_add[r] : 0.f;
  float best[SG_KTH_MAX];
#pragma unroll
  for (int i = 0; i < SG_KTH_MAX; i++) best[i] = INFINITY;
  const float* row = D + (long long)r * ld;
  for (int c = lane; c < cols; c += 64) {
    float v = fmaxf(row[c] + add, 0.f);
    if (v < best[SG_KTH_MAX - 1]) {
#pragma unroll
      for (int i = 0; i < SG_KTH_MAX; i++) {        // insertion into the ascending list (only the first k entries matter)
        const float b = best[i];
        const bool lt = v < b;
        best[i] = lt ? v : b;
        v = lt ? b : v;
      }
    }
  }
  float kth = INFINITY;
  for (int round = 0; round < k; round++) {
    const float mine = best[0];
    const float m = -wave_max(-mine);
    kth = m;
    // exactly one lane holding the minimum pops it (lowest lane among equals)
    const unsigned long long owners = __ballot(mine == m);
    const int first = __ffsll((long long)owners) - 1;
    if (lane == first) {
#pragma unroll
      for (int i = 0; i < SG_KTH_MAX - 1; i++) best[i] = best[i + 1];
      best[SG_KTH_MAX - 1] = INFINITY;
    }
  }
  if (lane == 0) out[r] = kth;
}
extern "C" int sg_kth_smallest_rows(const float* D, long long ld, int rows, int cols, int k, const float* row_add, float* out, sg_stream_t s) {
  SG_CHECK(D && out && rows > 0 && cols > 0 && k > 0 && k <= SG_KTH_MAX && k <= cols, "sg_kth_smallest_rows: bad args (k <= 16)");
  hipLaunchKernelGGL(k_kth_smallest_rows, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, D, ld, rows, cols, k, row_add, out);
  SG_LAUNCH_CHECK();
  return 0;
}

// One pass over a row block of the real-vs-fake squared distances d2[r][c] = max(0, D[r][c] + row_add[r]):
//   col_cnt[c] += #{ r : d2 < r2_row[r] }      (precision / density: fake c inside the k-NN ball of real r)      integer atomics: deterministic
//   row_any[r]  = any_c d2 < r2_col[c]         (recall: real r inside the ball of some fake c)
//   row_min[r]  = min_c d2                     (coverage: compared with r2_row[r] by the caller)
__global__ __launch_bounds__(256) void k_prdc_rows(const float* D, long long ld, int rows, int cols, const float* row_add, const float* r2_row, const float* r2_col,
                                                   int* col_cnt, uint8_t* row_any, float* row_min) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float add = row_add[r], rr = r2_row[r];
  const float* row = D + (long long)r * ld;
  float mn = INFINITY;
  int any = 0;
  for (int c = lane; c < cols; c += 64) {
    const float v = fmaxf(row[c] + add, 0.f);
    mn = fminf(mn, v);
    any |= (v < r2_col[c]) ? 1 : 0;
    if (v < rr) atomicAdd(col_cnt + c, 1);
  }
  mn = -wave_max(-mn);
  const unsigned long long a = __ballot(any != 0);
  if (lane == 0) { row_min[r] = mn; row_any[r] = a ? 1 : 0; }
}
extern "C" int sg_prdc_rows(const float* D, long long ld, int rows, int cols, const float* row_add, const float* r2_row, const float* r2_col,
                            int* col_cnt, uint8_t* row_any, float* row_min, sg_stream_t s) {
  SG_CHECK(D && row_add && r2_row && r2_col && col_cnt && row_any && row_min && rows > 0 && cols > 0, "sg_prdc_rows: bad args");
  hipLaunchKernelGGL(k_prdc_rows, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, D, ld, rows, cols, row_add, r2_row, r2_col, col_cnt, row_any, row_min);
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- PIL resizers ("clean" = bicubic, "friendly" = bilinear for InceptionV3_tf; reference src/utils/resize.py:39-78: Image.resize on mode 'F'
// single-channel images). Pillow resamples separably -- horizontal pass, then vertical -- with per-output-pixel coefficient windows whose
// support is scaled by the reduction factor (antialiasing); for 32-bit images it accumulates in double and stores float after each pass.
// The coefficient tables (bounds[o] = {first source index, count}, kk[o][ksize] doubles, normalised) are computed on the host exactly as
// Pillow's precompute_coeffs does (metrics.py pil_coeffs); the two kernels below only apply them.
// pass 1: uint8-quantised NCHW fp32 input (quantisation of ops.quantize_images applied on load) -> tmp [N][C][H][OW] fp32
__global__ __launch_bounds__(256) void k_pil_resample_h(const float* x, float* tmp, int N, int C, int H, int W, int OW, const int* bounds, const double* kk, int ksize, int quantize) {
  const long long total = (long long)N * C * H * OW;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ox = (int)(i % OW);
    const long long row = i / OW;                      // (n, c, h)
    const int x0 = bounds[2 * ox], cnt = bounds[2 * ox + 1];
    const float* p = x + row * W + x0;
    const double* k = kk + (long long)ox * ksize;
    double ss = 0.0;
    for (int j = 0; j < cnt; j++) { const float v = quantize ? quant_u8(p[j]) : truncf(p[j]); ss += (double)v * k[j]; }
    tmp[i] = (float)ss;
  }
}
// pass 2: tmp [N][C][H][OW] -> out [N][OH][OW][C] NHWC, normalised (v / 255 - 0.5) / 0.5 (no clipping: PIL's bicubic overshoot is kept,
// unlike the legacy resizer)
template <typename T> __global__ __launch_bounds__(256) void k_pil_resample_v(const float* tmp, T* out, int N, int C, int H, int OH, int OW, const int* bounds, const double* kk, int ksize) {
  const long long total = (long long)N * OH * OW * C;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C; const int ox = (int)(t % OW); t /= OW; const int oy = (int)(t % OH); const int n = (int)(t / OH);
    const int y0 = bounds[2 * oy], cnt = bounds[2 * oy + 1];
    const float* p = tmp + (((long long)n * C + c) * H + y0) * OW + ox;
    const double* k = kk + (long long)oy * ksize;
    double ss = 0.0;
    for (int j = 0; j < cnt; j++) ss += (double)p[(long long)j * OW] * k[j];
    float v = (float)ss;
    v = (v / 255.0f - 0.5f) / 0.5f;
    out[i] = from_f<T>(v);
  }
}
extern "C" int sg_pil_resize_normalize(int dtype, const float* x, void* out, float* tmp, int N, int C, int H, int W, int OH, int OW,
                                       const int* bounds_h, const double* kk_h, int ksize_h, const int* bounds_v, const double* kk_v, int ksize_v,
                                       int quantize, sg_stream_t s) {
  SG_CHECK(x && out && tmp && bounds_h && kk_h && bounds_v && kk_v && N > 0 && C > 0 && ksize_h > 0 && ksize_v > 0, "sg_pil_resize_normalize: bad args");
  hipStream_t st = (hipStream_t)s;
  hipLaunchKernelGGL(k_pil_resample_h, dim3(grid1d((long long)N * C * H * OW)), dim3(256), 0, st, x, tmp, N, C, H, W, OW, bounds_h, kk_h, ksize_h, quantize);
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_pil_resample_v<T>, dim3(grid1d((long long)N * OH * OW * C)), dim3(256), 0, st, (const float*)tmp, (T*)out, N, C, H, OH, OW, bounds_v, kk_v, ksize_v));
  SG_LAUNCH_CHECK();
  return 0;
}
