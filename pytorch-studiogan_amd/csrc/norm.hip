// norm.hip -- BatchNorm2d(eps=1e-4, momentum=0.1) and ConditionalBatchNorm2d forward/backward on NHWC tensors
// (reference src/utils/ops.py:14-28,227-228). Statistics are produced as fp64 partial sums {sum x, sum x^2}
// so that the cross-rank reduction of sync-BN is ONE small all-reduce between sg_bn_partial_stats and
// sg_bn_finalize (reference src/models/model.py:161-165 uses nn.SyncBatchNorm's all_gather instead).
// HBM-bound: every kernel streams 16-byte vectors along C.
#include "common.h"
#include "../../include/sgamd.h"

#define DISPATCH_T(dtype, ...)                                          \
  if ((dtype) == SG_DTYPE_F32) { typedef float T; __VA_ARGS__; }        \
  else if ((dtype) == SG_DTYPE_BF16) { typedef bf16_t T; __VA_ARGS__; } \
  else { sg_set_error("bad dtype"); return -1; }

// ---- statistics ------------------------------------------------------------------------------------------
template <typename T> __global__ __launch_bounds__(256) void k_bn_partial(const T* x, int ldx, long long rows, int C, double* partial, long long rpb) {
  __shared__ double sm[2][4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  long long r0 = blockIdx.y * rpb, r1 = r0 + rpb; if (r1 > rows) r1 = rows;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    for (long long r = r0 + ry; r < r1; r += 4) {
      const double v = (double)to_f<T>(x[r * ldx + c]);
      s1 += v; s2 += v * v;
    }
  }
  sm[0][ry][cx] = s1; sm[1][ry][cx] = s2;
  __syncthreads();
  if (ry == 0 && c < C) {
    atomicAdd(partial + 2 * c, sm[0][0][cx] + sm[0][1][cx] + sm[0][2][cx] + sm[0][3][cx]);
    atomicAdd(partial + 2 * c + 1, sm[1][0][cx] + sm[1][1][cx] + sm[1][2][cx] + sm[1][3][cx]);
  }
}
// Streaming variant: a thread owns one 16-byte channel vector and walks pixels (16-byte loads instead of 2-byte ones);
// fp32 partials are flushed into fp64 every 32 pixels, lanes of a block are combined through LDS, one fp64 atomic per
// channel and block.
template <typename T> __global__ __launch_bounds__(256) void k_bn_partial_stream(const T* x, int ldx, long long rows, int C, double* partial, long long rpb) {
  constexpr int V = ET<T>::VEC;
  extern __shared__ double smd[];                       // [lanes_p][C][2]
  const int CV = C / V;
  const int lanes_p = blockDim.x / CV;
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  long long r0 = blockIdx.x * rpb, r1 = r0 + rpb; if (r1 > rows) r1 = rows;
  double d1[V], d2[V];
#pragma unroll
  for (int e = 0; e < V; e++) { d1[e] = 0.0; d2[e] = 0.0; }
  for (long long rb = r0 + pl; rb < r1; rb += 32ll * lanes_p) {
    float s1[V], s2[V];
#pragma unroll
    for (int e = 0; e < V; e++) { s1[e] = 0.f; s2[e] = 0.f; }
    for (int k = 0; k < 32; k += 4) {                        // four loads in flight per lane (see k_bn_apply_stream); same summation order
      u32x4 raw[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const long long r = rb + (long long)(k + i) * lanes_p;
        raw[i] = r < r1 ? *(const u32x4*)(x + r * ldx + cv * V) : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float xv[V];
        unpack16<T>(raw[i], xv);
#pragma unroll
        for (int e = 0; e < V; e++) { s1[e] += xv[e]; s2[e] += xv[e] * xv[e]; }
      }
      if (rb + (long long)(k + 4) * lanes_p >= r1) break;
    }
#pragma unroll
    for (int e = 0; e < V; e++) { d1[e] += (double)s1[e]; d2[e] += (double)s2[e]; }
  }
#pragma unroll
  for (int e = 0; e < V; e++) { smd[((long long)pl * C + cv * V + e) * 2] = d1[e]; smd[((long long)pl * C + cv * V + e) * 2 + 1] = d2[e]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    double a = 0.0;
    for (int l = 0; l < lanes_p; l++) a += smd[(long long)l * C * 2 + i];
    atomicAdd(partial + i, a);
  }
}
extern "C" int sg_bn_partial_stats(int dtype, const void* x, int ldx, long long rows, int C, double* partial, sg_stream_t s) {
  SgProfScope prof((hipStream_t)s, (double)rows * C * (dtype == SG_DTYPE_BF16 ? 2.0 : 4.0), 4);
  SG_CHECK(x && partial && rows > 0 && C > 0, "sg_bn_partial_stats: bad args");
  bool done = false;
  DISPATCH_T(dtype, {
    constexpr int V = ET<T>::VEC;
    const int CV = C / V;
    if (C % V == 0 && ldx % V == 0 && ((uintptr_t)x & 15) == 0 && CV <= 128 && rows >= 4096) {
      const int lanes_p = 256 / CV;
      long long rpb = (rows + 1023) / 1024; if (rpb < 32ll * lanes_p) rpb = 32ll * lanes_p;
      const int gx = (int)((rows + rpb - 1) / rpb);
      hipLaunchKernelGGL(k_bn_partial_stream<T>, dim3(gx), dim3(CV * lanes_p), (size_t)lanes_p * C * 2 * sizeof(double), (hipStream_t)s, (const T*)x, ldx, rows, C, partial, rpb);
      done = true;
    }
  });
  if (done) { SG_LAUNCH_CHECK(); return 0; }
  int ct = (C + 63) / 64;
  long long want = 2048 / ct; if (want < 1) want = 1;
  long long rpb = (rows + want - 1) / want; if (rpb < 64) rpb = 64;
  int gy = (int)((rows + rpb - 1) / rpb);
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_bn_partial<T>, dim3(ct, gy), dim3(256), 0, (hipStream_t)s, (const T*)x, ldx, rows, C, partial, rpb));
  SG_LAUNCH_CHECK();
  return 0;
}
// partial[2 c + {0, 1}] += sum over the tile rows of stats[row][c][{0, 1}] (the per-tile sums a convolution epilogue wrote: conv_v2.h
// sg_conv_epilogue `stats`). Block (64 channels x 4 row phases), fixed summation order per block, fp64; blocks of different row groups meet in
// fp64 atomics like k_bn_partial_stream's.
__global__ __launch_bounds__(256) void k_bn_stats_from_tiles(const float* stats, int nrows, int C, double* partial, int rpb) {
  __shared__ double sm[2][4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  int r0 = blockIdx.y * rpb, r1 = r0 + rpb;
  if (r1 > nrows) r1 = nrows;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    for (int r = r0 + ry; r < r1; r += 4) {
      const float2 v = *(const float2*)(stats + ((long long)r * C + c) * 2);
      s1 += (double)v.x; s2 += (double)v.y;
    }
  }
  sm[0][ry][cx] = s1; sm[1][ry][cx] = s2;
  __syncthreads();
  if (ry == 0 && c < C) {
    atomicAdd(partial + 2 * c, (sm[0][0][cx] + sm[0][1][cx]) + (sm[0][2][cx] + sm[0][3][cx]));
    atomicAdd(partial + 2 * c + 1, (sm[1][0][cx] + sm[1][1][cx]) + (sm[1][2][cx] + sm[1][3][cx]));
  }
}
extern "C" int sg_bn_stats_from_tiles(const float* stats, int nrows, int C, double* partial, sg_stream_t s) {
  SG_CHECK(stats && partial && nrows > 0 && C > 0, "sg_bn_stats_from_tiles: bad args");
  SgProfScope prof((hipStream_t)s, (double)nrows * C * 8.0, 4);
  int gy = (nrows + 255) / 256;
  if (gy > 64) gy = 64;
  const int rpb = (nrows + gy - 1) / gy;
  hipLaunchKernelGGL(k_bn_stats_from_tiles, dim3((C + 63) / 64, gy), dim3(256), 0, (hipStream_t)s, stats, nrows, C, partial, rpb);
  SG_LAUNCH_CHECK();
  return 0;
}
__global__ void k_bn_finalize(const double* partial, double count, int C, float eps, float momentum, float* mean, float* invstd, float* rm, float* rv) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double m = partial[2 * c] / count;
  double var = partial[2 * c + 1] / count - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rm) {
    const double unb = (count > 1.0) ? var * count / (count - 1.0) : var;
    rm[c] = (1.f - momentum) * rm[c] + momentum * (float)m;
    rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unb;
  }
}
extern "C" int sg_bn_finalize(const double* partial, double count, int C, float eps, float momentum, float* mean, float* invstd, float* running_mean, float* running_var, sg_stream_t s) {
  SG_CHECK(partial && mean && invstd && count > 0, "sg_bn_finalize: bad args");
  SG_CHECK((running_mean == nullptr) == (running_var == nullptr), "sg_bn_finalize: running stats must come as a pair");
  hipLaunchKernelGGL(k_bn_finalize, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)s, partial, count, C, eps, momentum, mean, invstd, running_mean, running_var);
  SG_LAUNCH_CHECK();
  return 0;
}
__global__ void k_bn_from_running(const float* rm, const float* rv, int C, float eps, float* mean, float* invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = rm[c];
  invstd[c] = 1.f / sqrtf(rv[c] + eps);
}
extern "C" int sg_bn_from_running(const float* running_mean, const float* running_var, int C, float eps, float* mean, float* invstd, sg_stream_t s) {
  SG_CHECK(running_mean && running_var && mean && invstd, "sg_bn_from_running: null");
  hipLaunchKernelGGL(k_bn_from_running, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)s, running_mean, running_var, C, eps, mean, invstd);
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- apply -------------------------------------------------------------------------------------------------
// one thread = one 16-byte vector of channels; requires C % VEC == 0 (checked on the host; scalar kernel otherwise)
template <typename T, bool VECP> __global__ __launch_bounds__(256) void k_bn_apply(const T* x, T* y, int N, long long HW, int C, const float* mean, const float* invstd, const float* gain, const float* bias, int gsn, int relu) {
  constexpr int V = VECP ? ET<T>::VEC : 1;
  const int CV = C / V;
  const long long total = (long long)N * HW * CV;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cv = (int)(i % CV); const long long pix = i / CV; const int n = (int)(pix / HW);
    const int c0 = cv * V;
    float xv[V];
    if (VECP) unpack16<T>(*(const u32x4*)(x + pix * C + c0), xv); else xv[0] = to_f<T>(x[pix * C + c0]);
#pragma unroll
    for (int e = 0; e < V; e++) {
      const int c = c0 + e;
      // same evaluation order as the backward's recomputation (xhat * gain + bias)
      const float xh = (xv[e] - mean[c]) * invstd[c];
      const float ga = gain ? gain[(long long)n * gsn + c] : 1.f;
      const float bi = bias ? bias[(long long)n * gsn + c] : 0.f;
      float v = xh * ga + bi;
      if (relu) v = fmaxf(v, 0.f);
      xv[e] = v;
    }
    if (VECP) *(u32x4*)(y + pix * C + c0) = pack16<T>(xv); else y[pix * C + c0] = from_f<T>(xv[0]);
  }
}
static inline int grid_for(long long total) { long long b = (total + 255) / 256; if (b > 256 * 32) b = 256 * 32; if (b < 1) b = 1; return (int)b; }
template <typename T> static bool vec_ok(const void* a, const void* b, int C) {
  return (C % ET<T>::VEC == 0) && ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0;
}
// Streaming variant: grid (pixel chunks, N). A thread owns ONE 16-byte channel vector of ONE sample, so its
// scale/shift (8 or 4 channels) are computed once and the loop body is load -> fma -> store (the generic kernel
// re-reads 4 parameters per channel per element, which made it VALU/L1-bound at ~40 % of HBM speed).
template <typename T, int UN, bool NT> __global__ __launch_bounds__(256) void k_bn_apply_stream(const T* x, T* y, long long HW, int C, const float* mean, const float* invstd, const float* gain, const float* bias, int gsn, int relu, int ppb) {
  constexpr int V = ET<T>::VEC;
  const int CV = C / V;
  const int n = blockIdx.y;
  const int lanes_p = blockDim.x / CV;                // pixels per block iteration (blockDim.x == CV * lanes_p)
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  float mu[V], is[V], ga[V], bi[V];
#pragma unroll
  for (int e = 0; e < V; e++) {
    const int c = cv * V + e;
    mu[e] = mean[c]; is[e] = invstd[c];
    ga[e] = gain ? gain[(long long)n * gsn + c] : 1.f;
    bi[e] = bias ? bias[(long long)n * gsn + c] : 0.f;
  }
  const long long p0 = (long long)blockIdx.x * ppb;
  long long p1 = p0 + ppb; if (p1 > HW) p1 = HW;
  const T* xs = x + (long long)n * HW * C + cv * V;
  T* ys = y + (long long)n * HW * C + cv * V;
  // Four pixels per trip, all four loads issued before the first use: with one 16-byte load in flight per lane the 32 resident waves
  // of a CU hold 32 KB, which at ~2 us of HBM latency is 4.2 TB/s for the chip -- exactly what the r02 roofline_hbm leg measured for
  // the batch-norm family (0.53 of peak). Same arithmetic per element, so results are unchanged.
  auto one = [&](const u32x4& raw, long long pix) {
    float xv[V];
    unpack16<T>(raw, xv);
#pragma unroll
    for (int e = 0; e < V; e++) {
      float v = ((xv[e] - mu[e]) * is[e]) * ga[e] + bi[e];   // same evaluation order as the backward's recomputation
      if (relu) v = fmaxf(v, 0.f);
      xv[e] = v;
    }
    const u32x4 o = pack16<T>(xv);
    if (NT) __builtin_nontemporal_store(o, (u32x4*)(ys + pix * C)); else *(u32x4*)(ys + pix * C) = o;
  };
  long long pix = p0 + pl;
  for (; pix + (long long)(UN - 1) * lanes_p < p1; pix += (long long)UN * lanes_p) {
    u32x4 r[UN];
#pragma unroll
    for (int i = 0; i < UN; i++) r[i] = NT ? __builtin_nontemporal_load((const u32x4*)(xs + (pix + (long long)i * lanes_p) * C)) : *(const u32x4*)(xs + (pix + (long long)i * lanes_p) * C);
#pragma unroll
    for (int i = 0; i < UN; i++) one(r[i], pix + (long long)i * lanes_p);
  }
  for (; pix < p1; pix += lanes_p) one(*(const u32x4*)(xs + pix * C), pix);
}
template <typename T> static void bn_apply_stream_launch(int variant, dim3 grid, dim3 blk, hipStream_t st, const T* x, T* y, long long HW, int C, const float* mean, const float* invstd, const float* gain,
                                                         const float* bias, int gsn, int relu, int ppb) {
  if (variant == 0) hipLaunchKernelGGL((k_bn_apply_stream<T, 4, false>), grid, blk, 0, st, x, y, HW, C, mean, invstd, gain, bias, gsn, relu, ppb);
  else if (variant == 1) hipLaunchKernelGGL((k_bn_apply_stream<T, 8, false>), grid, blk, 0, st, x, y, HW, C, mean, invstd, gain, bias, gsn, relu, ppb);
  else if (variant == 2) hipLaunchKernelGGL((k_bn_apply_stream<T, 4, true>), grid, blk, 0, st, x, y, HW, C, mean, invstd, gain, bias, gsn, relu, ppb);
  else hipLaunchKernelGGL((k_bn_apply_stream<T, 8, true>), grid, blk, 0, st, x, y, HW, C, mean, invstd, gain, bias, gsn, relu, ppb);
}
extern "C" int sg_bn_apply(int dtype, const void* x, void* y, int N, long long HW, int C, const float* mean, const float* invstd, const float* gain, const float* bias, int gb_stride_n, int relu, sg_stream_t s) {
  SgProfScope prof((hipStream_t)s, 2.0 * N * (double)HW * C * (dtype == SG_DTYPE_BF16 ? 2.0 : 4.0), 4);
  SG_CHECK(x && y && mean && invstd, "sg_bn_apply: null");
  DISPATCH_T(dtype, {
    const int CV = C / ET<T>::VEC;
    if (vec_ok<T>(x, y, C) && CV <= 256 && N <= 65535 && HW >= 16) {      // (HW 64 -> 16: the 4 x 4 maps of a generator's first block took the generic kernel at 0.56 TB/s, 45 us for 25 MB)
      const int lanes_p = 256 / CV;
      // Variant 0 = four loads in flight, 1 = eight, 2 = four + non-temporal accesses, 3 = eight + non-temporal; `want` workgroups (SG_BN_APPLY="<variant><workgroups / 1024>", A/B).
      // Measured (tools/bn_bench.py, profiles/r06_bn_apply_variants_r7f.txt): ALONE, this pass streams tensors far beyond the 256 MB Infinity Cache 7-12 % faster with non-temporal
      // accesses, eight loads in flight and 4096 workgroups (4.64 -> 5.10 TB/s on BigGAN-128's 128^2 maps). IN THE STEP no gain could be shown: A, B, A, B
      // alternations read +1.0 ms for the non-temporal arm (profiles/r06_bn_apply_step_ab_r7l.txt), but an ABBA series showed every second PROCESS on such a box running 1.5 ms
      // faster whatever the variant (profiles/r06_bn_apply_reverse_walk_r7r.txt): corrected, the effect lies between -0.9 and +0.2 ms. Unresolved, so the default stays variant 0
      // (the pass's output is the next convolution's input; what it leaves in the Infinity Cache may matter as much as its own 0.1 ms).
      static const char* ev = getenv("SG_BN_APPLY");
      const int variant = ev && ev[0] >= '0' && ev[0] <= '3' ? ev[0] - '0' : 0;
      const long long want = ev && ev[0] && ev[1] >= '1' && ev[1] <= '9' ? 1024ll * (ev[1] - '0') : 2048;
      long long chunks = want / N; if (chunks < 1) chunks = 1;
      long long ppb = (HW + chunks - 1) / chunks; if (ppb < 4 * lanes_p) ppb = 4 * lanes_p;
      const int gx = (int)((HW + ppb - 1) / ppb);
      bn_apply_stream_launch<T>(variant, dim3(gx, N), dim3(CV * lanes_p), (hipStream_t)s, (const T*)x, (T*)y, HW, C, mean, invstd, gain, bias, gb_stride_n, relu, (int)ppb);
    } else if (vec_ok<T>(x, y, C)) hipLaunchKernelGGL((k_bn_apply<T, true>), dim3(grid_for((long long)N * HW * C / ET<T>::VEC)), dim3(256), 0, (hipStream_t)s, (const T*)x, (T*)y, N, HW, C, mean, invstd, gain, bias, gb_stride_n, relu);
    else hipLaunchKernelGGL((k_bn_apply<T, false>), dim3(grid_for((long long)N * HW * C)), dim3(256), 0, (hipStream_t)s, (const T*)x, (T*)y, N, HW, C, mean, invstd, gain, bias, gb_stride_n, relu);
  });
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- backward ----------------------------------------------------------------------------------------------
// stage 1: sums[n][c] = {sum_hw dy', sum_hw dy' * xhat}; grid (channel tiles, hw chunks, N); caller zeroes sums
template <typename T> __global__ __launch_bounds__(256) void k_bn_bwd_reduce(const T* x, const T* dy, long long HW, int C, const float* mean, const float* invstd, const float* gain, const float* bias, int gsn, int relu, float* sums, long long rpb) {
  __shared__ float sm[2][4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const int n = blockIdx.z;
  long long r0 = blockIdx.y * rpb, r1 = r0 + rpb; if (r1 > HW) r1 = HW;
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const float mu = mean[c], is = invstd[c];
    const float ga = gain ? gain[(long long)n * gsn + c] : 1.f;
    const float bi = bias ? bias[(long long)n * gsn + c] : 0.f;
    const T* xp = x + (long long)n * HW * C + c;
    const T* gp = dy + (long long)n * HW * C + c;
    for (long long r = r0 + ry; r < r1; r += 4) {
      const float xh = (to_f<T>(xp[r * C]) - mu) * is;
      float g = to_f<T>(gp[r * C]);
      if (relu && !(xh * ga + bi > 0.f)) g = 0.f;
      s1 += g; s2 += g * xh;
    }
  }
  sm[0][ry][cx] = s1; sm[1][ry][cx] = s2;
  __syncthreads();
  if (ry == 0 && c < C) {
    float* o = sums + ((long long)n * C + c) * 2;
    unsafeAtomicAdd(o, sm[0][0][cx] + sm[0][1][cx] + sm[0][2][cx] + sm[0][3][cx]);
    unsafeAtomicAdd(o + 1, sm[1][0][cx] + sm[1][1][cx] + sm[1][2][cx] + sm[1][3][cx]);
  }
}
// Streaming variant of stage 1 (16-byte loads; thread = one channel vector of one sample; LDS combine; float atomics)
template <typename T> __global__ __launch_bounds__(256) void k_bn_bwd_reduce_stream(const T* x, const T* dy, long long HW, int C, const float* mean, const float* invstd, const float* gain, const float* bias, int gsn, int relu, float* sums, int ppb) {
  constexpr int V = ET<T>::VEC;
  extern __shared__ float smf[];                        // [lanes_p][C][2]
  const int CV = C / V;
  const int n = blockIdx.y;
  const int lanes_p = blockDim.x / CV;
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  float mu[V], is[V], ga[V], bi[V], s1[V], s2[V];
#pragma unroll
  for (int e = 0; e < V; e++) {
    const int c = cv * V + e;
    mu[e] = mean[c]; is[e] = invstd[c];
    ga[e] = gain ? gain[(long long)n * gsn + c] : 1.f;
    bi[e] = bias ? bias[(long long)n * gsn + c] : 0.f;
    s1[e] = 0.f; s2[e] = 0.f;
  }
  const long long p0 = (long long)blockIdx.x * ppb;
  long long p1 = p0 + ppb; if (p1 > HW) p1 = HW;
  const long long base = (long long)n * HW * C + cv * V;
  auto one = [&](const u32x4& rx, const u32x4& rg) {
    float xv[V], gv[V];
    unpack16<T>(rx, xv);
    unpack16<T>(rg, gv);
#pragma unroll
    for (int e = 0; e < V; e++) {
      const float xh = (xv[e] - mu[e]) * is[e];
      float g = gv[e];
      if (relu && !(xh * ga[e] + bi[e] > 0.f)) g = 0.f;
      s1[e] += g; s2[e] += g * xh;
    }
  };
  long long pix = p0 + pl;                                   // same pixel order as before (sums unchanged), eight loads in flight
  for (; pix + 3ll * lanes_p < p1; pix += 4ll * lanes_p) {
    u32x4 rx[4], rg[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { rx[i] = *(const u32x4*)(x + base + (pix + (long long)i * lanes_p) * C); rg[i] = *(const u32x4*)(dy + base + (pix + (long long)i * lanes_p) * C); }
#pragma unroll
    for (int i = 0; i < 4; i++) one(rx[i], rg[i]);
  }
  for (; pix < p1; pix += lanes_p) one(*(const u32x4*)(x + base + pix * C), *(const u32x4*)(dy + base + pix * C));
#pragma unroll
  for (int e = 0; e < V; e++) { smf[((long long)pl * C + cv * V + e) * 2] = s1[e]; smf[((long long)pl * C + cv * V + e) * 2 + 1] = s2[e]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    float a = 0.f;
    for (int l = 0; l < lanes_p; l++) a += smf[(long long)l * C * 2 + i];
    unsafeAtomicAdd(sums + (long long)n * C * 2 + i, a);
  }
}
extern "C" int sg_bn_bwd_reduce(int dtype, const void* x, const void* dy, int N, long long HW, int C, const float* mean, const float* invstd, const float* gain, const float* bias, int gb_stride_n, int relu, float* sums, sg_stream_t s) {
  SgProfScope prof((hipStream_t)s, 2.0 * N * (double)HW * C * (dtype == SG_DTYPE_BF16 ? 2.0 : 4.0), 4);
  SG_CHECK(x && dy && mean && invstd && sums, "sg_bn_bwd_reduce: null");
  SG_CHECK(N <= 65535, "sg_bn_bwd_reduce: batch too large for grid.z");
  bool done = false;
  DISPATCH_T(dtype, {
    constexpr int V = ET<T>::VEC;
    const int CV = C / V;
    if (vec_ok<T>(x, dy, C) && CV <= 256 && HW >= 16) {
      const int lanes_p = 256 / CV;
      long long chunks = 2048 / N; if (chunks < 1) chunks = 1;
      long long ppb = (HW + chunks - 1) / chunks; if (ppb < 8 * lanes_p) ppb = 8 * lanes_p;
      const int gx = (int)((HW + ppb - 1) / ppb);
      hipLaunchKernelGGL(k_bn_bwd_reduce_stream<T>, dim3(gx, N), dim3(CV * lanes_p), (size_t)lanes_p * C * 2 * sizeof(float), (hipStream_t)s, (const T*)x, (const T*)dy, HW, C, mean, invstd, gain, bias, gb_stride_n, relu, sums, (int)ppb);
      done = true;
    }
  });
  if (done) { SG_LAUNCH_CHECK(); return 0; }
  int ct = (C + 63) / 64;
  long long want = 2048 / ((long long)ct * N); if (want < 1) want = 1;
  long long rpb = (HW + want - 1) / want; if (rpb < 16) rpb = 16;
  int gy = (int)((HW + rpb - 1) / rpb);
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_bn_bwd_reduce<T>, dim3(ct, gy, N), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)dy, HW, C, mean, invstd, gain, bias, gb_stride_n, relu, sums, rpb));
  SG_LAUNCH_CHECK();
  return 0;
}
// grid (C / 64), block (64 channels x 16 sample groups): the per-sample partial sums of sg_bn_bwd_reduce -> per-channel fp64 terms of
// the batch-statistics gradient, plus the (conditional) gain / bias gradients. One thread per channel walking all N samples (the
// first version) was a 256-long dependent chain per launch: 226 us average in the r01 trace for a few KB of data.
__global__ __launch_bounds__(1024) void k_bn_bwd_finalize(const float* sums, int N, int C, const float* gain, int gsn, float* dgain, float* dbias, double* chan) {
  __shared__ double sa0[16][64], sa1[16][64];
  __shared__ float sg0[16][64], sg1[16][64];
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double a0 = 0.0, a1 = 0.0; float g0 = 0.f, g1 = 0.f;
  if (c < C) {
    for (int n = grp; n < N; n += 16) {
      const float s1 = sums[((long long)n * C + c) * 2], s2 = sums[((long long)n * C + c) * 2 + 1];
      const float ga = gain ? gain[(long long)n * gsn + c] : 1.f;
      a0 += (double)ga * s1; a1 += (double)ga * s2;
      if (gsn) {     // per-sample gain / bias gradients at the row pitch of gain / bias themselves (gsn == C, or 2 C for the packed [gain | bias] rows of a cBN)
        if (dgain) dgain[(long long)n * gsn + c] += s2;
        if (dbias) dbias[(long long)n * gsn + c] += s1;
      } else { g0 += s1; g1 += s2; }
    }
  }
  sa0[grp][cl] = a0; sa1[grp][cl] = a1; sg0[grp][cl] = g0; sg1[grp][cl] = g1;
  __syncthreads();
  if (grp == 0 && c < C) {
    a0 = 0.0; a1 = 0.0; g0 = 0.f; g1 = 0.f;
    for (int g = 0; g < 16; g++) { a0 += sa0[g][cl]; a1 += sa1[g][cl]; g0 += sg0[g][cl]; g1 += sg1[g][cl]; }   // fixed order: deterministic
    if (!gsn) { if (dgain) dgain[c] += g1; if (dbias) dbias[c] += g0; }
    chan[2 * c] = a0; chan[2 * c + 1] = a1;
  }
}
extern "C" int sg_bn_bwd_finalize(const float* sums, int N, int C, const float* gain, int gb_stride_n, float* dgain, float* dbias, double* chan, sg_stream_t s) {
  SG_CHECK(sums && chan, "sg_bn_bwd_finalize: null");
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)s, sums, N, C, gain, gb_stride_n, dgain, dbias, chan);
  SG_LAUNCH_CHECK();
  return 0;
}
// res (optional): a tensor of dx's shape ADDED to the result -- the gradient the same input received through another branch (a residual block's
// skip path), so that the sum autograd would run as its own elementwise launch rides in this one.
template <typename T, bool VECP> __global__ __launch_bounds__(256) void k_bn_bwd_apply(const T* x, const T* dy, T* dx, int N, long long HW, int C, const float* mean, const float* invstd, const float* gain, const float* bias, int gsn, int relu, const double* chan, double count, int use_batch, const T* res) {
  constexpr int V = VECP ? ET<T>::VEC : 1;
  const int CV = C / V;
  const long long total = (long long)N * HW * CV;
  const float invc = (float)(1.0 / count);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cv = (int)(i % CV); const long long pix = i / CV; const int n = (int)(pix / HW);
    const int c0 = cv * V;
    float xv[V], gv[V];
    if (VECP) { unpack16<T>(*(const u32x4*)(x + pix * C + c0), xv); unpack16<T>(*(const u32x4*)(dy + pix * C + c0), gv); }
    else { xv[0] = to_f<T>(x[pix * C + c0]); gv[0] = to_f<T>(dy[pix * C + c0]); }
#pragma unroll
    for (int e = 0; e < V; e++) {
      const int c = c0 + e;
      const float is = invstd[c];
      const float xh = (xv[e] - mean[c]) * is;
      const float ga = gain ? gain[(long long)n * gsn + c] : 1.f;
      const float bi = bias ? bias[(long long)n * gsn + c] : 0.f;
      float g = gv[e];
      if (relu && !(xh * ga + bi > 0.f)) g = 0.f;
      float d = ga * g;
      if (use_batch) d -= ((float)chan[2 * c] + xh * (float)chan[2 * c + 1]) * invc;
      xv[e] = d * is;
    }
    if (res) {
      float rv[V];
      if (VECP) unpack16<T>(*(const u32x4*)(res + pix * C + c0), rv); else rv[0] = to_f<T>(res[pix * C + c0]);
#pragma unroll
      for (int e = 0; e < V; e++) xv[e] += rv[e];
    }
    if (VECP) *(u32x4*)(dx + pix * C + c0) = pack16<T>(xv); else dx[pix * C + c0] = from_f<T>(xv[0]);
  }
}
// Streaming variant (same ownership as k_bn_apply_stream): a thread keeps the 6 per-channel coefficients of its 16-byte
// channel vector in registers; the loop body is 2 loads -> ~6 flops per element -> 1 store.
template <typename T> __global__ __launch_bounds__(256) void k_bn_bwd_apply_stream(const T* x, const T* dy, T* dx, long long HW, int C, const float* mean, const float* invstd, const float* gain, const float* bias, int gsn, int relu, const double* chan, double count, int use_batch, int ppb, const T* res) {
  constexpr int V = ET<T>::VEC;
  const int CV = C / V;
  const int n = blockIdx.y;
  const int lanes_p = blockDim.x / CV;
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  const float invc = (float)(1.0 / count);
  float mu[V], is[V], ga[V], bi[V], k0[V], k1[V];
#pragma unroll
  for (int e = 0; e < V; e++) {
    const int c = cv * V + e;
    mu[e] = mean[c]; is[e] = invstd[c];
    ga[e] = gain ? gain[(long long)n * gsn + c] : 1.f;
    bi[e] = bias ? bias[(long long)n * gsn + c] : 0.f;
    k0[e] = use_batch ? (float)chan[2 * c] : 0.f;
    k1[e] = use_batch ? (float)chan[2 * c + 1] : 0.f;
  }
  const long long p0 = (long long)blockIdx.x * ppb;
  long long p1 = p0 + ppb; if (p1 > HW) p1 = HW;
  const long long base = (long long)n * HW * C + cv * V;
  auto one = [&](const u32x4& rx, const u32x4& rg, long long pix) {
    float xv[V], gv[V];
    unpack16<T>(rx, xv);
    unpack16<T>(rg, gv);
#pragma unroll
    for (int e = 0; e < V; e++) {
      const float xh = (xv[e] - mu[e]) * is[e];
      float g = gv[e];
      if (relu && !(xh * ga[e] + bi[e] > 0.f)) g = 0.f;
      float d = ga[e] * g;
      if (use_batch) d -= (k0[e] + xh * k1[e]) * invc;     // same evaluation order as k_bn_bwd_apply
      xv[e] = d * is[e];
    }
    if (res) {                                                // the skip path's gradient w.r.t. the same input (see k_bn_bwd_apply)
      float rv[V];
      unpack16<T>(*(const u32x4*)(res + base + pix * C), rv);
#pragma unroll
      for (int e = 0; e < V; e++) xv[e] += rv[e];
    }
    *(u32x4*)(dx + base + pix * C) = pack16<T>(xv);
  };
  long long pix = p0 + pl;                                   // four pixels = eight loads in flight per lane (see k_bn_apply_stream)
  for (; pix + 3ll * lanes_p < p1; pix += 4ll * lanes_p) {
    u32x4 rx[4], rg[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { rx[i] = *(const u32x4*)(x + base + (pix + (long long)i * lanes_p) * C); rg[i] = *(const u32x4*)(dy + base + (pix + (long long)i * lanes_p) * C); }
#pragma unroll
    for (int i = 0; i < 4; i++) one(rx[i], rg[i], pix + (long long)i * lanes_p);
  }
  for (; pix < p1; pix += lanes_p) one(*(const u32x4*)(x + base + pix * C), *(const u32x4*)(dy + base + pix * C), pix);
}
extern "C" int sg_bn_bwd_apply(int dtype, const void* x, const void* dy, void* dx, int N, long long HW, int C, const float* mean, const float* invstd, const float* gain, const float* bias, int gb_stride_n, int relu, const double* chan, double count, int use_batch_stats, sg_stream_t s) {
  return sg_bn_bwd_apply_res(dtype, x, dy, dx, N, HW, C, mean, invstd, gain, bias, gb_stride_n, relu, chan, count, use_batch_stats, nullptr, s);
}
extern "C" int sg_bn_bwd_apply_res(int dtype, const void* x, const void* dy, void* dx, int N, long long HW, int C, const float* mean, const float* invstd, const float* gain, const float* bias, int gb_stride_n, int relu, const double* chan, double count, int use_batch_stats, const void* res, sg_stream_t s) {
  SgProfScope prof((hipStream_t)s, (res ? 4.0 : 3.0) * N * (double)HW * C * (dtype == SG_DTYPE_BF16 ? 2.0 : 4.0), 4);
  SG_CHECK(x && dy && dx && mean && invstd && chan && count > 0, "sg_bn_bwd_apply: bad args");
  SG_CHECK(!res || ((((uintptr_t)res) & 15) == 0), "sg_bn_bwd_apply_res: res must be 16-byte aligned");
  DISPATCH_T(dtype, {
    const int CV = C / ET<T>::VEC;
    if (vec_ok<T>(x, dy, C) && ((((uintptr_t)dx) & 15) == 0) && CV <= 256 && N <= 65535 && HW >= 16) {
      const int lanes_p = 256 / CV;
      long long chunks = 2048 / N; if (chunks < 1) chunks = 1;
      long long ppb = (HW + chunks - 1) / chunks; if (ppb < 4 * lanes_p) ppb = 4 * lanes_p;
      const int gx = (int)((HW + ppb - 1) / ppb);
      hipLaunchKernelGGL(k_bn_bwd_apply_stream<T>, dim3(gx, N), dim3(CV * lanes_p), 0, (hipStream_t)s, (const T*)x, (const T*)dy, (T*)dx, HW, C, mean, invstd, gain, bias, gb_stride_n, relu, chan, count, use_batch_stats, (int)ppb, (const T*)res);
    } else if (vec_ok<T>(x, dy, C) && ((((uintptr_t)dx) & 15) == 0)) hipLaunchKernelGGL((k_bn_bwd_apply<T, true>), dim3(grid_for((long long)N * HW * C / ET<T>::VEC)), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)dy, (T*)dx, N, HW, C, mean, invstd, gain, bias, gb_stride_n, relu, chan, count, use_batch_stats, (const T*)res);
    else hipLaunchKernelGGL((k_bn_bwd_apply<T, false>), dim3(grid_for((long long)N * HW * C)), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)dy, (T*)dx, N, HW, C, mean, invstd, gain, bias, gb_stride_n, relu, chan, count, use_batch_stats, (const T*)res);
  });
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- second-order backward (WGAN-GP: gradient of the gradient penalty through BN's data gradient) -------------------
// First order (per channel, gain g_c, r = invstd, N = count, g = dy' after the ReLU mask):
//     dx = g_c r ( g - mean(g) - xhat mean(g xhat) )
// Given u = dL/d(dx), with  ub = mean(u), a = mean(u xhat), gb = mean(g), c = mean(g xhat), m = mean(u g):
//     dL/d(dy) = mask * g_c r ( u - ub - xhat a )
//     dL/dx    = g_c r^2 ( -c (u - ub) - a (g - gb) + xhat (3 a c - m + ub gb) )
//     dL/dg_c  = r N ( m - ub gb - a c )                       (reference: torch autograd through F.batch_norm's backward,
//                                                               reached from utils/losses.py:268-275,301-316)
// stage 1: sums[n][c][5] = {sum u, sum u xhat, sum g, sum g xhat, sum u g}; caller zeroes sums
template <typename T> __global__ __launch_bounds__(256) void k_bn_bwd2_reduce(const T* x, const T* dy, const T* u, long long HW, int C, const float* mean, const float* invstd, const float* gain, const float* bias, int relu, float* sums, long long rpb) {
  __shared__ float sm[5][4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const int n = blockIdx.z;
  long long r0 = blockIdx.y * rpb, r1 = r0 + rpb; if (r1 > HW) r1 = HW;
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const float mu = mean[c], is = invstd[c];
    const float ga = gain ? gain[c] : 1.f;
    const float bi = bias ? bias[c] : 0.f;
    const long long base = (long long)n * HW * C + c;
    for (long long r = r0 + ry; r < r1; r += 4) {
      const float xh = (to_f<T>(x[base + r * C]) - mu) * is;
      float g = to_f<T>(dy[base + r * C]);
      if (relu && !(xh * ga + bi > 0.f)) g = 0.f;
      const float uu = to_f<T>(u[base + r * C]);
      s[0] += uu; s[1] += uu * xh; s[2] += g; s[3] += g * xh; s[4] += uu * g;
    }
  }
#pragma unroll
  for (int k = 0; k < 5; k++) sm[k][ry][cx] = s[k];
  __syncthreads();
  if (ry == 0 && c < C) {
    float* o = sums + ((long long)n * C + c) * 5;
#pragma unroll
    for (int k = 0; k < 5; k++) unsafeAtomicAdd(o + k, sm[k][0][cx] + sm[k][1][cx] + sm[k][2][cx] + sm[k][3][cx]);
  }
}
extern "C" int sg_bn_bwd2_reduce(int dtype, const void* x, const void* dy, const void* u, int N, long long HW, int C, const float* mean, const float* invstd, const float* gain, const float* bias, int relu, float* sums, sg_stream_t s) {
  SG_CHECK(x && dy && u && mean && invstd && sums, "sg_bn_bwd2_reduce: null");
  SG_CHECK(N <= 65535, "sg_bn_bwd2_reduce: batch too large for grid.z");
  int ct = (C + 63) / 64;
  long long want = 2048 / ((long long)ct * N); if (want < 1) want = 1;
  long long rpb = (HW + want - 1) / want; if (rpb < 16) rpb = 16;
  int gy = (int)((HW + rpb - 1) / rpb);
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_bn_bwd2_reduce<T>, dim3(ct, gy, N), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)dy, (const T*)u, HW, C, mean, invstd, gain, bias, relu, sums, rpb));
  SG_LAUNCH_CHECK();
  return 0;
}
// stage 2: chan[c][5] (fp64) = sum over n
__global__ void k_bn_bwd2_finalize(const float* sums, int N, int C, double* chan) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a[5] = {0, 0, 0, 0, 0};
  for (int n = 0; n < N; n++)
    for (int k = 0; k < 5; k++) a[k] += (double)sums[((long long)n * C + c) * 5 + k];
  for (int k = 0; k < 5; k++) chan[5 * c + k] = a[k];
}
extern "C" int sg_bn_bwd2_finalize(const float* sums, int N, int C, double* chan, sg_stream_t s) {
  SG_CHECK(sums && chan, "sg_bn_bwd2_finalize: null");
  hipLaunchKernelGGL(k_bn_bwd2_finalize, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)s, sums, N, C, chan);
  SG_LAUNCH_CHECK();
  return 0;
}
// gain gradient from this rank's own sums (chan_local) and the statistics of the whole (possibly cross-rank) batch (chan)
__global__ void k_bn_bwd2_dgain(const double* chan_local, const double* chan, double count, const float* invstd, int C, int use_batch, float* dgain) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double r = invstd[c];
  double v;
  if (use_batch) {
    const double gb = chan[5 * c + 2] / count, cc = chan[5 * c + 3] / count;
    v = r * (chan_local[5 * c + 4] - gb * chan_local[5 * c + 0] - cc * chan_local[5 * c + 1]);
  } else v = r * chan_local[5 * c + 4];
  dgain[c] += (float)v;
}
extern "C" int sg_bn_bwd2_dgain(const double* chan_local, const double* chan, double count, const float* invstd, int C, int use_batch_stats, float* dgain, sg_stream_t s) {
  SG_CHECK(chan_local && chan && invstd && dgain && count > 0, "sg_bn_bwd2_dgain: bad args");
  hipLaunchKernelGGL(k_bn_bwd2_dgain, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)s, chan_local, chan, count, invstd, C, use_batch_stats, dgain);
  SG_LAUNCH_CHECK();
  return 0;
}
// stage 3: elementwise; g_dy and/or g_x may be null
template <typename T> __global__ __launch_bounds__(256) void k_bn_bwd2_apply(const T* x, const T* dy, const T* u, T* g_dy, T* g_x, long long total, int C, const float* mean, const float* invstd, const float* gain, const float* bias, int relu, const double* chan, double count, int use_batch) {
  const float invc = (float)(1.0 / count);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float is = invstd[c];
    const float xh = (to_f<T>(x[i]) - mean[c]) * is;
    const float ga = gain ? gain[c] : 1.f;
    const float bi = bias ? bias[c] : 0.f;
    const bool on = !(relu && !(xh * ga + bi > 0.f));
    const float g = on ? to_f<T>(dy[i]) : 0.f;
    const float uu = to_f<T>(u[i]);
    float ub = 0.f, a = 0.f, gb = 0.f, cc = 0.f, m = 0.f;
    if (use_batch) {
      ub = (float)chan[5 * c] * invc; a = (float)chan[5 * c + 1] * invc; gb = (float)chan[5 * c + 2] * invc;
      cc = (float)chan[5 * c + 3] * invc; m = (float)chan[5 * c + 4] * invc;
    }
    if (g_dy) g_dy[i] = from_f<T>(on ? ga * is * (uu - ub - xh * a) : 0.f);
    if (g_x) g_x[i] = from_f<T>(use_batch ? ga * is * is * (-cc * (uu - ub) - a * (g - gb) + xh * (3.f * a * cc - m + ub * gb)) : 0.f);
  }
}
extern "C" int sg_bn_bwd2_apply(int dtype, const void* x, const void* dy, const void* u, void* g_dy, void* g_x, int N, long long HW, int C, const float* mean, const float* invstd, const float* gain, const float* bias, int relu, const double* chan, double count, int use_batch_stats, sg_stream_t s) {
  SG_CHECK(x && dy && u && mean && invstd && chan && count > 0 && (g_dy || g_x), "sg_bn_bwd2_apply: bad args");
  const long long total = (long long)N * HW * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_bn_bwd2_apply<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)dy, (const T*)u, (T*)g_dy, (T*)g_x, total, C, mean, invstd, gain, bias, relu, chan, count, use_batch_stats));
  SG_LAUNCH_CHECK();
  return 0;
}
