// elementwise.hip -- HBM-bound glue of the hot path: layout changes at the NCHW boundary, pooling, softmax,
// reductions, embedding gather/scatter, the projection-discriminator head and the adversarial losses.
// All kernels are coalesced along the channel (fastest NHWC) dimension; wave = 64.
#include "common.h"
#include "../../include/sgamd.h"

static inline int nblk(long long n, int bs) {
  long long b = (n + bs - 1) / bs;
  if (b > 65535ll * 32) b = 65535ll * 32;
  if (b < 1) b = 1;
  return (int)b;
}
#define DISPATCH_T(dtype, ...)                                          \
  if ((dtype) == SG_DTYPE_F32) { typedef float T; __VA_ARGS__; }        \
  else if ((dtype) == SG_DTYPE_BF16) { typedef bf16_t T; __VA_ARGS__; } \
  else { sg_set_error("bad dtype"); return -1; }

// ---- layout -------------------------------------------------------------------------------------------
template <typename T> __global__ void k_nchw_to_nhwc(const float* src, T* dst, int N, int C, int HW, int ldo) {
  long long total = (long long)N * HW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    // iterate in source order (coalesced fp32 reads); C is small (3) at the image boundary
    int hw = (int)(i % HW); long long t = i / HW; int c = (int)(t % C); int n = (int)(t / C);
    dst[((long long)n * HW + hw) * ldo + c] = from_f<T>(src[i]);
  }
}
// padded rows (ldo > C: the RGB image as 8-channel pixels): a thread owns one PIXEL -- C coalesced plane reads, one contiguous row store that carries the zero
// padding, so the destination needs no fill launch in front (round 6: the scattered 2-byte stores + the fill were 54 us per image batch, seven times per C3 step)
template <typename T> __global__ void k_nchw_to_nhwc_pad(const float* src, T* dst, int N, int C, int HW, int ldo) {
  const long long total = (long long)N * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int hw = (int)(i % HW); const long long n = i / HW;
    const float* sp = src + n * C * HW + hw;
    T* dp = dst + i * ldo;
    if (sizeof(T) == 2 && ldo == 8 && C <= 8) {
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; c++) v[c] = c < C ? sp[(long long)c * HW] : 0.f;
      u32x4 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
      *(u32x4*)dp = o;
    } else {
      for (int c = 0; c < ldo; c++) dp[c] = from_f<T>(c < C ? sp[(long long)c * HW] : 0.f);
    }
  }
}
extern "C" int sg_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int H, int W, int ldo, sg_stream_t s) {
  SG_CHECK(src && dst && ldo >= C, "sg_nchw_to_nhwc: null / ldo < C");
  long long total = (long long)N * C * H * W;
  if (ldo > C) {      // every one of the ldo channels of a row is written (zeros behind C)
    DISPATCH_T(dtype, hipLaunchKernelGGL(k_nchw_to_nhwc_pad<T>, dim3(nblk((long long)N * H * W, 256)), dim3(256), 0, (hipStream_t)s, src, (T*)dst, N, C, H * W, ldo));
    SG_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_nchw_to_nhwc<T>, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)s, src, (T*)dst, N, C, H * W, ldo));
  SG_LAUNCH_CHECK();
  return 0;
}
template <typename T> __global__ void k_nhwc_to_nchw(const T* src, float* dst, int N, int C, int HW, int lds, int do_tanh) {
  long long total = (long long)N * HW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int hw = (int)(i % HW); long long t = i / HW; int c = (int)(t % C); int n = (int)(t / C);
    float v = to_f<T>(src[((long long)n * HW + hw) * lds + c]);
    if (do_tanh) v = tanhf(v);
    dst[i] = v;
  }
}
extern "C" int sg_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int H, int W, int lds, int apply_tanh, sg_stream_t s) {
  SG_CHECK(src && dst, "sg_nhwc_to_nchw: null");
  long long total = (long long)N * C * H * W;
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_nhwc_to_nchw<T>, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)s, (const T*)src, dst, N, C, H * W, lds, apply_tanh));
  SG_LAUNCH_CHECK();
  return 0;
}
template <typename T> __global__ void k_nchw_grad_to_nhwc(const float* dy, const float* y, T* dst, int N, int C, int HW, int ldo, int do_tanh) {
  long long total = (long long)N * HW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int hw = (int)(i % HW); long long t = i / HW; int c = (int)(t % C); int n = (int)(t / C);
    float g = dy[i];
    if (do_tanh) { float yy = y[i]; g *= (1.f - yy * yy); }
    dst[((long long)n * HW + hw) * ldo + c] = from_f<T>(g);
  }
}
extern "C" int sg_nchw_grad_to_nhwc(int dtype, const float* dy, const float* y, void* dst, int N, int C, int H, int W, int ldo, int apply_tanh, sg_stream_t s) {
  SG_CHECK(dy && dst && (y || !apply_tanh) && ldo >= C, "sg_nchw_grad_to_nhwc: bad args");
  long long total = (long long)N * C * H * W;
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_nchw_grad_to_nhwc<T>, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)s, dy, y, (T*)dst, N, C, H * W, ldo, apply_tanh));
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- 2x2 pooling ---------------------------------------------------------------------------------------
template <typename T> __global__ void k_avgpool2_fwd(const T* x, T* y, int N, int H, int W, int C) {
  const int H2 = H / 2, W2 = W / 2;
  long long total = (long long)N * H2 * W2 * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C); long long t = i / C; int w2 = (int)(t % W2); t /= W2; int h2 = (int)(t % H2); int n = (int)(t / H2);
    const T* p = x + (((long long)n * H + 2 * h2) * W + 2 * w2) * C + c;
    float v = to_f<T>(p[0]) + to_f<T>(p[C]) + to_f<T>(p[(long long)W * C]) + to_f<T>(p[(long long)W * C + C]);
    y[i] = from_f<T>(0.25f * v);
  }
}
// ---- 16-byte bf16 variants of the elementwise layout kernels (round 2) ------------------------------------------------------------------
// The scalar kernels move 2 bytes per thread behind three or four 64-bit divisions; they were 20 % of BigGAN-deep's kernel time
// (k_relu_mask 10.6 %, k_avgpool2_bwd 3.9 %, k_slice_up_fwd 2.9 %, k_avgpool2_fwd 1.3 %: gpurun session O) and 2.5 ms of the C3 step (max-pool).
// A thread owns 8 channels of one pixel; same arithmetic in the same order as the scalar kernels -> identical results.
static inline bool ew_v8_ok(int dtype, const void* a, const void* b, int C, int lda, int ldb, long long total_vec) {
  return dtype == SG_DTYPE_BF16 && C % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0 && total_vec > 0 && total_vec < (1ll << 31);
}
__global__ __launch_bounds__(256) void k_avgpool2_fwd_v8(const bf16_t* x, bf16_t* y, int N, int H, int W, int C) {
  const int H2 = H / 2, W2 = W / 2, CV = C / 8;
  const unsigned total = (unsigned)N * H2 * W2 * CV;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned q = i / CV, cv = i - q * CV;
    const unsigned w2 = q % W2, t = q / W2, h2 = t % H2, n = t / H2;
    const bf16_t* p = x + (((long long)n * H + 2 * h2) * W + 2 * w2) * C + cv * 8;
    float a[8], b[8], c[8], d[8];
    unpack16<bf16_t>(*(const u32x4*)p, a); unpack16<bf16_t>(*(const u32x4*)(p + C), b);
    unpack16<bf16_t>(*(const u32x4*)(p + (long long)W * C), c); unpack16<bf16_t>(*(const u32x4*)(p + (long long)W * C + C), d);
#pragma unroll
    for (int e = 0; e < 8; e++) a[e] = 0.25f * (((a[e] + b[e]) + c[e]) + d[e]);
    *(u32x4*)(y + (long long)q * C + cv * 8) = pack16<bf16_t>(a);
  }
}
__global__ __launch_bounds__(256) void k_avgpool2_bwd_v8(const bf16_t* dy, bf16_t* dx, int N, int H, int W, int C) {
  const int H2 = H / 2, W2 = W / 2, CV = C / 8;
  const unsigned total = (unsigned)N * H * W * CV;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned pix = i / CV, cv = i - pix * CV;
    const unsigned w = pix % W, t = pix / W, h = t % H, n = t / H;
    float g[8];
    unpack16<bf16_t>(*(const u32x4*)(dy + (((long long)n * H2 + (h >> 1)) * W2 + (w >> 1)) * C + cv * 8), g);
#pragma unroll
    for (int e = 0; e < 8; e++) g[e] = 0.25f * g[e];
    *(u32x4*)(dx + (long long)pix * C + cv * 8) = pack16<bf16_t>(g);
  }
}
extern "C" int sg_avgpool2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, sg_stream_t s) {
  SG_CHECK(x && y && H % 2 == 0 && W % 2 == 0, "sg_avgpool2_fwd: bad args");
  long long total = (long long)N * (H / 2) * (W / 2) * C;
  if (ew_v8_ok(dtype, x, y, C, 8, 8, total / 8)) {
    hipLaunchKernelGGL(k_avgpool2_fwd_v8, dim3(nblk(total / 8, 256)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (bf16_t*)y, N, H, W, C);
    SG_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_avgpool2_fwd<T>, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)s, (const T*)x, (T*)y, N, H, W, C));
  SG_LAUNCH_CHECK();
  return 0;
}
template <typename T> __global__ void k_avgpool2_bwd(const T* dy, T* dx, int N, int H, int W, int C) {
  // H, W are the dims of dx
  const int H2 = H / 2, W2 = W / 2;
  long long total = (long long)N * H * W * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C); long long t = i / C; int w = (int)(t % W); t /= W; int h = (int)(t % H); int n = (int)(t / H);
    float g = to_f<T>(dy[(((long long)n * H2 + (h >> 1)) * W2 + (w >> 1)) * C + c]);
    dx[i] = from_f<T>(0.25f * g);
  }
}
extern "C" int sg_avgpool2_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, sg_stream_t s) {
  SG_CHECK(dy && dx && H % 2 == 0 && W % 2 == 0, "sg_avgpool2_bwd: bad args");
  long long total = (long long)N * H * W * C;
  if (ew_v8_ok(dtype, dy, dx, C, 8, 8, total / 8)) {
    hipLaunchKernelGGL(k_avgpool2_bwd_v8, dim3(nblk(total / 8, 256)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C);
    SG_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_avgpool2_bwd<T>, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)s, (const T*)dy, (T*)dx, N, H, W, C));
  SG_LAUNCH_CHECK();
  return 0;
}
template <typename T> __global__ void k_maxpool2_fwd(const T* x, int ldx, T* y, int ldy, uint8_t* idx, int N, int H, int W, int C) {
  const int H2 = H / 2, W2 = W / 2;
  long long total = (long long)N * H2 * W2 * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C); long long q = i / C; long long t = q; int w2 = (int)(t % W2); t /= W2; int h2 = (int)(t % H2); int n = (int)(t / H2);
    const T* p = x + (((long long)n * H + 2 * h2) * W + 2 * w2) * ldx + c;
    // first maximum in (dy,dx) row-major order wins, like torch's max_pool2d
    float v0 = to_f<T>(p[0]), v1 = to_f<T>(p[ldx]), v2 = to_f<T>(p[(long long)W * ldx]), v3 = to_f<T>(p[(long long)W * ldx + ldx]);
    float m = v0; int a = 0;
    if (v1 > m) { m = v1; a = 1; }
    if (v2 > m) { m = v2; a = 2; }
    if (v3 > m) { m = v3; a = 3; }
    y[q * ldy + c] = from_f<T>(m);
    if (idx) idx[i] = (uint8_t)a;
  }
}
// bf16, C % 8 == 0, 16-byte aligned rows: a thread owns 8 channels of one output pixel -- four 16-byte loads, one 16-byte store, 8 index
// bytes; 32-bit index arithmetic (the scalar kernel: 2 bytes per thread and four 64-bit divisions per element, 0.9 + 1.6 ms per step)
__global__ __launch_bounds__(256) void k_maxpool2_fwd_v8(const bf16_t* x, int ldx, bf16_t* y, int ldy, uint8_t* idx, int N, int H, int W, int C) {
  const int H2 = H / 2, W2 = W / 2, CV = C / 8;
  const unsigned total = (unsigned)N * H2 * W2 * CV;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned q = i / CV, cv = i - q * CV;
    const unsigned w2 = q % W2, t = q / W2, h2 = t % H2, n = t / H2;
    const bf16_t* p = x + (((long long)n * H + 2 * h2) * W + 2 * w2) * ldx + cv * 8;
    const u32x4 r0 = *(const u32x4*)p, r1 = *(const u32x4*)(p + ldx), r2 = *(const u32x4*)(p + (long long)W * ldx), r3 = *(const u32x4*)(p + (long long)W * ldx + ldx);
    float v0[8], v1[8], v2[8], v3[8], m[8];
    unpack16<bf16_t>(r0, v0); unpack16<bf16_t>(r1, v1); unpack16<bf16_t>(r2, v2); unpack16<bf16_t>(r3, v3);
    uint32_t a_lo = 0, a_hi = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      // first maximum in (dy,dx) row-major order wins, like torch's max_pool2d
      float mm = v0[e]; uint32_t a = 0;
      if (v1[e] > mm) { mm = v1[e]; a = 1; }
      if (v2[e] > mm) { mm = v2[e]; a = 2; }
      if (v3[e] > mm) { mm = v3[e]; a = 3; }
      m[e] = mm;
      if (e < 4) a_lo |= a << (8 * e); else a_hi |= a << (8 * (e - 4));
    }
    *(u32x4*)(y + (long long)q * ldy + cv * 8) = pack16<bf16_t>(m);
    if (idx) { u32x2 av = {a_lo, a_hi}; *(u32x2*)(idx + (long long)q * C + cv * 8) = av; }
  }
}
__global__ __launch_bounds__(256) void k_maxpool2_bwd_v8(const bf16_t* dy, int ldy, const uint8_t* idx, bf16_t* dx, int ldx, int N, int H, int W, int C) {
  const int H2 = H / 2, W2 = W / 2, CV = C / 8;
  const unsigned total = (unsigned)N * H * W * CV;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned pix = i / CV, cv = i - pix * CV;
    const unsigned w = pix % W, t = pix / W, h = t % H, n = t / H;
    const long long q = ((long long)n * H2 + (h >> 1)) * W2 + (w >> 1);
    const uint32_t pos = ((h & 1) << 1) | (w & 1);
    const u32x2 av = *(const u32x2*)(idx + q * C + cv * 8);
    const u32x4 g = *(const u32x4*)(dy + q * ldy + cv * 8);
    u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const uint32_t a0 = ((d < 2 ? av[0] : av[1]) >> (16 * (d & 1))) & 0xffu, a1 = ((d < 2 ? av[0] : av[1]) >> (16 * (d & 1) + 8)) & 0xffu;
      o[d] = (a0 == pos ? (g[d] & 0xffffu) : 0u) | (a1 == pos ? (g[d] & 0xffff0000u) : 0u);
    }
    *(u32x4*)(dx + (long long)pix * ldx + cv * 8) = o;
  }
}
static inline bool pool_v8_ok(int dtype, const void* a, int lda, const void* b, int ldb, const void* idx, int C, long long total_vec) {
  return dtype == SG_DTYPE_BF16 && C % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0 && (((uintptr_t)idx) & 7) == 0 &&
         total_vec < (1ll << 31);
}
extern "C" int sg_maxpool2_fwd(int dtype, const void* x, int ldx, void* y, int ldy, uint8_t* idx, int N, int H, int W, int C, sg_stream_t s) {
  SG_CHECK(x && y && H % 2 == 0 && W % 2 == 0, "sg_maxpool2_fwd: bad args");
  long long total = (long long)N * (H / 2) * (W / 2) * C;
  if (pool_v8_ok(dtype, x, ldx, y, ldy, idx, C, total / 8)) {
    hipLaunchKernelGGL(k_maxpool2_fwd_v8, dim3(nblk(total / 8, 256)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, idx, N, H, W, C);
    SG_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_maxpool2_fwd<T>, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)s, (const T*)x, ldx, (T*)y, ldy, idx, N, H, W, C));
  SG_LAUNCH_CHECK();
  return 0;
}
template <typename T> __global__ void k_maxpool2_bwd(const T* dy, int ldy, const uint8_t* idx, T* dx, int ldx, int N, int H, int W, int C) {
  const int H2 = H / 2, W2 = W / 2;
  long long total = (long long)N * H * W * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C); long long pix = i / C; long long t = pix; int w = (int)(t % W); t /= W; int h = (int)(t % H); int n = (int)(t / H);
    long long q = ((long long)n * H2 + (h >> 1)) * W2 + (w >> 1);
    int pos = ((h & 1) << 1) | (w & 1);
    float g = (idx[q * C + c] == pos) ? to_f<T>(dy[q * ldy + c]) : 0.f;
    dx[pix * ldx + c] = from_f<T>(g);
  }
}
extern "C" int sg_maxpool2_bwd(int dtype, const void* dy, int ldy, const uint8_t* idx, void* dx, int ldx, int N, int H, int W, int C, sg_stream_t s) {
  SG_CHECK(dy && dx && idx, "sg_maxpool2_bwd: null");
  long long total = (long long)N * H * W * C;
  if (pool_v8_ok(dtype, dy, ldy, dx, ldx, idx, C, total / 8)) {
    hipLaunchKernelGGL(k_maxpool2_bwd_v8, dim3(nblk(total / 8, 256)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)dy, ldy, idx, (bf16_t*)dx, ldx, N, H, W, C);
    SG_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_maxpool2_bwd<T>, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)s, (const T*)dy, ldy, idx, (T*)dx, ldx, N, H, W, C));
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- row softmax (attention, reference src/utils/ops.py:94) ----------------------------------------------
// one 256-thread workgroup per row; logits arrive in fp32 from the QK^T epilogue, probabilities leave as T
template <typename T> __global__ __launch_bounds__(256) void k_softmax_rows(const float* s, T* p, int cols) {
  __shared__ float sm[4];
  const float* row = s + (long long)blockIdx.x * cols;
  T* out = p + (long long)blockIdx.x * cols;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, row[c]);
  m = block_max_256(m, sm);
  float sum = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) sum += __expf(row[c] - m);
  sum = block_sum_256(sum, sm);
  const float inv = 1.f / sum;
  for (int c = threadIdx.x; c < cols; c += 256) out[c] = from_f<T>(__expf(row[c] - m) * inv);
}
// single-pass variant for rows of exactly 1024 logits (the 64x64 attention of the 128-px models): each thread keeps
// its 4 logits in registers -> one 16-byte load and one 8/16-byte store per thread, 6 B/logit of HBM traffic
template <typename T> __global__ __launch_bounds__(256) void k_softmax_rows_1024(const float* s, T* p) {
  __shared__ float sm[4];
  const f32x4 v = ((const f32x4*)(s + (long long)blockIdx.x * 1024))[threadIdx.x];
  float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
  m = block_max_256(m, sm);
  float e[4];
#pragma unroll
  for (int i = 0; i < 4; i++) e[i] = __expf(v[i] - m);
  const float sum = block_sum_256(e[0] + e[1] + e[2] + e[3], sm);
  const float inv = 1.f / sum;
  T* o = p + (long long)blockIdx.x * 1024 + threadIdx.x * 4;
#pragma unroll
  for (int i = 0; i < 4; i++) o[i] = from_f<T>(e[i] * inv);
}
extern "C" int sg_softmax_rows(int dtype, const float* s_in, void* p_out, long long rows, int cols, sg_stream_t st) {
  SG_CHECK(s_in && p_out && rows > 0 && rows < (1ll << 31) && cols > 0, "sg_softmax_rows: bad args");
  if (cols == 1024 && ((((uintptr_t)s_in) | ((uintptr_t)p_out)) & 15) == 0) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(k_softmax_rows_1024<T>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)st, s_in, (T*)p_out));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL(k_softmax_rows<T>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)st, s_in, (T*)p_out, cols));
  }
  SG_LAUNCH_CHECK();
  return 0;
}
template <typename T> __global__ __launch_bounds__(256) void k_softmax_rows_bwd(const T* p, const float* dp, T* ds, int cols) {
  __shared__ float sm[4];
  const T* pr = p + (long long)blockIdx.x * cols;
  const float* dr = dp + (long long)blockIdx.x * cols;
  T* out = ds + (long long)blockIdx.x * cols;
  float dot = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) dot += to_f<T>(pr[c]) * dr[c];
  dot = block_sum_256(dot, sm);
  for (int c = threadIdx.x; c < cols; c += 256) out[c] = from_f<T>(to_f<T>(pr[c]) * (dr[c] - dot));
}
template <typename T> __global__ __launch_bounds__(256) void k_softmax_rows_bwd_1024(const T* p, const float* dp, T* ds) {
  __shared__ float sm[4];
  const long long base = (long long)blockIdx.x * 1024 + threadIdx.x * 4;
  const f32x4 g = *(const f32x4*)(dp + base);
  float pv[4];
#pragma unroll
  for (int i = 0; i < 4; i++) pv[i] = to_f<T>(p[base + i]);
  float dot = pv[0] * g[0] + pv[1] * g[1] + pv[2] * g[2] + pv[3] * g[3];
  dot = block_sum_256(dot, sm);
#pragma unroll
  for (int i = 0; i < 4; i++) ds[base + i] = from_f<T>(pv[i] * (g[i] - dot));
}
extern "C" int sg_softmax_rows_bwd(int dtype, const void* p, const float* dp, void* ds, long long rows, int cols, sg_stream_t st) {
  SG_CHECK(p && dp && ds && rows > 0 && rows < (1ll << 31) && cols > 0, "sg_softmax_rows_bwd: bad args");
  if (cols == 1024 && (((uintptr_t)dp) & 15) == 0) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(k_softmax_rows_bwd_1024<T>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)st, (const T*)p, dp, (T*)ds));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL(k_softmax_rows_bwd<T>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)st, (const T*)p, dp, (T*)ds, cols));
  }
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- axpby / dot / column sums ---------------------------------------------------------------------------
template <typename T> __global__ void k_axpby(const T* x, T* y, long long n, float a, float b) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = a * to_f<T>(x[i]);
    if (b != 0.f) v += b * to_f<T>(y[i]);
    y[i] = from_f<T>(v);
  }
}
extern "C" int sg_axpby(int dtype, const void* x, void* y, long long n, float a, float b, sg_stream_t s) {
  SG_CHECK(x && y, "sg_axpby: null");
  if (n <= 0) return 0;
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_axpby<T>, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)s, (const T*)x, (T*)y, n, a, b));
  SG_LAUNCH_CHECK();
  return 0;
}
template <typename T> __global__ __launch_bounds__(256) void k_dot(const T* x, const T* y, long long n, float* out, float scale, const float* scale_ptr) {
  __shared__ float sm[4];
  float acc = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc += to_f<T>(x[i]) * to_f<T>(y[i]);
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) {
    float sc = scale; if (scale_ptr) sc *= *scale_ptr;
    unsafeAtomicAdd(out, acc * sc);
  }
}
// 16-byte loads, four pairs in flight per thread (the element-wise walk above moved the two 201 MB operands of the attention gate's gradient at
// 1.6 TB/s: 252 us per call, five calls per C3 step in profiles/r03_bench_biggan128_bs256_kerneltrace_final.txt)
template <typename T> __global__ __launch_bounds__(256) void k_dot_vec(const T* x, const T* y, long long nvec, float* out, float scale, const float* scale_ptr) {
  constexpr int V = ET<T>::VEC;
  __shared__ float sm[4];
  const long long stride = (long long)gridDim.x * 256;
  float acc = 0.f;
  long long i = blockIdx.x * 256ll + threadIdx.x;
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    u32x4 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { a[k] = *(const u32x4*)(x + (i + k * stride) * V); b[k] = *(const u32x4*)(y + (i + k * stride) * V); }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float xa[V], ya[V];
      unpack16<T>(a[k], xa); unpack16<T>(b[k], ya);
#pragma unroll
      for (int e = 0; e < V; e++) acc += xa[e] * ya[e];
    }
  }
  for (; i < nvec; i += stride) {
    float xa[V], ya[V];
    unpack16<T>(*(const u32x4*)(x + i * V), xa); unpack16<T>(*(const u32x4*)(y + i * V), ya);
#pragma unroll
    for (int e = 0; e < V; e++) acc += xa[e] * ya[e];
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) {
    float sc = scale; if (scale_ptr) sc *= *scale_ptr;
    unsafeAtomicAdd(out, acc * sc);
  }
}
extern "C" int sg_dot(int dtype, const void* x, const void* y, long long n, float* out, float scale, const float* scale_ptr, sg_stream_t s) {
  SG_CHECK(x && y && out, "sg_dot: null");
  bool done = false;
  DISPATCH_T(dtype, {
    constexpr int V = ET<T>::VEC;
    if (n % V == 0 && n >= (1ll << 16) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
      const long long nvec = n / V;
      int blocks = nblk(nvec, 256 * 4); if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(k_dot_vec<T>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)y, nvec, out, scale, scale_ptr);
      done = true;
    }
  });
  if (done) { SG_LAUNCH_CHECK(); return 0; }
  int blocks = nblk(n, 256 * 8); if (blocks > 1024) blocks = 1024;
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_dot<T>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)y, n, out, scale, scale_ptr));
  SG_LAUNCH_CHECK();
  return 0;
}
// block = 64 channel lanes x 4 row lanes; grid = (channel tiles, row chunks)
template <typename T> __global__ __launch_bounds__(256) void k_colsum(const T* x, int ldx, const T* mask, int ldm, long long rows, int C, float* out, float alpha, long long rows_per_block) {
  __shared__ float sm[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  long long r0 = blockIdx.y * rows_per_block, r1 = r0 + rows_per_block; if (r1 > rows) r1 = rows;
  float acc = 0.f;
  if (c < C) {
    for (long long r = r0 + ry; r < r1; r += 4) {
      float v = to_f<T>(x[r * ldx + c]);
      if (mask) { if (!(to_f<T>(mask[r * ldm + c]) > 0.f)) v = 0.f; }
      acc += v;
    }
  }
  sm[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && c < C) unsafeAtomicAdd(out + c, alpha * (sm[0][cx] + sm[1][cx] + sm[2][cx] + sm[3][cx]));
}
// Streaming variant (no mask): a thread owns one 16-byte channel vector and walks rows; lanes combined through LDS.
template <typename T> __global__ __launch_bounds__(256) void k_colsum_stream(const T* x, int ldx, long long rows, int C, float* out, float alpha, long long rpb) {
  constexpr int V = ET<T>::VEC;
  extern __shared__ float cs_sm[];                      // [lanes_p][C]
  const int CV = C / V;
  const int lanes_p = blockDim.x / CV;
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  long long r0 = blockIdx.x * rpb, r1 = r0 + rpb; if (r1 > rows) r1 = rows;
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; e++) acc[e] = 0.f;
  long long r = r0 + pl;                                    // four loads in flight per lane (norm.hip, k_bn_apply_stream); same summation order
  for (; r + 3ll * lanes_p < r1; r += 4ll * lanes_p) {
    u32x4 raw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) raw[i] = *(const u32x4*)(x + (r + (long long)i * lanes_p) * ldx + cv * V);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float xv[V];
      unpack16<T>(raw[i], xv);
#pragma unroll
      for (int e = 0; e < V; e++) acc[e] += xv[e];
    }
  }
  for (; r < r1; r += lanes_p) {
    float xv[V];
    unpack16<T>(*(const u32x4*)(x + r * ldx + cv * V), xv);
#pragma unroll
    for (int e = 0; e < V; e++) acc[e] += xv[e];
  }
#pragma unroll
  for (int e = 0; e < V; e++) cs_sm[pl * C + cv * V + e] = acc[e];
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    float a = 0.f;
    for (int l = 0; l < lanes_p; l++) a += cs_sm[l * C + i];
    unsafeAtomicAdd(out + i, alpha * a);
  }
}
extern "C" int sg_colsum(int dtype, const void* x, int ldx, const void* mask, int ldm, long long rows, int C, float* out, float alpha, sg_stream_t s) {
  SG_CHECK(x && out && rows > 0 && C > 0, "sg_colsum: bad args");
  bool done = false;
  DISPATCH_T(dtype, {
    constexpr int V = ET<T>::VEC;
    const int CV = C / V;
    if (!mask && C % V == 0 && ldx % V == 0 && ((uintptr_t)x & 15) == 0 && CV <= 256 && rows >= 2048) {
      const int lanes_p = 256 / CV;
      long long rpb = (rows + 1023) / 1024; if (rpb < 8ll * lanes_p) rpb = 8ll * lanes_p;
      const int gx = (int)((rows + rpb - 1) / rpb);
      hipLaunchKernelGGL(k_colsum_stream<T>, dim3(gx), dim3(CV * lanes_p), (size_t)lanes_p * C * sizeof(float), (hipStream_t)s, (const T*)x, ldx, rows, C, out, alpha, rpb);
      done = true;
    }
  });
  if (done) { SG_LAUNCH_CHECK(); return 0; }
  int ct = (C + 63) / 64;
  long long want = 2048 / ct; if (want < 1) want = 1;
  long long rpb = (rows + want - 1) / want; if (rpb < 64) rpb = 64;
  int ry = (int)((rows + rpb - 1) / rpb);
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_colsum<T>, dim3(ct, ry), dim3(256), 0, (hipStream_t)s, (const T*)x, ldx, (const T*)mask, ldm, rows, C, out, alpha, rpb));
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- embedding (reference src/utils/ops.py:191-192) ------------------------------------------------------
__global__ void k_embedding_fwd(const float* table, const int64_t* idx, float* out, int B, int dim, int num) {
  long long total = (long long)B * dim;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int b = (int)(i / dim), d = (int)(i % dim);
    long long r = idx[b];
    out[i] = (r >= 0 && r < num) ? table[r * dim + d] : 0.f;
  }
}
extern "C" int sg_embedding_fwd(const float* table, const int64_t* idx, float* out, int B, int dim, int num, sg_stream_t s) {
  SG_CHECK(table && idx && out, "sg_embedding_fwd: null");
  hipLaunchKernelGGL(k_embedding_fwd, dim3(nblk((long long)B * dim, 256)), dim3(256), 0, (hipStream_t)s, table, idx, out, B, dim, num);
  SG_LAUNCH_CHECK();
  return 0;
}
__global__ void k_embedding_bwd(const float* dout, const int64_t* idx, float* dtable, int B, int dim, int num) {
  long long total = (long long)B * dim;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int b = (int)(i / dim), d = (int)(i % dim);
    long long r = idx[b];
    if (r >= 0 && r < num) unsafeAtomicAdd(dtable + r * dim + d, dout[i]);
  }
}
extern "C" int sg_embedding_bwd(const float* dout, const int64_t* idx, float* dtable, int B, int dim, int num, sg_stream_t s) {
  SG_CHECK(dout && idx && dtable, "sg_embedding_bwd: null");
  hipLaunchKernelGGL(k_embedding_bwd, dim3(nblk((long long)B * dim, 256)), dim3(256), 0, (hipStream_t)s, dout, idx, dtable, B, dim, num);
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- discriminator head (reference src/models/big_resnet.py:359-363,386-387) -----------------------------
template <typename T> __global__ void k_relu_sum_hw_fwd(const T* x, float* h, int B, int HW, int C) {
  long long total = (long long)B * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int b = (int)(i / C), c = (int)(i % C);
    const T* p = x + (long long)b * HW * C + c;
    float acc = 0.f;
    for (int k = 0; k < HW; k++) acc += fmaxf(to_f<T>(p[(long long)k * C]), 0.f);
    h[i] = acc;
  }
}
extern "C" int sg_relu_sum_hw_fwd(int dtype, const void* x, float* h, int B, int HW, int C, sg_stream_t s) {
  SG_CHECK(x && h, "sg_relu_sum_hw_fwd: null");
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_relu_sum_hw_fwd<T>, dim3(nblk((long long)B * C, 256)), dim3(256), 0, (hipStream_t)s, (const T*)x, h, B, HW, C));
  SG_LAUNCH_CHECK();
  return 0;
}
template <typename T> __global__ void k_relu_sum_hw_bwd(const T* x, const float* dh, T* dx, int B, int HW, int C) {
  long long total = (long long)B * HW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C); int b = (int)(i / ((long long)HW * C));
    float g = (to_f<T>(x[i]) > 0.f) ? dh[(long long)b * C + c] : 0.f;
    dx[i] = from_f<T>(g);
  }
}
extern "C" int sg_relu_sum_hw_bwd(int dtype, const void* x, const float* dh, void* dx, int B, int HW, int C, sg_stream_t s) {
  SG_CHECK(x && dh && dx, "sg_relu_sum_hw_bwd: null");
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_relu_sum_hw_bwd<T>, dim3(nblk((long long)B * HW * C, 256)), dim3(256), 0, (hipStream_t)s, (const T*)x, dh, (T*)dx, B, HW, C));
  SG_LAUNCH_CHECK();
  return 0;
}
__global__ __launch_bounds__(256) void k_pd_head_fwd(const float* h, const float* w1, const float* b1, const float* emb, float* adv, int C) {
  __shared__ float sm[4];
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    float w = w1[c];
    if (emb) w += emb[(long long)b * C + c];
    acc += h[(long long)b * C + c] * w;
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) adv[b] = acc + (b1 ? b1[0] : 0.f);
}
extern "C" int sg_pd_head_fwd(const float* h, const float* w1, const float* b1, const float* emb, float* adv, int B, int C, sg_stream_t s) {
  SG_CHECK(h && w1 && adv, "sg_pd_head_fwd: null");
  hipLaunchKernelGGL(k_pd_head_fwd, dim3(B), dim3(256), 0, (hipStream_t)s, h, w1, b1, emb, adv, C);
  SG_LAUNCH_CHECK();
  return 0;
}
// block = 32 channels x 8 batch lanes: a thread walks its eighth of the batch with independent, unrolled loads; the weight-gradient partials of
// the 8 lanes are combined through LDS in a fixed order (deterministic, no atomics). The one-thread-per-channel loop this replaces walked the
// whole batch with a load -> store -> load dependency per sample: 199 us per call at batch 256 x 1536 channels (r03 kernel trace), all latency.
__global__ __launch_bounds__(256) void k_pd_head_bwd(const float* __restrict__ h, const float* __restrict__ w1, const float* __restrict__ emb,
                                                     const float* __restrict__ dadv, float* __restrict__ dh, float* __restrict__ dw1,
                                                     float* __restrict__ db1, float* __restrict__ demb, int B, int C) {
  __shared__ float sm[8][32];
  const int cl = threadIdx.x & 31, bl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float aw = 0.f;
  if (c < C) {
    const float w = w1[c];
#pragma unroll 4
    for (int b = bl; b < B; b += 8) {
      const float g = dadv[b];
      const float hv = h[(long long)b * C + c];
      float wt = w;
      if (emb) { wt += emb[(long long)b * C + c]; demb[(long long)b * C + c] = g * hv; }
      dh[(long long)b * C + c] = g * wt;
      aw += g * hv;
    }
  }
  sm[bl][cl] = aw;
  __syncthreads();
  if (bl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) t += sm[k][cl];
    dw1[c] += t;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && db1) {
    float ab = 0.f;
    for (int b = 0; b < B; b++) ab += dadv[b];
    db1[0] += ab;
  }
}
extern "C" int sg_pd_head_bwd(const float* h, const float* w1, const float* emb, const float* dadv, float* dh, float* dw1, float* db1, float* demb, int B, int C, sg_stream_t s) {
  SG_CHECK(h && w1 && dadv && dh && dw1 && (!emb || demb), "sg_pd_head_bwd: null");
  hipLaunchKernelGGL(k_pd_head_bwd, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)s, h, w1, emb, dadv, dh, dw1, db1, demb, B, C);
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- adversarial losses (reference src/utils/losses.py:197-239) ------------------------------------------
__device__ __forceinline__ float softplusf(float x) { return (x > 20.f) ? x : log1pf(__expf(x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }
__global__ __launch_bounds__(256) void k_loss_d(int kind, const float* real, const float* fake, int B, float* loss, float* d_real, float* d_fake) {
  __shared__ float sm[4];
  float acc = 0.f;
  const float inv = 1.f / (float)B;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float r = real[b], f = fake[b];
    if (kind == 0) {  // hinge
      acc += fmaxf(1.f - r, 0.f) + fmaxf(1.f + f, 0.f);
      d_real[b] = (1.f - r > 0.f) ? -inv : 0.f;
      d_fake[b] = (1.f + f > 0.f) ? inv : 0.f;
    } else if (kind == 1) {  // wasserstein
      acc += f - r; d_real[b] = -inv; d_fake[b] = inv;
    } else {  // vanilla: softplus(-r) + softplus(f)
      acc += softplusf(-r) + softplusf(f);
      d_real[b] = -sigmoidf(-r) * inv; d_fake[b] = sigmoidf(f) * inv;
    }
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) loss[0] = acc * inv;
}
extern "C" int sg_loss_d(int kind, const float* real, const float* fake, int B, float* loss, float* d_real, float* d_fake, sg_stream_t s) {
  SG_CHECK(real && fake && loss && d_real && d_fake && B > 0, "sg_loss_d: bad args");
  hipLaunchKernelGGL(k_loss_d, dim3(1), dim3(256), 0, (hipStream_t)s, kind, real, fake, B, loss, d_real, d_fake);
  SG_LAUNCH_CHECK();
  return 0;
}
__global__ __launch_bounds__(256) void k_loss_g(int kind, const float* fake, int B, float* loss, float* d_fake) {
  __shared__ float sm[4];
  float acc = 0.f;
  const float inv = 1.f / (float)B;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float f = fake[b];
    if (kind == 2) { acc += softplusf(-f); d_fake[b] = -sigmoidf(-f) * inv; }
    else { acc += -f; d_fake[b] = -inv; }
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) loss[0] = acc * inv;
}
extern "C" int sg_loss_g(int kind, const float* fake, int B, float* loss, float* d_fake, sg_stream_t s) {
  SG_CHECK(fake && loss && d_fake && B > 0, "sg_loss_g: bad args");
  hipLaunchKernelGGL(k_loss_g, dim3(1), dim3(256), 0, (hipStream_t)s, kind, fake, B, loss, d_fake);
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- dtype conversion, residual add with ReLU (DiscBlock identity skip sees the in-place ReLU'd input) -----
template <typename TS, typename TD> __global__ void k_convert(const TS* x, TD* y, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = from_f<TD>(to_f<TS>(x[i]));
}
extern "C" int sg_convert(int src_dtype, int dst_dtype, const void* x, void* y, long long n, sg_stream_t s) {
  SG_CHECK(x && y, "sg_convert: null");
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)s;
  dim3 g(nblk(n, 256)), b(256);
  if (src_dtype == SG_DTYPE_F32 && dst_dtype == SG_DTYPE_BF16) hipLaunchKernelGGL((k_convert<float, bf16_t>), g, b, 0, st, (const float*)x, (bf16_t*)y, n);
  else if (src_dtype == SG_DTYPE_BF16 && dst_dtype == SG_DTYPE_F32) hipLaunchKernelGGL((k_convert<bf16_t, float>), g, b, 0, st, (const bf16_t*)x, (float*)y, n);
  else if (src_dtype == SG_DTYPE_F32 && dst_dtype == SG_DTYPE_F32) hipLaunchKernelGGL((k_convert<float, float>), g, b, 0, st, (const float*)x, (float*)y, n);
  else if (src_dtype == SG_DTYPE_BF16 && dst_dtype == SG_DTYPE_BF16) hipLaunchKernelGGL((k_convert<bf16_t, bf16_t>), g, b, 0, st, (const bf16_t*)x, (bf16_t*)y, n);
  else { sg_set_error("sg_convert: bad dtype"); return -1; }
  SG_LAUNCH_CHECK();
  return 0;
}
// out = a + relu(x)
template <typename T> __global__ void k_add_relu(const T* a, const T* x, T* out, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = from_f<T>(to_f<T>(a[i]) + fmaxf(to_f<T>(x[i]), 0.f));
}
extern "C" int sg_add_relu(int dtype, const void* a, const void* x, void* out, long long n, sg_stream_t s) {
  SG_CHECK(a && x && out, "sg_add_relu: null");
  if (n <= 0) return 0;
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_add_relu<T>, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)s, (const T*)a, (const T*)x, (T*)out, n));
  SG_LAUNCH_CHECK();
  return 0;
}
// dx = dy * (x > 0)
template <typename T> __global__ void k_relu_mask(const T* dy, const T* x, T* dx, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dx[i] = (to_f<T>(x[i]) > 0.f) ? dy[i] : from_f<T>(0.f);
}
__global__ __launch_bounds__(256) void k_relu_mask_v8(const bf16_t* dy, const bf16_t* x, bf16_t* dx, unsigned nv) {
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < nv; i += gridDim.x * 256u) {
    const u32x4 g = *(const u32x4*)(dy + 8ll * i);
    float xv[8];
    unpack16<bf16_t>(*(const u32x4*)(x + 8ll * i), xv);
    u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; d++) o[d] = (xv[2 * d] > 0.f ? (g[d] & 0xffffu) : 0u) | (xv[2 * d + 1] > 0.f ? (g[d] & 0xffff0000u) : 0u);
    *(u32x4*)(dx + 8ll * i) = o;
  }
}
extern "C" int sg_relu_mask(int dtype, const void* dy, const void* x, void* dx, long long n, sg_stream_t s) {
  SG_CHECK(dy && x && dx, "sg_relu_mask: null");
  if (n <= 0) return 0;
  if (dtype == SG_DTYPE_BF16 && n % 8 == 0 && (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx) & 15) == 0 && n / 8 < (1ll << 31)) {
    hipLaunchKernelGGL(k_relu_mask_v8, dim3(nblk(n / 8, 256)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, (unsigned)(n / 8));
    SG_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_relu_mask<T>, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)s, (const T*)dy, (const T*)x, (T*)dx, n));
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- WGAN-GP pieces (reference src/utils/losses.py:268-275,301-316) -----------------------------------------------
// out[b,c] = sum_hw t[b,hw,c] * (x[b,hw,c] > 0): adjoint of sg_relu_sum_hw_bwd w.r.t. dh (second-order pass)
template <typename T> __global__ void k_masked_sum_hw(const T* t, const T* x, float* out, int B, int HW, int C) {
  long long total = (long long)B * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int b = (int)(i / C), c = (int)(i % C);
    const long long base = (long long)b * HW * C + c;
    float acc = 0.f;
    for (int k = 0; k < HW; k++) acc += (to_f<T>(x[base + (long long)k * C]) > 0.f) ? to_f<T>(t[base + (long long)k * C]) : 0.f;
    out[i] = acc;
  }
}
extern "C" int sg_masked_sum_hw(int dtype, const void* t, const void* x, float* out, int B, int HW, int C, sg_stream_t s) {
  SG_CHECK(t && x && out, "sg_masked_sum_hw: null");
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_masked_sum_hw<T>, dim3(nblk((long long)B * C, 256)), dim3(256), 0, (hipStream_t)s, (const T*)t, (const T*)x, out, B, HW, C));
  SG_LAUNCH_CHECK();
  return 0;
}
// interpolates[b,:] = alpha[b] * real[b,:] + (1 - alpha[b]) * fake[b,:]      (losses.py:303-308)
__global__ void k_interp_rows(const float* real, const float* fake, const float* alpha, float* out, long long n, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const float a = alpha[i / n];
    out[i] = a * real[i] + (1.f - a) * fake[i];
  }
}
extern "C" int sg_interp_rows(const float* real, const float* fake, const float* alpha, float* out, int B, long long n, sg_stream_t s) {
  SG_CHECK(real && fake && alpha && out && B > 0 && n > 0, "sg_interp_rows: bad args");
  hipLaunchKernelGGL(k_interp_rows, dim3(nblk((long long)B * n, 256)), dim3(256), 0, (hipStream_t)s, real, fake, alpha, out, n, (long long)B * n);
  SG_LAUNCH_CHECK();
  return 0;
}
// norms[b] = ||grads[b,:]||_2 (one block per sample); penalty = mean_b (norms[b] - 1)^2     (losses.py:313-315)
__global__ __launch_bounds__(256) void k_gp_norms(const float* grads, long long n, float* norms) {
  __shared__ float sm[4];
  const float* p = grads + (long long)blockIdx.x * n;
  float acc = 0.f;
  for (long long i = threadIdx.x; i < n; i += 256) acc += p[i] * p[i];
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) norms[blockIdx.x] = sqrtf(acc);
}
// kind 0: WGAN-GP  mean_b (||g_b|| - 1)^2   (reference utils/losses.py:301-316; DRA :319-335 is the same functional)
// kind 1: R1       0.5 mean_b ||g_b||^2      (:355-361)
// kind 2: maxGP    max_b ||g_b||^2           (:338-352); norms[B] receives the (first) arg-max row for the backward pass
__global__ __launch_bounds__(256) void k_gp_loss(int kind, float* norms, int B, float* loss) {
  __shared__ float sm[4];
  if (kind == 2) {
    if (threadIdx.x == 0) {
      float best = -1.f; int bi = 0;
      for (int b = 0; b < B; b++) { const float v = norms[b] * norms[b]; if (v > best) { best = v; bi = b; } }
      loss[0] = best; norms[B] = (float)bi;
    }
    return;
  }
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    if (kind == 0) { const float d = norms[b] - 1.f; acc += d * d; }
    else acc += 0.5f * norms[b] * norms[b];
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) loss[0] = acc / (float)B;
}
extern "C" int sg_gp_fwd(int kind, const float* grads, int B, long long n, float* norms, float* loss, sg_stream_t s) {
  SG_CHECK(grads && norms && loss && B > 0 && n > 0 && kind >= 0 && kind <= 2, "sg_gp_fwd: bad args");
  hipLaunchKernelGGL(k_gp_norms, dim3(B), dim3(256), 0, (hipStream_t)s, grads, n, norms);
  hipLaunchKernelGGL(k_gp_loss, dim3(1), dim3(256), 0, (hipStream_t)s, kind, norms, B, loss);
  SG_LAUNCH_CHECK();
  return 0;
}
// d penalty / d grads[b,:] = gout * { (2/B) (1 - 1/norms[b]) | 1/B | 2 [b == argmax] } grads[b,:]
__global__ void k_gp_bwd(int kind, const float* grads, const float* norms, const float* gout, float* dgrads, int B, long long n, long long total) {
  const float go = gout[0];
  const int bmax = kind == 2 ? (int)norms[B] : -1;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / n);
    float f;
    if (kind == 0) f = go * 2.f / (float)B * (1.f - 1.f / norms[b]);
    else if (kind == 1) f = go / (float)B;
    else f = (b == bmax) ? 2.f * go : 0.f;
    dgrads[i] = f * grads[i];
  }
}
extern "C" int sg_gp_bwd(int kind, const float* grads, const float* norms, const float* gout, float* dgrads, int B, long long n, sg_stream_t s) {
  SG_CHECK(grads && norms && gout && dgrads && B > 0 && n > 0 && kind >= 0 && kind <= 2, "sg_gp_bwd: bad args");
  hipLaunchKernelGGL(k_gp_bwd, dim3(nblk((long long)B * n, 256)), dim3(256), 0, (hipStream_t)s, kind, grads, norms, gout, dgrads, B, n, (long long)B * n);
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- BigGAN-deep skip connections (reference src/models/big_resnet_deep_legacy.py:53-56,74-77,236-238) ------------------------
// y[n][h][w][c] = x[n][h/up][w/up][c] for c < C: the generator's channel-slice (+ nearest x2) skip
template <typename T> __global__ void k_slice_up_fwd(const T* x, T* y, int Hs, int Ws, int ldx, int C, int up, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long long t = i / C;
    const int w = (int)(t % (Ws * up)); t /= (Ws * up);
    const int h = (int)(t % (Hs * up)); const int n = (int)(t / (Hs * up));
    y[i] = x[(((long long)n * Hs + h / up) * Ws + w / up) * ldx + c];
  }
}
__global__ __launch_bounds__(256) void k_slice_up_fwd_v8(const bf16_t* x, bf16_t* y, int Hs, int Ws, int ldx, int C, int up, unsigned total) {
  const int CV = C / 8, Wo = Ws * up, Ho = Hs * up;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned pix = i / CV, cv = i - pix * CV;
    const unsigned w = pix % Wo, t = pix / Wo, h = t % Ho, n = t / Ho;
    *(u32x4*)(y + (long long)pix * C + cv * 8) = *(const u32x4*)(x + (((long long)n * Hs + h / up) * Ws + w / up) * ldx + cv * 8);
  }
}
__global__ __launch_bounds__(256) void k_slice_up_bwd_v8(const bf16_t* dy, bf16_t* dx, int Hs, int Ws, int ldx, int C, int up, unsigned total) {
  const int LV = ldx / 8;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned pix = i / LV, cv = i - pix * LV;
    const unsigned ws = pix % Ws, t = pix / Ws, hs = t % Hs, n = t / Hs;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = 0.f;
    if ((int)cv * 8 < C) {
      for (int a = 0; a < up; a++)
        for (int b = 0; b < up; b++) {
          float v[8];
          unpack16<bf16_t>(*(const u32x4*)(dy + (((long long)n * Hs * up + hs * up + a) * (Ws * up) + ws * up + b) * C + cv * 8), v);
#pragma unroll
          for (int e = 0; e < 8; e++) acc[e] += v[e];
        }
    }
    *(u32x4*)(dx + (long long)pix * ldx + cv * 8) = pack16<bf16_t>(acc);
  }
}
__global__ __launch_bounds__(256) void k_copy_channels_v8(const bf16_t* src, int lds, bf16_t* dst, int ldd, int C, unsigned total) {
  const int CV = C / 8;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned r = i / CV, cv = i - r * CV;
    *(u32x4*)(dst + (long long)r * ldd + cv * 8) = *(const u32x4*)(src + (long long)r * lds + cv * 8);
  }
}
extern "C" int sg_slice_up_fwd(int dtype, const void* x, void* y, int N, int Hs, int Ws, int ldx, int C, int up, sg_stream_t s) {
  SG_CHECK(x && y && C > 0 && C <= ldx && (up == 1 || up == 2), "sg_slice_up_fwd: bad args");
  const long long total = (long long)N * Hs * up * Ws * up * C;
  if (ew_v8_ok(dtype, x, y, C, ldx, 8, total / 8)) {
    hipLaunchKernelGGL(k_slice_up_fwd_v8, dim3(nblk(total / 8, 256)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (bf16_t*)y, Hs, Ws, ldx, C, up, (unsigned)(total / 8));
    SG_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_slice_up_fwd<T>, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)s, (const T*)x, (T*)y, Hs, Ws, ldx, C, up, total));
  SG_LAUNCH_CHECK();
  return 0;
}
// dx[n][hs][ws][c] = sum over the up x up block of dy (c < C), 0 for the channels the slice dropped
template <typename T> __global__ void k_slice_up_bwd(const T* dy, T* dx, int Hs, int Ws, int ldx, int C, int up, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % ldx); long long t = i / ldx;
    const int ws = (int)(t % Ws); t /= Ws;
    const int hs = (int)(t % Hs); const int n = (int)(t / Hs);
    float acc = 0.f;
    if (c < C) {
      for (int a = 0; a < up; a++)
        for (int b = 0; b < up; b++)
          acc += to_f<T>(dy[(((long long)n * Hs * up + hs * up + a) * (Ws * up) + ws * up + b) * C + c]);
    }
    dx[i] = from_f<T>(acc);
  }
}
extern "C" int sg_slice_up_bwd(int dtype, const void* dy, void* dx, int N, int Hs, int Ws, int ldx, int C, int up, sg_stream_t s) {
  SG_CHECK(dy && dx && C > 0 && C <= ldx && (up == 1 || up == 2), "sg_slice_up_bwd: bad args");
  const long long total = (long long)N * Hs * Ws * ldx;
  if (ew_v8_ok(dtype, dy, dx, C, ldx, 8, total / 8)) {
    hipLaunchKernelGGL(k_slice_up_bwd_v8, dim3(nblk(total / 8, 256)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)dy, (bf16_t*)dx, Hs, Ws, ldx, C, up, (unsigned)(total / 8));
    SG_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_slice_up_bwd<T>, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)s, (const T*)dy, (T*)dx, Hs, Ws, ldx, C, up, total));
  SG_LAUNCH_CHECK();
  return 0;
}
// dst[r][0..C) = src[r][0..C) with independent row pitches (channel-concat skip of the discriminator)
template <typename T> __global__ void k_copy_channels(const T* src, int lds, T* dst, int ldd, int C, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); const long long r = i / C;
    dst[r * ldd + c] = src[r * lds + c];
  }
}
extern "C" int sg_copy_channels(int dtype, const void* src, int ld_src, void* dst, int ld_dst, long long rows, int C, sg_stream_t s) {
  SG_CHECK(src && dst && rows > 0 && C > 0 && C <= ld_src && C <= ld_dst, "sg_copy_channels: bad args");
  const long long total = rows * C;
  if (ew_v8_ok(dtype, src, dst, C, ld_src, ld_dst, total / 8)) {
    hipLaunchKernelGGL(k_copy_channels_v8, dim3(nblk(total / 8, 256)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, C, (unsigned)(total / 8));
    SG_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_copy_channels<T>, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)s, (const T*)src, ld_src, (T*)dst, ld_dst, C, total));
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- LeCam regulariser (reference src/utils/losses.py:262-265; anchors = LeCamEMA.D_fake / D_real, src/utils/ops.py:106-133) ------
// loss = mean relu(real - ema_fake)^2 + mean relu(ema_real - fake)^2, with its gradient w.r.t. the logits
__global__ __launch_bounds__(256) void k_lecam(const float* real, const float* fake, int B, float ema_real, float ema_fake, float* loss,
                                               float* d_real, float* d_fake) {
  __shared__ float sm[4];
  float acc = 0.f;
  const float inv = 1.f / (float)B;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float a = fmaxf(real[b] - ema_fake, 0.f), c = fmaxf(ema_real - fake[b], 0.f);
    acc += a * a + c * c;
    d_real[b] = 2.f * a * inv;
    d_fake[b] = -2.f * c * inv;
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) loss[0] = acc * inv;
}
extern "C" int sg_lecam(const float* real, const float* fake, int B, float ema_real, float ema_fake, float* loss, float* d_real, float* d_fake,
                        sg_stream_t s) {
  SG_CHECK(real && fake && loss && d_real && d_fake && B > 0, "sg_lecam: bad args");
  hipLaunchKernelGGL(k_lecam, dim3(1), dim3(256), 0, (hipStream_t)s, real, fake, B, ema_real, ema_fake, loss, d_real, d_fake);
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- uint8 input path (reference src/data_util.py:92-94,141: ToTensor + Normalize(0.5, 0.5) of the HDF5 / in-memory uint8 images,
// optional horizontal flip): [N][H][W][3] uint8 -> T [N][H][W][cpad] = (x / 255 - 0.5) / 0.5, zero-filled channels, the layout the
// discriminator's stem reads. Same fp32 operations as the host transform, so the result equals converting its fp32 output.
template <typename T> __global__ void k_u8_to_nhwc(const uint8_t* x, const uint8_t* flip, T* y, int H, int W, int cpad, long long npix) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const long long nh = i / W;
    const long long n = nh / H;
    const int ws = (flip && flip[n]) ? (W - 1 - w) : w;
    const uint8_t* p = x + (nh * W + ws) * 3;
    T* o = y + i * cpad;
    for (int c = 0; c < cpad; c++) {
      float v = 0.f;
      if (c < 3) v = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p[c], 255.f), 0.5f), 0.5f);
      o[c] = from_f<T>(v);
    }
  }
}
extern "C" int sg_u8_to_nhwc(int dtype, const uint8_t* x, const uint8_t* flip, void* y, int N, int H, int W, int cpad, sg_stream_t s) {
  SG_CHECK(x && y && N > 0 && H > 0 && W > 0 && cpad >= 3, "sg_u8_to_nhwc: bad args");
  const long long npix = (long long)N * H * W;
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_u8_to_nhwc<T>, dim3(nblk(npix, 256)), dim3(256), 0, (hipStream_t)s, x, flip, (T*)y, H, W, cpad, npix));
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- the dataset resident in HBM: a training "basket" is one gather ------------------------------------------------------------------------
// dst[b][h][w][c] = src[idx[b]][h][flip[b] ? W - 1 - w : w][c] for uint8 images [N][H][W][3] (the HDF5 / in-memory layout of reference
// src/utils/hdf5.py:35-97, src/data_util.py:102-142); replaces the DataLoader's per-sample __getitem__ + RandomHorizontalFlip + collate +
// host-to-device copy for a data set that fits the 288 GB of HBM (ImageNet-128 as uint8: 63 GB). labels gathered alongside.
__global__ __launch_bounds__(256) void k_gather_images_u8(const uint8_t* src, const int64_t* idx, const uint8_t* flip, uint8_t* dst, int B, int H, int W,
                                                          const int64_t* lab_src, int64_t* lab_dst) {
  const long long npix = (long long)B * H * W;
  for (long long p = blockIdx.x * 256ll + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
    const int w = (int)(p % W);
    const long long t = p / W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    const long long n = idx[b];
    const int ws = (flip && flip[b]) ? (W - 1 - w) : w;
    const uint8_t* s = src + ((n * H + h) * W + ws) * 3;
    uint8_t* d = dst + p * 3;
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    if (lab_dst && h == 0 && w == 0) lab_dst[b] = lab_src[n];
  }
}
extern "C" int sg_gather_images_u8(const uint8_t* src, const int64_t* idx, const uint8_t* flip, uint8_t* dst, int B, int H, int W,
                                   const int64_t* lab_src, int64_t* lab_dst, sg_stream_t s) {
  SG_CHECK(src && idx && dst && B > 0 && H > 0 && W > 0, "sg_gather_images_u8: bad args");
  SG_CHECK((lab_src == nullptr) == (lab_dst == nullptr), "sg_gather_images_u8: labels in and out go together");
  const long long npix = (long long)B * H * W;
  hipLaunchKernelGGL(k_gather_images_u8, dim3(nblk(npix, 256)), dim3(256), 0, (hipStream_t)s, src, idx, flip, dst, B, H, W, lab_src, lab_dst);
  SG_LAUNCH_CHECK();
  return 0;
}
