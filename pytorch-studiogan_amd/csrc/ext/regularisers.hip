// regularisers.hip (ext) -- three small HBM-bound operators of the discriminator update:
//   sg_clamp_flat     weight clipping of WGAN (reference src/worker.py:489-492: `for p in Dis.parameters(): p.data.clamp_(-wc_bound, wc_bound)` -- one torch launch
//                     per parameter tensor) as ONE pass over the flat parameter arena the fused optimiser already owns
//   sg_select_rows    adaptive pseudo augmentation (reference src/utils/apa_aug.py:10-21): out[n] = flag[n] ? a[n] : b[n] per image -- the reference
//                     builds fake * flag + real * (1 - flag) from three elementwise launches and syncs the host on an allclose() first
//   sg_sign_count     the ADA / APA overfitting heuristic's accumulator (src/worker.py:285-289,478-481): acc[0] += sum_b sign(logit_b), acc[1] += B, on the device
//                     (the reference pulls the sum to the host with .item() in every micro-step)
#include "../common.h"
#include "../../../include/sgamd.h"

__global__ __launch_bounds__(256) void k_clamp_flat(float* p, long long n, float lo, float hi) {
  const long long nv = n >> 2;
  f32x4* pv = (f32x4*)p;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    f32x4 v = pv[i];
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = fminf(fmaxf(v[e], lo), hi);
    pv[i] = v;
  }
  for (long long i = (nv << 2) + blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = fminf(fmaxf(p[i], lo), hi);
}
extern "C" int sg_clamp_flat(float* p, long long n, float lo, float hi, sg_stream_t s) {
  SG_CHECK(p && n > 0 && lo <= hi && (((uintptr_t)p) & 15) == 0, "sg_clamp_flat: bad args (16-byte aligned buffer, lo <= hi)");
  long long blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_clamp_flat, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, p, n, lo, hi);
  SG_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void k_select_rows(const uint8_t* flag, const float* a, const float* b, float* out, long long row, long long total) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) out[i] = flag[i / row] ? a[i] : b[i];
}
extern "C" int sg_select_rows(const uint8_t* flag, const float* a, const float* b, float* out, int N, long long row, sg_stream_t s) {
  SG_CHECK(flag && a && b && out && N > 0 && row > 0, "sg_select_rows: bad args");
  const long long total = (long long)N * row;
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_select_rows, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, flag, a, b, out, row, total);
  SG_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void k_sign_count(const float* logit, int B, float* acc) {
  __shared__ float sm[4];
  float a = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) { const float v = logit[b]; a += (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }
  a = block_sum_256(a, sm);
  if (threadIdx.x == 0) { acc[0] += a; acc[1] += (float)B; }
}
extern "C" int sg_sign_count(const float* logit, int B, float* acc, sg_stream_t s) {
  SG_CHECK(logit && acc && B > 0, "sg_sign_count: bad args");
  hipLaunchKernelGGL(k_sign_count, dim3(1), dim3(256), 0, (hipStream_t)s, logit, B, acc);
  SG_LAUNCH_CHECK();
  return 0;
}
