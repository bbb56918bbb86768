// optim.hip -- torch.optim.Adam step fused with the generator EMA, over flat fp32 arenas (one launch per network
// instead of one small kernel per parameter/buffer: reference src/config.py:541-563, src/utils/ema.py:27-40).
// HBM-bound: 16 B read + 12 B written per parameter (+8 B with the EMA copy).
#include "common.h"
#include "../../include/sgamd.h"

// torch.lerp(start, end, w): w < 0.5 ? start + w*(end-start) : end - (end-start)*(1-w)
__device__ __forceinline__ float torch_lerp(float start, float end, float w) {
  const float d = end - start;
  return (w < 0.5f) ? (start + w * d) : (end - d * (1.f - w));
}

__global__ __launch_bounds__(256) void k_adam_ema(float* p, const float* g, float* m, float* v, float* ema, long long n, float lr,
                                                   float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                                                   float ema_decay, float grad_scale) {
  const long long n4 = n >> 2;
  const float step_size = lr / bc1;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 pv = ((f32x4*)p)[i], gv = ((const f32x4*)g)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
    f32x4 ev; if (ema) ev = ((f32x4*)ema)[i];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float gr = gv[e] * grad_scale;
      if (wd != 0.f) gr += wd * pv[e];
      mv[e] = torch_lerp(mv[e], gr, 1.f - beta1);
      vv[e] = vv[e] * beta2 + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pv[e] = pv[e] - step_size * (mv[e] / denom);
      if (ema) ev[e] = torch_lerp(pv[e], ev[e], ema_decay);
    }
    ((f32x4*)p)[i] = pv; ((f32x4*)m)[i] = mv; ((f32x4*)v)[i] = vv;
    if (ema) ((f32x4*)ema)[i] = ev;
  }
  // tail
  if (blockIdx.x == 0) {
    for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
      float gr = g[i] * grad_scale;
      if (wd != 0.f) gr += wd * p[i];
      const float mm = torch_lerp(m[i], gr, 1.f - beta1);
      const float vv = v[i] * beta2 + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(vv) / bc2_sqrt + eps;
      const float pn = p[i] - step_size * (mm / denom);
      m[i] = mm; v[i] = vv; p[i] = pn;
      if (ema) ema[i] = torch_lerp(pn, ema[i], ema_decay);
    }
  }
}
extern "C" int sg_adam_ema(float* p, const float* g, float* m, float* v, float* ema, long long n, float lr, float beta1, float beta2,
                           float eps, float wd, int step, float ema_decay, float grad_scale, sg_stream_t s) {
  SgProfScope prof((hipStream_t)s, (double)n * (ema ? 36.0 : 28.0), 6);      // p, m, v read + write, g read (+ ema read + write)
  SG_CHECK(p && g && m && v && n > 0 && step >= 1, "sg_adam_ema: bad args");
  SG_CHECK(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v) | ((uintptr_t)ema)) & 15) == 0, "sg_adam_ema: arenas must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long long blocks = (n / 4 + 255) / 256; if (blocks > 256 * 16) blocks = 256 * 16; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_adam_ema, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, p, g, m, v, ema, n, lr, beta1, beta2, eps, wd, (float)bc1,
                     (float)sqrt(bc2), ema_decay, grad_scale);
  SG_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void k_ema_lerp(const float* src, float* ema, long long n, float decay) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) ema[i] = torch_lerp(src[i], ema[i], decay);
}
extern "C" int sg_ema_lerp(const float* src, float* ema, long long n, float decay, sg_stream_t s) {
  SgProfScope prof((hipStream_t)s, (double)(n > 0 ? n : 0) * 12.0, 6);
  SG_CHECK(src && ema, "sg_ema_lerp: null");
  if (n <= 0) return 0;
  long long blocks = (n + 255) / 256; if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(k_ema_lerp, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, src, ema, n, decay);
  SG_LAUNCH_CHECK();
  return 0;
}
