// losses.hip (ext) -- the adversarial losses and the generator regulariser of the reference that the first three loss kinds (csrc/elementwise.hip
// sg_loss_d / sg_loss_g: hinge, wasserstein, vanilla) did not cover:
//   * least squares (LSGAN; reference src/utils/losses.py:216-223 d_ls / g_ls, `adv_loss: "least_square"` in configs/*/LSGAN.yaml)
//   * feature matching (reference src/utils/losses.py:254-259 feature_matching_loss, LOSS.apply_fm of the MHGAN configurations, src/worker.py:588-596):
//     mean_c | mean_b fake_h[b][c] - mean_b real_h[b][c] | on the discriminator's pooled features h [B][C]
// ("logistic", losses.py:207-213, is the vanilla kind: softplus(-r) + softplus(f); the host mirror maps it.)
// Like their siblings: the loss is a mean, so value and analytic gradient come out of the same launch.
#include "../common.h"
#include "../../../include/sgamd.h"

__global__ __launch_bounds__(256) void k_loss_ls_d(const float* real, const float* fake, int B, float* loss, float* d_real, float* d_fake) {
  __shared__ float sm[4];
  float acc = 0.f;
  const float inv = 1.f / (float)B;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float r = real[b] - 1.f, f = fake[b];
    acc += 0.5f * r * r + 0.5f * f * f;
    d_real[b] = r * inv;
    d_fake[b] = f * inv;
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) loss[0] = acc * inv;
}
__global__ __launch_bounds__(256) void k_loss_ls_g(const float* fake, int B, float* loss, float* d_fake) {
  __shared__ float sm[4];
  float acc = 0.f;
  const float inv = 1.f / (float)B;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float f = fake[b] - 1.f;
    acc += 0.5f * f * f;
    d_fake[b] = f * inv;
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) loss[0] = acc * inv;
}
extern "C" int sg_loss_ls_d(const float* real, const float* fake, int B, float* loss, float* d_real, float* d_fake, sg_stream_t s) {
  SG_CHECK(real && fake && loss && d_real && d_fake && B > 0, "sg_loss_ls_d: bad args");
  hipLaunchKernelGGL(k_loss_ls_d, dim3(1), dim3(256), 0, (hipStream_t)s, real, fake, B, loss, d_real, d_fake);
  SG_LAUNCH_CHECK();
  return 0;
}
extern "C" int sg_loss_ls_g(const float* fake, int B, float* loss, float* d_fake, sg_stream_t s) {
  SG_CHECK(fake && loss && d_fake && B > 0, "sg_loss_ls_g: bad args");
  hipLaunchKernelGGL(k_loss_ls_g, dim3(1), dim3(256), 0, (hipStream_t)s, fake, B, loss, d_fake);
  SG_LAUNCH_CHECK();
  return 0;
}

// feature matching: one lane per feature column c (coalesced over c for every b), block partial sums of |diff_c| -> part[blockIdx.x]; the gradient needs only
// the column's sign, so it is written in the same pass: d_fake[b][c] = sign(diff_c) / (B * C)
__global__ __launch_bounds__(256) void k_fm_cols(const float* real, const float* fake, int B, int C, float* part, float* d_fake) {
  __shared__ float sm[4];
  const int c = blockIdx.x * 256 + threadIdx.x;
  float a = 0.f;
  if (c < C) {
    float sr = 0.f, sf = 0.f;
    for (int b = 0; b < B; b++) { sr += real[(long long)b * C + c]; sf += fake[(long long)b * C + c]; }
    const float diff = sf / (float)B - sr / (float)B;
    a = fabsf(diff);
    const float g = (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) / ((float)B * (float)C);
    for (int b = 0; b < B; b++) d_fake[(long long)b * C + c] = g;
  }
  a = block_sum_256(a, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = a;
}
__global__ __launch_bounds__(256) void k_fm_final(const float* part, int parts, int C, float* loss) {
  __shared__ float sm[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < parts; i += 256) a += part[i];
  a = block_sum_256(a, sm);
  if (threadIdx.x == 0) loss[0] = a / (float)C;
}
extern "C" int sg_fm_work_floats(int C) { return C > 0 ? (C + 255) / 256 : 0; }
extern "C" int sg_fm_loss(const float* real_h, const float* fake_h, int B, int C, float* work, float* loss, float* d_fake, sg_stream_t s) {
  SG_CHECK(real_h && fake_h && work && loss && d_fake && B > 0 && C > 0, "sg_fm_loss: bad args");
  const int parts = (C + 255) / 256;
  hipLaunchKernelGGL(k_fm_cols, dim3(parts), dim3(256), 0, (hipStream_t)s, real_h, fake_h, B, C, work, d_fake);
  SG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_fm_final, dim3(1), dim3(256), 0, (hipStream_t)s, work, parts, C, loss);
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- InfoGAN's Q-head losses (reference src/models/big_resnet.py:337-344,373-377, src/utils/losses.py:369-375, src/worker.py:607-618) -------------------------
// sg_exp_fwd / _bwd: the continuous code's variance head, var = exp(linear(h)); dx = dy * y.
// sg_normal_nll: nll = -mean_b sum_k [ -0.5 log(2 pi var + 1e-6) - (x - mu)^2 / (2 var + 1e-6) ] with the gradients w.r.t. mu and var from the same launch.
__global__ __launch_bounds__(256) void k_exp_fwd(const float* x, float* y, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = expf(x[i]);
}
__global__ __launch_bounds__(256) void k_exp_bwd(const float* dy, const float* y, float* dx, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dx[i] = dy[i] * y[i];
}
extern "C" int sg_exp_fwd(const float* x, float* y, long long n, sg_stream_t s) {
  SG_CHECK(x && y && n > 0, "sg_exp_fwd: bad args");
  long long b = (n + 255) / 256; if (b > 4096) b = 4096;
  hipLaunchKernelGGL(k_exp_fwd, dim3((int)b), dim3(256), 0, (hipStream_t)s, x, y, n);
  SG_LAUNCH_CHECK();
  return 0;
}
extern "C" int sg_exp_bwd(const float* dy, const float* y, float* dx, long long n, sg_stream_t s) {
  SG_CHECK(dy && y && dx && n > 0, "sg_exp_bwd: bad args");
  long long b = (n + 255) / 256; if (b > 4096) b = 4096;
  hipLaunchKernelGGL(k_exp_bwd, dim3((int)b), dim3(256), 0, (hipStream_t)s, dy, y, dx, n);
  SG_LAUNCH_CHECK();
  return 0;
}
__global__ __launch_bounds__(256) void k_normal_nll(const float* x, const float* mu, const float* var, int B, int K, float* loss, float* dmu, float* dvar) {
  __shared__ float sm[4];
  const float TWO_PI = 6.283185307179586f;
  const int n = B * K;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = x[i] - mu[i], v = var[i];
    const float a = v * TWO_PI + 1e-6f, b2 = v * 2.f + 1e-6f;
    acc += -0.5f * logf(a) - d * d / b2;
    // d nll / d mu = -(1/B) d logli / d mu,  d logli / d mu = 2 d / b2;  d logli / d var = -0.5 * 2 pi / a + 2 d^2 / b2^2
    dmu[i] = -(2.f * d / b2) / (float)B;
    dvar[i] = -(-0.5f * TWO_PI / a + 2.f * d * d / (b2 * b2)) / (float)B;
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) loss[0] = -acc / (float)B;
}
extern "C" int sg_normal_nll(const float* x, const float* mu, const float* var, int B, int K, float* loss, float* dmu, float* dvar, sg_stream_t s) {
  SG_CHECK(x && mu && var && loss && dmu && dvar && B > 0 && K > 0, "sg_normal_nll: bad args");
  hipLaunchKernelGGL(k_normal_nll, dim3(1), dim3(256), 0, (hipStream_t)s, x, mu, var, B, K, loss, dmu, dvar);
  SG_LAUNCH_CHECK();
  return 0;
}
