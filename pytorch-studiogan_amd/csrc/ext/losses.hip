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
