// second_order.hip (ext) -- the three small operators the create_graph pass through SelfAttention needs beyond the first-order kernels (R1 / gradient
// penalties on a discriminator WITH attention: reference src/utils/losses.py:301-316,355-361 through src/utils/ops.py:83-103, which torch autograd
// differentiates twice on its own). The first-order path (csrc/attn.hip: streaming softmax, nothing materialised) is untouched; a create_graph backward
// re-evaluates the block from differentiable primitives (functional.BmmFn / SoftmaxRowsFn / MaxPool2Fn / ScalePtrFn), and these are their missing adjoints:
//   sg_maxpool2_gather     y[q][c] = x[window(q)][idx[q][c]][c]          (adjoint of sg_maxpool2_bwd: the pooling with its argmax held fixed)
//   sg_softmax_rows_bwd2   gP = u * (dP - <P, dP>) - dP * <u, P>          (derivative of dS = P * (dP - <P, dP>) with respect to P, contracted with u)
//   sg_scale_by_ptr        y = sigma[0] * x                                (the attention block's learnt output gain as a device scalar)
#include "../common.h"
#include "../../../include/sgamd.h"

template <typename T> __global__ void k_maxpool2_gather(const T* x, int ldx, const uint8_t* idx, T* y, int ldy, int N, int H, int W, int C) {
  const int H2 = H / 2, W2 = W / 2;
  const long long total = (long long)N * H2 * W2 * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long q = i / C;
    long long t = q;
    const int w2 = (int)(t % W2); t /= W2;
    const int h2 = (int)(t % H2);
    const int n = (int)(t / H2);
    const int a = idx[i];
    y[q * ldy + c] = x[(((long long)n * H + 2 * h2 + (a >> 1)) * W + 2 * w2 + (a & 1)) * ldx + c];
  }
}
extern "C" int sg_maxpool2_gather(int dtype, const void* x, int ldx, const uint8_t* idx, void* y, int ldy, int N, int H, int W, int C, sg_stream_t s) {
  SG_CHECK(x && idx && y && N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && ldx >= C && ldy >= C, "sg_maxpool2_gather: bad args");
  SG_CHECK(dtype == SG_DTYPE_F32 || dtype == SG_DTYPE_BF16, "sg_maxpool2_gather: bad dtype");
  const long long total = (long long)N * (H / 2) * (W / 2) * C;
  long long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (dtype == SG_DTYPE_F32) hipLaunchKernelGGL(k_maxpool2_gather<float>, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, (const float*)x, ldx, idx, (float*)y, ldy, N, H, W, C);
  else hipLaunchKernelGGL(k_maxpool2_gather<bf16_t>, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, ldx, idx, (bf16_t*)y, ldy, N, H, W, C);
  SG_LAUNCH_CHECK();
  return 0;
}

// one wave per row: two dot products over the row, then the elementwise combination
__global__ __launch_bounds__(256) void k_softmax_rows_bwd2(const float* P, const float* dP, const float* u, float* gP, long long rows, int cols) {
  const long long r = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = P + r * cols;
  const float* d = dP + r * cols;
  const float* v = u + r * cols;
  float s = 0.f, t = 0.f;
  for (int c = lane; c < cols; c += 64) { const float pc = p[c]; s += pc * d[c]; t += pc * v[c]; }
  s = wave_sum(s);
  t = wave_sum(t);
  float* o = gP + r * cols;
  for (int c = lane; c < cols; c += 64) o[c] = v[c] * (d[c] - s) - d[c] * t;
}
extern "C" int sg_softmax_rows_bwd2(const float* P, const float* dP, const float* u, float* gP, long long rows, int cols, sg_stream_t s) {
  SG_CHECK(P && dP && u && gP && rows > 0 && cols > 0, "sg_softmax_rows_bwd2: bad args");
  SG_CHECK((rows + 3) / 4 < (1ll << 31), "sg_softmax_rows_bwd2: too many rows");
  hipLaunchKernelGGL(k_softmax_rows_bwd2, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)s, P, dP, u, gP, rows, cols);
  SG_LAUNCH_CHECK();
  return 0;
}

template <typename T> __global__ void k_scale_by_ptr(const T* x, const float* sigma, T* y, long long n) {
  const float g = sigma[0];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = from_f<T>(g * to_f<T>(x[i]));
}
extern "C" int sg_scale_by_ptr(int dtype, const void* x, const float* sigma, void* y, long long n, sg_stream_t s) {
  SG_CHECK(x && sigma && y && n > 0, "sg_scale_by_ptr: bad args");
  SG_CHECK(dtype == SG_DTYPE_F32 || dtype == SG_DTYPE_BF16, "sg_scale_by_ptr: bad dtype");
  long long blocks = (n + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (dtype == SG_DTYPE_F32) hipLaunchKernelGGL(k_scale_by_ptr<float>, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, (const float*)x, sigma, (float*)y, n);
  else hipLaunchKernelGGL(k_scale_by_ptr<bf16_t>, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, sigma, (bf16_t*)y, n);
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- tanh of the generator's output image in a create_graph pass (latent optimisation, reference src/utils/losses.py:278-298: the gradient of D(G(z)) with
// respect to z is itself differentiated) -- fp32 NCHW:  sg_tanh_bwd  t = dy * (1 - y^2)      sg_tanh_bwd2  out = -2 g dy y  (d t / d y contracted with g)
__global__ __launch_bounds__(256) void k_tanh_bwd(const float* dy, const float* y, float* t, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) { const float v = y[i]; t[i] = dy[i] * (1.f - v * v); }
}
__global__ __launch_bounds__(256) void k_tanh_bwd2(const float* g, const float* dy, const float* y, float* out, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = -2.f * g[i] * dy[i] * y[i];
}
extern "C" int sg_tanh_bwd(const float* dy, const float* y, float* t, long long n, sg_stream_t s) {
  SG_CHECK(dy && y && t && n > 0, "sg_tanh_bwd: bad args");
  long long b = (n + 255) / 256; if (b > 16384) b = 16384;
  hipLaunchKernelGGL(k_tanh_bwd, dim3((int)b), dim3(256), 0, (hipStream_t)s, dy, y, t, n);
  SG_LAUNCH_CHECK();
  return 0;
}
extern "C" int sg_tanh_bwd2(const float* g, const float* dy, const float* y, float* out, long long n, sg_stream_t s) {
  SG_CHECK(g && dy && y && out && n > 0, "sg_tanh_bwd2: bad args");
  long long b = (n + 255) / 256; if (b > 16384) b = 16384;
  hipLaunchKernelGGL(k_tanh_bwd2, dim3((int)b), dim3(256), 0, (hipStream_t)s, g, dy, y, out, n);
  SG_LAUNCH_CHECK();
  return 0;
}
