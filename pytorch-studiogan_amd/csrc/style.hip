// style.hip -- the StyleGAN2/3 native operators of the reference as gfx950 kernels:
//   sg_bias_act   = reference src/utils/style_ops/bias_act.cu:23-147 (bias + activation + gain + clamp, and the first / second order
//                   gradient evaluators its autograd wrapper bias_act.py:124-205 calls with grad = 1 / 2)
//   sg_upfirdn2d  = reference src/utils/style_ops/upfirdn2d.cu:29-104 (pad -> zero-insertion upsample -> FIR filter -> decimate, in one pass)
// Both are HBM-bound streaming operators: 16-byte loads / stores along the contiguous dimension, the per-element channel index of the
// bias kept out of the inner loop (one division per 16-byte vector), the FIR taps restricted to the polyphase component that meets
// non-zero samples ((fh / upy) x (fw / upx) multiply-adds per output, not fh x fw), filter coefficients in LDS.
// Host mirrors with the reference's signatures: pytorch-studiogan_amd/style_ops/{bias_act,upfirdn2d,filtered_lrelu}.py.
#include "common.h"
#include "../../include/sgamd.h"

// ---- bias_act ---------------------------------------------------------------------------------------------------------------------------
// Activation codes = the reference's cuda_idx (bias_act.py:20-30): 1 linear 2 relu 3 lrelu 4 tanh 5 sigmoid 6 elu 7 selu 8 softplus 9 swish.
// MODE 0: y = clamp(gain * act(x + b));  MODE 1: y = dy-like input x times act'(.) gain, zero where the forward output was clamped;
// MODE 2: second derivative term. For modes 1/2 the activation's derivative is expressed through the FORWARD OUTPUT yref (/ gain) where the
// reference does so (relu .. softplus) and through the forward input xref + b for swish.
struct BiasActArgs {
  const void* x; const void* b; const void* xref; const void* yref; const void* dy; void* y;
  long long n, step_b; int size_b;
  float alpha, gain, clamp;
};
__device__ __forceinline__ float ba_exp(float v) { return __expf(v); }
template <int ACT, int MODE>
__device__ __forceinline__ float bias_act_eval(float x, float xr, float yref, float dy, float alpha, float gain, float clamp) {
  // x: MODE 0 the biased input; MODE 1/2 the incoming gradient. xr: biased forward input (swish). yref: forward output.
  const float SELU_S = 1.0507009873554804934193349852946f, SELU_A = 1.6732632423543772848170429916717f;
  const float yy = gain != 0.f ? yref / gain : 0.f;
  float r = 0.f;
  if (ACT == 1) { r = MODE <= 1 ? x : 0.f; }
  if (ACT == 2) { r = MODE == 0 ? fmaxf(x, 0.f) : (MODE == 1 ? (yy > 0.f ? x : 0.f) : 0.f); }
  if (ACT == 3) { r = MODE == 0 ? (x > 0.f ? x : x * alpha) : (MODE == 1 ? (yy > 0.f ? x : x * alpha) : 0.f); }
  if (ACT == 4) {
    if (MODE == 0) { const float c = ba_exp(x), d = 1.f / c; r = x < -80.f ? -1.f : (x > 80.f ? 1.f : (c - d) / (c + d)); }
    if (MODE == 1) r = x * (1.f - yy * yy);
    if (MODE == 2) r = x * (1.f - yy * yy) * (-2.f * yy);
  }
  if (ACT == 5) {
    if (MODE == 0) r = x < -80.f ? 0.f : 1.f / (ba_exp(-x) + 1.f);
    if (MODE == 1) r = x * yy * (1.f - yy);
    if (MODE == 2) r = x * yy * (1.f - yy) * (1.f - 2.f * yy);
  }
  if (ACT == 6) {
    if (MODE == 0) r = x >= 0.f ? x : ba_exp(x) - 1.f;
    if (MODE == 1) r = yy >= 0.f ? x : x * (yy + 1.f);
    if (MODE == 2) r = yy >= 0.f ? 0.f : x * (yy + 1.f);
  }
  if (ACT == 7) {
    if (MODE == 0) r = x >= 0.f ? SELU_S * x : (SELU_S * SELU_A) * (ba_exp(x) - 1.f);
    if (MODE == 1) r = yy >= 0.f ? x * SELU_S : x * (yy + SELU_S * SELU_A);
    if (MODE == 2) r = yy >= 0.f ? 0.f : x * (yy + SELU_S * SELU_A);
  }
  if (ACT == 8) {
    if (MODE == 0) r = x > 80.f ? x : __logf(ba_exp(x) + 1.f);
    if (MODE == 1) r = x * (1.f - ba_exp(-yy));
    if (MODE == 2) { const float c = ba_exp(-yy); r = x * c * (1.f - c); }
  }
  if (ACT == 9) {
    if (MODE == 0) r = x < -80.f ? 0.f : x / (ba_exp(-x) + 1.f);
    else {
      const float c = ba_exp(xr), d = c + 1.f;
      if (MODE == 1) r = xr > 40.f ? x : x * c * (xr + d) / (d * d);
      else r = xr > 40.f ? 0.f : x * c * (xr * (2.f - d) + 2.f * d) / (d * d * d);
      yref = xr < -80.f ? 0.f : xr / (ba_exp(-xr) + 1.f) * gain;      // the clamp test below needs the forward output, recomputed from x
    }
  }
  r *= gain * dy;
  if (clamp >= 0.f) {
    if (MODE == 0) r = (r > -clamp && r < clamp) ? r : (r >= 0.f ? clamp : -clamp);
    else r = (yref > -clamp && yref < clamp) ? r : 0.f;
  }
  return r;
}
template <typename T> __device__ __forceinline__ void ba_load(const void* p, long long i, int cnt, float* o) {
  if (!p) return;
  const T* q = (const T*)p + i;
  if (cnt == ET<T>::VEC) unpack16<T>(*(const u32x4*)q, o);
  else for (int e = 0; e < cnt; e++) o[e] = to_f<T>(q[e]);
}
template <typename T, int ACT, int MODE>
__global__ __launch_bounds__(256) void k_bias_act(BiasActArgs a) {
  constexpr int V = ET<T>::VEC;
  const long long nvec = (a.n + V - 1) / V;
  for (long long vi = blockIdx.x * 256ll + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * 256) {
    const long long i0 = vi * V;
    const int cnt = (a.n - i0) >= V ? V : (int)(a.n - i0);
    float x[V], xr[V], yr[V], dy[V];
#pragma unroll
    for (int e = 0; e < V; e++) { x[e] = 0.f; xr[e] = 0.f; yr[e] = 0.f; dy[e] = 1.f; }
    ba_load<T>(a.x, i0, cnt, x);
    ba_load<T>(a.xref, i0, cnt, xr);
    ba_load<T>(a.yref, i0, cnt, yr);
    ba_load<T>(a.dy, i0, cnt, dy);
    if (a.b) {
      // channel of element i: (i / step_b) % size_b -- one division per vector, then a running (remainder, channel) pair
      long long q = i0 / a.step_b;
      long long rem = i0 - q * a.step_b;
      int ch = (int)(q % a.size_b);
#pragma unroll
      for (int e = 0; e < V; e++) {
        const float bv = to_f<T>(((const T*)a.b)[ch]);
        if (MODE == 0) x[e] += bv; else xr[e] += bv;
        if (++rem == a.step_b) { rem = 0; if (++ch == a.size_b) ch = 0; }
      }
    }
    float r[V];
#pragma unroll
    for (int e = 0; e < V; e++) r[e] = bias_act_eval<ACT, MODE>(x[e], xr[e], yr[e], dy[e], a.alpha, a.gain, a.clamp);
    T* o = (T*)a.y + i0;
    if (cnt == V) {
      u32x4 pk;
      if (V == 4) { pk[0] = __float_as_uint(r[0]); pk[1] = __float_as_uint(r[1]); pk[2] = __float_as_uint(r[2]); pk[3] = __float_as_uint(r[3]); }
      else { pk[0] = pack2bf(r[0], r[1]); pk[1] = pack2bf(r[2], r[3]); pk[2] = pack2bf(r[4 % V], r[5 % V]); pk[3] = pack2bf(r[6 % V], r[7 % V]); }
      *(u32x4*)o = pk;
    } else {
      for (int e = 0; e < cnt; e++) o[e] = from_f<T>(r[e]);
    }
  }
}
template <typename T, int ACT> static void bias_act_launch_mode(const BiasActArgs& a, int mode, hipStream_t st) {
  const long long nvec = (a.n + ET<T>::VEC - 1) / ET<T>::VEC;
  long long blocks = (nvec + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (blocks < 1) blocks = 1;
  if (mode == 0) hipLaunchKernelGGL((k_bias_act<T, ACT, 0>), dim3((int)blocks), dim3(256), 0, st, a);
  else if (mode == 1) hipLaunchKernelGGL((k_bias_act<T, ACT, 1>), dim3((int)blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_bias_act<T, ACT, 2>), dim3((int)blocks), dim3(256), 0, st, a);
}
template <typename T> static int bias_act_launch(const BiasActArgs& a, int act, int mode, hipStream_t st) {
  switch (act) {
    case 1: bias_act_launch_mode<T, 1>(a, mode, st); break;
    case 2: bias_act_launch_mode<T, 2>(a, mode, st); break;
    case 3: bias_act_launch_mode<T, 3>(a, mode, st); break;
    case 4: bias_act_launch_mode<T, 4>(a, mode, st); break;
    case 5: bias_act_launch_mode<T, 5>(a, mode, st); break;
    case 6: bias_act_launch_mode<T, 6>(a, mode, st); break;
    case 7: bias_act_launch_mode<T, 7>(a, mode, st); break;
    case 8: bias_act_launch_mode<T, 8>(a, mode, st); break;
    case 9: bias_act_launch_mode<T, 9>(a, mode, st); break;
    default: sg_set_error("sg_bias_act: activation code must be 1..9 (bias_act.py activation_funcs cuda_idx)"); return -1;
  }
  return 0;
}
extern "C" int sg_bias_act(int dtype, const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y, long long n,
                           long long step_b, int size_b, int grad, int act, float alpha, float gain, float clamp, sg_stream_t s) {
  SG_CHECK(x && y && n > 0, "sg_bias_act: x / y must be non-null, n > 0");
  SG_CHECK(grad >= 0 && grad <= 2, "sg_bias_act: grad must be 0, 1 or 2");
  SG_CHECK(!b || (step_b > 0 && size_b > 0), "sg_bias_act: bias needs step_b > 0 and size_b > 0");
  SG_CHECK((((uintptr_t)x | (uintptr_t)y | (uintptr_t)xref | (uintptr_t)yref | (uintptr_t)dy) & 15) == 0, "sg_bias_act: tensors must be 16-byte aligned");
  BiasActArgs a;
  a.x = x; a.b = b; a.xref = xref; a.yref = yref; a.dy = dy; a.y = y; a.n = n; a.step_b = b ? step_b : 1; a.size_b = b ? size_b : 1;
  a.alpha = alpha; a.gain = gain; a.clamp = clamp;
  int rc;
  if (dtype == SG_DTYPE_F32) rc = bias_act_launch<float>(a, act, grad, (hipStream_t)s);
  else if (dtype == SG_DTYPE_BF16) rc = bias_act_launch<bf16_t>(a, act, grad, (hipStream_t)s);
  else { sg_set_error("sg_bias_act: fp32 / bf16 only"); return -1; }
  if (rc) return rc;
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- upfirdn2d --------------------------------------------------------------------------------------------------------------------------
// x [N*C][H][W] (NCHW planes), f [fh][fw] fp32, y [N*C][Ho][Wo]:
//   y[p][oy][ox] = gain * sum_{fy, fx} F[fy][fx] * xu[oy * downy + fy - pady0][ox * downx + fx - padx0]
// xu = x with (up - 1) zeros after every sample; F = f flipped in both axes unless flip_filter (the reference's ref path flips the filter and
// then runs a cross-correlation, upfirdn2d.py:196-205: without the flag the operator is a true convolution).
// A tap (fy, fx) meets a real sample iff (oy * downy + fy - pady0) % upy == 0 (and likewise in x): only that polyphase component is visited.
struct UpfirdnArgs {
  const void* x; const float* f; void* y;
  int planes, H, W, Ho, Wo, fh, fw;
  int upx, upy, downx, downy, padx0, pady0;
  int flip; float gain;
};
template <typename T, int OPT /* outputs per thread along x */>
__global__ __launch_bounds__(256) void k_upfirdn2d(UpfirdnArgs a) {
  extern __shared__ float sf[];                       // the (flipped-as-needed) filter, gain folded in
  for (int i = threadIdx.x; i < a.fh * a.fw; i += 256) {
    const int fy = i / a.fw, fx = i - fy * a.fw;
    const int sy = a.flip ? fy : a.fh - 1 - fy, sx = a.flip ? fx : a.fw - 1 - fx;
    sf[i] = a.f[sy * a.fw + sx] * a.gain;
  }
  __syncthreads();
  const int wq = (a.Wo + OPT - 1) / OPT;
  const long long total = (long long)a.planes * a.Ho * wq;
  for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int xq = (int)(t % wq);
    const long long r = t / wq;
    const int oy = (int)(r % a.Ho);
    const int pl = (int)(r / a.Ho);
    const T* xp = (const T*)a.x + (long long)pl * a.H * a.W;
    // first tap row that meets a sample: u = oy * downy + fy - pady0 = 0 (mod upy), u >= 0
    const int uy0 = oy * a.downy - a.pady0;
    int fy0 = ((-uy0) % a.upy + a.upy) % a.upy;
    float acc[OPT];
#pragma unroll
    for (int e = 0; e < OPT; e++) acc[e] = 0.f;
    for (int fy = fy0; fy < a.fh; fy += a.upy) {
      const int u = uy0 + fy;
      if (u < 0) continue;
      const int iy = u / a.upy;
      if (iy >= a.H) break;
      const T* xrow = xp + (long long)iy * a.W;
      const float* frow = sf + fy * a.fw;
#pragma unroll
      for (int e = 0; e < OPT; e++) {
        const int ox = xq * OPT + e;
        if (ox >= a.Wo) break;
        const int ux0 = ox * a.downx - a.padx0;
        const int fx0 = ((-ux0) % a.upx + a.upx) % a.upx;
        float s = 0.f;
        for (int fx = fx0; fx < a.fw; fx += a.upx) {
          const int v = ux0 + fx;
          if (v < 0) continue;
          const int ix = v / a.upx;
          if (ix >= a.W) break;
          s += frow[fx] * to_f<T>(xrow[ix]);
        }
        acc[e] += s;
      }
    }
    T* yp = (T*)a.y + ((long long)pl * a.Ho + oy) * a.Wo + xq * OPT;
#pragma unroll
    for (int e = 0; e < OPT; e++) if (xq * OPT + e < a.Wo) yp[e] = from_f<T>(acc[e]);
  }
}
extern "C" int sg_upfirdn2d(int dtype, const void* x, const float* f, void* y, int planes, int H, int W, int fh, int fw, int upx, int upy,
                            int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip_filter, float gain, sg_stream_t s) {
  SG_CHECK(x && f && y && planes > 0 && H > 0 && W > 0, "sg_upfirdn2d: bad tensor arguments");
  SG_CHECK(fh >= 1 && fw >= 1 && fh * fw <= 4096, "sg_upfirdn2d: filter must be 1..4096 taps");
  SG_CHECK(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1, "sg_upfirdn2d: up / down factors must be >= 1");
  const int Wo = (W * upx + padx0 + padx1 - fw + downx) / downx, Ho = (H * upy + pady0 + pady1 - fh + downy) / downy;
  SG_CHECK(Wo >= 1 && Ho >= 1, "sg_upfirdn2d: empty output (upsampled + padded image smaller than the filter)");
  UpfirdnArgs a;
  a.x = x; a.f = f; a.y = y; a.planes = planes; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.fh = fh; a.fw = fw;
  a.upx = upx; a.upy = upy; a.downx = downx; a.downy = downy; a.padx0 = padx0; a.pady0 = pady0; a.flip = flip_filter ? 1 : 0; a.gain = gain;
  const long long total = (long long)planes * Ho * ((Wo + 3) / 4);
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  const size_t lds = sizeof(float) * fh * fw;
  if (dtype == SG_DTYPE_F32) hipLaunchKernelGGL((k_upfirdn2d<float, 4>), dim3((int)blocks), dim3(256), lds, (hipStream_t)s, a);
  else if (dtype == SG_DTYPE_BF16) hipLaunchKernelGGL((k_upfirdn2d<bf16_t, 4>), dim3((int)blocks), dim3(256), lds, (hipStream_t)s, a);
  else { sg_set_error("sg_upfirdn2d: fp32 / bf16 only"); return -1; }
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- filtered_lrelu, forward, separable filters: ONE launch ------------------------------------------------------------------------------
// reference src/utils/style_ops/filtered_lrelu.cu (the 1284-line tiled CUDA kernel) / filtered_lrelu.py:120-155 (its definition as a chain):
//   y = downfir_fd( clamp( lrelu( upfir_fu(x + b; up, padding, gain up^2), slope ) * gain ) ; down )
// Per workgroup: one output tile of one (n, c) plane. Everything between the input tile and the output tile lives in LDS as fp32:
//   in [IH][IW] (x + b, zeros outside the image) -> horizontal up-FIR -> t1 [IH][AW] -> vertical up-FIR + leaky ReLU + gain + clamp -> mid [AH][AW]
//   -> horizontal down-FIR -> t2 [AH][TW] -> vertical down-FIR -> y tile [TH][TW]
// so the up-sampled intermediate (up^2 x the input) never reaches HBM -- the chain of four launches writes and reads it twice.
// The up-FIR visits only the polyphase taps that meet a real sample (fu_n / up multiply-adds per value and axis).
struct FlreluArgs {
  const void* x; const void* b; const float* fu; const float* fd; void* y;
  int planes, C, H, W, Ho, Wo;
  int fu_n, fd_n, up, down, px0, py0;
  int flip; float gain, slope, clamp;
  int TH, TW, AH, AW, IH, IW;          // tile extents: output, activated intermediate, input
};
template <typename T>
__global__ __launch_bounds__(256) void k_filtered_lrelu(FlreluArgs a) {
  extern __shared__ float fl_sm[];
  float* s_fu = fl_sm;                                  // [fu_n] flipped as needed, x up (per-axis gain)
  float* s_fd = s_fu + a.fu_n;                          // [fd_n]
  float* s_in = s_fd + a.fd_n;                          // [IH][IW]
  float* s_t1 = s_in + a.IH * a.IW;                     // [IH][AW]
  float* s_mid = s_t1 + a.IH * a.AW;                    // [AH][AW]
  float* s_t2 = s_mid + a.AH * a.AW;                    // [AH][TW]
  const int tid = threadIdx.x;
  for (int i = tid; i < a.fu_n; i += 256) s_fu[i] = a.fu[a.flip ? i : a.fu_n - 1 - i] * (float)a.up;
  for (int i = tid; i < a.fd_n; i += 256) s_fd[i] = a.fd[a.flip ? i : a.fd_n - 1 - i];
  const int tx = (a.Wo + a.TW - 1) / a.TW, ty = (a.Ho + a.TH - 1) / a.TH;
  int bid = blockIdx.x;
  const int bx = bid % tx; bid /= tx;
  const int by = bid % ty;
  const int pl = bid / ty;
  const int ox0 = bx * a.TW, oy0 = by * a.TH;
  const int jx0 = ox0 * a.down, jy0 = oy0 * a.down;     // first intermediate (up-sampled + filtered) coordinate of the tile
  // first input sample any up-FIR tap of the tile can meet: smallest k >= 0 with k * up >= j0 - p0
  auto first_in = [&](int j0, int p0) { const int v = j0 - p0; return v <= 0 ? 0 : (v + a.up - 1) / a.up; };
  const int ix0 = first_in(jx0, a.px0), iy0 = first_in(jy0, a.py0);
  const T* xp = (const T*)a.x + (long long)pl * a.H * a.W;
  const float bias = a.b ? to_f<T>(((const T*)a.b)[pl % a.C]) : 0.f;
  for (int e = tid; e < a.IH * a.IW; e += 256) {
    const int r = e / a.IW, c = e - r * a.IW;
    const int iy = iy0 + r, ix = ix0 + c;
    s_in[e] = (iy < a.H && ix < a.W) ? to_f<T>(xp[(long long)iy * a.W + ix]) + bias : 0.f;
  }
  __syncthreads();
  // horizontal up-FIR: t1[r][j] = sum_t fu'[t] * xu[jx0 + j + t - px0]
  for (int e = tid; e < a.IH * a.AW; e += 256) {
    const int r = e / a.AW, j = e - r * a.AW;
    const int u0 = jx0 + j - a.px0;                     // zero-inserted coordinate of tap 0
    int t = ((-u0) % a.up + a.up) % a.up;               // first tap on a real sample
    float acc = 0.f;
    for (; t < a.fu_n; t += a.up) {
      const int u = u0 + t;
      if (u < 0) continue;
      const int c = u / a.up - ix0;
      if (c >= a.IW) break;
      acc += s_fu[t] * s_in[r * a.IW + c];              // (columns beyond the image hold zeros)
    }
    s_t1[e] = acc;
  }
  __syncthreads();
  // vertical up-FIR, then leaky ReLU, gain, clamp
  for (int e = tid; e < a.AH * a.AW; e += 256) {
    const int i = e / a.AW, j = e - i * a.AW;
    const int u0 = jy0 + i - a.py0;
    int t = ((-u0) % a.up + a.up) % a.up;
    float acc = 0.f;
    for (; t < a.fu_n; t += a.up) {
      const int u = u0 + t;
      if (u < 0) continue;
      const int r = u / a.up - iy0;
      if (r >= a.IH) break;
      acc += s_fu[t] * s_t1[r * a.AW + j];
    }
    float v = (acc > 0.f ? acc : acc * a.slope) * a.gain;
    if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
    s_mid[e] = v;
  }
  __syncthreads();
  // horizontal down-FIR: t2[i][ox] = sum_t fd'[t] * mid[i][ox * down + t]
  for (int e = tid; e < a.AH * a.TW; e += 256) {
    const int i = e / a.TW, ox = e - i * a.TW;
    float acc = 0.f;
    const float* m = s_mid + i * a.AW + ox * a.down;
    for (int t = 0; t < a.fd_n; t++) acc += s_fd[t] * m[t];
    s_t2[e] = acc;
  }
  __syncthreads();
  T* yp = (T*)a.y + (long long)pl * a.Ho * a.Wo;
  for (int e = tid; e < a.TH * a.TW; e += 256) {
    const int oy = e / a.TW, ox = e - oy * a.TW;
    if (oy0 + oy >= a.Ho || ox0 + ox >= a.Wo) continue;
    float acc = 0.f;
    for (int t = 0; t < a.fd_n; t++) acc += s_fd[t] * s_t2[(oy * a.down + t) * a.TW + ox];
    yp[(long long)(oy0 + oy) * a.Wo + ox0 + ox] = from_f<T>(acc);
  }
}
// fu / fd: separable fp32 filters on the device (fu_n / fd_n taps, >= 1; a one-tap {1} filter is the identity). Returns -3 when the tile does
// not fit the kernel's LDS budget (the caller then runs the chain of sg_bias_act / sg_upfirdn2d launches, which has no such limit).
extern "C" int sg_filtered_lrelu(int dtype, const void* x, const float* fu, const float* fd, const void* b, void* y, int N, int C, int H, int W,
                                 int fu_n, int fd_n, int up, int down, int px0, int px1, int py0, int py1, float gain, float slope, float clamp,
                                 int flip_filter, sg_stream_t s) {
  SG_CHECK(x && fu && fd && y && N > 0 && C > 0 && H > 0 && W > 0, "sg_filtered_lrelu: bad tensor arguments");
  SG_CHECK(fu_n >= 1 && fd_n >= 1 && up >= 1 && down >= 1 && gain > 0.f && slope >= 0.f, "sg_filtered_lrelu: bad filter / factor / gain arguments");
  const int Wo = (W * up + px0 + px1 - (fu_n - 1) - (fd_n - 1) + (down - 1)) / down;
  const int Ho = (H * up + py0 + py1 - (fu_n - 1) - (fd_n - 1) + (down - 1)) / down;
  SG_CHECK(Wo >= 1 && Ho >= 1, "sg_filtered_lrelu: empty output");
  FlreluArgs a;
  a.x = x; a.b = b; a.fu = fu; a.fd = fd; a.y = y; a.planes = N * C; a.C = C; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  a.fu_n = fu_n; a.fd_n = fd_n; a.up = up; a.down = down; a.px0 = px0; a.py0 = py0; a.flip = flip_filter ? 1 : 0;
  a.gain = gain; a.slope = slope; a.clamp = clamp;
  a.TH = Ho < 16 ? Ho : 16; a.TW = Wo < 16 ? Wo : 16;
  a.AH = (a.TH - 1) * down + fd_n; a.AW = (a.TW - 1) * down + fd_n;
  a.IH = (a.AH + fu_n - 1 + up - 1) / up + 1; a.IW = (a.AW + fu_n - 1 + up - 1) / up + 1;
  const size_t lds = sizeof(float) * ((size_t)fu_n + fd_n + (size_t)a.IH * a.IW + (size_t)a.IH * a.AW + (size_t)a.AH * a.AW + (size_t)a.AH * a.TW);
  if (lds > 64 * 1024) { sg_set_error("sg_filtered_lrelu: tile does not fit the LDS budget (use the operator chain)"); return -3; }
  const long long blocks = (long long)a.planes * ((Ho + a.TH - 1) / a.TH) * ((Wo + a.TW - 1) / a.TW);
  SG_CHECK(blocks < (1ll << 31), "sg_filtered_lrelu: too many tiles");
  if (dtype == SG_DTYPE_F32) hipLaunchKernelGGL(k_filtered_lrelu<float>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)s, a);
  else if (dtype == SG_DTYPE_BF16) hipLaunchKernelGGL(k_filtered_lrelu<bf16_t>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)s, a);
  else { sg_set_error("sg_filtered_lrelu: fp32 / bf16 only"); return -1; }
  SG_LAUNCH_CHECK();
  return 0;
}
