// ada.hip (ext) -- the image-side operators of the adaptive discriminator augmentation pipeline ("DiffAug / ADA"; reference
// src/utils/ada_aug.py:178-353, AdaAugment.forward; cfgs.AUG.series_augment for AUG.apply_ada, src/config.py:590-591). The pipeline's per-image parameters
// (a 3 x 3 inverse geometric transform and a 4 x 4 colour transform, composed from ~25 random draws) are made by the host mirror with the reference's own
// calls; what touches image bytes is here, fp32 NCHW, each with its exact adjoint (gather form, no atomics: bit-identical between runs):
//   sg_reflect_pad2d_fwd / _bwd     F.pad(mode='reflect') by per-call margins (ada_aug.py:265: the margins follow the transformed image corners)
//   sg_affine_sample_fwd / _bwd     F.affine_grid(align_corners=False) + grid_sample(bilinear, zeros padding) in one pass, no [N][Ho][Wo][2] grid in HBM
//                                   (ada_aug.py:276-277; the 2x up- / down-sampling around it is sg_upfirdn2d, csrc/style.hip)
//   sg_color_affine_fwd / _bwd      y = M[:, :3, :3] x + M[:, :3, 3] per image on RGB planes, or the luma form on one plane (ada_aug.py:339-347)
#include "../common.h"
#include "../../../include/sgamd.h"

__device__ __forceinline__ int ada_reflect(int k, int L) {
  if (k < 0) k = -k;
  if (k >= L) k = 2 * (L - 1) - k;
  return k < 0 ? 0 : (k > L - 1 ? L - 1 : k);
}

__global__ __launch_bounds__(256) void k_reflect_pad_fwd(const float* x, float* y, int H, int W, int l, int t, int Ho, int Wo, long long total) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i % Wo);
    const long long r = i / Wo;
    const int row = (int)(r % Ho);
    const long long pl = r / Ho;
    y[i] = x[(pl * H + ada_reflect(row - t, H)) * W + ada_reflect(j - l, W)];
  }
}
// dx[p][q] = sum over the padded positions that read (p, q): per axis the direct position and its (at most two) mirror images inside the padded extent
__device__ __forceinline__ int ada_pad_preimages(int p, int L, int lo, int Lo, int* out) {
  int n = 0;
  int k = p + lo;                      // direct
  if (k >= 0 && k < Lo) out[n++] = k;
  if (p > 0) { k = -p + lo; if (k >= 0 && k < Lo) out[n++] = k; }                              // mirrored at 0
  if (p < L - 1) { k = 2 * (L - 1) - p + lo; if (k >= 0 && k < Lo) out[n++] = k; }             // mirrored at L - 1
  return n;
}
__global__ __launch_bounds__(256) void k_reflect_pad_bwd(const float* dy, float* dx, int H, int W, int l, int t, int Ho, int Wo, long long total) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int q = (int)(i % W);
    const long long r = i / W;
    const int p = (int)(r % H);
    const long long pl = r / H;
    int ri[3], ci[3];
    const int nr = ada_pad_preimages(p, H, t, Ho, ri), nc = ada_pad_preimages(q, W, l, Wo, ci);
    float a = 0.f;
    for (int u = 0; u < nr; u++)
      for (int v = 0; v < nc; v++) a += dy[(pl * Ho + ri[u]) * Wo + ci[v]];
    dx[i] = a;
  }
}
static inline int nblk64(long long n) { long long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b)); }
static int pad_check(int planes, int H, int W, int l, int r, int t, int b) {
  return planes > 0 && H > 0 && W > 0 && l >= 0 && r >= 0 && t >= 0 && b >= 0 && l < W && r < W && t < H && b < H;
}
extern "C" int sg_reflect_pad2d_fwd(const float* x, float* y, int planes, int H, int W, int l, int r, int t, int b, sg_stream_t s) {
  SG_CHECK(x && y && pad_check(planes, H, W, l, r, t, b), "sg_reflect_pad2d_fwd: bad args (margins must be < the image size)");
  const int Ho = H + t + b, Wo = W + l + r;
  const long long total = (long long)planes * Ho * Wo;
  hipLaunchKernelGGL(k_reflect_pad_fwd, dim3(nblk64(total)), dim3(256), 0, (hipStream_t)s, x, y, H, W, l, t, Ho, Wo, total);
  SG_LAUNCH_CHECK();
  return 0;
}
extern "C" int sg_reflect_pad2d_bwd(const float* dy, float* dx, int planes, int H, int W, int l, int r, int t, int b, sg_stream_t s) {
  SG_CHECK(dy && dx && pad_check(planes, H, W, l, r, t, b), "sg_reflect_pad2d_bwd: bad args");
  const int Ho = H + t + b, Wo = W + l + r;
  const long long total = (long long)planes * H * W;
  hipLaunchKernelGGL(k_reflect_pad_bwd, dim3(nblk64(total)), dim3(256), 0, (hipStream_t)s, dy, dx, H, W, l, t, Ho, Wo, total);
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- affine grid + bilinear sampling ---------------------------------------------------------------------------------------------------------------------
// theta [N][2][3] maps normalised OUTPUT coordinates (x_o = (2 j + 1) / Wo - 1, y_o likewise) to normalised INPUT coordinates; the input pixel coordinate is
// ix = ((gx + 1) Wi - 1) / 2 (align_corners = False), samples outside the input read 0
__global__ __launch_bounds__(256) void k_affine_sample_fwd(const float* x, const float* theta, float* y, int C, int Hi, int Wi, int Ho, int Wo) {
  const int n = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Ho * Wo) return;
  const int i = t / Wo, j = t - i * Wo;
  const float* th = theta + 6 * n;
  const float xo = (2.f * j + 1.f) / (float)Wo - 1.f, yo = (2.f * i + 1.f) / (float)Ho - 1.f;
  const float gx = th[0] * xo + th[1] * yo + th[2], gy = th[3] * xo + th[4] * yo + th[5];
  const float ix = ((gx + 1.f) * (float)Wi - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)Hi - 1.f) * 0.5f;
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  const bool vx0 = x0 >= 0 && x0 < Wi, vx1 = x0 + 1 >= 0 && x0 + 1 < Wi, vy0 = y0 >= 0 && y0 < Hi, vy1 = y0 + 1 >= 0 && y0 + 1 < Hi;
  for (int c = 0; c < C; c++) {
    const float* p = x + ((long long)(n * C + c) * Hi) * Wi;
    float a = 0.f;
    if (vy0 && vx0) a += p[(long long)y0 * Wi + x0] * (wy0 * wx0);
    if (vy0 && vx1) a += p[(long long)y0 * Wi + x0 + 1] * (wy0 * wx1);
    if (vy1 && vx0) a += p[(long long)(y0 + 1) * Wi + x0] * (wy1 * wx0);
    if (vy1 && vx1) a += p[(long long)(y0 + 1) * Wi + x0 + 1] * (wy1 * wx1);
    y[((long long)(n * C + c) * Ho + i) * Wo + j] = a;
  }
}
// adjoint by gathering: input pixel (p, q) receives w(dy) from every output pixel whose sample point lies within one pixel of it. The map is affine, so those
// output pixels lie in the parallelogram A^-1 ([q - 1, q + 1] x [p - 1, p + 1]); its bounding box is scanned and the bilinear weight evaluated exactly as the
// forward evaluates it (same floor / fraction arithmetic), so forward and backward are adjoint to rounding.
__global__ __launch_bounds__(256) void k_affine_sample_bwd(const float* dy, const float* theta, float* dx, int C, int Hi, int Wi, int Ho, int Wo) {
  const int n = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Hi * Wi) return;
  const int p = t / Wi, q = t - p * Wi;
  const float* th = theta + 6 * n;
  // pixel-space forward map: ix = ax * j + bx * i + cx, iy = ay * j + by * i + cy
  const float sx = 0.5f * (float)Wi, sy = 0.5f * (float)Hi;
  const float ax = sx * th[0] * 2.f / (float)Wo, bx = sx * th[1] * 2.f / (float)Ho;
  const float ay = sy * th[3] * 2.f / (float)Wo, by = sy * th[4] * 2.f / (float)Ho;
  const float cx = sx * (th[0] * (1.f / (float)Wo - 1.f) + th[1] * (1.f / (float)Ho - 1.f) + th[2] + 1.f) - 0.5f;
  const float cy = sy * (th[3] * (1.f / (float)Wo - 1.f) + th[4] * (1.f / (float)Ho - 1.f) + th[5] + 1.f) - 0.5f;
  const float det = ax * by - bx * ay;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (fabsf(det) > 1e-12f) {
    // corners of the footprint square in output space
    float jmin = 3.0e38f, jmax = -3.0e38f, imin = 3.0e38f, imax = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float ux = (float)q + ((k & 1) ? 1.f : -1.f) - cx, uy = (float)p + ((k & 2) ? 1.f : -1.f) - cy;
      const float jj = (by * ux - bx * uy) / det, ii = (-ay * ux + ax * uy) / det;
      jmin = fminf(jmin, jj); jmax = fmaxf(jmax, jj); imin = fminf(imin, ii); imax = fmaxf(imax, ii);
    }
    int j0 = (int)floorf(jmin) - 1, j1 = (int)ceilf(jmax) + 1; j0 = j0 < 0 ? 0 : j0; j1 = j1 > Wo - 1 ? Wo - 1 : j1;
    int i0 = (int)floorf(imin) - 1, i1 = (int)ceilf(imax) + 1; i0 = i0 < 0 ? 0 : i0; i1 = i1 > Ho - 1 ? Ho - 1 : i1;
    for (int i = i0; i <= i1; i++)
      for (int j = j0; j <= j1; j++) {
        // the forward's own arithmetic for this output pixel
        const float xo = (2.f * j + 1.f) / (float)Wo - 1.f, yo = (2.f * i + 1.f) / (float)Ho - 1.f;
        const float gx = th[0] * xo + th[1] * yo + th[2], gy = th[3] * xo + th[4] * yo + th[5];
        const float ix = ((gx + 1.f) * (float)Wi - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)Hi - 1.f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        float wx, wy;
        if (q == x0) wx = 1.f - (ix - fx); else if (q == x0 + 1) wx = ix - fx; else continue;
        if (p == y0) wy = 1.f - (iy - fy); else if (p == y0 + 1) wy = iy - fy; else continue;
        const float w = wy * wx;
        for (int c = 0; c < C; c++) acc[c] += dy[((long long)(n * C + c) * Ho + i) * Wo + j] * w;
      }
  }
  for (int c = 0; c < C; c++) dx[((long long)(n * C + c) * Hi + p) * Wi + q] = acc[c];
}
extern "C" int sg_affine_sample_fwd(const float* x, const float* theta, float* y, int N, int C, int Hi, int Wi, int Ho, int Wo, sg_stream_t s) {
  SG_CHECK(x && theta && y && N > 0 && C > 0 && C <= 4 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "sg_affine_sample_fwd: bad args (1 <= C <= 4)");
  hipLaunchKernelGGL(k_affine_sample_fwd, dim3((Ho * Wo + 255) / 256, N), dim3(256), 0, (hipStream_t)s, x, theta, y, C, Hi, Wi, Ho, Wo);
  SG_LAUNCH_CHECK();
  return 0;
}
extern "C" int sg_affine_sample_bwd(const float* dy, const float* theta, float* dx, int N, int C, int Hi, int Wi, int Ho, int Wo, sg_stream_t s) {
  SG_CHECK(dy && theta && dx && N > 0 && C > 0 && C <= 4 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "sg_affine_sample_bwd: bad args (1 <= C <= 4)");
  hipLaunchKernelGGL(k_affine_sample_bwd, dim3((Hi * Wi + 255) / 256, N), dim3(256), 0, (hipStream_t)s, dy, theta, dx, C, Hi, Wi, Ho, Wo);
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- per-image colour transform ------------------------------------------------------------------------------------------------------------------------------
// M [N][3][4]: y_c = sum_k M[c][k] x_k + M[c][3] on three planes (transpose = 1: y_c = sum_k M[k][c] x_k, no offset: the adjoint); C == 1: y = x * a[n] + b[n]
// with (a, b) = M[n][0][0], M[n][0][3] prepared by the caller (the reference's luma form, ada_aug.py:343-345)
__global__ __launch_bounds__(256) void k_color_affine(const float* x, const float* M, float* y, int C, int HW, int transpose) {
  const int n = blockIdx.y;
  const float* m = M + 12 * n;
  const float* xi = x + (long long)n * C * HW;
  float* yo = y + (long long)n * C * HW;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < HW; t += gridDim.x * 256) {
    if (C == 1) {
      yo[t] = transpose ? xi[t] * m[0] : __fadd_rn(__fmul_rn(xi[t], m[0]), m[3]);
    } else {
      const float r = xi[t], g = xi[HW + t], b = xi[2 * HW + t];
      if (transpose) {
        yo[t] = m[0] * r + m[4] * g + m[8] * b;
        yo[HW + t] = m[1] * r + m[5] * g + m[9] * b;
        yo[2 * HW + t] = m[2] * r + m[6] * g + m[10] * b;
      } else {
        yo[t] = m[0] * r + m[1] * g + m[2] * b + m[3];
        yo[HW + t] = m[4] * r + m[5] * g + m[6] * b + m[7];
        yo[2 * HW + t] = m[8] * r + m[9] * g + m[10] * b + m[11];
      }
    }
  }
}
extern "C" int sg_color_affine(const float* x, const float* M, float* y, int N, int C, int HW, int transpose, sg_stream_t s) {
  SG_CHECK(x && M && y && N > 0 && (C == 1 || C == 3) && HW > 0, "sg_color_affine: RGB (3 planes) or L (1 plane) images");
  int bx = (HW + 255) / 256;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(k_color_affine, dim3(bx, N), dim3(256), 0, (hipStream_t)s, x, M, y, C, HW, transpose);
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- image-space filtering, additive noise, cutout (ada_aug.py:352-416) --------------------------------------------------------------------------------------
// sg_fir_reflect: y[n][c][..] = sum_t taps[n][t] x[n][c][reflect(pos + t - T/2)] along one axis (axis 0: columns, 1: rows) -- one of the two passes of the per-image
// separable amplification filter over a reflect-padded image (conv2d is a correlation: no tap flip). transpose = 1: the adjoint (each source sample gathers from its
// direct and mirrored positions).
__global__ __launch_bounds__(256) void k_fir_reflect(const float* x, const float* taps, float* y, int C, int H, int W, int T, int axis, int transpose) {
  const int n = blockIdx.y;
  const int HW = H * W;
  const float* w = taps + (long long)n * T;
  const int p = T / 2, Lax = axis ? H : W, stride = axis ? W : 1;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < C * HW; i += gridDim.x * 256) {
    const int c = i / HW, r = i - c * HW;
    const int row = r / W, col = r - row * W;
    const int pos = axis ? row : col;
    const float* line = x + ((long long)(n * C + c) * H + (axis ? 0 : row)) * W + (axis ? col : 0);
    float a = 0.f;
    if (!transpose) {
      for (int t = 0; t < T; t++) a += w[t] * line[(long long)ada_reflect(pos + t - p, Lax) * stride];
    } else {
      // sources k (padded coordinate) that read this sample: k = pos, -pos, 2 (L - 1) - pos; output index j = k - t + p
      int ks[3], nk = 0;
      ks[nk++] = pos;
      if (pos > 0) ks[nk++] = -pos;
      if (pos < Lax - 1) ks[nk++] = 2 * (Lax - 1) - pos;
      for (int u = 0; u < nk; u++)
        for (int t = 0; t < T; t++) {
          const int j = ks[u] - t + p;
          if (j >= 0 && j < Lax && ks[u] >= -p && ks[u] <= Lax - 1 + p) a += w[t] * line[(long long)j * stride];
        }
    }
    y[((long long)(n * C + c) * H + row) * W + col] = a;
  }
}
extern "C" int sg_fir_reflect(const float* x, const float* taps, float* y, int N, int C, int H, int W, int T, int axis, int transpose, sg_stream_t s) {
  SG_CHECK(x && taps && y && x != y && N > 0 && C > 0 && H > 0 && W > 0 && T > 0 && (T & 1) && (axis == 0 || axis == 1), "sg_fir_reflect: bad args (odd tap count)");
  SG_CHECK(T / 2 < (axis ? H : W), "sg_fir_reflect: the reflect padding (T / 2) must be smaller than the image");
  int bx = (C * H * W + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(k_fir_reflect, dim3(bx, N), dim3(256), 0, (hipStream_t)s, x, taps, y, C, H, W, T, axis, transpose);
  SG_LAUNCH_CHECK();
  return 0;
}
// y = (x + noise * sigma[n]) * mask_n; mask_n(i, j) = 0 inside the cutout rectangle |(j + 0.5) / W - cut[n][0]| < cut[n][2] / 2 and |(i + 0.5) / H - cut[n][1]| < cut[n][3] / 2
// (ada_aug.py:393-416). noise / sigma and cut may be NULL (that operator off); the adjoint is the same call with noise = NULL.
__global__ __launch_bounds__(256) void k_noise_cutout(const float* x, const float* noise, const float* sigma, const float* cut, float* y, int C, int H, int W) {
  const int n = blockIdx.y;
  const int HW = H * W;
  const float sg = sigma ? sigma[n] : 0.f;
  float cx = 0.f, cy = 0.f, hx = -1.f, hy = -1.f;
  if (cut) { cx = cut[4 * n]; cy = cut[4 * n + 1]; hx = __fdiv_rn(cut[4 * n + 2], 2.f); hy = __fdiv_rn(cut[4 * n + 3], 2.f); }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < C * HW; i += gridDim.x * 256) {
    const int r = i % HW, row = r / W, col = r - row * W;
    const long long o = (long long)n * C * HW + i;
    float v = x[o];
    if (noise) v = __fadd_rn(v, __fmul_rn(noise[o], sg));
    if (cut) {
      const bool keep_x = fabsf(__fsub_rn(__fdiv_rn((float)col + 0.5f, (float)W), cx)) >= hx;
      const bool keep_y = fabsf(__fsub_rn(__fdiv_rn((float)row + 0.5f, (float)H), cy)) >= hy;
      if (!(keep_x || keep_y)) v = __fmul_rn(v, 0.f);
    }
    y[o] = v;
  }
}
extern "C" int sg_ada_noise_cutout(const float* x, const float* noise, const float* sigma, const float* cut, float* y, int N, int C, int H, int W, sg_stream_t s) {
  SG_CHECK(x && y && N > 0 && C > 0 && H > 0 && W > 0 && (!noise == !sigma), "sg_ada_noise_cutout: bad args (noise and sigma come together)");
  int bx = (C * H * W + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(k_noise_cutout, dim3(bx, N), dim3(256), 0, (hipStream_t)s, x, noise, sigma, cut, y, C, H, W);
  SG_LAUNCH_CHECK();
  return 0;
}
