// augment.hip -- differentiable image augmentations of the discriminator's inputs, and the consistency regularisers' squared-error loss
// (the callers on the image side of the training step):
//   * DiffAugment (reference src/utils/diffaug.py:35-102, wired as cfgs.AUG.series_augment at src/config.py:586-587 and called on the real and the fake
//     batch in front of every discriminator forward, src/worker.py:276-278,549-550): brightness, saturation, contrast, integer translation with zero fill,
//     cutout;
//   * the CR / bCR augmentation (reference src/utils/cr.py:13-48, cfgs.AUG.parallel_augment, src/worker.py:326-354): horizontal flip and integer translation
//     over a reflect-padded image;
//   * torch.nn.MSELoss between two logit / embedding / image tensors (the `l2_loss` of src/worker.py:116,329-361,603).
//
// The reference runs 5 + 9 elementwise / gather launches over fp32 NCHW images per DiffAugment call (three reductions among them) and builds an int64 index
// grid of N x H x W x 3 entries for each translation. Here one call is ONE gather pass (plus one partial-sum pass when contrast is on): every output pixel
// knows its source pixel in closed form, the per-image contrast mean is a fixed-order two-level sum (bit-identical between runs and ranks), and the backward
// is the transposed gather -- each source pixel collects its (at most 3 x 3, reflect mode) images -- so no atomics anywhere.
//
// Fixed operator order inside one call: brightness -> saturation -> contrast -> flip -> translation -> cutout (the reference's "color,translation,cutout"
// policy and cr.py's flip -> translation; the host mirror splits other policy orders into several calls). All arithmetic is fp32 with the reference's
// operation order and no fused multiply-adds, so the only deviation from the reference's CPU result is the summation order of the contrast mean.
// HBM-bound: reads x once (twice with contrast), writes y once: 8-12 bytes per element against the reference's ~120.
#include "../common.h"
#include "../../../include/sgamd.h"

#define AUG_PARTS_MAX 16

struct AugGeom { int tx, ty, cx, cy, flip; };

__device__ __forceinline__ AugGeom aug_geom(const sg_aug_desc& d, int n) {
  AugGeom g = {0, 0, 0, 0, 0};
  if (d.geom) { const int* p = d.geom + 5 * n; g.tx = p[0]; g.ty = p[1]; g.cx = p[2]; g.cy = p[3]; g.flip = p[4]; }
  return g;
}
// source index of output index i along an axis of length L shifted by t: -1 = the zero fill
__device__ __forceinline__ int aug_src(int i, int t, int L, int ops) {
  int k = i + t;
  if (ops & SG_AUG_TRANSLATE_REFLECT) {                                  // F.pad(mode='reflect'): no edge repeat
    if (k < 0) k = -k; else if (k >= L) k = 2 * (L - 1) - k;
    return k < 0 ? 0 : (k > L - 1 ? L - 1 : k);                           // (|t| < L by contract; a table that breaks it must not read outside the image)
  }
  if (ops & SG_AUG_TRANSLATE) return (k >= 0 && k < L) ? k : -1;                                                 // clamp into the 1-pixel zero frame
  return i;
}
// output indices whose source is p (transposed gather); returns their count (<= 3)
__device__ __forceinline__ int aug_preimages(int p, int t, int L, int ops, int* out) {
  if (!(ops & (SG_AUG_TRANSLATE | SG_AUG_TRANSLATE_REFLECT))) { out[0] = p; return 1; }
  int n = 0, i = p - t;
  if (i >= 0 && i < L) out[n++] = i;
  if (ops & SG_AUG_TRANSLATE_REFLECT) {
    if (p > 0) { i = -p - t; if (i >= 0 && i < L) out[n++] = i; }
    if (p < L - 1) { i = 2 * (L - 1) - p - t; if (i >= 0 && i < L) out[n++] = i; }
  }
  return n;
}
// cutout window [lo, hi] along an axis (diffaug.py:88-95: the window's indices are CLAMPED into the image, not cropped)
__device__ __forceinline__ void aug_cut(int centre, int size, int L, int& lo, int& hi) {
  const int a = centre - size / 2, b = a + size - 1;
  lo = a < 0 ? 0 : (a > L - 1 ? L - 1 : a);
  hi = b < 0 ? 0 : (b > L - 1 ? L - 1 : b);
}
// colour chain of one pixel up to (not including) the contrast step: v[c] <- saturation(brightness(v[c]))
template <int C> __device__ __forceinline__ void aug_bright_sat(float (&v)[C], int ops, float b, float s) {
  if (ops & SG_AUG_BRIGHTNESS) {
#pragma unroll
    for (int c = 0; c < C; c++) v[c] = __fadd_rn(v[c], b);
  }
  if (ops & SG_AUG_SATURATION) {
    float m = v[0];
#pragma unroll
    for (int c = 1; c < C; c++) m = __fadd_rn(m, v[c]);
    m = __fdiv_rn(m, (float)C);
#pragma unroll
    for (int c = 0; c < C; c++) v[c] = __fadd_rn(__fmul_rn(__fsub_rn(v[c], m), s), m);
  }
}

// ---- per-image partial sums of the saturated image (the contrast mean): part[n][blockIdx.x] ---------------------------------------------------------------
template <int C> __global__ __launch_bounds__(256) void k_aug_sum(sg_aug_desc d, const float* x, float* part) {
  __shared__ float sm[4];
  const int n = blockIdx.y, HW = d.H * d.W;
  const float b = d.color ? d.color[3 * n] : 0.f, s = d.color ? d.color[3 * n + 1] : 1.f;
  const float* xi = x + (long long)n * C * HW;
  float acc = 0.f;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < HW; t += gridDim.x * 256) {
    float v[C];
    _Pragma("unroll") for (int c = 0; c < C; c++) v[c] = xi[(long long)c * HW + t];
    aug_bright_sat<C>(v, d.ops, b, s);
    _Pragma("unroll") for (int c = 0; c < C; c++) acc += v[c];
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) part[n * gridDim.x + blockIdx.x] = acc;
}

// ---- forward: one thread = V adjacent output columns of one row, all channels ------------------------------------------------------------------------------
template <int C, int V> __global__ __launch_bounds__(256) void k_aug_fwd(sg_aug_desc d, const float* x, float* y, const float* part, int parts) {
  const int n = blockIdx.y, HW = d.H * d.W, Wv = d.W / V;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= d.H * Wv) return;
  const int i = t / Wv, j0 = (t - i * Wv) * V;
  const AugGeom g = aug_geom(d, n);
  const float b = d.color ? d.color[3 * n] : 0.f, s = d.color ? d.color[3 * n + 1] : 1.f, cc = d.color ? d.color[3 * n + 2] : 1.f;
  float M = 0.f;
  if (d.ops & SG_AUG_CONTRAST) {
    for (int k = 0; k < parts; k++) M += part[n * parts + k];            // fixed order: the same value in every thread, run and rank
    M = __fdiv_rn(M, (float)C * (float)HW);
  }
  int clo = 0, chi = -1, rlo = 0, rhi = -1;
  if (d.ops & SG_AUG_CUTOUT) { aug_cut(g.cx, d.cut_h, d.H, rlo, rhi); aug_cut(g.cy, d.cut_w, d.W, clo, chi); }
  const int si = aug_src(i, g.tx, d.H, d.ops);
  const float* xi = x + (long long)n * C * HW;
  float* yo = y + (long long)n * C * HW + (long long)i * d.W + j0;
  float o[C][V];
#pragma unroll
  for (int v = 0; v < V; v++) {
    const int j = j0 + v;
    int sj = aug_src(j, g.ty, d.W, d.ops);
    if (sj >= 0 && (d.ops & SG_AUG_FLIP) && g.flip) sj = d.W - 1 - sj;
    const bool cut = (d.ops & SG_AUG_CUTOUT) && i >= rlo && i <= rhi && j >= clo && j <= chi;
    if (si < 0 || sj < 0 || cut) {
      _Pragma("unroll") for (int c = 0; c < C; c++) o[c][v] = 0.f;                      // zero frame of the translation / cutout mask (x * 0)
      continue;
    }
    float px[C];
    _Pragma("unroll") for (int c = 0; c < C; c++) px[c] = xi[(long long)c * HW + (long long)si * d.W + sj];
    aug_bright_sat<C>(px, d.ops, b, s);
    if (d.ops & SG_AUG_CONTRAST) {
      _Pragma("unroll") for (int c = 0; c < C; c++) px[c] = __fadd_rn(__fmul_rn(__fsub_rn(px[c], M), cc), M);
    }
    _Pragma("unroll") for (int c = 0; c < C; c++) o[c][v] = px[c];
  }
  _Pragma("unroll") for (int c = 0; c < C; c++) {
    if constexpr (V == 4) {
      f32x4 r = {o[c][0], o[c][1], o[c][2], o[c][3]};
      *(f32x4*)(yo + (long long)c * HW) = r;
    } else {
      yo[(long long)c * HW] = o[c][0];
    }
  }
}

// ---- backward, pass 1 (contrast only): partial sums over the image of the gradient that reaches the colour chain ------------------------------------------
template <int C> __global__ __launch_bounds__(256) void k_aug_bwd_sum(sg_aug_desc d, const float* dy, float* part) {
  __shared__ float sm[4];
  const int n = blockIdx.y, HW = d.H * d.W;
  const AugGeom g = aug_geom(d, n);
  int clo = 0, chi = -1, rlo = 0, rhi = -1;
  if (d.ops & SG_AUG_CUTOUT) { aug_cut(g.cx, d.cut_h, d.H, rlo, rhi); aug_cut(g.cy, d.cut_w, d.W, clo, chi); }
  const float* gi = dy + (long long)n * C * HW;
  float acc = 0.f;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < HW; t += gridDim.x * 256) {
    const int i = t / d.W, j = t - i * d.W;
    const bool cut = (d.ops & SG_AUG_CUTOUT) && i >= rlo && i <= rhi && j >= clo && j <= chi;
    if (cut || aug_src(i, g.tx, d.H, d.ops) < 0 || aug_src(j, g.ty, d.W, d.ops) < 0) continue;
    _Pragma("unroll") for (int c = 0; c < C; c++) acc += gi[(long long)c * HW + t];
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) part[n * gridDim.x + blockIdx.x] = acc;
}

// ---- backward, pass 2: one thread = V adjacent SOURCE columns of one row; dx = colour^T (gather^T (mask * dy)) --------------------------------------------
template <int C, int V> __global__ __launch_bounds__(256) void k_aug_bwd(sg_aug_desc d, const float* dy, float* dx, const float* part, int parts) {
  const int n = blockIdx.y, HW = d.H * d.W, Wv = d.W / V;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= d.H * Wv) return;
  const int p = t / Wv, q0 = (t - p * Wv) * V;
  const AugGeom g = aug_geom(d, n);
  const float s = d.color ? d.color[3 * n + 1] : 1.f, cc = d.color ? d.color[3 * n + 2] : 1.f;
  float G = 0.f;
  if (d.ops & SG_AUG_CONTRAST) {
    for (int k = 0; k < parts; k++) G += part[n * parts + k];
    G = G / ((float)C * (float)HW);
  }
  int clo = 0, chi = -1, rlo = 0, rhi = -1;
  if (d.ops & SG_AUG_CUTOUT) { aug_cut(g.cx, d.cut_h, d.H, rlo, rhi); aug_cut(g.cy, d.cut_w, d.W, clo, chi); }
  int ri[3];
  const int nr = aug_preimages(p, g.tx, d.H, d.ops, ri);
  const float* gi = dy + (long long)n * C * HW;
  float* xo = dx + (long long)n * C * HW + (long long)p * d.W + q0;
  float o[C][V];
#pragma unroll
  for (int v = 0; v < V; v++) {
    const int q = q0 + v;
    const int qf = ((d.ops & SG_AUG_FLIP) && g.flip) ? d.W - 1 - q : q;     // column of this source pixel in the flipped image
    int ci[3];
    const int nc = aug_preimages(qf, g.ty, d.W, d.ops, ci);
    float a[C];
    _Pragma("unroll") for (int c = 0; c < C; c++) a[c] = 0.f;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if (r >= nr || k >= nc) continue;
        const int i = ri[r], j = ci[k];
        if ((d.ops & SG_AUG_CUTOUT) && i >= rlo && i <= rhi && j >= clo && j <= chi) continue;
        _Pragma("unroll") for (int c = 0; c < C; c++) a[c] += gi[(long long)c * HW + (long long)i * d.W + j];
      }
    if (d.ops & SG_AUG_CONTRAST) {
      _Pragma("unroll") for (int c = 0; c < C; c++) a[c] = cc * a[c] + (1.f - cc) * G;
    }
    if (d.ops & SG_AUG_SATURATION) {
      float m = a[0];
      _Pragma("unroll") for (int c = 1; c < C; c++) m += a[c];
      m = m / (float)C;
      _Pragma("unroll") for (int c = 0; c < C; c++) a[c] = s * a[c] + (1.f - s) * m;
    }
    _Pragma("unroll") for (int c = 0; c < C; c++) o[c][v] = a[c];
  }
  _Pragma("unroll") for (int c = 0; c < C; c++) {
    if constexpr (V == 4) {
      f32x4 r = {o[c][0], o[c][1], o[c][2], o[c][3]};
      *(f32x4*)(xo + (long long)c * HW) = r;
    } else {
      xo[(long long)c * HW] = o[c][0];
    }
  }
}

#define AUG_DISPATCH_C(c, ...)                              \
  do {                                                      \
    switch (c) {                                            \
      case 1: { constexpr int CC = 1; __VA_ARGS__; } break; \
      case 2: { constexpr int CC = 2; __VA_ARGS__; } break; \
      case 3: { constexpr int CC = 3; __VA_ARGS__; } break; \
      default: { constexpr int CC = 4; __VA_ARGS__; } break; \
    }                                                       \
  } while (0)

static int aug_check(const sg_aug_desc* d, const char** why) {
  if (!d || d->N <= 0 || d->C <= 0 || d->C > 4 || d->H <= 0 || d->W <= 0) { *why = "sg_augment: N, H, W > 0 and 1 <= C <= 4 (image batches)"; return -1; }
  const int known = SG_AUG_BRIGHTNESS | SG_AUG_SATURATION | SG_AUG_CONTRAST | SG_AUG_FLIP | SG_AUG_TRANSLATE | SG_AUG_TRANSLATE_REFLECT | SG_AUG_CUTOUT;
  if (d->ops & ~known) { *why = "sg_augment: unknown operator bit"; return -1; }
  if ((d->ops & SG_AUG_TRANSLATE) && (d->ops & SG_AUG_TRANSLATE_REFLECT)) { *why = "sg_augment: zero-fill and reflect translation are exclusive"; return -1; }
  if ((d->ops & (SG_AUG_BRIGHTNESS | SG_AUG_SATURATION | SG_AUG_CONTRAST)) && !d->color) { *why = "sg_augment: colour operators need the colour table"; return -1; }
  if ((d->ops & (SG_AUG_FLIP | SG_AUG_TRANSLATE | SG_AUG_TRANSLATE_REFLECT | SG_AUG_CUTOUT)) && !d->geom) { *why = "sg_augment: geometric operators need the geometry table"; return -1; }
  if ((d->ops & SG_AUG_CUTOUT) && (d->cut_h <= 0 || d->cut_w <= 0)) { *why = "sg_augment: cutout window must be positive"; return -1; }
  if ((d->ops & SG_AUG_TRANSLATE_REFLECT) && (d->max_t < 0 || d->max_t >= d->H || d->max_t >= d->W)) { *why = "sg_augment: reflect translation needs 0 <= max_t < min(H, W)"; return -1; }
  return 0;
}
static int aug_parts(const sg_aug_desc* d) {
  const int n = (d->H * d->W + 1023) / 1024;
  return n < 1 ? 1 : (n > AUG_PARTS_MAX ? AUG_PARTS_MAX : n);
}

extern "C" int sg_augment_work_floats(const sg_aug_desc* d) { return d && d->N > 0 ? d->N * AUG_PARTS_MAX : 0; }

extern "C" int sg_augment_fwd(const sg_aug_desc* d, const float* x, float* y, float* work, sg_stream_t s) {
  const char* why = nullptr;
  if (aug_check(d, &why)) { sg_set_error(why); return -1; }
  SG_CHECK(x && y && x != y, "sg_augment_fwd: x, y must be distinct device buffers");
  SG_CHECK(!(d->ops & SG_AUG_CONTRAST) || work, "sg_augment_fwd: contrast needs the work buffer (sg_augment_work_floats)");
  const int parts = aug_parts(d);
  if (d->ops & SG_AUG_CONTRAST) {
    AUG_DISPATCH_C(d->C, hipLaunchKernelGGL(k_aug_sum<CC>, dim3(parts, d->N), dim3(256), 0, (hipStream_t)s, *d, x, work));
    SG_LAUNCH_CHECK();
  }
  if (d->W % 4 == 0 && ((uintptr_t)y & 15) == 0) {
    AUG_DISPATCH_C(d->C, hipLaunchKernelGGL((k_aug_fwd<CC, 4>), dim3((d->H * (d->W / 4) + 255) / 256, d->N), dim3(256), 0, (hipStream_t)s, *d, x, y, work, parts));
  } else {
    AUG_DISPATCH_C(d->C, hipLaunchKernelGGL((k_aug_fwd<CC, 1>), dim3((d->H * d->W + 255) / 256, d->N), dim3(256), 0, (hipStream_t)s, *d, x, y, work, parts));
  }
  SG_LAUNCH_CHECK();
  return 0;
}

extern "C" int sg_augment_bwd(const sg_aug_desc* d, const float* dy, float* dx, float* work, sg_stream_t s) {
  const char* why = nullptr;
  if (aug_check(d, &why)) { sg_set_error(why); return -1; }
  SG_CHECK(dy && dx && dy != dx, "sg_augment_bwd: dy, dx must be distinct device buffers");
  SG_CHECK(!(d->ops & SG_AUG_CONTRAST) || work, "sg_augment_bwd: contrast needs the work buffer (sg_augment_work_floats)");
  const int parts = aug_parts(d);
  if (d->ops & SG_AUG_CONTRAST) {
    AUG_DISPATCH_C(d->C, hipLaunchKernelGGL(k_aug_bwd_sum<CC>, dim3(parts, d->N), dim3(256), 0, (hipStream_t)s, *d, dy, work));
    SG_LAUNCH_CHECK();
  }
  if (d->W % 4 == 0 && ((uintptr_t)dx & 15) == 0) {
    AUG_DISPATCH_C(d->C, hipLaunchKernelGGL((k_aug_bwd<CC, 4>), dim3((d->H * (d->W / 4) + 255) / 256, d->N), dim3(256), 0, (hipStream_t)s, *d, dy, dx, work, parts));
  } else {
    AUG_DISPATCH_C(d->C, hipLaunchKernelGGL((k_aug_bwd<CC, 1>), dim3((d->H * d->W + 255) / 256, d->N), dim3(256), 0, (hipStream_t)s, *d, dy, dx, work, parts));
  }
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- mean squared error (torch.nn.MSELoss, reduction 'mean') ---------------------------------------------------------------------------------------------
// forward: fixed-order two-level sum (<= 1024 partials, then one workgroup in fp64); backward: da = gout * 2 (a - b) / n, db = -da
#define MSE_PARTS_MAX 1024
__global__ __launch_bounds__(256) void k_mse_part(const float* a, const float* b, long long n, float* part) {
  __shared__ float sm[4];
  float acc = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) { const float dlt = a[i] - b[i]; acc += dlt * dlt; }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_mse_final(const float* part, int parts, long long n, float* loss) {
  __shared__ double smd[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < parts; i += 256) acc += (double)part[i];
  acc = block_sum_256_d(acc, smd);
  if (threadIdx.x == 0) loss[0] = (float)(acc / (double)n);
}
__global__ __launch_bounds__(256) void k_mse_bwd(const float* a, const float* b, const float* gout, long long n, float* da, float* db) {
  const float k = gout[0] * 2.f / (float)n;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = k * (a[i] - b[i]);
    if (da) da[i] = v;
    if (db) db[i] = -v;
  }
}
extern "C" int sg_mse_work_floats(void) { return MSE_PARTS_MAX; }
extern "C" int sg_mse_fwd(const float* a, const float* b, long long n, float* work, float* loss, sg_stream_t s) {
  SG_CHECK(a && b && work && loss && n > 0, "sg_mse_fwd: bad args");
  long long parts = (n + 2047) / 2048;
  if (parts > MSE_PARTS_MAX) parts = MSE_PARTS_MAX;
  hipLaunchKernelGGL(k_mse_part, dim3((int)parts), dim3(256), 0, (hipStream_t)s, a, b, n, work);
  SG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_mse_final, dim3(1), dim3(256), 0, (hipStream_t)s, work, (int)parts, n, loss);
  SG_LAUNCH_CHECK();
  return 0;
}
extern "C" int sg_mse_bwd(const float* a, const float* b, const float* gout, long long n, float* da, float* db, sg_stream_t s) {
  SG_CHECK(a && b && gout && n > 0 && (da || db), "sg_mse_bwd: bad args");
  long long blocks = (n + 1023) / 1024;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_mse_bwd, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, a, b, gout, n, da, db);
  SG_LAUNCH_CHECK();
  return 0;
}
