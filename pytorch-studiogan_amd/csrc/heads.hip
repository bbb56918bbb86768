// heads.hip -- class-conditioning heads and their losses (reference src/utils/losses.py:40-165,242-252; heads of
// src/models/big_resnet.py:307-333,380-413): auxiliary classifier (AC: cross entropy), ContraGAN (2C: conditional contrastive loss),
// ReACGAN (D2D-CE: data-to-data cross entropy), multi-hinge (MH: Crammer-Singer), multi-discriminator (MD: per-class logit).
// The tensors are [batch, classes] / [batch, embed] / [batch, batch]: every kernel is one wave per row, forward and the analytic
// gradient in the same launch (the losses are means over rows, so the gradient of the mean is known when the row is done).
#include "common.h"
#include "../../include/sgamd.h"

// y[r] = x[r] / max(|x[r]|, eps); inv[r] = +-1 / max(|x[r]|, eps)        (torch.nn.functional.normalize(dim=1); cosine-similarity rows)
// The SIGN of inv[r] is the "clamped" flag of the backward: negative = |x[r]| < eps, the row was divided by the constant eps.
__global__ __launch_bounds__(256) void k_row_normalize_fwd(const float* x, float* y, float* inv, int rows, int cols, float eps) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int c = lane; c < cols; c += 64) { const float v = x[(long long)r * cols + c]; acc += v * v; }
  acc = wave_sum(acc);
  const float nrm = sqrtf(acc);
  const float iv = 1.f / fmaxf(nrm, eps);
  for (int c = lane; c < cols; c += 64) y[(long long)r * cols + c] = x[(long long)r * cols + c] * iv;
  if (lane == 0) inv[r] = nrm < eps ? -iv : iv;
}
// dx = inv * (dy - y <y, dy>); rows whose norm was clamped (inv < 0): y = x / eps is linear in x, so dx = dy / eps -- what autograd
// gives for x / clamp_min(|x|, eps) (zero-initialised or collapsed embeddings / proxies)
__global__ __launch_bounds__(256) void k_row_normalize_bwd(const float* y, const float* inv, const float* dy, float* dx, int rows, int cols) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float dot = 0.f;
  for (int c = lane; c < cols; c += 64) dot += y[(long long)r * cols + c] * dy[(long long)r * cols + c];
  dot = wave_sum(dot);
  float iv = inv[r];
  if (iv < 0.f) { iv = -iv; dot = 0.f; }
  for (int c = lane; c < cols; c += 64) dx[(long long)r * cols + c] = iv * (dy[(long long)r * cols + c] - y[(long long)r * cols + c] * dot);
}
extern "C" int sg_row_normalize_fwd(const float* x, float* y, float* inv, int rows, int cols, float eps, sg_stream_t s) {
  SG_CHECK(x && y && inv && rows > 0 && cols > 0, "sg_row_normalize_fwd: bad args");
  hipLaunchKernelGGL(k_row_normalize_fwd, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, x, y, inv, rows, cols, eps);
  SG_LAUNCH_CHECK();
  return 0;
}
extern "C" int sg_row_normalize_bwd(const float* y, const float* inv, const float* dy, float* dx, int rows, int cols, sg_stream_t s) {
  SG_CHECK(y && inv && dy && dx && rows > 0 && cols > 0, "sg_row_normalize_bwd: bad args");
  hipLaunchKernelGGL(k_row_normalize_bwd, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, y, inv, dy, dx, rows, cols);
  SG_LAUNCH_CHECK();
  return 0;
}

// p[r] = <a[r], b[r]>
__global__ __launch_bounds__(256) void k_row_dot(const float* a, const float* b, float* p, int rows, int cols) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int c = lane; c < cols; c += 64) acc += a[(long long)r * cols + c] * b[(long long)r * cols + c];
  acc = wave_sum(acc);
  if (lane == 0) p[r] = acc;
}
extern "C" int sg_row_dot(const float* a, const float* b, float* p, int rows, int cols, sg_stream_t s) {
  SG_CHECK(a && b && p && rows > 0 && cols > 0, "sg_row_dot: bad args");
  hipLaunchKernelGGL(k_row_dot, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, a, b, p, rows, cols);
  SG_LAUNCH_CHECK();
  return 0;
}
// out[r][c] = (acc ? out[r][c] : 0) + g[r] * x[r][c]
__global__ __launch_bounds__(256) void k_row_scale(const float* g, const float* x, float* out, long long n, int cols, int acc) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = g[i / cols] * x[i];
    out[i] = acc ? out[i] + v : v;
  }
}
extern "C" int sg_row_scale(const float* g, const float* x, float* out, int rows, int cols, int accumulate, sg_stream_t s) {
  SG_CHECK(g && x && out && rows > 0 && cols > 0, "sg_row_scale: bad args");
  const long long n = (long long)rows * cols;
  long long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_row_scale, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, g, x, out, n, cols, accumulate);
  SG_LAUNCH_CHECK();
  return 0;
}

// cross entropy, mean over rows (torch.nn.CrossEntropyLoss): row_loss[r] = logsumexp(z[r]) - z[r][label]; dz = (softmax - onehot) / rows
__global__ __launch_bounds__(256) void k_xent(const float* z, const int64_t* label, int rows, int cols, float* row_loss, float* dz) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* zr = z + (long long)r * cols;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, zr[c]);
  m = wave_max(m);
  float se = 0.f;
  for (int c = lane; c < cols; c += 64) se += expf(zr[c] - m);
  se = wave_sum(se);
  const long long tl = label[r];
  const bool bad = tl < 0 || tl >= cols;          // a label outside the head (e.g. an ADC label 2C-1 against a C-wide head): poison, never read out of bounds
  const int t = bad ? 0 : (int)tl;
  const float lse = m + logf(se);
  if (lane == 0) row_loss[r] = bad ? NAN : lse - zr[t];
  const float invr = 1.f / rows;
  for (int c = lane; c < cols; c += 64) dz[(long long)r * cols + c] = bad ? NAN : (expf(zr[c] - lse) - (c == t ? 1.f : 0.f)) * invr;
}
// Crammer-Singer multi-hinge (losses.py:242-252): row_loss = relu(1 + max_{c != label} z[c] - z[label]); dz of the mean
__global__ __launch_bounds__(256) void k_crammer_singer(const float* z, const int64_t* label, int rows, int cols, float* row_loss, float* dz) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* zr = z + (long long)r * cols;
  const long long tl = label[r];
  const bool bad = tl < 0 || tl >= cols;          // out-of-range label: NaN loss / gradient instead of an out-of-bounds read
  const int t = bad ? 0 : (int)tl;
  float m = -INFINITY; int arg = 0x7fffffff;
  for (int c = lane; c < cols; c += 64) { if (c != t) { const float v = zr[c]; if (v > m) { m = v; arg = c; } } }
  // wave argmax; among equal values the LOWEST index (what torch.max returns on the masked row)
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64); const int oa = __shfl_xor(arg, o, 64);
    if (om > m || (om == m && oa < arg)) { m = om; arg = oa; }
  }
  const float l = 1.f + m - zr[t];
  const bool on = l > 0.f;
  if (lane == 0) row_loss[r] = bad ? NAN : (on ? l : 0.f);
  const float invr = 1.f / rows;
  for (int c = lane; c < cols; c += 64) dz[(long long)r * cols + c] = bad ? NAN : (on ? ((c == arg ? invr : 0.f) - (c == t ? invr : 0.f)) : 0.f);
}
// fixed-order mean of the row losses (deterministic): loss[0] = sum_r row_loss[r] / rows
__global__ __launch_bounds__(256) void k_mean_rows(const float* row_loss, int rows, float* loss) {
  __shared__ float sm[4];
  float acc = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) acc += row_loss[r];
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) loss[0] = acc / rows;
}
extern "C" int sg_class_loss(int kind, const float* z, const int64_t* label, int rows, int cols, float* row_loss, float* loss, float* dz, sg_stream_t s) {
  SG_CHECK(z && label && row_loss && loss && dz && rows > 0 && cols > 1, "sg_class_loss: bad args");
  if (kind == 0) hipLaunchKernelGGL(k_xent, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, z, label, rows, cols, row_loss, dz);
  else if (kind == 1) hipLaunchKernelGGL(k_crammer_singer, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, z, label, rows, cols, row_loss, dz);
  else { sg_set_error("sg_class_loss: kind 0 (cross entropy) or 1 (Crammer-Singer)"); return -1; }
  hipLaunchKernelGGL(k_mean_rows, dim3(1), dim3(256), 0, (hipStream_t)s, (const float*)row_loss, rows, loss);
  SG_LAUNCH_CHECK();
  return 0;
}

// Contrastive conditioning losses over S = cos(embed_i, embed_j) [B][B] and p_i = cos(embed_i, proxy_i); T = temperature.
//  kind 0 (ConditionalContrastiveLoss, losses.py:50-97):  e_ij = exp(S_ij / T) (j != i), a_i = exp(p_i / T)
//        loss_i = -log( (a_i + sum_{j != i, y_j == y_i} e_ij) / (a_i + sum_{j != i} e_ij) )
//  kind 1 (Data2DataCrossEntropyLoss, losses.py:100-165): s_ij = (S_ij + m_p - 1) / T (j != i), mx_i = max_j s_ij (constant),
//        q_ij = [y_j != y_i] exp(relu(s_ij) - mx_i), pos_i = relu((m_p - p_i) / T), loss_i = pos_i + log(exp(-pos_i) + sum_j q_ij)
// Outputs: row_loss[i], dS[i][j] and dp[i] of the MEAN over rows. One wave per row.
__global__ __launch_bounds__(256) void k_contrastive_rows(int kind, const float* S, const float* p, const int64_t* label, int B, float T, float m_p,
                                                          float* row_loss, float* dS, float* dp) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  const int lane = threadIdx.x & 63;
  const float* Si = S + (long long)i * B;
  float* dSi = dS + (long long)i * B;
  const long long yi = label[i];
  const float invT = 1.f / T, invB = 1.f / B;
  if (kind == 0) {
    float all = 0.f, pos = 0.f;
    for (int j = lane; j < B; j += 64) {
      if (j == i) continue;
      const float e = expf(Si[j] * invT);
      all += e;
      if (label[j] == yi) pos += e;
    }
    all = wave_sum(all); pos = wave_sum(pos);
    const float a = expf(p[i] * invT);
    const float num = a + pos, den = a + all;
    if (lane == 0) { row_loss[i] = -logf(num / den); dp[i] = invB * a * invT * (1.f / den - 1.f / num); }
    for (int j = lane; j < B; j += 64) {
      float g = 0.f;
      if (j != i) {
        const float e = expf(Si[j] * invT);
        g = invB * e * invT * (1.f / den - (label[j] == yi ? 1.f / num : 0.f));
      }
      dSi[j] = g;
    }
  } else {
    float mx = -INFINITY;
    for (int j = lane; j < B; j += 64) if (j != i) mx = fmaxf(mx, (Si[j] + m_p - 1.f) * invT);
    mx = wave_max(mx);
    float q = 0.f;
    for (int j = lane; j < B; j += 64) {
      if (j == i || label[j] == yi) continue;
      const float sij = (Si[j] + m_p - 1.f) * invT;
      q += expf(fmaxf(sij, 0.f) - mx);
    }
    q = wave_sum(q);
    const float u = (m_p - p[i]) * invT;
    const float pos = fmaxf(u, 0.f);
    const float ep = expf(-pos);
    const float den = ep + q;
    if (lane == 0) {
      row_loss[i] = pos + logf(den);
      // d loss / d pos = 1 - ep / den; d pos / d p = -1/T where u > 0
      dp[i] = (u > 0.f) ? invB * (1.f - ep / den) * (-invT) : 0.f;
    }
    for (int j = lane; j < B; j += 64) {
      float g = 0.f;
      if (j != i && label[j] != yi) {
        const float sij = (Si[j] + m_p - 1.f) * invT;
        if (sij > 0.f) g = invB * expf(sij - mx) / den * invT;      // relu: no gradient where s_ij <= 0 (the term is then the constant exp(-mx))
      }
      dSi[j] = g;
    }
  }
}
extern "C" int sg_contrastive_loss(int kind, const float* S, const float* p, const int64_t* label, int B, float temperature, float m_p,
                                   float* row_loss, float* loss, float* dS, float* dp, sg_stream_t s) {
  SG_CHECK(S && p && label && row_loss && loss && dS && dp && B > 1 && temperature > 0.f, "sg_contrastive_loss: bad args");
  SG_CHECK(kind == 0 || kind == 1, "sg_contrastive_loss: kind 0 (2C) or 1 (D2D-CE)");
  hipLaunchKernelGGL(k_contrastive_rows, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)s, kind, S, p, label, B, temperature, m_p, row_loss, dS, dp);
  hipLaunchKernelGGL(k_mean_rows, dim3(1), dim3(256), 0, (hipStream_t)s, (const float*)row_loss, B, loss);
  SG_LAUNCH_CHECK();
  return 0;
}

// out[r] = z[r][label[r]]  /  dz[r][c] = (c == label[r]) * g[r]        (multi-discriminator head: adv_output[idx, label])
__global__ __launch_bounds__(256) void k_gather_cols(const float* z, const int64_t* label, int rows, int cols, float* out) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < rows) { const long long t = label[r]; out[r] = (t < 0 || t >= cols) ? NAN : z[(long long)r * cols + t]; }
}
__global__ __launch_bounds__(256) void k_scatter_cols(const float* g, const int64_t* label, int rows, int cols, float* dz) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < (long long)rows * cols; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    dz[i] = (c == (int)label[r]) ? g[r] : 0.f;
  }
}
extern "C" int sg_gather_cols(const float* z, const int64_t* label, int rows, int cols, float* out, sg_stream_t s) {
  SG_CHECK(z && label && out && rows > 0 && cols > 0, "sg_gather_cols: bad args");
  hipLaunchKernelGGL(k_gather_cols, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)s, z, label, rows, cols, out);
  SG_LAUNCH_CHECK();
  return 0;
}
extern "C" int sg_scatter_cols(const float* g, const int64_t* label, int rows, int cols, float* dz, sg_stream_t s) {
  SG_CHECK(g && label && dz && rows > 0 && cols > 0, "sg_scatter_cols: bad args");
  long long blocks = ((long long)rows * cols + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_scatter_cols, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, g, label, rows, cols, dz);
  SG_LAUNCH_CHECK();
  return 0;
}
