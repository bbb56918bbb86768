// sn.hip -- spectral normalisation for every SN layer of a network in a handful of batched launches.
// Follows torch.nn.utils.spectral_norm (installed torch/nn/utils/spectral_norm.py: compute_weight), which is what
// reference src/utils/ops.py:195-224 wraps around Conv2d / Linear / Embedding with eps=1e-6:
//     v <- normalize(W^T u, eps) ; u <- normalize(W v, eps)        (only while module.training)
//     sigma = u . (W v) ; W_sn = W / sigma
// The same pass casts W_sn to the compute dtype and writes the two operand images the convolution engine wants
// ([Cout][R][S][Cin] for forward, [Cin][R-1-r][S-1-s][Cout] for the data gradient), so the normalised weight is
// never re-laid-out again. Work is weight-sized and HBM-bound (3 reads of W per forward).
// All reductions use a fixed order (no atomics): replicas on different GPUs stay bit-identical, which is what
// lets the data-parallel path skip the reference's per-forward buffer broadcast (DDP broadcast_buffers).
#include "common.h"
#include "../../include/sgamd.h"

#define SN_SPLITS 16      // row groups of the W^T u pass (bank.py WeightBank.SN_SPLITS sizes the workspace with the same number)

// element (o, k) of the weight viewed as the [rows][cols] matrix spectral norm works on. Conv2d / Linear / Embedding:
// natural layout (dim 0 = rows). ConvTranspose2d (torch uses dim=1): weight is [Cin][Cout][R][S], rows = Cout.
template <class LAYER> __device__ __forceinline__ long long sn_widx(const LAYER& l, int o, int k) {
  if (!l.trans) return (long long)o * l.cols + k;
  const int c = k / l.RS, rs = k - c * l.RS;
  return ((long long)c * l.rows + o) * l.RS + rs;
}

// Flat work tables (round 6). The launches below used to be (tiles of the LARGEST layer) x layers: a table that holds [1536 x 13824] convolutions next to [96 x 864] ones, a
// [24576 x 20] linear layer and a [1000 x 1536] embedding is mostly empty workgroups (and needed two calls to keep the linear layers from inflating the convolutions' grids).
// Here every launch is a 1-D grid of exactly the tiles that exist: the host sums the per-layer tile counts into `start` (kernel argument by value, scalar loads), a workgroup
// finds its layer by bisection.
#define SN_MAXL 64
struct sn_flat { int n; int start[SN_MAXL + 1]; };
__device__ __forceinline__ int sn_find(const sn_flat& F, int b) {     // the layer li with start[li] <= b < start[li + 1] (empty layers have start[li] == start[li + 1])
  int lo = 0, hi = F.n;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (b >= F.start[mid]) lo = mid; else hi = mid; }
  return lo;
}
__host__ __device__ __forceinline__ bool sn_wtu_skinny(const sg_sn_layer& l) { return !l.trans && (l.cols & 3) == 0 && l.cols <= 64 && ((uintptr_t)l.w & 15) == 0; }

// flat grid of (col tiles of 1024) x SN_SPLITS per power-iterating layer: partial[split][k] = sum over the split's rows of W[o][k] u[o].
// A thread owns 4 adjacent columns (one 16-byte load per row, 4 KB contiguous per row and block) and keeps 8 rows in flight (16 measured SLOWER, 129 -> 153 us per launch, session r3n): the
// one-float-per-thread column walk this replaces touched 1 KB per 55 KB row and ran at 0.35-0.95 TB/s (r01 kernel trace).
__global__ __launch_bounds__(256) void k_sn_wtu(const sg_sn_layer* L, float* work, const sn_flat F) {
  const int li = sn_find(F, blockIdx.x);
  const sg_sn_layer l = L[li];
  if (!l.apply_sn || !l.do_power_iter) return;
  const int tiles_x = (l.cols + 1023) >> 10, local = blockIdx.x - F.start[li];
  const int by = local / tiles_x, bx = local - by * tiles_x;
  const int per = (l.rows + SN_SPLITS - 1) / SN_SPLITS;
  int o0 = by * per, o1 = o0 + per; if (o1 > l.rows) o1 = l.rows;
  float* out = work + l.work_off + (long long)by * l.cols;
  if (sn_wtu_skinny(l)) {
    // tall and skinny (BigGAN's linear0: 24576 x 20): the column-owner mapping below leaves 5 threads walking 1536 rows each -- 119 us for 2 MB (round 6 trace of
    // tools/sn_bench.py). Here R = 256 / (cols / 4) threads share a column vector, each takes every R-th row of the split, and the R partial sums meet in LDS in
    // lane order (fixed order: deterministic).
    __shared__ f32x4 red[256];
    const int c4 = l.cols >> 2, R = 256 / c4;
    const int r = threadIdx.x / c4, k4 = threadIdx.x - r * c4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (r < R) {
      const float* wp = l.w + 4 * k4;
      int o = o0 + r;
      for (; o + 3 * R < o1; o += 4 * R) {
        const f32x4 w0 = *(const f32x4*)(wp + (long long)o * l.cols), w1 = *(const f32x4*)(wp + (long long)(o + R) * l.cols);
        const f32x4 w2 = *(const f32x4*)(wp + (long long)(o + 2 * R) * l.cols), w3 = *(const f32x4*)(wp + (long long)(o + 3 * R) * l.cols);
        const float u0 = l.u[o], u1 = l.u[o + R], u2 = l.u[o + 2 * R], u3 = l.u[o + 3 * R];
        acc += w0 * u0; acc += w1 * u1; acc += w2 * u2; acc += w3 * u3;
      }
      for (; o < o1; o += R) acc += *(const f32x4*)(wp + (long long)o * l.cols) * l.u[o];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (r == 0 && k4 < c4) {
      f32x4 t = red[k4];
      for (int q = 1; q < R; q++) t += red[q * c4 + k4];
      out[4 * k4] = t[0]; out[4 * k4 + 1] = t[1]; out[4 * k4 + 2] = t[2]; out[4 * k4 + 3] = t[3];
    }
    return;
  }
  if (!l.trans && (l.cols & 3) == 0 && ((uintptr_t)l.w & 15) == 0) {
    const int k = (bx * 256 + threadIdx.x) * 4;
    if (k >= l.cols) return;
    const float* wp = l.w + k;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int o = o0;
    for (; o + 8 <= o1; o += 8) {
      f32x4 r[8]; float uu[8];
#pragma unroll
      for (int e = 0; e < 8; e++) { r[e] = *(const f32x4*)(wp + (long long)(o + e) * l.cols); uu[e] = l.u[o + e]; }
#pragma unroll
      for (int e = 0; e < 8; e++) { acc[0] += r[e][0] * uu[e]; acc[1] += r[e][1] * uu[e]; acc[2] += r[e][2] * uu[e]; acc[3] += r[e][3] * uu[e]; }
    }
    for (; o < o1; o++) {
      const f32x4 r = *(const f32x4*)(wp + (long long)o * l.cols); const float uu = l.u[o];
      acc[0] += r[0] * uu; acc[1] += r[1] * uu; acc[2] += r[2] * uu; acc[3] += r[3] * uu;
    }
    out[k] = acc[0]; out[k + 1] = acc[1]; out[k + 2] = acc[2]; out[k + 3] = acc[3];   // (work offsets are only 4-byte aligned)
    return;
  }
  for (int k = bx * 1024 + threadIdx.x; k < l.cols && k < (bx + 1) * 1024; k += 256) {
    float acc = 0.f;
    for (int o = o0; o < o1; o++) acc += l.w[sn_widx(l, o, k)] * l.u[o];
    out[k] = acc;
  }
}
// One workgroup per layer finishes a power-iteration vector (k_sn_v, k_sn_u): these are latency chains (a 13824-column layer was 54 dependent trips of 16 loads at 256 threads,
// 27 us per launch with the matrix-sized passes waiting behind it -- r7b trace), so the workgroup is as wide as the hardware allows. Fixed summation order.
#define SN_NT 1024
__device__ __forceinline__ float block_sum_nt(float v, float* sm /* SN_NT / 64 floats */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = sm[0];
#pragma unroll
  for (int w = 1; w < SN_NT / 64; w++) t += sm[w];
  return t;
}
// grid (layers): v = normalize(sum of partials)
__global__ __launch_bounds__(SN_NT) void k_sn_v(const sg_sn_layer* L, float* work, float eps) {
  __shared__ float sm[SN_NT / 64];
  const sg_sn_layer l = L[blockIdx.x];
  if (!l.apply_sn || !l.do_power_iter) return;
  float* part = work + l.work_off;
  float nn = 0.f;
  for (int k = threadIdx.x; k < l.cols; k += SN_NT) {
    float t = 0.f;
#pragma unroll
    for (int s = 0; s < SN_SPLITS; s++) t += part[(long long)s * l.cols + k];
    part[k] = t;  // same thread re-reads it below
    nn += t * t;
  }
  nn = block_sum_nt(nn, sm);
  const float inv = 1.f / fmaxf(sqrtf(nn), eps);
  for (int k0 = threadIdx.x; k0 < l.cols; k0 += SN_NT * 8) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { const int k = k0 + SN_NT * e; t[e] = k < l.cols ? part[k] : 0.f; }
#pragma unroll
    for (int e = 0; e < 8; e++) { const int k = k0 + SN_NT * e; if (k < l.cols) l.v[k] = t[e] * inv; }
  }
}
// flat grid of (row tiles of 4) per SN layer: one wave per row, t_u[o] = W[o,:] . v
__global__ __launch_bounds__(256) void k_sn_wv(const sg_sn_layer* L, float* work, const sn_flat F) {
  const int li = sn_find(F, blockIdx.x);
  const sg_sn_layer l = L[li];
  if (!l.apply_sn) return;
  const int o = (blockIdx.x - F.start[li]) * 4 + (threadIdx.x >> 6);
  if (o >= l.rows) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  if (!l.trans && (l.cols & 3) == 0 && ((uintptr_t)l.w & 15) == 0 && ((uintptr_t)l.v & 15) == 0) {
    // 16 bytes per lane, four loads of the row in flight (round 3: the 4-byte walk moved the 352 MB of D's weights at 2.7 TB/s)
    const float* wr = l.w + (long long)o * l.cols;
    f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
    int k = lane * 4;
    for (; k + 768 < l.cols; k += 1024) {
      f32x4 w0 = *(const f32x4*)(wr + k), w1 = *(const f32x4*)(wr + k + 256), w2 = *(const f32x4*)(wr + k + 512), w3 = *(const f32x4*)(wr + k + 768);
      f32x4 v0 = *(const f32x4*)(l.v + k), v1 = *(const f32x4*)(l.v + k + 256), v2 = *(const f32x4*)(l.v + k + 512), v3 = *(const f32x4*)(l.v + k + 768);
      a4 += w0 * v0; a4 += w1 * v1; a4 += w2 * v2; a4 += w3 * v3;
    }
    for (; k < l.cols; k += 256) a4 += *(const f32x4*)(wr + k) * *(const f32x4*)(l.v + k);
    acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  } else {
    for (int k = lane; k < l.cols; k += 64) acc += l.w[sn_widx(l, o, k)] * l.v[k];
  }
  acc = wave_sum(acc);
  if (lane == 0) work[l.work_off + (long long)SN_SPLITS * l.cols + o] = acc;
}
// grid (layers): u, sigma
__global__ __launch_bounds__(SN_NT) void k_sn_u(const sg_sn_layer* L, float* work, float eps) {
  __shared__ float sm[SN_NT / 64];
  const sg_sn_layer l = L[blockIdx.x];
  if (!l.apply_sn) { if (threadIdx.x == 0) l.sigma[0] = 1.f; return; }
  const float* tu = work + l.work_off + (long long)SN_SPLITS * l.cols;
  float sig;
  if (l.do_power_iter) {
    // (eight loads in flight per thread: a 24576-row linear layer is 96 trips of this one block; one load per trip was most of the kernel's 67 us on the
    // generator's linear table -- round 6 trace of tools/sn_bench.py. Same per-thread summation order.)
    float nn = 0.f;
    for (int o0 = threadIdx.x; o0 < l.rows; o0 += SN_NT * 8) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; e++) { const int o = o0 + SN_NT * e; t[e] = o < l.rows ? tu[o] : 0.f; }
#pragma unroll
      for (int e = 0; e < 8; e++) nn += t[e] * t[e];
    }
    nn = block_sum_nt(nn, sm);
    const float inv = 1.f / fmaxf(sqrtf(nn), eps);
    float dot = 0.f;
    for (int o0 = threadIdx.x; o0 < l.rows; o0 += SN_NT * 8) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; e++) { const int o = o0 + SN_NT * e; t[e] = o < l.rows ? tu[o] : 0.f; }
#pragma unroll
      for (int e = 0; e < 8; e++) { const int o = o0 + SN_NT * e; if (o < l.rows) { const float un = t[e] * inv; l.u[o] = un; dot += un * t[e]; } }
    }
    sig = block_sum_nt(dot, sm);
  } else {
    float dot = 0.f;
    for (int o = threadIdx.x; o < l.rows; o += SN_NT) dot += l.u[o] * tu[o];
    sig = block_sum_nt(dot, sm);
  }
  if (threadIdx.x == 0) l.sigma[0] = sig;
  if (l.u_snap) {
    for (int o0 = threadIdx.x; o0 < l.rows; o0 += SN_NT * 8) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; e++) { const int o = o0 + SN_NT * e; t[e] = o < l.rows ? l.u[o] : 0.f; }
#pragma unroll
      for (int e = 0; e < 8; e++) { const int o = o0 + SN_NT * e; if (o < l.rows) l.u_snap[o] = t[e]; }
    }
  }
  if (l.v_snap) {
    // eight loads in flight per thread: one load -> store per iteration (the two vectors may alias as far as the compiler knows) was 54
    // dependent round trips on a 13824-column layer, most of this kernel's 48 us
    for (int k0 = threadIdx.x; k0 < l.cols; k0 += SN_NT * 8) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; e++) { const int k = k0 + SN_NT * e; t[e] = k < l.cols ? l.v[k] : 0.f; }
#pragma unroll
      for (int e = 0; e < 8; e++) { const int k = k0 + SN_NT * e; if (k < l.cols) l.v_snap[k] = t[e]; }
    }
  }
}
// grid (element tiles, layers): W / sigma -> operand images
template <typename T> __global__ __launch_bounds__(256) void k_sn_pack(const sg_sn_layer* L) {
  const sg_sn_layer l = L[blockIdx.y];
  const int rows_out = l.rows_pad > l.rows ? l.rows_pad : l.rows;
  const long long total = (long long)rows_out * l.cols;
  const float sig = l.sigma[0];
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int o = (int)(i / l.cols), k = (int)(i % l.cols);
    const int c = k / l.RS, rs = k - c * l.RS;
    float val = 0.f;
    if (o < l.rows) {
      val = l.w[sn_widx(l, o, k)] / sig;
      if (l.w_f32) l.w_f32[i] = val;
      if (l.w_dgrad) ((T*)l.w_dgrad)[((long long)c * l.RS + (l.dgrad_noflip ? rs : (l.RS - 1 - rs))) * rows_out + o] = from_f<T>(val);
    }
    if (l.w_fwd) ((T*)l.w_fwd)[((long long)o * l.RS + rs) * (l.Cin_pad > l.Cin ? l.Cin_pad : l.Cin) + c] = from_f<T>(val);
  }
}

// Tiled replacement of k_sn_pack (its scattered 2-byte stores ran at ~1/15 of HBM speed):
//  * k_sn_pack_rows: one block per weight row -> fp32 natural copy and the forward image [o][rs][c]; the [c][rs] -> [rs][c]
//    shuffle of the row goes through LDS so both the read and the write are contiguous.
//  * k_sn_pack_dgrad: 64(o) x 128(k) tiles transposed through LDS -> data-gradient image [c][rs'][o], 128-byte rows of o.
template <typename T> __global__ __launch_bounds__(256) void k_sn_pack_rows(const sg_sn_layer* L, const sn_flat F) {
  extern __shared__ __attribute__((aligned(16))) char sn_raw[];
  T* row = (T*)sn_raw;
  const int li = sn_find(F, blockIdx.x);
  const sg_sn_layer l = L[li];
  const int nblk = F.start[li + 1] - F.start[li];          // workgroups of this layer (<= 2048: rows beyond that are walked)
  const int rows_out = l.rows_pad > l.rows ? l.rows_pad : l.rows;
  const float sig = l.sigma[0];
  const float inv = 1.f / sig;
  const int cp = l.Cin_pad > l.Cin ? l.Cin_pad : l.Cin, colsp = l.RS * cp;
  // fast path (round 2): 16-byte global accesses on both sides and no per-element division -- the read side takes 4 fp32 per lane,
  // the write side 8 outputs of one tap (cp % 8 == 0: eight consecutive image elements share their tap), gathered from the LDS row
  const bool vec = !l.trans && (l.cols % 4 == 0) && (cp % 8 == 0) && sizeof(T) == 2 && ((reinterpret_cast<uintptr_t>(l.w) & 15) == 0) &&
                   (!l.w_f32 || (reinterpret_cast<uintptr_t>(l.w_f32) & 15) == 0) && (!l.w_fwd || (reinterpret_cast<uintptr_t>(l.w_fwd) & 15) == 0);
  for (int o = blockIdx.x - F.start[li]; o < rows_out; o += nblk) {
    if (vec) {
      const float* src = l.w + (long long)o * l.cols;
      for (int k = threadIdx.x * 4; k < l.cols; k += 1024) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (o < l.rows) {
          v = *(const f32x4*)(src + k);
          // same operation as the scalar path (a division, not a multiplication by the reciprocal: bit-identical images)
          v[0] = v[0] / sig; v[1] = v[1] / sig; v[2] = v[2] / sig; v[3] = v[3] / sig;
          if (l.w_f32) *(f32x4*)(l.w_f32 + (long long)o * l.cols + k) = v;
        }
        u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *(u32x2*)((bf16_t*)row + k) = pk;
      }
    } else {
      for (int k = threadIdx.x; k < l.cols; k += 256) {
        float val = 0.f;
        if (o < l.rows) {
          val = l.w[sn_widx(l, o, k)] / sig;
          if (l.w_f32) l.w_f32[(long long)o * l.cols + k] = val;
        }
        row[k] = from_f<T>(val);
      }
    }
    __syncthreads();
    if (l.w_fwd) {
      T* dst = (T*)l.w_fwd + (long long)o * colsp;
      if (vec) {
        const bf16_t* r16 = (const bf16_t*)row;
        int j = threadIdx.x * 8;
        int rs = j / cp, c = j - rs * cp;
        const int drs = 2048 / cp, dc = 2048 - drs * cp;
        for (; j < colsp; j += 2048) {
          uint32_t w4[4];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int c0 = c + 2 * i;
            const uint32_t lo = c0 < l.Cin ? r16[c0 * l.RS + rs] : 0u, hi = c0 + 1 < l.Cin ? r16[(c0 + 1) * l.RS + rs] : 0u;
            w4[i] = lo | (hi << 16);
          }
          u32x4 v = {w4[0], w4[1], w4[2], w4[3]};
          *(u32x4*)((bf16_t*)dst + j) = v;
          c += dc; rs += drs;
          if (c >= cp) { c -= cp; rs++; }
        }
      } else {
        for (int j = threadIdx.x; j < colsp; j += 256) {
          const int rs = j / cp, c = j - rs * cp;
          dst[j] = c < l.Cin ? row[c * l.RS + rs] : from_f<T>(0.f);
        }
      }
    }
    __syncthreads();
  }
  (void)inv;
}
// The data-gradient image of a layer whose forward image [o][rs][c] was just written by k_sn_pack_rows is a transposition of THAT image (the same bf16 values, 2 bytes
// read per element instead of the master's 4, no division): `elem` = bytes per image element.
__host__ __device__ __forceinline__ bool sn_dgrad_from_image(const sg_sn_layer& l, int elem) {
  const int rows_out = l.rows_pad > l.rows ? l.rows_pad : l.rows, cp = l.Cin_pad > l.Cin ? l.Cin_pad : l.Cin;
  return elem == 2 && l.w_fwd && !l.trans && (rows_out % 8 == 0) && (cp % 8 == 0) && ((reinterpret_cast<uintptr_t>(l.w_fwd) & 15) == 0) &&
         ((reinterpret_cast<uintptr_t>(l.w_dgrad) & 15) == 0);
}
__host__ __device__ __forceinline__ int sn_dgrad_extent(const sg_sn_layer& l, int elem) {      // columns the 128-wide tiles of a layer walk
  return sn_dgrad_from_image(l, elem) ? l.RS * (l.Cin_pad > l.Cin ? l.Cin_pad : l.Cin) : l.cols;
}
template <typename T> __global__ __launch_bounds__(256) void k_sn_pack_dgrad(const sg_sn_layer* L, const sn_flat F) {
  __shared__ T tile[64][130];
  const int li = sn_find(F, blockIdx.x);
  const sg_sn_layer l = L[li];
  if (!l.w_dgrad) return;
  const bool img = sn_dgrad_from_image(l, (int)sizeof(T));
  const int tiles_k = (sn_dgrad_extent(l, (int)sizeof(T)) + 127) >> 7, local = blockIdx.x - F.start[li];
  const int by = local / tiles_k, bx = local - by * tiles_k;
  const int o0 = by * 64, k0 = bx * 128;
  const int rows_out = l.rows_pad > l.rows ? l.rows_pad : l.rows;    // pitch of the image; columns >= rows stay zero
  if (img) {      // (tiles over rows_out: the zero rows of the forward image's padding become the zero columns of this one)
    const int cp = l.Cin_pad > l.Cin ? l.Cin_pad : l.Cin, colsp = l.RS * cp;
    for (int e = threadIdx.x; e < 64 * 16; e += 256) {
      const int oo = e >> 4, jj = (e & 15) * 8, o = o0 + oo, j = k0 + jj;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (o < rows_out && j < colsp) v = *(const u32x4*)((const bf16_t*)l.w_fwd + (long long)o * colsp + j);
      uint32_t* tp = (uint32_t*)&tile[oo][jj];
      tp[0] = v[0]; tp[1] = v[1]; tp[2] = v[2]; tp[3] = v[3];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 8 * 128; e += 256) {
      const int og = e & 7, jj = e >> 3, o = o0 + og * 8, j = k0 + jj;
      if (j < colsp && o < rows_out) {
        const int rs = j / cp, c = j - rs * cp;
        if (c < l.Cin) {
          const bf16_t* tp = (const bf16_t*)&tile[0][0];
          uint32_t w4[4];
#pragma unroll
          for (int i = 0; i < 4; i++) w4[i] = (uint32_t)tp[(og * 8 + 2 * i) * 130 + jj] | ((uint32_t)tp[(og * 8 + 2 * i + 1) * 130 + jj] << 16);
          u32x4 v = {w4[0], w4[1], w4[2], w4[3]};
          *(u32x4*)((bf16_t*)l.w_dgrad + ((long long)c * l.RS + (l.dgrad_noflip ? rs : (l.RS - 1 - rs))) * rows_out + o) = v;
        }
      }
    }
    return;
  }
  if (o0 >= l.rows || k0 >= l.cols) return;
  const float sig = l.sigma[0];
  // fast path (round 2): 16-byte accesses on both sides -- 4 fp32 per lane in, 8 couts of one (c, tap) row per lane out
  const bool vec = sizeof(T) == 2 && !l.trans && (l.cols % 4 == 0) && (rows_out % 8 == 0) && ((reinterpret_cast<uintptr_t>(l.w) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(l.w_dgrad) & 15) == 0);
  if (vec) {
    for (int e = threadIdx.x; e < 64 * 32; e += 256) {
      const int oo = e >> 5, kk = (e & 31) * 4, o = o0 + oo, k = k0 + kk;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (o < l.rows && k < l.cols) {
        v = *(const f32x4*)(l.w + (long long)o * l.cols + k);
        v[0] = v[0] / sig; v[1] = v[1] / sig; v[2] = v[2] / sig; v[3] = v[3] / sig;     // the scalar path's operation: bit-identical images
      }
      bf16_t* tp = (bf16_t*)&tile[oo][kk];
      *(uint32_t*)tp = pack2bf(v[0], v[1]);
      *(uint32_t*)(tp + 2) = pack2bf(v[2], v[3]);
    }
  } else {
    const int kk = threadIdx.x & 127, k = k0 + kk;
    for (int p = 0; p < 32; p++) {
      const int oo = p * 2 + (threadIdx.x >> 7), o = o0 + oo;
      float val = 0.f;
      if (o < l.rows && k < l.cols) val = l.w[sn_widx(l, o, k)] / sig;
      tile[oo][kk] = from_f<T>(val);
    }
  }
  __syncthreads();
  if (vec) {
    for (int e = threadIdx.x; e < 8 * 128; e += 256) {
      const int og = e & 7, kk = e >> 3, o = o0 + og * 8, k = k0 + kk;
      if (k < l.cols && o < rows_out) {
        const int c = k / l.RS, rs = k - c * l.RS;
        const bf16_t* tp = (const bf16_t*)&tile[0][0];
        uint32_t w4[4];
#pragma unroll
        for (int i = 0; i < 4; i++) w4[i] = (uint32_t)tp[(og * 8 + 2 * i) * 130 + kk] | ((uint32_t)tp[(og * 8 + 2 * i + 1) * 130 + kk] << 16);
        u32x4 v = {w4[0], w4[1], w4[2], w4[3]};
        *(u32x4*)((bf16_t*)l.w_dgrad + ((long long)c * l.RS + (l.dgrad_noflip ? rs : (l.RS - 1 - rs))) * rows_out + o) = v;
      }
    }
  } else {
    const int oo = threadIdx.x & 63, o = o0 + oo;
    for (int p = 0; p < 32; p++) {
      const int kk = p * 4 + (threadIdx.x >> 6), k = k0 + kk;
      if (k < l.cols && o < l.rows) {
        const int c = k / l.RS, rs = k - c * l.RS;
        ((T*)l.w_dgrad)[((long long)c * l.RS + (l.dgrad_noflip ? rs : (l.RS - 1 - rs))) * rows_out + o] = tile[oo][kk];
      }
    }
  }
}
template <typename T> static void sn_pack_launch(const sg_sn_layer* layers_dev, const sg_sn_layer* layers_host, int n, int max_rows_out, int max_rows, int max_cols, long long max_elems, hipStream_t st) {
  if ((size_t)max_cols * sizeof(T) <= 60 * 1024) {
    sn_flat Fr, Fd;
    Fr.n = Fd.n = n; Fr.start[0] = Fd.start[0] = 0;
    for (int i = 0; i < n; i++) {
      const sg_sn_layer& l = layers_host[i];
      const int ro = l.rows_pad > l.rows ? l.rows_pad : l.rows;
      Fr.start[i + 1] = Fr.start[i] + (ro > 2048 ? 2048 : ro);
      const bool img = sn_dgrad_from_image(l, (int)sizeof(T));
      Fd.start[i + 1] = Fd.start[i] + (l.w_dgrad ? ((sn_dgrad_extent(l, (int)sizeof(T)) + 127) / 128) * (((img ? ro : l.rows) + 63) / 64) : 0);
    }
    hipLaunchKernelGGL(k_sn_pack_rows<T>, dim3(Fr.start[n]), dim3(256), (size_t)max_cols * sizeof(T), st, layers_dev, Fr);
    if (Fd.start[n] > 0) hipLaunchKernelGGL(k_sn_pack_dgrad<T>, dim3(Fd.start[n]), dim3(256), 0, st, layers_dev, Fd);
  } else {
    long long tiles = (max_elems + 256 * 8 - 1) / (256 * 8); if (tiles > 4096) tiles = 4096;
    hipLaunchKernelGGL(k_sn_pack<T>, dim3((int)tiles, n), dim3(256), 0, st, layers_dev);
  }
}

// layers [i0, i1) of a table: the launch sequence of one forward (power iteration, sigma, operand images)
template <typename T> static void sn_forward_range(const sg_sn_layer* layers_dev, const sg_sn_layer* layers_host, int n, float eps, float* work, hipStream_t st) {
  int max_rows = 1, max_cols = 1, max_rows_out = 1; long long max_elems = 1; bool any_sn = false, any_pi = false;
  for (int i = 0; i < n; i++) {
    const sg_sn_layer& l = layers_host[i];
    if (l.apply_sn) { any_sn = true; if (l.do_power_iter) any_pi = true; }
    if (l.rows > max_rows) max_rows = l.rows;
    if (l.cols > max_cols) max_cols = l.cols;
    const int ro = l.rows_pad > l.rows ? l.rows_pad : l.rows;
    if (ro > max_rows_out) max_rows_out = ro;
    const long long e = (long long)ro * l.cols;
    if (e > max_elems) max_elems = e;
  }
  sn_flat Fu, Fv;
  Fu.n = Fv.n = n; Fu.start[0] = Fv.start[0] = 0;
  for (int i = 0; i < n; i++) {
    const sg_sn_layer& l = layers_host[i];
    Fu.start[i + 1] = Fu.start[i] + (l.apply_sn && l.do_power_iter ? ((l.cols + 1023) / 1024) * SN_SPLITS : 0);
    Fv.start[i + 1] = Fv.start[i] + (l.apply_sn ? (l.rows + 3) / 4 : 0);
  }
  if (any_pi) {
    hipLaunchKernelGGL(k_sn_wtu, dim3(Fu.start[n]), dim3(256), 0, st, layers_dev, work, Fu);
    hipLaunchKernelGGL(k_sn_v, dim3(n), dim3(SN_NT), 0, st, layers_dev, work, eps);
  }
  if (any_sn) hipLaunchKernelGGL(k_sn_wv, dim3(Fv.start[n]), dim3(256), 0, st, layers_dev, work, Fv);
  hipLaunchKernelGGL(k_sn_u, dim3(n), dim3(SN_NT), 0, st, layers_dev, work, eps);
  sn_pack_launch<T>(layers_dev, layers_host, n, max_rows_out, max_rows, max_cols, max_elems, st);
}
// (Round 5, measured and removed: walking the table in runs of layers that fit the 256 MB Infinity Cache, each run through its whole launch sequence, so that the
// later passes of a run would find its weights on the die -- SLOWER on the D table, forward 430 -> 512 us and backward 588 -> 1177 us: the extra launches cost
// more than the cache returns. What does pay is keeping layers of very different shapes out of one launch: the grids are sized by the largest layer of the
// table, and one [24576 x 20] linear layer next to [1536 x 13824] convolutions turned k_sn_pack_dgrad into 1.7 M workgroups, all but 50 k of them empty
// (generator forward 869 us): bank.py hands the convolutions and the linear / embedding layers over as separate tables.)
extern "C" int sg_sn_forward(int dtype, const sg_sn_layer* layers_dev, const sg_sn_layer* layers_host, int n, float eps, float* work, long long work_floats, sg_stream_t s) {
  SG_CHECK(layers_dev && layers_host && n > 0 && work, "sg_sn_forward: bad args");
  SG_CHECK(dtype == SG_DTYPE_F32 || dtype == SG_DTYPE_BF16, "sg_sn_forward: bad dtype");
  for (int i = 0; i < n; i++) {
    const sg_sn_layer& l = layers_host[i];
    SG_CHECK(l.w && l.sigma && l.rows > 0 && l.cols > 0 && l.RS > 0 && l.Cin * l.RS == l.cols, "sg_sn_forward: bad layer");
    if (l.apply_sn) {
      SG_CHECK(l.u && l.v, "sg_sn_forward: SN layer without u/v");
      SG_CHECK(l.work_off >= 0 && l.work_off + (long long)SN_SPLITS * l.cols + l.rows <= work_floats, "sg_sn_forward: workspace too small");
    }
  }
  hipStream_t st = (hipStream_t)s;
  double bytes = 0.0;      // algorithmic: W^T u and W v read the fp32 weight once each, the pack reads it again and writes the operand images
  for (int i = 0; i < n; i++) {
    const sg_sn_layer& l = layers_host[i];
    const double e = (double)l.rows * l.cols, es = dtype == SG_DTYPE_BF16 ? 2.0 : 4.0;
    bytes += e * (4.0 * ((l.apply_sn && l.do_power_iter ? 1 : 0) + (l.apply_sn ? 1 : 0) + 1) + (l.w_fwd ? es : 0.0) + (l.w_dgrad ? es : 0.0) + (l.w_f32 ? 4.0 : 0.0));
  }
  SgProfScope prof(st, bytes, 3);
  for (int i0 = 0; i0 < n; i0 += SN_MAXL) {        // (a flat table holds SN_MAXL layers)
    const int cnt = n - i0 < SN_MAXL ? n - i0 : SN_MAXL;
    if (dtype == SG_DTYPE_F32) sn_forward_range<float>(layers_dev + i0, layers_host + i0, cnt, eps, work, st);
    else sn_forward_range<bf16_t>(layers_dev + i0, layers_host + i0, cnt, eps, work, st);
  }
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- backward: dW = (dWt - <dWt, W/sigma> u v^T) / sigma --------------------------------------------------
#define SNB_BLOCKS 512     // block partials of <dWt, W> per layer (64 blocks left 3/4 of the chip idle on the big layers: 0.97 TB/s; round 5, tools/sn_bench.py: 1024 / 2048 / 4096
                           // are SLOWER -- D backward 576 / 585 / 728 / 1145 us -- every block re-sums the partials and the per-item staging gets shorter)
__device__ __forceinline__ long long snb_src_index(const sg_sn_bwd_layer& l, int o, int k) {
  if (l.natural == 1) return (long long)o * l.cols + k;
  const int c = k / l.RS, rs = k - c * l.RS;
  if (l.natural == 2) return ((long long)c * l.RS + rs) * l.rows + o;   // [Cin][R][S][Cout]: weight gradient of a transposed conv
  return ((long long)o * l.RS + rs) * (l.Cin_pad > l.Cin ? l.Cin_pad : l.Cin) + c;
}
// ---- tile kernels (round 2) ---------------------------------------------------------------------------------------------------
// k_snb_dot / k_snb_apply walk a flat element index with a 64-bit division per element and fetch the two layouts of a weight
// ([o][rs][c] image of the gradient, [o][c][rs] master) with 36-byte-strided gathers: 0.9 / 1.8 TB/s in the r02 trace (388 + 398 us per
// D backward). Here a work item is (row o, chunk of <= 128 input channels): its RS x 128 gradient values are read as RS contiguous
// 512-byte runs (16 bytes per lane) into LDS at a pitch = 8 (mod 32) floats, then walked in the master's order -- LDS reads at <= 3-way
// conflicts, global reads / read-modify-writes of 128 RS contiguous floats, (c, rs) advanced incrementally, 32-bit arithmetic. 9 KB of
// LDS per block keeps the occupancy of a streaming kernel (a first version staged whole rows: 55 KB, two blocks per CU, SLOWER than
// the kernels it replaced -- session J). Natural-layout weights (linear, embedding) take the same walk without the staging.
// Transposed-convolution weights and RS > 16 stay with the old kernels.
// (chunk of 256 channels = 1 KB runs, 17 KB of LDS: 128 -> 256 took the backward of BigGAN-128's G / D tables from 466 / 574 us to 421 / 519 us, 512 lost a third to occupancy --
// round 6, tools/sn_bench.py over three builds)
#ifndef SNB_CW
#define SNB_CW 256
#endif
#ifndef SNB_ST
#define SNB_ST (SNB_CW + 8)
#endif
// (SNB_ST = chunk + 8 floats: pitch = 8 (mod 32))
__host__ __device__ __forceinline__ bool snb_row_ok(const sg_sn_bwd_layer& l) {
  if (l.trans) return false;
  if (l.natural == 1) return true;
  if (l.natural != 0) return false;
  const int cp = l.Cin_pad > l.Cin ? l.Cin_pad : l.Cin;
  return (cp % 4 == 0) && ((reinterpret_cast<uintptr_t>(l.dwt) & 15) == 0) && l.RS <= 16;
}
// grid (SNB_BLOCKS, layers); APPLY = false: block partials of <dWt, W> into work; true: dw += (dWt - coef u v^T) / sigma
template <bool APPLY> __global__ __launch_bounds__(256) void k_snb_rows(const sg_sn_bwd_layer* L, float* work, int nb) {
  __shared__ __attribute__((aligned(16))) float tile[16 * SNB_ST];
  __shared__ float sm[4];
  __shared__ float coef_sm;
  const sg_sn_bwd_layer l = L[blockIdx.y];
  if (!snb_row_ok(l)) return;
  if (!APPLY && !l.apply_sn) return;
  float sig = 1.f, coef = 0.f;
  if (APPLY && l.apply_sn) {
    sig = l.sigma[0];
    if (threadIdx.x < 64) {      // fixed-order sum of the block partials (every block computes the same value)
      float t = 0.f;
      for (int b = threadIdx.x; b < nb; b += 64) t += work[(long long)blockIdx.y * nb + b];
      t = wave_sum(t);
      if (threadIdx.x == 0) coef_sm = t;
    }
    __syncthreads();
    coef = coef_sm / sig;        // <dWt, W/sigma>
  }
  const float inv = 1.f / sig;
  const int RS = l.RS, cols = l.cols;
  float acc = 0.f;
  if (l.natural == 1) {
    // items = (row, chunk of 1024 columns)
    const int nch = (cols + 1023) / 1024;
    const int items = l.rows * nch;
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
      const int o = it / nch, k0 = (it - o * nch) * 1024;
      const long long base = (long long)o * cols;
      const float uo = (APPLY && l.apply_sn) ? l.u[o] * coef : 0.f;
      for (int k = k0 + threadIdx.x; k < cols && k < k0 + 1024; k += 256) {
        const float g = l.dwt[base + k];
        if (APPLY) l.dw[base + k] += l.apply_sn ? (g - uo * l.v[k]) * inv : g;
        else acc += g * l.w[base + k];
      }
    }
  } else {
    const int cp = l.Cin_pad > l.Cin ? l.Cin_pad : l.Cin;
    const int nch = (l.Cin + SNB_CW - 1) / SNB_CW;               // chunks of real channels (padding channels carry no gradient)
    const int items = l.rows * nch;
    // per-thread walks: master order k' = c RS + rs (k' += 256) and image order j = rs SNB_CW + c (4 floats per step, j += 1024)
    const int c_init = threadIdx.x / RS, rs_init = threadIdx.x - c_init * RS;
    const int dc = 256 / RS, drs = 256 - dc * RS;
    const int j_init = threadIdx.x * 4;
    const int rs2_init = j_init / SNB_CW, c2_init = j_init - rs2_init * SNB_CW;
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
      const int o = it / nch, c0 = (it - o * nch) * SNB_CW;
      const int cw = (cp - c0 < SNB_CW) ? cp - c0 : SNB_CW;      // stored channels of this chunk (multiple of 4)
      const int cwr = (l.Cin - c0 < SNB_CW) ? l.Cin - c0 : SNB_CW; // real channels
      const float* grow = l.dwt + ((long long)o * RS) * cp + c0;
      {
        int rs = rs2_init, c = c2_init;
        for (int j = j_init; j < RS * SNB_CW; j += 1024) {
          if (c < cw) *(f32x4*)(tile + rs * SNB_ST + c) = *(const f32x4*)(grow + (long long)rs * cp + c);
          rs += 1024 / SNB_CW;
        }
      }
      __syncthreads();
      const long long base = (long long)o * cols + (long long)c0 * RS;
      const float uo = (APPLY && l.apply_sn) ? l.u[o] * coef : 0.f;
      const int kn = cwr * RS;
      int c = c_init, rs = rs_init;
      for (int k = threadIdx.x; k < kn; k += 256) {
        const float g = tile[rs * SNB_ST + c];
        if (APPLY) l.dw[base + k] += l.apply_sn ? (g - uo * l.v[c0 * RS + k]) * inv : g;
        else acc += g * l.w[base + k];
        c += dc; rs += drs;
        if (rs >= RS) { rs -= RS; c++; }
      }
      __syncthreads();
    }
  }
  if (!APPLY) {
    acc = block_sum_256(acc, sm);
    if (threadIdx.x == 0) work[(long long)blockIdx.y * nb + blockIdx.x] = acc;
  }
}
// grid (SNB_BLOCKS, layers): block partials of <dWt, W>
__global__ __launch_bounds__(256) void k_snb_dot(const sg_sn_bwd_layer* L, float* work, int nb) {
  __shared__ float sm[4];
  const sg_sn_bwd_layer l = L[blockIdx.y];
  if (!l.apply_sn || snb_row_ok(l)) return;
  const long long total = (long long)l.rows * l.cols;
  float acc = 0.f;
  const int cp = l.Cin_pad > l.Cin ? l.Cin_pad : l.Cin;
  if (l.natural == 0 && cp == l.Cin) {
    // walk dwt in ITS order ([o][rs][c], contiguous reads of the big scratch) and gather w ([o][c][rs], L2-friendly 9-float strides)
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += nb * 256ll) {
      const int o = (int)(i / l.cols), j = (int)(i % l.cols);
      const int rs = j / l.Cin, c = j - rs * l.Cin;
      acc += l.dwt[i] * l.w[sn_widx(l, o, c * l.RS + rs)];
    }
  } else {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += nb * 256ll) {
      const int o = (int)(i / l.cols), k = (int)(i % l.cols);
      acc += l.dwt[snb_src_index(l, o, k)] * l.w[sn_widx(l, o, k)];
    }
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) work[(long long)blockIdx.y * nb + blockIdx.x] = acc;
}
// grid (tiles, layers)
__global__ __launch_bounds__(256) void k_snb_apply(const sg_sn_bwd_layer* L, const float* work, int nb) {
  const sg_sn_bwd_layer l = L[blockIdx.y];
  if (snb_row_ok(l)) return;
  const long long total = (long long)l.rows * l.cols;
  float sig = 1.f, coef = 0.f;
  if (l.apply_sn) {
    sig = l.sigma[0];
    float d = 0.f;
    // fixed-order sum of the block partials, spread over the first wave (every block computes the same value)
    __shared__ float coef_sm;
    if (threadIdx.x < 64) {
      float t = 0.f;
      for (int b = threadIdx.x; b < nb; b += 64) t += work[(long long)blockIdx.y * nb + b];
      t = wave_sum(t);
      if (threadIdx.x == 0) coef_sm = t;
    }
    __syncthreads();
    d = coef_sm;
    coef = d / sig;  // <dWt, W/sigma>
  }
  const float inv = 1.f / sig;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int o = (int)(i / l.cols), k = (int)(i % l.cols);
    float g = l.dwt[snb_src_index(l, o, k)];
    if (l.apply_sn) g = (g - coef * l.u[o] * l.v[k]) * inv;
    l.dw[sn_widx(l, o, k)] += g;
  }
}
extern "C" int sg_sn_backward(const sg_sn_bwd_layer* layers_dev, const sg_sn_bwd_layer* layers_host, int n, float* work, long long work_floats, sg_stream_t s) {
  SG_CHECK(layers_dev && layers_host && n > 0 && work, "sg_sn_backward: bad args");
  const int nb = SNB_BLOCKS;
  SG_CHECK((long long)n * nb <= work_floats, "sg_sn_backward: workspace too small");
  for (int i = 0; i < n; i++) {
    const sg_sn_bwd_layer& l = layers_host[i];
    SG_CHECK(l.dwt && l.dw && l.rows > 0 && l.cols > 0 && l.RS > 0 && l.Cin * l.RS == l.cols, "sg_sn_backward: bad layer");
    if (l.apply_sn) SG_CHECK(l.w && l.u && l.v && l.sigma, "sg_sn_backward: SN layer without state");
  }
  hipStream_t st = (hipStream_t)s;
  double bytes = 0.0;      // dot: dWt + W; apply: dWt + read-modify-write of dW
  for (int i = 0; i < n; i++) bytes += (double)layers_host[i].rows * layers_host[i].cols * 4.0 * ((layers_host[i].apply_sn ? 2 : 0) + 3);
  SgProfScope prof(st, bytes, 3);
  long long max_elems = 1;
  bool any_row = false, any_old = false, any_old_sn = false, any_row_sn = false;
  for (int i = 0; i < n; i++) {
    const sg_sn_bwd_layer& l = layers_host[i];
    const long long e = (long long)l.rows * l.cols;
    if (e > max_elems) max_elems = e;
    if (snb_row_ok(l)) { any_row = true; if (l.apply_sn) any_row_sn = true; }
    else { any_old = true; if (l.apply_sn) any_old_sn = true; }
  }
  if (any_old_sn) hipLaunchKernelGGL(k_snb_dot, dim3(nb, n), dim3(256), 0, st, layers_dev, work, nb);
  if (any_row && any_row_sn) hipLaunchKernelGGL(k_snb_rows<false>, dim3(nb, n), dim3(256), 0, st, layers_dev, work, nb);
  if (any_old) {
    long long tiles = (max_elems + 256 * 8 - 1) / (256 * 8); if (tiles > 4096) tiles = 4096;
    hipLaunchKernelGGL(k_snb_apply, dim3((int)tiles, n), dim3(256), 0, st, layers_dev, (const float*)work, nb);
  }
  if (any_row) hipLaunchKernelGGL(k_snb_rows<true>, dim3(nb, n), dim3(256), 0, st, layers_dev, work, nb);
  SG_LAUNCH_CHECK();
  return 0;
}
