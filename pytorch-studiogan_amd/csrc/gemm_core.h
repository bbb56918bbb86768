// gemm_core.h -- the one MFMA contraction engine every dense op on the StudioGAN hot path runs on
// (conv fwd / dgrad / wgrad as implicit GEMM, linear layers, the self-attention matmuls).
//
//   OUT[j][i] (+)= alpha * sum_k P(i,k) * Q(j,k)        i contiguous in memory
//
// P is fed to the MFMA "A" operand (rows = i), Q to the "B" operand (cols = j), so that one lane of the
// 32x32 accumulator tile owns four CONSECUTIVE i for a single j: the epilogue (bias / ReLU-mask / residual
// / 2x2 pooling / store) then moves 8-16 contiguous bytes per lane instead of 2-byte scattered stores.
//
// gfx950 specifics (see /opt/skills/guides/cdna_hip_programming.md §3, MI355X_MICROARCH.md §LDS):
//   * bf16: v_mfma_f32_32x32x16_bf16, lane l holds A[i=l&31][k=8*(l>>5)..+7], B[k=8*(l>>5)..+7][j=l&31]
//   * f32 : v_mfma_f32_32x32x2_f32 (exact fp32 fma chain),   A[i=l&31][k=l>>5],   B[k=l>>5][j=l&31]
//   * C/D : col j = l&31, row i = (reg&3) + 8*(reg>>2) + 4*(l>>5)
//   * operands whose reduction index is the SLOW memory index (wgrad: k = pixel; P^T V style products) are
//     staged as [k][row] images and fetched with ds_read_b64_tr_b16 (hardware 4x16 transpose read); images
//     whose reduction index is contiguous are staged as [row][k] with an 80-byte row pitch (conflict-free
//     ds_read_b128 for the 16-lane service groups).
//   * 256 threads = 4 waves per workgroup, register-staged double-buffered LDS, one barrier per k-tile,
//     XCD-aware (bijective) tile order so that neighbouring tiles share an L2.
#pragma once
#include "common.h"

#define SG_KC_STRIDE 80  // bytes per row of a [row][k] LDS image: 64 B payload + 16 B skew

__host__ __device__ constexpr int sg_mc_stride(int B, int es) {
  // [k][row] image pitch. bf16: pitch == 64 (mod 128) bytes puts the 4 k-rows of one ds_read_b64_tr_b16
  // lane-group on 4 disjoint 16-bank windows. fp32 reads are lane-contiguous; any 16B-aligned pitch works.
  int s = B * es;
  if (es == 2) {
    while ((s % 128) != 64) s += 16;
    return s;
  }
  return s + 16;
}

template <typename T> __device__ __forceinline__ void set_elem(u32x4& v, int e, T x);
template <> __device__ __forceinline__ void set_elem<float>(u32x4& v, int e, float x) { v[e] = __float_as_uint(x); }
template <> __device__ __forceinline__ void set_elem<bf16_t>(u32x4& v, int e, bf16_t x) {
  v[e >> 1] |= ((uint32_t)x) << (16 * (e & 1));
}

// ---------------------------------------------------------------------------------------------------
// Operand loaders. A loader hands out 16-byte chunks:
//   KC form ("k contiguous"):  chunk = elements (row, k .. k+VEC-1)      -> LDS image [row][k]
//   MC form ("row contiguous"): chunk = elements (row .. row+VEC-1, k)    -> LDS image [k][row]
// init() is called once per thread-chunk, load() every k-tile, advance() steps k by BK.
// ---------------------------------------------------------------------------------------------------
template <typename T, bool FAST = false> struct StridedKC {
  static constexpr bool KC = true;
  static constexpr int VEC = ET<T>::VEC, BK = ET<T>::BK;
  const T* base; long long bstride; int ld; int rows; int K; int vec_ok;
  struct Chunk { const T* p; int k; bool valid; };
  __device__ __forceinline__ void set_batch(int b) { base += (long long)b * bstride; }
  __device__ __forceinline__ void init(Chunk& ch, int row, int k) const {
    ch.valid = row < rows; ch.p = base + (long long)row * ld; ch.k = k;
  }
  __device__ __forceinline__ u32x4 load(const Chunk& ch) const {
    u32x4 v = zero16();
    if (FAST || vec_ok) {
      const bool ok = ch.valid & (ch.k < K);
      const u32x4 t = *(const u32x4*)(ok ? (ch.p + ch.k) : base);
      return ok ? t : v;
    }
    if constexpr (FAST) return v;
    if (!ch.valid || ch.k >= K) return v;
#pragma unroll
    for (int e = 0; e < VEC; e++) if (ch.k + e < K) set_elem<T>(v, e, ch.p[ch.k + e]);
    return v;
  }
  __device__ __forceinline__ void advance(Chunk& ch) const { ch.k += BK; }
};

template <typename T, bool FAST = false> struct StridedMC {
  static constexpr bool KC = false;
  static constexpr int VEC = ET<T>::VEC, BK = ET<T>::BK;
  const T* base; long long bstride; int ld; int rows; int K; int vec_ok;
  struct Chunk { int row; int k; };
  __device__ __forceinline__ void set_batch(int b) { base += (long long)b * bstride; }
  __device__ __forceinline__ void init(Chunk& ch, int row, int k) const { ch.row = row; ch.k = k; }
  __device__ __forceinline__ u32x4 load(const Chunk& ch) const {
    u32x4 v = zero16();
    if (FAST || vec_ok) {
      const bool ok = (ch.k < K) & (ch.row < rows);
      const u32x4 t = *(const u32x4*)(ok ? (base + (long long)ch.k * ld + ch.row) : base);
      return ok ? t : v;
    }
    if constexpr (FAST) return v;
    if (ch.k >= K || ch.row >= rows) return v;
    const T* p = base + (long long)ch.k * ld + ch.row;
#pragma unroll
    for (int e = 0; e < VEC; e++) if (ch.row + e < rows) set_elem<T>(v, e, p[e]);
    return v;
  }
  __device__ __forceinline__ void advance(Chunk& ch) const { ch.k += BK; }
};

#define SG_PIX_RELU 1       // apply ReLU to the activation while loading
#define SG_PIX_UPSAMPLE 2   // tensor is stored at (Hin/2, Win/2); read with nearest-neighbour x2
#define SG_PIX_QUAD 4       // rows enumerate output pixels quad-major (n, h/2, w/2, dy, dx) for fused 2x2 pooling
#define SG_PIX_TRANSPOSED 8 // gather form of a transposed (fractionally strided) convolution

// geometry of one activation tensor as seen by a convolution
template <typename T> struct PixGeom {
  const T* x;
  int N, Hs, Ws;        // stored spatial dims
  int Hin, Win;         // logical dims (== stored, or 2x stored with SG_PIX_UPSAMPLE)
  int C, ldx;           // channels used, element pitch between pixels
  int Ho, Wo;           // output grid of the convolution
  int R, S, stride, pad_h, pad_w;
  int flags;
  int vec_ok;           // C % VEC == 0 && ldx % VEC == 0 && 16B-aligned base
  int wshift, hshift;   // log2(Wo), log2(Ho) or -1

  __device__ __forceinline__ void pix_decompose(int pix, int& n, int& ho, int& wo) const {
    if (wshift >= 0 && hshift >= 0) {
      wo = pix & (Wo - 1); int t = pix >> wshift; ho = t & (Ho - 1); n = t >> hshift;
    } else {
      wo = pix % Wo; int t = pix / Wo; ho = t % Ho; n = t / Ho;
    }
  }
  // branch-free form for the vector path: offset (elements) of the tapped pixel, 0 + valid=false when it is padding.
  // (No early return: a branch in front of every global load makes the compiler wait for each load before issuing
  // the next one -- measured 3-5x on the weight-gradient k-loop.)
  __device__ __forceinline__ unsigned tap_off(int n, int ho, int wo, int r, int s, bool& valid) const {
    int h, w;
    bool ok = true;
    if (flags & SG_PIX_TRANSPOSED) {
      int hn = ho + pad_h - r, wn = wo + pad_w - s;
      ok = (hn >= 0) & (wn >= 0);
      if (stride == 2) { ok &= (((hn | wn) & 1) == 0); hn >>= 1; wn >>= 1; }
      else if (stride != 1) { ok &= ((hn % stride) == 0) & ((wn % stride) == 0); hn /= stride; wn /= stride; }
      h = hn; w = wn;
    } else {
      h = ho * stride - pad_h + r; w = wo * stride - pad_w + s;
    }
    ok &= ((unsigned)h < (unsigned)Hin) & ((unsigned)w < (unsigned)Win);
    if (flags & SG_PIX_UPSAMPLE) { h >>= 1; w >>= 1; }
    valid = ok;
    const unsigned off = ((unsigned)(n * Hs + h) * (unsigned)Ws + (unsigned)w) * (unsigned)ldx;
    return ok ? off : 0u;
  }
  // address of input pixel feeding output (n,ho,wo) through tap (r,s); returns false when the tap is padding
  __device__ __forceinline__ bool tap(int n, int ho, int wo, int r, int s, const T*& p) const {
    int h, w;
    if (flags & SG_PIX_TRANSPOSED) {
      int hn = ho + pad_h - r, wn = wo + pad_w - s;
      if (hn < 0 || wn < 0) return false;
      if (stride == 2) { if ((hn | wn) & 1) return false; hn >>= 1; wn >>= 1; }
      else if (stride != 1) { if (hn % stride || wn % stride) return false; hn /= stride; wn /= stride; }
      h = hn; w = wn;
    } else {
      h = ho * stride - pad_h + r; w = wo * stride - pad_w + s;
    }
    if (h < 0 || h >= Hin || w < 0 || w >= Win) return false;
    if (flags & SG_PIX_UPSAMPLE) { h >>= 1; w >>= 1; }
    p = x + (unsigned)(((unsigned)(n * Hs + h) * (unsigned)Ws + (unsigned)w) * (unsigned)ldx);  // host checks the tensor has < 2^31 elements
    return true;
  }
};

// pixel-side operand of forward / data-gradient convolution: row = output pixel, k = (r,s,c)
template <typename T, bool FAST = false> struct ConvPixKC {
  static constexpr bool KC = true;
  static constexpr int VEC = ET<T>::VEC, BK = ET<T>::BK;
  PixGeom<T> g; int rows; int K;
  struct Chunk { int n, ho, wo; int r, s, c; bool valid; };
  __device__ __forceinline__ void set_batch(int) {}
  __device__ __forceinline__ void init(Chunk& ch, int row, int k) const {
    ch.valid = row < rows;
    if (g.flags & SG_PIX_QUAD) {
      int q = row >> 2, dy = (row >> 1) & 1, dx = row & 1;
      int hq, wq;
      int Wq = g.Wo >> 1, Hq = g.Ho >> 1;
      wq = q % Wq; int t = q / Wq; hq = t % Hq; ch.n = t / Hq;
      ch.ho = 2 * hq + dy; ch.wo = 2 * wq + dx;
    } else {
      g.pix_decompose(row, ch.n, ch.ho, ch.wo);
    }
    int rs = k / g.C; ch.c = k - rs * g.C; ch.r = rs / g.S; ch.s = rs - ch.r * g.S;
  }
  __device__ __forceinline__ u32x4 load(const Chunk& ch) const {
    u32x4 v = zero16();
    if (FAST || g.vec_ok) {
      bool ok;
      const unsigned off = g.tap_off(ch.n, ch.ho, ch.wo, ch.r, ch.s, ok);
      ok = ok & ch.valid & (ch.r < g.R);
      const u32x4 t = *(const u32x4*)(g.x + (ok ? off + (unsigned)ch.c : 0u));   // unconditional load, select after
      v = ok ? t : v;
    } else if constexpr (!FAST) {
      if (!ch.valid || ch.r >= g.R) return v;
      int r = ch.r, s = ch.s, c = ch.c;
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        if (r < g.R) {
          const T* p;
          if (g.tap(ch.n, ch.ho, ch.wo, r, s, p)) set_elem<T>(v, e, p[c]);
        }
        c++; if (c == g.C) { c = 0; s++; if (s == g.S) { s = 0; r++; } }
      }
    }
    if (g.flags & SG_PIX_RELU) v = relu16<T>(v);
    return v;
  }
  __device__ __forceinline__ void advance(Chunk& ch) const {
    ch.c += BK;
    while (ch.c >= g.C) { ch.c -= g.C; ch.s++; if (ch.s == g.S) { ch.s = 0; ch.r++; } }
  }
};

// activation operand of weight-gradient convolution: row = (r,s,c) flattened, k = output pixel (n,ho,wo).
// With R=S=1 it is also the loader for the output-gradient operand (row = cout), including the
// "gradient of a fused 2x2 pooling" view (SG_PIX_UPSAMPLE).
template <typename T, bool FAST = false> struct ConvPixMC {
  static constexpr bool KC = false;
  static constexpr int VEC = ET<T>::VEC, BK = ET<T>::BK;
  PixGeom<T> g; int rows; int K;  // rows = R*S*C, K = N*Ho*Wo
  struct Chunk { int r, s, c; int pix; bool valid; };
  __device__ __forceinline__ void set_batch(int) {}
  __device__ __forceinline__ void init(Chunk& ch, int row, int k) const {
    ch.valid = row < rows;
    int rs = row / g.C; ch.c = row - rs * g.C; ch.r = rs / g.S; ch.s = rs - ch.r * g.S;
    ch.pix = k;
  }
  __device__ __forceinline__ u32x4 load(const Chunk& ch) const {
    u32x4 v = zero16();
    int n, ho, wo;
    if (FAST || g.vec_ok) {
      const bool inb = ch.valid & (ch.pix < K);
      g.pix_decompose(inb ? ch.pix : 0, n, ho, wo);
      bool ok;
      const unsigned off = g.tap_off(n, ho, wo, ch.r, ch.s, ok);
      ok = ok & inb;
      const u32x4 t = *(const u32x4*)(g.x + (ok ? off + (unsigned)ch.c : 0u));   // unconditional load, select after
      v = ok ? t : v;
    } else if constexpr (!FAST) {
      if (!ch.valid || ch.pix >= K) return v;
      g.pix_decompose(ch.pix, n, ho, wo);
      int r = ch.r, s = ch.s, c = ch.c;
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        if (r < g.R) {
          const T* p;
          if (g.tap(n, ho, wo, r, s, p)) set_elem<T>(v, e, p[c]);
        }
        c++; if (c == g.C) { c = 0; s++; if (s == g.S) { s = 0; r++; } }
      }
    }
    if (g.flags & SG_PIX_RELU) v = relu16<T>(v);
    return v;
  }
  __device__ __forceinline__ void advance(Chunk& ch) const { ch.pix += BK; }
};

// ---------------------------------------------------------------------------------------------------
// Epilogue: out[j][i0..i0+3] = beta*res + mask( alpha * pool(acc) + bias[i] )
// ---------------------------------------------------------------------------------------------------
#define SG_EPI_OUT_F32 1    // store fp32 regardless of T
#define SG_EPI_ATOMIC 2     // atomicAdd fp32 (split-K weight gradients)
#define SG_EPI_POOL 4       // sum the 4 rows of each j-quad (rows must be quad-major), output row = j>>2
#define SG_EPI_RELU 8       // ReLU on the result
#define SG_EPI_RES_F32 16   // residual operand is fp32

template <typename T> struct Epilogue {
  void* out; long long out_bstride; int ldo;
  const float* bias;
  const void* res; long long res_bstride; int ldr; float beta;
  const T* mask; long long mask_bstride; int ldm;
  float alpha; const float* alpha_ptr;
  int flags; int I; int J;
  long long split_stride;  // != 0: k-split s writes its partial tile to out + s*split_stride (two-stage split-K)

  __device__ __forceinline__ void set_batch(int b) {
    if (flags & (SG_EPI_OUT_F32 | SG_EPI_ATOMIC)) out = (float*)out + (long long)b * out_bstride;
    else out = (T*)out + (long long)b * out_bstride;
    if (res) {
      if (flags & SG_EPI_RES_F32) res = (const float*)res + (long long)b * res_bstride;
      else res = (const T*)res + (long long)b * res_bstride;
    }
    if (mask) mask += (long long)b * mask_bstride;
  }

  // lane owns (j, i0..i0+3)
  __device__ __forceinline__ void store(int j, int i0, float v[4], float a) const {
    if (prep(j, i0, v, a)) write(j, i0, v);
  }
  // everything but the final store: pooling, scale, bias, mask, residual, ReLU. Returns false when this lane has nothing to
  // write; j becomes the OUTPUT row (pooled index with SG_EPI_POOL). Every lane of the wave must call it (pool shuffles).
  __device__ __forceinline__ bool prep(int& j, int i0, float v[4], float a) const {
    const int lane = threadIdx.x & 63;
    if (flags & SG_EPI_POOL) {
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = quad_sum(v[e]);
      if (lane & 3) return false;
      j >>= 2;
    }
    const int Jout = (flags & SG_EPI_POOL) ? (J >> 2) : J;
    if (j >= Jout || i0 >= I) return false;
    const int ne = (I - i0) < 4 ? (I - i0) : 4;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float x = v[e] * a;
      if (bias && e < ne) x += bias[i0 + e];
      v[e] = x;
    }
    if (mask) {
      const T* m = mask + (long long)j * ldm + i0;
#pragma unroll
      for (int e = 0; e < 4; e++) if (e < ne) { if (!(to_f<T>(m[e]) > 0.f)) v[e] = 0.f; }
    }
    if (res) {
      if (flags & SG_EPI_RES_F32) {
        const float* r = (const float*)res + (long long)j * ldr + i0;
#pragma unroll
        for (int e = 0; e < 4; e++) if (e < ne) v[e] += beta * r[e];
      } else {
        const T* r = (const T*)res + (long long)j * ldr + i0;
#pragma unroll
        for (int e = 0; e < 4; e++) if (e < ne) v[e] += beta * to_f<T>(r[e]);
      }
    }
    if (flags & SG_EPI_RELU) {
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
    }
    return true;
  }
  __device__ __forceinline__ void write(int j, int i0, const float v[4]) const {
    const int ne = (I - i0) < 4 ? (I - i0) : 4;
    if (flags & SG_EPI_ATOMIC) {
      float* o = (float*)out + (long long)j * ldo + i0;
#pragma unroll
      for (int e = 0; e < 4; e++) if (e < ne) unsafeAtomicAdd(o + e, v[e]);
    } else if ((flags & SG_EPI_OUT_F32) || sizeof(T) == 4) {
      float* o = (float*)out + (long long)j * ldo + i0;
      if (ne == 4 && ((ldo & 3) == 0) && ((((uintptr_t)out) & 15) == 0)) {
        f32x4 t = {v[0], v[1], v[2], v[3]};
        *(f32x4*)o = t;
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) if (e < ne) o[e] = v[e];
      }
    } else {
      bf16_t* o = (bf16_t*)out + (long long)j * ldo + i0;
      if (ne == 4 && ((ldo & 3) == 0) && ((((uintptr_t)out) & 7) == 0)) {
        u32x2 t;
        t[0] = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        t[1] = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        *(u32x2*)o = t;
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) if (e < ne) o[e] = f2bf(v[e]);
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------
// LDS fragment fetch
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// bf16, [row][k] image: 8 consecutive k for row (l&31): one ds_read_b128
__device__ __forceinline__ bf16x8_t frag_kc_bf16(const char* tile, int row0, int ks) {
  const int l = threadIdx.x & 63;
  const char* p = tile + (row0 + (l & 31)) * SG_KC_STRIDE + (ks * 16 + 8 * (l >> 5)) * 2;
  u32x4 v = *(const u32x4*)p;
  return __builtin_bit_cast(bf16x8_t, v);
}
// bf16, [k][row] image: hardware transpose read. Within each 16-lane group, lane t supplies the address of
// 4 contiguous rows (8 bytes) M[k0 + (t>>2)][r16 + 4*(t&3) ..+3] and receives M[k0+0..3][r16 + t].
template <bool TR>
__device__ __forceinline__ bf16x8_t frag_mc_bf16(const char* tile, int pitch, int row0, int ks) {
  const int l = threadIdx.x & 63;
  if (TR) {
    const int g16 = l >> 4, t = l & 15;
    const int r16 = row0 + 16 * (g16 & 1);
    const int k0 = ks * 16 + 8 * (g16 >> 1);
    const char* p = tile + (k0 + (t >> 2)) * pitch + (r16 + 4 * (t & 3)) * 2;
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * pitch));
    s16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return __builtin_bit_cast(bf16x8_t, r);
  } else {
    // reference gather (8 x ds_read_u16), kept as the cross-check for the transpose-read path
    const int k0 = ks * 16 + 8 * (l >> 5);
    const int row = row0 + (l & 31);
    s16x8 r;
#pragma unroll
    for (int e = 0; e < 8; e++) r[e] = *(const short*)(tile + (k0 + e) * pitch + row * 2);
    return __builtin_bit_cast(bf16x8_t, r);
  }
}
__device__ __forceinline__ float frag_kc_f32(const char* tile, int row0, int kk) {
  const int l = threadIdx.x & 63;
  return *(const float*)(tile + (row0 + (l & 31)) * SG_KC_STRIDE + (kk * 2 + (l >> 5)) * 4);
}
__device__ __forceinline__ float frag_mc_f32(const char* tile, int pitch, int row0, int kk) {
  const int l = threadIdx.x & 63;
  return *(const float*)(tile + (kk * 2 + (l >> 5)) * pitch + (row0 + (l & 31)) * 4);
}

// fp32 [row][k] image, SPLIT = 3 ("bf16x3"): the 8 consecutive k of row (l & 31) that a 32x32x16 bf16 MFMA wants from this lane, as two bf16 vectors
// x = hi + lo + O(2^-17 |x|): hi = bf16(x) (round to nearest even), lo = bf16(x - hi). Two ds_read_b128 (80-byte pitch: conflict-free per 16-lane group).
__device__ __forceinline__ void frag_kc_f32_split(const char* tile, int row0, bf16x8_t& hi, bf16x8_t& lo) {
  const int l = threadIdx.x & 63;
  const char* p = tile + (row0 + (l & 31)) * SG_KC_STRIDE + 32 * (l >> 5);
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 16);
  float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  u32x4 h, r;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const uint32_t hp = pack2bf(x[2 * e], x[2 * e + 1]);
    const float h0 = __uint_as_float(hp << 16), h1 = __uint_as_float(hp & 0xffff0000u);
    h[e] = hp;
    r[e] = pack2bf(x[2 * e] - h0, x[2 * e + 1] - h1);
  }
  hi = __builtin_bit_cast(bf16x8_t, h);
  lo = __builtin_bit_cast(bf16x8_t, r);
}

// the same from a [k][row] image (weight gradient: k = pixel): eight ds_read_b32 at the image pitch; neighbouring lanes read neighbouring rows (conflict-free)
__device__ __forceinline__ void frag_mc_f32_split(const char* tile, int pitch, int row0, bf16x8_t& hi, bf16x8_t& lo) {
  const int l = threadIdx.x & 63;
  const char* p = tile + (8 * (l >> 5)) * pitch + (row0 + (l & 31)) * 4;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; e++) x[e] = *(const float*)(p + e * pitch);
  u32x4 h, r;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const uint32_t hp = pack2bf(x[2 * e], x[2 * e + 1]);
    const float h0 = __uint_as_float(hp << 16), h1 = __uint_as_float(hp & 0xffff0000u);
    h[e] = hp;
    r[e] = pack2bf(x[2 * e] - h0, x[2 * e + 1] - h1);
  }
  hi = __builtin_bit_cast(bf16x8_t, h);
  lo = __builtin_bit_cast(bf16x8_t, r);
}

// ---------------------------------------------------------------------------------------------------
// The kernel
// ---------------------------------------------------------------------------------------------------
// SPLIT (fp32 operands, either LDS form): 0 = v_mfma_f32_32x32x2_f32, the exact fp32 FMA chain (64 cycles per SIMD and k-pair: 1/16 of the bf16 rate);
// 3 = "bf16x3": every fp32 operand element is split into two bf16 terms at fragment time and a k-tile of 16 runs as THREE v_mfma_f32_32x32x16_bf16
// (lo*hi + hi*lo + hi*hi, fp32 accumulation) = 96 cycles per SIMD instead of 512. Dropped: lo*lo and the second-order split remainders, ~2^-16 relative per
// product with random sign -- 60x finer than the TF32 convolutions (10-bit mantissa) torch runs the reference's "fp32" evaluation with on its usual hardware.
template <typename T, class LP, class LQ, int BI, int BJ, int WI, int WJ, bool TR, int SPLIT = 0>
__global__ __launch_bounds__(256) void sg_gemm_kernel(LP lp, LQ lq, Epilogue<T> epi, int I, int J, int K,
                                                       int klen, int tilesI, int tilesJ) {
  constexpr int VEC = ET<T>::VEC, BK = ET<T>::BK, ES = sizeof(T);
  constexpr int P_PITCH = LP::KC ? SG_KC_STRIDE : sg_mc_stride(BI, ES);
  constexpr int Q_PITCH = LQ::KC ? SG_KC_STRIDE : sg_mc_stride(BJ, ES);
  constexpr int PB = LP::KC ? BI * SG_KC_STRIDE : BK * P_PITCH;
  constexpr int QB = LQ::KC ? BJ * SG_KC_STRIDE : BK * Q_PITCH;
  constexpr int NP = (BI * 4 + 255) / 256, NQ = (BJ * 4 + 255) / 256;
  constexpr int TI = BI / WI / 32, TJ = BJ / WJ / 32;
  static_assert(WI * WJ == 4, "4 waves");
  static_assert(BI % (WI * 32) == 0 && BJ % (WJ * 32) == 0, "tile/wave mismatch");

  __shared__ __attribute__((aligned(16))) char smem[2 * (PB + QB)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware bijective remap: consecutive logical tiles stay on one XCD (shared L2) -- guide §5.5 T1
  const int nt = tilesI * tilesJ;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tI = bid % tilesI, tJ = bid / tilesI;
  const int i0 = tI * BI, j0 = tJ * BJ;
  const int k_begin = blockIdx.y * klen;
  const int k_end = (k_begin + klen < K) ? (k_begin + klen) : K;
  if (gridDim.z > 1) { lp.set_batch(blockIdx.z); lq.set_batch(blockIdx.z); epi.set_batch(blockIdx.z); }
  if (epi.split_stride) epi.out = (float*)epi.out + (long long)blockIdx.y * epi.split_stride;

  typename LP::Chunk pc[NP];
  typename LQ::Chunk qc[NQ];
  int p_off[NP], q_off[NQ];
#pragma unroll
  for (int n = 0; n < NP; n++) {
    int c = tid + 256 * n;
    if (c >= BI * 4) c = BI * 4 - 1;  // duplicate work, harmless (same data, same address)
    int row, ko;
    if (LP::KC) { row = c >> 2; ko = (c & 3) * VEC; p_off[n] = row * SG_KC_STRIDE + (c & 3) * 16; }
    else { constexpr int CPR = BI / VEC; row = (c % CPR) * VEC; ko = c / CPR; p_off[n] = ko * P_PITCH + (c % CPR) * 16; }
    lp.init(pc[n], i0 + row, k_begin + ko);
  }
#pragma unroll
  for (int n = 0; n < NQ; n++) {
    int c = tid + 256 * n;
    if (c >= BJ * 4) c = BJ * 4 - 1;
    int row, ko;
    if (LQ::KC) { row = c >> 2; ko = (c & 3) * VEC; q_off[n] = row * SG_KC_STRIDE + (c & 3) * 16; }
    else { constexpr int CPR = BJ / VEC; row = (c % CPR) * VEC; ko = c / CPR; q_off[n] = ko * Q_PITCH + (c % CPR) * 16; }
    lq.init(qc[n], j0 + row, k_begin + ko);
  }

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; a++)
#pragma unroll
    for (int b = 0; b < TJ; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  const int wi = wave % WI, wj = wave / WI;
  const int wi0 = wi * (BI / WI), wj0 = wj * (BJ / WJ);

  u32x4 pr[NP], qr[NQ];
#pragma unroll
  for (int n = 0; n < NP; n++) pr[n] = lp.load(pc[n]);
#pragma unroll
  for (int n = 0; n < NQ; n++) qr[n] = lq.load(qc[n]);
#pragma unroll
  for (int n = 0; n < NP; n++) *(u32x4*)(smem + p_off[n]) = pr[n];
#pragma unroll
  for (int n = 0; n < NQ; n++) *(u32x4*)(smem + PB + q_off[n]) = qr[n];
  __syncthreads();

  const int nk = (k_end - k_begin + BK - 1) / BK;
  for (int it = 0; it < nk; it++) {
    const char* ps = smem + (it & 1) * (PB + QB);
    const char* qs = ps + PB;
    const bool more = (it + 1 < nk);
    if (more) {
#pragma unroll
      for (int n = 0; n < NP; n++) { lp.advance(pc[n]); pr[n] = lp.load(pc[n]); }
#pragma unroll
      for (int n = 0; n < NQ; n++) { lq.advance(qc[n]); qr[n] = lq.load(qc[n]); }
    }
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        bf16x8_t pf[TI], qf[TJ];
#pragma unroll
        for (int a = 0; a < TI; a++)
          pf[a] = LP::KC ? frag_kc_bf16(ps, wi0 + a * 32, ks) : frag_mc_bf16<TR>(ps, P_PITCH, wi0 + a * 32, ks);
#pragma unroll
        for (int b = 0; b < TJ; b++)
          qf[b] = LQ::KC ? frag_kc_bf16(qs, wj0 + b * 32, ks) : frag_mc_bf16<TR>(qs, Q_PITCH, wj0 + b * 32, ks);
#pragma unroll
        for (int a = 0; a < TI; a++)
#pragma unroll
          for (int b = 0; b < TJ; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[a], qf[b], acc[a][b], 0, 0, 0);
      }
    } else if constexpr (SPLIT == 3) {
      bf16x8_t ph[TI], pl[TI], qh[TJ], ql[TJ];
#pragma unroll
      for (int a = 0; a < TI; a++) {
        if constexpr (LP::KC) frag_kc_f32_split(ps, wi0 + a * 32, ph[a], pl[a]); else frag_mc_f32_split(ps, P_PITCH, wi0 + a * 32, ph[a], pl[a]);
      }
#pragma unroll
      for (int b = 0; b < TJ; b++) {
        if constexpr (LQ::KC) frag_kc_f32_split(qs, wj0 + b * 32, qh[b], ql[b]); else frag_mc_f32_split(qs, Q_PITCH, wj0 + b * 32, qh[b], ql[b]);
      }
#pragma unroll
      for (int a = 0; a < TI; a++)
#pragma unroll
        for (int b = 0; b < TJ; b++) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl[a], qh[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph[a], ql[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph[a], qh[b], acc[a][b], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int kk = 0; kk < 8; kk++) {
        float pf[TI], qf[TJ];
#pragma unroll
        for (int a = 0; a < TI; a++)
          pf[a] = LP::KC ? frag_kc_f32(ps, wi0 + a * 32, kk) : frag_mc_f32(ps, P_PITCH, wi0 + a * 32, kk);
#pragma unroll
        for (int b = 0; b < TJ; b++)
          qf[b] = LQ::KC ? frag_kc_f32(qs, wj0 + b * 32, kk) : frag_mc_f32(qs, Q_PITCH, wj0 + b * 32, kk);
#pragma unroll
        for (int a = 0; a < TI; a++)
#pragma unroll
          for (int b = 0; b < TJ; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[a], qf[b], acc[a][b], 0, 0, 0);
      }
    }
    if (more) {
      char* pd = smem + ((it + 1) & 1) * (PB + QB);
#pragma unroll
      for (int n = 0; n < NP; n++) *(u32x4*)(pd + p_off[n]) = pr[n];
#pragma unroll
      for (int n = 0; n < NQ; n++) *(u32x4*)(pd + PB + q_off[n]) = qr[n];
    }
    __syncthreads();
  }

  float a = epi.alpha;
  if (epi.alpha_ptr) a *= *epi.alpha_ptr;
#pragma unroll
  for (int ta = 0; ta < TI; ta++)
#pragma unroll
    for (int tb = 0; tb < TJ; tb++) {
      const int j = j0 + wj0 + tb * 32 + (lane & 31);
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int ii = i0 + wi0 + ta * 32 + 8 * g4 + 4 * (lane >> 5);
        float v[4] = {acc[ta][tb][4 * g4 + 0], acc[ta][tb][4 * g4 + 1], acc[ta][tb][4 * g4 + 2], acc[ta][tb][4 * g4 + 3]};
        epi.store(j, ii, v, a);
      }
    }
}

// host-side launch helper -------------------------------------------------------------------------
struct SgTileCfg { int BI, BJ; };

template <typename T, class LP, class LQ, int BI, int BJ, int WI, int WJ, bool TR = true, int SPLIT = 0>
static inline void sg_launch_gemm(const LP& lp, const LQ& lq, const Epilogue<T>& epi, int I, int J, int K,
                                  int splits, int batch, hipStream_t stream) {
  constexpr int BK = ET<T>::BK;
  const int tilesI = (I + BI - 1) / BI, tilesJ = (J + BJ - 1) / BJ;
  int klen = K;
  if (splits > 1) {
    klen = (K + splits - 1) / splits;
    klen = ((klen + BK - 1) / BK) * BK;
    splits = (K + klen - 1) / klen;
  } else splits = 1;
  dim3 grid(tilesI * tilesJ, splits, batch);
  hipLaunchKernelGGL((sg_gemm_kernel<T, LP, LQ, BI, BJ, WI, WJ, TR, SPLIT>), grid, dim3(256), 0, stream, lp, lq, epi, I, J, K, klen,
                     tilesI, tilesJ);
}
