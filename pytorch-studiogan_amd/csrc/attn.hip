// attn.hip -- fused score kernels of the self-attention block (reference src/utils/ops.py:83-103), bf16 path.
//
// The unfused chain materialises the fp32 score matrix S = theta . phi^T ([B, HW, HW/4]: 4.3 GB at BigGAN-128's 64x64
// attention, batch 256), re-reads it for the softmax and writes the bf16 probabilities: ~13 GB of HBM traffic per forward for
// 0.26 TFLOP. Here the scores never leave the registers:
//
//   sg_attn_probs_fwd : P = softmax_k(theta_q . phi_k) (bf16) and the row log-sum-exp (fp32), two MFMA passes over the keys
//                       (pass 1: online max / sum, pass 2: normalise and store) -- HBM: write P once.
//   sg_attn_ds_bwd    : dS = P * (dP - delta), P recomputed from theta/phi/lse, dP = dO . V^T by MFMA, delta_q = sum_k P dP
//                       accumulated in fp32 by a first pass over the keys (exact row-sum-zero property of the softmax
//                       Jacobian; the <dO, O> shortcut would inherit O's bf16 rounding) -- HBM: write dS once; no fp32 dP, no P read.
//
// The surrounding products (O = P V, dV = P^T dO, dtheta = dS phi, dphi = dS^T theta) stay on the batched GEMM engine.
//
// Layout: one workgroup = 4 waves = 128 queries of one image; MFMA 32x32x16 with A = keys (rows), B = queries (rows), so a
// lane owns ONE query (l & 31) and 16 of the 32 keys of a block: the softmax statistics are in-lane plus one cross-half
// exchange. Key-side operands ([rows][32 channels] bf16 = 64-byte rows) are staged by LDS-DMA with the 16-byte chunk index
// XOR-ed by (row >> 2) & 3 on the source side (conflict-free ds_read_b128, MI355X_MICROARCH.md §LDS).
#include "common.h"
#include "../../include/sgamd.h"

// Inner-loop unrolling of the streaming kernels (k_attn_fwd_flash pass 2, k_attn_bwd_q, k_attn_bwd_k): left to the compiler these loops were unrolled
// eight times and their fragment loads hoisted -- 276 registers for k_attn_fwd_flash<3>, 324 / 432 for k_attn_bwd_k<2> / <3>: ONE wave per SIMD under kernels whose
// MFMA -> exp -> pack -> MFMA chains have nothing but other waves to hide behind (round 5, tools/isa_mix.py).
#ifndef AT_UNROLL
#define AT_UNROLL 1
#endif
// workgroups per CU the staging areas allow ((1 + NCG) * 16 KiB each): the register budget is set to match
#ifndef AT_WAVES
#define AT_WAVES(NCG) ((NCG) <= 2 ? 3 : 2)
#endif
typedef __attribute__((address_space(1))) const void* at_gptr_t;
typedef __attribute__((address_space(3))) void* at_lptr_t;
typedef __bf16 at_bf16x8 __attribute__((ext_vector_type(8)));
typedef float at_f32x16 __attribute__((ext_vector_type(16)));

__device__ u32x4 sg_attn_zero[4];

// stage `rows` rows of 32 channels (64 B) starting at channel c0 of a [rows][ld] bf16 matrix into a lane-linear LDS image
template <int NW> __device__ __forceinline__ void at_stage(char* img, const bf16_t* src, int rows, int ld, int c0, int C, int wave, int lane) {
  const int r16 = lane >> 2;
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);     // logical 16-byte chunk this lane fetches (row >> 2 == lane >> 4 inside a group)
  const int c = c0 + chunk * 8;
  for (int g = wave; g < rows / 16; g += NW) {
    const int row = g * 16 + r16;
    const bf16_t* p = (c < C) ? (src + (long long)row * ld + c) : (const bf16_t*)sg_attn_zero;
    __builtin_amdgcn_global_load_lds((at_gptr_t)p, (at_lptr_t)(img + g * 1024), 16, 0, 0);
  }
}
// MFMA A fragment: rows = 32 keys of block kb, k = 8 channels starting at 16 t + 8 h of the staged 32
__device__ __forceinline__ at_bf16x8 at_frag(const char* img, int kb, int t, int lane) {
  const int row = kb * 32 + (lane & 31);
  const int slot = (2 * t + (lane >> 5)) ^ ((row >> 2) & 3);
  u32x4 v = *(const u32x4*)(img + row * 64 + slot * 16);
  return __builtin_bit_cast(at_bf16x8, v);
}
// B fragment straight from global memory: row `row` of a [.][ld] matrix, 8 channels at c (zero beyond C)
__device__ __forceinline__ u32x4 at_gfrag(const bf16_t* base, long long row, int ld, int c, int C) {
  u32x4 z = {0u, 0u, 0u, 0u};
  return (c < C) ? *(const u32x4*)(base + row * ld + c) : z;
}
// exchange between the two lane halves of a query: v_permlane32_swap_b32 (gfx950) swaps lanes 32-63 of one register with lanes 0-31 of another in the vector pipe; with both
// registers = v, every lane ends up holding {its own value, its partner's} in the pair, in either order -- which a maximum or a sum does not care about. (__shfl_xor(v, 32) is a
// ds_bpermute_b32: an LDS round trip on the per-block MFMA -> max -> exp -> MFMA chain of the streaming forward.)
__device__ __forceinline__ float at_half_max(float v) {
  const uint32_t u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float at_half_sum(float v) {
  const uint32_t u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// 16 fp32 values of one query (keys {0-3, 8-11, 16-19, 24-27} + 4 h of a 32-key block) -> 16 CONTIGUOUS bf16 keys (16 h ..)
__device__ __forceinline__ void at_store16(bf16_t* dst, const float* p, int h) {
  uint32_t own[8];
#pragma unroll
  for (int i = 0; i < 8; i++) own[i] = (uint32_t)f2bf(p[2 * i]) | ((uint32_t)f2bf(p[2 * i + 1]) << 16);
  uint32_t recv[4];
#pragma unroll
  for (int i = 0; i < 4; i++) recv[i] = __shfl_xor(h ? own[i] : own[4 + i], 32, 64);
  u32x4 a, b;
  if (h == 0) { a = {own[0], own[1], recv[0], recv[1]}; b = {own[2], own[3], recv[2], recv[3]}; }
  else        { a = {recv[0], recv[1], own[4], own[5]}; b = {recv[2], recv[3], own[6], own[7]}; }
  *(u32x4*)dst = a;
  *(u32x4*)(dst + 8) = b;
}

// grid (HW / 128, B), 256 threads. LDS: HW4 * 64 bytes.
__global__ __launch_bounds__(256) void k_attn_probs_fwd(const bf16_t* theta, const bf16_t* phi, bf16_t* P, float* lse, int HW, int HW4, int Dp) {
  extern __shared__ __attribute__((aligned(16))) char at_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int q = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const int h = lane >> 5;
  at_stage<4>(at_smem, phi + (long long)b * HW4 * Dp, HW4, Dp, 0, Dp, wave, lane);
  const u32x4 q0 = at_gfrag(theta, (long long)b * HW + q, Dp, 8 * h, Dp);
  const u32x4 q1 = at_gfrag(theta, (long long)b * HW + q, Dp, 16 + 8 * h, Dp);
  const at_bf16x8 qf0 = __builtin_bit_cast(at_bf16x8, q0), qf1 = __builtin_bit_cast(at_bf16x8, q1);
  __syncthreads();
  const int nb = HW4 / 32;
  float m = -3.0e38f, l = 0.f;
  for (int kb = 0; kb < nb; kb++) {
    at_f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; r++) s[r] = 0.f;
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(at_smem, kb, 0, lane), qf0, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(at_smem, kb, 1, lane), qf1, s, 0, 0, 0);
    float bm = s[0];
#pragma unroll
    for (int r = 1; r < 16; r++) bm = fmaxf(bm, s[r]);
    const float mn = fmaxf(m, bm);
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) acc += __expf(s[r] - mn);
    l = l * __expf(m - mn) + acc;
    m = mn;
  }
  {  // combine the two halves of the wave (same query, disjoint keys)
    const float mo = at_half_max(m);
    l = at_half_sum(l * __expf(m - mo));
    m = mo;
  }
  const float inv = 1.f / l;
  if (h == 0) lse[(long long)b * HW + q] = m + __logf(l);
  bf16_t* prow = P + ((long long)b * HW + q) * HW4 + 16 * h;
  for (int kb = 0; kb < nb; kb++) {
    at_f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; r++) s[r] = 0.f;
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(at_smem, kb, 0, lane), qf0, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(at_smem, kb, 1, lane), qf1, s, 0, 0, 0);
    float p[16];
#pragma unroll
    for (int r = 0; r < 16; r++) p[r] = __expf(s[r] - m) * inv;
    at_store16(prow + kb * 32, p, h);
  }
}


// ---- fused forward: O = softmax(theta phi^T) V without a trip of the probabilities through HBM for the product --------------------------
// Same two passes over the keys as k_attn_probs_fwd (identical P: the bf16 probabilities are what the MFMA consumes, and what is stored
// for the backward when STORE_P), plus in pass 2, per 32-key block, O^T[c][q] += V^T[c][k] P[k][q] on the MFMA: the probabilities go from
// the score accumulators straight into the B operand. A lane owns keys {0-3, 8-11, 16-19, 24-27} + 4h of a block for its query; the
// contraction index of the second product is simply taken in THAT order (slot (h, e) <-> key 4h + e, 8 + 4h + e - 4, ...), and the V^T
// fragments are gathered in the same order by ds_read_b64_tr_b16 from the [key][32 channels] chunk image -- no cross-lane exchange.
// LDS: phi of the whole image (HW4 * 64 B) + one 256-key chunk of V (NCG * 16 KiB).
__device__ __forceinline__ at_bf16x8 at_vfrag(const char* img, int kbase, int lane) {
  // V^T fragment: channel = lane & 31 of this 32-channel image, keys kbase .. kbase + 3 (elements 0-3) and kbase + 8 .. + 11 (elements 4-7);
  // kbase is a multiple of 4, so the four rows one 16-lane group reads share their swizzle key
  const int g16 = lane >> 4, t = lane & 15;
  const int row = kbase + (t >> 2);
  const int slot = (2 * (g16 & 1) + ((t & 3) >> 1)) ^ ((row >> 2) & 3);
  const char* p = img + row * 64 + slot * 16 + 8 * (t & 1);
  const int slot2 = (2 * (g16 & 1) + ((t & 3) >> 1)) ^ (((row + 8) >> 2) & 3);
  const char* p2 = img + (row + 8) * 64 + slot2 * 16 + 8 * (t & 1);
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p2);
  s16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return __builtin_bit_cast(at_bf16x8, r);
}
template <int NCG, bool STORE_P>
__global__ __launch_bounds__(256) void k_attn_fwd_fused(const bf16_t* theta, const bf16_t* phi, const bf16_t* g, bf16_t* P, float* lse, bf16_t* O,
                                                        int HW, int HW4, int Dp, int Cg) {
  constexpr int KC = 256;
  extern __shared__ __attribute__((aligned(16))) char at_smem[];
  char* kimg = at_smem;
  char* vimg = at_smem + HW4 * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int q = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const int h = lane >> 5;
  at_stage<4>(kimg, phi + (long long)b * HW4 * Dp, HW4, Dp, 0, Dp, wave, lane);
  const u32x4 q0 = at_gfrag(theta, (long long)b * HW + q, Dp, 8 * h, Dp);
  const u32x4 q1 = at_gfrag(theta, (long long)b * HW + q, Dp, 16 + 8 * h, Dp);
  const at_bf16x8 qf0 = __builtin_bit_cast(at_bf16x8, q0), qf1 = __builtin_bit_cast(at_bf16x8, q1);
  __syncthreads();
  const int nb = HW4 / 32;
  float m = -3.0e38f, l = 0.f;
  for (int kb = 0; kb < nb; kb++) {
    at_f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; r++) s[r] = 0.f;
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(kimg, kb, 0, lane), qf0, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(kimg, kb, 1, lane), qf1, s, 0, 0, 0);
    float bm = s[0];
#pragma unroll
    for (int r = 1; r < 16; r++) bm = fmaxf(bm, s[r]);
    const float mn = fmaxf(m, bm);
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) acc += __expf(s[r] - mn);
    l = l * __expf(m - mn) + acc;
    m = mn;
  }
  {
    const float mo = at_half_max(m);
    l = at_half_sum(l * __expf(m - mo));
    m = mo;
  }
  const float inv = 1.f / l;
  if (h == 0) lse[(long long)b * HW + q] = m + __logf(l);
  bf16_t* prow = STORE_P ? (P + ((long long)b * HW + q) * HW4 + 16 * h) : nullptr;
  at_f32x16 o[NCG];
#pragma unroll
  for (int cg = 0; cg < NCG; cg++)
#pragma unroll
    for (int r = 0; r < 16; r++) o[cg][r] = 0.f;
  for (int k0 = 0; k0 < HW4; k0 += KC) {
    __syncthreads();                                             // previous V chunk fully consumed
#pragma unroll
    for (int cg = 0; cg < NCG; cg++) at_stage<4>(vimg + cg * KC * 64, g + ((long long)b * HW4 + k0) * Cg, KC, Cg, cg * 32, Cg, wave, lane);
    __syncthreads();
    for (int kc = 0; kc < KC / 32; kc++) {
      const int kb = k0 / 32 + kc;
      at_f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; r++) s[r] = 0.f;
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(kimg, kb, 0, lane), qf0, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(kimg, kb, 1, lane), qf1, s, 0, 0, 0);
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; r++) p[r] = __expf(s[r] - m) * inv;
      if (STORE_P) at_store16(prow + kb * 32, p, h);
      u32x4 pa, pb;                                               // regs 0-7 / 8-15 = contraction slots of the two k-steps
#pragma unroll
      for (int i = 0; i < 4; i++) { pa[i] = pack2bf(p[2 * i], p[2 * i + 1]); pb[i] = pack2bf(p[8 + 2 * i], p[8 + 2 * i + 1]); }
      const at_bf16x8 pfa = __builtin_bit_cast(at_bf16x8, pa), pfb = __builtin_bit_cast(at_bf16x8, pb);
#pragma unroll
      for (int cg = 0; cg < NCG; cg++) {
        const char* vi = vimg + cg * KC * 64;
        o[cg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_vfrag(vi, kc * 32 + 4 * h, lane), pfa, o[cg], 0, 0, 0);
        o[cg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_vfrag(vi, kc * 32 + 16 + 4 * h, lane), pfb, o[cg], 0, 0, 0);
      }
    }
  }
  // O^T tile: lane = (query, h) holds channels cg * 32 + 8 * (r >> 2) + 4 h + (r & 3): four consecutive channels per register group
  bf16_t* orow = O + ((long long)b * HW + q) * Cg;
#pragma unroll
  for (int cg = 0; cg < NCG; cg++)
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
      const int c0 = cg * 32 + 8 * g4 + 4 * h;
      if (c0 < Cg) {
        u32x2 v = {pack2bf(o[cg][4 * g4 + 0], o[cg][4 * g4 + 1]), pack2bf(o[cg][4 * g4 + 2], o[cg][4 * g4 + 3])};
        *(u32x2*)(orow + c0) = v;
      }
    }
}

// ---- fused forward when nothing stores P: ONE pass over the keys, running maximum with a deferred rescale ------------------------------------------------
// History. k_attn_fwd_fused<., false> (whole key image in LDS, two exp passes): 811 us on D's attention (B 256, 4096 queries x 1024 keys, 48 channels). Round 2:
// keys and values streamed in 256-key chunks (LDS (1 + NCG) * 16 KiB), pass 1 = running row maximum only (scores + 16 v_max per block), pass 2 = p = exp2(s log2e -
// m log2e), row sum and O' = sum p V with the unnormalised bf16 p as the MFMA operand, O = O' / l. Round 5: the maximum pass is gone. The running maximum m of a
// query is only RAISED when a block's maximum exceeds it by more than AT_THR (then l and the O' accumulators of that query are rescaled by exp(m_old - m_new), as in
// online softmax); below the threshold the stale m is kept and p = exp(s - m) <= e^AT_THR simply carries a common factor that cancels in O' / l and in
// lse = m + log l. bf16 p has fp32's exponent range, l and O' accumulate in fp32: nothing is lost to the factor. With a threshold the rescale runs once per query
// (at the first block) instead of on almost every block for some query of the wave; what is saved per 32-key block is the first pass's two score MFMAs, its two
// fragment reads and its 16 v_max, and per chunk one key staging and two barriers.
// m is kept equal in the two lane halves of a query (they hold disjoint keys of the block, and the second product contracts over both halves' probabilities).
#ifndef AT_THR
#define AT_THR 8.0f
#endif
template <int NCG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(AT_WAVES(NCG), AT_WAVES(NCG)))) void k_attn_fwd_flash(const bf16_t* theta, const bf16_t* phi, const bf16_t* g, float* lse, bf16_t* O, float* O32,
                                                        int HW, int HW4, int Dp, int Cg) {
  constexpr int KC = 256;
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) char at_smem[];
  char* kimg = at_smem;
  char* vimg = at_smem + KC * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int q = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const int h = lane >> 5;
  const long long qrow = (long long)b * HW + q;
  const at_bf16x8 qf0 = __builtin_bit_cast(at_bf16x8, at_gfrag(theta, qrow, Dp, 8 * h, Dp));
  const at_bf16x8 qf1 = __builtin_bit_cast(at_bf16x8, at_gfrag(theta, qrow, Dp, 16 + 8 * h, Dp));
  float m = -3.0e38f;          // running reference maximum of this query (natural-log units), equal in both lane halves
  float m2 = m * LOG2E;
  float l = 0.f;
  at_f32x16 o[NCG];
#pragma unroll
  for (int cg = 0; cg < NCG; cg++)
#pragma unroll
    for (int r = 0; r < 16; r++) o[cg][r] = 0.f;
  for (int k0 = 0; k0 < HW4; k0 += KC) {
    __syncthreads();
    at_stage<4>(kimg, phi + ((long long)b * HW4 + k0) * Dp, KC, Dp, 0, Dp, wave, lane);
#pragma unroll
    for (int cg = 0; cg < NCG; cg++) at_stage<4>(vimg + cg * KC * 64, g + ((long long)b * HW4 + k0) * Cg, KC, Cg, cg * 32, Cg, wave, lane);
    __syncthreads();
#pragma unroll AT_UNROLL
    for (int kc = 0; kc < KC / 32; kc++) {
      at_f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; r++) s[r] = 0.f;
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(kimg, kc, 0, lane), qf0, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(kimg, kc, 1, lane), qf1, s, 0, 0, 0);
      float bm = s[0];
#pragma unroll
      for (int r = 1; r < 16; r++) bm = fmaxf(bm, s[r]);
      bm = at_half_max(bm);
      if (bm > m + AT_THR) {                                      // (the first block always: m starts at -3e38)
        const float alpha = __builtin_amdgcn_exp2f((m - bm) * LOG2E);
        l *= alpha;
#pragma unroll
        for (int cg = 0; cg < NCG; cg++)
#pragma unroll
          for (int r = 0; r < 16; r++) o[cg][r] *= alpha;
        m = bm;
        m2 = m * LOG2E;
      }
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; r++) { p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], LOG2E, -m2)); l += p[r]; }
      u32x4 pa, pb;                                               // regs 0-7 / 8-15 = contraction slots of the two k-steps
#pragma unroll
      for (int i = 0; i < 4; i++) { pa[i] = pack2bf(p[2 * i], p[2 * i + 1]); pb[i] = pack2bf(p[8 + 2 * i], p[8 + 2 * i + 1]); }
      const at_bf16x8 pfa = __builtin_bit_cast(at_bf16x8, pa), pfb = __builtin_bit_cast(at_bf16x8, pb);
#pragma unroll
      for (int cg = 0; cg < NCG; cg++) {
        const char* vi = vimg + cg * KC * 64;
        o[cg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_vfrag(vi, kc * 32 + 4 * h, lane), pfa, o[cg], 0, 0, 0);
        o[cg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_vfrag(vi, kc * 32 + 16 + 4 * h, lane), pfb, o[cg], 0, 0, 0);
      }
    }
  }
  l = at_half_sum(l);
  // (the MFMA contracts over the key slots of BOTH lane halves, so o[] is already complete for the channels this lane holds; only the
  // row sum is per half and needs the exchange -- the reference maximum is common to both halves by construction)
  const float inv = 1.f / l;
  if (h == 0) lse[qrow] = m + __logf(l);
  bf16_t* orow = O + qrow * Cg;
#pragma unroll
  for (int cg = 0; cg < NCG; cg++)
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
      const int c0 = cg * 32 + 8 * g4 + 4 * h;
      if (c0 < Cg) {
        const f32x4 f = {o[cg][4 * g4 + 0] * inv, o[cg][4 * g4 + 1] * inv, o[cg][4 * g4 + 2] * inv, o[cg][4 * g4 + 3] * inv};
        u32x2 v = {pack2bf(f[0], f[1]), pack2bf(f[2], f[3])};
        *(u32x2*)(orow + c0) = v;
        if (O32) *(f32x4*)(O32 + qrow * Cg + c0) = f;             // unrounded copy for the backward's delta_q = dO_q . O_q
      }
    }
}

// grid (HW / 128, B), 256 threads. Keys in chunks of KC = 256: LDS = (1 + NCG) * 16 KiB.
template <int NCG> __global__ __launch_bounds__(256) void k_attn_ds_bwd(const bf16_t* theta, const bf16_t* phi, const bf16_t* g, const bf16_t* dO, const float* lse, bf16_t* dS, int HW, int HW4, int Dp, int Cg) {
  constexpr int KC = 256;
  extern __shared__ __attribute__((aligned(16))) char at_smem[];
  char* kimg = at_smem;
  char* vimg = at_smem + KC * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int q = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const int h = lane >> 5;
  const long long qrow = (long long)b * HW + q;
  const at_bf16x8 qf0 = __builtin_bit_cast(at_bf16x8, at_gfrag(theta, qrow, Dp, 8 * h, Dp));
  const at_bf16x8 qf1 = __builtin_bit_cast(at_bf16x8, at_gfrag(theta, qrow, Dp, 16 + 8 * h, Dp));
  at_bf16x8 df[NCG][2];
#pragma unroll
  for (int cg = 0; cg < NCG; cg++)
#pragma unroll
    for (int t = 0; t < 2; t++) df[cg][t] = __builtin_bit_cast(at_bf16x8, at_gfrag(dO, qrow, Cg, cg * 32 + 16 * t + 8 * h, Cg));
  float delta = 0.f;
  const float ls = lse[qrow];
  bf16_t* drow = dS + qrow * HW4 + 16 * h;
  for (int pass = 0; pass < 2; pass++) {
  for (int k0 = 0; k0 < HW4; k0 += KC) {
    __syncthreads();                                           // previous chunk fully consumed
    at_stage<4>(kimg, phi + ((long long)b * HW4 + k0) * Dp, KC, Dp, 0, Dp, wave, lane);
#pragma unroll
    for (int cg = 0; cg < NCG; cg++) at_stage<4>(vimg + cg * KC * 64, g + ((long long)b * HW4 + k0) * Cg, KC, Cg, cg * 32, Cg, wave, lane);
    __syncthreads();
    for (int kb = 0; kb < KC / 32; kb++) {
      at_f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(kimg, kb, 0, lane), qf0, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(kimg, kb, 1, lane), qf1, s, 0, 0, 0);
#pragma unroll
      for (int cg = 0; cg < NCG; cg++)
#pragma unroll
        for (int t = 0; t < 2; t++)
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(vimg + cg * KC * 64, kb, t, lane), df[cg][t], dp, 0, 0, 0);
      if (pass == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) delta += __expf(s[r] - ls) * dp[r];
      } else {
        float o[16];
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] = __expf(s[r] - ls) * (dp[r] - delta);
        at_store16(drow + k0 + kb * 32, o, h);
      }
    }
  }
  if (pass == 0) delta = at_half_sum(delta);
  }
}


// ---- fused backward, query side: delta, dS (registers only) and dtheta = dS phi ------------------------------------------------------------
// k_attn_ds_bwd with the product that consumed dS moved inside: dtheta^T[d][q] += phi^T[d][k] dS[k][q] per 32-key block, dS going from the
// score accumulators into the MFMA B operand (same key order trick as the fused forward); delta_q is also written out for the key side.
// With the forward output O at hand, delta_q = sum_k P_qk dP_qk = sum_c dO_qc O_qc (the row identity flash attention uses) is a dot product of
// two rows the lane pair already touches, and the first of the two key passes (16 v_exp + 6 MFMAs per block just for delta) disappears.
// O comes as the UNROUNDED fp32 copy the forward keeps for this purpose: with the bf16 output the error of dtheta against fp64 grew from
// ~2e-2 to 4.4e-2 on the near-uniform softmax of tests/test_kernels_gpu.py::test_attention_core (session F), because dS = P (dP - delta)
// subtracts nearly equal numbers there.
template <int NCG> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(AT_WAVES(NCG), AT_WAVES(NCG)))) void k_attn_bwd_q(const bf16_t* theta, const bf16_t* phi, const bf16_t* g, const bf16_t* dO, const float* Oin,
                                                                        const float* lse, float* delta_out, bf16_t* dtheta, int HW, int HW4, int Dp, int Cg) {
  constexpr int KC = 256;
  extern __shared__ __attribute__((aligned(16))) char at_smem[];
  char* kimg = at_smem;
  char* vimg = at_smem + KC * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int q = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const int h = lane >> 5;
  const long long qrow = (long long)b * HW + q;
  const at_bf16x8 qf0 = __builtin_bit_cast(at_bf16x8, at_gfrag(theta, qrow, Dp, 8 * h, Dp));
  const at_bf16x8 qf1 = __builtin_bit_cast(at_bf16x8, at_gfrag(theta, qrow, Dp, 16 + 8 * h, Dp));
  at_bf16x8 df[NCG][2];
#pragma unroll
  for (int cg = 0; cg < NCG; cg++)
#pragma unroll
    for (int t = 0; t < 2; t++) df[cg][t] = __builtin_bit_cast(at_bf16x8, at_gfrag(dO, qrow, Cg, cg * 32 + 16 * t + 8 * h, Cg));
  float delta = 0.f;
  const float ls2 = lse[qrow] * 1.4426950408889634f;
  if (Oin) {
#pragma unroll
    for (int cg = 0; cg < NCG; cg++)
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int c0 = cg * 32 + 16 * t + 8 * h;
        if (c0 < Cg) {                                               // Cg % 8 == 0: groups of 8 channels are whole
          const f32x4 oa = *(const f32x4*)(Oin + qrow * Cg + c0), ob = *(const f32x4*)(Oin + qrow * Cg + c0 + 4);
          const u32x4 dv = __builtin_bit_cast(u32x4, df[cg][t]);
          delta += oa[0] * __uint_as_float(dv[0] << 16) + oa[1] * __uint_as_float(dv[0] & 0xffff0000u) + oa[2] * __uint_as_float(dv[1] << 16) + oa[3] * __uint_as_float(dv[1] & 0xffff0000u)
                 + ob[0] * __uint_as_float(dv[2] << 16) + ob[1] * __uint_as_float(dv[2] & 0xffff0000u) + ob[2] * __uint_as_float(dv[3] << 16) + ob[3] * __uint_as_float(dv[3] & 0xffff0000u);
        }
      }
    delta = at_half_sum(delta);
  }
  at_f32x16 dth;
#pragma unroll
  for (int r = 0; r < 16; r++) dth[r] = 0.f;
  for (int pass = Oin ? 1 : 0; pass < 2; pass++) {
    for (int k0 = 0; k0 < HW4; k0 += KC) {
      __syncthreads();
      at_stage<4>(kimg, phi + ((long long)b * HW4 + k0) * Dp, KC, Dp, 0, Dp, wave, lane);
#pragma unroll
      for (int cg = 0; cg < NCG; cg++) at_stage<4>(vimg + cg * KC * 64, g + ((long long)b * HW4 + k0) * Cg, KC, Cg, cg * 32, Cg, wave, lane);
      __syncthreads();
#pragma unroll AT_UNROLL
      for (int kb = 0; kb < KC / 32; kb++) {
        at_f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(kimg, kb, 0, lane), qf0, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(kimg, kb, 1, lane), qf1, s, 0, 0, 0);
#pragma unroll
        for (int cg = 0; cg < NCG; cg++)
#pragma unroll
          for (int t = 0; t < 2; t++)
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(vimg + cg * KC * 64, kb, t, lane), df[cg][t], dp, 0, 0, 0);
        if (pass == 0) {
#pragma unroll
          for (int r = 0; r < 16; r++) delta += __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], 1.4426950408889634f, -ls2)) * dp[r];
        } else {
          float o[16];
#pragma unroll
          for (int r = 0; r < 16; r++) o[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], 1.4426950408889634f, -ls2)) * (dp[r] - delta);
          u32x4 da, db;
#pragma unroll
          for (int i = 0; i < 4; i++) { da[i] = pack2bf(o[2 * i], o[2 * i + 1]); db[i] = pack2bf(o[8 + 2 * i], o[8 + 2 * i + 1]); }
          dth = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_vfrag(kimg, kb * 32 + 4 * h, lane), __builtin_bit_cast(at_bf16x8, da), dth, 0, 0, 0);
          dth = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_vfrag(kimg, kb * 32 + 16 + 4 * h, lane), __builtin_bit_cast(at_bf16x8, db), dth, 0, 0, 0);
        }
      }
    }
    if (pass == 0) delta = at_half_sum(delta);
  }
  if (h == 0) delta_out[qrow] = delta;
  bf16_t* trow = dtheta + qrow * Dp;
#pragma unroll
  for (int g4 = 0; g4 < 4; g4++) {
    const int c0 = 8 * g4 + 4 * h;
    if (c0 < Dp) {
      u32x2 v = {pack2bf(dth[4 * g4 + 0], dth[4 * g4 + 1]), pack2bf(dth[4 * g4 + 2], dth[4 * g4 + 3])};
      *(u32x2*)(trow + c0) = v;
    }
  }
}

// ---- fused backward, key side: dphi = dS^T theta, dg = P^T dO with P and dS recomputed per (32 queries x 32 keys) block ----------------------
// One workgroup = 4 waves = 128 keys of one image; a lane owns ONE key (l & 31) and 16 of the 32 queries of a block (the transposed
// orientation of the query-side kernels: scores = theta (rows) x phi (columns)). Query-side operands (theta, dO; lse, delta) are staged by
// LDS-DMA in chunks of 256 queries; they feed the score / dP products as k-contiguous fragments and the two accumulating products as
// transposed fragments of the same images. No P and no dS ever exist in HBM.
template <int NCG> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(AT_WAVES(NCG), AT_WAVES(NCG)))) void k_attn_bwd_k(const bf16_t* theta, const bf16_t* phi, const bf16_t* g, const bf16_t* dO, const float* lse,
                                                                        const float* delta, bf16_t* dphi, bf16_t* dg, int HW, int HW4, int Dp, int Cg) {
  constexpr int QC = 256;
  extern __shared__ __attribute__((aligned(16))) char at_smem[];
  char* timg = at_smem;
  char* oimg = at_smem + QC * 64;
  float* st = (float*)(at_smem + (1 + NCG) * QC * 64);
  float* dl = st + QC;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int key = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const int h = lane >> 5;
  const long long krow = (long long)b * HW4 + key;
  const at_bf16x8 kf0 = __builtin_bit_cast(at_bf16x8, at_gfrag(phi, krow, Dp, 8 * h, Dp));
  const at_bf16x8 kf1 = __builtin_bit_cast(at_bf16x8, at_gfrag(phi, krow, Dp, 16 + 8 * h, Dp));
  at_bf16x8 gf[NCG][2];
#pragma unroll
  for (int cg = 0; cg < NCG; cg++)
#pragma unroll
    for (int t = 0; t < 2; t++) gf[cg][t] = __builtin_bit_cast(at_bf16x8, at_gfrag(g, krow, Cg, cg * 32 + 16 * t + 8 * h, Cg));
  at_f32x16 dph, dgt[NCG];
#pragma unroll
  for (int r = 0; r < 16; r++) dph[r] = 0.f;
#pragma unroll
  for (int cg = 0; cg < NCG; cg++)
#pragma unroll
    for (int r = 0; r < 16; r++) dgt[cg][r] = 0.f;
  for (int q0 = 0; q0 < HW; q0 += QC) {
    __syncthreads();
    at_stage<4>(timg, theta + ((long long)b * HW + q0) * Dp, QC, Dp, 0, Dp, wave, lane);
#pragma unroll
    for (int cg = 0; cg < NCG; cg++) at_stage<4>(oimg + cg * QC * 64, dO + ((long long)b * HW + q0) * Cg, QC, Cg, cg * 32, Cg, wave, lane);
    if (tid < QC) { st[tid] = lse[(long long)b * HW + q0 + tid] * 1.4426950408889634f; dl[tid] = delta[(long long)b * HW + q0 + tid]; }
    __syncthreads();
#pragma unroll AT_UNROLL
    for (int qb = 0; qb < QC / 32; qb++) {
      at_f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(timg, qb, 0, lane), kf0, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(timg, qb, 1, lane), kf1, s, 0, 0, 0);
#pragma unroll
      for (int cg = 0; cg < NCG; cg++)
#pragma unroll
        for (int t = 0; t < 2; t++)
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(oimg + cg * QC * 64, qb, t, lane), gf[cg][t], dp, 0, 0, 0);
      float pr[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int qi = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;       // the query of accumulator register r
        pr[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], 1.4426950408889634f, -st[qi]));
        ds[r] = pr[r] * (dp[r] - dl[qi]);
      }
      u32x4 pa, pb, da, db;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        pa[i] = pack2bf(pr[2 * i], pr[2 * i + 1]); pb[i] = pack2bf(pr[8 + 2 * i], pr[8 + 2 * i + 1]);
        da[i] = pack2bf(ds[2 * i], ds[2 * i + 1]); db[i] = pack2bf(ds[8 + 2 * i], ds[8 + 2 * i + 1]);
      }
#pragma unroll
      for (int cg = 0; cg < NCG; cg++) {
        const char* oi = oimg + cg * QC * 64;
        dgt[cg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_vfrag(oi, qb * 32 + 4 * h, lane), __builtin_bit_cast(at_bf16x8, pa), dgt[cg], 0, 0, 0);
        dgt[cg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_vfrag(oi, qb * 32 + 16 + 4 * h, lane), __builtin_bit_cast(at_bf16x8, pb), dgt[cg], 0, 0, 0);
      }
      dph = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_vfrag(timg, qb * 32 + 4 * h, lane), __builtin_bit_cast(at_bf16x8, da), dph, 0, 0, 0);
      dph = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_vfrag(timg, qb * 32 + 16 + 4 * h, lane), __builtin_bit_cast(at_bf16x8, db), dph, 0, 0, 0);
    }
  }
  bf16_t* prow = dphi + krow * Dp;
  bf16_t* grow = dg + krow * Cg;
#pragma unroll
  for (int g4 = 0; g4 < 4; g4++) {
    const int c0 = 8 * g4 + 4 * h;
    if (c0 < Dp) {
      u32x2 v = {pack2bf(dph[4 * g4 + 0], dph[4 * g4 + 1]), pack2bf(dph[4 * g4 + 2], dph[4 * g4 + 3])};
      *(u32x2*)(prow + c0) = v;
    }
  }
#pragma unroll
  for (int cg = 0; cg < NCG; cg++)
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
      const int c0 = cg * 32 + 8 * g4 + 4 * h;
      if (c0 < Cg) {
        u32x2 v = {pack2bf(dgt[cg][4 * g4 + 0], dgt[cg][4 * g4 + 1]), pack2bf(dgt[cg][4 * g4 + 2], dgt[cg][4 * g4 + 3])};
        *(u32x2*)(grow + c0) = v;
      }
    }
}

static bool at_ok(int B, int HW, int HW4, int Dp) {
  return B > 0 && B <= 65535 && HW % 128 == 0 && HW4 % 256 == 0 && Dp % 8 == 0 && Dp >= 8 && Dp <= 32;
}
extern "C" int sg_attn_fused_ok(int B, int HW, int HW4, int Dp, int Cg) {
  return (at_ok(B, HW, HW4, Dp) && HW4 * 64 <= 128 * 1024 && Cg % 8 == 0 && Cg <= 128) ? 1 : 0;
}
extern "C" int sg_attn_fwd_fused_ok(int B, int HW, int HW4, int Dp, int Cg) {
  const int ncg = (Cg + 31) / 32;
  return (at_ok(B, HW, HW4, Dp) && Cg % 8 == 0 && Cg >= 8 && Cg <= 128 && HW4 * 64 + ncg * 16384 <= 160 * 1024) ? 1 : 0;
}
// the P == NULL form of sg_attn_fwd_fused (k_attn_fwd_flash) streams keys AND values in 256-key chunks: (1 + ncg) * 16 KiB of LDS whatever HW4 is
// (BigGAN-deep-256's discriminator attends over 128 x 128 = 16384 queries x 4096 keys: reference src/models/big_resnet_deep_legacy.py:80-95)
extern "C" int sg_attn_fwd_flash_ok(int B, int HW, int HW4, int Dp, int Cg) {
  return (at_ok(B, HW, HW4, Dp) && Cg % 8 == 0 && Cg >= 8 && Cg <= 128) ? 1 : 0;
}
// O = softmax(theta phi^T) g in one launch; P == NULL: the probabilities are not stored (no backward will ask for them)
extern "C" int sg_attn_fwd_fused(const void* theta, const void* phi, const void* g, void* P, float* lse, void* O, float* O32, int B, int HW, int HW4, int Dp, int Cg, sg_stream_t s) {
  SG_CHECK(theta && phi && g && lse && O, "sg_attn_fwd_fused: null");
  SG_CHECK(!(P && O32), "sg_attn_fwd_fused: the fp32 copy of O belongs to the path that does not store P");
  static const bool two_pass = [] { const char* e = getenv("SG_ATTN_FLASH"); return e && e[0] == '0'; }();    // A/B switch: the first fused forward
  const bool flash = !P && !(two_pass && !O32);
  SG_CHECK((flash ? sg_attn_fwd_flash_ok(B, HW, HW4, Dp, Cg) : sg_attn_fwd_fused_ok(B, HW, HW4, Dp, Cg)) == 1, "sg_attn_fwd_fused: unsupported shape");
  const int ncg = (Cg + 31) / 32;
  const int lds = HW4 * 64 + ncg * 16384;
  SgProfScope prof((hipStream_t)s, (double)B * HW * ((P ? (double)HW4 * 2.0 : 0.0) + (Dp + Cg) * 2.0 + 4.0) + (double)B * HW4 * (Dp + Cg) * 2.0, 5);
  const dim3 grid(HW / 128, B), blk(256);
  hipStream_t st = (hipStream_t)s;
#define ATF_LAUNCH(N, SP)                                                                                                                 \
  {                                                                                                                                        \
    static bool done = false;                                                                                                              \
    if (!done) { SG_CHECK(hipFuncSetAttribute((const void*)k_attn_fwd_fused<N, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess, "sg_attn_fwd_fused: LDS attribute"); done = true; } \
    hipLaunchKernelGGL((k_attn_fwd_fused<N, SP>), grid, blk, lds, st, (const bf16_t*)theta, (const bf16_t*)phi, (const bf16_t*)g, (bf16_t*)P, lse, (bf16_t*)O, HW, HW4, Dp, Cg); \
  }
#define ATL_LAUNCH(N)                                                                                                                     \
  {                                                                                                                                        \
    static bool done = false;                                                                                                              \
    if (!done) { SG_CHECK(hipFuncSetAttribute((const void*)k_attn_fwd_flash<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess, "sg_attn_fwd_fused: LDS attribute"); done = true; } \
    hipLaunchKernelGGL((k_attn_fwd_flash<N>), grid, blk, (1 + N) * 16384, st, (const bf16_t*)theta, (const bf16_t*)phi, (const bf16_t*)g, lse, (bf16_t*)O, O32, HW, HW4, Dp, Cg); \
  }
  if (P) { if (ncg == 1) ATF_LAUNCH(1, true) else if (ncg == 2) ATF_LAUNCH(2, true) else if (ncg == 3) ATF_LAUNCH(3, true) else ATF_LAUNCH(4, true) }
  else if (!flash) { if (ncg == 1) ATF_LAUNCH(1, false) else if (ncg == 2) ATF_LAUNCH(2, false) else if (ncg == 3) ATF_LAUNCH(3, false) else ATF_LAUNCH(4, false) }
  else { if (ncg == 1) ATL_LAUNCH(1) else if (ncg == 2) ATL_LAUNCH(2) else if (ncg == 3) ATL_LAUNCH(3) else ATL_LAUNCH(4) }
#undef ATL_LAUNCH
#undef ATF_LAUNCH
  SG_LAUNCH_CHECK();
  return 0;
}
// fused backward (no P, no dS in HBM): dtheta [B][HW][Dp], dphi [B][HW4][Dp], dg [B][HW4][Cg] (pooled keys / values), delta = fp32 scratch [B][HW]
extern "C" int sg_attn_bwd_fused_ok(int B, int HW, int HW4, int Dp, int Cg) {
  return (at_ok(B, HW, HW4, Dp) && HW % 256 == 0 && HW4 % 128 == 0 && Cg % 8 == 0 && Cg >= 8 && Cg <= 128) ? 1 : 0;
}
extern "C" int sg_attn_bwd_fused(const void* theta, const void* phi, const void* g, const void* dO, const float* O32, const float* lse, float* delta, void* dtheta, void* dphi,
                                 void* dg, int B, int HW, int HW4, int Dp, int Cg, sg_stream_t s) {
  SG_CHECK(theta && phi && g && dO && lse && delta && dtheta && dphi && dg, "sg_attn_bwd_fused: null");
  SG_CHECK(sg_attn_bwd_fused_ok(B, HW, HW4, Dp, Cg) == 1, "sg_attn_bwd_fused: unsupported shape");
  const int ncg = (Cg + 31) / 32;
  const int lds_q = (1 + ncg) * 256 * 64, lds_k = (1 + ncg) * 256 * 64 + 2 * 256 * 4;
  SgProfScope prof((hipStream_t)s, 2.0 * ((double)B * HW * ((Dp + Cg) * 2.0 + 8.0) + (double)B * HW4 * (Dp + Cg) * 2.0) + (double)B * (HW + HW4) * (Dp * 2.0) + (double)B * HW4 * Cg * 2.0, 5);
  hipStream_t st = (hipStream_t)s;
#define ATB_LAUNCH(N)                                                                                                                     \
  {                                                                                                                                        \
    static bool done = false;                                                                                                              \
    if (!done) {                                                                                                                           \
      SG_CHECK(hipFuncSetAttribute((const void*)k_attn_bwd_q<N>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_q) == hipSuccess, "sg_attn_bwd_fused: LDS attribute"); \
      SG_CHECK(hipFuncSetAttribute((const void*)k_attn_bwd_k<N>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_k) == hipSuccess, "sg_attn_bwd_fused: LDS attribute"); \
      done = true;                                                                                                                         \
    }                                                                                                                                      \
    hipLaunchKernelGGL(k_attn_bwd_q<N>, dim3(HW / 128, B), dim3(256), lds_q, st, (const bf16_t*)theta, (const bf16_t*)phi, (const bf16_t*)g, (const bf16_t*)dO, O32, lse, delta, (bf16_t*)dtheta, HW, HW4, Dp, Cg); \
    hipLaunchKernelGGL(k_attn_bwd_k<N>, dim3(HW4 / 128, B), dim3(256), lds_k, st, (const bf16_t*)theta, (const bf16_t*)phi, (const bf16_t*)g, (const bf16_t*)dO, lse, (const float*)delta, (bf16_t*)dphi, (bf16_t*)dg, HW, HW4, Dp, Cg); \
  }
  if (ncg == 1) ATB_LAUNCH(1) else if (ncg == 2) ATB_LAUNCH(2) else if (ncg == 3) ATB_LAUNCH(3) else ATB_LAUNCH(4)
#undef ATB_LAUNCH
  SG_LAUNCH_CHECK();
  return 0;
}
extern "C" int sg_attn_probs_fwd(const void* theta, const void* phi, void* P, float* lse, int B, int HW, int HW4, int Dp, sg_stream_t s) {
  SG_CHECK(theta && phi && P && lse, "sg_attn_probs_fwd: null");
  SG_CHECK(at_ok(B, HW, HW4, Dp) && HW4 * 64 <= 128 * 1024, "sg_attn_probs_fwd: unsupported shape");
  const int lds = HW4 * 64;
  SgProfScope prof((hipStream_t)s, (double)B * HW * ((double)HW4 * 2.0 + Dp * 2.0 + 4.0) + (double)B * HW4 * Dp * 2.0, 5);   // P written once (bf16), theta / phi read, lse
  static int attr = 0;
  if (attr < lds) {
    SG_CHECK(hipFuncSetAttribute((const void*)k_attn_probs_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess, "sg_attn_probs_fwd: LDS attribute");
    attr = 128 * 1024;
  }
  hipLaunchKernelGGL(k_attn_probs_fwd, dim3(HW / 128, B), dim3(256), lds, (hipStream_t)s, (const bf16_t*)theta, (const bf16_t*)phi, (bf16_t*)P, lse, HW, HW4, Dp);
  SG_LAUNCH_CHECK();
  return 0;
}
extern "C" int sg_attn_ds_bwd(const void* theta, const void* phi, const void* g, const void* dO, const float* lse, void* dS, int B, int HW, int HW4, int Dp, int Cg, sg_stream_t s) {
  SG_CHECK(theta && phi && g && dO && lse && dS, "sg_attn_ds_bwd: null");
  SG_CHECK(at_ok(B, HW, HW4, Dp) && Cg % 8 == 0 && Cg >= 8 && Cg <= 128, "sg_attn_ds_bwd: unsupported shape");
  const int ncg = (Cg + 31) / 32;
  SgProfScope prof((hipStream_t)s, (double)B * HW * ((double)HW4 * 2.0 + (Dp + Cg) * 2.0 + 4.0) + (double)B * HW4 * (Dp + Cg) * 2.0, 5);   // dS written once
  const int lds = (1 + ncg) * 256 * 64;
  const dim3 grid(HW / 128, B), blk(256);
  hipStream_t st = (hipStream_t)s;
#define AT_LAUNCH(N)                                                                                                                       \
  {                                                                                                                                        \
    static bool done = false;                                                                                                              \
    if (!done) { SG_CHECK(hipFuncSetAttribute((const void*)k_attn_ds_bwd<N>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess, "sg_attn_ds_bwd: LDS attribute"); done = true; } \
    hipLaunchKernelGGL(k_attn_ds_bwd<N>, grid, blk, lds, st, (const bf16_t*)theta, (const bf16_t*)phi, (const bf16_t*)g, (const bf16_t*)dO, lse, (bf16_t*)dS, HW, HW4, Dp, Cg); \
  }
  if (ncg == 1) AT_LAUNCH(1) else if (ncg == 2) AT_LAUNCH(2) else if (ncg == 3) AT_LAUNCH(3) else AT_LAUNCH(4)
#undef AT_LAUNCH
  SG_LAUNCH_CHECK();
  return 0;
}
