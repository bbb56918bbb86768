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
__device__ __forceinline__ float at_half_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float at_half_sum(float v) { return v + __shfl_xor(v, 32, 64); }
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

static bool at_ok(int B, int HW, int HW4, int Dp) {
  return B > 0 && B <= 65535 && HW % 128 == 0 && HW4 % 256 == 0 && Dp % 8 == 0 && Dp >= 8 && Dp <= 32;
}
extern "C" int sg_attn_fused_ok(int B, int HW, int HW4, int Dp, int Cg) {
  return (at_ok(B, HW, HW4, Dp) && HW4 * 64 <= 128 * 1024 && Cg % 8 == 0 && Cg <= 128) ? 1 : 0;
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
