// Common device helpers for the gfx950 (MI355X / CDNA4) StudioGAN hot-path kernels.
// Wave = 64 lanes everywhere in this tree; nothing here is written for 32-wide warps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 storage (torch.bfloat16 bit pattern)
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// `s_setprio 1` around the MFMA clusters of the lean weight-gradient kernels (wgrad_ql.h, wgrad_v3l.h): their workgroups share a CU in different phases (one in
// its MFMA cluster while another issues its LDS-DMA or sits at its barrier), the structure cdna_hip_programming.md T5 prices the hint for. Measured in round 5
// (same box, profiles/r05_variant_ab_layer_tables_b.txt): -1.0 % / -1.7 % on the two weight-gradient layer tables, -0.7 ms on the C3 step: always on.
#define SG_PRIO_UP() __builtin_amdgcn_s_setprio(1)
#define SG_PRIO_DOWN() __builtin_amdgcn_s_setprio(0)

#define SG_DTYPE_F32 0
#define SG_DTYPE_BF16 1

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, same rounding as torch's float->bfloat16
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two floats -> packed bf16 pair with the gfx950 hardware conversion (v_cvt_pk_bf16_f32, round-to-nearest-even: the same value as
// f2bf for every finite input; 1 instruction per pair instead of ~12)
typedef __bf16 sg_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float sg_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float a, float b) {
  sg_f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sg_bf16x2_t));
}

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

// element traits: VEC = elements per 16-byte vector, BK = GEMM k-tile in elements (always 64 bytes)
template <typename T> struct ET;
template <> struct ET<float> { static constexpr int VEC = 4, BK = 16; };
template <> struct ET<bf16_t> { static constexpr int VEC = 8, BK = 32; };

__device__ __forceinline__ u32x4 zero16() { u32x4 z = {0u, 0u, 0u, 0u}; return z; }

// ReLU on a packed 16-byte vector
// 16-byte vector <-> fp32 lanes (4 floats or 8 bf16)
template <typename T> __device__ __forceinline__ void unpack16(u32x4 v, float* o);
template <> __device__ __forceinline__ void unpack16<float>(u32x4 v, float* o) {
#pragma unroll
  for (int i = 0; i < 4; i++) o[i] = __uint_as_float(v[i]);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(u32x4 v, float* o) {
#pragma unroll
  for (int i = 0; i < 4; i++) { o[2 * i] = __uint_as_float(v[i] << 16); o[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ u32x4 pack16(const float* o);
template <> __device__ __forceinline__ u32x4 pack16<float>(const float* o) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; i++) v[i] = __float_as_uint(o[i]);
  return v;
}
template <> __device__ __forceinline__ u32x4 pack16<bf16_t>(const float* o) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; i++) v[i] = (uint32_t)f2bf(o[2 * i]) | ((uint32_t)f2bf(o[2 * i + 1]) << 16);
  return v;
}


template <typename T> __device__ __forceinline__ u32x4 relu16(u32x4 v);
template <> __device__ __forceinline__ u32x4 relu16<float>(u32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; i++) v[i] = (v[i] & 0x80000000u) ? 0u : v[i];
  return v;
}
template <> __device__ __forceinline__ u32x4 relu16<bf16_t>(u32x4 v) {
  // bf16 ReLU == signed 16-bit max with 0 (sign bit set -> 0, everything else unchanged): one v_pk_max_i16 per dword. The
  // mask/select form it replaces compiled to ~6 VALU per dword, ~190 VALU per 24-MFMA tap of the halo kernel.
  typedef short s16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t x = v[i];       // (bit_cast straight from the vector element miscompiles with hipcc 7.2: go through a scalar)
    s16x2_t a = __builtin_bit_cast(s16x2_t, x);
    const s16x2_t z = {0, 0};
    a = __builtin_elementwise_max(a, z);
    v[i] = __builtin_bit_cast(uint32_t, a);
  }
  return v;
}

// sum over the 4 lanes of a quad (lanes 4q..4q+3), result in every lane: two v_add_f32_dpp (quad_perm), no LDS crossbar round trip.
// Same values as v += shfl_xor(v,1); v += shfl_xor(v,2) (the additions commute), which compiled to two dependent ds_bpermute.
__device__ __forceinline__ float quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  return v;
}

// wave-level sum (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-level sum for blockDim.x == 256 (4 waves); result valid in all threads
__device__ __forceinline__ float block_sum_256(float v, float* sm /* >= 4 floats */) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ double block_sum_256_d(double v, double* sm) {
  v = wave_sum_d(v);
  int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ float block_max_256(float v, float* sm) {
  v = wave_max(v);
  int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// error reporting shared by every C-ABI entry point (capi.cpp owns the storage)
extern "C" void sg_set_error(const char* msg);
extern int g_sg_f32_mode;      // fp32 arithmetic of the generic engine (conv.hip sg_set_f32_mode): 0 exact fp32 MFMA, 3 bf16x3 split
extern "C" int sg_prof_begin(hipStream_t st, double flops, int kind);
extern "C" int sg_prof_begin_q(hipStream_t st, double flops, int kind);
extern "C" void sg_prof_end(hipStream_t st, int slot);
extern "C" void sg_prof_set_executed(int slot, double flops);
extern "C" void sg_prof_tag(int slot, int engine, double alg_bytes);
// kernel families of the convolution engine, as the launch profiler reports them (sg_prof_collect_tags; names in bench.py ENGINES)
enum { SG_ENG_OTHER = 0, SG_ENG_CONV_SK, SG_ENG_CONV_RS, SG_ENG_CONV_V4, SG_ENG_CONV_V3, SG_ENG_CONV_V2, SG_ENG_CONV_GEMM, SG_ENG_CONV_V4_SKIP, SG_ENG_CONV_Q,
       SG_ENG_CONV_Q_SKIP, SG_ENG_WGRAD_SK, SG_ENG_WGRAD_V3, SG_ENG_WGRAD_V2, SG_ENG_WGRAD_GEMM, SG_ENG_WGRAD_Q, SG_ENG_COUNT };
// scope guard around an entry point's launches for the in-library profiler (capi.hip): kinds 0-2 = MFMA contraction engine (work in
// FLOP), kinds 3-6 = HBM-bound families (work in algorithmic BYTES): 3 spectral norm, 4 batch norm, 5 attention scores, 6 Adam / EMA
struct SgProfScope {
  int slot; hipStream_t st;
  SgProfScope(hipStream_t s, double work, int kind) : slot(sg_prof_begin(s, work, kind)), st(s) {}
  ~SgProfScope() { sg_prof_end(st, slot); }
};
#define SG_CHECK(cond, msg)                                   \
  do {                                                        \
    if (!(cond)) { sg_set_error(msg); return -1; }            \
  } while (0)
#define SG_LAUNCH_CHECK()                                     \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) { sg_set_error(hipGetErrorString(e__)); return -2; } \
  } while (0)
