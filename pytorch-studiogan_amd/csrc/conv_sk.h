// conv_sk.h -- "small-K" streaming forward / data-gradient kernel: 1x1 convolutions with <= 192 input channels and the 3x3 RGB
// stem (8 padded channels, K = 72), bf16.
//
// These layers are HBM-bound (a 96 -> 192 1x1 at 64^2 moves 0.3 GB for 39 GFLOP), but the tile kernels (conv_v2.h) ran them at
// 1.7-2.3 TB/s: one 160 KB workgroup per CU, and per tile a prologue (pixel decode, halo masks), one or two k-tiles behind a
// cold DMA, a block-wide staged epilogue and the workgroup hand-over -- all serialised, 14 us per 512-pixel tile of the stem.
// Here nothing is per-tile and nothing is block-wide after start-up:
//   * the WEIGHTS of a cout tile (<= 192 x K) are staged into LDS once per workgroup (row pitch = odd multiple of 16 bytes:
//     conflict-free ds_read_b128) and stay there; the workgroup's waves then run independently of each other (no barrier)
//   * a wave owns row blocks of 32*TJ pixels, strided over the grid. The MFMA B operand of pixel row (lane & 31), k-half
//     (lane >> 5) is 16 contiguous bytes of NHWC memory (1x1: channels 8c..8c+7; stem: the 8 channels of tap c), so a lane
//     fetches its own fragments with buffer_load_dwordx4 straight into registers -- no LDS round trip for the pixel operand.
//     Halo taps, channel tails and rows past the problem are out-of-range offsets: the buffer unit returns zeros for them.
//   * software pipeline per wave: the fragment loads of row block n+1 are issued (unconditionally -- past-the-end blocks are all
//     out-of-range -- so the compiler's vmcnt bookkeeping stays exact) before the MFMAs and the epilogue of block n
//   * epilogue = sg_conv_epilogue's arithmetic (pool, scale, bias, mask | residual, ReLU, bf16 pack) on a wave-private staging
//     tile: 16-byte coalesced pre-load of the mask / residual rows, in-place update, 16-byte coalesced stores
//   * ReLU-on-load is one v_pk_max_i16 per dword against a floor register (0 or -32768): no template split
#pragma once
#include "conv_v2.h"

struct ConvSkParams {
  const bf16_t* x; const bf16_t* w;
  int C, ldx, Hs, Ws;          // source tensor: channels, pixel pitch (elements), spatial size
  int Ho, Wo, wshift, hshift;  // output size (= source x2 with upsample-on-load), powers of two
  int mode3;                   // 1: 3x3 / pad 1 with C == 8 (16-byte chunk c = tap c), 0: 1x1
  int flags;                   // SG_PIX_RELU | SG_PIX_UPSAMPLE | SG_PIX_QUAD
  int I, J, K;
  unsigned xbytes;
  unsigned obytes, sbytes;     // extents of the output tensor and of the mask / residual tensor (< 2^31: bit 31 of an offset = skip)
  int nrb;                     // row blocks of 32 * TJ pixels
};

// TI x 32 couts per accumulator pass, NP passes over the same pixel fragments (cout tile = TI * NP * 32)
template <int TI, int NP, int TJ, int KS, int NW, bool POOL>
__global__ __launch_bounds__(64 * NW) void sg_conv_sk_kernel(ConvSkParams p, Epilogue<bf16_t> epi) {
  constexpr int BI = TI * NP * 32;
  constexpr int NCH = 2 * KS;              // 16-byte k-chunks
  constexpr int WP = KS * 32 + 16;         // weight row pitch in LDS
  constexpr int CP = TI * 64 + 16;         // staging row pitch: the couts of ONE pass (the passes stage and store one after the other)
  constexpr int ROWS = 32 * TJ;
  constexpr int CPR = TI * 4;              // 16-byte chunks per output row of a pass
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wsm = smem;
  float* const sbias = (float*)(smem + BI * WP);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* const stg = smem + BI * WP + BI * 4 + wave * (ROWS * CP);
  const int i0 = blockIdx.y * BI;

  // ---- weights and bias of this cout tile -> LDS (once) -------------------------------------------------------------------------
  for (int idx = tid; idx < BI * NCH; idx += 64 * NW) {
    const int row = idx / NCH, c = idx - row * NCH;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (i0 + row < p.I && c * 8 < p.K) v = *(const u32x4*)(p.w + (long long)(i0 + row) * p.K + c * 8);
    *(u32x4*)(wsm + row * WP + c * 16) = v;
  }
  for (int i = tid; i < BI; i += 64 * NW) sbias[i] = (epi.bias && i0 + i < epi.I) ? epi.bias[i0 + i] : 0.f;
  __syncthreads();

  const auto rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
  const auto rso = __builtin_amdgcn_make_buffer_rsrc(epi.out, 0, (int)p.obytes, 0x00020000);
  const auto rss = __builtin_amdgcn_make_buffer_rsrc((void*)(epi.mask ? (const void*)epi.mask : epi.res), 0, (int)p.sbytes, 0x00020000);
  constexpr int NIT = ((POOL ? ROWS / 4 : ROWS) * CPR + 63) / 64;      // 16-byte pieces of a staged pass tile per lane
  const int frow = lane & 31, fhi = lane >> 5;
  const unsigned ldx2 = 2u * (unsigned)p.ldx;
  const bool up = (p.flags & SG_PIX_UPSAMPLE) != 0, quad = (p.flags & SG_PIX_QUAD) != 0;
  typedef short s16x2_t __attribute__((ext_vector_type(2)));
  const short fl = (p.flags & SG_PIX_RELU) ? (short)0 : (short)-32768;
  const s16x2_t floor2 = {fl, fl};

  // per-lane k-chunk constants: chunk c = 2 ks + fhi. Stem: chunk = tap (tr, ts) = (c / 3, c % 3), displacement (tr-1, ts-1) pixels;
  // 1x1: chunk = channels 8c.., no displacement (tr = ts = 1 keeps the halo test trivially true): one branch-free form for both
  int ctr[KS], cts[KS];
  unsigned cdelta[KS];
  bool cok[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ks++) {
    const int c = 2 * ks + fhi;
    const int tr3 = (c * 11) >> 5;           // c / 3 for c < 32
    const int ts3 = c - 3 * tr3;
    ctr[ks] = p.mode3 ? tr3 : 1;
    cts[ks] = p.mode3 ? ts3 : 1;
    cdelta[ks] = p.mode3 ? (unsigned)((tr3 - 1) * p.Ws + (ts3 - 1)) * ldx2 : (unsigned)c * 16u;
    cok[ks] = p.mode3 ? (c < 9) : (c * 8 < p.C);
  }

  // issue the fragment loads of row block rb (all out of range when rb >= nrb: rows >= J)
  auto fetch = [&](int rb, u32x4 (&q)[TJ][KS]) {
#pragma unroll
    for (int b = 0; b < TJ; b++) {
      const int row = rb * ROWS + b * 32 + frow;
      int n, ho, wo;
      if (quad) {
        const int qd = row >> 2, dy = (row >> 1) & 1, dx = row & 1;
        const int wq = qd & ((p.Wo >> 1) - 1);
        const int t = qd >> (p.wshift - 1);
        const int hq = t & ((p.Ho >> 1) - 1);
        n = t >> (p.hshift - 1);
        ho = 2 * hq + dy; wo = 2 * wq + dx;
      } else {
        wo = row & (p.Wo - 1); const int t = row >> p.wshift; ho = t & (p.Ho - 1); n = t >> p.hshift;
      }
      const bool rok = row < p.J;
      const int hs = up ? (ho >> 1) : ho, ws = up ? (wo >> 1) : wo;
      const unsigned pixoff = ((unsigned)(n * p.Hs + hs) * (unsigned)p.Ws + (unsigned)ws) * ldx2;
      if constexpr (KS == 6) {       // the only sub-step count the stem (K = 72) maps to: tap displacement + halo test per chunk
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
          const bool ok = rok && cok[ks] && (unsigned)(ho + ctr[ks] - 1) < (unsigned)p.Ho && (unsigned)(wo + cts[ks] - 1) < (unsigned)p.Wo;
          unsigned off = pixoff + cdelta[ks];
          off = ok ? off : 0x80000000u;
          q[b][ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)off, 0, 0));
        }
      } else {                       // 1x1 only: chunk c of the pixel's own row (the offsets fold into the instruction's immediate)
        const unsigned base = rok ? pixoff + (unsigned)fhi * 16u : 0x80000000u;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
          const unsigned off = ((2 * ks + fhi) * 8 < p.C) ? base + (unsigned)ks * 32u : 0x80000000u;
          q[b][ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)off, 0, 0));
        }
      }
    }
  };

  float al = epi.alpha;
  if (epi.alpha_ptr) al *= *epi.alpha_ptr;
  constexpr bool pool = POOL;
  const bool relu_out = (epi.flags & SG_EPI_RELU) != 0;
  const bool pre_mask = epi.mask != nullptr, pre_res = epi.res != nullptr;
  const int Jout = pool ? (epi.J >> 2) : epi.J;
  const int ncr = ((epi.I - i0 < BI ? epi.I - i0 : BI) + 7) >> 3;   // 16-byte chunks of an output row that exist
  constexpr int rows_out_full = ROWS;

  auto compute_store = [&](int rb, u32x4 (&q)[TJ][KS]) {
    // ReLU-on-load once per row block (not per pass)
#pragma unroll
    for (int b = 0; b < TJ; b++)
#pragma unroll
      for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const uint32_t xw = q[b][ks][e];
          s16x2_t t = __builtin_bit_cast(s16x2_t, xw);
          t = __builtin_elementwise_max(t, floor2);
          q[b][ks][e] = __builtin_bit_cast(uint32_t, t);
        }
    const int r0 = rb * ROWS;
    constexpr int rows_out = pool ? rows_out_full / 4 : rows_out_full;
    const int jbase = pool ? (r0 >> 2) : r0;
#pragma unroll
    for (int ps = 0; ps < NP; ps++) {
      f32x16 acc[TI][TJ];
#pragma unroll
      for (int a = 0; a < TI; a++)
#pragma unroll
        for (int b = 0; b < TJ; b++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        bf16x8_t pf[TI], qf[TJ];
#pragma unroll
        for (int a = 0; a < TI; a++) {
          const u32x4 v = *(const u32x4*)(wsm + ((ps * TI + a) * 32 + frow) * WP + (2 * ks + fhi) * 16);
          pf[a] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int b = 0; b < TJ; b++) qf[b] = __builtin_bit_cast(bf16x8_t, q[b][ks]);
#pragma unroll
        for (int a = 0; a < TI; a++)
#pragma unroll
          for (int b = 0; b < TJ; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[a], qf[b], acc[a][b], 0, 0, 0);
      }
      // ---- wave-private epilogue of this pass (couts i0 + ps*TI*32 ..) ---------------------------------------------------------
      const int ic0 = ps * TI * 32;                    // first cout of the pass inside the tile
      const int ncp = ncr - ps * CPR;                  // 16-byte chunks of this pass that exist (cout tail)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the previous staging reads are done (LDS is in order per wave)
      // The tile copies below are straight-line code with a compile-time number of buffer instructions (lanes with nothing to do
      // carry an out-of-range offset: loads return zeros, stores are dropped). vmcnt is one in-order counter for loads AND stores:
      // only with an exact count can the wait in front of the next block's MFMAs leave this block's stores in flight (a
      // data-dependent loop made it drain them -- a full HBM write round trip per 32-pixel row block).
      if (pre_mask || pre_res) {
        const int ld = pre_mask ? epi.ldm : epi.ldr;
        u32x4 pre[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {          // all loads first, one wait
          const int idx = lane + 64 * it;
          const int r = idx / CPR, c = idx - r * CPR;
          const int jg = jbase + r;
          const bool ok = idx < rows_out * CPR && jg < Jout && c < ncp;
          const unsigned off = ok ? ((unsigned)jg * (unsigned)ld + (unsigned)(i0 + ic0 + c * 8)) * 2u : 0x80000000u;
          pre[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rss, (int)off, 0, 0));
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
          const int idx = lane + 64 * it;
          const int r = idx / CPR, c = idx - r * CPR;
          if (idx < rows_out * CPR) *(u32x4*)(stg + r * CP + c * 16) = pre[it];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int ta = 0; ta < TI; ta++)
#pragma unroll
        for (int tb = 0; tb < TJ; tb++) {
          const int jl = tb * 32 + (lane & 31);
#pragma unroll
          for (int g4 = 0; g4 < 4; g4++) {
            const int il = ta * 32 + 8 * g4 + 4 * (lane >> 5);       // cout inside the pass
            float v[4] = {acc[ta][tb][4 * g4 + 0], acc[ta][tb][4 * g4 + 1], acc[ta][tb][4 * g4 + 2], acc[ta][tb][4 * g4 + 3]};
            if (pool) {
#pragma unroll
              for (int e = 0; e < 4; e++) v[e] = quad_sum(v[e]);
            }
            const int jo = pool ? (jl >> 2) : jl;
            const bool act = (!pool || (lane & 3) == 0) && (jbase + jo < Jout);
            if (act) {
              char* loc = stg + jo * CP + il * 2;
#pragma unroll
              for (int e = 0; e < 4; e++) v[e] *= al;
              {
                const f32x4 bv = *(const f32x4*)(sbias + ic0 + il);
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] += bv[e];
              }
              if (pre_mask) {
                const u32x2 m = *(const u32x2*)loc;
#pragma unroll
                for (int e = 0; e < 4; e++) { const bf16_t h = (bf16_t)((m[e >> 1] >> (16 * (e & 1))) & 0xffffu); if (!(bf2f(h) > 0.f)) v[e] = 0.f; }
              }
              if (pre_res) {
                const u32x2 r = *(const u32x2*)loc;
#pragma unroll
                for (int e = 0; e < 4; e++) { const bf16_t h = (bf16_t)((r[e >> 1] >> (16 * (e & 1))) & 0xffffu); v[e] += epi.beta * bf2f(h); }
              }
              if (relu_out) {
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
              }
              u32x2 t;
              t[0] = pack2bf(v[0], v[1]);
              t[1] = pack2bf(v[2], v[3]);
              *(u32x2*)loc = t;
            }
          }
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int idx = lane + 64 * it;
        const int r = idx / CPR, c = idx - r * CPR;
        const int jg = jbase + r;
        const bool ok = idx < rows_out * CPR && jg < Jout && c < ncp;
        const unsigned off = ok ? ((unsigned)jg * (unsigned)epi.ldo + (unsigned)(i0 + ic0 + c * 8)) * 2u : 0x80000000u;
        const int rr = idx < rows_out * CPR ? r : 0;          // (rows past the tile: any valid staging address; the store is dropped)
        const u32x4 v = *(const u32x4*)(stg + rr * CP + c * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rso, (int)off, 0, 0);
      }
    }
  };

  const int stride = gridDim.x * NW;
  int rb = blockIdx.x * NW + wave;
  u32x4 qa[TJ][KS], qb[TJ][KS];
  if (rb >= p.nrb) return;
  fetch(rb, qa);
  // as many (dropped: out-of-range) stores as one row block issues, so that the loop is entered with the same vmcnt history it has
  // on its back edge -- otherwise the compiler's merged state assumes "no stores in flight" and the first wait of every other
  // iteration drains them
#pragma unroll
  for (int it = 0; it < NP * NIT; it++) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    __builtin_amdgcn_raw_buffer_store_b128(z, rso, (int)(0x80000000u + 16u * it), 0, 0);     // (distinct offsets: identical stores get merged)
  }
  while (true) {
    fetch(rb + stride, qb);
    compute_store(rb, qa);
    rb += stride;
    if (rb >= p.nrb) break;
    fetch(rb + stride, qa);
    compute_store(rb, qb);
    rb += stride;
    if (rb >= p.nrb) break;
  }
}

template <int TI, int NP, int TJ, int KS, bool POOL>
static inline int sg_launch_conv_sk_tp(ConvSkParams p, const Epilogue<bf16_t>& e, hipStream_t st) {
  constexpr int BI = TI * NP * 32, WP = KS * 32 + 16, CP = TI * 64 + 16, ROWS = 32 * TJ;
  // 8 waves per CU (2 per SIMD: the register budget): two 4-wave workgroups when their LDS fits twice, else one 8-wave workgroup
  // sharing one copy of the weights
  constexpr int lds4 = BI * WP + BI * 4 + 4 * ROWS * CP;
  constexpr int NW = (2 * lds4 <= 160 * 1024) ? 4 : 8;
  constexpr int lds = BI * WP + BI * 4 + NW * ROWS * CP;
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)sg_conv_sk_kernel<TI, NP, TJ, KS, NW, POOL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
    attr_done = true;
  }
  p.nrb = (p.J + ROWS - 1) / ROWS;
  const int tilesI = (p.I + BI - 1) / BI;
  int gx = (p.nrb + NW - 1) / NW;
  const int cap = (256 * (8 / NW) + tilesI - 1) / tilesI;
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL((sg_conv_sk_kernel<TI, NP, TJ, KS, NW, POOL>), dim3(gx, tilesI), dim3(64 * NW), lds, st, p, e);
  return 0;
}
template <int TI, int NP, int TJ, int KS>
static inline int sg_launch_conv_sk_t(const ConvSkParams& p, const Epilogue<bf16_t>& e, hipStream_t st) {
  return (e.flags & SG_EPI_POOL) ? sg_launch_conv_sk_tp<TI, NP, TJ, KS, true>(p, e, st) : sg_launch_conv_sk_tp<TI, NP, TJ, KS, false>(p, e, st);
}

// cout tile 32 / 64 / 96 (one pass) or 128 / 192 (two passes of 64 / 96); KS in {1,3,6,12} (K <= 16/48/96/192); TJ = 2 (64-pixel
// row blocks) only where accumulators + two fragment sets stay under 256 registers (2 waves per SIMD)
static inline int sg_launch_conv_sk(const ConvSkParams& p, const Epilogue<bf16_t>& e, hipStream_t st) {
  const int ti = p.I <= 32 ? 1 : p.I <= 64 ? 2 : p.I <= 96 ? 3 : p.I <= 128 ? 4 : 6;
  const int ks = p.K <= 16 ? 1 : p.K <= 48 ? 3 : p.K <= 96 ? 6 : 12;
#define SG_SK_CASE(T_, TI_, NP_, KS_, TJ_) if (ti == T_ && ks == KS_) return sg_launch_conv_sk_t<TI_, NP_, TJ_, KS_>(p, e, st);
  SG_SK_CASE(1, 1, 1, 1, 2) SG_SK_CASE(1, 1, 1, 3, 2) SG_SK_CASE(1, 1, 1, 6, 2) SG_SK_CASE(1, 1, 1, 12, 1)
  SG_SK_CASE(2, 2, 1, 1, 2) SG_SK_CASE(2, 2, 1, 3, 2) SG_SK_CASE(2, 2, 1, 6, 1) SG_SK_CASE(2, 2, 1, 12, 1)
  SG_SK_CASE(3, 3, 1, 1, 1) SG_SK_CASE(3, 3, 1, 3, 1) SG_SK_CASE(3, 3, 1, 6, 1) SG_SK_CASE(3, 3, 1, 12, 1)
  SG_SK_CASE(4, 2, 2, 1, 2) SG_SK_CASE(4, 2, 2, 3, 2) SG_SK_CASE(4, 2, 2, 6, 1) SG_SK_CASE(4, 2, 2, 12, 1)
  SG_SK_CASE(6, 3, 2, 1, 1) SG_SK_CASE(6, 3, 2, 3, 1) SG_SK_CASE(6, 3, 2, 6, 1) SG_SK_CASE(6, 2, 3, 12, 1)
#undef SG_SK_CASE
  return -1;
}
