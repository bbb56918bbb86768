// conv_v3.h -- "halo" forward / data-gradient kernel for 3x3 stride-1 pad-1 convolutions (bf16, >= 64 input channels).
//
// conv_v2.h stages one im2col k-tile per step: for a 3x3 filter every input pixel of the tile travels L2 -> LDS nine times
// (once per tap), and with two LDS buffers only ONE k-tile of DMA is ever in flight -- the k-loop runs at
// (bytes in flight) / (memory latency), ~40 KB/us per CU, far below what the MFMAs could consume on the <= 192-channel layers.
// Here the pixel operand is staged ONCE per 64-channel slice as a PATCH: the tile's pixels plus one image row (+1 pixel) of
// halo on either side, in raster order, [pixel][64 channels] = 128-byte rows with the same source-side XOR swizzle as conv_v2.
// The nine taps then read their MFMA fragments from the same patch at SHIFTED row indices (row + (tr-1)*W + (ts-1)); pixels
// outside the image are redirected to a zero line. Only the weight tile (BI x 64) is streamed per tap. L2 -> LDS bytes per
// (tile, slice): patch (BJ + 2W + 16) * 128 + 9 * BI * 128   instead of   9 * (BJ + BI) * 128   (1.9x fewer on 192 -> 192 @64^2,
// 3.3x fewer on 96 -> 96 @128^2), and the next slice's patch is in flight during all nine taps of the current one.
//
//   * raster row order: patch row of tile pixel p, tap (tr,ts) = (p - j0) + W + 8 + (tr-1)*W + (ts-1)
//   * quad row order (2x2 pooling epilogue): same patch (the tile still covers BJ/W whole image rows), per-lane row base
//   * nearest x2 upsample on load: the patch holds SOURCE pixels (4x fewer); row = base + ((ph+tr-1)>>1)*Ws + ((pw+ts-1)>>1)
//   * fragment addresses: (row << 7) | ((chunk ^ (row >> 1 & 7)) << 4): any 16 rows with distinct residues mod 16 are
//     conflict-free for ds_read_b128, so every shift is as good as the unshifted read
//   * epilogue: sg_conv_epilogue (conv_v2.h)
#pragma once
#include <type_traits>
#include "conv_v2.h"

struct ConvV3Params {
  const bf16_t* x; const bf16_t* w;
  int W;                  // source image width (Ws): power of two >= 4; patch halo = one source row + 8 pixels either side
  int wlog;               // log2(W)
  int C, ldx;
  int Ho, Wo, wshift, hshift;
  int flags;
  int I, J, K;
  int nslice;             // ceil(C / 64)
  int npix_src;           // N * Hs * Ws
  int npx;                // patch pixels = BJ(/4 with upsample) + 2 W + 16
  unsigned xbytes, wbytes;
  int zero_off, bias_off; // LDS byte offsets of the zero line / bias vector (behind the output staging area)
  int dump_off;           // 1 KiB of LDS that absorbs the DMA pieces a wave has no use for (see "branch-free DMA" below)
  int pm4, psh;           // image-row parity term of the chunk swizzle (quad row order, W >= 16; see conv_v4.h): chunk ^= ((pixel >> psh) & pm4), pm4 = 4 or 0
};

// W3: three weight buffers (PB2 only). The weights of tap t+1 are then complete and visible one barrier EARLIER than they are
// needed, so the first fragments of tap t+1 are requested before the barrier that ends tap t: the matrix pipe does not drain at
// every tap boundary while all eight waves sit behind the barrier and then queue for the LDS at once.
//
// NKL: 16-channel sub-steps of the LAST 64-channel slice -- 4 (C % 64 == 0), 2 (C % 64 == 32: the 96-channel layers) or 0 = decided at
// run time (any other C). Straight-line taps: with the sub-step count a run-time value and the DMA pieces behind wave-dependent
// branches, every tap was a chain of small basic blocks, and at each join the compiler's s_waitcnt insertion had to assume the
// shorter path: the MFMAs of sub-step k waited for the fragment loads of sub-step k + 1 that had just been issued in front of them
// (s_waitcnt lgkmcnt(1) / lgkmcnt(0) right before every group of MFMAs in the r02 disassembly) -- the register double buffer of the
// fragments bought nothing. Now the sub-step count is a compile-time constant per slice (the slice body is instantiated for 4 and
// for NKL) and the DMA is branch-free: every wave issues every piece of every tap; a piece it has no use for (cout rows beyond the
// tile, patch groups beyond the patch, the prefetch behind the last slice) goes, with an out-of-range source offset (zero fill,
// no memory traffic), to a 1 KiB dump area of LDS. A tap is one basic block and the compiler counts its own loads exactly.
template <int BI, int WJ, int WI, int BJ, bool RELU, bool UP, bool PB2, bool W3 = false, int NKL = 0>
__global__ __launch_bounds__(64 * WJ * WI) void sg_conv_v3_kernel(ConvV3Params p, Epilogue<bf16_t> epi, int tilesI, int tilesJ) {
  static_assert(!W3 || PB2, "W3 needs the double patch buffer");
  constexpr int NW = WJ * WI;
  constexpr int PB = BI * 128;                 // one weight tile (BI couts x 64 channels)
  constexpr int NPI = (BI / 8 + NW - 1) / NW;  // weight DMA pieces per wave per tap (upper bound)
  constexpr int TJ = BJ / WJ / 32, TI = BI / WI / 32;
  static_assert(NW == 8 || NW == 4, "8 waves (2 per SIMD) or 4 waves (1 per SIMD, 512 registers: 12-16 accumulator blocks per wave)");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = tilesI * tilesJ;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tI = bid % tilesI, tJ = bid / tilesI;
  const int i0 = tI * BI, j0 = tJ * BJ;
  const int patch_bytes = p.npx * 128;
  char* const pbufs = smem + (PB2 ? 2 : 1) * patch_bytes;      // the two weight buffers sit behind the patch buffer(s)
  float* sbias = (float*)(smem + p.bias_off);
  if (epi.bias) {
    for (int i = tid; i < BI; i += 64 * NW) sbias[i] = (i0 + i < epi.I) ? epi.bias[i0 + i] : 0.f;
  }
  if (tid < 32) ((unsigned*)(smem + p.zero_off))[tid] = 0u;

  const auto rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
  const auto rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.wbytes, 0x00020000);
  const int sub = lane >> 3;
  const int lc = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);       // logical 16-byte chunk this lane fetches (DMA group g: key = 4g + sub/2)
  const unsigned ldx2 = 2u * (unsigned)p.ldx;

  // ---- patch DMA: groups of 8 consecutive source pixels, group g = wave + 8 i ----------------------------------------------
  const int P0 = (UP ? (j0 >> 2) : j0) - p.W - 8;                    // raster index of patch row 0
  const int ngroups = p.npx >> 3;
  const int npw = (ngroups - wave + NW - 1) / NW;                    // groups of this wave (wave-uniform)
  const int pix0 = P0 + 8 * wave + sub;                              // this lane's pixel in group i = 0
  const unsigned poff0 = (unsigned)pix0 * ldx2 + (unsigned)lc * 16u;
  const unsigned pstep = 8u * NW * ldx2;
  // slice s: channel chunk lc of the slice exists when s*64 + lc*8 < C
  char* const dump = smem + p.dump_off;
  auto patch_piece = [&](int buf, int s, int i, bool en) {           // en: wave-uniform; false = the piece goes to the dump area
    const int pix = pix0 + 8 * NW * i;
    unsigned off = poff0;
    asm volatile("" : "+v"(off));     // (same: one live address per piece in flight, not one per piece of the slice)
    off += (unsigned)i * pstep + (unsigned)s * 128u;
    const int lce = lc ^ ((pix >> p.psh) & p.pm4);                  // quad row order: the image-row parity joins the swizzle key
    off += (unsigned)((lce - lc) * 16);
    const bool ok = en && ((unsigned)pix < (unsigned)p.npix_src) && (s * 64 + lce * 8 < p.C);
    off = ok ? off : 0x80000000u;
    char* dst = en ? smem + buf * patch_bytes + (wave + NW * i) * 1024 : dump;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (sg_lptr_t)dst, 16, (int)off, 0, 0, 0);
  };

  // ---- weight DMA: BI rows x 64 channels of tap t, slice s ------------------------------------------------------------------
  unsigned woff[NPI];
#pragma unroll
  for (int i = 0; i < NPI; i++) {
    const int g = wave + NW * i;
    const int row = i0 + 8 * g + sub;
    woff[i] = ((g < BI / 8) && (row < p.I)) ? (unsigned)row * (unsigned)p.K * 2u : 0x40000000u;
  }
  auto weight_piece = [&](int buf, int s, int t, int j, bool en) {
    const unsigned kw = (en && s * 64 + lc * 8 < p.C) ? (unsigned)((t * p.C + s * 64 + lc * 8) * 2) : 0x40000000u;   // 2^30 (+ 2^30) >= wbytes: zero fill
    en = en && (NW * (j + 1) <= BI / 8 || wave + NW * j < BI / 8);
    char* dst = en ? pbufs + buf * PB + (wave + NW * j) * 1024 : dump;
    unsigned wo = woff[j];
    asm volatile("" : "+v"(wo));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (sg_lptr_t)dst, 16, (int)(wo + kw), 0, 0, 0);
  };
  auto weight_tile = [&](int buf, int s, int t) {
#pragma unroll
    for (int j = 0; j < NPI; j++) weight_piece(buf, s, t, j, true);
  };

  // ---- fragment rows of this lane ---------------------------------------------------------------------------------------------
  const int wj = wave % WJ, wi = wave / WJ;
  const int wj0 = wj * (BJ / WJ), wi0 = wi * (BI / WI);
  const int frow = lane & 31, fhi = lane >> 5;
  int rb[TJ];             // patch row of the centre pixel
  unsigned qinv[TJ];      // bit t set = tap t reads outside the image (or the row is outside the problem)
  int rs0[TJ], rs2[TJ], cs0[TJ], cs2[TJ];   // row / column displacement of taps tr = 0, 2 and ts = 0, 2 (tr = ts = 1: none)
#pragma unroll
  for (int b = 0; b < TJ; b++) {
    const int row = j0 + wj0 + b * 32 + frow;
    int n, ho, wo;
    if (p.flags & SG_PIX_QUAD) {
      const int q = row >> 2, dy = (row >> 1) & 1, dx = row & 1;
      const int wq = q & ((p.Wo >> 1) - 1);
      const int t = q >> (p.wshift - 1);
      const int hq = t & ((p.Ho >> 1) - 1);
      n = t >> (p.hshift - 1);
      ho = 2 * hq + dy; wo = 2 * wq + dx;
    } else {
      wo = row & (p.Wo - 1); const int t = row >> p.wshift; ho = t & (p.Ho - 1); n = t >> p.hshift;
    }
    unsigned m = 0;
    if (row < p.J) {
#pragma unroll
      for (int rr = 0; rr < 3; rr++)
#pragma unroll
        for (int ss = 0; ss < 3; ss++)
          if ((unsigned)(ho - 1 + rr) < (unsigned)p.Ho && (unsigned)(wo - 1 + ss) < (unsigned)p.Wo) m |= 1u << (rr * 3 + ss);
    }
    qinv[b] = ~m;
    if (UP) {
      const int Hs = p.Ho >> 1;
      const int spc = ((n * Hs + (ho >> 1)) << p.wlog) + (wo >> 1);
      rb[b] = spc - P0;
      rs0[b] = (ho & 1) ? 0 : -p.W; rs2[b] = (ho & 1) ? p.W : 0;
      cs0[b] = (wo & 1) ? 0 : -1;  cs2[b] = (wo & 1) ? 1 : 0;
    } else {
      rb[b] = (((n << p.hshift) + ho) << p.wshift) + wo - P0;
      rs0[b] = -p.W; rs2[b] = p.W; cs0[b] = -1; cs2[b] = 1;
    }
  }

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; a++)
#pragma unroll
    for (int b = 0; b < TJ; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  const int nslice = p.nslice;
  // sub-steps of the last slice when NKL == 0 (run-time): a 96-channel layer's second slice is half empty, and running its zero
  // half cost 25 % of the layer's MFMAs and fragment reads
  const int crem_last = p.C - (nslice - 1) * 64;
  const int nks_last_rt = crem_last >= 64 ? 4 : ((crem_last + 15) >> 4);

  if constexpr (W3) {
    // ---- three weight buffers: buffer of step (9 s + t) = t % 3; DMA runs two taps ahead; fragments one sub-step ahead ACROSS taps --
    for (int i = 0; i < npw; i++) patch_piece(0, 0, i, true);
    weight_tile(0, 0, 0);
    weight_tile(1, 0, 1);
    __syncthreads();
    const int ppt8 = (npw + 7) >> 3;                 // patch pieces of the next slice per tap, taps 0..7 (<= 2: launcher)
    bf16x8_t pf[2][TI], qf[2][TJ];
    unsigned qa[TJ];                                 // fragment byte offsets (from smem) of the tap whose sub-step 0 is loaded next
    auto make_qa = [&](int t, int poff) {            // t: compile-time tap; poff: byte offset of the slice's patch buffer
      const int tr = t / 3, ts = t % 3;
#pragma unroll
      for (int b = 0; b < TJ; b++) {
        int row = rb[b];
        asm volatile("" : "+v"(row));   // keeps the nine taps' address arithmetic inside their taps (hoisted out of the slice loop it spilled)
        if (tr == 0) row += rs0[b];
        if (tr == 2) row += rs2[b];
        if (ts == 0) row += cs0[b];
        if (ts == 2) row += cs2[b];
        const int ipar = ((row + P0) >> p.psh) & p.pm4;
        unsigned a = (unsigned)poff + (((unsigned)row << 7) | ((unsigned)((fhi ^ (row >> 1) ^ ipar) & 7) << 4));
        a = ((qinv[b] >> t) & 1u) ? (unsigned)p.zero_off : a;
        qa[b] = a;
      }
    };
    auto load = [&](int ks, int slot, const char* ps) {
#pragma unroll
      for (int a = 0; a < TI; a++) {
        const int row = wi0 + a * 32 + frow;
        const int ch = (ks * 2 + fhi) ^ ((row >> 1) & 7);
        u32x4 v = *(const u32x4*)(ps + row * 128 + ch * 16);
        pf[slot][a] = __builtin_bit_cast(bf16x8_t, v);
      }
#pragma unroll
      for (int b = 0; b < TJ; b++) {
        u32x4 v = *(const u32x4*)(smem + (qa[b] ^ (unsigned)(ks * 32)));
        if (RELU) v = relu16<bf16_t>(v);
        qf[slot][b] = __builtin_bit_cast(bf16x8_t, v);
      }
    };
    make_qa(0, 0);
    load(0, 0, pbufs);
    auto run_slice = [&](int s, auto nkc) __attribute__((always_inline)) {
      constexpr int NKC = decltype(nkc)::value;      // sub-steps of this slice; 0: run-time
      const int nks = NKC ? NKC : nks_last_rt;
      const bool next_slice = s + 1 < nslice;
      const int poff_cur = (s & 1) * patch_bytes, poff_nxt = ((s + 1) & 1) * patch_bytes;
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const char* ps = pbufs + (t % 3) * PB;
        auto dma_piece = [&](int k) {                // piece k of this tap: weights two taps ahead, then a share of the next slice's patch
          if (k < NPI) {
            if (t + 2 < 9) weight_piece((t + 2) % 3, s, t + 2, k, true);
            else weight_piece((t + 2) % 3, s + 1, t + 2 - 9, k, next_slice);
          } else if (t < 8) {
            const int i = t * ppt8 + (k - NPI);
            patch_piece((s + 1) & 1, s + 1, i, next_slice && (k - NPI) < ppt8 && i < npw);
          }
        };
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
          if (ks < 3 && ks + 1 < nks) load(ks + 1, (ks + 1) & 1, ps);
          if (ks < nks) {
#pragma unroll
            for (int a = 0; a < TI; a++)
#pragma unroll
              for (int b = 0; b < TJ; b++) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[ks & 1][a], qf[ks & 1][b], acc[a][b], 0, 0, 0);
                // a piece costs 60-180 issue cycles: one behind each of the first sub-step's MFMAs, where the matrix pipe is busy anyway
                if (ks == 0 && a * TJ + b < NPI + 2) {
                  __builtin_amdgcn_sched_barrier(0);
                  dma_piece(a * TJ + b);
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
            if (ks == 0) {
#pragma unroll
              for (int k = TI * TJ; k < NPI + 2; k++) dma_piece(k);
            }
          }
          __builtin_amdgcn_sched_barrier(0);       // the scheduler keeps the software pipeline as written: fragments of sub-step k + 1, then the MFMAs
                                                   // of sub-step k (left alone it pulls the loads of several sub-steps to the tap's head and spills)
        }
        // sub-step 0 of the next tap (its weights were complete at the PREVIOUS barrier; the patch of the next slice at barrier 7)
        if (t < 8) { make_qa(t + 1, poff_cur); load(0, 0, pbufs + ((t + 1) % 3) * PB); }
        else if (next_slice) { make_qa(0, poff_nxt); load(0, 0, pbufs); }
        __syncthreads();
      }
    };
    if constexpr (NKL == 4) {
      for (int s = 0; s < nslice; s++) run_slice(s, std::integral_constant<int, 4>{});
    } else {                                         // last slice peeled (an if / else of the two bodies inside one loop spilled registers)
      for (int s = 0; s + 1 < nslice; s++) run_slice(s, std::integral_constant<int, 4>{});
      run_slice(nslice - 1, std::integral_constant<int, NKL>{});
    }
  } else {
  // ---- prologue: patch of slice 0 and the weights of (slice 0, tap 0) ------------------------------------------------------------
  for (int i = 0; i < npw; i++) patch_piece(0, 0, i, true);
  weight_tile(0, 0, 0);
  __syncthreads();

  const int ppt = (npw + 8) / 9;                   // patch pieces of the next slice issued per tap (PB2; <= 2: launcher)
  auto run_slice = [&](int s, auto nkc) __attribute__((always_inline)) {
    constexpr int NKC = decltype(nkc)::value;      // sub-steps of this slice; 0: run-time
    const int nks = NKC ? NKC : nks_last_rt;
    const char* patch = smem + (PB2 ? (s & 1) : 0) * patch_bytes;
    const bool next_slice = s + 1 < nslice;
    const int step0 = 9 * s;
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const int step = step0 + t;
      // prefetch: weights of the next (slice, tap); a share (<= 2 pieces: launcher) of the next slice's patch. DMA piece k of this
      // tap: k < NPI = weight piece k, k = NPI, NPI + 1 = patch pieces. A piece costs 60-180 issue cycles: one piece behind each of
      // the first sub-step's MFMAs, where the matrix pipe is busy anyway.
      auto dma_piece = [&](int k) {
        if (k < NPI) {
          if (t < 8) weight_piece((step + 1) & 1, s, t + 1, k, true);
          else weight_piece((step + 1) & 1, s + 1, 0, k, next_slice);
        } else if (PB2) {
          const int i = t * ppt + (k - NPI);
          patch_piece((s + 1) & 1, s + 1, i, next_slice && (k - NPI) < ppt && i < npw);
        }
      };
      const char* ps = pbufs + (step & 1) * PB;
      // fragment base addresses of this tap
      const int tr = t / 3, ts = t % 3;            // compile-time after unrolling
      unsigned qa[TJ];
#pragma unroll
      for (int b = 0; b < TJ; b++) {
        int row = rb[b];
        asm volatile("" : "+v"(row));   // keeps the nine taps' address arithmetic inside their taps (hoisted out of the slice loop it spilled)
        if (tr == 0) row += rs0[b];
        if (tr == 2) row += rs2[b];
        if (ts == 0) row += cs0[b];
        if (ts == 2) row += cs2[b];
        const int ipar = ((row + P0) >> p.psh) & p.pm4;
        unsigned a = ((unsigned)row << 7) | ((unsigned)((fhi ^ (row >> 1) ^ ipar) & 7) << 4);
        a = ((qinv[b] >> t) & 1u) ? (unsigned)(p.zero_off - (PB2 ? (s & 1) : 0) * patch_bytes) : a;
        qa[b] = a;
      }
      bf16x8_t pf[2][TI], qf[2][TJ];
      auto load = [&](int ks, int slot) {
#pragma unroll
        for (int a = 0; a < TI; a++) {
          const int row = wi0 + a * 32 + frow;
          const int ch = (ks * 2 + fhi) ^ ((row >> 1) & 7);
          u32x4 v = *(const u32x4*)(ps + row * 128 + ch * 16);
          pf[slot][a] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int b = 0; b < TJ; b++) {
          u32x4 v = *(const u32x4*)(patch + (qa[b] ^ (unsigned)(ks * 32)));
          if (RELU) v = relu16<bf16_t>(v);
          qf[slot][b] = __builtin_bit_cast(bf16x8_t, v);
        }
      };
      load(0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        if (ks < 3 && ks + 1 < nks) load(ks + 1, (ks + 1) & 1);
        if (ks < nks) {
#pragma unroll
          for (int a = 0; a < TI; a++)
#pragma unroll
            for (int b = 0; b < TJ; b++) {
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[ks & 1][a], qf[ks & 1][b], acc[a][b], 0, 0, 0);
              if (ks == 0 && a * TJ + b < NPI + 2) {
                __builtin_amdgcn_sched_barrier(0);
                dma_piece(a * TJ + b);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          if (ks == 0) {   // fewer MFMAs in a sub-step than pieces: the rest behind the sub-step
#pragma unroll
            for (int k = TI * TJ; k < NPI + 2; k++) dma_piece(k);
          }
        }
        __builtin_amdgcn_sched_barrier(0);         // (see the three-buffer path)
      }
      __syncthreads();
    }
    if (!PB2 && next_slice) {                      // single patch buffer: the next slice's patch cannot overlap the taps
      for (int i = 0; i < npw; i++) patch_piece(0, s + 1, i, true);
      __syncthreads();
    }
  };
  if constexpr (NKL == 4) {
    for (int s = 0; s < nslice; s++) run_slice(s, std::integral_constant<int, 4>{});
  } else {
    for (int s = 0; s + 1 < nslice; s++) run_slice(s, std::integral_constant<int, 4>{});
    run_slice(nslice - 1, std::integral_constant<int, NKL>{});
  }

  }   // !W3

  float al = epi.alpha;
  if (epi.alpha_ptr) al *= *epi.alpha_ptr;
  sg_conv_epilogue<BI, BJ, NW, TI, TJ>(acc, smem, sbias, epi, i0, j0, wi0, wj0, al);
}

template <int BI, int WJ, int WI, int BJ, bool RELU, bool UP, bool PB2, bool W3, int NKL>
static inline int sg_launch_conv_v3r(ConvV3Params p, const Epilogue<bf16_t>& e, hipStream_t st) {
  const int patch_bytes = p.npx * 128;
  int body = (PB2 ? 2 : 1) * patch_bytes + (W3 ? 3 : 2) * BI * 128;
  const int stage = BJ * (BI * 2 + 16);
  if (stage > body) body = stage;
  p.zero_off = body; p.bias_off = body + 128; p.dump_off = body + 128 + BI * 4;
  const int lds = body + 128 + BI * 4 + 1024;
  if (lds > 160 * 1024) return -1;
  static int attr_lds = 0;
  if (lds > attr_lds) {
    if (hipFuncSetAttribute((const void*)sg_conv_v3_kernel<BI, WJ, WI, BJ, RELU, UP, PB2, W3, NKL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
    attr_lds = 160 * 1024;
  }
  const int tilesI = (p.I + BI - 1) / BI, tilesJ = (p.J + BJ - 1) / BJ;
  hipLaunchKernelGGL((sg_conv_v3_kernel<BI, WJ, WI, BJ, RELU, UP, PB2, W3, NKL>), dim3(tilesI * tilesJ), dim3(64 * WJ * WI), lds, st, p, e, tilesI, tilesJ);
  return 0;
}
// LDS need of a configuration (bytes), or -1 when it does not fit
static inline int sg_conv_v3_lds(int BI, int BJ, int npx, bool pb2, bool w3 = false) {
  int body = (pb2 ? 2 : 1) * npx * 128 + (w3 ? 3 : 2) * BI * 128;
  const int stage = BJ * (BI * 2 + 16);
  if (stage > body) body = stage;
  const int lds = body + 128 + BI * 4 + 1024;
  return lds <= 160 * 1024 ? lds : -1;
}
template <int BI, int WJ, int WI, int BJ, int NKL>
static inline int sg_launch_conv_v3(const ConvV3Params& p, const Epilogue<bf16_t>& e, hipStream_t st) {
  const bool up = (p.flags & SG_PIX_UPSAMPLE) != 0, relu = (p.flags & SG_PIX_RELU) != 0;
  const bool pb2 = p.nslice > 1 && sg_conv_v3_lds(BI, BJ, p.npx, true) > 0;
  if (!pb2 && sg_conv_v3_lds(BI, BJ, p.npx, false) < 0) return -1;
  if constexpr (BJ == 256 && (BI == 192 || BI == 128)) {      // the deep-layer configurations: three weight buffers when they fit
    static int w3_mode = -1;
    if (w3_mode < 0) { const char* e3 = getenv("SG_V3_W3"); w3_mode = (e3 && e3[0] == '0') ? 0 : 1; }
    if (w3_mode && pb2 && sg_conv_v3_lds(BI, BJ, p.npx, true, true) > 0 && (((p.npx >> 3) + WJ * WI - 1) / (WJ * WI) + 7) / 8 <= 2) {
      if (relu) return up ? sg_launch_conv_v3r<BI, WJ, WI, BJ, true, true, true, true, NKL>(p, e, st) : sg_launch_conv_v3r<BI, WJ, WI, BJ, true, false, true, true, NKL>(p, e, st);
      return up ? sg_launch_conv_v3r<BI, WJ, WI, BJ, false, true, true, true, NKL>(p, e, st) : sg_launch_conv_v3r<BI, WJ, WI, BJ, false, false, true, true, NKL>(p, e, st);
    }
  }
#define SG_V3_CASE(R_, U_, P_) if (relu == R_ && up == U_ && pb2 == P_) return sg_launch_conv_v3r<BI, WJ, WI, BJ, R_, U_, P_, false, NKL>(p, e, st);
  SG_V3_CASE(false, false, false) SG_V3_CASE(false, false, true) SG_V3_CASE(false, true, false) SG_V3_CASE(false, true, true)
  SG_V3_CASE(true, false, false) SG_V3_CASE(true, false, true) SG_V3_CASE(true, true, false) SG_V3_CASE(true, true, true)
#undef SG_V3_CASE
  return -1;
}
// tile choice -> instantiation, for one value of NKL (explicitly instantiated in conv_v3.hip (4) and conv_v3b.hip (2): two translation units, built in parallel)
// four-wave variants (one wave per SIMD, 96 x 128 / 64 x 128 register tiles: 7 fragment reads per 12 MFMAs instead of 5 per 6) of the
// deep-layer tiles; instantiated in conv_v3c.hip
template <int NKL>
int sg_conv_v3_dispatch_nw4(int best, int BJ, const ConvV3Params& p, const Epilogue<bf16_t>& e, hipStream_t st) {
  if ((((p.npx >> 3) + 3) / 4 + 8) / 9 > 2) return -1;      // at most two patch pieces per tap and wave
  if (best == 192 && BJ == 256) return sg_launch_conv_v3<192, 2, 2, 256, NKL>(p, e, st);
  if (best == 128 && BJ == 256) return sg_launch_conv_v3<128, 2, 2, 256, NKL>(p, e, st);
  return -1;
}
template <int NKL>
int sg_conv_v3_dispatch(int best, int BJ, const ConvV3Params& p, const Epilogue<bf16_t>& e, hipStream_t st) {
  if (best == 192) return sg_launch_conv_v3<192, 4, 2, 256, NKL>(p, e, st);
  if (best == 128) return sg_launch_conv_v3<128, 4, 2, 256, NKL>(p, e, st);
  if (best == 32) return sg_launch_conv_v3<32, 8, 1, 512, NKL>(p, e, st);
  if (BJ == 512) return sg_launch_conv_v3<96, 8, 1, 512, NKL>(p, e, st);
  return sg_launch_conv_v3<96, 8, 1, 256, NKL>(p, e, st);
}
