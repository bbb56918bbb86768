// wgrad_sk.h -- streaming weight gradient for the THIN layers of the networks (bf16):
//   * 1x1 convolutions with <= 192 input and <= 96 output channels (attention theta / phi / g / o, the 1x1 skips of the first blocks),
//   * the 3x3 RGB stem of the discriminators (8 padded input channels -> <= 96 couts),
//   * the 3x3 RGB output layer of the generators (<= 96 channels -> 8 padded couts).
// Replaces autograd's convolution_backward (weight part) for reference src/utils/ops.py:165-173 on those shapes.
//
// These problems are HBM-bound (0.04-0.15 ms of traffic at batch 256) but a 256 x 128 MFMA tile is 3-28 % full on them: the tile
// kernels ran them at 0.27-2.1 ms (profiles/r02 layer table, "rgb_in" / "rgb_out" / k1 rows). Here the whole dW tile
// (I = taps*Cin <= 192 rows, J = Cout <= 96 columns) lives in the accumulators of ONE wave, and every wave streams its own pixel
// chunks with no workgroup barrier at all:
//   chunk = 32 consecutive output pixels; both operands are copied as they lie in memory ([pixel][channel] rows) into a wave-private
//   double-buffered LDS image by LDS-DMA (buffer descriptors: halo / tail / padded channels = out-of-range offsets = zeros), the
//   wave waits for ITS OWN previous chunk with a counted vmcnt, reads the MFMA fragments with ds_read_b64_tr_b16 (pixel-major ->
//   k-contiguous, the hardware 4x4 transpose) and issues NI x NJ MFMAs per 16 pixels.
//   3x3 layers: the thin operand (8 channels = 16 bytes per pixel) is staged as a 3-row halo patch; the nine taps are nine address
//   offsets of the transpose read, so the "row" index of the MFMA is (tap, channel) without any im2col copy. The generator's RGB
//   layer is the same kernel with the roles of x and dy exchanged: dW[co][tap][ci] = sum_p x[p][ci] * dy[p - tap][co].
//   Every wave writes its fp32 partial tile once; k_splitk_reduce sums them in a fixed order (deterministic gradients).
#pragma once
#include "gemm_core.h"
#include "conv_v2.h"

struct WgradSkParams {
  const bf16_t* a; const bf16_t* b;   // A: rows of the MFMA tile (dW's I index); B: columns (J index)
  int lda, ldb;                       // channel pitch of a stored pixel (elements)
  int CA, CB;                         // real channels of A (ONE mode) / B
  int H, W, wshift, hshift;           // output-pixel raster, powers of two
  int upA, upB;                       // operand stored at half resolution: pixel (h, w) reads (h >> 1, w >> 1)
  int HsA, WsA, HsB, WsB;             // stored extents
  int nchunk;                         // N * H * W / 32
  int sgn;                            // TAPS: A is read at pixel + sgn * tap
  int swap;                           // TAPS, generator RGB layer: tile (row = (tap, co), col = ci) -> dW[co][tap][ci]
  int arelu;                          // ReLU on the A operand (ONE mode)
  unsigned abytes, bbytes;            // descriptor extents
  int I, J;                           // dW is [J][I] fp32 (swap: [8][9][J])
  float* work; long long n;           // partial tiles: work + wave * n
  float alpha; const float* alpha_ptr;
};

__device__ __forceinline__ void sk_tr_read(unsigned addr, int off, u32x2& lo) {
  // inline asm: hipcc would put `s_waitcnt vmcnt(0)` in front of the builtin form while an LDS-DMA is in flight (see wgrad_v2.h)
  const unsigned a = addr + (unsigned)off;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=&v"(lo) : "v"(a));
}

// NI x NJ blocks of 32 x 32; TAPS: A = 8-channel operand with nine taps (NI == 3)
template <int NI, int NJ, bool TAPS>
__global__ __launch_bounds__(256) void sg_wgrad_sk_kernel(WgradSkParams p) {
  constexpr int APITCH = TAPS ? 16 : NI * 64;        // bytes per pixel row of the A image
  constexpr int BPITCH = NJ * 64;
  constexpr int ABUF = TAPS ? 2048 : 32 * APITCH;    // TAPS: [3 rows][40 pixels][16 B] = 1920 B in two DMA pieces
  constexpr int BBUF = 32 * BPITCH;
  constexpr int NPA = ABUF / 1024, NPB = BBUF / 1024;
  constexpr int WSZ = 2 * (ABUF + BBUF);
  constexpr int TROW = 640;                          // TAPS: bytes per patch row (40 pixels): rows land 32 banks apart
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  char* wbase = smem + wave * WSZ;
  const auto rsa = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.abytes, 0x00020000);
  const auto rsb = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, (int)p.bbytes, 0x00020000);

  // (the DMA lane coordinates -- pixel / byte column of piece j for this lane -- are recomputed per issue from compile-time pitches:
  // keeping them live cost the 6 x 3 instantiation 36 registers and pushed it into scratch)
  auto pix_off = [&](int pix, int up, int Hs, int Ws, int ld) -> unsigned {
    int w = pix & (p.W - 1);
    const int t = pix >> p.wshift;
    int h = t & (p.H - 1);
    const int n = t >> p.hshift;
    if (up) { h >>= 1; w >>= 1; }
    return ((unsigned)(n * Hs + h) * (unsigned)Ws + (unsigned)w) * (unsigned)ld * 2u;
  };
  auto issue = [&](int buf, int c) {
    char* ab = wbase + buf * (ABUF + BBUF);
    char* bb = ab + ABUF;
    const int p0 = c * 32;
    if (TAPS) {
      const int w0 = p0 & (p.W - 1);
      const int t = p0 >> p.wshift;
      const int h = t & (p.H - 1);
      const int n = t >> p.hshift;
#pragma unroll
      for (int j = 0; j < NPA; j++) {
        const int o = j * 1024 + lane * 16;
        const int prw = o / TROW, pcl = (o % TROW) >> 4;            // patch row / column of this lane's 16 bytes
        const int hh = h + prw - 1, ww = w0 + pcl - 1;
        const bool ok = (prw < 3) & (pcl < 34) & ((unsigned)hh < (unsigned)p.H) & ((unsigned)ww < (unsigned)p.W);
        unsigned off = ((unsigned)(n * p.H + hh) * (unsigned)p.W + (unsigned)ww) * (unsigned)p.lda * 2u;
        off = ok ? off : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (sg_lptr_t)(ab + j * 1024), 16, (int)off, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NPA; j++) {
        const int o = j * 1024 + lane * 16;
        const int px = o / APITCH, cb = o % APITCH;
        unsigned off = pix_off(p0 + px, p.upA, p.HsA, p.WsA, p.lda) + (unsigned)cb;
        off = ((cb >> 1) < p.CA) ? off : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (sg_lptr_t)(ab + j * 1024), 16, (int)off, 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < NPB; j++) {
      const int o = j * 1024 + lane * 16;
      const int px = o / BPITCH, cb = o % BPITCH;
      unsigned off = pix_off(p0 + px, p.upB, p.HsB, p.WsB, p.ldb) + (unsigned)cb;
      off = ((cb >> 1) < p.CB) ? off : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (sg_lptr_t)(bb + j * 1024), 16, (int)off, 0, 0, 0);
    }
  };

  // ---- loop-invariant fragment addresses (one transpose read = 4 pixel rows x 16 channels per 16-lane group) ---------------
  const int g16 = lane >> 4, t16 = lane & 15;
  const int prow = 8 * (g16 >> 1) + (t16 >> 2);                 // pixel row inside a 16-pixel k-step (second read: + 4)
  const int csub = 16 * (g16 & 1) + 4 * (t16 & 3);              // channel inside a 32-channel block
  const unsigned wb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)wbase;
  const unsigned bfrag = wb + ABUF + prow * BPITCH + csub * 2;
  unsigned afrag[NI];
#pragma unroll
  for (int a = 0; a < NI; a++) {
    if (TAPS) {
      int tap = 4 * a + 2 * (g16 & 1) + ((t16 & 3) >> 1);
      if (tap > 8) tap = 8;                                      // rows 72..95 of the tile are never stored
      const int dr = tap / 3 - 1, ds = tap % 3 - 1;
      afrag[a] = wb + (1 + p.sgn * dr) * TROW + (1 + prow + p.sgn * ds) * 16 + 8 * (t16 & 1);
    } else {
      afrag[a] = wb + prow * APITCH + (a * 32 + csub) * 2;
    }
  }

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int a = 0; a < NI; a++)
#pragma unroll
    for (int b = 0; b < NJ; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  constexpr int NP = NPA + NPB;
  const uint32_t relu_bound = p.arelu ? 0u : 0x80008000u;
  int c = gw;
  if (c < p.nchunk) issue(0, c);
  int buf = 0;
  for (; c < p.nchunk; c += nw) {
    const bool more = (c + nw) < p.nchunk;
    if (more) issue(buf ^ 1, c + nw);
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned bo = (unsigned)(buf * (ABUF + BBUF));
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      // A blocks in groups of <= 3 (the 6 x 3 tile keeps 288 accumulators: fragments of all 6 row blocks at once spilled)
      constexpr int AG = NI < 3 ? NI : 3;
#pragma unroll
      for (int a0 = 0; a0 < NI; a0 += AG) {
        u32x2 al[AG], ah[AG], bl[NJ], bh[NJ];
#pragma unroll
        for (int aa = 0; aa < AG; aa++) {
          const int a = a0 + aa;
          if (TAPS) { sk_tr_read(afrag[a] + bo, ks * 256, al[aa]); sk_tr_read(afrag[a] + bo, ks * 256 + 64, ah[aa]); }
          else { sk_tr_read(afrag[a] + bo, ks * 16 * APITCH, al[aa]); sk_tr_read(afrag[a] + bo, (ks * 16 + 4) * APITCH, ah[aa]); }
        }
#pragma unroll
        for (int b = 0; b < NJ; b++) {
          sk_tr_read(bfrag + bo, b * 64 + ks * 16 * BPITCH, bl[b]);
          sk_tr_read(bfrag + bo, b * 64 + (ks * 16 + 4) * BPITCH, bh[b]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bf16x8_t af[AG], bf[NJ];
#pragma unroll
        for (int aa = 0; aa < AG; aa++) {
          asm volatile("" : "+v"(al[aa]), "+v"(ah[aa]));      // the MFMAs below must not be scheduled above the wait
          u32x4 v = {al[aa][0], al[aa][1], ah[aa][0], ah[aa][1]};
          if (!TAPS) {            // branch-free ReLU switch: signed 16-bit max with 0 (ReLU of bf16) or with -32768 (identity)
            typedef short sk_s16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const uint32_t xq = v[q];
              sk_s16x2 x2 = __builtin_bit_cast(sk_s16x2, xq);
              x2 = __builtin_elementwise_max(x2, __builtin_bit_cast(sk_s16x2, relu_bound));
              v[q] = __builtin_bit_cast(uint32_t, x2);
            }
          }
          af[aa] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int b = 0; b < NJ; b++) {
          asm volatile("" : "+v"(bl[b]), "+v"(bh[b]));
          u32x4 v = {bl[b][0], bl[b][1], bh[b][0], bh[b][1]};
          bf[b] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int aa = 0; aa < AG; aa++)
#pragma unroll
          for (int b = 0; b < NJ; b++)
            acc[a0 + aa][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[aa], bf[b], acc[a0 + aa][b], 0, 0, 0);
      }
    }
    buf ^= 1;
  }

  // ---- the wave's partial tile --------------------------------------------------------------------------------------------
  float al = p.alpha;
  if (p.alpha_ptr) al *= *p.alpha_ptr;
  float* out = p.work + (long long)gw * p.n;
#pragma unroll
  for (int a = 0; a < NI; a++)
#pragma unroll
    for (int b = 0; b < NJ; b++) {
      const int col = b * 32 + (lane & 31);
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int row = a * 32 + 8 * g4 + 4 * (lane >> 5);
        if (p.swap) {
          const int tap = row >> 3, co = row & 7;       // co .. co + 3
          if (tap < 9 && col < p.J) {
#pragma unroll
            for (int e = 0; e < 4; e++) out[((co + e) * 9 + tap) * p.J + col] = acc[a][b][4 * g4 + e] * al;
          }
        } else if (col < p.J && row < p.I) {
          f32x4 v = {acc[a][b][4 * g4 + 0] * al, acc[a][b][4 * g4 + 1] * al, acc[a][b][4 * g4 + 2] * al, acc[a][b][4 * g4 + 3] * al};
          *(f32x4*)(out + (long long)col * p.I + row) = v;
        }
      }
    }
}

template <int NI, int NJ, bool TAPS>
static inline int sg_launch_wgrad_sk_t(const WgradSkParams& p, int nwg, hipStream_t st) {
  constexpr int ABUF = TAPS ? 2048 : 32 * NI * 64, BBUF = 32 * NJ * 64;
  constexpr int LDS = 4 * 2 * (ABUF + BBUF);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)sg_wgrad_sk_kernel<NI, NJ, TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1;
    attr_done = true;
  }
  hipLaunchKernelGGL((sg_wgrad_sk_kernel<NI, NJ, TAPS>), dim3(nwg), dim3(256), LDS, st, p);
  return 0;
}

// number of waves (= partial tiles) the launch uses for a problem of nchunk 32-pixel chunks
static inline int sg_wgrad_sk_waves(int nchunk) {
  int nw = nchunk / 8;
  if (nw > 1024) nw = 1024;
  if (nw < 4) nw = 4;
  return (nw + 3) & ~3;
}

static inline int sg_launch_wgrad_sk(const WgradSkParams& p, bool taps, int NI, int NJ, hipStream_t st) {
  const int nwg = sg_wgrad_sk_waves(p.nchunk) / 4;
  if (taps) {
    if (NJ == 1) return sg_launch_wgrad_sk_t<3, 1, true>(p, nwg, st);
    if (NJ == 2) return sg_launch_wgrad_sk_t<3, 2, true>(p, nwg, st);
    return sg_launch_wgrad_sk_t<3, 3, true>(p, nwg, st);
  }
#define SK_CASE(A, B) if (NI == A && NJ == B) return sg_launch_wgrad_sk_t<A, B, false>(p, nwg, st);
  SK_CASE(1, 1) SK_CASE(1, 2) SK_CASE(1, 3) SK_CASE(2, 1) SK_CASE(2, 2) SK_CASE(2, 3)
  SK_CASE(3, 1) SK_CASE(3, 2) SK_CASE(3, 3) SK_CASE(6, 1) SK_CASE(6, 2) SK_CASE(6, 3)
#undef SK_CASE
  return -1;
}
