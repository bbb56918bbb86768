// conv_rs.h -- "row-streaming" forward / data-gradient kernel for 3x3 stride-1 pad-1 convolutions with FEW output channels (<= 32) on
// 128-pixel-wide images: the generator's RGB layer (96 -> 3, written as 8 channels) and the data gradient of the discriminator's RGB
// stem (96 -> 8) of the ImageNet-128 configurations. bf16, C = 64 or 96 input channels.
//
// Why a kernel of its own: with 8 output channels the problem is pure streaming (805 MB in, 67 MB out at batch 256: 0.11 ms of HBM time,
// 22 GFLOP), but the halo kernel (conv_v3.h <32, 8, 1, 512>) is built around a cout tile worth streaming weights for: it reloads a 4 KB
// weight tile and crosses a barrier for each of its 18 (slice, tap) steps with 4-8 MFMAs per wave in between, and its single patch buffer
// (100 KB) makes both slice loads of a tile a full stop -- 0.45 ms per launch = 48 TFLOP/s (profiles/r03_conv_layer_table_a_*.txt), six
// launches per BigGAN-128 step.
//
// Here nothing is re-staged:
//   * the WEIGHTS live in registers for the whole kernel: the A fragment of (tap t, 16-channel sub-step ks) is 4 VGPRs per lane,
//     9 * C / 16 fragments = 216 registers at C = 96 -- one wave per SIMD (4 waves per workgroup, one workgroup per CU) has 512
//   * a workgroup walks DOWN a strip of image rows. Image rows are staged once each, by LDS-DMA, into a ring of five row buffers
//     ([1 zero pixel][128 pixels][1 zero pixel] x (2 C + 16) bytes: the 16-byte pad makes the pixel pitch an odd number of 16-byte units, so
//     the 16 lanes of a ds_read_b128 service group -- 16 consecutive pixels, same chunk -- fall into 16 different bank groups); output row r
//     reads rows r - 1, r, r + 1 at column offsets -1, 0, +1: the nine taps are nine immediate offsets from three row bases.
//     Rows above / below the image and the pad chunk come in as zeros through out-of-range DMA offsets; the two pad pixels are zeroed once
//   * one barrier per output row. Step j: wait for everything but the newest row in flight (s_waitcnt vmcnt(PPW)), barrier, store the
//     PREVIOUS row's outputs (held in registers: each lane owns 4 couts of one pixel = one 8-byte store; a wave's stores are contiguous),
//     then 9 * C / 16 x (ds_read_b128, MFMA 32x32x16) against the resident weights, the PPW pieces of row j + 4 behind the first MFMAs. The DMA of a row has two full steps
//     to land. The wait count is exact under either ordering model of stores against loads: loads retire in order among themselves, so
//     "at most PPW outstanding" leaves only pieces of the newest row whether or not the store has retired
//   * per step and CU: 54 MFMAs x 32 cycles per SIMD (1728 cycles; 3/4 of each 32-cout MFMA is padding at 8 couts) against 4 waves x 54
//     conflict-free ds_read_b128 = 864 LDS cycles, and 24.6 KB of HBM per 0.75 us step is ~8 TB/s over 256 CUs: matrix pipe and HBM run
//     out together, ~0.11 ms per launch at batch 256 by that count; measured 0.195 ms (4.5 TB/s), see the note at the DMA issue below
//   * epilogue in registers: scale, bias, ReLU, bf16 pack (no mask / residual / pooling: the launcher leaves those to conv_v3.h)
#pragma once
#include "conv_v2.h"

struct ConvRsParams {
  const bf16_t* x; const bf16_t* w;
  int H;                  // image rows (the width is 128)
  int ldx;                // pixel pitch of x (elements)
  int I, K;               // output channels (multiple of 8, <= 32), K = 9 C
  int SH;                 // output rows per workgroup (divides H)
  int spi;                // strips per image = H / SH
  unsigned xbytes;
};

template <int NKT, bool RELU>        // NKT: 16-channel sub-steps per tap (C = 16 NKT); RELU: ReLU on load
__global__ __launch_bounds__(256) void sg_conv_rs_kernel(ConvRsParams p, Epilogue<bf16_t> epi) {
  constexpr int W = 128, C = 16 * NKT, PITCH = 2 * C + 16, ROWB = (W + 2) * PITCH, NRING = 5;
  constexpr int NPIECE = W * PITCH / 1024, PPW = (NPIECE + 3) / 4;
  static_assert((W * PITCH) % 1024 == 0, "a row of pixels is a whole number of 1 KB DMA pieces");
  static_assert((PITCH / 16) % 2 == 1, "pixel pitch = odd number of 16-byte units");
  extern __shared__ __attribute__((aligned(16))) char smem[];      // ring[NRING][ROWB] | dump[1024]
  char* const dump = smem + NRING * ROWB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fhi = lane >> 5;
  const int n = blockIdx.x / p.spi, r0 = (blockIdx.x - n * p.spi) * p.SH;

  // the pad pixels either side of every ring row (never written by the DMA)
  for (int i = tid; i < NRING * 2 * (PITCH / 4); i += 256) {
    const int row = i / (2 * (PITCH / 4)), rem = i - row * (2 * (PITCH / 4));
    const int side = rem / (PITCH / 4), wd = rem - side * (PITCH / 4);
    ((uint32_t*)(smem + row * ROWB + side * (W + 1) * PITCH))[wd] = 0u;
  }

  // ---- row DMA: piece q = wave + 4 i covers bytes [q KB, (q + 1) KB) of the row's pixel area; lane -> (pixel, 16-byte chunk) is the same for every row
  const auto rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
  unsigned poff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; i++) {
    const int q = wave + 4 * i;
    const int b = q * 1024 + lane * 16;
    const int pix = b / PITCH, ch = (b - pix * PITCH) >> 4;
    poff[i] = (q < NPIECE && ch < C / 8) ? (unsigned)((pix * p.ldx + ch * 8) * 2) : 0x80000000u;
  }
  const unsigned rowbytes = (unsigned)(W * p.ldx * 2);
  // piece i of image row r0 - 1 + rho into ring slot rho % NRING
  auto issue_piece = [&](int rho, int i) {
    const int r = r0 - 1 + rho;
    const bool rv = ((unsigned)r < (unsigned)p.H) && (rho <= p.SH + 1);      // outside the image / behind the strip: zeros (into a free slot)
    const unsigned rbase = (unsigned)(n * p.H + r) * rowbytes;
    char* const slot = smem + (rho % NRING) * ROWB + PITCH;
    const int q = wave + 4 * i;
    unsigned off = poff[i];
    asm volatile("" : "+v"(off));
    off = rv ? off + rbase : 0x80000000u;               // (an invalid lane stays >= 2^31: xbytes < 2^31)
    char* dst = (q < NPIECE) ? slot + q * 1024 : dump;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (sg_lptr_t)dst, 16, (int)off, 0, 0, 0);
  };
  for (int rho = 0; rho < 4; rho++) {
#pragma unroll
    for (int i = 0; i < PPW; i++) issue_piece(rho, i);
  }

  // ---- weights: all 9 * NKT A fragments of this lane's cout row, for the whole kernel ------------------------------------------------
  bf16x8_t wf[9][NKT];
  {
    // every lane loads (rows >= I re-read row 0 and are zeroed by a select): a branch around each load would wait for each one in turn
    const bool wv = frow < p.I;
    const bf16_t* wrow = p.w + (long long)(wv ? frow : 0) * p.K + fhi * 8;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int ks = 0; ks < NKT; ks++) {
        u32x4 v = *(const u32x4*)(wrow + t * C + ks * 16);
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = wv ? v[i] : 0u;
        wf[t][ks] = __builtin_bit_cast(bf16x8_t, v);
      }
  }

  // ---- epilogue constants ---------------------------------------------------------------------------------------------------------
  const int ig = p.I >> 3;                              // 8-cout groups (1 .. 4)
  float al = epi.alpha;
  if (epi.alpha_ptr) al *= *epi.alpha_ptr;
  float bia[4][4];
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int e = 0; e < 4; e++) bia[g][e] = (epi.bias && g < ig) ? epi.bias[8 * g + 4 * fhi + e] : 0.f;
  const bool relu_out = (epi.flags & SG_EPI_RELU) != 0;
  const int col = wave * 32 + frow;
  bf16_t* const obase = (bf16_t*)epi.out + ((long long)(n * p.H + r0) * W + col) * epi.ldo + 4 * fhi;
  const long long ostep = (long long)W * epi.ldo;
  u32x2 outreg[4];
#pragma unroll
  for (int g = 0; g < 4; g++) outreg[g] = (u32x2){0u, 0u};
  auto store_row = [&](int j) {
    bf16_t* o = obase + (long long)j * ostep;
#pragma unroll
    for (int g = 0; g < 4; g++)
      if (g < ig) *(u32x2*)(o + 8 * g) = outreg[g];
  };

  const unsigned lb = (unsigned)(col * PITCH + fhi * 16);      // pixel col - 1 of a ring row (the row starts with the pad pixel)

  __syncthreads();                                      // pad pixels written (also drains the prologue's loads: rows 0..3, weights)

  int s0 = 0;                                           // ring slot of row rho = j
  for (int j = 0; j < p.SH; j++) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    __builtin_amdgcn_s_barrier();                       // rows <= j + 2 have landed for every wave; every wave is done with row j - 1
    if (j > 0) store_row(j - 1);
    int s1 = s0 + 1; if (s1 >= NRING) s1 -= NRING;
    int s2 = s1 + 1; if (s2 >= NRING) s2 -= NRING;
    const unsigned base[3] = {(unsigned)(s0 * ROWB) + lb, (unsigned)(s1 * ROWB) + lb, (unsigned)(s2 * ROWB) + lb};
    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    // two accumulators (even / odd MFMA): no dependent back-to-back MFMAs on the single wave of the SIMD; fragments one tap ahead
    bf16x8_t qf[2][NKT];
    auto load_tap = [&](int t, int slot) {
      const char* ps = smem + base[t / 3] + (t % 3) * PITCH;
#pragma unroll
      for (int ks = 0; ks < NKT; ks++) {
        u32x4 v = *(const u32x4*)(ps + ks * 32);
        if (RELU) v = relu16<bf16_t>(v);
        qf[slot][ks] = __builtin_bit_cast(bf16x8_t, v);
      }
    };
    load_tap(0, 0);
#pragma unroll
    for (int t = 0; t < 9; t++) {
      if (t < 8) load_tap(t + 1, (t + 1) & 1);
#pragma unroll
      for (int ks = 0; ks < NKT; ks++) {
        acc[ks & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][ks], qf[t & 1][ks], acc[ks & 1], 0, 0, 0);
        // the PPW pieces of row j + 4, one behind each of the step's first MFMAs (a piece costs 60-180 issue cycles). Measured against the
        // first version, which issued all of them in front of the step's first MFMA: 0.198 -> 0.195 ms per launch at batch 256 (r3l / r3m) --
        // nothing; the kernel waits on the memory system (4.5 TB/s, HBM traffic by PMC = 1.0003 x the algorithmic bytes), not on issue slots
        if (t * NKT + ks < PPW) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(j + 4, t * NKT + ks);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);                // keep the software pipeline as written: fragments of tap t + 1, then the MFMAs of tap t
    }
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (g < ig) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          v[e] = (acc[0][4 * g + e] + acc[1][4 * g + e]) * al + bia[g][e];
          if (relu_out) v[e] = fmaxf(v[e], 0.f);
        }
        outreg[g][0] = pack2bf(v[0], v[1]);
        outreg[g][1] = pack2bf(v[2], v[3]);
      }
    }
    s0 = s1;
  }
  store_row(p.SH - 1);
}

// launcher: 0 = launched, -1 = not launched
template <int NKT, bool RELU>
static inline int sg_launch_conv_rs(const ConvRsParams& p, const Epilogue<bf16_t>& e, int nstrips, hipStream_t st) {
  constexpr int PITCH = 2 * 16 * NKT + 16, ROWB = 130 * PITCH;
  const int lds = 5 * ROWB + 1024;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)sg_conv_rs_kernel<NKT, RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
    attr_done = true;
  }
  hipLaunchKernelGGL((sg_conv_rs_kernel<NKT, RELU>), dim3(nstrips), dim3(256), lds, st, p, e);
  return 0;
}
