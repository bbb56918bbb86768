// conv_rs96.h -- row-streaming 96 -> 96 channel 3x3 kernel. Written at the end of round 3 without GPU time; first executed in round 4
// (tools/sessions/r4a.sh): parity green against CPU fp64 and the halo kernel (tests/test_conv_v2_gpu.py), plain variant 0.986 -> 0.811 ms per launch
// at batch 256, pooling variant slower than the halo kernel (profiles/r04_conv_rs96_first_run.txt). Its index arithmetic is also replayed lane by
// lane on the CPU (tools/rs96_model.py, tests/test_host_cpu.py).
//
// The row-streaming structure of conv_rs.h for the 96 -> 96 channel 3x3 convolutions at 128 x 128 (bf16): the layers that carry most of the
// FLOPs of the BigGAN-128 discriminator's first block and of the generator's last one, and that the halo kernels run at 0.29-0.38 of the MFMA
// peak (732-945 TFLOP/s in profiles/r03_conv_layer_table_a_swz_parity_on.txt against 1200+ on the deep layers): K = 864 is short, so a tile's
// prologue / epilogue / per-tap weight traffic never amortise.
//
// Here the WEIGHTS never move: the register file of a CU holds 512 KB, the 96 x 864 bf16 weights are 166 KB.
//   * 4 waves, one per SIMD. Waves 0..2 are CONSUMERS: wave w owns cout tile w (32 couts) and keeps its 54 A fragments (9 taps x 6 sub-steps of
//     16 channels, 4 VGPRs each = 216 registers) for the whole kernel; per output row it runs 4 pixel tiles x 54 = 216 MFMAs 32x32x16 against B
//     fragments read from a ring of image rows in LDS (one ds_read_b128 per MFMA, an A fragment serves 4 MFMAs in a row). Wave 3 is the
//     PRODUCER: it issues the LDS-DMA of the row four steps ahead (26 pieces of 1 KB) and waits for the row two steps ahead -- the consumers
//     never touch the vector-memory counter, so their waits are LDS waits only. 3 of 4 matrix pipes busy: ceiling 0.75 of peak.
//   * ring of five row buffers as in conv_rs.h ([zero pixel][128 pixels][zero pixel] x 208 B), one s_barrier per output row
//   * epilogue in registers, straight to global memory: a lane owns 4 couts of one pixel per 8-cout group = 8-byte stores, the four groups of a
//     wave fill one 64-byte line per pixel. POOL: the accumulators simply keep running over an even / odd row pair, the horizontal pair is one
//     DPP add, even lanes store (out = 0.25-scaled by the caller's alpha, as sg_conv_epilogue does)
//   * supported: bias, scale, ReLU on load, ReLU on store, 2x2 average pooling. Mask / residual / upsample-on-load stay with conv_v4.h.
#pragma once
#include "conv_v2.h"

struct ConvRs96Params {
  const bf16_t* x; const bf16_t* w;
  int H;                  // image rows (the width is 128)
  int ldx;                // pixel pitch of x (elements)
  int K;                  // 9 * 96
  int SH;                 // output rows per workgroup (divides H; even with pooling)
  int spi;                // strips per image
  unsigned xbytes;
};

template <bool RELU, bool POOL>
__global__ __launch_bounds__(256) void sg_conv_rs96_kernel(ConvRs96Params p, Epilogue<bf16_t> epi) {
  constexpr int W = 128, C = 96, NKT = 6, PITCH = 2 * C + 16, ROWB = (W + 2) * PITCH, NRING = 5;
  constexpr int NPIECE = W * PITCH / 1024;               // 26
  extern __shared__ __attribute__((aligned(16))) char smem[];      // ring[NRING][ROWB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fhi = lane >> 5;
  const int n = blockIdx.x / p.spi, r0 = (blockIdx.x - n * p.spi) * p.SH;

  for (int i = tid; i < NRING * 2 * (PITCH / 4); i += 256) {       // the pad pixels either side of every ring row
    const int row = i / (2 * (PITCH / 4)), rem = i - row * (2 * (PITCH / 4));
    const int side = rem / (PITCH / 4), wd = rem - side * (PITCH / 4);
    ((uint32_t*)(smem + row * ROWB + side * (W + 1) * PITCH))[wd] = 0u;
  }

  if (wave == 3) {
    // ---- producer: all 26 pieces of a row; lane -> (pixel, 16-byte chunk) of piece q is the same for every row ----------------------------
    const auto rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
    unsigned poff[NPIECE];
#pragma unroll
    for (int q = 0; q < NPIECE; q++) {
      const int b = q * 1024 + lane * 16;
      const int pix = b / PITCH, ch = (b - pix * PITCH) >> 4;
      poff[q] = (ch < C / 8) ? (unsigned)((pix * p.ldx + ch * 8) * 2) : 0x80000000u;
    }
    const unsigned rowbytes = (unsigned)(W * p.ldx * 2);
    auto issue_row = [&](int rho) {                      // image row r0 - 1 + rho into ring slot rho % NRING
      const int r = r0 - 1 + rho;
      const bool rv = ((unsigned)r < (unsigned)p.H) && (rho <= p.SH + 1);
      const unsigned rbase = (unsigned)(n * p.H + r) * rowbytes;
      char* const slot = smem + (rho % NRING) * ROWB + PITCH;
#pragma unroll
      for (int q = 0; q < NPIECE; q++) {
        unsigned off = poff[q];
        asm volatile("" : "+v"(off));
        off = rv ? off + rbase : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (sg_lptr_t)(slot + q * 1024), 16, (int)off, 0, 0, 0);
      }
    };
    for (int rho = 0; rho < 4; rho++) issue_row(rho);
    __syncthreads();                                     // (drains rows 0..3)
    for (int j = 0; j < p.SH; j++) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");      // everything but the newest row (j + 3) has landed
      __builtin_amdgcn_s_barrier();                      // rows <= j + 2 visible to the consumers; they are done with row j - 1
      issue_row(j + 4);
    }
    return;
  }

  // ---- consumers -----------------------------------------------------------------------------------------------------------------------
  const int co0 = 32 * wave;
  bf16x8_t wf[9][NKT];
  {
    const bf16_t* wrow = p.w + (long long)(co0 + frow) * p.K + fhi * 8;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int ks = 0; ks < NKT; ks++) {
        const u32x4 v = *(const u32x4*)(wrow + t * C + ks * 16);
        wf[t][ks] = __builtin_bit_cast(bf16x8_t, v);
      }
  }
  float al = epi.alpha;
  if (epi.alpha_ptr) al *= *epi.alpha_ptr;
  float bia[4][4];
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int e = 0; e < 4; e++) bia[g][e] = epi.bias ? epi.bias[co0 + 8 * g + 4 * fhi + e] : 0.f;
  const bool relu_out = (epi.flags & SG_EPI_RELU) != 0;
  const unsigned lb = (unsigned)(frow * PITCH + fhi * 16);          // pixel tile 0, column offset -1 (the row starts with the pad pixel)
  const int Wout = POOL ? W / 2 : W;
  // output pixel of this lane in pixel tile pt: column pt * 32 + frow (pooled: (pt * 32 + frow) / 2, even lanes only)
  bf16_t* const obase = (bf16_t*)epi.out + ((long long)(n * (POOL ? p.H / 2 : p.H) + (POOL ? r0 / 2 : r0)) * Wout + (POOL ? frow / 2 : frow)) * epi.ldo + co0 + 4 * fhi;
  const long long ostep = (long long)Wout * epi.ldo;

  f32x16 acc[4];
  __syncthreads();                                       // pad pixels written, rows 0..3 landed (the producer waited for them)

  int s0 = 0;
  for (int j = 0; j < p.SH; j++) {
    __builtin_amdgcn_s_barrier();
    int s1 = s0 + 1; if (s1 >= NRING) s1 -= NRING;
    int s2 = s1 + 1; if (s2 >= NRING) s2 -= NRING;
    const unsigned base[3] = {(unsigned)(s0 * ROWB) + lb, (unsigned)(s1 * ROWB) + lb, (unsigned)(s2 * ROWB) + lb};
    if (!POOL || (j & 1) == 0) {
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    }
    bf16x8_t qf[2][4];                                   // B fragments of one (tap, sub-step): four pixel tiles; one sub-step ahead
    auto load_step = [&](int t, int ks, int slot) {
      const char* ps = smem + base[t / 3] + (t % 3) * PITCH + ks * 32;
#pragma unroll
      for (int pt = 0; pt < 4; pt++) {
        u32x4 v = *(const u32x4*)(ps + pt * 32 * PITCH);
        if (RELU) v = relu16<bf16_t>(v);
        qf[slot][pt] = __builtin_bit_cast(bf16x8_t, v);
      }
    };
    load_step(0, 0, 0);
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int ks = 0; ks < NKT; ks++) {
        const int s = t * NKT + ks;
        if (s + 1 < 9 * NKT) load_step((s + 1) / NKT, (s + 1) % NKT, (s + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);               // (left alone the scheduler sinks these reads behind three of the four MFMAs: one MFMA of cover)
#pragma unroll
        for (int pt = 0; pt < 4; pt++)
          acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][ks], qf[s & 1][pt], acc[pt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);               // fragments of sub-step s + 1, then the four MFMAs of sub-step s
      }
    if (!POOL || (j & 1) == 1) {
      bf16_t* o = obase + (long long)(POOL ? (j >> 1) : j) * ostep;
#pragma unroll
      for (int pt = 0; pt < 4; pt++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float a = acc[pt][4 * g + e];
            if (POOL)                                    // + the horizontal neighbour (quad_perm [1,0,3,2]): lanes 2k, 2k + 1 are pixels 2k, 2k + 1
              a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true));
            v[e] = a * al + bia[g][e];
            if (relu_out) v[e] = fmaxf(v[e], 0.f);
          }
          u32x2 t2;
          t2[0] = pack2bf(v[0], v[1]);
          t2[1] = pack2bf(v[2], v[3]);
          if (!POOL || (lane & 1) == 0) *(u32x2*)(o + (long long)(POOL ? pt * 16 : pt * 32) * epi.ldo + 8 * g) = t2;
        }
      }
    }
    s0 = s1;
  }
}

template <bool RELU, bool POOL>
static inline int sg_launch_conv_rs96(const ConvRs96Params& p, const Epilogue<bf16_t>& e, int nstrips, hipStream_t st) {
  const int lds = 5 * 130 * 208;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)sg_conv_rs96_kernel<RELU, POOL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
    attr_done = true;
  }
  hipLaunchKernelGGL((sg_conv_rs96_kernel<RELU, POOL>), dim3(nstrips), dim3(256), lds, st, p, e);
  return 0;
}
