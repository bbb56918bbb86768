// wgrad_v3.h -- "halo" weight gradient for the 3x3 layers (bf16, stride 1, pad 1, W >= 8): first written for the layers whose activations
// stream from HBM (96 / 192 / 384-channel blocks of BigGAN at 32^2 .. 128^2, the 64 .. 256-channel blocks of the ResNet GANs), then
// extended to the 16^2 and 8^2 layers (chunk = 4 / 8 whole image rows).
//
//   dW[co][tap][ci] (+)= alpha * sum_pix dy'[pix][co] * x'[pix + tap][ci]
//
// wgrad_v2.h stages an im2col tile per k-step: every one of the nine taps re-fetches the same activation pixels from L2 (85 FLOP per
// staged byte) and the 864 x 96 problem of the 96-channel layers fills 63 % of its 256 x 128 MFMA tile -- 500-550 TFLOP/s on those layers
// (profiles/r02 layer table), MFMA busy 0.36-0.46 with SQ_WAIT_ANY 0.37-0.46 (profiles/r02_conv_sq_counters_baseline.txt). Here:
//   * a workgroup owns ONE 32-channel slice of the input and ONE tile of 32 * NB output channels (NB = 3 or 2), for ALL nine taps: the
//     result is [9 taps x 32 ci] x [32 NB co] -- an exact fit for every channel count that is a multiple of 32 / 32 NB;
//   * it walks chunks of 64 output pixels (64 x 1 or 32 x 2): the input slice is staged ONCE per chunk as a raster patch with a one-pixel
//     halo ([rows + 2][cols + 2] pixels x 64 B, zeros outside the image straight from the buffer descriptor), the nine taps are nine
//     shifted transpose reads of that patch (ds_read_b64_tr_b16, pixel-major -> k-contiguous): ~140 FLOP per staged byte;
//   * the 9 NB (tap, cout block) products of 32 x 32 are dealt to 4 waves as  wave w: taps 2w, 2w + 1 with every cout block, plus
//     (tap 8, cout block w) for w < NB  -- 7, 7, 7, 6 MFMAs per 16 pixels for NB = 3. Which taps a wave owns is only an LDS address, the
//     code is the same for all waves. 2 NB + 1 accumulators = 112 registers: three workgroups (12 waves, 3 per SIMD) share a CU, each
//     with a double-buffered 2 x 25 KB staging area; one barrier per chunk.
// Partial tiles go to the deterministic two-stage reduction (k_splitk_reduce), like wgrad_v2.h.
#pragma once
#include "gemm_core.h"
#include "conv_v2.h"

struct WgradV3Params {
  const bf16_t* x; const bf16_t* dy;
  int xHs, xWs, ldx, x_up, x_relu;
  int gHs, gWs, ldg, g_up;
  int N, H, W;                   // output-pixel raster (= dy's logical extent = x's logical extent: stride 1, pad 1)
  int C, Cout;                   // full channel counts (dW is [Cout][9][C])
  int nci, nco;                  // channel slices of 32, cout tiles of 32 * NB
  int nchunk;                    // N * H * W / 64
  int splits;                    // workgroups per (ci slice, co tile); chunk c goes to split c % splits
  unsigned xbytes, gbytes;
  float* out; long long split_stride;   // partial s at out + s * split_stride
  long long bias_off;            // >= 0: the workgroups of channel slice 0 also write sum_pix dy[pix][co] * bias_scale to out[s * split_stride + bias_off + co]
  float bias_scale;              // 0.25 when dy is the pooled gradient read four times (g_up), else 1
  float alpha; const float* alpha_ptr;
};

template <int OFF> __device__ __forceinline__ void w3_tr_read(unsigned addr, u32x2& v) {      // asm: see wgrad_v2.h (no compiler vmcnt(0) in front of it)
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF));
}
typedef short w3_s16x2 __attribute__((ext_vector_type(2)));
// one k-step (16 pixels) of a chunk: 2 NB + 1 MFMAs from 3 activation fragments (taps t0, t1, 8) and NB + 1 gradient fragments
template <int NB, int WC, int KS>
__device__ __forceinline__ void w3_kstep(f32x16* acc, unsigned a0, unsigned a1, unsigned a2, unsigned b0, unsigned bx, uint32_t relu_bound, bool extra,
                                         float* csum, bool do_csum) {
  constexpr int PW = WC + 2, GPITCH = NB * 64;
  // patch byte offset of pixels KS * 16 .. of the chunk raster. WC == 4 (round 4): a chunk is FOUR whole 4 x 4 images, each with its own 6 x 6
  // halo patch; a 16-pixel k-step is one image, and the second half of a fragment (+ 4 pixels) is the next image row
  constexpr int KX = WC == 4 ? KS * 36 * 64 : ((KS * 16) / WC) * PW * 64 + ((KS * 16) % WC) * 64;
  constexpr int A2 = WC == 4 ? PW * 64 : 256;
  constexpr int KG = KS * 16 * GPITCH;
  u32x2 al[3], ah[3], bl[NB], bh[NB], xl, xh;
  w3_tr_read<KX>(a0, al[0]); w3_tr_read<KX + A2>(a0, ah[0]);
  w3_tr_read<KX>(a1, al[1]); w3_tr_read<KX + A2>(a1, ah[1]);
  w3_tr_read<KX>(a2, al[2]); w3_tr_read<KX + A2>(a2, ah[2]);
  w3_tr_read<KG>(b0, bl[0]); w3_tr_read<KG + 4 * GPITCH>(b0, bh[0]);
  w3_tr_read<KG + 64>(b0, bl[1]); w3_tr_read<KG + 64 + 4 * GPITCH>(b0, bh[1]);
  if constexpr (NB == 3) { w3_tr_read<KG + 128>(b0, bl[2]); w3_tr_read<KG + 128 + 4 * GPITCH>(b0, bh[2]); }
  w3_tr_read<KG>(bx, xl); w3_tr_read<KG + 4 * GPITCH>(bx, xh);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  bf16x8_t af[3], bf[NB], xf;
#pragma unroll
  for (int s = 0; s < 3; s++) {
    asm volatile("" : "+v"(al[s]), "+v"(ah[s]));
    u32x4 v = {al[s][0], al[s][1], ah[s][0], ah[s][1]};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t xq = v[q];
      w3_s16x2 x2 = __builtin_bit_cast(w3_s16x2, xq);
      x2 = __builtin_elementwise_max(x2, __builtin_bit_cast(w3_s16x2, relu_bound));
      v[q] = __builtin_bit_cast(uint32_t, x2);
    }
    af[s] = __builtin_bit_cast(bf16x8_t, v);
  }
#pragma unroll
  for (int b = 0; b < NB; b++) {
    asm volatile("" : "+v"(bl[b]), "+v"(bh[b]));
    u32x4 v = {bl[b][0], bl[b][1], bh[b][0], bh[b][1]};
    bf[b] = __builtin_bit_cast(bf16x8_t, v);
  }
  { asm volatile("" : "+v"(xl), "+v"(xh)); u32x4 v = {xl[0], xl[1], xh[0], xh[1]}; xf = __builtin_bit_cast(bf16x8_t, v); }
  if (do_csum) {       // bias gradient (one wave of the slice-0 workgroups): this lane's 8 pixels of cout b * 32 + (lane & 31)
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const u32x4 v = __builtin_bit_cast(u32x4, bf[b]);
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 4; q++) t += __uint_as_float(v[q] << 16) + __uint_as_float(v[q] & 0xffff0000u);
      csum[b] += t;
    }
  }
#pragma unroll
  for (int b = 0; b < NB; b++) {
    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[b], acc[b], 0, 0, 0);
    acc[NB + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[b], acc[NB + b], 0, 0, 0);
  }
  if (extra) acc[2 * NB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], xf, acc[2 * NB], 0, 0, 0);
}

// NB = 32-wide cout blocks per tile (2 or 3), WC = chunk width in pixels: 64 = one row segment, 32 / 16 / 8 = 2 / 4 / 8 whole rows of a 32 / 16 / 8-wide image
template <int NB, int WC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void sg_wgrad_v3_kernel(WgradV3Params p) {
  constexpr int NIMG = WC == 4 ? 4 : 1;             // images per chunk (WC == 4: four whole 4 x 4 images)
  constexpr int RC = 64 / WC / NIMG;                // image rows per chunk (per image part)
  constexpr int PW = WC + 2, PR = RC + 2;           // patch extent in pixels (per image part)
  constexpr int XBYTES = NIMG * PR * PW * 64;       // patch: 64 B (32 channels) per pixel
  constexpr int NPX = (XBYTES + 1023) / 1024;       // LDS-DMA pieces of the patch
  constexpr int GPITCH = NB * 64;
  constexpr int NPG = 64 * GPITCH / 1024;           // pieces of the dy tile (4 NB)
  constexpr int GOFF = NPX * 1024;
  constexpr int BUF = GOFF + NPG * 1024;            // one staging buffer
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
  const auto rsg = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.gbytes, 0x00020000);
  // hardware deals consecutive workgroup ids round-robin to the 8 XCDs: renumber so that one XCD (one L2) runs a contiguous range of
  // logical ids -- the channel slices of a split, which read the same dy pixels, and neighbouring splits, which share halo rows.
  int bid = blockIdx.x;
  { const int G = gridDim.x; if ((G & 7) == 0) bid = (bid & 7) * (G >> 3) + (bid >> 3); }
  const int tiles = p.nci * p.nco;
  const int split = bid / tiles;
  const int tl = bid - split * tiles;
  const int cis = tl % p.nci, cot = tl / p.nci;
  const int ci0 = cis * 32, co0 = cot * (32 * NB);
  const int cpr = WC == 4 ? 1 : p.W / WC;           // chunks per image-row group
  const int cpi = WC == 4 ? 1 : (p.H / RC) * cpr;   // chunks per image (WC == 4: a chunk is four images)

  auto issue = [&](int c, int buf) {
    int n, h0, w0;
    if (WC == 4) { n = 4 * c; h0 = 0; w0 = 0; }
    else { n = c / cpi; const int rem = c - n * cpi; const int rg = rem / cpr, cx = rem - rg * cpr; h0 = rg * RC; w0 = cx * WC; }
    char* base = smem + buf * BUF;
    for (int j = wave; j < NPX; j += 4) {
      const int o = j * 1024 + lane * 16;
      const int pp0 = o >> 6, cb = o & 63;
      const int kimg = pp0 / (PR * PW), pp = pp0 - kimg * (PR * PW);
      const int pr = pp / PW, pc = pp - pr * PW;
      int hh = h0 + pr - 1, ww = w0 + pc - 1;
      const bool ok = (kimg < NIMG) & ((unsigned)hh < (unsigned)p.H) & ((unsigned)ww < (unsigned)p.W);
      if (p.x_up) { hh >>= 1; ww >>= 1; }
      unsigned off = (((unsigned)((n + kimg) * p.xHs + hh) * (unsigned)p.xWs + (unsigned)ww) * (unsigned)p.ldx + (unsigned)ci0) * 2u + (unsigned)cb;
      off = ok ? off : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (sg_lptr_t)(base + j * 1024), 16, (int)off, 0, 0, 0);
    }
    for (int j = wave; j < NPG; j += 4) {
      const int o = j * 1024 + lane * 16;
      const int px = o / GPITCH, cb = o - px * GPITCH;
      int kimg = 0, cr, cc;
      if (WC == 4) { kimg = px >> 4; cr = (px >> 2) & 3; cc = px & 3; } else { cr = px / WC; cc = px - cr * WC; }
      int hh = h0 + cr, ww = w0 + cc;
      if (p.g_up) { hh >>= 1; ww >>= 1; }
      const unsigned off = (((unsigned)((n + kimg) * p.gHs + hh) * (unsigned)p.gWs + (unsigned)ww) * (unsigned)p.ldg + (unsigned)co0) * 2u + (unsigned)cb;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, (sg_lptr_t)(base + GOFF + j * 1024), 16, (int)off, 0, 0, 0);
    }
  };

  // loop-invariant fragment addresses. One transpose read = 4 pixel rows x 16 channels per 16-lane group; lane result: channel
  // 16 (g16 & 1) + 4 (t & 3) .. + 3 of the block, pixel 8 (g16 >> 1) + (t >> 2) (second read: + 4 pixels).
  const int g16 = lane >> 4, t16 = lane & 15;
  const int prow = 8 * (g16 >> 1) + (t16 >> 2);
  const int csub = 16 * (g16 & 1) + 4 * (t16 & 3);
  const unsigned sb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
  // taps of this wave: 2w, 2w + 1 and 8; tap t = (dr, ds) is the patch pixel (row + dr, col + ds) of output pixel (row, col)
  const int t0 = 2 * wave, t1 = 2 * wave + 1;
  // (a 16-pixel k-step is part of one image row for WC >= 16 and two whole rows for WC = 8: the lane's pixel is (prow / WC, prow % WC) of it)
  const int ppix = (prow / WC) * PW + (prow % WC);
  const unsigned a0 = sb + (((t0 / 3) * PW + (t0 % 3)) + ppix) * 64 + csub * 2;
  const unsigned a1 = sb + (((t1 / 3) * PW + (t1 % 3)) + ppix) * 64 + csub * 2;
  const unsigned a2 = sb + ((2 * PW + 2) + ppix) * 64 + csub * 2;
  const unsigned b0 = sb + GOFF + prow * GPITCH + csub * 2;
  const bool extra = wave < NB;                                      // (tap 8, cout block `wave`)
  const unsigned bx = b0 + (extra ? wave : 0) * 64;
  const uint32_t relu_bound = p.x_relu ? 0u : 0x80008000u;           // signed 16-bit max with 0 = ReLU of bf16, with -32768 = identity

  f32x16 acc[2 * NB + 1];
#pragma unroll
  for (int s = 0; s < 2 * NB + 1; s++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[s][r] = 0.f;

  // bias gradient: wave 3 (the wave without a seventh product for NB = 3) of the workgroups that own channel slice 0
  const bool do_csum = p.bias_off >= 0 && cis == 0 && wave == 3;
  float csum[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) csum[b] = 0.f;

  int buf = 0;
  if (split < p.nchunk) issue(split, 0);
  for (int c = split; c < p.nchunk; c += p.splits) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // chunk c has landed everywhere; every wave is done with the other buffer
    if (c + p.splits < p.nchunk) issue(c + p.splits, buf ^ 1);
    const unsigned bo = (unsigned)(buf * BUF);
    w3_kstep<NB, WC, 0>(acc, a0 + bo, a1 + bo, a2 + bo, b0 + bo, bx + bo, relu_bound, extra, csum, do_csum);
    w3_kstep<NB, WC, 1>(acc, a0 + bo, a1 + bo, a2 + bo, b0 + bo, bx + bo, relu_bound, extra, csum, do_csum);
    w3_kstep<NB, WC, 2>(acc, a0 + bo, a1 + bo, a2 + bo, b0 + bo, bx + bo, relu_bound, extra, csum, do_csum);
    w3_kstep<NB, WC, 3>(acc, a0 + bo, a1 + bo, a2 + bo, b0 + bo, bx + bo, relu_bound, extra, csum, do_csum);
    buf ^= 1;
  }

  float al = p.alpha;
  if (p.alpha_ptr) al *= *p.alpha_ptr;
  float* out = p.out + (long long)split * p.split_stride;
  auto store = [&](const f32x16& a, int tap, int b) {
    const int co = co0 + b * 32 + (lane & 31);
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
      const int ci = ci0 + 8 * g4 + 4 * (lane >> 5);
      f32x4 v = {a[4 * g4 + 0] * al, a[4 * g4 + 1] * al, a[4 * g4 + 2] * al, a[4 * g4 + 3] * al};
      *(f32x4*)(out + ((long long)co * 9 + tap) * p.C + ci) = v;
    }
  };
#pragma unroll
  for (int b = 0; b < NB; b++) { store(acc[b], t0, b); store(acc[NB + b], t1, b); }
  if (extra) store(acc[2 * NB], 8, wave);
  if (do_csum) {
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const float t = csum[b] + __shfl_xor(csum[b], 32, 64);        // the two k-halves of the wave hold different pixels of the same cout
      if (lane < 32) out[p.bias_off + co0 + b * 32 + lane] = t * p.bias_scale;
    }
  }
}

template <int NB, int WC>
static inline int sg_launch_wgrad_v3_t(const WgradV3Params& p, hipStream_t st) {
  constexpr int NIMG = WC == 4 ? 4 : 1, RC = 64 / WC / NIMG, XB = NIMG * (RC + 2) * (WC + 2) * 64;
  constexpr int LDS = 2 * (((XB + 1023) / 1024) * 1024 + 64 * NB * 64);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)sg_wgrad_v3_kernel<NB, WC>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1;
    attr_done = true;
  }
  hipLaunchKernelGGL((sg_wgrad_v3_kernel<NB, WC>), dim3(p.nci * p.nco * p.splits), dim3(256), LDS, st, p);
  return 0;
}
static inline int sg_launch_wgrad_v3(const WgradV3Params& p, int NB, hipStream_t st) {
  const int wc = p.W >= 64 ? 64 : p.W;
  if (NB == 3) {
    switch (wc) { case 64: return sg_launch_wgrad_v3_t<3, 64>(p, st); case 32: return sg_launch_wgrad_v3_t<3, 32>(p, st);
                  case 16: return sg_launch_wgrad_v3_t<3, 16>(p, st); case 8: return sg_launch_wgrad_v3_t<3, 8>(p, st);
                  case 4: return sg_launch_wgrad_v3_t<3, 4>(p, st); }
  } else {
    switch (wc) { case 64: return sg_launch_wgrad_v3_t<2, 64>(p, st); case 32: return sg_launch_wgrad_v3_t<2, 32>(p, st);
                  case 16: return sg_launch_wgrad_v3_t<2, 16>(p, st); case 8: return sg_launch_wgrad_v3_t<2, 8>(p, st);
                  case 4: return sg_launch_wgrad_v3_t<2, 4>(p, st); }
  }
  return -1;
}
