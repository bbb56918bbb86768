// wgrad_v3l.h -- the halo weight-gradient kernel for plain operands (the "lean" rewrite of wgrad_v3.h's loop). Written at the end of round 4 without GPU time,
// from the static instruction mix of that loop (tools/isa_mix.py, profiles/r04_isa_mix.txt: 9.5 vector-ALU instructions per MFMA), checked on the CPU interpreter
// (tests/test_hipemu_cpu.py); first GPU run in round 5 (profiles/r05_variant_ab_layer_tables_b.txt, same box): the wgrad_v3 layers of C3 -8..-12 %, dW bit-identical
// (tests/test_conv_v2_gpu.py). The default since; SG_WGRAD_V3_LEAN=0 selects wgrad_v3.h's kernel everywhere (it still serves the upsampled operands).
//
// Same tiling, staging, fragment addresses and result layout as sg_wgrad_v3_kernel. Three changes:
//   * the LDS-DMA addresses of a lane's pieces are computed once per workgroup, not once per chunk (see the kernel; plain operands only);
//   * ReLU-on-load is a template parameter: the layers without one (every generator layer: the ReLU sits in the batch-norm apply) issue no
//     v_pk_max_i16 at all -- 12 per k-step of 7 MFMAs in the shipped loop, which clamps against -32768 when there is nothing to clamp;
//   * the bias gradient is summed by the waves that own a seventh product, from the gradient fragment they already hold for it, with one
//     v_dot2c_f32_bf16 against (1, 1) per dword: 4 vector instructions per k-step on three waves, instead of 36 (unpack + add of three
//     fragments) on the one wave of the slice-0 workgroups that every chunk barrier then waits for. Summation order differs from the shipped
//     kernel's (pairs first): results agree to fp32 rounding, not bit for bit.
#pragma once
#include "wgrad_v3.h"

typedef __bf16 w3l_bf2 __attribute__((ext_vector_type(2)));
typedef WgradV3Params WgradV3LParams;

// one k-step (16 pixels) of a chunk, as w3_kstep: 2 NB + 1 MFMAs from 3 activation fragments (taps t0, t1, 8) and NB + 1 gradient fragments
template <int NB, int WC, int KS, bool RELU>
__device__ __forceinline__ void w3l_kstep(f32x16* acc, unsigned a0, unsigned a1, unsigned a2, unsigned b0, unsigned bx, bool extra, float& csum, bool do_csum) {
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
    if constexpr (RELU) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t xq = v[q];                                                  // (bit_cast straight from a vector element miscompiles: common.h relu16)
        w3_s16x2 x2 = __builtin_bit_cast(w3_s16x2, xq);
        x2 = __builtin_elementwise_max(x2, __builtin_bit_cast(w3_s16x2, 0u));      // signed 16-bit max with 0 = ReLU of a bf16 pair
        v[q] = __builtin_bit_cast(uint32_t, x2);
      }
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
  if (do_csum) {       // bias gradient of cout block `wave` from the fragment of the seventh product: this lane's 8 pixels of cout (lane & 31)
    const u32x4 v = __builtin_bit_cast(u32x4, xf);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t gq = v[q];
      csum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(w3l_bf2, gq), __builtin_bit_cast(w3l_bf2, 0x3f803f80u), csum, false);
    }
  }
  SG_PRIO_UP();
#pragma unroll
  for (int b = 0; b < NB; b++) {
    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[b], acc[b], 0, 0, 0);
    acc[NB + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[b], acc[NB + b], 0, 0, 0);
  }
  if (extra) acc[2 * NB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], xf, acc[2 * NB], 0, 0, 0);
  SG_PRIO_DOWN();
}

template <int NB, int WC, bool RELU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void sg_wgrad_v3l_kernel(WgradV3LParams p) {
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

  // ---- LDS-DMA addresses (plain operands only: the launcher keeps x_up / g_up problems on the shipped kernel). The shipped kernel derives
  // (image, row, column) of every 16-byte piece from its byte offset again for every chunk: divisions by constants and 32-bit multiplies, ~20
  // vector instructions per piece of which 4-5 run at quarter rate, 6-7 pieces per wave and chunk of 28 MFMAs. The pieces of a lane are the
  // same for every chunk: their byte offset RELATIVE to the chunk's first pixel and their (row, column) displacement are computed ONCE here;
  // per chunk a piece costs two adds, two compares and a select.
  constexpr int NIX = (NPX + 3) / 4;                // x pieces per wave (piece j = wave + 4 i)
  unsigned xrel[NIX]; int xrc[NIX];         // relative byte offset; (row displacement << 16) | (column displacement & 0xffff)
#pragma unroll
  for (int i = 0; i < NIX; i++) {
    const int j = wave + 4 * i;
    const int o = j * 1024 + lane * 16;
    const int pp0 = o >> 6, cb = o & 63;
    const int kimg = pp0 / (PR * PW), pp = pp0 - kimg * (PR * PW);
    const int pr = pp / PW, pc = pp - pr * PW;
    const int dr = pr - 1, dc = pc - 1;
    xrel[i] = (unsigned)(((kimg * p.H + dr) * p.W + dc) * p.ldx * 2 + cb);
    const bool inside = (j < NPX) & (kimg < NIMG);
    xrc[i] = ((inside ? dr : -0x4000) << 16) | (dc & 0xffff);      // (a piece beyond the patch: a row that is never inside [0, H))
  }
  unsigned grel[NB];                                // dy pieces: j = wave + 4 i, i < NB (NPG = 4 NB), always inside the tensor
#pragma unroll
  for (int i = 0; i < NB; i++) {
    const int o = (wave + 4 * i) * 1024 + lane * 16;
    const int px = o / GPITCH, cb = o - px * GPITCH;
    int kimg = 0, cr, cc;
    if (WC == 4) { kimg = px >> 4; cr = (px >> 2) & 3; cc = px & 3; } else { cr = px / WC; cc = px - cr * WC; }
    grel[i] = (unsigned)(((kimg * p.H + cr) * p.W + cc) * p.ldg * 2 + cb);
  }

  auto issue = [&](int c, int buf) {
    int n, h0, w0;
    if (WC == 4) { n = 4 * c; h0 = 0; w0 = 0; }
    else { n = c / cpi; const int rem = c - n * cpi; const int rg = rem / cpr, cx = rem - rg * cpr; h0 = rg * RC; w0 = cx * WC; }
    char* base = smem + buf * BUF;
    const unsigned pix0 = (unsigned)(n * p.H + h0) * (unsigned)p.W + (unsigned)w0;       // the chunk's first pixel (wave-uniform: scalar arithmetic)
    const unsigned bx = pix0 * (unsigned)p.ldx * 2u + (unsigned)ci0 * 2u;
    const unsigned bg = pix0 * (unsigned)p.ldg * 2u + (unsigned)co0 * 2u;
#pragma unroll
    for (int i = 0; i < NIX; i++) {
      const int j = wave + 4 * i;
      if (j < NPX) {
        int rc = xrc[i];
        asm volatile("" : "+v"(rc));                 // (keeps the unpacking inside the loop: hoisted, it would cost the registers the packing saves)
        const bool ok = ((unsigned)(h0 + (rc >> 16)) < (unsigned)p.H) & ((unsigned)(w0 + (int)(short)rc) < (unsigned)p.W);
        const unsigned off = ok ? bx + xrel[i] : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (sg_lptr_t)(base + j * 1024), 16, (int)off, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, (sg_lptr_t)(base + GOFF + (wave + 4 * i) * 1024), 16, (int)(bg + grel[i]), 0, 0, 0);
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

  f32x16 acc[2 * NB + 1];
#pragma unroll
  for (int s = 0; s < 2 * NB + 1; s++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[s][r] = 0.f;

  // bias gradient: the waves with a seventh product (cout block `wave`) of the workgroups that own channel slice 0
  const bool do_csum = p.bias_off >= 0 && cis == 0 && extra;
  float csum = 0.f;

  int buf = 0;
  if (split < p.nchunk) issue(split, 0);
  for (int c = split; c < p.nchunk; c += p.splits) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // chunk c has landed everywhere; every wave is done with the other buffer
    if (c + p.splits < p.nchunk) issue(c + p.splits, buf ^ 1);
    const unsigned bo = (unsigned)(buf * BUF);
    w3l_kstep<NB, WC, 0, RELU>(acc, a0 + bo, a1 + bo, a2 + bo, b0 + bo, bx + bo, extra, csum, do_csum);
    w3l_kstep<NB, WC, 1, RELU>(acc, a0 + bo, a1 + bo, a2 + bo, b0 + bo, bx + bo, extra, csum, do_csum);
    w3l_kstep<NB, WC, 2, RELU>(acc, a0 + bo, a1 + bo, a2 + bo, b0 + bo, bx + bo, extra, csum, do_csum);
    w3l_kstep<NB, WC, 3, RELU>(acc, a0 + bo, a1 + bo, a2 + bo, b0 + bo, bx + bo, extra, csum, do_csum);
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
    const float t = csum + __shfl_xor(csum, 32, 64);                 // the two k-halves of the wave hold different pixels of the same cout
    if (lane < 32) out[p.bias_off + co0 + wave * 32 + lane] = t * p.bias_scale;
  }
}

template <int NB, int WC, bool RELU>
static inline int sg_launch_wgrad_v3l_t(const WgradV3LParams& p, hipStream_t st) {
  constexpr int NIMG = WC == 4 ? 4 : 1, RC = 64 / WC / NIMG, XB = NIMG * (RC + 2) * (WC + 2) * 64;
  constexpr int LDS = 2 * (((XB + 1023) / 1024) * 1024 + 64 * NB * 64);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)sg_wgrad_v3l_kernel<NB, WC, RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1;
    attr_done = true;
  }
  hipLaunchKernelGGL((sg_wgrad_v3l_kernel<NB, WC, RELU>), dim3(p.nci * p.nco * p.splits), dim3(256), LDS, st, p);
  return 0;
}
template <int NB, int WC>
static inline int sg_launch_wgrad_v3l_r(const WgradV3LParams& p, hipStream_t st) {
  return p.x_relu ? sg_launch_wgrad_v3l_t<NB, WC, true>(p, st) : sg_launch_wgrad_v3l_t<NB, WC, false>(p, st);
}
static inline int sg_launch_wgrad_v3l(const WgradV3Params& p0, int NB, hipStream_t st) {
  if (p0.x_up || p0.g_up) return sg_launch_wgrad_v3(p0, NB, st);      // operands read through a 2x nearest upsampling: the shipped kernel
  const WgradV3LParams& p = p0;
  const int wc = p.W >= 64 ? 64 : p.W;
  if (NB == 3) {
    switch (wc) { case 64: return sg_launch_wgrad_v3l_r<3, 64>(p, st); case 32: return sg_launch_wgrad_v3l_r<3, 32>(p, st);
                  case 16: return sg_launch_wgrad_v3l_r<3, 16>(p, st); case 8: return sg_launch_wgrad_v3l_r<3, 8>(p, st);
                  case 4: return sg_launch_wgrad_v3l_r<3, 4>(p, st); }
  } else {
    switch (wc) { case 64: return sg_launch_wgrad_v3l_r<2, 64>(p, st); case 32: return sg_launch_wgrad_v3l_r<2, 32>(p, st);
                  case 16: return sg_launch_wgrad_v3l_r<2, 16>(p, st); case 8: return sg_launch_wgrad_v3l_r<2, 8>(p, st);
                  case 4: return sg_launch_wgrad_v3l_r<2, 4>(p, st); }
  }
  return -1;
}
