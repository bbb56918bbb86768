// wgrad_v2.h -- second-generation weight-gradient kernel for the hot bf16 shapes.
//
//   dW[co][(tap,ci)] (+)= alpha * sum_pix dy'[pix][co] * x'[pix + tap][ci]
//
// Same contraction/epilogue as gemm_core.h's ConvPixMC x ConvPixMC path, restructured like conv_v2.h:
//   * 512 threads = 8 waves, tile = 256 (tap,ci) rows x 128 couts, BK = 64 pixels, two LDS buffers filled by
//     global_load_lds_dwordx4 (one instruction = 4 pixels x 128 channels = 1 KiB); 4(I) x 2(J) waves, 64x64 per wave
//   * both operands have the reduction index (pixel) as the slow memory index -> LDS image [pixel][channel] read with
//     ds_read_b64_tr_b16. The 4 pixel-rows one transpose-read touches must sit on 4 disjoint 16-bank windows; with a
//     lane-linear DMA image this is done on the SOURCE side: lane (pixel p, slot s) fetches channel chunk s ^ 4*(p&3)
//     and the fragment read applies the same involution (conflict-free, MI355X_MICROARCH.md §LDS)
//   * a lane's (tap, channel) is fixed for the whole k-loop; per k-tile it decomposes TWO pixel indices (shifts) and
//     derives six source byte offsets into two buffer descriptors; halo / tails set bit 31 (out of range: the DMA writes zeros)
#pragma once
#include "gemm_core.h"
#include "conv_v2.h"

struct WgradV2Params {
  const bf16_t* x; const bf16_t* dy;
  int xHs, xWs, C, ldx, x_up, x_relu, Hin, Win;
  int gHs, gWs, Cout, ldg, g_up;
  int Ho, Wo, wshift, hshift;
  int R, S, pad_h, pad_w;
  int I, J, K, klen;
  unsigned xbytes, gbytes;   // descriptor extents (< 2^31)
};

__device__ __forceinline__ bf16x8_t sg_frag_tr_swz(const char* img, int col0, int ks) {
  // img: [64 pixels][128 channels] bf16 (256-byte rows), swizzled slots. Returns the MFMA fragment
  // {pixel = ks*16 + 8*(l>>5) + 0..7} x {channel col0 + (l&31)}.
  const int l = threadIdx.x & 63;
  const int g16 = l >> 4, t = l & 15;
  const int prow = ks * 16 + 8 * (g16 >> 1) + (t >> 2);
  const int col16 = col0 + 16 * (g16 & 1);
  const int slot = ((col16 >> 3) + ((t & 3) >> 1)) ^ (4 * (t >> 2));
  const char* p = img + prow * 256 + slot * 16 + 8 * (t & 1);
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 256));
  s16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return __builtin_bit_cast(bf16x8_t, r);
}

// The same fragment through inline asm. hipcc treats the transpose-read builtin as an LDS access that may alias the LDS-DMA in
// flight and puts `s_waitcnt vmcnt(0)` in front of it -- i.e. it drains the NEXT tile's prefetch right after issuing it, which
// serialised copy and compute in this kernel. The asm form is invisible to that analysis; the data dependence on the read is
// re-created by routing the registers through sg_lgkm_wait<N>() (an `s_waitcnt lgkmcnt(N)` that "modifies" them) before use.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int NF> struct WgFrags { u32x2 v[2 * NF]; };   // NF fragments (2 of P, TJ of Q) x {pixels 0..3 | 4..7 of the lane's 8}
__device__ __forceinline__ void sg_frag_tr_issue(const char* img, int col0, int ks, u32x2& lo, u32x2& hi) {
  const int l = threadIdx.x & 63;
  const int g16 = l >> 4, t = l & 15;
  const int prow = ks * 16 + 8 * (g16 >> 1) + (t >> 2);
  const int col16 = col0 + 16 * (g16 & 1);
  const int slot = ((col16 >> 3) + ((t & 3) >> 1)) ^ (4 * (t >> 2));
  const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(img + prow * 256 + slot * 16 + 8 * (t & 1));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=&v"(lo) : "v"(a));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=&v"(hi) : "v"(a));
}
template <int N> __device__ __forceinline__ void sg_lgkm_wait(WgFrags<4>& f) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(f.v[0]), "+v"(f.v[1]), "+v"(f.v[2]), "+v"(f.v[3]), "+v"(f.v[4]), "+v"(f.v[5]), "+v"(f.v[6]), "+v"(f.v[7])
               : "n"(N));
}
template <int N> __device__ __forceinline__ void sg_lgkm_wait(WgFrags<6>& f) {
  asm volatile("s_waitcnt lgkmcnt(%12)"
               : "+v"(f.v[0]), "+v"(f.v[1]), "+v"(f.v[2]), "+v"(f.v[3]), "+v"(f.v[4]), "+v"(f.v[5]), "+v"(f.v[6]), "+v"(f.v[7]),
                 "+v"(f.v[8]), "+v"(f.v[9]), "+v"(f.v[10]), "+v"(f.v[11])
               : "n"(N));
}
__device__ __forceinline__ bf16x8_t sg_frag_join(u32x2 lo, u32x2 hi) {
  u32x4 r = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8_t, r);
}

// BJ = 128 (4 x 2 waves of 64 x 64) or 256 (64 x 128 per wave: 8 MFMAs per 12 transpose reads instead of 4 per 8, and a third
// fewer DMA pieces per MFMA; for Cout % 256 == 0)
// NBUF = 2: the DMA of k-tile kt+1 overlaps the MFMAs of k-tile kt and is drained at the barrier that ends kt. NBUF = 3 (BJ = 128
// only: 3 x 48 KB of LDS): k-tile kt+2 is requested before the MFMAs of kt and only kt+1 has to have landed at that barrier
// (`s_waitcnt vmcnt(<pieces of one tile>)` + a bare s_barrier -- LDS-DMA stays in flight across it), i.e. two k-tiles of latency
// cover instead of one. Counter evidence for the change: profiles/r02_conv_sq_counters_baseline.txt, SQ_WAIT_ANY = 0.37-0.46 of the
// wave cycles of this kernel at MFMA busy 0.36-0.46.
template <bool XRELU, int BJ, int NBUF = 2>
__global__ __launch_bounds__(512) void sg_wgrad_v2_kernel(WgradV2Params p, Epilogue<bf16_t> epi, int tilesI, int tilesJ) {
  constexpr int IMG = 64 * 256;              // one [64][128] bf16 image
  constexpr int NQI = BJ / 128;              // Q images (128 couts each)
  constexpr int TJ = BJ / 2 / 32;            // 32-wide cout blocks per wave
  constexpr int BUF = (2 + NQI) * IMG;       // P sub-image 0, P sub-image 1, Q image(s)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
  const auto rsg = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.gbytes, 0x00020000);
  const int nt = tilesI * tilesJ;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tI = bid % tilesI, tJ = bid / tilesI;
  const int i0 = tI * 256, j0 = tJ * BJ;
  const int k_begin = blockIdx.y * p.klen;
  const int k_end = (k_begin + p.klen < p.K) ? (k_begin + p.klen) : p.K;
  if (epi.split_stride) epi.out = (float*)epi.out + (long long)blockIdx.y * epi.split_stride;

  // ---- per-lane DMA state (fixed over the k-loop) ---------------------------------------------------------
  const int lrow = lane >> 4;                                    // pixel row inside a 4-pixel DMA group
  const int lc = (lane & 15) ^ (4 * (lrow & 3));                 // logical 8-channel chunk this lane fetches
  int pr[2], ps[2]; unsigned pc[2]; bool pv[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int irow = i0 + 128 * h + 8 * lc;
    pv[h] = irow < p.I;
    const int tap = irow / p.C;
    pc[h] = (unsigned)(irow - tap * p.C);
    pr[h] = tap / p.S - p.pad_h;
    ps[h] = tap % p.S - p.pad_w;
  }
  int qcol[NQI]; bool qv[NQI];
#pragma unroll
  for (int h = 0; h < NQI; h++) { qcol[h] = j0 + 128 * h + 8 * lc; qv[h] = qcol[h] < p.J; }

  auto issue = [&](int buf, int kt) {
    char* base = smem + buf * BUF;
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const int pg = wave + 8 * e;                               // pixel group (4 pixels) inside the 64-pixel tile
      const int pix = k_begin + kt * 64 + 4 * pg + lrow;
      const bool inb = pix < k_end;
      const int pp = inb ? pix : 0;
      const int wo = pp & (p.Wo - 1);
      const int t = pp >> p.wshift;
      const int ho = t & (p.Ho - 1);
      const int n = t >> p.hshift;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        int hh = ho + pr[h], ww = wo + ps[h];
        const bool ok = inb & pv[h] & ((unsigned)hh < (unsigned)p.Hin) & ((unsigned)ww < (unsigned)p.Win);
        if (p.x_up) { hh >>= 1; ww >>= 1; }
        unsigned off = (((unsigned)(n * p.xHs + hh) * (unsigned)p.xWs + (unsigned)ww) * (unsigned)p.ldx + pc[h]) * 2u;
        off = ok ? off : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (sg_lptr_t)(base + h * IMG + pg * 1024), 16, (int)off, 0, 0, 0);
      }
      {
        const int hg = p.g_up ? (ho >> 1) : ho, wg = p.g_up ? (wo >> 1) : wo;
        const unsigned pixb = ((unsigned)(n * p.gHs + hg) * (unsigned)p.gWs + (unsigned)wg) * (unsigned)p.ldg;
#pragma unroll
        for (int h = 0; h < NQI; h++) {
          unsigned off = (pixb + (unsigned)qcol[h]) * 2u;
          off = (inb & qv[h]) ? off : 0x80000000u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, (sg_lptr_t)(base + (2 + h) * IMG + pg * 1024), 16, (int)off, 0, 0, 0);
        }
      }
    }
  };

  f32x16 acc[2][TJ];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < TJ; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  const int wi = wave & 3, wj = wave >> 2;
  const int nk = (k_end - k_begin + 63) / 64;
  constexpr int NPIECE = 2 * (2 + NQI);      // LDS-DMA instructions one issue() makes per wave
  issue(0, 0);
  if (NBUF == 3 && nk > 1) issue(1, 1);
  if (NBUF == 3) {
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  } else {
    __syncthreads();
  }
  int cur = 0, nxt2 = 2;                     // NBUF == 3: buffer of k-tile kt, buffer k-tile kt+2 goes to
  for (int kt = 0; kt < nk; kt++) {
    if (NBUF == 3) { if (kt + 2 < nk) issue(nxt2, kt + 2); }
    else if (kt + 1 < nk) issue((kt + 1) & 1, kt + 1);
    const char* base = smem + (NBUF == 3 ? cur : (kt & 1)) * BUF;
    const char* pimg = base + (wi >> 1) * IMG;
    const char* qbase = base + 2 * IMG;
    // fragments of sub-step ks+1 are requested before the MFMAs of sub-step ks; lgkmcnt(NR) = "everything but those NR reads"
    constexpr int NF = 2 + TJ, NR = 2 * NF;
    WgFrags<NF> fr[2];
    auto load = [&](int ks, WgFrags<NF>& f) {
      sg_frag_tr_issue(pimg, 64 * (wi & 1), ks, f.v[0], f.v[1]);
      sg_frag_tr_issue(pimg, 64 * (wi & 1) + 32, ks, f.v[2], f.v[3]);
#pragma unroll
      for (int b = 0; b < TJ; b++) {
        const int col = (BJ / 2) * wj + 32 * b;                       // cout inside the tile
        sg_frag_tr_issue(qbase + (col >> 7) * IMG, col & 127, ks, f.v[4 + 2 * b], f.v[5 + 2 * b]);
      }
    };
    load(0, fr[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      if (ks < 3) { load(ks + 1, fr[(ks + 1) & 1]); sg_lgkm_wait<NR>(fr[ks & 1]); }
      else sg_lgkm_wait<0>(fr[ks & 1]);
      WgFrags<NF>& f = fr[ks & 1];
      bf16x8_t pf[2], qf[TJ];
#pragma unroll
      for (int a = 0; a < 2; a++) {
        pf[a] = sg_frag_join(f.v[2 * a], f.v[2 * a + 1]);
        if (XRELU) {
          u32x4 v = __builtin_bit_cast(u32x4, pf[a]);
          v = relu16<bf16_t>(v);
          pf[a] = __builtin_bit_cast(bf16x8_t, v);
        }
      }
#pragma unroll
      for (int b = 0; b < TJ; b++) qf[b] = sg_frag_join(f.v[4 + 2 * b], f.v[5 + 2 * b]);
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < TJ; b++)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[a], qf[b], acc[a][b], 0, 0, 0);
    }
    if (NBUF == 3) {
      // k-tile kt+1 must be complete in every wave's view, kt+2 (just requested) may stay in flight; all fragment reads of kt are done
      // (consumed by the MFMAs above through lgkmcnt waits), so its buffer can be refilled after the barrier
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = (cur == 2) ? 0 : cur + 1;
      nxt2 = (nxt2 == 2) ? 0 : nxt2 + 1;
    } else {
      __syncthreads();
    }
  }

  float al = epi.alpha;
  if (epi.alpha_ptr) al *= *epi.alpha_ptr;
#pragma unroll
  for (int ta = 0; ta < 2; ta++)
#pragma unroll
    for (int tb = 0; tb < TJ; tb++) {
      const int j = j0 + (BJ / 2) * wj + tb * 32 + (lane & 31);
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int ii = i0 + 64 * wi + ta * 32 + 8 * g4 + 4 * (lane >> 5);
        float v[4] = {acc[ta][tb][4 * g4 + 0], acc[ta][tb][4 * g4 + 1], acc[ta][tb][4 * g4 + 2], acc[ta][tb][4 * g4 + 3]};
        epi.store(j, ii, v, al);
      }
    }
}

template <bool XRELU, int BJ, int NBUF = 2>
static inline int sg_launch_wgrad_v2r(const WgradV2Params& p, const Epilogue<bf16_t>& e, int splits, hipStream_t st) {
  constexpr int LDS = NBUF * (2 + BJ / 128) * 64 * 256;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)sg_wgrad_v2_kernel<XRELU, BJ, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1;
    attr_done = true;
  }
  const int tilesI = (p.I + 255) / 256, tilesJ = (p.J + BJ - 1) / BJ;
  hipLaunchKernelGGL((sg_wgrad_v2_kernel<XRELU, BJ, NBUF>), dim3(tilesI * tilesJ, splits), dim3(512), LDS, st, p, e, tilesI, tilesJ);
  return 0;
}
// cout tile of the plan (the launcher and wgrad_plan must agree): 256 when the couts fill it AND every workgroup still gets a long
// k-loop. Measured (tools/conv_bench.py, MI355X): 1536 -> 1536 @8^2 +17 %, 768 -> 768 @16^2 +12 %, 1536 -> 768 @16^2 +7 % (85-205
// k-tiles per workgroup), but 768 -> 1536 @8^2 -13 %, 1x1 -19 %, the 4x4 layers -8 % (11-52 k-tiles: the 256 KB partial tile and the
// prologue are not amortised). SG_WGRAD_BJ256=0 / force: never / whenever Cout % 256 == 0.
static inline int sg_wgrad_v2_bj(int I, int J, int K) {
  const char* e = getenv("SG_WGRAD_BJ256");      // (read per call: the tests switch it)
  const int mode = !e ? 1 : (e[0] == '0' ? 0 : (e[0] == 'f' ? 2 : 1));
  if (!mode || J % 256) return 128;
  if (mode == 2) return 256;
  const int tiles = ((I + 255) / 256) * (J / 256);
  int splits = (768 + tiles - 1) / tiles;
  if (splits < 1) splits = 1;
  return (K / splits / 64 >= 80) ? 256 : 128;
}
static inline int sg_launch_wgrad_v2(const WgradV2Params& p, const Epilogue<bf16_t>& e, int splits, hipStream_t st) {
  if (sg_wgrad_v2_bj(p.I, p.J, p.K) == 256)
    return p.x_relu ? sg_launch_wgrad_v2r<true, 256>(p, e, splits, st) : sg_launch_wgrad_v2r<false, 256>(p, e, splits, st);
  // SG_WGRAD_NBUF=2: the two-buffer loop (A/B switch); default three buffers
  static int nbuf = -1;
  if (nbuf < 0) { const char* e3 = getenv("SG_WGRAD_NBUF"); nbuf = (e3 && e3[0] == '2') ? 2 : 3; }
  if (nbuf == 3) return p.x_relu ? sg_launch_wgrad_v2r<true, 128, 3>(p, e, splits, st) : sg_launch_wgrad_v2r<false, 128, 3>(p, e, splits, st);
  return p.x_relu ? sg_launch_wgrad_v2r<true, 128>(p, e, splits, st) : sg_launch_wgrad_v2r<false, 128>(p, e, splits, st);
}
