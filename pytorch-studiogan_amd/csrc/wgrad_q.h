// wgrad_q.h -- weight gradient of the "quad" convolutions (conv_q.h): the gradient with respect to the QUAD filter image
//
//   dq[co][view][tap][ci] = sum over the low-resolution grid (n, i, j) of  G_view[n, i, j, co] * X_view[n, i + ti - ea, j + tj - eb, ci]
//
//   POOL form: G = dy (low resolution), X_view = parity view (a, b) of the fine input (ReLU on load optional), (ea, eb) = (a, b)
//   UP form:   G_view = parity view (a, b) of the fine dy, X = the low-resolution input,            (ea, eb) = (1 - a, 1 - b)
//
// 16 C MACs per low-resolution position and output channel instead of the 36 C of the 3x3 weight gradient over the fine grid; the 3x3
// gradient is the transpose of sg_quad_pack's filter sums applied to dq (k_quad_reduce_fold in conv_q.hip, fused with the split-K reduction).
//
// Structure: wgrad_v3.h's (one barrier per chunk of 64 pixels, LDS-DMA staging into two buffers, ds_read_b64_tr_b16 fragments, partial
// tiles to the deterministic two-stage reduction). A workgroup owns (view, S slices of 32 input channels, 32 NB output channels); wave w owns
// TAP w with every (slice, cout block) pair: S NB MFMAs per 16 pixels from S activation fragments and NB gradient fragments (S = 2, NB = 3:
// 10 transpose reads per 6 MFMAs; wgrad_v3.h: 14 per 7). The patch needs ONE halo row and ONE halo column (the taps of a view reach to one
// side only): its origin is the chunk's first pixel shifted by (-ea, -eb). Each slice is its own 64-byte-pitch plane (a 128-byte pitch would
// put pixels p and p + 2 of a transpose read on the same banks).
#pragma once
#include "gemm_core.h"
#include "conv_v2.h"

struct WgradQParams {
  const bf16_t* x; const bf16_t* dy;
  int form;                      // 0 = POOL, 1 = UP
  int ldx, ldg, x_relu;
  int N, H, W, wlog;             // low-resolution grid (W a power of two; W == 4 needs H == 4)
  int C, Cout;                   // full channel counts (dq is [Cout][16][C])
  int nci, nco;                  // input-channel groups of 32 S, cout tiles of 32 NB
  int nchunk;                    // N * H * W / 64
  int splits;                    // workgroups per (view, ci group, co tile); chunk c goes to split c % splits
  unsigned xbytes, gbytes;
  float* out; long long split_stride;   // partial s at out + s * split_stride
  long long bias_off;            // >= 0: sum_pix dy[pix][co] of the workgroup's pixels goes to out[s * split_stride + bias_off + view * Cout + co]
                                 // (ci group 0 only; POOL: view 0 only -- every view reads the same dy)
  float alpha; const float* alpha_ptr;
};

template <int OFF> __device__ __forceinline__ void wq_tr_read(unsigned addr, u32x2& v) {      // asm: see wgrad_v2.h (no compiler vmcnt(0) in front of it)
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF));
}
typedef short wq_s16x2 __attribute__((ext_vector_type(2)));

template <int NB, int WC, int S, int KS>
__device__ __forceinline__ void wq_kstep(f32x16* acc, unsigned a0, unsigned b0, uint32_t relu_bound, float* csum, bool do_csum) {
  constexpr int NIMG = WC == 4 ? 4 : 1, RC = 64 / WC, RCI = RC / NIMG, PW = WC + 1, PPI = (RCI + 1) * PW;
  constexpr int XB = ((NIMG * PPI * 64 + 1023) / 1024) * 1024;                 // one slice plane of the patch
  constexpr int GPITCH = NB * 64;
  constexpr int KX = WC == 4 ? KS * PPI * 64 : (((KS * 16) / WC) * PW + ((KS * 16) % WC)) * 64;
  constexpr int A2 = WC == 4 ? PW * 64 : 256;                                  // the second half of the fragment: + 4 pixels (WC == 4: the next image row)
  constexpr int KG = KS * 16 * GPITCH;
  u32x2 al[S], ah[S], bl[NB], bh[NB];
  wq_tr_read<KX>(a0, al[0]); wq_tr_read<KX + A2>(a0, ah[0]);
  if constexpr (S == 2) { wq_tr_read<KX + XB>(a0, al[1]); wq_tr_read<KX + XB + A2>(a0, ah[1]); }
  wq_tr_read<KG>(b0, bl[0]); wq_tr_read<KG + 4 * GPITCH>(b0, bh[0]);
  wq_tr_read<KG + 64>(b0, bl[1]); wq_tr_read<KG + 64 + 4 * GPITCH>(b0, bh[1]);
  if constexpr (NB == 3) { wq_tr_read<KG + 128>(b0, bl[2]); wq_tr_read<KG + 128 + 4 * GPITCH>(b0, bh[2]); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  bf16x8_t af[S], bf[NB];
#pragma unroll
  for (int s = 0; s < S; s++) {
    asm volatile("" : "+v"(al[s]), "+v"(ah[s]));
    u32x4 v = {al[s][0], al[s][1], ah[s][0], ah[s][1]};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t xq = v[q];
      wq_s16x2 x2 = __builtin_bit_cast(wq_s16x2, xq);
      x2 = __builtin_elementwise_max(x2, __builtin_bit_cast(wq_s16x2, relu_bound));
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
  if (do_csum) {       // bias gradient: this lane's 8 pixels of cout b * 32 + (lane & 31)
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
  for (int s = 0; s < S; s++)
#pragma unroll
    for (int b = 0; b < NB; b++)
      acc[s * NB + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], bf[b], acc[s * NB + b], 0, 0, 0);
}

// NB = 32-wide cout blocks per tile (2 or 3), WC = chunk width in low-resolution pixels (64, 32, 16, 8: 64 / WC whole image rows; 4: four
// whole 4 x 4 images), S = 32-channel input slices per workgroup (1 or 2)
template <int NB, int WC, int S>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void sg_wgrad_q_kernel(WgradQParams p) {
  constexpr int NIMG = WC == 4 ? 4 : 1;             // images per chunk
  constexpr int RC = 64 / WC, RCI = RC / NIMG;      // chunk rows, rows per image part
  constexpr int PW = WC + 1, PR = RCI + 1;          // patch extent (per image part) in pixels
  constexpr int PPI = PR * PW;
  constexpr int NPX = (NIMG * PPI * 64 + 1023) / 1024;   // LDS-DMA pieces of one slice plane
  constexpr int XB = NPX * 1024;
  constexpr int GPITCH = NB * 64;
  constexpr int NPG = 64 * GPITCH / 1024;           // pieces of the dy tile (4 NB)
  constexpr int GOFF = S * XB;
  constexpr int BUF = GOFF + NPG * 1024;            // one staging buffer
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
  const auto rsg = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.gbytes, 0x00020000);
  int bid = blockIdx.x;
  { const int G = gridDim.x; if ((G & 7) == 0) bid = (bid & 7) * (G >> 3) + (bid >> 3); }
  const int tiles = 4 * p.nci * p.nco;
  const int split = bid / tiles;
  int tl = bid - split * tiles;
  const int cis = tl % p.nci; tl /= p.nci;
  const int view = tl & 3, cot = tl >> 2;
  const int ci0 = cis * (32 * S), co0 = cot * (32 * NB);
  const bool pool = p.form == 0;
  const int va = view >> 1, vb = view & 1;
  const int ea = pool ? va : 1 - va, eb = pool ? vb : 1 - vb;
  const int cpr = WC == 4 ? 1 : p.W / WC;           // chunks per image-row group
  const int cpi = WC == 4 ? 1 : (p.H / RC) * cpr;   // chunks per image (WC == 4: a chunk is four images)
  const int W2 = 2 * p.W;

  // element offset of low-resolution pixel (n, hh, ww) in a tensor of pixel pitch ld: plain, or through the parity view (va, vb) of the fine tensor
  auto lowoff = [&](int n, int hh, int ww, int ld) -> unsigned { return ((unsigned)(n * p.H + hh) * (unsigned)p.W + (unsigned)ww) * (unsigned)ld; };
  auto viewoff = [&](int n, int hh, int ww, int ld) -> unsigned { return ((unsigned)((n * p.H + hh) * 2 + va) * (unsigned)W2 + (unsigned)(2 * ww + vb)) * (unsigned)ld; };

  auto issue = [&](int c, int buf) {
    int n, h0, w0;
    if (WC == 4) { n = 4 * c; h0 = 0; w0 = 0; }
    else { n = c / cpi; const int rem = c - n * cpi; const int rg = rem / cpr, cx = rem - rg * cpr; h0 = rg * RC; w0 = cx * WC; }
    char* base = smem + buf * BUF;
#pragma unroll
    for (int s = 0; s < S; s++) {
      for (int j = wave; j < NPX; j += 4) {
        const int o = j * 1024 + lane * 16;
        const int pp = o >> 6, cb = o & 63;
        const int k = pp / PPI, pq = pp - k * PPI;
        const int pr = pq / PW, pc = pq - pr * PW;
        const int hh = h0 + pr - ea, ww = w0 + pc - eb;
        const bool ok = (k < NIMG) & ((unsigned)hh < (unsigned)p.H) & ((unsigned)ww < (unsigned)p.W);
        unsigned off = ((pool ? viewoff(n + k, hh, ww, p.ldx) : lowoff(n + k, hh, ww, p.ldx)) + (unsigned)(ci0 + 32 * s)) * 2u + (unsigned)cb;
        off = ok ? off : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (sg_lptr_t)(base + s * XB + j * 1024), 16, (int)off, 0, 0, 0);
      }
    }
    for (int j = wave; j < NPG; j += 4) {
      const int o = j * 1024 + lane * 16;
      const int px = o / GPITCH, cb = o - px * GPITCH;
      int k = 0, cr, cc;
      if (WC == 4) { k = px >> 4; cr = (px >> 2) & 3; cc = px & 3; } else { cr = px / WC; cc = px - cr * WC; }
      const int hh = h0 + cr, ww = w0 + cc;
      const unsigned off = ((pool ? lowoff(n + k, hh, ww, p.ldg) : viewoff(n + k, hh, ww, p.ldg)) + (unsigned)co0) * 2u + (unsigned)cb;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, (sg_lptr_t)(base + GOFF + j * 1024), 16, (int)off, 0, 0, 0);
    }
  };

  // loop-invariant fragment addresses. One transpose read = 4 pixel rows x 16 channels per 16-lane group; lane result: channel
  // 16 (g16 & 1) + 4 (t & 3) .. + 3 of the block, pixel 8 (g16 >> 1) + (t >> 2) (second read: + 4 pixels).
  const int g16 = lane >> 4, t16 = lane & 15;
  const int prow = 8 * (g16 >> 1) + (t16 >> 2);
  const int csub = 16 * (g16 & 1) + 4 * (t16 & 3);
  const unsigned sb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
  // tap of this wave: (ti, tj) = (wave >> 1, wave & 1) reads patch pixel (row + ti, col + tj) of chunk pixel (row, col)
  const int ppix = (prow / WC) * PW + (prow % WC);
  const unsigned a0 = sb + (((wave >> 1) * PW + (wave & 1)) + ppix) * 64 + csub * 2;
  const unsigned b0 = sb + GOFF + prow * GPITCH + csub * 2;
  const uint32_t relu_bound = p.x_relu ? 0u : 0x80008000u;           // signed 16-bit max with 0 = ReLU of bf16, with -32768 = identity

  f32x16 acc[S * NB];
#pragma unroll
  for (int s = 0; s < S * NB; s++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[s][r] = 0.f;

  const bool do_csum = p.bias_off >= 0 && cis == 0 && wave == 3 && (!pool || view == 0);
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
    wq_kstep<NB, WC, S, 0>(acc, a0 + bo, b0 + bo, relu_bound, csum, do_csum);
    wq_kstep<NB, WC, S, 1>(acc, a0 + bo, b0 + bo, relu_bound, csum, do_csum);
    wq_kstep<NB, WC, S, 2>(acc, a0 + bo, b0 + bo, relu_bound, csum, do_csum);
    wq_kstep<NB, WC, S, 3>(acc, a0 + bo, b0 + bo, relu_bound, csum, do_csum);
    buf ^= 1;
  }

  float al = p.alpha;
  if (p.alpha_ptr) al *= *p.alpha_ptr;
  float* out = p.out + (long long)split * p.split_stride;
  const int vt = view * 4 + wave;
#pragma unroll
  for (int s = 0; s < S; s++)
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const f32x16& a = acc[s * NB + b];
      const int co = co0 + b * 32 + (lane & 31);
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int ci = ci0 + 32 * s + 8 * g4 + 4 * (lane >> 5);
        f32x4 v = {a[4 * g4 + 0] * al, a[4 * g4 + 1] * al, a[4 * g4 + 2] * al, a[4 * g4 + 3] * al};
        *(f32x4*)(out + ((long long)co * 16 + vt) * p.C + ci) = v;
      }
    }
  if (do_csum) {
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const float t = csum[b] + __shfl_xor(csum[b], 32, 64);        // the two k-halves of the wave hold different pixels of the same cout
      if (lane < 32) out[p.bias_off + (long long)view * p.Cout + co0 + b * 32 + lane] = t;
    }
  }
}

template <int NB, int WC, int S>
static inline int sg_launch_wgrad_q_t(const WgradQParams& p, hipStream_t st) {
  constexpr int NIMG = WC == 4 ? 4 : 1, RCI = (64 / WC) / NIMG, PPI = (RCI + 1) * (WC + 1);
  constexpr int XB = ((NIMG * PPI * 64 + 1023) / 1024) * 1024;
  constexpr int LDS = 2 * (S * XB + 64 * NB * 64);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)sg_wgrad_q_kernel<NB, WC, S>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1;
    attr_done = true;
  }
  hipLaunchKernelGGL((sg_wgrad_q_kernel<NB, WC, S>), dim3(4 * p.nci * p.nco * p.splits), dim3(256), LDS, st, p);
  return 0;
}
template <int NB, int S>
static inline int sg_launch_wgrad_q_s(const WgradQParams& p, hipStream_t st) {
  const int wc = p.W >= 64 ? 64 : p.W;
  switch (wc) {
    case 64: return sg_launch_wgrad_q_t<NB, 64, S>(p, st);
    case 32: return sg_launch_wgrad_q_t<NB, 32, S>(p, st);
    case 16: return sg_launch_wgrad_q_t<NB, 16, S>(p, st);
    case 8: return sg_launch_wgrad_q_t<NB, 8, S>(p, st);
    case 4: return sg_launch_wgrad_q_t<NB, 4, S>(p, st);
  }
  return -1;
}
static inline int sg_launch_wgrad_q(const WgradQParams& p, int NB, int S, hipStream_t st) {
  if (NB == 3) return S == 2 ? sg_launch_wgrad_q_s<3, 2>(p, st) : sg_launch_wgrad_q_s<3, 1>(p, st);
  return S == 2 ? sg_launch_wgrad_q_s<2, 2>(p, st) : sg_launch_wgrad_q_s<2, 1>(p, st);
}
