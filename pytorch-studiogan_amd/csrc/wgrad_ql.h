// wgrad_ql.h -- the quad weight-gradient kernel (the "lean" rewrite of wgrad_q.h's loop, the twin of wgrad_v3l.h). Written in round 4 without GPU time from
// the static instruction mix of the round-4 loop (tools/isa_mix.py: 9.5 vector-ALU instructions per MFMA), checked lane by lane on the CPU interpreter against it
// (tests/test_hipemu_cpu.py: same MFMAs in the same order -> dq bit for bit); first GPU run in round 5 (profiles/r05_variant_ab_layer_tables_b.txt, same box):
// the ten quad layers of C3 6.29 -> 5.00 ms (-20 %), bit-identical on the GPU too (tests/test_quad_gpu.py). The default since; SG_WGRAD_Q_LEAN=0 selects
// wgrad_q.h's kernel (the reference of that test).
//
// Same tiling, staging, fragment addresses and result layout as sg_wgrad_q_kernel. Three changes:
//   * the LDS-DMA addresses of a lane's pieces are computed once per workgroup, not once per chunk (see the kernel): the shipped loop spends
//     ~25 vector instructions per piece, 7 of them quarter-rate integer multiplies, 7-8 pieces per wave and chunk of 24 MFMAs;
//   * ReLU-on-load is a template parameter (the UP form's input is the low-resolution generator activation behind a batch norm: no ReLU there;
//     the shipped loop clamps against -32768 when there is nothing to clamp: 4 S v_pk_max_i16 per k-step);
//   * the bias gradient: wave w < NB sums cout block w from the gradient fragment it already holds, one v_dot2_f32_bf16 against (1, 1) per
//     dword (4 per k-step on NB waves) instead of 12 NB unpack + add instructions on wave 3 alone, which every chunk barrier then waits for.
//     Summation order differs from the shipped kernel's (pairs first): fp32 rounding, not bit for bit.
#pragma once
#include "wgrad_q.h"

typedef __bf16 wql_bf2 __attribute__((ext_vector_type(2)));
typedef WgradQParams WgradQLParams;

// fragments of one k-step (16 pixels): S activation fragments (one per slice) and NB gradient fragments, each as two transpose reads
template <int NB, int S> struct WqlFrags { u32x2 al[S], ah[S], bl[NB], bh[NB]; };
template <int NB, int WC, int S, int KS>
__device__ __forceinline__ void wql_issue(unsigned a0, unsigned b0, WqlFrags<NB, S>& f) {      // 2 (S + NB) transpose reads, no wait
  constexpr int NIMG = WC == 4 ? 4 : 1, RC = 64 / WC, RCI = RC / NIMG, PW = WC + 1, PPI = (RCI + 1) * PW;
  constexpr int XB = ((NIMG * PPI * 64 + 1023) / 1024) * 1024;                 // one slice plane of the patch
  constexpr int GPITCH = NB * 64;
  constexpr int KX = WC == 4 ? KS * PPI * 64 : (((KS * 16) / WC) * PW + ((KS * 16) % WC)) * 64;
  constexpr int A2 = WC == 4 ? PW * 64 : 256;                                  // the second half of the fragment: + 4 pixels (WC == 4: the next image row)
  constexpr int KG = KS * 16 * GPITCH;
  wq_tr_read<KX>(a0, f.al[0]); wq_tr_read<KX + A2>(a0, f.ah[0]);
  if constexpr (S == 2) { wq_tr_read<KX + XB>(a0, f.al[1]); wq_tr_read<KX + XB + A2>(a0, f.ah[1]); }
  wq_tr_read<KG>(b0, f.bl[0]); wq_tr_read<KG + 4 * GPITCH>(b0, f.bh[0]);
  wq_tr_read<KG + 64>(b0, f.bl[1]); wq_tr_read<KG + 64 + 4 * GPITCH>(b0, f.bh[1]);
  if constexpr (NB == 3) { wq_tr_read<KG + 128>(b0, f.bl[2]); wq_tr_read<KG + 128 + 4 * GPITCH>(b0, f.bh[2]); }
}
// the S NB MFMAs of a k-step from fragments that HAVE landed (the caller's s_waitcnt lgkmcnt covers them)
template <int NB, int S, bool RELU>
__device__ __forceinline__ void wql_consume(f32x16* acc, WqlFrags<NB, S>& f, int cblk, float& csum) {
  bf16x8_t af[S], bf[NB];
#pragma unroll
  for (int s = 0; s < S; s++) {
    asm volatile("" : "+v"(f.al[s]), "+v"(f.ah[s]));          // (the registers are used behind the wait, not before it)
    u32x4 v = {f.al[s][0], f.al[s][1], f.ah[s][0], f.ah[s][1]};
    if constexpr (RELU) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t xq = v[q];                                                  // (bit_cast straight from a vector element miscompiles: common.h relu16)
        wq_s16x2 x2 = __builtin_bit_cast(wq_s16x2, xq);
        x2 = __builtin_elementwise_max(x2, __builtin_bit_cast(wq_s16x2, 0u));      // signed 16-bit max with 0 = ReLU of a bf16 pair
        v[q] = __builtin_bit_cast(uint32_t, x2);
      }
    }
    af[s] = __builtin_bit_cast(bf16x8_t, v);
  }
#pragma unroll
  for (int b = 0; b < NB; b++) {
    asm volatile("" : "+v"(f.bl[b]), "+v"(f.bh[b]));
    u32x4 v = {f.bl[b][0], f.bl[b][1], f.bh[b][0], f.bh[b][1]};
    bf[b] = __builtin_bit_cast(bf16x8_t, v);
  }
  // bias gradient: cout block cblk (= this wave's index, or -1) from the fragment already in registers: this lane's 8 pixels of cout (lane & 31)
#pragma unroll
  for (int b = 0; b < NB; b++)
    if (b == cblk) {
      const u32x4 v = __builtin_bit_cast(u32x4, bf[b]);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t gq = v[q];
        csum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wql_bf2, gq), __builtin_bit_cast(wql_bf2, 0x3f803f80u), csum, false);
      }
    }
  SG_PRIO_UP();
#pragma unroll
  for (int s = 0; s < S; s++)
#pragma unroll
    for (int b = 0; b < NB; b++)
      acc[s * NB + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], bf[b], acc[s * NB + b], 0, 0, 0);
  SG_PRIO_DOWN();
}
// The four k-steps of a chunk as a two-deep register pipeline -- the reads of k-step s + 1 are in flight while the MFMAs of k-step s issue; the
// counted wait leaves exactly those 2 (S + NB) reads outstanding (LDS operations of a wave return in order). One exposed LDS round trip per chunk
// instead of four; 2 (S + NB) more registers (wgrad_v2.h has run such a pipeline since round 2).
template <int NB, int WC, int S, bool RELU>
__device__ __forceinline__ void wql_chunk_pipelined(f32x16* acc, unsigned a0, unsigned b0, int cblk, float& csum) {
  constexpr int NR = 2 * (S + NB);
  static_assert(NR <= 15, "lgkmcnt is a 4-bit counter");
  WqlFrags<NB, S> f0, f1;
  wql_issue<NB, WC, S, 0>(a0, b0, f0);
  wql_issue<NB, WC, S, 1>(a0, b0, f1);
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR) : "memory");
  wql_consume<NB, S, RELU>(acc, f0, cblk, csum);
  wql_issue<NB, WC, S, 2>(a0, b0, f0);
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR) : "memory");
  wql_consume<NB, S, RELU>(acc, f1, cblk, csum);
  wql_issue<NB, WC, S, 3>(a0, b0, f1);
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR) : "memory");
  wql_consume<NB, S, RELU>(acc, f0, cblk, csum);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wql_consume<NB, S, RELU>(acc, f1, cblk, csum);
}

// NB = 32-wide cout blocks per tile (2 or 3), WC = chunk width in low-resolution pixels (64, 32, 16, 8: 64 / WC whole image rows; 4: four
// whole 4 x 4 images), S = 32-channel input slices per workgroup (1 or 2)
template <int NB, int WC, int S, bool RELU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void sg_wgrad_ql_kernel(WgradQLParams p) {
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

  // ---- LDS-DMA addresses. The shipped kernel derives (image, row, column) of every 16-byte piece from its byte offset again for every chunk:
  // two divisions by constants and three 32-bit multiplies per piece, ~25 vector instructions of which 7 run at quarter rate -- more vector-pipe
  // time per chunk than the chunk's 24 MFMAs take on the matrix pipe. The pieces of a lane are the same for every chunk: their offset RELATIVE to
  // the chunk's first pixel and their (row, column) displacement are computed ONCE here; per chunk a piece costs two adds, two compares and a select.
  // pixel (n + k, h0 + dr, w0 + dc) relative to (n, h0, w0), in pixels of the stored tensor: plain low-resolution, or through the parity view
  const int W2 = 2 * p.W;
  constexpr int NIX = (NPX + 3) / 4;                // x pieces per wave and slice (piece j = wave + 4 i)
  unsigned xrel[NIX]; int xrc[NIX];         // relative byte offset; (row displacement << 16) | (column displacement & 0xffff)
#pragma unroll
  for (int i = 0; i < NIX; i++) {
    const int j = wave + 4 * i;
    const int o = j * 1024 + lane * 16;
    const int pp = o >> 6, cb = o & 63;
    const int k = pp / PPI, pq = pp - k * PPI;
    const int pr = pq / PW, pc = pq - pr * PW;
    const int dr = pr - ea, dc = pc - eb;
    const int pixd = pool ? ((k * p.H + dr) * 2 * W2 + 2 * dc) : ((k * p.H + dr) * p.W + dc);
    xrel[i] = (unsigned)(pixd * p.ldx * 2 + cb);
    const bool inside = (j < NPX) & (k < NIMG);
    xrc[i] = ((inside ? dr : -0x4000) << 16) | (dc & 0xffff);      // (a piece beyond the patch: a row that is never inside [0, H))
  }
  unsigned grel[NB];                                // dy pieces: j = wave + 4 i, i < NB (NPG = 4 NB), always inside the tensor
#pragma unroll
  for (int i = 0; i < NB; i++) {
    const int o = (wave + 4 * i) * 1024 + lane * 16;
    const int px = o / GPITCH, cb = o - px * GPITCH;
    int k = 0, cr, cc;
    if (WC == 4) { k = px >> 4; cr = (px >> 2) & 3; cc = px & 3; } else { cr = px / WC; cc = px - cr * WC; }
    const int pixd = pool ? ((k * p.H + cr) * p.W + cc) : ((k * p.H + cr) * 2 * W2 + 2 * cc);
    grel[i] = (unsigned)(pixd * p.ldg * 2 + cb);
  }
  const unsigned vsel = (unsigned)(va * W2 + vb);   // first pixel of the parity view inside its 2 x 2 cell row pair

  auto issue = [&](int c, int buf) {
    int n, h0, w0;
    if (WC == 4) { n = 4 * c; h0 = 0; w0 = 0; }
    else { n = c / cpi; const int rem = c - n * cpi; const int rg = rem / cpr, cx = rem - rg * cpr; h0 = rg * RC; w0 = cx * WC; }
    char* base = smem + buf * BUF;
    // the chunk's first pixel (wave-uniform: scalar arithmetic)
    const unsigned plow = ((unsigned)(n * p.H + h0) * (unsigned)p.W + (unsigned)w0);
    const unsigned pview = ((unsigned)(n * p.H + h0) * 2u * (unsigned)W2 + 2u * (unsigned)w0) + vsel;
    const unsigned bx = (pool ? pview : plow) * (unsigned)p.ldx * 2u + (unsigned)ci0 * 2u;
    const unsigned bg = (pool ? plow : pview) * (unsigned)p.ldg * 2u + (unsigned)co0 * 2u;
#pragma unroll
    for (int i = 0; i < NIX; i++) {
      const int j = wave + 4 * i;
      if (j < NPX) {
        int rc = xrc[i];
        asm volatile("" : "+v"(rc));                 // (keeps the unpacking inside the loop: hoisted, it would cost the registers the packing saves)
        const bool ok = ((unsigned)(h0 + (rc >> 16)) < (unsigned)p.H) & ((unsigned)(w0 + (int)(short)rc) < (unsigned)p.W);
        const unsigned off = ok ? bx + xrel[i] : 0x80000000u;
#pragma unroll
        for (int s = 0; s < S; s++)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (sg_lptr_t)(base + s * XB + j * 1024), 16, (int)(off + 64u * s), 0, 0, 0);
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
  // tap of this wave: (ti, tj) = (wave >> 1, wave & 1) reads patch pixel (row + ti, col + tj) of chunk pixel (row, col)
  const int ppix = (prow / WC) * PW + (prow % WC);
  const unsigned a0 = sb + (((wave >> 1) * PW + (wave & 1)) + ppix) * 64 + csub * 2;
  const unsigned b0 = sb + GOFF + prow * GPITCH + csub * 2;

  f32x16 acc[S * NB];
#pragma unroll
  for (int s = 0; s < S * NB; s++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[s][r] = 0.f;

  // bias gradient: waves 0 .. NB - 1 of the workgroups that own channel group 0 (POOL: of view 0 -- every view reads the same dy)
  const bool do_csum = p.bias_off >= 0 && cis == 0 && wave < NB && (!pool || view == 0);
  const int cblk = do_csum ? wave : -1;
  float csum = 0.f;

  int buf = 0;
  if (split < p.nchunk) issue(split, 0);
  for (int c = split; c < p.nchunk; c += p.splits) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // chunk c has landed everywhere; every wave is done with the other buffer
    if (c + p.splits < p.nchunk) issue(c + p.splits, buf ^ 1);
    const unsigned bo = (unsigned)(buf * BUF);
    wql_chunk_pipelined<NB, WC, S, RELU>(acc, a0 + bo, b0 + bo, cblk, csum);
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
    const float t = csum + __shfl_xor(csum, 32, 64);                // the two k-halves of the wave hold different pixels of the same cout
    if (lane < 32) out[p.bias_off + (long long)view * p.Cout + co0 + wave * 32 + lane] = t;
  }
}

template <int NB, int WC, int S, bool RELU>
static inline int sg_launch_wgrad_ql_t(const WgradQLParams& p, hipStream_t st) {
  constexpr int NIMG = WC == 4 ? 4 : 1, RCI = (64 / WC) / NIMG, PPI = (RCI + 1) * (WC + 1);
  constexpr int XB = ((NIMG * PPI * 64 + 1023) / 1024) * 1024;
  constexpr int LDS = 2 * (S * XB + 64 * NB * 64);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)sg_wgrad_ql_kernel<NB, WC, S, RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1;
    attr_done = true;
  }
  hipLaunchKernelGGL((sg_wgrad_ql_kernel<NB, WC, S, RELU>), dim3(4 * p.nci * p.nco * p.splits), dim3(256), LDS, st, p);
  return 0;
}
template <int NB, int S, bool RELU>
static inline int sg_launch_wgrad_ql_s(const WgradQLParams& p, hipStream_t st) {
  const int wc = p.W >= 64 ? 64 : p.W;
  switch (wc) {
    case 64: return sg_launch_wgrad_ql_t<NB, 64, S, RELU>(p, st);
    case 32: return sg_launch_wgrad_ql_t<NB, 32, S, RELU>(p, st);
    case 16: return sg_launch_wgrad_ql_t<NB, 16, S, RELU>(p, st);
    case 8: return sg_launch_wgrad_ql_t<NB, 8, S, RELU>(p, st);
    case 4: return sg_launch_wgrad_ql_t<NB, 4, S, RELU>(p, st);
  }
  return -1;
}
template <bool RELU>
static inline int sg_launch_wgrad_ql_r(const WgradQLParams& p, int NB, int S, hipStream_t st) {
  if (NB == 3) return S == 2 ? sg_launch_wgrad_ql_s<3, 2, RELU>(p, st) : sg_launch_wgrad_ql_s<3, 1, RELU>(p, st);
  return S == 2 ? sg_launch_wgrad_ql_s<2, 2, RELU>(p, st) : sg_launch_wgrad_ql_s<2, 1, RELU>(p, st);
}
static inline int sg_launch_wgrad_ql(const WgradQParams& p, int NB, int S, hipStream_t st) {
  return p.x_relu ? sg_launch_wgrad_ql_r<true>(p, NB, S, st) : sg_launch_wgrad_ql_r<false>(p, NB, S, st);
}
