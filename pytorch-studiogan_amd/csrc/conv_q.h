// conv_q.h -- "quad" forward / data-gradient kernel: the 3x3 / pad-1 convolutions that sit next to a 2x resampling, computed through the
// exact identity that folds the resampling into the filter (bf16, C % 32 == 0):
//
//   POOL form   avgpool2(conv3x3(x; w))           = conv4x4, stride 2, pad 1 (x; w'),   w'[u][v] = 1/4 sum_{i,j in {0,1}} w[u - i][v - j]
//               (the discriminator blocks' `conv2d2` + `average_pooling`, reference src/models/big_resnet.py:177-192,221-242; and the data
//               gradient of the UP form)
//   UP form     conv3x3(nearest_up2(x); w)[2i+a, 2j+b] = conv2x2(x; w''_ab) at (i, j)  (four phase filters of 2 x 2 taps, sums of the 3 x 3 taps
//               that land on the same source pixel) -- the generator blocks' `F.interpolate` + `conv2d1` (big_resnet.py:28-42) and the data
//               gradient of the POOL form
//
// Either way a low-resolution position costs 16 C MACs per output channel instead of 36 C: 2.25 x fewer MFMAs for the same result (the
// filter sums are taken in fp32 from the normalised weights and rounded to bf16 once: csrc/conv_q.hip sg_quad_pack).
//
// One kernel serves both forms. Everything is indexed on the LOW-resolution grid [N][Hl][Wl] (raster index q):
//   * the fine tensor [N][2 Hl][2 Wl][C] is four "views" (a, b) = (row parity, column parity): view pixel q lives at fine pixel
//     fp(q) = (q >> wlog) * 4 Wl + a * 2 Wl + 2 (q & (Wl - 1)) + b. POOL reads its input through the four views (one after the other, as
//     four times C / 32 channel slices of the same accumulators); UP writes its output (and reads its ReLU mask / residual) through ONE view
//     per workgroup (blockIdx decides the phase);
//   * per (view, 32-channel slice) the pixel operand is staged once as a raster patch with a one-row halo, exactly like conv_v4.h, and the
//     FOUR taps (ti, tj) read it at low-resolution offsets (ti - ea, tj - eb), (ea, eb) = (a, b) for POOL and (1 - a, 1 - b) for UP;
//   * weights: the quad image [Cout][view][tap][C] (K = 16 C), one 32 NB x 32 tile per (view, slice, tap), two taps ahead through four buffers
//     with counted waits (conv_v4.h's scheme; four buffers because a slice has four taps).
// Tile = 256 low-resolution pixels x 32 NB couts, 4 waves, <= 53 KB of LDS: three workgroups per CU.
#pragma once
#include <type_traits>
#include "conv_v2.h"

struct ConvQParams {
  const bf16_t* x; const bf16_t* w;
  int form;               // 0 = POOL (input through the four views), 1 = UP (output through one view per workgroup)
  int Wl, wlog, hlog;     // low-resolution width (power of two >= 4), log2 Wl, log2 Hl
  int C, ldx;             // input channels, pixel pitch of x (elements)
  int I, J, K;            // couts, N * Hl * Wl, 16 C
  int nslice;             // C / 32
  int npx;                // patch pixels (multiple of 16) >= BJ + 2 Wl + 16
  int bj;                 // pixel tile: 256 or 128
  int gj;                 // tile order: pixel tiles per group (1 = cout tiles / phases of ONE pixel tile are neighbours; > 1 = weight-stationary groups, see the kernel)
  int patchb;             // bytes of one patch buffer (npx * 64)
  int flags;              // SG_PIX_RELU
  unsigned xbytes, wbytes;
  int wgt_off, zero_off, bias_off;
  // fused 1x1 skip (SKIP instantiations, POOL form): acc += sum over the four views of conv1x1(relu?(x2_view); w2), w2 = 1/4 of the skip filter
  const bf16_t* x2; const bf16_t* w2; const float* bias2;
  int C2, ldx2, nslice2;
  int c2x8;               // the skip input has 8 channels (the RGB image): ONE slice whose four 16-byte chunks are the four views; C2 = 32 for the filter
  int skip_norelu;        // ReLU on load applies to the main input only
  unsigned x2bytes, w2bytes;
  float* stats;           // optional [tilesJ * nph][I][2]: per-tile batch-norm statistics of the result (sg_conv_epilogue)
};

// TJW = 32-pixel blocks per wave: 2 (tile of 256 low-resolution pixels) or 1 (128: twice the workgroups for the layers whose whole low-resolution
// grid is a few thousand positions -- the 1536-channel 8 x 8 -> 4 x 4 tail has 256 tiles of 256 x 96 at batch 256, one per CU where three fit)
//
// SKIP (POOL form): the discriminator block's 1x1 skip convolution rides in the same launch (reference src/models/big_resnet.py:221-242:
// `x0 = conv2d0(x0); x0 = average_pooling(x0); out = x + x0`): avgpool2(conv1x1(x2)) = sum over the four parity views of conv1x1(x2_view; w2 / 4),
// i.e. 4 C2 / 32 more K-slices of the same accumulators with ONE tap each (no halo), staged through two slots of the operand area behind the
// main loop. These slices are bound by the staging traffic, not by their 4 NB MFMAs: what they read is the fine skip input, once -- what the
// separate 1x1 launch read from HBM as well, without its output round trip (0.09-0.18 ms per block tail at batch 256,
// profiles/r04_dfwd_timeline_e.txt).
//
// NPMIN > 0: the DOUBLE-BUFFERED variant (two workgroups per CU instead of three). The single-buffered loop stops at every slice boundary to
// reload the patch -- every four taps here, against nine in conv_v4.h. With two patch buffers the next slice's patch is requested at the first
// tap of the current one and has the whole slice to land; the weights then run THREE taps ahead so that the vector-memory counter (which
// retires in order) never has to drain the patch early: per slice the issue order is W3, P', W0', W1', W2' and the wait in front of the
// barrier that ends tap t needs the weights of tap t + 1 only: vmcnt(n_p + 2 n_w) behind taps 0-2 (leaves P' and two weight tiles in flight),
// vmcnt(2 n_w) behind tap 3 (P' and W0' have landed). NPMIN = the smallest number of patch pieces a wave issues (compile-time immediate;
// waves with one piece more only wait a little earlier than they must).
//
// Measured and removed in round 5 (profiles/r05_variant_ab_layer_tables_b.txt, same box, the ten quad layers of C3): weights three taps ahead in the
// single-buffered loop (+1.2 %), the four taps of a slice as two pairs with one barrier per pair (+-0), a one-sided patch halo (-8 % staged bytes: -0.7 %,
// inside the box noise, 0 on the step), 512-pixel tiles at two workgroups per CU (+3.4 % / +7.4 %). Neither the weight tile's latency, nor the barrier
// count, nor the bytes staged per MFMA bound this loop.
template <int NB, bool RELU, int TJW = 2, bool SKIP = false, int NPMIN = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NPMIN > 0 ? 2 : 3, NPMIN > 0 ? 2 : 3))) void sg_conv_q_kernel(ConvQParams p, Epilogue<bf16_t> epi, int tilesI, int tilesJ, int nph) {
  constexpr bool DB = NPMIN > 0;
  constexpr int BI = 32 * NB, BJ = 128 * TJW, NW = 4, TI = NB, TJ = TJW;
  constexpr int PB = BI * 64;                  // one weight tile (BI couts x 32 channels)
  constexpr int NWP = BI / 16;                 // weight DMA pieces per tap (16 rows each): 6 or 4
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = tilesI * tilesJ * nph;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // gj == 1: the phases and cout tiles of one pixel tile are neighbours in launch order: they share the patch through the XCD's L2. That order walks through
  // ALL weight sets (cout tile, phase) once per pixel tile: fine while the whole quad image stays in the 4 MB L2, but the deep layers' images are 19-75 MB and every
  // pixel tile then streams them from the Infinity Cache again (profiles/r06_quad_dispatch_traffic_q.txt: 1.7-3.7 GB of L2 misses per launch for 0.1 GB of tensors).
  // gj > 1: groups of gj pixel tiles; inside a group the pixel tile runs fastest, so the gj workgroups that start together share ONE weight set and walk its
  // channel slices in step (one L2 miss per gj readers), while the group's patches (gj x <= 0.5 MB) stay in L2 for the other weight sets.
  int tI, ph, tJ;
  if (p.gj > 1) {
    const int nI = tilesI * nph, per = nI * p.gj;
    const int grp = bid / per, r = bid - grp * per;
    const int left = tilesJ - grp * p.gj, gs = left < p.gj ? left : p.gj;
    const int iI = r / gs;
    tJ = grp * p.gj + (r - iI * gs);
    tI = iI % tilesI; ph = iI / tilesI;
  } else {
    tI = bid % tilesI;
    const int rest = bid / tilesI;
    ph = rest % nph; tJ = rest / nph;
  }
  const int i0 = tI * BI, j0 = tJ * BJ;
  char* const pbufs = smem + p.wgt_off;
  float* sbias = (float*)(smem + p.bias_off);
  if (epi.bias) {
    for (int i = tid; i < BI; i += 64 * NW) {
      float b = (i0 + i < epi.I) ? epi.bias[i0 + i] : 0.f;
      if (SKIP && p.bias2 && i0 + i < epi.I) b += p.bias2[i0 + i];
      sbias[i] = b;
    }
  }
  if (tid < 32) ((unsigned*)(smem + p.zero_off))[tid] = 0u;

  const auto rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
  const auto rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.wbytes, 0x00020000);
  // DMA piece = 1 KiB = 16 rows x 64 B, LDS linear in lane order: lane -> (row sub = lane >> 2, physical chunk lane & 3); the logical
  // 16-byte chunk it fetches is the swizzle inverse: lc = (lane & 3) ^ (row >> 2 & 3), and row = 16 g + sub gives (sub >> 2) & 3.
  const int sub = lane >> 2;
  const int lc = (lane & 3) ^ ((lane >> 4) & 3);
  const unsigned ldxb = 2u * (unsigned)p.ldx;
  const bool pool = p.form == 0;
  const int wmask = p.Wl - 1;

  // ---- patch DMA: groups of 16 consecutive low-resolution pixels, group g = wave + 4 i ---------------------------------------------
  // Round 5 ("lean" forward loop, after the weight-gradient kernels): what a lane's LDS-DMA pieces read is the same for every slice of a view up to the
  // slice's channel offset, which is wave-uniform -- the per-lane byte offsets (with their bounds decision folded in) are computed once per VIEW and held in
  // registers (pvo), the slice offset rides in the instruction's scalar offset. The loop this replaces re-derived (pixel -> fine pixel -> byte offset, one
  // 64-bit multiply-add each) for every piece of every slice: ~8 vector instructions per piece, one of them quarter-rate, 6-7 pieces per wave and slice of 48 MFMAs.
  const int P0 = j0 - p.Wl - 8;                                      // raster index of patch row 0
  const int ngroups = p.npx >> 4;
  const int pix0 = P0 + 16 * wave + sub;
  constexpr int MAXPG = 7;                                           // pieces per wave held in registers: (BJ + 2 Wl + 16) / 64 rounded up for Wl <= 64
  unsigned pvo[MAXPG];
  int pview = 0;
  auto patch_offset = [&](int view, int i) -> unsigned {
    const int vadd = pool ? ((view >> 1) * 2 * p.Wl + (view & 1)) : 0;
    const int pix = pix0 + 64 * i;
    // POOL: view pixel -> fine pixel; out-of-range low-resolution indices (tile halo beyond the tensor) read zeros
    const unsigned src = pool ? (unsigned)(((pix >> p.wlog) << (p.wlog + 2)) + ((pix & wmask) << 1) + vadd) : (unsigned)pix;
    return ((unsigned)pix < (unsigned)p.J) ? src * ldxb + (unsigned)(lc * 16) : 0x80000000u;
  };
  auto patch_offsets = [&](int view) {
    pview = view;
#pragma unroll
    for (int i = 0; i < MAXPG; i++) pvo[i] = patch_offset(view, i);
  };
  auto patch_slice = [&](int s, int pbase = 0) {
#pragma unroll
    for (int i = 0; i < MAXPG; i++) {
      const int g = wave + NW * i;
      if (g < ngroups) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (sg_lptr_t)(smem + pbase + g * 1024), 16, (int)pvo[i], s * 64, 0, 0);
    }
    for (int g = wave + NW * MAXPG; g < ngroups; g += NW)            // (Wl = 128 only: the eighth and ninth piece, offsets on the fly)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (sg_lptr_t)(smem + pbase + g * 1024), 16, (int)patch_offset(pview, (g - wave) / NW), s * 64, 0, 0);
  };
  // ---- weight DMA: BI rows x 32 channels of (view, slice s, tap t): per-lane row offsets once per workgroup, (view, tap, slice) through the scalar offset ----
  unsigned wvo[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = i0 + 16 * (wave + NW * i) + sub;
    wvo[i] = (row < p.I) ? ((unsigned)row * (unsigned)p.K + (unsigned)(lc * 8)) * 2u : 0x80000000u;
  }
  auto weight_tile = [&](int buf, int view, int s, int t) {
    const int so = ((view * 4 + t) * p.C + s * 32) * 2;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int g = wave + NW * i;
      if (g < NWP) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (sg_lptr_t)(pbufs + buf * PB + g * 1024), 16, (int)wvo[i], so, 0, 0);
    }
  };

  // ---- fragment rows of this lane ----------------------------------------------------------------------------------------------------
  const int wj0 = wave * (32 * TJ);
  const int frow = lane & 31, fhi = lane >> 5;
  int rb[TJ];             // patch row of the centre pixel
  unsigned qval[TJ];      // bit (dr + 1) * 3 + (dc + 1) set = the low-resolution neighbour (dr, dc) is inside the image
#pragma unroll
  for (int b = 0; b < TJ; b++) {
    const int row = j0 + wj0 + b * 32 + frow;
    const int wo = row & wmask, ho = (row >> p.wlog) & ((1 << p.hlog) - 1);
    unsigned m = 0;
    if (row < p.J) {
#pragma unroll
      for (int rr = 0; rr < 3; rr++)
#pragma unroll
        for (int ss = 0; ss < 3; ss++)
          if ((unsigned)(ho - 1 + rr) < (unsigned)(1 << p.hlog) && (unsigned)(wo - 1 + ss) < (unsigned)p.Wl) m |= 1u << (rr * 3 + ss);
    }
    qval[b] = m;
    rb[b] = row - P0;
  }
  // weight fragment addresses: row = cout a * 32 + frow, chunk (ks * 2 + fhi) ^ (row >> 2 & 3); ks = 1 is the address ^ 32
  unsigned wa[TI];
#pragma unroll
  for (int a = 0; a < TI; a++) {
    const int row = a * 32 + frow;
    wa[a] = (unsigned)(row * 64 + ((fhi ^ ((row >> 2) & 3)) << 4));
  }

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; a++)
#pragma unroll
    for (int b = 0; b < TJ; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  // virtual slices: POOL walks (view 0..3) x (C / 32); UP walks the C / 32 slices of its one view (= phase)
  const int nv = pool ? 4 : 1;
  const int nslice = p.nslice;
  const int nvs = nv * nslice;
  const bool two = wave + NW < NWP;                                  // this wave issues two weight pieces per tap
  int view = pool ? 0 : ph, s = 0;
  // patch fragment addresses of the four taps of a view (k-step 1 is the address ^ 32): they depend on the view's tap origin only -- computed once per
  // view instead of once per tap (10 vector instructions per 32-pixel block and tap in the loop this replaces). Invalid taps (image border) read the zero line.
  unsigned qat[4][TJ];
  auto tap_addresses = [&](int view) {
    const int ea = pool ? (view >> 1) : 1 - (view >> 1);
    const int eb = pool ? (view & 1) : 1 - (view & 1);
    const int org = -ea * p.Wl - eb;                                 // patch-row displacement of tap (0, 0)
    const int vbit0 = (1 - ea) * 3 + (1 - eb);                       // validity bit of tap (0, 0)
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int ti = t >> 1, tj = t & 1;
#pragma unroll
      for (int b = 0; b < TJ; b++) {
        const int row = rb[b] + org + ti * p.Wl + tj;
        unsigned a = ((unsigned)row << 6) | ((unsigned)((fhi ^ (row >> 2)) & 3) << 4);
        a = ((qval[b] >> (vbit0 + ti * 3 + tj)) & 1u) ? a : (unsigned)p.zero_off;
        qat[t][b] = a;
      }
    }
  };
  patch_offsets(view);
  tap_addresses(view);
  patch_slice(0);
  weight_tile(0, view, 0, 0);
  weight_tile(1, view, 0, 1);
  if (DB) weight_tile(2, view, 0, 2);
  __syncthreads();
  for (int vs = 0; vs < nvs; vs++) {
    const bool next_slice = vs + 1 < nvs;
    const unsigned pb = DB ? (unsigned)((vs & 1) * p.patchb) : 0u;
    int nview = view, ns = s + 1;
    if (ns == nslice) { ns = 0; nview = view + 1; }
#pragma unroll
    for (int t = 0; t < 4; t++) {
      // weights of the tap after next: buffer (t + 2) % 4 was last read two taps ago, every wave is past two barriers since
      const bool issue = (t + 2 < 4) || next_slice;
      if constexpr (DB) {
        if (t == 0) {
          weight_tile(3, view, s, 3);
          if (next_slice) {
            if (ns == 0) patch_offsets(nview);       // (POOL form: the next slice belongs to the next view)
            patch_slice(ns, ((vs + 1) & 1) * p.patchb);
          }
        } else if (next_slice) {
          weight_tile(t - 1, nview, ns, t - 1);
        }
      } else {
        if (t + 2 < 4) weight_tile(t + 2, view, s, t + 2);
        else if (next_slice) weight_tile(t - 2, nview, ns, t - 2);
      }
      const char* ps = pbufs + t * PB;
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        bf16x8_t pf[TI], qf[TJ];
#pragma unroll
        for (int a = 0; a < TI; a++) {
          u32x4 v = *(const u32x4*)(ps + (wa[a] ^ (unsigned)(ks * 32)));
          pf[a] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int b = 0; b < TJ; b++) {
          unsigned qaddr = qat[t][b] ^ (unsigned)(ks * 32);
          if constexpr (DB) qaddr += (qaddr < (unsigned)p.zero_off) ? pb : 0u;      // (the zero line is not double-buffered)
          u32x4 v = *(const u32x4*)(smem + qaddr);
          if (RELU) v = relu16<bf16_t>(v);
          qf[b] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int a = 0; a < TI; a++)
#pragma unroll
          for (int b = 0; b < TJ; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[a], qf[b], acc[a][b], 0, 0, 0);
      }
      if constexpr (DB) {
        // counted waits (see the kernel comment): n_w = 2 for the waves that issue two weight pieces per tap, else 1
        if (next_slice) {
          if (t < 3) { if (two) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPMIN + 4) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPMIN + 2) : "memory"); }
          else { if (two) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
        } else {
          if (t == 0) { if (two) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
          else if (t == 1) { if (two) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (t == 3 && next_slice && ns == 0) tap_addresses(nview);
      } else if (t == 3 && next_slice) {
        // slice boundary: the patch is single-buffered -- everyone must be done reading it, then it is reloaded (a full stop for this
        // workgroup; the other two workgroups of the CU keep the matrix pipe busy)
        __syncthreads();
        if (ns == 0) { patch_offsets(nview); tap_addresses(nview); }      // (POOL form: the next slice belongs to the next view)
        patch_slice(ns);
        __syncthreads();
      } else {
        if (!issue) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (two) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    view = nview; s = ns;
  }

  if constexpr (SKIP) {
    // the main loop ended behind vmcnt(0) + barrier: the operand area is free. Slot i: patch at i * P2B, weights at 2 * P2B + i * PB.
    const auto rsx2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x2, 0, (int)p.x2bytes, 0x00020000);
    const auto rsw2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (int)p.w2bytes, 0x00020000);
    const unsigned ldx2b = 2u * (unsigned)p.ldx2;
    constexpr int P2B = BJ * 64, NG2 = BJ / 16;
    char* const w2bufs = smem + 2 * P2B;
    const int n2 = p.c2x8 ? 1 : 4 * p.nslice2;
    auto issue2 = [&](int vs2, int slot) {
      // (c2x8: the lane's logical chunk IS the view -- the slice holds view 0 | 1 | 2 | 3 of an 8-channel pixel)
      const int v2 = p.c2x8 ? lc : vs2 / p.nslice2, s2 = p.c2x8 ? 0 : vs2 - (vs2 / p.nslice2) * p.nslice2;
      const int vadd = (v2 >> 1) * 2 * p.Wl + (v2 & 1);
      for (int g = wave; g < NG2; g += NW) {
        const int pix = j0 + 16 * g + sub;
        const unsigned src = (unsigned)(((pix >> p.wlog) << (p.wlog + 2)) + ((pix & wmask) << 1) + vadd);
        unsigned off = src * ldx2b + (p.c2x8 ? 0u : (unsigned)(s2 * 64 + lc * 16));
        off = (pix < p.J) ? off : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx2, (sg_lptr_t)(smem + slot * P2B + g * 1024), 16, (int)off, 0, 0, 0);
      }
      for (int g = wave; g < NWP; g += NW) {
        const int row = i0 + 16 * g + sub;
        unsigned off = ((unsigned)row * (unsigned)p.C2 + (unsigned)(s2 * 32 + lc * 8)) * 2u;
        off = (row < p.I) ? off : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw2, (sg_lptr_t)(w2bufs + slot * PB + g * 1024), 16, (int)off, 0, 0, 0);
      }
    };
    issue2(0, 0);
    int slot = 0;
    for (int vs2 = 0; vs2 < n2; vs2++) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                    // slice vs2 has landed for every wave; every wave is done with slice vs2 - 1
      if (vs2 + 1 < n2) issue2(vs2 + 1, slot ^ 1);
      const char* ps = w2bufs + slot * PB;
      const unsigned pbase = (unsigned)(slot * P2B);
      unsigned qa[TJ];
#pragma unroll
      for (int b = 0; b < TJ; b++) {
        const int row = wj0 + b * 32 + frow;             // tile-local pixel (the skip patch has no halo)
        unsigned a = pbase + (((unsigned)row << 6) | ((unsigned)((fhi ^ (row >> 2)) & 3) << 4));
        a = ((qval[b] >> 4) & 1u) ? a : (unsigned)p.zero_off;        // the centre bit is clear only for rows beyond the problem
        qa[b] = a;
      }
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        bf16x8_t pf[TI], qf[TJ];
#pragma unroll
        for (int a = 0; a < TI; a++) {
          u32x4 v = *(const u32x4*)(ps + (wa[a] ^ (unsigned)(ks * 32)));
          pf[a] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int b = 0; b < TJ; b++) {
          u32x4 v = *(const u32x4*)(smem + (qa[b] ^ (unsigned)(ks * 32)));
          if (RELU && !p.skip_norelu) v = relu16<bf16_t>(v);
          qf[b] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int a = 0; a < TI; a++)
#pragma unroll
          for (int b = 0; b < TJ; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[a], qf[b], acc[a][b], 0, 0, 0);
      }
      slot ^= 1;
    }
  }

  float al = epi.alpha;
  if (epi.alpha_ptr) al *= *epi.alpha_ptr;
  if (pool) {
    sg_conv_epilogue<BI, BJ, NW, TI, TJ>(acc, smem, sbias, epi, i0, j0, 0, wj0, al, true, 0, 0, p.stats, p.I, tJ);
  } else {
    // UP: output / mask / residual rows go through the view of this workgroup's phase
    sg_conv_epilogue<BI, BJ, NW, TI, TJ, true>(acc, smem, sbias, epi, i0, j0, 0, wj0, al, true, p.wlog, (ph >> 1) * 2 * p.Wl + (ph & 1),
                                               p.stats, p.I, tJ * nph + ph);
  }
}

// LDS need (bytes) of a configuration
static inline int sg_conv_q_lds(int NB, int BJ, int npx, bool skip, bool db, int* wgt_off, int* zero_off, int* bias_off) {
  const int BI = 32 * NB;
  const int woff = npx * 64 * (db ? 2 : 1);
  const int ops = woff + 4 * BI * 64;
  const int stage = BJ * (BI * 2 + 16);
  const int skp = skip ? 2 * BJ * 64 + 2 * BI * 64 : 0;      // two staging slots of the fused skip (patch + weights each)
  int body = ops > stage ? ops : stage;
  if (skp > body) body = skp;
  if (wgt_off) *wgt_off = woff;
  if (zero_off) *zero_off = body;
  if (bias_off) *bias_off = body + 128;
  return body + 128 + BI * 4;
}
template <int NB, bool RELU, int TJW, bool SKIP, int NPMIN>
static inline int sg_launch_conv_qr(const ConvQParams& p0, const Epilogue<bf16_t>& e, hipStream_t st) {
  constexpr int BI = 32 * NB, BJ = 128 * TJW;
  ConvQParams p = p0;
  p.patchb = p.npx * 64;
  const int lds = sg_conv_q_lds(NB, BJ, p.npx, SKIP, NPMIN > 0, &p.wgt_off, &p.zero_off, &p.bias_off);
  if (lds > 80 * 1024) return -1;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)sg_conv_q_kernel<NB, RELU, TJW, SKIP, NPMIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return -1;
    attr_done = true;
  }
  const int tilesI = (p.I + BI - 1) / BI, tilesJ = (p.J + BJ - 1) / BJ, nph = p.form == 0 ? 1 : 4;
  hipLaunchKernelGGL((sg_conv_q_kernel<NB, RELU, TJW, SKIP, NPMIN>), dim3(tilesI * tilesJ * nph), dim3(256), lds, st, p, e, tilesI, tilesJ, nph);
  return 0;
}
template <int NB, bool SKIP, bool RELU>
static inline int sg_launch_conv_qd(const ConvQParams& p, const Epilogue<bf16_t>& e, int db, hipStream_t st) {
  if (p.bj == 128) return sg_launch_conv_qr<NB, RELU, 1, SKIP, 0>(p, e, st);
  if (db) {       // double-buffered patch: the smallest per-wave piece count is a compile-time immediate of the counted waits
    const int npmin = (p.npx >> 4) >> 2;
    if (npmin == 4) return sg_launch_conv_qr<NB, RELU, 2, SKIP, 4>(p, e, st);
    if (npmin == 5) return sg_launch_conv_qr<NB, RELU, 2, SKIP, 5>(p, e, st);
    if (npmin == 6) return sg_launch_conv_qr<NB, RELU, 2, SKIP, 6>(p, e, st);
  }
  return sg_launch_conv_qr<NB, RELU, 2, SKIP, 0>(p, e, st);
}
template <int NB>
static inline int sg_launch_conv_q(const ConvQParams& p, const Epilogue<bf16_t>& e, int db, hipStream_t st) {
  const bool relu = (p.flags & SG_PIX_RELU) != 0;
  if (p.x2) return relu ? sg_launch_conv_qd<NB, true, true>(p, e, db, st) : sg_launch_conv_qd<NB, true, false>(p, e, db, st);
  return relu ? sg_launch_conv_qd<NB, false, true>(p, e, db, st) : sg_launch_conv_qd<NB, false, false>(p, e, db, st);
}
