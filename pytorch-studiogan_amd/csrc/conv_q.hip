// conv_q.hip -- the "quad" convolutions (conv_q.h, wgrad_q.h): C ABI, filter-image transforms, split-K reduction fused with the back-transform.
//
// A 3x3 / pad-1 convolution followed by 2x2 average pooling (discriminator blocks, reference src/models/big_resnet.py:177-192,221-242) is
// a 4x4 / stride-2 convolution with the filter w'[u][v] = 1/4 sum_{i,j in {0,1}} w[u-i][v-j]; a 3x3 / pad-1 convolution of a nearest-x2
// upsampled tensor (generator blocks, big_resnet.py:28-42) is four 2x2 convolutions of the source tensor, one per output parity, whose
// taps are sums of the 3x3 taps that read the same source pixel. Both cost 16 C MACs per low-resolution position instead of 36 C. The data
// gradient of one form is the other form with a transformed filter, the weight gradients are taken with respect to the 16-tap filter and
// mapped back through the transpose of the filter sums -- everything the 3x3 formulation computes, 2.25 x fewer MFMAs.
#include "conv_common.h"
#include "conv_q.h"
#include "wgrad_q.h"
#include "wgrad_ql.h"

// ---- filter transforms ---------------------------------------------------------------------------------------------------------------
// Per dimension: which 3x3 tap indices r feed quad tap ti of parity a. (POOL: w'[u] = (w[u] + w[u-1]) / 2 at u = 2 ti - a + 1;
// UP: the 3x3 taps d in {-1, 0, 1} of fine row 2i + a that land on source row i + a - 1 + ti.)
__device__ __host__ static inline int quad_pat(int pool_like, int a, int t) {
  // bit r set = tap r contributes
  const int P[2][2] = {{3, 4}, {1, 6}};        // POOL-like
  const int U[2][2] = {{1, 6}, {3, 4}};        // UP-like
  return pool_like ? P[a][t] : U[a][t];
}
// mode 0: POOL forward image from [M][r][s][Cs];  mode 1: UP forward image;
// mode 2: data-gradient image of POOL (run by the UP form) from the FLIPPED transposed 3x3 image [Cin][2-r][2-s][Cout];
// mode 3: data-gradient image of UP (run by the POOL form) from the flipped transposed image.
// With the flipped source, mode 2 has the UP pattern at scale 1/4 and mode 3 the POOL pattern at scale 1 (transposes of each other's forward).
static inline bool quad_mode_pool_like(int mode) { return mode == 0 || mode == 3; }
static inline float quad_mode_scale(int mode) { return (mode == 0 || mode == 2) ? 0.25f : 1.f; }

template <typename T>
__global__ __launch_bounds__(256) void k_quad_pack(const T* src, T* dst, long long nvec, int cv, int pool_like, float scale) {
  constexpr int V = ET<T>::VEC;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const long long m = i / cv;
  const int c = (int)(i - m * cv) * V;
  const int Cs = cv * V;
  const T* s0 = src + m * 9 * Cs + c;
  float w[9][V];
#pragma unroll
  for (int k = 0; k < 9; k++) unpack16<T>(*(const u32x4*)(s0 + (long long)k * Cs), w[k]);
  T* d0 = dst + m * 16 * Cs + c;
#pragma unroll
  for (int view = 0; view < 4; view++)
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int pr = quad_pat(pool_like, view >> 1, t >> 1), pc = quad_pat(pool_like, view & 1, t & 1);
      float o[V];
#pragma unroll
      for (int e = 0; e < V; e++) o[e] = 0.f;
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int s = 0; s < 3; s++)
          if (((pr >> r) & 1) && ((pc >> s) & 1)) {
#pragma unroll
            for (int e = 0; e < V; e++) o[e] += w[r * 3 + s][e];
          }
#pragma unroll
      for (int e = 0; e < V; e++) o[e] *= scale;
      *(u32x4*)(d0 + (long long)(view * 4 + t) * Cs) = pack16<T>(o);
    }
}

extern "C" int sg_quad_pack(int dtype, int mode, const void* src, void* dst, int M, int Cs, sg_stream_t stream) {
  SG_CHECK(src && dst && M > 0 && Cs > 0 && mode >= 0 && mode <= 3, "sg_quad_pack: bad arguments");
  SG_CHECK(aligned16(src) && aligned16(dst), "sg_quad_pack: 16-byte aligned images");
  hipStream_t st = (hipStream_t)stream;
  const int pl = quad_mode_pool_like(mode) ? 1 : 0;
  const float sc = quad_mode_scale(mode);
  if (dtype == SG_DTYPE_BF16) {
    SG_CHECK(Cs % 8 == 0, "sg_quad_pack: bf16 rows are whole 16-byte vectors (Cs % 8 == 0)");
    const long long nvec = (long long)M * (Cs / 8);
    hipLaunchKernelGGL(k_quad_pack<bf16_t>, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, nvec, Cs / 8, pl, sc);
  } else if (dtype == SG_DTYPE_F32) {
    SG_CHECK(Cs % 4 == 0, "sg_quad_pack: fp32 rows are whole 16-byte vectors (Cs % 4 == 0)");
    const long long nvec = (long long)M * (Cs / 4);
    hipLaunchKernelGGL(k_quad_pack<float>, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, st, (const float*)src, (float*)dst, nvec, Cs / 4, pl, sc);
  } else {
    sg_set_error("sg_quad_pack: bad dtype");
    return -1;
  }
  SG_LAUNCH_CHECK();
  return 0;
}

// All quad images of a network in ONE launch (the weight bank packs them right behind sg_sn_forward; one launch per image was 60 us x 135
// launches per C3 step in profiles/r04_bench_biggan128_bs256_kerneltrace_e.txt). Block b finds its item by a scan of the (short) table.
// A thread owns ONE output vector (row m, quad tap vt, 8 / 4 channels): it sums the one, two or four 3x3 taps that feed it and stores 16 bytes.
// (The first version gave a thread all 16 outputs of its (m, channels): 9 loads, 16 stores, 108 VGPRs -- 1.8 TB/s; the 3x3 rows are re-read from
// L2 here, the HBM traffic is the same.)
template <typename T>
__global__ __launch_bounds__(256) void k_quad_pack_batch(const sg_quad_item* items, int n) {
  constexpr int V = ET<T>::VEC;
  long long base = 0;
  int it = 0;
  long long blk = blockIdx.x;
  for (; it < n; it++) {
    const long long per = items[it].mode == 5 ? (long long)items[it].M * 4 : (long long)items[it].M * (items[it].Cs / V) * (items[it].mode == 4 ? 1 : 16);
    const long long nb = (per + 255) / 256;
    if (blk < base + nb) break;
    base += nb;
  }
  if (it >= n) return;
  const sg_quad_item q = items[it];
  const int cv = q.Cs / V;
  const long long i = (blk - base) * 256 + threadIdx.x;
  if (q.mode == 5) {           // the 8-channel (RGB) skip filter x 1/4, once per view: [M][8] -> [M][4 views][8] (conv_q.h c2x8); Cs = 8
    if (i >= (long long)q.M * 4) return;
    float o[V];
    unpack16<T>(*(const u32x4*)((const T*)q.src + (i >> 2) * V), o);
#pragma unroll
    for (int e = 0; e < V; e++) o[e] *= 0.25f;
    *(u32x4*)((T*)q.dst + i * V) = pack16<T>(o);
    return;
  }
  if (q.mode == 4) {           // the 1x1 skip filter of a pooled block tail, x 1/4 (exact in bf16): [M][Cs] -> [M][Cs]
    if (i >= (long long)q.M * cv) return;
    float o[V];
    unpack16<T>(*(const u32x4*)((const T*)q.src + i * V), o);
#pragma unroll
    for (int e = 0; e < V; e++) o[e] *= 0.25f;
    *(u32x4*)((T*)q.dst + i * V) = pack16<T>(o);
    return;
  }
  if (i >= (long long)q.M * cv * 16) return;
  const bool pool_like = q.mode == 0 || q.mode == 3;
  const float scale = (q.mode == 0 || q.mode == 2) ? 0.25f : 1.f;
  const int c = (int)(i % cv) * V;
  const long long mv = i / cv;
  const int vt = (int)(mv & 15);
  const long long m = mv >> 4;
  const int view = vt >> 2, t = vt & 3;
  const int pr = quad_pat(pool_like, view >> 1, t >> 1), pc = quad_pat(pool_like, view & 1, t & 1);
  const T* s0 = (const T*)q.src + m * 9 * q.Cs + c;
  float o[V];
#pragma unroll
  for (int e = 0; e < V; e++) o[e] = 0.f;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int s = 0; s < 3; s++)
      if (((pr >> r) & 1) && ((pc >> s) & 1)) {
        float w[V];
        unpack16<T>(*(const u32x4*)(s0 + (long long)(r * 3 + s) * q.Cs), w);
#pragma unroll
        for (int e = 0; e < V; e++) o[e] += w[e];
      }
#pragma unroll
  for (int e = 0; e < V; e++) o[e] *= scale;
  *(u32x4*)((T*)q.dst + (m * 16 + vt) * q.Cs + c) = pack16<T>(o);
}
extern "C" int sg_quad_pack_batch(int dtype, const sg_quad_item* items_dev, const sg_quad_item* items_host, int n, sg_stream_t stream) {
  SG_CHECK(items_dev && items_host && n > 0 && n <= 64, "sg_quad_pack_batch: bad arguments");
  const int V = dtype == SG_DTYPE_BF16 ? 8 : 4;
  long long blocks = 0;
  for (int i = 0; i < n; i++) {
    const sg_quad_item& q = items_host[i];
    SG_CHECK(q.src && q.dst && q.M > 0 && q.Cs > 0 && q.Cs % V == 0 && q.mode >= 0 && q.mode <= 5 && (q.mode != 5 || (q.Cs == 8 && dtype == SG_DTYPE_BF16)) && aligned16(q.src) && aligned16(q.dst), "sg_quad_pack_batch: bad item");
    blocks += ((q.mode == 5 ? (long long)q.M * 4 : (long long)q.M * (q.Cs / V) * (q.mode == 4 ? 1 : 16)) + 255) / 256;
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SG_DTYPE_BF16) hipLaunchKernelGGL(k_quad_pack_batch<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, items_dev, n);
  else if (dtype == SG_DTYPE_F32) hipLaunchKernelGGL(k_quad_pack_batch<float>, dim3((unsigned)blocks), dim3(256), 0, st, items_dev, n);
  else { sg_set_error("sg_quad_pack_batch: bad dtype"); return -1; }
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- forward / data gradient ---------------------------------------------------------------------------------------------------------
static bool convq_plan(const sg_convq_desc* d, ConvQParams& p, Epilogue<bf16_t>& e, int& NB) {
  if (d->dtype != SG_DTYPE_BF16 || (d->form != SG_Q_POOL && d->form != SG_Q_UP)) return false;
  const char* mode = getenv("SG_CONV_Q");
  if (mode && mode[0] == '0') return false;
  if (d->C < 32 || d->C % 32 || d->ldx % 8 || !aligned16(d->x) || !aligned16(d->wq)) return false;
  const int wlog = ilog2_exact(d->Wl), hlog = ilog2_exact(d->Hl);
  if (wlog < 2 || hlog < 1) return false;
  const long long J = (long long)d->N * d->Hl * d->Wl;
  const long long npix_x = d->form == SG_Q_POOL ? 4 * J : J;
  if (4 * J >= (1ll << 29)) return false;
  const long long xbytes = ((npix_x - 1) * d->ldx + d->C) * 2, wbytes = (long long)d->Cout * 16 * d->C * 2;
  if (xbytes >= (1ll << 31) || wbytes >= (1ll << 31)) return false;
  if ((d->epi_flags & ~SG_EPI_RELU) || (d->pix_flags & ~SG_PIX_RELU)) return false;
  if ((d->ldo & 7) || !aligned16(d->out)) return false;
  if (d->mask && ((d->ldm & 7) || !aligned16(d->mask))) return false;
  if (d->res && ((d->ldr & 7) || !aligned16(d->res))) return false;
  if (d->Cout % 96 == 0) NB = 3; else if (d->Cout % 64 == 0) NB = 2; else return false;
  p.x = (const bf16_t*)d->x; p.w = (const bf16_t*)d->wq;
  p.form = d->form; p.Wl = d->Wl; p.wlog = wlog; p.hlog = hlog;
  p.C = d->C; p.ldx = d->ldx; p.I = d->Cout; p.J = (int)J; p.K = 16 * d->C; p.nslice = d->C / 32;
  {
    // pixel tile: 256 low-resolution positions, or 128 when that leaves fewer than two workgroups per CU (SG_CONV_Q_BJ=128 / 256 forces one: tests, A/B)
    const int nph = d->form == SG_Q_POOL ? 1 : 4;
    const long long tiles256 = (long long)(d->Cout / (32 * NB)) * ((J + 255) / 256) * nph;
    p.bj = tiles256 < 512 ? 128 : 256;
    if (const char* bj = getenv("SG_CONV_Q_BJ")) { if (bj[0] == '1') p.bj = 128; else if (bj[0] == '2') p.bj = 256; }
  }
  p.npx = ((p.bj + 2 * d->Wl + 16) + 15) & ~15;
  // tile order (conv_q.h): weight-stationary groups once the quad image no longer fits the L2 next to the patches (SG_CONV_Q_GJ=n forces n: A/B)
  p.gj = wbytes > (2ll << 20) ? 8 : 1;
  if (const char* gj = getenv("SG_CONV_Q_GJ")) { const int v = atoi(gj); if (v >= 1 && v <= 64) p.gj = v; }
  p.flags = d->pix_flags;
  p.xbytes = (unsigned)xbytes; p.wbytes = (unsigned)wbytes;
  p.wgt_off = p.zero_off = p.bias_off = 0;
  p.x2 = nullptr; p.w2 = nullptr; p.bias2 = nullptr; p.C2 = p.ldx2 = p.nslice2 = 0; p.x2bytes = p.w2bytes = 0; p.c2x8 = p.skip_norelu = 0;
  if (d->x2) {
    const bool c8 = d->C2 == 8;          // the RGB image: one slice = the four views of an 8-channel pixel, filter [Cout][4][8] (sg_quad_pack_batch mode 5)
    if (d->form != SG_Q_POOL || !d->w2q || (!c8 && (d->C2 < 32 || d->C2 % 32)) || d->ldx2 % 8 || !aligned16(d->x2) || !aligned16(d->w2q)) return false;
    if (d->bias2 && !d->bias) return false;
    const int c2w = c8 ? 32 : d->C2;
    const long long x2bytes = ((4 * J - 1) * d->ldx2 + d->C2) * 2, w2bytes = (long long)d->Cout * c2w * 2;
    if (x2bytes >= (1ll << 31) || w2bytes >= (1ll << 31)) return false;
    p.x2 = (const bf16_t*)d->x2; p.w2 = (const bf16_t*)d->w2q; p.bias2 = d->bias2;
    p.C2 = c2w; p.ldx2 = d->ldx2; p.nslice2 = c2w / 32; p.x2bytes = (unsigned)x2bytes; p.w2bytes = (unsigned)w2bytes;
    p.c2x8 = c8 ? 1 : 0; p.skip_norelu = d->x2_norelu ? 1 : 0;
  }
  p.stats = d->stats;
  e.out = d->out; e.out_bstride = 0; e.ldo = d->ldo; e.bias = d->bias;
  e.res = d->res; e.res_bstride = 0; e.ldr = d->ldr; e.beta = d->beta;
  e.mask = (const bf16_t*)d->mask; e.mask_bstride = 0; e.ldm = d->ldm; e.split_stride = 0;
  e.alpha = d->alpha; e.alpha_ptr = d->alpha_ptr; e.flags = d->epi_flags; e.I = d->Cout; e.J = (int)J;
  return true;
}
extern "C" int sg_conv2d_q_ok(const sg_convq_desc* d) {
  if (!d || !d->x || !d->wq || !d->out) return 0;
  ConvQParams p; Epilogue<bf16_t> e; int NB;
  return convq_plan(d, p, e, NB) ? 1 : 0;
}
// rows of the per-tile statistics buffer a launch with d->stats writes ([rows][Cout][2] floats): pixel tiles x phases; 0 = not eligible
extern "C" int sg_conv2d_q_stat_rows(const sg_convq_desc* d) {
  if (!d || !d->x || !d->wq || !d->out) return 0;
  ConvQParams p; Epilogue<bf16_t> e; int NB;
  if (!convq_plan(d, p, e, NB)) return 0;
  const int sbj = p.bj > 256 ? 256 : p.bj;          // (512-pixel tiles write the statistics rows of the 256-pixel tiling)
  return ((p.J + sbj - 1) / sbj) * (d->form == SG_Q_POOL ? 1 : 4);
}
extern "C" int sg_conv2d_q(const sg_convq_desc* d, sg_stream_t stream) {
  SG_CHECK(d && d->x && d->wq && d->out, "sg_conv2d_q: null pointer");
  ConvQParams p; Epilogue<bf16_t> e; int NB = 0;
  SG_CHECK(convq_plan(d, p, e, NB), "sg_conv2d_q: problem not eligible for the quad kernel (ask sg_conv2d_q_ok first)");
  hipStream_t st = (hipStream_t)stream;
  // algorithmic work = the 3x3 convolution over the fine grid this launch stands for; executed = 16 C MACs per low-resolution position
  const double c2 = d->x2 ? (double)d->C2 : 0.0;      // the fused skip stands for a 1x1 convolution over the fine grid
  const int prof = sg_prof_begin_q(st, 2.0 * (double)d->Cout * 4.0 * (double)p.J * (9.0 * (double)d->C + c2), 0);
  sg_prof_set_executed(prof, 2.0 * (double)d->Cout * (double)p.J * (16.0 * (double)d->C + 4.0 * c2));
  // double-buffered-patch variant (conv_q.h NPMIN; two workgroups per CU). Measured (profiles/r04_quad_bench_l_db.txt): slower than the three
  // single-buffered workgroups everywhere (sum of the C3 layers 4.79 -> 5.07 ms forward, 4.97 -> 5.43 ms data gradient) EXCEPT on the 4 x 4
  // low-resolution grids of the UP form (0.349 -> 0.307 and 0.367 -> 0.318 ms), whose patches are 16 images of 16 pixels and whose slices are
  // pure weight streaming: default on there only. SG_CONV_Q_DB=1 / 0 forces it on / off (read per call: the tests switch it).
  const char* ev_db = getenv("SG_CONV_Q_DB");
  int db_mode = ev_db ? (ev_db[0] == '1' ? 1 : 0) : ((d->Wl == 4 && d->form == SG_Q_UP) ? 1 : 0);
  const int rc = NB == 3 ? sg_launch_conv_q<3>(p, e, db_mode, st) : sg_launch_conv_q<2>(p, e, db_mode, st);
  {
    // algorithmic HBM bytes: input (+ skip input), quad filter(s), result (+ mask / residual), bf16
    const double J = (double)p.J, jin = d->form == SG_Q_POOL ? 4.0 * J : J, jout = d->form == SG_Q_POOL ? J : 4.0 * J;
    sg_prof_tag(prof, d->x2 ? SG_ENG_CONV_Q_SKIP : SG_ENG_CONV_Q,
                2.0 * (jin * d->C + (d->x2 ? 4.0 * J * d->C2 + (double)d->Cout * d->C2 : 0.0) + 16.0 * d->C * d->Cout + jout * d->Cout * (1.0 + (d->mask ? 1.0 : 0.0) + (d->res ? 1.0 : 0.0))));
  }
  sg_prof_end(st, prof);
  SG_CHECK(rc == 0, "sg_conv2d_q: launch failed");
  SG_LAUNCH_CHECK();
  return 0;
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------------------
struct QPlan { bool ok; int NB, S, nci, nco, splits, nchunk; long long n, stride; };
static QPlan wgradq_plan(const sg_convq_wgrad_desc* d) {
  QPlan s; s.ok = false; s.NB = s.S = s.nci = s.nco = s.splits = s.nchunk = 0; s.n = s.stride = 0;
  const char* mode = getenv("SG_WGRAD_Q");
  if (mode && mode[0] == '0') return s;
  if (d->dtype != SG_DTYPE_BF16 || (d->form != SG_Q_POOL && d->form != SG_Q_UP)) return s;
  if (ilog2_exact(d->Wl) < 2 || ilog2_exact(d->Hl) < 0) return s;
  if (d->Wl == 4 && (d->Hl != 4 || d->N % 4)) return s;
  if (d->Wl < 64 && d->Wl > 4 && d->Hl % (64 / d->Wl)) return s;      // a chunk = 64 / W whole image rows
  if (d->C % 32 || d->ldx % 8 || d->ldg % 8 || !aligned16(d->x) || !aligned16(d->dy)) return s;
  if (d->x_flags & ~SG_PIX_RELU) return s;
  if (d->Cout % 96 == 0) s.NB = 3; else if (d->Cout % 64 == 0) s.NB = 2; else return s;
  const long long K = (long long)d->N * d->Hl * d->Wl;
  if (K % 64 || 4 * K >= (1ll << 29)) return s;
  const long long xpix = d->form == SG_Q_POOL ? 4 * K : K, gpix = d->form == SG_Q_POOL ? K : 4 * K;
  if (xpix * d->ldx * 2 >= (1ll << 31) || gpix * d->ldg * 2 >= (1ll << 31)) return s;
  s.S = (d->C % 64 == 0) ? 2 : 1;
  s.nci = d->C / (32 * s.S); s.nco = d->Cout / (32 * s.NB); s.nchunk = (int)(K / 64);
  s.n = 16ll * d->C * d->Cout;
  s.stride = s.n + (d->dbias ? 4ll * d->Cout : 0);
  const int tiles = 4 * s.nci * s.nco;
  int sp = d->splits > 0 ? d->splits : (512 >= tiles ? 512 / tiles : 1);   // two workgroups per CU: one full wave of 512
  const int maxs = s.nchunk / 8 > 0 ? s.nchunk / 8 : 1;                   // at least eight chunks per workgroup
  if (sp > maxs) sp = maxs;
  if (sp > 512) sp = 512;
  s.splits = sp;
  s.ok = true;
  return s;
}
extern "C" int sg_conv2d_q_wgrad_plan(const sg_convq_wgrad_desc* d, int* splits, long long* work_floats) {
  SG_CHECK(d && splits && work_floats, "sg_conv2d_q_wgrad_plan: null");
  const QPlan s = wgradq_plan(d);
  *splits = s.ok ? s.splits : 0;
  *work_floats = s.ok ? (long long)s.splits * s.stride : 0;
  return 0;
}

// dw[m][r][s][c] += sum over splits and over the quad taps that contain 3x3 tap (r, s) of scale * partial[split][m][view][tap][c]
// (the transpose of k_quad_pack), fixed summation order. Block = (row m, 64 channels): thread (vt, c4) sums the splits of one quad tap
// for 4 channels, then 9 x 16 threads fold through LDS. Blocks >= nmain reduce the bias-gradient partials.
__global__ __launch_bounds__(256) void k_quad_reduce_fold(const float* partial, float* dw, int splits, long long stride, int M, int C, int pool_like, float scale,
                                                          float* dbias, long long bias_off, int nviews, int nmain) {
  __shared__ f32x4 sm[16][16];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= nmain) {
    const int co = ((int)blockIdx.x - nmain) * 256 + tid;
    if (co < M) {
      float t = 0.f;
      for (int s = 0; s < splits; s++)
        for (int v = 0; v < nviews; v++) t += partial[(long long)s * stride + bias_off + (long long)v * M + co];
      dbias[co] += t;
    }
    return;
  }
  const int cb = C > 64 ? (C + 63) / 64 : 1;
  const int m = blockIdx.x / cb, c0 = (blockIdx.x - m * cb) * 64;
  const int vt = tid >> 4, c4 = tid & 15;
  const int c = c0 + 4 * c4;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (c < C) {
    const float* src = partial + ((long long)m * 16 + vt) * C + c;
    int s = 0;
    for (; s + 3 < splits; s += 4) {
      const f32x4 v0 = *(const f32x4*)(src + (long long)s * stride), v1 = *(const f32x4*)(src + (long long)(s + 1) * stride);
      const f32x4 v2 = *(const f32x4*)(src + (long long)(s + 2) * stride), v3 = *(const f32x4*)(src + (long long)(s + 3) * stride);
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; s < splits; s++) a0 += *(const f32x4*)(src + (long long)s * stride);
  }
  sm[vt][c4] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (tid < 144 && c < C) {
    const int rs = tid >> 4, r = rs / 3, s3 = rs - 3 * r;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int view = 0; view < 4; view++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int pr = quad_pat(pool_like, view >> 1, q >> 1), pc = quad_pat(pool_like, view & 1, q & 1);
        if (((pr >> r) & 1) && ((pc >> s3) & 1)) t += sm[view * 4 + q][c4];
      }
    float* dst = dw + ((long long)m * 9 + rs) * C + c;
    f32x4 o = *(f32x4*)dst;
    *(f32x4*)dst = o + t * scale;
  }
}

extern "C" int sg_conv2d_q_wgrad(const sg_convq_wgrad_desc* d, sg_stream_t stream) {
  SG_CHECK(d && d->x && d->dy && d->dw && d->work, "sg_conv2d_q_wgrad: null pointer");
  const QPlan s = wgradq_plan(d);
  SG_CHECK(s.ok, "sg_conv2d_q_wgrad: problem not eligible (ask sg_conv2d_q_wgrad_plan first)");
  SG_CHECK(d->work_floats >= (long long)s.splits * s.stride, "sg_conv2d_q_wgrad: workspace too small");
  SG_CHECK(d->C % 4 == 0 && aligned16(d->dw) && aligned16(d->work), "sg_conv2d_q_wgrad: 16-byte aligned fp32 rows");
  hipStream_t st = (hipStream_t)stream;
  const long long K = (long long)d->N * d->Hl * d->Wl;
  WgradQParams p;
  p.x = (const bf16_t*)d->x; p.dy = (const bf16_t*)d->dy;
  p.form = d->form; p.ldx = d->ldx; p.ldg = d->ldg; p.x_relu = (d->x_flags & SG_PIX_RELU) ? 1 : 0;
  p.N = d->N; p.H = d->Hl; p.W = d->Wl; p.wlog = ilog2_exact(d->Wl); p.C = d->C; p.Cout = d->Cout;
  p.nci = s.nci; p.nco = s.nco; p.nchunk = s.nchunk; p.splits = s.splits;
  const long long xpix = d->form == SG_Q_POOL ? 4 * K : K, gpix = d->form == SG_Q_POOL ? K : 4 * K;
  p.xbytes = (unsigned)(((xpix - 1) * d->ldx + d->C) * 2);
  p.gbytes = (unsigned)(((gpix - 1) * d->ldg + d->Cout) * 2);
  p.out = d->work; p.split_stride = s.stride;
  p.bias_off = d->dbias ? s.n : -1;
  p.alpha = d->alpha; p.alpha_ptr = d->alpha_ptr;
  const int prof = sg_prof_begin(st, 2.0 * (double)d->Cout * 4.0 * (double)K * 9.0 * (double)d->C, 1);
  sg_prof_set_executed(prof, 2.0 * (double)d->Cout * (double)K * 16.0 * (double)d->C);
  // the lean kernel (wgrad_ql.h; round 5, same box: the ten quad layers of C3 6.29 -> 5.00 ms, profiles/r05_variant_ab_layer_tables_b.txt).
  // SG_WGRAD_Q_LEAN=0 (read per call) selects the round-4 kernel it replaced: the bit-identity reference of tests/test_quad_gpu.py
  const char* lean = getenv("SG_WGRAD_Q_LEAN");
  const int rc = (lean && lean[0] == '0') ? sg_launch_wgrad_q(p, s.NB, s.S, st) : sg_launch_wgrad_ql(p, s.NB, s.S, st);
  if (rc == 0) {
    const bool pl = d->form == SG_Q_POOL;
    const int cb = d->C > 64 ? (d->C + 63) / 64 : 1;
    const int nmain = d->Cout * cb;
    const int nbias = d->dbias ? (d->Cout + 255) / 256 : 0;
    hipLaunchKernelGGL(k_quad_reduce_fold, dim3(nmain + nbias), dim3(256), 0, st, (const float*)d->work, d->dw, s.splits, s.stride, d->Cout, d->C,
                       pl ? 1 : 0, pl ? 0.25f : 1.f, d->dbias, s.n, pl ? 1 : 4, nmain);
  }
  // algorithmic HBM bytes: x and dy once (bf16), the fp32 3x3 gradient read and written
  sg_prof_tag(prof, SG_ENG_WGRAD_Q, 2.0 * ((double)xpix * d->C + (double)gpix * d->Cout) + 8.0 * 9.0 * (double)d->C * d->Cout);
  sg_prof_end(st, prof);
  SG_CHECK(rc == 0, "sg_conv2d_q_wgrad: launch failed");
  SG_LAUNCH_CHECK();
  return 0;
}
