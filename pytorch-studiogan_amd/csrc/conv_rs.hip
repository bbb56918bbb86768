// conv_rs.hip -- dispatcher of the row-streaming kernel (conv_rs.h): 3x3 / stride 1 / pad 1, <= 32 output channels, 128-pixel-wide images, bf16.
#include "conv_common.h"
#include "conv_rs.h"
#include "conv_rs96.h"

static long long g_rs_launches = 0;

// the 96 -> 96 channel row-streaming kernel (conv_rs96.h). Round 4, first execution (profiles/r04_conv_rs96_first_run.txt): parity green; the
// plain variant 0.986 -> 0.811 ms per launch at batch 256 (858 TFLOP/s: the generator's last 3x3) -- on by default; the pooling variant is
// SLOWER than the halo kernel (1.027 vs 0.926 ms) and the pooled layers run on the quad kernel anyway: only with SG_CONV_RS96=1 / force.
// SG_CONV_RS96=0 switches the kernel off.
static bool conv_fwd_rs96_try(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int K, int pflags, hipStream_t st) {
  const char* m = getenv("SG_CONV_RS96");
  if (m && m[0] == '0') return false;
  const bool force = m && m[0] == 'f';
  const bool pool = (e.flags & SG_EPI_POOL) != 0;
  if (pool && !(m && (m[0] == '1' || m[0] == 'f'))) return false;
  if (d->stride != 1 || d->R != 3 || d->S != 3 || d->pad_h != 1 || d->pad_w != 1) return false;
  if (pflags & (SG_PIX_TRANSPOSED | SG_PIX_UPSAMPLE)) return false;
  if (((pflags & SG_PIX_QUAD) != 0) != pool) return false;
  if (d->C != 96 || I != 96 || d->ldx % 8 || !aligned16(d->x) || !aligned16(d->w)) return false;
  if (d->Ws != 128 || d->Wo != 128 || d->Ho != d->Hs || d->Hs % 8) return false;
  if (e.mask || e.res || (e.flags & ~(SG_EPI_RELU | SG_EPI_POOL)) || (e.ldo & 3) || (((uintptr_t)e.out) & 7)) return false;
  const long long xbytes = (((long long)d->N * d->Hs * d->Ws - 1) * d->ldx + d->C) * 2;
  if (xbytes >= (1ll << 31)) return false;
  int SH = d->Hs;
  while (SH > 8 && (long long)d->N * (d->Hs / SH) < 256 && SH % 2 == 0 && d->Hs % (SH / 2) == 0) SH /= 2;
  if (const char* sh = getenv("SG_CONV_RS_SH")) { const int v = atoi(sh); if (v > 0 && d->Hs % v == 0 && (!pool || v % 2 == 0)) SH = v; }
  const int nstrips = d->N * (d->Hs / SH);
  if (nstrips < 64 && !force) return false;
  ConvRs96Params p;
  p.x = (const bf16_t*)d->x; p.w = (const bf16_t*)d->w;
  p.H = d->Hs; p.ldx = d->ldx; p.K = K; p.SH = SH; p.spi = d->Hs / SH; p.xbytes = (unsigned)xbytes;
  const bool relu = (pflags & SG_PIX_RELU) != 0;
  int rc;
  if (pool) rc = relu ? sg_launch_conv_rs96<true, true>(p, e, nstrips, st) : sg_launch_conv_rs96<false, true>(p, e, nstrips, st);
  else rc = relu ? sg_launch_conv_rs96<true, false>(p, e, nstrips, st) : sg_launch_conv_rs96<false, false>(p, e, nstrips, st);
  return rc == 0;
}

// returns false when the problem is not eligible (the caller falls through to the halo kernel). SG_CONV_RS=0 disables it, =force takes small batches too.
bool sg_conv_fwd_rs_try(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st) {
  const char* m = getenv("SG_CONV_RS");                 // (read per call: the tests switch it)
  if (m && m[0] == '0') return false;
  if (conv_fwd_rs96_try(d, e, I, K, pflags, st)) { __atomic_fetch_add(&g_rs_launches, 1ll, __ATOMIC_RELAXED); return true; }
  const bool force = m && m[0] == 'f';                  // force: skip the "enough strips to fill the chip" rule (tests); SG_CONV_RS_SH=<rows> fixes the strip height
  if (d->stride != 1 || d->R != 3 || d->S != 3 || d->pad_h != 1 || d->pad_w != 1) return false;
  if (pflags & (SG_PIX_TRANSPOSED | SG_PIX_UPSAMPLE | SG_PIX_QUAD)) return false;
  if (d->C != 96 && d->C != 64) return false;
  if (d->ldx % 8 || !aligned16(d->x) || !aligned16(d->w)) return false;
  if (d->Ws != 128 || d->Wo != 128 || d->Ho != d->Hs || d->Hs % 8) return false;
  if (I % 8 || I > 32 || I < 8) return false;
  if (e.mask || e.res || (e.flags & ~SG_EPI_RELU) || (e.ldo & 3) || (((uintptr_t)e.out) & 7)) return false;
  const long long xbytes = (((long long)d->N * d->Hs * d->Ws - 1) * d->ldx + d->C) * 2;
  if (xbytes >= (1ll << 31)) return false;
  // strip height: whole images when the batch alone fills the chip, else the tallest strip that gives >= 256 workgroups (>= 8 rows:
  // a strip pays 2 halo rows and a 55 KB weight fetch)
  int SH = d->Hs;
  while (SH > 8 && (long long)d->N * (d->Hs / SH) < 256 && SH % 2 == 0 && d->Hs % (SH / 2) == 0) SH /= 2;
  if (const char* sh = getenv("SG_CONV_RS_SH")) { const int v = atoi(sh); if (v > 0 && d->Hs % v == 0) SH = v; }
  const int nstrips = d->N * (d->Hs / SH);
  if (nstrips < 64 && !force) return false;
  ConvRsParams p;
  p.x = (const bf16_t*)d->x; p.w = (const bf16_t*)d->w;
  p.H = d->Hs; p.ldx = d->ldx; p.I = I; p.K = K; p.SH = SH; p.spi = d->Hs / SH; p.xbytes = (unsigned)xbytes;
  const bool relu = (pflags & SG_PIX_RELU) != 0;
  int rc;
  if (d->C == 96) rc = relu ? sg_launch_conv_rs<6, true>(p, e, nstrips, st) : sg_launch_conv_rs<6, false>(p, e, nstrips, st);
  else rc = relu ? sg_launch_conv_rs<4, true>(p, e, nstrips, st) : sg_launch_conv_rs<4, false>(p, e, nstrips, st);
  if (rc == 0) __atomic_fetch_add(&g_rs_launches, 1ll, __ATOMIC_RELAXED);
  return rc == 0;
}
extern "C" long long sg_conv_rs_launches(void) { return __atomic_load_n(&g_rs_launches, __ATOMIC_RELAXED); }
