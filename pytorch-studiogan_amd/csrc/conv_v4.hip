// conv_v4.hip -- dispatcher of the small-workgroup halo kernel (conv_v4.h): 3x3 / stride 1 / pad 1 forward and data gradient, bf16.
#include "conv_common.h"
#include "conv_v4.h"

// SG_CONV_V4=0 disables it; =force: no minimum tile count (tests: the kernel a batch-256 problem gets, at small batch); =all: in addition
// no channel rule (every eligible shape); default: the short-K layers (C <= 384), which conv_v3.h's one-workgroup-per-CU tiles serve worst.
// sk != nullptr: the fused 1x1 skip (conv_v4.h SKIP); dry: eligibility only, nothing is launched
bool sg_conv_fwd_v4_skip_try(const sg_conv_fwd_desc* d, const sg_conv_skip_desc* sk, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st, bool dry);
bool sg_conv_fwd_v4_try(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st) {
  return sg_conv_fwd_v4_skip_try(d, nullptr, e, I, J, K, pflags, st, false);
}
bool sg_conv_fwd_v4_skip_try(const sg_conv_fwd_desc* d, const sg_conv_skip_desc* sk, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st, bool dry) {
  const char* mode = getenv("SG_CONV_V4");
  if (mode && mode[0] == '0') return false;
  const bool all = mode && mode[0] == 'a', force = all || (mode && mode[0] == 'f') || sk != nullptr;   // a fused skip has no other engine to fall to
  if (d->stride != 1 || (pflags & SG_PIX_TRANSPOSED) || d->R != 3 || d->S != 3 || d->pad_h != 1 || d->pad_w != 1) return false;
  if (d->C < 32 || d->C % 32 || d->ldx % 8 || !aligned16(d->x) || !aligned16(d->w)) return false;
  if (!all && !sk && d->C > 384) return false;
  const bool up = (pflags & SG_PIX_UPSAMPLE) != 0, quad = (pflags & SG_PIX_QUAD) != 0;
  long long x2bytes = 0, w2bytes = 0;
  if (sk) {
    if (up || !sk->x2 || !sk->w2 || sk->C2 < 32 || sk->C2 % 32 || sk->ldx2 % 8 || !aligned16(sk->x2) || !aligned16(sk->w2)) return false;
    if (sk->bias2 && !d->bias) return false;
    if (sk->x2_up && ((d->Ho & 1) || (d->Wo & 1) || quad)) return false;
    const long long npix2 = (long long)d->N * (sk->x2_up ? (d->Ho / 2) * (d->Wo / 2) : d->Ho * d->Wo);
    x2bytes = ((npix2 - 1) * sk->ldx2 + sk->C2) * 2; w2bytes = (long long)I * sk->C2 * 2;
    if (x2bytes >= (1ll << 31) || w2bytes >= (1ll << 31)) return false;
  }
  if (d->Ho != d->Hs * (up ? 2 : 1) || d->Wo != d->Ws * (up ? 2 : 1)) return false;
  const int wshift = ilog2_exact(d->Wo), hshift = ilog2_exact(d->Ho);
  if (wshift < 0 || hshift < 0 || d->Ws < 4 || d->Hs < 2) return false;
  const long long xbytes = (((long long)d->N * d->Hs * d->Ws - 1) * d->ldx + d->C) * 2, wbytes = (long long)I * K * 2;
  if (xbytes >= (1ll << 31) || wbytes >= (1ll << 31)) return false;
  if ((e.flags & (SG_EPI_ATOMIC | SG_EPI_OUT_F32)) || (e.ldo & 7) || !aligned16(e.out)) return false;
  // (mask AND residual together: sg_conv_epilogue condenses the mask tile to register bits, then stages the residual tile)
  if (e.mask && ((e.ldm & 7) || !aligned16(e.mask))) return false;
  if (e.res && ((e.flags & SG_EPI_RES_F32) || (e.ldr & 7) || !aligned16(e.res))) return false;
  int NB;
  if (I % 96 == 0) NB = 3; else if (I % 64 == 0) NB = 2; else return false;
  // (a 512-pixel tile -- 12 accumulator blocks per wave, two workgroups per CU -- ran no faster than this one in round 2, profiles/r02_conv_layer_table_v4_bj512_q.txt,
  // and again on conv_q.h in round 5: removed)
  const int BJ = 256;
  const int tiles = (I / (32 * NB)) * ((J + BJ - 1) / BJ);
  if (!force && tiles < 768) return false;   // one full wave of workgroups (three per CU)
  if ((quad || up || (sk && sk->x2_up)) && (BJ % (2 * d->Wo))) return false;         // the tile must cover whole pairs of image rows
  if (J % d->Wo) return false;
  {
    // the launcher's own limit, asked here so that a dry run is authoritative (the caller no longer swallows a launch failure): operand area + staged epilogue
    // of one workgroup within 80 KiB
    const int npx = (((up ? BJ / 4 : BJ) + 2 * d->Ws + 16) + 15) & ~15;
    if (sg_conv_v4_lds(NB, npx, nullptr, nullptr, nullptr, sk ? (sk->x2_up ? 3 * 64 * 64 : 2 * 256 * 64) : 0) > 80 * 1024) return false;
  }
  if (dry) return true;
  ConvV4Params p;
  p.x = (const bf16_t*)d->x; p.w = (const bf16_t*)d->w;
  p.W = d->Ws; p.wlog = ilog2_exact(d->Ws); p.C = d->C; p.ldx = d->ldx;
  p.Ho = d->Ho; p.Wo = d->Wo; p.wshift = wshift; p.hshift = hshift; p.flags = pflags;
  p.I = I; p.J = J; p.K = K; p.nslice = d->C / 32;
  p.npix_src = d->N * d->Hs * d->Ws;
  p.npx = (((up ? BJ / 4 : BJ) + 2 * d->Ws + 16) + 15) & ~15;
  p.xbytes = (unsigned)xbytes; p.wbytes = (unsigned)wbytes;
  p.wgt_off = p.zero_off = p.bias_off = 0;
  {   // image-row parity in the chunk swizzle (conv_v4.h): quad row order with W >= 16; SG_SWZ_PAR=0 switches it off (A/B)
    static int par_mode = -1;
    if (par_mode < 0) { const char* ep = getenv("SG_SWZ_PAR"); par_mode = (ep && ep[0] == '0') ? 0 : 1; }
    const bool on = par_mode && quad && p.wlog >= 4;
    p.pm2 = on ? 2 : 0;
    p.psh = on ? p.wlog - 1 : 0;
  }
  p.x2 = nullptr; p.w2 = nullptr; p.bias2 = nullptr; p.C2 = p.ldx2 = p.up2 = p.nslice2 = p.npix2 = 0; p.x2bytes = p.w2bytes = 0;
  p.stats = nullptr;
  if (sk) {
    p.stats = (e.flags & SG_EPI_POOL) ? nullptr : sk->stats;     // (tile rows = output rows; the pooled tail of a D block feeds no batch norm)
    p.x2 = (const bf16_t*)sk->x2; p.w2 = (const bf16_t*)sk->w2; p.bias2 = sk->bias2;
    p.C2 = sk->C2; p.ldx2 = sk->ldx2; p.up2 = sk->x2_up ? 1 : 0; p.nslice2 = sk->C2 / 32;
    p.npix2 = d->N * (sk->x2_up ? (d->Ho / 2) * (d->Wo / 2) : d->Ho * d->Wo);
    p.x2bytes = (unsigned)x2bytes; p.w2bytes = (unsigned)w2bytes;
    return (NB == 3 ? sg_launch_conv_v4_skip<3>(p, e, st) : sg_launch_conv_v4_skip<2>(p, e, st)) == 0;
  }
  return (NB == 3 ? sg_launch_conv_v4<3, 2>(p, e, st) : sg_launch_conv_v4<2, 2>(p, e, st)) == 0;
}
