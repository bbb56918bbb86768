// conv_v4.hip -- dispatcher of the small-workgroup halo kernel (conv_v4.h): 3x3 / stride 1 / pad 1 forward and data gradient, bf16.
#include "conv_common.h"
#include "conv_v4.h"

// SG_CONV_V4=0 disables it; =force: no minimum tile count (tests: the kernel a batch-256 problem gets, at small batch); =all: in addition
// no channel rule (every eligible shape); default: the short-K layers (C <= 384), which conv_v3.h's one-workgroup-per-CU tiles serve worst.
bool sg_conv_fwd_v4_try(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st) {
  const char* mode = getenv("SG_CONV_V4");
  if (mode && mode[0] == '0') return false;
  const bool all = mode && mode[0] == 'a', force = all || (mode && mode[0] == 'f');
  if (d->stride != 1 || (pflags & SG_PIX_TRANSPOSED) || d->R != 3 || d->S != 3 || d->pad_h != 1 || d->pad_w != 1) return false;
  if (d->C < 32 || d->C % 32 || d->ldx % 8 || !aligned16(d->x) || !aligned16(d->w)) return false;
  if (!all && d->C > 384) return false;
  const bool up = (pflags & SG_PIX_UPSAMPLE) != 0, quad = (pflags & SG_PIX_QUAD) != 0;
  if (d->Ho != d->Hs * (up ? 2 : 1) || d->Wo != d->Ws * (up ? 2 : 1)) return false;
  const int wshift = ilog2_exact(d->Wo), hshift = ilog2_exact(d->Ho);
  if (wshift < 0 || hshift < 0 || d->Ws < 4 || d->Hs < 2) return false;
  const long long xbytes = (((long long)d->N * d->Hs * d->Ws - 1) * d->ldx + d->C) * 2, wbytes = (long long)I * K * 2;
  if (xbytes >= (1ll << 31) || wbytes >= (1ll << 31)) return false;
  if ((e.flags & (SG_EPI_ATOMIC | SG_EPI_OUT_F32)) || (e.ldo & 7) || !aligned16(e.out)) return false;
  if (e.mask && e.res) return false;
  if (e.mask && ((e.ldm & 7) || !aligned16(e.mask))) return false;
  if (e.res && ((e.flags & SG_EPI_RES_F32) || (e.ldr & 7) || !aligned16(e.res))) return false;
  int NB;
  if (I % 96 == 0) NB = 3; else if (I % 64 == 0) NB = 2; else return false;
  // SG_CONV_V4_BJ=512: the 512-pixel tile (12 accumulator blocks per wave, two workgroups per CU) where the shape allows it (A/B switch)
  const char* bj = getenv("SG_CONV_V4_BJ");
  int BJ = (bj && bj[0] == '5') ? 512 : 256;
  if (BJ == 512 && (J % 512 || ((quad || up) && (512 % (2 * d->Wo))))) BJ = 256;
  const int tiles = (I / (32 * NB)) * ((J + BJ - 1) / BJ);
  if (!force && tiles < (BJ == 512 ? 512 : 768)) return false;   // one full wave of workgroups (two / three per CU)
  if ((quad || up) && (BJ % (2 * d->Wo))) return false;         // the tile must cover whole pairs of image rows
  if (J % d->Wo) return false;
  ConvV4Params p;
  p.x = (const bf16_t*)d->x; p.w = (const bf16_t*)d->w;
  p.W = d->Ws; p.wlog = ilog2_exact(d->Ws); p.C = d->C; p.ldx = d->ldx;
  p.Ho = d->Ho; p.Wo = d->Wo; p.wshift = wshift; p.hshift = hshift; p.flags = pflags;
  p.I = I; p.J = J; p.K = K; p.nslice = d->C / 32;
  p.npix_src = d->N * d->Hs * d->Ws;
  p.npx = (((up ? BJ / 4 : BJ) + 2 * d->Ws + 16) + 15) & ~15;
  p.xbytes = (unsigned)xbytes; p.wbytes = (unsigned)wbytes;
  p.wgt_off = p.zero_off = p.bias_off = 0;
  int rc;
  if (BJ == 512) rc = NB == 3 ? sg_launch_conv_v4<3, 4>(p, e, st) : sg_launch_conv_v4<2, 4>(p, e, st);
  else rc = NB == 3 ? sg_launch_conv_v4<3, 2>(p, e, st) : sg_launch_conv_v4<2, 2>(p, e, st);
  return rc == 0;
}
