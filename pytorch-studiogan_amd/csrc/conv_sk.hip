// conv_sk.hip -- dispatcher of the small-K streaming kernel (conv_sk.h): 1x1 convolutions with <= 192 channels and the RGB stem, bf16.
#include "conv_common.h"
#include "conv_sk.h"

// small-K streaming kernel (conv_sk.h): 1x1 convolutions with <= 192 input channels and the 3x3 stem over 8 padded channels.
// SG_CONV_SK=0 disables it, =force skips the problem-size heuristic (tests).
bool sg_conv_fwd_sk_try(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st) {
  const char* mode = getenv("SG_CONV_SK");
  if (mode && mode[0] == '0') return false;
  const bool force = mode && mode[0] == 'f';
  if (d->stride != 1 || (pflags & SG_PIX_TRANSPOSED)) return false;
  const bool up = (pflags & SG_PIX_UPSAMPLE) != 0;
  const bool one = d->R == 1 && d->S == 1 && d->pad_h == 0 && d->pad_w == 0 && d->C <= 192;
  const bool stem = d->R == 3 && d->S == 3 && d->pad_h == 1 && d->pad_w == 1 && d->C == 8 && !up;
  if (!one && !stem) return false;
  if (d->C % 8 || d->ldx % 8 || I % 8 || I > 384 || !aligned16(d->x) || !aligned16(d->w)) return false;
  if (d->Ho != d->Hs * (up ? 2 : 1) || d->Wo != d->Ws * (up ? 2 : 1)) return false;
  const int wshift = ilog2_exact(d->Wo), hshift = ilog2_exact(d->Ho);
  if (wshift < 1 || hshift < 1) return false;
  const long long xbytes = (((long long)d->N * d->Hs * d->Ws - 1) * d->ldx + d->C) * 2;
  if (xbytes >= (1ll << 31) || J >= (1 << 30)) return false;
  if ((e.flags & (SG_EPI_ATOMIC | SG_EPI_OUT_F32)) || (e.ldo & 7) || !aligned16(e.out)) return false;
  if (e.mask && e.res) return false;
  if (e.mask && ((e.ldm & 7) || !aligned16(e.mask))) return false;
  if (e.res && ((e.flags & SG_EPI_RES_F32) || (e.ldr & 7) || !aligned16(e.res))) return false;
  if (J < 16384 && !force) return false;
  // output / mask / residual go through buffer descriptors too (32-bit offsets, bit 31 = "no access")
  const bool pooled = (e.flags & SG_EPI_POOL) != 0;
  const long long jout = pooled ? (J >> 2) : J;
  const long long obytes = ((jout - 1) * e.ldo + I) * 2;
  const long long sbytes = e.mask ? ((jout - 1) * e.ldm + I) * 2 : (e.res ? ((jout - 1) * e.ldr + I) * 2 : 16);
  if (obytes >= (1ll << 31) || sbytes >= (1ll << 31)) return false;
  ConvSkParams p;
  p.obytes = (unsigned)obytes; p.sbytes = (unsigned)sbytes;
  p.x = (const bf16_t*)d->x; p.w = (const bf16_t*)d->w;
  p.C = d->C; p.ldx = d->ldx; p.Hs = d->Hs; p.Ws = d->Ws;
  p.Ho = d->Ho; p.Wo = d->Wo; p.wshift = wshift; p.hshift = hshift;
  p.mode3 = stem ? 1 : 0; p.flags = pflags;
  p.I = I; p.J = J; p.K = K; p.xbytes = (unsigned)xbytes; p.nrb = 0;
  return sg_launch_conv_sk(p, e, st) == 0;
}

