// conv_v3.hip -- dispatcher of the halo kernel (conv_v3.h): 3x3 / stride 1 / pad 1 forward and data gradient, bf16.
#include "conv_common.h"
#include "conv_v3.h"
extern template int sg_conv_v3_dispatch<2>(int, int, const ConvV3Params&, const Epilogue<bf16_t>&, hipStream_t);   // conv_v3b.hip
extern template int sg_conv_v3_dispatch_nw4<4>(int, int, const ConvV3Params&, const Epilogue<bf16_t>&, hipStream_t);   // conv_v3c.hip

// halo kernel (conv_v3.h) for 3x3 / stride 1 / pad 1 with >= 64 input channels; returns false when the problem is not eligible.
// SG_CONV_V3=0 disables it, =force skips the tile-count heuristic (tests), =all also takes the shapes the default table leaves to v2.
bool sg_conv_fwd_v3_try(const sg_conv_fwd_desc* d, const Epilogue<bf16_t>& e, int I, int J, int K, int pflags, hipStream_t st) {
  const char* mode = getenv("SG_CONV_V3");
  if (mode && mode[0] == '0') return false;
  const bool force = mode && mode[0] == 'f';
  if (d->stride != 1 || (pflags & SG_PIX_TRANSPOSED) || d->R != 3 || d->S != 3 || d->pad_h != 1 || d->pad_w != 1) return false;
  if (d->C < 64 || d->C % 32 || d->ldx % 8 || !aligned16(d->x) || !aligned16(d->w)) return false;   // slices of 64 channels, the last one whole or half
  const bool up = (pflags & SG_PIX_UPSAMPLE) != 0, quad = (pflags & SG_PIX_QUAD) != 0;
  if (d->Ho != d->Hs * (up ? 2 : 1) || d->Wo != d->Ws * (up ? 2 : 1)) return false;
  const int wshift = ilog2_exact(d->Wo), hshift = ilog2_exact(d->Ho);
  if (wshift < 0 || hshift < 0 || d->Ws < 4 || d->Hs < 2) return false;
  const long long xbytes = (((long long)d->N * d->Hs * d->Ws - 1) * d->ldx + d->C) * 2, wbytes = (long long)I * K * 2;
  if (xbytes >= (1ll << 31) || wbytes >= (1ll << 30)) return false;
  if ((e.flags & (SG_EPI_ATOMIC | SG_EPI_OUT_F32)) || (e.ldo & 7) || !aligned16(e.out)) return false;
  // (mask AND residual together: sg_conv_epilogue condenses the mask tile to register bits, then stages the residual tile)
  if (e.mask && ((e.ldm & 7) || !aligned16(e.mask))) return false;
  if (e.res && ((e.flags & SG_EPI_RES_F32) || (e.ldr & 7) || !aligned16(e.res))) return false;
  const int tj = (J + 255) / 256;
  const int cands[3] = {192, 128, 96};
  int best = 0, best_tiles = 0;
  for (int c = 0; c < 3; c++) {
    if (I % cands[c]) continue;
    const int tiles = (I / cands[c]) * tj;
    if (tiles >= 512) { best = cands[c]; best_tiles = tiles; break; }
    if (tiles > best_tiles) { best = cands[c]; best_tiles = tiles; }
  }
  if (I <= 32 && I % 8 == 0 && (J >= 512 * 256 || force)) { best = 32; best_tiles = (J + 511) / 512; }   // narrow outputs (G's RGB layer, 8 padded couts): HBM-bound, one cout tile
  // fewer tiles than CUs: still taken for long reductions (K >= 1152, >= 16 tiles) -- the alternative is the generic engine, whose 128 x 128
  // tiles fill the chip no better and run 3-4x slower per tile (the 1024-channel 8^2 / 4^2 layers of a batch-64 ResNet: 246 us per
  // launch = 78 TFLOP/s in the session-O trace of the WGAN-GP workload)
  if (!best || (best_tiles < 160 && !force && !(K >= 1152 && best_tiles >= 16))) return false;
  // force (tests): take the tile the batch-256 problem gets, so the benchmarked instantiation is the one under test at small batch
  int BJ = (best == 32 || (best == 96 && (J >= 512 * 256 || (force && J % 512 == 0)))) ? 512 : 256;
  if (best == 96) { const char* bj = getenv("SG_V3_BJ96"); if (bj && bj[0] == '2') BJ = 256; }   // A/B: 256-pixel tiles (double patch buffer) for the 96-wide layers
  if ((quad || up) && (BJ % (2 * d->Wo))) return false;     // the tile must cover whole (pairs of) image rows
  if (J % d->Wo) return false;
  ConvV3Params p;
  p.x = (const bf16_t*)d->x; p.w = (const bf16_t*)d->w;
  p.W = d->Ws; p.wlog = ilog2_exact(d->Ws); p.C = d->C; p.ldx = d->ldx;
  p.Ho = d->Ho; p.Wo = d->Wo; p.wshift = wshift; p.hshift = hshift; p.flags = pflags;
  p.I = I; p.J = J; p.K = K; p.nslice = (d->C + 63) / 64;
  p.npix_src = d->N * d->Hs * d->Ws;
  p.npx = (up ? BJ / 4 : BJ) + 2 * d->Ws + 16;
  p.xbytes = (unsigned)xbytes; p.wbytes = (unsigned)wbytes;
  p.zero_off = 0; p.bias_off = 0; p.dump_off = 0;
  {   // image-row parity in the chunk swizzle: quad row order with W >= 16 (conv_v4.h has the derivation); SG_SWZ_PAR=0 switches it off (A/B)
    static int par_mode = -1;
    if (par_mode < 0) { const char* ep = getenv("SG_SWZ_PAR"); par_mode = (ep && ep[0] == '0') ? 0 : 1; }
    const bool on = par_mode && quad && p.wlog >= 4;
    p.pm4 = on ? 4 : 0;
    p.psh = on ? p.wlog - 2 : 0;
  }
  if (((p.npx >> 3) + 7) / 8 >= 19) return false;           // would need more than 2 patch pieces per tap and wave (never with <= 160 KB of LDS)
  {   // SG_V3_NW4=1: four-wave variants of the 192 / 128-wide tiles (A/B switch)
    const char* e4 = getenv("SG_V3_NW4");
    const bool nw4 = e4 && e4[0] == '1';
    if (nw4 && d->C % 64 == 0 && (best == 192 || best == 128) && BJ == 256 && sg_conv_v3_dispatch_nw4<4>(best, BJ, p, e, st) == 0) return true;
  }
  const int rc = (d->C % 64 == 0) ? sg_conv_v3_dispatch<4>(best, BJ, p, e, st) : sg_conv_v3_dispatch<2>(best, BJ, p, e, st);
  return rc == 0;
}
template int sg_conv_v3_dispatch<4>(int, int, const ConvV3Params&, const Epilogue<bf16_t>&, hipStream_t);
