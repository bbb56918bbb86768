// linear_group.hip -- many SMALL fp32 linear layers in one launch.
//
// BigGAN's conditional batch norms (reference src/utils/ops.py:14-28) each own two bias-free linear layers gain(y), bias(y) on a 148-wide conditioning
// vector y = [shared class embedding | z chunk] (src/models/big_resnet.py:139-163): ten [256 x 148] x [148 x 2C] products per generator forward, every one
// known when the forward starts. Each is a few hundred MFLOP -- a 32 us launch of the tile GEMM whose time is its latency (two or twenty-four workgroups on
// 256 CUs): 50 launches, 1.6 ms per C3 step (profiles/r06_bench_biggan128_bs256_kerneltrace_z.txt, sg_gemm_kernel<float, StridedKC...>). Here all items of a
// forward are one grid: block = 64 rows x 64 batch entries of ONE item, found through the items' running tile count.
//
//   out[b][o] = bias[o] + sum_k w[o][k] y[b][k]          (fp32 FMA chain, k ascending -- the order of the exact f32 MFMA path it replaces)
#include "common.h"
#include "../../include/sgamd.h"

#define LG_T 64
#define LG_KC 32
__global__ __launch_bounds__(256) void k_linear_group(const sg_linear_item* items, int n, int B) {
  __shared__ float Ws[LG_T][LG_KC + 1];
  __shared__ float Ys[LG_T][LG_KC + 1];
  int it = 0;
  int t0 = 0;
  for (; it < n; it++) {
    const int nt = (items[it].rows + LG_T - 1) / LG_T;
    if ((int)blockIdx.x < t0 + nt) break;
    t0 += nt;
  }
  if (it >= n) return;
  const sg_linear_item q = items[it];
  const int o0 = ((int)blockIdx.x - t0) * LG_T, b0 = blockIdx.y * LG_T;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < q.K; k0 += LG_KC) {
#pragma unroll
    for (int e = 0; e < (LG_T * LG_KC) / 256; e++) {
      const int idx = threadIdx.x + 256 * e;
      const int r = idx >> 5, kk = idx & 31;
      const bool kin = k0 + kk < q.K;
      Ws[r][kk] = (kin && o0 + r < q.rows) ? q.w[(long long)(o0 + r) * q.K + k0 + kk] : 0.f;
      Ys[r][kk] = (kin && b0 + r < B) ? q.y[(long long)(b0 + r) * q.ldy + k0 + kk] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < LG_KC; kk++) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { a[i] = Ws[tx + 16 * i][kk]; b[i] = Ys[ty + 16 * i][kk]; }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int b = b0 + ty + 16 * j;
    if (b >= B) continue;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int o = o0 + tx + 16 * i;
      if (o < q.rows) q.out[(long long)b * q.ldo + o] = acc[i][j] + (q.bias ? q.bias[o] : 0.f);
    }
  }
}

extern "C" int sg_linear_group(const sg_linear_item* items_dev, const sg_linear_item* items_host, int n, int B, sg_stream_t stream) {
  SG_CHECK(items_dev && items_host && n > 0 && B > 0, "sg_linear_group: bad arguments");
  long long tiles = 0;
  double flop = 0.0;
  for (int i = 0; i < n; i++) {
    const sg_linear_item& q = items_host[i];
    SG_CHECK(q.w && q.y && q.out && q.rows > 0 && q.K > 0 && q.ldy >= q.K && q.ldo >= q.rows, "sg_linear_group: bad item");
    tiles += (q.rows + LG_T - 1) / LG_T;
    flop += 2.0 * (double)q.rows * q.K * B;
  }
  SG_CHECK(tiles < (1ll << 30), "sg_linear_group: too many tiles");
  hipStream_t st = (hipStream_t)stream;
  const int prof = sg_prof_begin(st, flop, 2);
  hipLaunchKernelGGL(k_linear_group, dim3((unsigned)tiles, (unsigned)((B + LG_T - 1) / LG_T)), dim3(256), 0, st, items_dev, n, B);
  sg_prof_end(st, prof);
  SG_LAUNCH_CHECK();
  return 0;
}
