// linalg.hip -- the matrix square root of the Frechet distance on the device, in fp64.
// Reference src/metrics/fid.py:34-62 needs tr sqrtm(S1 S2) and gets it from scipy.linalg.sqrtm (complex Schur form of a 2048 x 2048
// product on the host, ~10 s). For symmetric positive definite S1 = L1 L1^T, S2 = L2 L2^T (Cholesky):
//     eig(S1 S2) = eig(L1^T S2 L1) = singular values^2 of B = L2^T L1      =>      tr sqrtm(S1 S2) = sum_i sigma_i(B)   (nuclear norm),
// and the singular values of B come from a one-sided (Hestenes) Jacobi iteration: rows of B are rotated pairwise until they are
// mutually orthogonal; their norms are then the singular values. Everything is dense fp64 FMA work on [n][n] matrices:
//   sg_chol_lower      in-place Cholesky (right-looking, one column + one rank-1 trailing update per step), *flag != 0 if not SPD
//   sg_dgemm_tn        C = A^T B
//   sg_jacobi_sweep    one sweep = n - 1 rounds of n / 2 disjoint row pairs (round-robin tournament), returns the largest
//                      |<b_p, b_q>| / (|b_p| |b_q|) seen (convergence measure) in *offd
//   sg_row_norm_sum    sum_i |row_i|
// The caller (metrics.frechet_inception_distance_device) falls back to the host formula when a covariance is not positive definite
// (fewer samples than dimensions), which is also where the reference takes its eps-regularised branch.
#include "common.h"
#include "../../include/sgamd.h"

__device__ __forceinline__ double block_sum_d(double v, double* sm) {   // blockDim.x == 256
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

// column j: A[j][j] = sqrt(A[j][j]); A[i][j] /= A[j][j] (i > j)
__global__ __launch_bounds__(256) void k_chol_col(double* A, int n, int j, int* flag) {
  __shared__ double piv;
  if (threadIdx.x == 0) {
    double d = A[(long long)j * n + j];
    if (!(d > 0.0)) { atomicExch(flag, j + 1); d = 1.0; }
    piv = sqrt(d);
    A[(long long)j * n + j] = piv;
  }
  __syncthreads();
  const double inv = 1.0 / piv;
  for (int i = j + 1 + blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) A[(long long)i * n + j] *= inv;
}
// trailing update of the lower triangle: A[i][k] -= A[i][j] A[k][j] for j < k <= i < n; 32 x 32 tiles
__global__ __launch_bounds__(256) void k_chol_update(double* A, int n, int j) {
  __shared__ double ci[32], ck[32];
  const int ti = blockIdx.y, tk = blockIdx.x;
  if (tk > ti) return;
  const int i0 = j + 1 + ti * 32, k0 = j + 1 + tk * 32;
  if (threadIdx.x < 32) { const int i = i0 + threadIdx.x; ci[threadIdx.x] = i < n ? A[(long long)i * n + j] : 0.0; }
  else if (threadIdx.x < 64) { const int k = k0 + threadIdx.x - 32; ck[threadIdx.x - 32] = k < n ? A[(long long)k * n + j] : 0.0; }
  __syncthreads();
  const int kk = threadIdx.x & 31;
  for (int ii = threadIdx.x >> 5; ii < 32; ii += 8) {
    const int i = i0 + ii, k = k0 + kk;
    if (i < n && k <= i) A[(long long)i * n + k] -= ci[ii] * ck[kk];
  }
}
__global__ __launch_bounds__(256) void k_zero_upper(double* A, int n) {
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < (long long)n * n; e += (long long)gridDim.x * 256) {
    const int i = (int)(e / n), k = (int)(e % n);
    if (k > i) A[e] = 0.0;
  }
}
extern "C" int sg_chol_lower(double* A, int n, int* flag, sg_stream_t s) {
  SG_CHECK(A && flag && n > 0, "sg_chol_lower: bad args");
  hipStream_t st = (hipStream_t)s;
  if (hipMemsetAsync(flag, 0, sizeof(int), st) != hipSuccess) { sg_set_error("sg_chol_lower: memset"); return -2; }
  for (int j = 0; j < n; j++) {
    const int rem = n - j - 1;
    hipLaunchKernelGGL(k_chol_col, dim3(1), dim3(256), 0, st, A, n, j, flag);      // one block: the pivot is read and rewritten by it alone
    if (rem > 0) {
      const int t = (rem + 31) / 32;
      hipLaunchKernelGGL(k_chol_update, dim3(t, t), dim3(256), 0, st, A, n, j);
    }
  }
  hipLaunchKernelGGL(k_zero_upper, dim3(1024), dim3(256), 0, st, A, n);
  SG_LAUNCH_CHECK();
  return 0;
}

// C[a][b] = sum_k A[k][a] B[k][b]   (all [n][n] row-major); 64 x 64 tile per block, 4 x 4 per thread, k-tiles of 16
__global__ __launch_bounds__(256) void k_dgemm_tn(const double* A, const double* B, double* C, int n) {
  __shared__ double sa[16][64], sb[16][64];
  const int a0 = blockIdx.y * 64, b0 = blockIdx.x * 64;
  const int ta = threadIdx.x >> 4, tb = threadIdx.x & 15;
  double acc[4][4] = {};
  for (int k0 = 0; k0 < n; k0 += 16) {
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
      const int kk = e >> 6, c = e & 63;
      const int k = k0 + kk;
      sa[kk][c] = (k < n && a0 + c < n) ? A[(long long)k * n + a0 + c] : 0.0;
      sb[kk][c] = (k < n && b0 + c < n) ? B[(long long)k * n + b0 + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
      double av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { av[i] = sa[kk][ta * 4 + i]; bv[i] = sb[kk][tb * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] += av[i] * bv[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int a = a0 + ta * 4 + i, b = b0 + tb * 4 + j;
      if (a < n && b < n) C[(long long)a * n + b] = acc[i][j];
    }
}
extern "C" int sg_dgemm_tn(const double* A, const double* B, double* C, int n, sg_stream_t s) {
  SG_CHECK(A && B && C && n > 0, "sg_dgemm_tn: bad args");
  hipLaunchKernelGGL(k_dgemm_tn, dim3((n + 63) / 64, (n + 63) / 64), dim3(256), 0, (hipStream_t)s, A, B, C, n);
  SG_LAUNCH_CHECK();
  return 0;
}

// one round of the tournament: block b rotates the row pair of slot b. n even. Rows p, q of M ([n][n] row-major).
__global__ __launch_bounds__(256) void k_jacobi_round(double* M, int n, int round, unsigned long long* offd_bits) {
  __shared__ double sm[4];
  const int m = n - 1, b = blockIdx.x;
  int p, q;
  if (b == 0) { p = n - 1; q = round % m; }
  else { p = (round + b) % m; q = (round - b + m) % m; }
  double* rp = M + (long long)p * n;
  double* rq = M + (long long)q * n;
  double a = 0.0, bb = 0.0, g = 0.0;
  for (int c = threadIdx.x; c < n; c += 256) { const double x = rp[c], y = rq[c]; a += x * x; bb += y * y; g += x * y; }
  a = block_sum_d(a, sm);
  bb = block_sum_d(bb, sm);
  g = block_sum_d(g, sm);
  const double den = sqrt(a * bb);
  const double off = den > 0.0 ? fabs(g) / den : 0.0;
  if (threadIdx.x == 0) atomicMax(offd_bits, (unsigned long long)__double_as_longlong(off));      // non-negative doubles order like integers
  if (off <= 1e-15 || g == 0.0) return;
  const double zeta = (bb - a) / (2.0 * g);
  const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
  for (int c = threadIdx.x; c < n; c += 256) {
    const double x = rp[c], y = rq[c];
    rp[c] = cs * x - sn * y;
    rq[c] = sn * x + cs * y;
  }
}
extern "C" int sg_jacobi_sweep(double* M, int n, double* offd, sg_stream_t s) {
  SG_CHECK(M && offd && n >= 2 && (n % 2) == 0, "sg_jacobi_sweep: n must be even");
  hipStream_t st = (hipStream_t)s;
  if (hipMemsetAsync(offd, 0, sizeof(double), st) != hipSuccess) { sg_set_error("sg_jacobi_sweep: memset"); return -2; }
  for (int r = 0; r < n - 1; r++) hipLaunchKernelGGL(k_jacobi_round, dim3(n / 2), dim3(256), 0, st, M, n, r, (unsigned long long*)offd);
  SG_LAUNCH_CHECK();
  return 0;
}

// out[0] = sum_i |row_i|_2   (one block; fixed summation order)
__global__ __launch_bounds__(256) void k_row_norm_sum(const double* M, int n, double* out) {
  __shared__ double sm[4];
  double total = 0.0;
  for (int r = 0; r < n; r++) {
    double a = 0.0;
    for (int c = threadIdx.x; c < n; c += 256) { const double x = M[(long long)r * n + c]; a += x * x; }
    a = block_sum_d(a, sm);
    total += sqrt(a);
  }
  if (threadIdx.x == 0) out[0] = total;
}
extern "C" int sg_row_norm_sum(const double* M, int n, double* out, sg_stream_t s) {
  SG_CHECK(M && out && n > 0, "sg_row_norm_sum: bad args");
  hipLaunchKernelGGL(k_row_norm_sum, dim3(1), dim3(256), 0, (hipStream_t)s, M, n, out);
  SG_LAUNCH_CHECK();
  return 0;
}
