// Multi-column responses and fantasies (the X_pend branch of the score classes).
//
// Reference behaviour covered:
//   scores/expected_improvement.lua:51-60   Y_pend = model:fantasize(nFantasies, X_obs, Y_obs, X_pend, hyp);
//                                           Y_obs = [Y_obs x nFantasies ; Y_pend]  -> predict with c = nFantasies columns
//   scores/expected_improvement.lua:83-85   ei:mean(2) over the fantasy columns (the score kernels take c columns)
// gp_regressor:fantasize is in the absent `gp` package: restated as a joint draw from the GP posterior at the
// pending points, y = mu_P + chol(Sigma_P) z, Sigma_P = K(Xp,Xp) - K(Xp,X) K^-1 K(X,Xp) (+ noise with
// b7_gp_opts.var_with_noise).  torch.randn's stream is not part of the reference: z comes from a documented
// counter-based generator (splitmix64 -> Box-Muller).
#include "b7_internal.h"
#include "gemm_f64.h"

namespace {

// mu[j][c] = mean + sum_k K*[j][k] alpha[k][c]   (alpha row-major Npad x yld, yld a multiple of 64)
using GM = GemmF64<128, 64, 16, 2, 2, true, 1>;
__global__ void __launch_bounds__(256)
    mean_multi_kernel(const double *__restrict__ ks, const double *__restrict__ alpha, int Npad, int yld, int ycols,
                      int64_t row0, int64_t Mtotal, double meanc, double *__restrict__ mu) {
  __shared__ __align__(16) double sm[2 * GM::STAGE_DOUBLES];
  d4_t acc[GM::TM][GM::TN] = {};
  GM::run(ks + (int64_t)blockIdx.x * 128 * Npad, Npad, alpha + blockIdx.y * 64, yld, 0, Npad, acc, sm);
#pragma unroll
  for (int i = 0; i < GM::TM; ++i)
#pragma unroll
    for (int j = 0; j < GM::TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t g = row0 + (int64_t)blockIdx.x * 128 + GM::out_row(i, r);
        const int cc = blockIdx.y * 64 + GM::out_col(j);
        if (g < Mtotal && cc < ycols) mu[g * ycols + cc] = meanc + acc[i][j][r];
      }
}

// C[m][n] = sum_k A[m][k] B[n][k]; all extents multiples of 64 (k of 16)
using GN = GemmF64<64, 64, 16, 2, 2, false, 1>;
__global__ void __launch_bounds__(256)
    gemm_nt_kernel(const double *__restrict__ A, int lda, const double *__restrict__ B, int ldb, double *__restrict__ C,
                   int ldc, int k) {
  __shared__ __align__(16) double sm[2 * GN::STAGE_DOUBLES];
  d4_t acc[GN::TM][GN::TN] = {};
  GN::run(A + (int64_t)blockIdx.y * 64 * lda, lda, B + (int64_t)blockIdx.x * 64 * ldb, ldb, 0, k, acc, sm);
  double *out = C + (int64_t)blockIdx.y * 64 * ldc + blockIdx.x * 64;
#pragma unroll
  for (int i = 0; i < GN::TM; ++i)
#pragma unroll
    for (int j = 0; j < GN::TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(int64_t)GN::out_row(i, r) * ldc + GN::out_col(j)] = acc[i][j][r];
}

// S = Kpp - G (+ diag_add) on the P x P corner, identity in the padding of the 64 x 64 block
__global__ void fantasy_cov_kernel(const double *__restrict__ kpp, const double *__restrict__ g, double *__restrict__ S,
                                   int P, double diag_add) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 64 * 64) return;
  const int i = e >> 6, j = e & 63;
  double v;
  if (i < P && j < P) {
    v = kpp[e] - g[e];
    if (i == j) v = v + diag_add;
  } else {
    v = (i == j) ? 1.0 : 0.0;
  }
  S[e] = v;
}

__global__ void add_diag_kernel(double *__restrict__ S, int ld, int n, double v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) S[(int64_t)i * ld + i] += v;
}

__device__ inline uint64_t splitmix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// z(k, s) ~ N(0,1): Box-Muller on two counter-based uniforms; u1 in (0, 1]
__device__ inline double counter_normal(uint64_t seed, uint64_t ctr) {
  const uint64_t a = splitmix64(seed + 0x9E3779B97F4A7C15ull * (2 * ctr + 1));
  const uint64_t b = splitmix64(seed + 0x9E3779B97F4A7C15ull * (2 * ctr + 2));
  const double u1 = (double)((a >> 11) + 1) * 1.1102230246251565404e-16;
  const double u2 = (double)(b >> 11) * 1.1102230246251565404e-16;
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}

// out[i][s] = mu[i] + sum_{k <= i} Lp[i][k] z(k, s)
__global__ void fantasy_sample_kernel(const double *__restrict__ Lp, const double *__restrict__ mu, int P, int n,
                                      uint64_t seed, double *__restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P * n) return;
  const int i = e / n, s = e - i * n;
  double acc = mu[i];
  for (int k = 0; k <= i; ++k) acc += Lp[i * 64 + k] * counter_normal(seed, (uint64_t)k * (uint64_t)n + (uint64_t)s);
  out[e] = acc;
}

}  // namespace

int launch_mean_multi(b7_ctx *c, const double *ks, int64_t row0, int64_t rows, int64_t Mtotal, double *mu) {
  PhaseScope ps(c, "mean");
  if (rows % 128) return b7_fail(c, B7_ERR_INVALID, "mean_multi: rows %lld not a multiple of 128", (long long)rows);
  hipLaunchKernelGGL(mean_multi_kernel, dim3((unsigned)(rows / 128), c->yld / 64), dim3(256), 0, c->stream, ks,
                     (const double *)c->alpha.p, c->Npad, c->yld, c->ycols, row0, Mtotal, c->mean, mu);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_gemm_nt(b7_ctx *c, const double *A, int lda, const double *B, int ldb, double *C, int ldc, int m, int n,
                   int k) {
  if (m % 64 || n % 64 || k % 16) return b7_fail(c, B7_ERR_INVALID, "gemm_nt: extents %d %d %d", m, n, k);
  hipLaunchKernelGGL(gemm_nt_kernel, dim3(n / 64, m / 64), dim3(256), 0, c->stream, A, lda, B, ldb, C, ldc, k);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_fantasy_cov(b7_ctx *c, const double *kpp, const double *g, double *S, int P, double diag_add) {
  hipLaunchKernelGGL(fantasy_cov_kernel, dim3(16), dim3(256), 0, c->stream, kpp, g, S, P, diag_add);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_add_diag(b7_ctx *c, double *S, int ld, int n, double v) {
  hipLaunchKernelGGL(add_diag_kernel, dim3(1), dim3(64), 0, c->stream, S, ld, n, v);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_fantasy_sample(b7_ctx *c, const double *Lp, const double *mu, int P, int n, uint64_t seed, double *out) {
  const int total = P * n;
  hipLaunchKernelGGL(fantasy_sample_kernel, dim3((total + 255) / 256), dim3(256), 0, c->stream, Lp, mu, P, n, seed,
                     out);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}
