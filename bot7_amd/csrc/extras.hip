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
#include <stdlib.h>

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

// ---- DNGO: basis network forward and the pieces of the Bayesian-linear head -------------------------------------------
// Reference: models/dngo.lua:155-171 pushes X through `network` in minibatches and keeps `self.basis.output`
// (the activations of the last hidden layer: nn.Linear / activation stacks from nnTools/builder.lua:118-151);
// :174 hands those features to gp.models.bayes_linear (absent `gp` package), restated in api.hip.
// One block = 32 inputs; activations ping-pong between two LDS buffers (dynamic, 2 x 32 x (maxw+1) doubles);
// thread t computes units (t / 32 + 8 i) of input (t % 32).  Weights (row-major out x in, nn.Linear's layout, each
// followed by its bias) stream from L2.
constexpr int MLP_MAXW = 256;
constexpr int MLP_IN = 32;
__device__ __forceinline__ double mlp_act(double v, int kind) {
  switch (kind) {
    case 1: return tanh(v);
    case 2: return v > 0.0 ? v : 0.0;
    case 3: return 1.0 / (1.0 + exp(-v));
    default: return v;
  }
}
__global__ void __launch_bounds__(256)
    mlp_forward_kernel(const double *__restrict__ X, int64_t M, int d, const double *__restrict__ net, int n_layers,
                       int d0, int d1, int d2, int d3, int d4, int activation, int stride, double *__restrict__ out,
                       int ld_out) {
  extern __shared__ __align__(16) double msm[];
  double *src = msm, *dst = msm + MLP_IN * stride;
  const int dims[5] = {d0, d1, d2, d3, d4};
  const int ci = threadIdx.x & (MLP_IN - 1), u0 = threadIdx.x / MLP_IN;
  constexpr int UG = 256 / MLP_IN;
  int64_t g = (int64_t)blockIdx.x * MLP_IN + ci;
  const bool live = g < M;
  if (!live) g = M - 1;
  for (int k = u0; k < d; k += UG) src[ci * stride + k] = X[g * d + k];
  __syncthreads();
  const double *w = net;
  for (int l = 0; l < n_layers; ++l) {
    const int nin = dims[l], nout = dims[l + 1];
    const double *bias = w + (size_t)nout * nin;
    for (int o = u0; o < nout; o += UG) {
      double s = bias[o];
      const double *wr = w + (size_t)o * nin;
      for (int k = 0; k < nin; ++k) s += wr[k] * src[ci * stride + k];
      dst[ci * stride + o] = mlp_act(s, activation);
    }
    __syncthreads();
    double *tmp = src;
    src = dst;
    dst = tmp;
    w = bias + nout;
  }
  const int z = dims[n_layers];
  if (live)
    for (int k = u0; k < ld_out; k += UG) out[g * ld_out + k] = (k < z) ? src[ci * stride + k] : 0.0;
}

// MFMA version for widths <= 128 (the usual DNGO shapes): a block owns 32 inputs; activations ping-pong in LDS
// ([32][stride], odd stride, zero-padded to multiples of 4 in K and 16 in N); each layer's weights are staged through
// LDS 64 output units at a time ([64][stride]); wave w computes M-tile (w & 1) x N-tiles 2 (w >> 1) + {0, 1} of the
// chunk as act (A operand, [m][k]) times W' (B operand, W is [n][k]); epilogue adds the bias, applies the
// activation and writes the next layer's input (padding columns written as zeros).
__global__ void __launch_bounds__(256)
    mlp_forward_mfma_kernel(const double *__restrict__ X, int64_t M, int d, const double *__restrict__ net,
                            int n_layers, int d0, int d1, int d2, int d3, int d4, int activation, int stride,
                            double *__restrict__ out, int ld_out) {
  extern __shared__ __align__(16) double msm[];
  double *src = msm, *dst = msm + 32 * stride, *wl = msm + 64 * stride;  // wl: [64][stride]
  const int dims[5] = {d0, d1, d2, d3, d4};
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  const int64_t g0 = (int64_t)blockIdx.x * 32;
  // input rows, zero padded to a multiple of 4 columns
  const int kin0 = (d + 3) & ~3;
  for (int e = tid; e < 32 * kin0; e += 256) {
    const int r = e / kin0, k = e - r * kin0;
    int64_t g = g0 + r;
    if (g > M - 1) g = M - 1;
    src[r * stride + k] = (k < d) ? X[g * d + k] : 0.0;
  }
  const double *w = net;
  for (int l = 0; l < n_layers; ++l) {
    const int nin = dims[l], nout = dims[l + 1];
    const int kpad = (nin + 3) & ~3, npad = (nout + 15) & ~15;
    const double *bias = w + (size_t)nout * nin;
    for (int n0 = 0; n0 < npad; n0 += 64) {
      __syncthreads();  // src complete (first chunk) / previous chunk's wl consumed
      for (int e = tid; e < 64 * kpad; e += 256) {
        const int r = e / kpad, k = e - r * kpad;
        wl[r * stride + k] = (n0 + r < nout && k < nin) ? w[(size_t)(n0 + r) * nin + k] : 0.0;
      }
      __syncthreads();
      const int mt = wave & 1;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int nt = (wave >> 1) * 2 + t;
        if (n0 + nt * 16 >= npad) continue;
        d4_t c = {0.0, 0.0, 0.0, 0.0};
        const double *pa = src + (mt * 16 + lr) * stride + lq;
        const double *pb = wl + (nt * 16 + lr) * stride + lq;
        for (int k4 = 0; k4 < kpad; k4 += 4) c = mfma_f64(pa[k4], pb[k4], c);
        const int col = n0 + nt * 16 + lr;
        const double bv = (col < nout) ? bias[col] : 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          dst[(mt * 16 + lq + 4 * r) * stride + col] = (col < nout) ? mlp_act(c[r] + bv, activation) : 0.0;
      }
    }
    __syncthreads();
    double *tmp = src;
    src = dst;
    dst = tmp;
    w = bias + nout;
  }
  const int z = dims[n_layers];
  for (int e = tid; e < 32 * ld_out; e += 256) {
    const int r = e / ld_out, k = e - r * ld_out;
    if (g0 + r < M) out[(g0 + r) * ld_out + k] = (k < z) ? src[r * stride + k] : 0.0;
  }
}

// y[row] = base + sum_k A[row][k] x[k]  (one wave per row)
__global__ void __launch_bounds__(256)
    gemv_rows_kernel(const double *__restrict__ A, int lda, const double *__restrict__ x, int n, double base,
                     int64_t row0, int64_t Mtotal, double *__restrict__ y) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row0 + row >= Mtotal) return;
  double s = 0.0;
  for (int k = lane; k < n; k += 64) s += A[row * lda + k] * x[k];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) y[row0 + row] = base + s;
}

// Zt[k][i] = Z[i][k] (Z: n x ldz row-major, Zt: zpad x nk, zero padded) -- the layout gemm_nt wants for Z'Z
__global__ void transpose_pad_kernel(const double *__restrict__ Z, int n, int ldz, int z, double *__restrict__ Zt,
                                     int zpad, int nk) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= zpad * nk) return;
  const int k = e / nk, i = e - k * nk;
  Zt[e] = (k < z && i < n) ? Z[(int64_t)i * ldz + k] : 0.0;
}

// K = beta * G + alpha_prec * I on the z x z corner, identity in the padding
__global__ void blr_assemble_kernel(const double *__restrict__ G, double *__restrict__ K, int z, int zpad,
                                    double alpha_prec, double beta) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= zpad * zpad) return;
  const int i = e / zpad, j = e - i * zpad;
  double v;
  if (i < z && j < z) {
    v = beta * G[e];
    if (i == j) v = v + alpha_prec;
  } else {
    v = (i == j) ? 1.0 : 0.0;
  }
  K[e] = v;
}

// ---- incremental fit: append one observation (same hypers) ----------------------------------------------------
// With K' = [K k; k' kappa]:  L' = [L 0; l' lambda],  l = L^-1 k,  lambda^2 = kappa - l'l,
//                             inv(L') = [inv(L) 0; -(inv(L)' l)'/lambda, 1/lambda].
// One block: reduces l'l in a fixed order, then writes row/column N of K and row N of L and L^-1.
// status[0] = 1 if lambda^2 is not positive (the caller then refits from scratch with the jitter schedule).
__global__ void __launch_bounds__(256)
    append_finalize_kernel(const double *__restrict__ krow, const double *__restrict__ lvec,
                           const double *__restrict__ uvec, const double *__restrict__ evec, double noise, int N,
                           int Npad, double *__restrict__ K, double *__restrict__ L, double *__restrict__ Linv,
                           int *__restrict__ status) {
  __shared__ double red[256], rede[256];
  __shared__ double lam_s;
  double s = 0.0, se = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) {
    s += lvec[i] * lvec[i];
    se += fabs(lvec[i]) * evec[i];
  }
  red[threadIdx.x] = s;
  rede[threadIdx.x] = se;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[threadIdx.x] += red[threadIdx.x + o];
      rede[threadIdx.x] += rede[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double kappa = krow[N] + noise;
    const double l2 = kappa - red[0];
    // The new pivot l2 = kappa - |Linv k|^2 must stand clear of the rounding error of its own evaluation:
    // |delta l_i| <= (N+2) u e_i with e = |Linv||k|, so |delta l2| <= 2 (N+2) u sum |l_i| e_i + (N+2) u |l|^2.
    // Below that the sign of l2 is noise (an exact duplicate of an observation with no noise term lands here)
    // and the extension is refused, as for a non-positive pivot; the caller refits with the jitter schedule.
    const double u = 1.1102230246251565e-16, g = (double)(N + 2) * u;
    const double tol = 2.0 * g * rede[0] + g * red[0];
    if (!(l2 > tol)) {
      status[0] = 1;
      lam_s = 0.0;
    } else {
      status[0] = 0;
      lam_s = sqrt(l2);
    }
  }
  __syncthreads();
  const double lam = lam_s;
  if (!(lam > 0.0)) return;
  for (int j = threadIdx.x; j < Npad; j += 256) {
    if (j < N) {
      K[(int64_t)N * Npad + j] = krow[j];
      K[(int64_t)j * Npad + N] = krow[j];
      L[(int64_t)N * Npad + j] = lvec[j];
      Linv[(int64_t)N * Npad + j] = -uvec[j] / lam;
    } else if (j == N) {
      K[(int64_t)N * Npad + N] = krow[N] + noise;
      L[(int64_t)N * Npad + N] = lam;
      Linv[(int64_t)N * Npad + N] = 1.0 / lam;
    }
  }
}

}  // namespace

int launch_append_finalize(b7_ctx *c, const double *krow, const double *lvec, const double *uvec, const double *evec,
                           int *status_dev) {
  hipLaunchKernelGGL(append_finalize_kernel, dim3(1), dim3(256), 0, c->stream, krow, lvec, uvec, evec, c->noise, c->N,
                     c->Npad, (double *)c->K.p, (double *)c->L.p, (double *)c->Linv.p, status_dev);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_mlp_forward(b7_ctx *c, const double *X, int64_t M, int d, const double *net_dev, const int *dims,
                       int n_layers, int activation, double *out, int ld_out) {
  PhaseScope ps(c, "basis");
  if (n_layers < 1 || n_layers > 4) return b7_fail(c, B7_ERR_UNSUPPORTED, "mlp: 1..4 weighted layers supported");
  int dd[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i <= n_layers; ++i) {
    dd[i] = dims[i];
    if (dims[i] < 1 || dims[i] > MLP_MAXW) return b7_fail(c, B7_ERR_UNSUPPORTED, "mlp: layer width %d > %d", dims[i], MLP_MAXW);
  }
  if (dims[0] != d) return b7_fail(c, B7_ERR_INVALID, "mlp: input width %d != grid dims %d", dims[0], d);
  if (M <= 0) return B7_OK;
  int maxw = 0;
  for (int i = 0; i <= n_layers; ++i) maxw = dd[i] > maxw ? dd[i] : maxw;
  if (maxw <= 128) {  // MFMA path
    const int stride = ((maxw + 15) & ~15) + 1;
    const int lds = (64 + 64) * stride * (int)sizeof(double);
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_forward_mfma_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(mlp_forward_mfma_kernel, dim3((unsigned)((M + 31) / 32)), dim3(256), lds, c->stream, X, M, d,
                       net_dev, n_layers, dd[0], dd[1], dd[2], dd[3], dd[4], activation, stride, out, ld_out);
    B7_HIP(c, hipGetLastError());
    return B7_OK;
  }
  const int stride = maxw + 1;
  const int lds = 2 * MLP_IN * stride * (int)sizeof(double);
  B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_forward_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(mlp_forward_kernel, dim3((unsigned)((M + MLP_IN - 1) / MLP_IN)), dim3(256), lds, c->stream, X, M,
                     d, net_dev, n_layers, dd[0], dd[1], dd[2], dd[3], dd[4], activation, stride, out, ld_out);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_gemv_rows(b7_ctx *c, const double *A, int lda, const double *x, int n, double base, int64_t row0,
                     int64_t rows, int64_t Mtotal, double *y) {
  PhaseScope ps(c, "mean");
  if (rows <= 0) return B7_OK;
  hipLaunchKernelGGL(gemv_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, c->stream, A, lda, x, n, base,
                     row0, Mtotal, y);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_transpose_pad(b7_ctx *c, const double *Z, int n, int ldz, int z, double *Zt, int zpad, int nk) {
  hipLaunchKernelGGL(transpose_pad_kernel, dim3((zpad * nk + 255) / 256), dim3(256), 0, c->stream, Z, n, ldz, z, Zt,
                     zpad, nk);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_blr_assemble(b7_ctx *c, const double *G, double *K, int z, int zpad, double alpha_prec, double beta) {
  hipLaunchKernelGGL(blr_assemble_kernel, dim3((zpad * zpad + 255) / 256), dim3(256), 0, c->stream, G, K, z, zpad,
                     alpha_prec, beta);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

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
