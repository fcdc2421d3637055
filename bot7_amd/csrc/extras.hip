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
#include "exp_table.h"
#ifndef B7_MLP_ABLATE
#define B7_MLP_ABLATE 0
#endif

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

// ---- the same network with its weights RESIDENT in LDS (the usual DNGO shapes: every layer's padded weights together fit) --
// mlp_forward_mfma_kernel re-stages every layer's weights for every 32 inputs (two block barriers per 64-unit chunk, an
// integer division per staged element) and calls ocml's tanh: one block took 26 us, 65 536 candidates 146 us
// (profiles/r03_cfg5_kernel_stats_before.csv).  Here a block of eight waves stages all layers once ([npad][kpad + 1] per
// layer, zero padded, odd stride) and then loops over 16-input tiles, one tile per wave at a time, with no block barrier
// after the staging: a wave loads its tile's A fragments into registers, so the layer's outputs overwrite its inputs in the
// wave's own [16][as] LDS strip; tanh is 1 - 2 / (e^{2|x|} + 1) with the table exponential of the covariance kernel
// (expm1 form below: no cancellation near zero).  The epilogue writes the z real feature columns (the padding columns of
// the feature matrix are zeroed once, when it is allocated) and, when the head's weights m are given, the posterior mean
// mean0 + phi . m of each input (models/dngo.lua:174's predictive mean; otherwise a second pass over the features).
__constant__ double exp2_tab_blr[256];   // [0, 128): 2^(j/128);  [128, 256): 2^(j/128) - 1, correctly rounded

// tanh|x| = t / (t + 2) with t = e^{2|x|} - 1:  n = rint(y 128/ln2) for y = 2|x| <= 45, r = y - n ln2/128, q = e^r - 1
// (degree-5 Taylor, |r| <= ln2/256), s = 2^(n >> 7):  t = s T (1 + q) - 1 = fma(s T, q, s T - 1), with (T - 1) taken from
// its own table when s = 1 (no cancellation for small arguments); the quotient by reciprocal estimate, two Newton steps
// and one correction (the compiler's IEEE division sequence is twice as long).  Against long-double tanh over 2e7 arguments
// in [1e-8, 30]: relative error <= 5.1 * 2^-53.  tab: the 256-entry table above, copied to LDS by the kernel.
// tanh of four values, stage by stage (cf. amp_exp_nonpos4 in covar.hip: one chain is ~30 dependent fp64 instructions, and
// the compiler emits independent chains one after the other unless it is stopped)
#define B7_STAGE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void tanh_fast4(const double (&x)[4], const double *__restrict__ tab, double (&out)[4]) {
  double y[4], nb[4], nf[4], r[4], p[4], q[4], tj[4], tm1[4], st[4], t[4], d[4], rc[4], e[4];
  int n[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) y[i] = 2.0 * __builtin_fabs(x[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) y[i] = (y[i] > 45.0) ? 45.0 : y[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) nb[i] = __builtin_fma(y[i], B7_EXP_INV, B7_EXP_MAGIC);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    n[i] = __double2loint(nb[i]);
    tj[i] = tab[n[i] & 127];
    tm1[i] = tab[128 + (n[i] & 127)];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) nf[i] = nb[i] - B7_EXP_MAGIC;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __builtin_fma(nf[i], -B7_EXP_HEAD, y[i]);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __builtin_fma(nf[i], -B7_EXP_TAIL, r[i]);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = __builtin_fma(r[i], 1.0 / 120.0, 1.0 / 24.0);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = __builtin_fma(p[i], r[i], 1.0 / 6.0);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = __builtin_fma(p[i], r[i], 0.5);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = __builtin_fma(p[i], r[i], 1.0);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = r[i] * p[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) st[i] = __builtin_ldexp(tj[i], n[i] >> 7);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) tm1[i] = ((n[i] >> 7) == 0) ? tm1[i] : st[i] - 1.0;
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = __builtin_fma(st[i], q[i], tm1[i]);   // e^y - 1
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = t[i] + 2.0;
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) rc[i] = __builtin_amdgcn_rcp(d[i]);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = __builtin_fma(-d[i], rc[i], 1.0);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) rc[i] = __builtin_fma(e[i], rc[i], rc[i]);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = __builtin_fma(-d[i], rc[i], 1.0);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) rc[i] = __builtin_fma(e[i], rc[i], rc[i]);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = t[i] * rc[i];
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = __builtin_fma(-d[i], q[i], t[i]);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = __builtin_fma(e[i], rc[i], q[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = __builtin_copysign(q[i], x[i]);
}
struct MlpResident {      // LDS layout (doubles), computed on the host
  int w_off[4], b_off[4]; // layer l: weights [npad_l][ks_l], biases [npad_l]
  int ks[4], kpad[4], npad[4];
  int m_off;              // the head's weights (z entries), when the mean is fused; else unused
  int tab_off;            // the 256-entry exponential table of tanh_fast
  int act_off, as;        // eight strips [16][as]
};
// NT_MAX: 16-column tiles of the widest layer output (4: widths <= 64, 8: <= 128); ACT: the activation, compile-time
template <int NT_MAX, int ACT>
__global__ void __launch_bounds__(512)
    mlp_resident_kernel(const double *__restrict__ X, int64_t M, int d, const double *__restrict__ net, int n_layers,
                        int d0, int d1, int d2, int d3, int d4, MlpResident lay, double *__restrict__ out,
                        int ld_out, const double *__restrict__ mvec, double mean0, double *__restrict__ mu) {
  extern __shared__ __align__(16) double rsm[];
  const int dims[5] = {d0, d1, d2, d3, d4};
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  {  // every layer's weights and biases, once per block: a wave takes rows wave, wave + 8, ...; all loads of a batch of
     // eight rows are issued before the first LDS store (a load -> store loop paid one memory round trip per row)
    const double *w = net;
    for (int l = 0; l < n_layers; ++l) {
      const int nin = dims[l], nout = dims[l + 1], ks = lay.ks[l], kp = lay.kpad[l], np = lay.npad[l];
      double *wl = rsm + lay.w_off[l], *bl = rsm + lay.b_off[l];
      for (int half = 0; half < 2; ++half) {   // rows wave + 8 i, i < 8, then i in [8, 16) (layers wider than 64 only)
        if (half * 64 >= np) break;
        double v[8][2];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int r = wave + 8 * (i + 8 * half), k = lane + 64 * h;
            v[i][h] = (r < nout && k < nin) ? w[(size_t)r * nin + k] : 0.0;
          }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int r = wave + 8 * (i + 8 * half), k = lane + 64 * h;
            if (r < np && k < kp) wl[r * ks + k] = v[i][h];
          }
      }
      const double *bias = w + (size_t)nout * nin;
      for (int r = tid; r < np; r += 512) bl[r] = (r < nout) ? bias[r] : 0.0;
      w = bias + nout;
    }
    const int z = dims[n_layers];
    if (mvec)
      for (int k = tid; k < z; k += 512) rsm[lay.m_off + k] = mvec[k];
    if (tid < 256) rsm[lay.tab_off + tid] = exp2_tab_blr[tid];
  }
  __syncthreads();
  const double *tab = rsm + lay.tab_off;
  double *act = rsm + lay.act_off + wave * 16 * lay.as;
  const int as = lay.as, z = dims[n_layers];
  const int64_t ntiles = (M + 15) / 16;
  // the inputs of a tile are fetched one tile ahead (a wave has one tile in flight and a global round trip at its start was
  // a fifth of the tile's time): lane (lr, lq) holds columns lq, lq + 4, ... of row lr, up to 8 of them (d <= 32), else the
  // strip is filled directly
  const int kin = (d + 3) & ~3;
  const bool pre_ok = kin <= 32;
  double xin[8];
  auto fetch = [&](int64_t t) {
    int64_t g = t * 16 + lr;
    if (g > M - 1) g = M - 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = lq + 4 * i;
      xin[i] = (k < d) ? X[g * d + k] : 0.0;
    }
  };
  const int64_t tile0 = (int64_t)blockIdx.x * 8 + wave, tstep = (int64_t)gridDim.x * 8;
  if (pre_ok && tile0 < ntiles) fetch(tile0);
  for (int64_t tile = tile0; tile < ntiles; tile += tstep) {
    const int64_t g0 = tile * 16;
    if (pre_ok) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (lq + 4 * i < kin) act[lr * as + lq + 4 * i] = xin[i];
      if (tile + tstep < ntiles) fetch(tile + tstep);
    } else {  // wide inputs: zero padded to a multiple of 4 columns, straight into the strip
      int64_t g = g0 + lr;
      if (g > M - 1) g = M - 1;
      for (int k = lq; k < kin; k += 4) act[lr * as + k] = (k < d) ? X[g * d + k] : 0.0;
    }
    // the layer loop is unrolled so that every layer's shape sits in its own scalar registers for the whole kernel: indexed
    // by a run-time l, each field of `lay` was a scalar load from the argument block (and a wait) per layer per tile
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l >= n_layers) break;
      const int nout = dims[l + 1], ks = lay.ks[l], ksteps = lay.kpad[l] >> 2, ntl = lay.npad[l] >> 4;
      const double *wl = rsm + lay.w_off[l], *bl = rsm + lay.b_off[l];
      // all n-tiles of the layer at once: NT_MAX independent accumulator chains, A and B fragments straight from LDS one
      // k-step ahead of their MFMAs; tiles beyond the layer's last recompute its last tile and are dropped.  Every read
      // of the strip precedes every write (one wave, program order), so the outputs overwrite the inputs in place.
      const double *pa = act + lr * as + lq;
      const double *pb[NT_MAX];
#pragma unroll
      for (int q = 0; q < NT_MAX; ++q) pb[q] = wl + ((q < ntl ? q : ntl - 1) * 16 + lr) * ks + lq;
      d4_t c[NT_MAX];
#pragma unroll
      for (int q = 0; q < NT_MAX; ++q) c[q] = d4_t{0.0, 0.0, 0.0, 0.0};
      double acur = pa[0], bcur[NT_MAX];
#pragma unroll
      for (int q = 0; q < NT_MAX; ++q) bcur[q] = pb[q][0];
#if B7_MLP_ABLATE & 2   // one k-step per layer
      for (int k4 = 0; k4 < 1; ++k4) {
#else
      for (int k4 = 0; k4 < ksteps; ++k4) {
#endif
        const int kn = (k4 + 1 < ksteps) ? 4 * (k4 + 1) : 4 * k4;   // the last step re-reads itself
        const double anxt = pa[kn];
        double bnxt[NT_MAX];
#pragma unroll
        for (int q = 0; q < NT_MAX; ++q) bnxt[q] = pb[q][kn];
#pragma unroll
        for (int q = 0; q < NT_MAX; ++q) c[q] = mfma_f64(acur, bcur[q], c[q]);
        acur = anxt;
#pragma unroll
        for (int q = 0; q < NT_MAX; ++q) bcur[q] = bnxt[q];
      }
#pragma unroll
      for (int q = 0; q < NT_MAX; ++q)
        if (q < ntl) {
          const int col = q * 16 + lr;
          const double bv = bl[col];
          double pre[4], post[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) pre[r] = c[q][r] + bv;
          if (ACT == 1) {
            tanh_fast4(pre, tab, post);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              post[r] = ACT == 2 ? (pre[r] > 0.0 ? pre[r] : 0.0) : ACT == 3 ? 1.0 / (1.0 + exp(-pre[r])) : pre[r];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) act[(lq + 4 * r) * as + col] = (col < nout) ? post[r] : 0.0;
        }
    }
#if B7_MLP_ABLATE & 1   // diagnostic builds only (tools/basis_ablate.py): no feature stores
    if (act[0] == 1.2345e-300)
#endif
    // features: the z real columns of 16 rows; all LDS reads of a 64-column slice first, then the stores
    for (int kb = 0; kb < z; kb += 64) {
      const int k = kb + lane;
      double v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = (k < z) ? act[r * as + k] : 0.0;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (k < z && g0 + r < M) out[(g0 + r) * ld_out + k] = v[r];
    }
    if (mvec) {  // posterior mean: four partial sums per row (k = part mod 4), folded in a fixed order
      const double *mv = rsm + lay.m_off;
      double sacc = 0.0;
      for (int k = lq; k < z; k += 4) sacc = __builtin_fma(act[lr * as + k], mv[k], sacc);
      sacc += __shfl_xor(sacc, 16);
      sacc += __shfl_xor(sacc, 32);
      if (lq == 0 && g0 + lr < M) mu[g0 + lr] = mean0 + sacc;
    }
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

// S heads over the same rows (blockIdx.y = s): y_s[row] = base[s] + sum_k A[row][k] x_s[k]; the same sums as gemv_rows_kernel
__global__ void __launch_bounds__(256)
    gemv_rows_batch_kernel(const double *__restrict__ A, int lda, const double *__restrict__ x, int64_t sx, int n,
                           const double *__restrict__ base, int64_t Mtotal, double *__restrict__ y, int64_t sy) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, sidx = blockIdx.y;
  if (row >= Mtotal) return;
  x += sidx * sx;
  double s = 0.0;
  for (int k = lane; k < n; k += 64) s += A[row * lda + k] * x[k];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) y[sidx * sy + row] = base[sidx] + s;
}

// The same for 64-column rows (the 64-padded features of a DNGO head) with ONE pass over A for all S heads: a THREAD takes a
// row (64 doubles in registers, 512 contiguous bytes) and, per head, walks the binary tree the wave butterfly of
// gemv_rows_kernel adds its 64 products in (level o adds element l and l + o) -- the same sums, no second read of the
// features (ten heads over 65536 x 64 features: 335 MB -> 34 MB of reads).
__global__ void __launch_bounds__(256)
    gemv_rows64_multi_kernel(const double *__restrict__ A, const double *__restrict__ x, int S, const double *__restrict__ base,
                             int64_t Mtotal, double *__restrict__ y, int64_t sy) {
  extern __shared__ double xs[];  // [S][64]
  for (int e = threadIdx.x; e < S * 64; e += blockDim.x) xs[e] = x[e];
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= Mtotal) return;
  double v[64];
  const d2_t *ap = reinterpret_cast<const d2_t *>(A + row * 64);
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const d2_t t = ap[k];
    v[2 * k] = t[0], v[2 * k + 1] = t[1];
  }
  for (int sidx = 0; sidx < S; ++sidx) {
    const double *xv = xs + sidx * 64;
    double p[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) p[k] = v[k] * xv[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int l = 0; l < o; ++l) p[l] = p[l] + p[l + o];
    y[sidx * sy + row] = base[sidx] + p[0];
  }
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

// 2^(j/128) and 2^(j/128) - 1 for tanh_fast, uploaded once per process and device
static int ensure_blr_tables(b7_ctx *c) {
  static bool done[64] = {false};
  if (c->device < 64 && done[c->device]) return B7_OK;
  B7_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(exp2_tab_blr), b7_exp2_tab, sizeof(b7_exp2_tab), 0));
  B7_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(exp2_tab_blr), b7_exp2m1_tab, sizeof(b7_exp2m1_tab), sizeof(b7_exp2_tab)));
  if (c->device < 64) done[c->device] = true;
  return B7_OK;
}

// mvec (nullable, device, z entries) + mean0: also write mu[i] = mean0 + phi_i . mvec (only the resident-weights kernel does
// that; *mean_done tells the caller whether it happened)
int launch_mlp_forward_mean(b7_ctx *c, const double *X, int64_t M, int d, const double *net_dev, const int *dims,
                            int n_layers, int activation, double *out, int ld_out, const double *mvec, double mean0,
                            double *mu, bool *mean_done) {
  PhaseScope ps(c, "basis");
  if (mean_done) *mean_done = false;
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
  {  // resident weights when all layers + eight activation strips fit the CU's LDS
    MlpResident lay{};
    int off = 0, maxk = 0, maxn = 0;
    for (int l = 0; l < n_layers; ++l) {
      lay.kpad[l] = (dd[l] + 3) & ~3;
      lay.npad[l] = (dd[l + 1] + 15) & ~15;
      lay.ks[l] = lay.kpad[l] + 1;
      lay.w_off[l] = off;
      off += lay.npad[l] * lay.ks[l];
      lay.b_off[l] = off;
      off += lay.npad[l];
      maxk = lay.kpad[l] > maxk ? lay.kpad[l] : maxk;
      maxn = lay.npad[l] > maxn ? lay.npad[l] : maxn;
    }
    lay.m_off = off;
    off += (dd[n_layers] + 1) & ~1;
    lay.tab_off = off;
    off += 256;
    lay.as = (maxn > maxk ? maxn : maxk) + 1;
    lay.act_off = off;
    off += 8 * 16 * lay.as;
    const size_t lds = sizeof(double) * (size_t)off;
    if (lds <= 160 * 1024 && maxn <= 128) {
      B7_TRY(ensure_blr_tables(c));
      const int64_t ntiles = (M + 15) / 16;
      int64_t blocks = (ntiles + 7) / 8;
      if (blocks > c->cus) blocks = c->cus;   // persistent: one block per CU stages the weights once
      using Kern = void (*)(const double *, int64_t, int, const double *, int, int, int, int, int, int, MlpResident, double *, int,
                            const double *, double, double *);
      static const Kern kerns[2][4] = {
          {mlp_resident_kernel<4, 0>, mlp_resident_kernel<4, 1>, mlp_resident_kernel<4, 2>, mlp_resident_kernel<4, 3>},
          {mlp_resident_kernel<8, 0>, mlp_resident_kernel<8, 1>, mlp_resident_kernel<8, 2>, mlp_resident_kernel<8, 3>}};
      const Kern kern = kerns[maxn <= 64 ? 0 : 1][activation >= 0 && activation <= 3 ? activation : 0];
      B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, c->stream, X, M, d, net_dev, n_layers, dd[0], dd[1], dd[2],
                         dd[3], dd[4], lay, out, ld_out, mvec, mean0, mu);
      B7_HIP(c, hipGetLastError());
      if (mean_done) *mean_done = mvec != nullptr;
      return B7_OK;
    }
  }
  if (maxw <= 128) {  // MFMA path, weights staged per layer chunk
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

int launch_mlp_forward(b7_ctx *c, const double *X, int64_t M, int d, const double *net_dev, const int *dims,
                       int n_layers, int activation, double *out, int ld_out) {
  return launch_mlp_forward_mean(c, X, M, d, net_dev, dims, n_layers, activation, out, ld_out, nullptr, 0.0, nullptr, nullptr);
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

int launch_gemv_rows_batch(b7_ctx *c, int S, const double *A, int lda, const double *x, int64_t sx, int n, const double *base_dev,
                           int64_t rows, double *y, int64_t sy) {
  PhaseScope ps(c, "mean");
  if (rows <= 0) return B7_OK;
  if (lda == 64 && n == 64 && sx == 64 && S <= 64) {
    hipLaunchKernelGGL(gemv_rows64_multi_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), sizeof(double) * 64 * (size_t)S, c->stream,
                       A, x, S, base_dev, rows, y, sy);
    B7_HIP(c, hipGetLastError());
    return B7_OK;
  }
  hipLaunchKernelGGL(gemv_rows_batch_kernel, dim3((unsigned)((rows + 3) / 4), S), dim3(256), 0, c->stream, A, lda, x, sx, n, base_dev,
                     rows, y, sy);
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
