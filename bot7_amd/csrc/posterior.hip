// Posterior variance over a chunk of candidates:  var_j = amp - || L^-1 k*_j ||^2.
//
// Reference arithmetic replaced: the variance half of model:predict(X_obs, Y_obs, X_hid, hyp, {mean, var})
// (call sites scores/expected_improvement.lua:63, scores/confidence_bound.lua:63; gp_regressor itself is in
// the absent `gp` package): textbook GP regression, V = L^-1 K(X*,X)', var = k** - colsumsq(V).
//
// This is the dominant kernel of the whole path: M * N^2 flops (4.19 MFLOP per candidate at N = 2048), all
// of it on v_mfma_f64_16x16x4_f64.  V is never stored: a block owns 128 candidates, walks the 128-row
// tiles of the explicit inverse factor top to bottom (k only up to the diagonal: Linv is lower triangular)
// and folds each finished 128x128 tile of V into per-candidate sums of squares held in registers.
// The summation order is fixed (n-tiles ascending, then a fixed shuffle/LDS tree), so results are bitwise
// reproducible run to run -- the arg-max downstream depends on that.
//
// Algorithmic work per launch: rows * Npad^2 flops (triangle exploited), bytes: the K* chunk is read
// (t+1)/T-weighted ~ (T+1)/2 times from L2/MALL/HBM (T = Npad/128 n-tiles), Linv once per block from L2.
#include "b7_internal.h"
#include "gemm_f64.h"

namespace {

using GP = GemmF64<128, 128, 16, 2, 2, false>;

__global__ void __launch_bounds__(256)
    post_kernel(const double *__restrict__ Linv, const double *__restrict__ ks, int Npad, int64_t row0,
                int64_t Mtotal, double amp, double var_add, int clamp, double var_min, double *__restrict__ var) {
  extern __shared__ __align__(16) double sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double *B = ks + (int64_t)blockIdx.x * 128 * Npad;  // this block's 128 candidate rows of K*
  double colss[GP::TN] = {};

  const int ntiles = Npad / 128;
  for (int t = 0; t < ntiles; ++t) {
    d4_t acc[GP::TM][GP::TN] = {};
    GP::run(Linv + (int64_t)t * 128 * Npad, Npad, B, Npad, 0, (t + 1) * 128, acc, sm);
#pragma unroll
    for (int j = 0; j < GP::TN; ++j) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < GP::TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][j][r] * acc[i][j][r];
      colss[j] += s;
    }
  }

  // lanes l, l^16, l^32, l^48 hold partial sums of the same candidate column
  double *red = sm;  // [2 (wm)][128]; GP::run ended with a barrier, LDS is free
#pragma unroll
  for (int j = 0; j < GP::TN; ++j) {
    double v = colss[j];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lane < 16) red[(wave / 2) * 128 + (wave % 2) * 64 + j * 16 + lane] = v;
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int64_t g = row0 + (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (g < Mtotal) {
      double ss = red[threadIdx.x] + red[128 + threadIdx.x];
      double v = (amp - ss) + var_add;
      if (clamp) v = (v < var_min) ? var_min : v;  // TH clamp: NaN passes through
      var[g] = v;
    }
  }
}

}  // namespace

int launch_post(b7_ctx *c, const double *ks, int64_t row0, int64_t rows, int64_t Mtotal, double *var) {
  PhaseScope ps(c, "post");
  if (rows % 128) return b7_fail(c, B7_ERR_INVALID, "post: rows %lld not a multiple of 128", (long long)rows);
  const int lds = GP::LDS_BYTES;
  B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(post_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(post_kernel, dim3((unsigned)(rows / 128)), dim3(256), lds, c->stream,
                     (const double *)c->Linv.p, ks, c->Npad, row0, Mtotal, c->amp,
                     c->opts.var_with_noise ? c->noise : 0.0, c->opts.var_clamp, c->opts.var_min, var);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}
