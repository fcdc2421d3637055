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
// How the tile shape was chosen (A/B in one process on MI355X, N = 2048, 262144 candidates, round 1; all variants
// bitwise identical): 128x128 tile, 1 wave/SIMD 44.6 TF | 2 blocks/CU 59.2 | + odd LDS stride 59.7 | + zero-strip skip
// 61.3 | both 61.6 | both on a 128(n) x 256(cand) tile with 8 waves 63.6 | + static priority 64.7 TF = 82 % of 78.6.
// PMC: MFMA pipe 84 % busy, effective clock 2.1 GHz (DVFS) -> 68 TF is the ceiling at that clock.
//
// Algorithmic work per launch: rows * Npad^2 flops (triangle exploited), bytes: the K* chunk is read
// (t+1)/T-weighted ~ (T+1)/2 times from L2/MALL/HBM (T = Npad/128 n-tiles), Linv once per block from L2.
#include "b7_internal.h"
#include "gemm_f64.h"

namespace {

// BM = rows of L^-1 per n-tile, BN = candidates per block, WM x WN waves of 64x64 accumulators each.
template <int BM, int BN, int WM, int WN, int MINW, int PAD, bool TRI, bool PRIO = false>
__global__ void __launch_bounds__(64 * WM * WN, MINW)
    post_kernel(const double *__restrict__ Linv, const double *__restrict__ ks, int Npad, int64_t row0,
                int64_t Mtotal, double base, double sgn, double var_add, int clamp, double var_min,
                double *__restrict__ var) {
  using GP = GemmF64<BM, BN, 16, WM, WN, false, PAD>;
  static_assert(GP::TM == 4 && GP::TN == 4, "64x64 per wave");
  extern __shared__ __align__(16) double sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double *B = ks + (int64_t)blockIdx.x * BN * Npad;  // this block's BN candidate rows of K*
  double colss[GP::TN] = {};
  // PRIO: with two waves per SIMD the second-dispatched half of the workgroup loses issue arbitration to the older
  // half at the start of every stage; one static priority raise for that half evens it out (the condition must be
  // wave-uniform for the scalar s_setprio to be conditional at all)
  if (PRIO && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 32 * WM * WN) __builtin_amdgcn_s_setprio(1);

  const int ntiles = Npad / BM;
  for (int t = 0; t < ntiles; ++t) {
    d4_t acc[GP::TM][GP::TN] = {};
    GP::template run<TRI>(Linv + (int64_t)t * BM * Npad, Npad, B, Npad, 0, (t + 1) * BM, acc, sm);
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

  // lanes l, l^16, l^32, l^48 hold partial sums of the same candidate column; then the WM row-waves
  double *red = sm;  // [WM][BN]; GP::run ended with a barrier, LDS is free
  const int wm = wave / WN, wn = wave % WN;
#pragma unroll
  for (int j = 0; j < GP::TN; ++j) {
    double v = colss[j];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lane < 16) red[wm * BN + wn * 64 + j * 16 + lane] = v;
  }
  __syncthreads();
  if (threadIdx.x < BN) {
    const int64_t g = row0 + (int64_t)blockIdx.x * BN + threadIdx.x;
    if (g < Mtotal) {
      double ss = red[threadIdx.x];
#pragma unroll
      for (int w = 1; w < WM; ++w) ss += red[w * BN + threadIdx.x];
      double v = (base + sgn * ss) + var_add;  // GP: amp - ss; Bayesian-linear head: 1/beta + ss
      if (clamp) v = (v < var_min) ? var_min : v;  // TH clamp: NaN passes through
      var[g] = v;
    }
  }
}

template <int BM, int BN, int WM, int WN, int MINW, int PAD, bool TRI, bool PRIO = false>
int launch_post_variant(b7_ctx *c, const double *ks, int64_t row0, int64_t rows, int64_t Mtotal, double *var) {
  using GP = GemmF64<BM, BN, 16, WM, WN, false, PAD>;
  auto kern = post_kernel<BM, BN, WM, WN, MINW, PAD, TRI, PRIO>;
  const int lds = GP::LDS_BYTES;
  B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const bool blr = c->model_kind == 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)(rows / BN)), dim3(64 * WM * WN), lds, c->stream, (const double *)c->Linv.p,
                     ks, c->Npad, row0, Mtotal, blr ? 0.0 : c->amp, blr ? 1.0 : -1.0,
                     blr ? c->noise : (c->opts.var_with_noise ? c->noise : 0.0), c->opts.var_clamp, c->opts.var_min, var);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

}  // namespace

int launch_post(b7_ctx *c, const double *ks, int64_t row0, int64_t rows, int64_t Mtotal, double *var) {
  PhaseScope ps(c, "post");
  if (rows % B7_MROWS) return b7_fail(c, B7_ERR_INVALID, "post: rows %lld not a multiple of %d", (long long)rows, B7_MROWS);
  // Two shapes, bitwise identical in output (the A/B ladder that led here is in DESIGN.md section 8): 128(n) x 256(cand)
  // tile with 8 waves, odd LDS stride, zero-strip skip and a static priority raise for the younger half of the waves;
  // with fewer 256-candidate workgroups than CUs the 128-wide tile fills the chip (N = 2048, M = 32768: 2.6 vs 4.2 ms;
  // N = 256: 63 vs 83 us; a 64-wide one was slower again, 4.1 ms)
  if (rows / 256 < c->cus) return launch_post_variant<128, 128, 2, 2, 2, 1, true>(c, ks, row0, rows, Mtotal, var);
  return launch_post_variant<128, 256, 2, 4, 2, 1, true, true>(c, ks, row0, rows, Mtotal, var);
}
