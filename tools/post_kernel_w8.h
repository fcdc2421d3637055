// The eight-wave posterior kernel of rounds 1-2 (two waves per SIMD, 64 x 64 accumulators per wave, compiler-scheduled):
// kept OUT of the shipped library, as the independent second implementation tools/post_probe.hip runs beside
// post_kernel_w4 against the host model of the arithmetic.  (diagnostic tool, not product)
#pragma once
#include "../bot7_amd/csrc/gemm_f64.h"

namespace w8 {
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
        for (int r = 0; r < 4; ++r) s = __builtin_fma(acc[i][j][r], acc[i][j][r], s);  // explicit: the order is part of the result
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
}  // namespace w8
