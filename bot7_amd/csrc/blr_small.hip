// The Bayesian-linear head of models/dngo.lua (:125-153 fit, :174-175 predict's weights) for z <= 64 basis features in ONE
// workgroup of ONE launch:   A = alpha_prec I + beta Z' Z,   b = Z' (beta (y - mean)),   A = L L',   L^-1,   m = A^-1 b.
//
// The general path of b7_blr_fit builds the same head from launches of its own -- a transposed copy of the features, a GEMM
// for Z'Z, the assembly of A, a matrix-vector product for b, a memset, the persistent factorisation, three triangular
// matrix-vector launches -- nine dispatches of 2 ... 16 us for a 50 x 50 system.  b7_blr_eval_nominate (one nomination per
// trial, the head refitted every time because the network is: bots/bayesopt.lua:65-66) takes this kernel instead when the
// padded width is one 64-block; a failed pivot is reported, and the caller redoes the fit through b7_blr_fit_x and its jitter
// schedule (utils/math.lua:159-218), exactly as before.
//
// Arithmetic: G = Z'Z per 16 x 16 tile as one chain of v_mfma_f64_16x16x4 over the observations in ascending order (tiles
// on and below the diagonal; the factor routine reads nothing above it), A = beta G (+ alpha_prec on the diagonal) as
// blr_assemble_kernel forms it, identity in the padding; the block through b7diag::diag_core (potrf_diag.h), the routine of
// every other factorisation here; t = L^-1 b and m = L^-T t as fma chains over ascending indices.
#include "b7_internal.h"
#include "potrf_diag.h"

namespace {
using namespace b7diag;  // NB = 64, DLD, TLD, diag_core
typedef double d2_t __attribute__((ext_vector_type(2)));

constexpr int ZLD = NB + 1;  // row stride of a 64-observation chunk of Z in LDS
constexpr int BLR_SMALL_LDS_DOUBLES = 2 * NB * DLD + 32 * TLD + NB * ZLD + 4 * NB;

__global__ void __launch_bounds__(256)
    blr_head_small_kernel(const double *__restrict__ Z, int N, int z, int ldz, const double *__restrict__ yv, double alpha_prec,
                          double beta, double *__restrict__ Lout, double *__restrict__ Linv, double *__restrict__ mvec,
                          double *__restrict__ bout, int *__restrict__ info, int *__restrict__ report) {
  extern __shared__ __align__(16) double sm[];
  __shared__ int inf[4];
  double *A = sm, *X = A + NB * DLD, *T = X + NB * DLD, *Zc = T + 32 * TLD, *bv = Zc + NB * ZLD, *tv = bv + NB, *yc = tv + NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  if (tid < 4) inf[tid] = 0;
  // the wave's tiles of G (on and below the diagonal, dealt round-robin): q = wave, wave + 4, wave + 8 (< 10)
  d4_t acc[3] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  // b_i = sum_n Z[n][i] yv[n]: thread (i = tid & 63, quarter = tid >> 6) sums rows 16 quarter .. of every chunk in ascending
  // order; the four quarters are added in order at the end
  double bsum = 0.0;
  // a chunk of 64 observations: sixteen elements per thread (rows (tid >> 6) + 4 i, column tid & 63), loaded from clamped
  // addresses so that all sixteen are in flight at once, zeroed outside N x z afterwards; the NEXT chunk's loads are issued
  // before this chunk is consumed
  const int col = tid & 63, rq = tid >> 6, colc = col < z ? col : z - 1;
  double pv[16], py = 0.0;
  auto fetch = [&](int n0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + rq + 4 * i, nc = n < N ? n : N - 1;
      pv[i] = Z[(int64_t)nc * ldz + colc];
    }
    if (tid < NB) py = yv[(n0 + tid < N) ? n0 + tid : N - 1];
  };
  fetch(0);
  for (int n0 = 0; n0 < N; n0 += NB) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = rq + 4 * i;
      Zc[r * ZLD + col] = (n0 + r < N && col < z) ? pv[i] : 0.0;
    }
    if (tid < NB) yc[tid] = (n0 + tid < N) ? py : 0.0;
    __syncthreads();
    if (n0 + NB < N) fetch(n0 + NB);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int q = wave + 4 * s;
      if (q < 10) {
        const int it = (q >= 1) + (q >= 3) + (q >= 6), jt = q - ((it * (it + 1)) >> 1);
        const double *ap = Zc + lq * ZLD + 16 * it + lr, *bp = Zc + lq * ZLD + 16 * jt + lr;
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) acc[s] = mfma_f64(ap[4 * k4 * ZLD], bp[4 * k4 * ZLD], acc[s]);
      }
    }
#pragma unroll
    for (int r = 16 * rq; r < 16 * rq + 16; ++r) bsum = __builtin_fma(Zc[r * ZLD + col], yc[r], bsum);
    __syncthreads();
  }
  // A = beta G + alpha_prec I on the z x z corner, identity in the padding (blr_assemble_kernel); X = 0
  for (int e = tid; e < NB * DLD; e += 256) X[e] = 0.0;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int q = wave + 4 * s;
    if (q < 10) {
      const int it = (q >= 1) + (q >= 3) + (q >= 6), jt = q - ((it * (it + 1)) >> 1);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int i = 16 * it + lq + 4 * rr, j = 16 * jt + lr;
        double v;
        if (i < z && j < z) {
          v = beta * acc[s][rr];
          if (i == j) v = v + alpha_prec;
        } else {
          v = (i == j) ? 1.0 : 0.0;
        }
        A[i * DLD + j] = v;
      }
    }
  }
  tv[tid & 63] = 0.0;
  T[tid] = bsum;  // T is free until the factor routine: the four quarters' partial sums, [quarter][i]
  __syncthreads();
  if (tid < NB) bv[tid] = tid < z ? ((T[tid] + T[64 + tid]) + (T[128 + tid] + T[192 + tid])) : 0.0;
  __syncthreads();
  if (tid < 256) {  // rows 0..15 x columns 48..63: I_16 (potrf_diag.h: the right-hand side of the inversion)
    const int i = tid >> 4, j = tid & 15;
    A[i * DLD + 48 + j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  diag_core<1, false>(A, X, T, 0, inf, nullptr);  // A -> L (lower), X = L^-1; ends with a barrier
  {
    // t = L^-1 b: four lanes per row, ascending columns within each quarter, then the quarters in order
    const int row = tid >> 2, part = tid & 3;
    double a = 0.0;
    for (int k = 16 * part; k < 16 * part + 16; ++k) a = __builtin_fma(X[row * DLD + k], bv[k], a);
    a += __shfl_xor(a, 1);
    a += __shfl_xor(a, 2);
    if (part == 0) tv[row] = a;
  }
  __syncthreads();
  if (tid < NB) {  // m_j = sum_{i >= j} L^-1[i][j] t_i, ascending i
    double a = 0.0;
    for (int i = tid; i < NB; ++i) a = __builtin_fma(X[i * DLD + tid], tv[i], a);
    mvec[tid] = tid < z ? a : 0.0;
    bout[tid] = bv[tid];
  }
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e >> 6, j = e & 63;
    Lout[e] = j <= i ? A[i * DLD + j] : 0.0;
    Linv[e] = j <= i ? X[i * DLD + j] : 0.0;
  }
  __syncthreads();
  if (tid < 4) {
    info[tid] = inf[tid];
    if (report) report[tid] = inf[tid];
  }
}


// ---- the GP fit's factorisation when the padded system is ONE 64-block (N <= 64) --------------------------------------
// K (+ eps I on the real rows, utils/math.lua:190) -> L, L^-1 (= the block's dinv), pivot report, and -- with one response
// column -- alpha = L^-T (L^-1 r) as well: the memset, the persistent launch and the three triangular matrix-vector launches of
// the general schedule in ONE workgroup (grid.x = fit for the S hyper samples of a nomination).  The block goes through
// b7diag::diag_core exactly as the persistent schedule's single panel does -- L and L^-1 are the same bits --; alpha is summed
// as blr_head_small_kernel sums the head's weights.
__global__ void __launch_bounds__(256)
    potrf_small64_kernel(const double *__restrict__ K, double *__restrict__ L, double *__restrict__ Linv, double *__restrict__ dinv,
                         const double *__restrict__ resid, double *__restrict__ alpha, int nreal, double extra,
                         int *__restrict__ info, int *__restrict__ report, int64_t sK, int64_t sL, int64_t sdinv, int64_t svec,
                         int sinfo) {
  extern __shared__ __align__(16) double sm[];
  __shared__ int inf[4];
  double *A = sm, *X = A + NB * DLD, *T = X + NB * DLD, *rv = T + 32 * TLD, *tv = rv + NB;
  const int b = blockIdx.x, tid = threadIdx.x;
  K += b * sK;
  L += b * sL;
  if (Linv) Linv += b * sL;
  dinv += b * sdinv;
  info += (int64_t)b * sinfo;
  if (report) report += (int64_t)b * sinfo;
  if (tid < 4) inf[tid] = 0;
  for (int e = tid; e < NB * NB / 2; e += 256) {
    const int r = e >> 5, c2 = 2 * (e & 31);
    d2_t v = *reinterpret_cast<const d2_t *>(K + (int64_t)r * NB + c2);
    if (r < nreal) {  // potrf_persist.hip: add_extra
      if (c2 == r) v[0] = v[0] + extra;
      if (c2 + 1 == r) v[1] = v[1] + extra;
    }
    A[r * DLD + c2] = v[0];
    A[r * DLD + c2 + 1] = v[1];
  }
  for (int e = tid; e < NB * DLD; e += 256) X[e] = 0.0;
  if (resid && tid < NB) rv[tid] = resid[b * svec + tid];
  __syncthreads();
  {
    const int i = tid >> 4, j = tid & 15;
    A[i * DLD + 48 + j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  diag_core<1, false>(A, X, T, 0, inf, nullptr);
  if (resid) {
    const int row = tid >> 2, part = tid & 3;
    double a = 0.0;
    for (int k = 16 * part; k < 16 * part + 16; ++k) a = __builtin_fma(X[row * DLD + k], rv[k], a);
    a += __shfl_xor(a, 1);
    a += __shfl_xor(a, 2);
    if (part == 0) tv[row] = a;
  }
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e >> 6, j = e & 63;
    const double x = j <= i ? X[i * DLD + j] : 0.0;
    L[e] = j <= i ? A[i * DLD + j] : 0.0;
    if (Linv) Linv[e] = x;
    dinv[e] = x;
  }
  __syncthreads();
  if (resid && tid < NB) {
    double a = 0.0;
    for (int i = tid; i < NB; ++i) a = __builtin_fma(X[i * DLD + tid], tv[i], a);
    alpha[b * svec + tid] = tid < nreal ? a : 0.0;
  }
  if (tid < 4) {
    info[tid] = inf[tid];
    if (report) report[tid] = inf[tid];
  }
}

}  // namespace

// Z: N x z features of the observations (row stride ldz), yv[N] = beta (y - mean), both on the device.  Writes L, L^-1
// (64 x 64, row stride 64), the head's weights and b into the context's fit slot, the pivot report into c->info and (when given)
// into mapped host memory.
int launch_blr_head_small(b7_ctx *c, const double *Z, int N, int z, int ldz, const double *yv, double alpha_prec, double beta,
                          int *report_dev) {
  PhaseScope ps(c, "potrf");
  if (z < 1 || z > NB || c->Npad != NB) return b7_fail(c, B7_ERR_INVALID, "blr_head_small: z %d, padded %d", z, c->Npad);
  const size_t lds = sizeof(double) * BLR_SMALL_LDS_DOUBLES;
  static bool attr_done[64] = {false};
  if (c->device >= 64 || !attr_done[c->device]) {
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(blr_head_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (c->device < 64) attr_done[c->device] = true;
  }
  hipLaunchKernelGGL(blr_head_small_kernel, dim3(1), dim3(256), lds, c->stream, Z, N, z, ldz, yv, alpha_prec, beta, (double *)c->L.p,
                     (double *)c->Linv.p, (double *)c->alpha.p, (double *)c->resid.p, (int *)c->info.p, report_dev);
  B7_HIP(c, hipGetLastError());
  c->linv_done = true;
  return B7_OK;
}

// B fits whose padded system is one 64-block: K_b -> L_b, Linv_b (nullable), dinv_b, info_b (+ a copy in mapped host memory
// when report_dev is given) and, when resid is given (one response column), alpha_b.  Strides in doubles / ints.
int launch_potrf_small(b7_ctx *c, int B, const double *K, double *L, double *Linv, double *dinv, const double *resid, double *alpha,
                       double extra, int *info, int *report_dev, int64_t sK, int64_t sL, int64_t sdinv, int64_t svec, int sinfo) {
  PhaseScope ps(c, "potrf");
  if (c->Npad != NB) return b7_fail(c, B7_ERR_INVALID, "potrf_small: padded size %d", c->Npad);
  const size_t lds = sizeof(double) * (2 * NB * DLD + 32 * TLD + 2 * NB);
  static bool attr_done[64] = {false};
  if (c->device >= 64 || !attr_done[c->device]) {
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_small64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (c->device < 64) attr_done[c->device] = true;
  }
  hipLaunchKernelGGL(potrf_small64_kernel, dim3(B), dim3(256), lds, c->stream, K, L, Linv, dinv, resid, alpha, c->N, extra, info,
                     report_dev, sK, sL, sdinv, svec, sinfo);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}
