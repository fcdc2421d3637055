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

// Batch mode (hyp3 != nullptr: b7_blr_eval_nominate_marg, the S (alpha, beta, mean) samples of a marginalised head): blockIdx.x =
// sample; hyp3 = [S alpha | S beta | S mean] on the device, yv holds the RAW responses and beta (y - mean) is formed here (the
// same rounded operations as the host forms for a single head); outputs are strided per sample (64 x 64 matrices, 64-vectors,
// 4 ints) and terms[3 s ..] receives what the evidence needs: sum log L_kk, b'm, sum (y - mean)^2.
__global__ void __launch_bounds__(256)
    blr_head_small_kernel(const double *__restrict__ Z, int N, int z, int ldz, const double *__restrict__ yv, double alpha_prec,
                          double beta, double *__restrict__ Lout, double *__restrict__ Linv, double *__restrict__ mvec,
                          double *__restrict__ bout, int *__restrict__ info, int *__restrict__ report,
                          const double *__restrict__ hyp3, int S, double *__restrict__ terms) {
  extern __shared__ __align__(16) double sm[];
  __shared__ int inf[4];
  double *A = sm, *X = A + NB * DLD, *T = X + NB * DLD, *Zc = T + 32 * TLD, *bv = Zc + NB * ZLD, *tv = bv + NB, *yc = tv + NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  double mean_s = 0.0;
  if (hyp3) {
    const int sidx = blockIdx.x;
    alpha_prec = hyp3[sidx], beta = hyp3[S + sidx], mean_s = hyp3[2 * S + sidx];
    Lout += (size_t)sidx * NB * NB, Linv += (size_t)sidx * NB * NB, mvec += (size_t)sidx * NB, bout += (size_t)sidx * NB;
    info += 4 * sidx;
    if (report) report += 4 * sidx;
  }
  double rr = 0.0;  // batch mode: sum (y - mean)^2 over this thread's share (threads 0..63, ascending chunks)
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
    if (tid < NB) {
      double yvn = py;
      if (hyp3) {  // beta (y - mean), and the evidence's sum of squared residuals on the way
        const double res = py - mean_s;
        yvn = beta * res;
        if (n0 + tid < N) rr = __builtin_fma(res, res, rr);
      }
      yc[tid] = (n0 + tid < N) ? yvn : 0.0;
    }
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
  if (terms && wave == 0) {  // the evidence's pieces (batch mode): sum_k log L_kk, b'm, sum (y - mean)^2, each a wave butterfly
    double ld = lane < z ? log(A[lane * DLD + lane]) : 0.0;
    double a2 = 0.0;
    for (int i = lane; i < NB; ++i) a2 = __builtin_fma(X[i * DLD + lane], tv[i], a2);   // m_lane, as above
    double qm = lane < z ? bv[lane] * a2 : 0.0;
    double r2 = rr;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ld += __shfl_xor(ld, o);
      qm += __shfl_xor(qm, o);
      r2 += __shfl_xor(r2, o);
    }
    if (lane == 0) {
      double *t3 = terms + 3 * (size_t)blockIdx.x;
      t3[0] = ld, t3[1] = qm, t3[2] = r2;
    }
  }
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
                     (double *)c->Linv.p, (double *)c->alpha.p, (double *)c->resid.p, (int *)c->info.p, report_dev,
                     (const double *)nullptr, 1, (double *)nullptr);
  B7_HIP(c, hipGetLastError());
  c->linv_done = true;
  return B7_OK;
}

// S heads over the same features (b7_blr_eval_nominate_marg): hyp3_dev = [S alpha | S beta | S mean], yraw_dev the raw responses;
// L, Linv [S][64 x 64], m, b [S][64], info [S][4] (+ report in mapped host memory), terms [S][3] (sum log L_kk, b'm, sum r^2).
int launch_blr_heads_small(b7_ctx *c, int S, const double *Z, int N, int z, int ldz, const double *yraw_dev, const double *hyp3_dev,
                           double *L, double *Linv, double *m, double *b, int *info, int *report_dev, double *terms) {
  PhaseScope ps(c, "potrf");
  if (z < 1 || z > NB || c->Npad != NB) return b7_fail(c, B7_ERR_INVALID, "blr_heads_small: z %d, padded %d", z, c->Npad);
  const size_t lds = sizeof(double) * BLR_SMALL_LDS_DOUBLES;
  static bool attr_done[64] = {false};
  if (c->device >= 64 || !attr_done[c->device]) {
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(blr_head_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (c->device < 64) attr_done[c->device] = true;
  }
  hipLaunchKernelGGL(blr_head_small_kernel, dim3(S), dim3(256), lds, c->stream, Z, N, z, ldz, yraw_dev, 0.0, 0.0, L, Linv, m, b, info,
                     report_dev, hyp3_dev, S, terms);
  B7_HIP(c, hipGetLastError());
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
