// GP fit linear algebra: blocked right-looking Cholesky, explicit triangular inverse, alpha = K^-1 (y - m).
//
// Reference arithmetic replaced:
//   utils/math.lua:165      torch.potrf(res, src, 'L')  (LAPACK dpotrf): K = L L'
//   utils/math.lua:168-202  the pcall/jitter loop around it needs to know THAT a pivot failed: the first
//                           non-positive pivot is reported (1-based) like dpotrf's info; the retry schedule
//                           itself runs on the host (api.hip) with the same eps arithmetic
//   alpha = L^-T L^-1 (Y - mean)  -- first half of gp_regressor:predict (call site
//                           scores/expected_improvement.lua:63); the op order inside the absent `gp`
//                           package is unknown, tolerance-checked against oracle/gp.py
//
// Structure (all extents padded to multiples of 64/128 by the caller; padding = identity).  The factorisation is a
// chain of small dependent launches (each costs >= 4 us however little it does), so the schedule keeps TWO launches
// per 64-wide panel on the critical path and hangs everything else on them as extra workgroups ("riders"):
//   diag(p)   workgroup 0 factors the 64x64 diagonal block in LDS and inverts it;
//             riders: the far part of panel p-1's trailing update (block columns >= p+1), and the K-chunk partial
//             products of row block p of inv(L)
//   near(p)   one workgroup per tile row below: L21 = A21 * inv(L11)' and panel p's update of block column p+1
//             only (all the next diag needs); riders: row block p of inv(L) = -inv(L_pp) * (sum of the partials)
// (B7_POTRF_SCHED=0 selects the earlier schedule: panel pairs, separate narrow / K = 128 trailing-update launches.)
// Without the inline inverse (b7_chol, or Npad > 8192) inv(L) comes from recursive doubling over block size
// s = 64,128,...: for each pair [A 0; B C] of already-inverted diagonal blocks, X = -inv(C) * (B * inv(A)); two
// batched MFMA GEMMs per level (launch_trtri).
// The explicit inverse is what lets the posterior variance be one GEMM with a fused column sum of squares
// (posterior.hip) instead of a triangular solve that would have to store L^-1 K*'.
#include "b7_internal.h"
#include "gemm_f64.h"
#include "potrf_diag.h"

#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>
#include <vector>

namespace {

using namespace b7diag;  // NB = 64, DLD, TLD, diag_core, block16_update, DIAG_LDS_BYTES
// 32-deep LDS stages: measured with s_memtime stamps, every stage of these 64x64-tile products carries ~1900 cycles
// of fixed cost (barrier + restage) next to 1024 cycles of MFMA per 16 of depth, so fewer, deeper stages win.
using G64NT = GemmF64<64, 64, 32, 2, 2, false>;
using G64NN = GemmF64<64, 64, 32, 2, 2, true>;
// Whole-K single-stage forms for the launches ON the critical path that have fewer tiles than CUs (panel solve,
// narrow update, the near part of the trailing update): one workgroup per CU has nothing to overlap a staged K loop
// with, so every stage's load latency is exposed (a K = 128 update cost >= 8.2 us however few its tiles); with all of
// K in LDS at once there is ONE load round trip and the MFMAs issue back to back.  Same k order: bit-identical.
using G64NT_K64 = GemmF64<64, 64, 64, 2, 2, false>;
using G64NT_K128 = GemmF64<64, 64, 128, 2, 2, false>;

// One 64x64 tile of a trailing update: A[I][J] -= L[I][kc0 .. kc0+kb) L[J][kc0 .. kc0+kb)', (I, J) the rem-th lower
// tile of block columns [j0, j0 + ncols) counted column by column.  The C tile is fetched before the product so
// its latency hides under the MFMAs.  sm: 2 * G64NT::STAGE_DOUBLES doubles of LDS.
template <bool STAMP, class G = G64NT>
__device__ __forceinline__ void syrk_tile(double *__restrict__ L, int ld, int kc0, int kb, int j0, int ncols, int nbt,
                                          int rem, double *__restrict__ sm, unsigned long long *__restrict__ stamps) {
  int J = j0;
  for (int c = 0; c < ncols; ++c, ++J) {
    const int h = nbt - J;
    if (rem < h) break;
    rem -= h;
  }
  const int I = J + rem;
  const double *a = L + ((int64_t)I * NB) * ld + (int64_t)kc0 * NB;
  const double *b = L + ((int64_t)J * NB) * ld + (int64_t)kc0 * NB;
  double *cblk = L + ((int64_t)I * NB) * ld + (int64_t)J * NB;
  d4_t cin[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cin[i][j][r] = cblk[(int64_t)G::out_row(i, r) * ld + G::out_col(j)];
  d4_t acc[2][2] = {};
  if (STAMP && threadIdx.x == 0) stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
  G::run(a, ld, b, ld, 0, kb * NB, acc, sm);
  if (STAMP && threadIdx.x == 0) stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cblk[(int64_t)G::out_row(i, r) * ld + G::out_col(j)] = cin[i][j][r] - acc[i][j][r];
}

// L <- block-lower part of K (upper 64x64 blocks zeroed).
// `extra` (the jitter of this attempt) is added to the first nreal diagonal entries: src + eps*I.
__global__ void __launch_bounds__(256)
    copy_lower_kernel(const double *__restrict__ K, double *__restrict__ L, int n, int nreal, double extra) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n * n) return;
  int i = (int)(e / n), j = (int)(e - (int64_t)i * n);
  double v = ((j / NB) <= (i / NB)) ? K[e] : 0.0;
  if (i == j && i < nreal) v = v + extra;
  L[e] = v;
}

template <int VAR, bool STAMP = false>
__global__ void __launch_bounds__(256)
    potrf_diag_kernel(double *__restrict__ L, int ld, int p, double *__restrict__ dinv, int *__restrict__ info,
                      unsigned long long *__restrict__ stamps, const double *__restrict__ Linv,
                      double *__restrict__ Wpart, int nt, int s_kc0, int s_kb, int s_j0, int s_ncols, int nbt) {
#define B7_DIAG_STAMP(i) \
  if (STAMP && threadIdx.x == 0) stamps[i] = __builtin_amdgcn_s_memtime()
  B7_DIAG_STAMP(0);
  extern __shared__ __align__(16) double dsm[];
  // workgroup 1 (when there is a tile below): copy A[p+1][p] to the scratch tile behind dinv.  The near kernel's
  // workgroups all read that tile while four of them overwrite it in place with L[p+1][p]; reading the copy made
  // HERE, one launch earlier, takes the race away at no cost to the factorisation.
  const int ncopy = (p + 1 < nbt && gridDim.x > 1) ? 1 : 0;
  if (ncopy && blockIdx.x == 1) {
    const double *below = L + ((int64_t)(p + 1) * NB) * ld + (int64_t)p * NB;
    double *aq = dinv + (int64_t)nbt * NB * NB;
    double2 w[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int e = threadIdx.x + 256 * t, i = e >> 5, j2 = (e & 31) * 2;
      w[t] = *reinterpret_cast<const double2 *>(below + (int64_t)i * ld + j2);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int e = threadIdx.x + 256 * t, i = e >> 5, j2 = (e & 31) * 2;
      *reinterpret_cast<double2 *>(aq + i * NB + j2) = w[t];
    }
    return;
  }
  const int rider = (int)blockIdx.x - 1 - ncopy;  // index among the riders proper
  if (rider >= nt) {  // a deferred trailing-update tile of the previous panel group (see launch_potrf)
    syrk_tile<false>(L, ld, s_kc0, s_kb, s_j0, s_ncols, nbt, rider - nt, dsm, nullptr);
    return;
  }
  if (blockIdx.x > 0) {
    int rem = rider, cch = 0;
    const int PC = (p + 1) / 2;
    for (; cch < PC; ++cch) {
      const int cnt = (2 * cch + 2 < p) ? 2 * cch + 2 : p;  // tiles j whose K range [64 j, 64 p) meets chunk cch
      if (rem < cnt) break;
      rem -= cnt;
    }
    const int j = rem;
    const int k0 = (128 * cch > NB * j) ? 128 * cch : NB * j;
    const int k1 = (128 * cch + 128 < NB * p) ? 128 * cch + 128 : NB * p;
    d4_t acc[2][2] = {};
    G64NN::run(L + ((int64_t)p * NB) * ld, ld, Linv + (int64_t)j * NB, ld, k0, k1, acc, dsm);
    double *out = Wpart + ((int64_t)cch * NB) * ld + (int64_t)j * NB;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(int64_t)G64NN::out_row(i, r) * ld + G64NN::out_col(jj)] = acc[i][jj][r];
    return;
  }
  double *A = dsm;                  // [64][DLD] working copy of the block, becomes L11
  double *X = dsm + NB * DLD;       // [64][DLD] inverse of L11 (zero above the diagonal)
  double *T = X + NB * DLD;         // [32][TLD] scratch of the last doubling level
  const int tid = threadIdx.x;
  double *blk = L + ((int64_t)p * NB) * ld + (int64_t)p * NB;
  {  // all eight 16-byte loads of a thread in flight at once (a load -> wait -> LDS store loop is 16 round trips)
    double2 v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int e = tid + 256 * t, i = e >> 5, j2 = (e & 31) * 2;
      v[t] = *reinterpret_cast<const double2 *>(blk + (int64_t)i * ld + j2);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int e = tid + 256 * t, i = e >> 5, j2 = (e & 31) * 2;
      // VAR 1: I_16 for the inversion recurrence lives in the block's unused upper-right 16x16 corner (rows 0..15,
      // columns 48..63: only blocks on or below the diagonal are ever touched); its neighbour (columns 32..47)
      // takes dummy stores
      if (VAR >= 1 && i < 16 && j2 >= 48) v[t] = make_double2(i == j2 - 48 ? 1.0 : 0.0, i == j2 - 47 ? 1.0 : 0.0);
      *reinterpret_cast<double2 *>(A + i * DLD + j2) = v[t];
      *reinterpret_cast<double2 *>(X + i * DLD + j2) = make_double2(0.0, 0.0);
    }
  }
  __syncthreads();
  B7_DIAG_STAMP(1);

  diag_core<VAR, STAMP>(A, X, T, p, info, stamps);
  B7_DIAG_STAMP(18);
  double *dv = dinv + (int64_t)p * NB * NB;
  {  // all LDS reads first, then 16-byte stores (same reason as the load above)
    double2 va[8], vx[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int e = tid + 256 * t, i = e >> 5, j2 = (e & 31) * 2;
      va[t] = *reinterpret_cast<const double2 *>(A + i * DLD + j2);
      vx[t] = *reinterpret_cast<const double2 *>(X + i * DLD + j2);
      if (j2 > i) va[t].x = 0.0;
      if (j2 + 1 > i) va[t].y = 0.0;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int e = tid + 256 * t, i = e >> 5, j2 = (e & 31) * 2;
      *reinterpret_cast<double2 *>(blk + (int64_t)i * ld + j2) = va[t];
      *reinterpret_cast<double2 *>(dv + i * NB + j2) = vx[t];
    }
  }
  B7_DIAG_STAMP(19);
#undef B7_DIAG_STAMP
}
// Riding workgroups (inverse partial products, deferred trailing-update tiles) may share the factorising CU: two
// workgroups of this size fit, and keeping that CU exclusive (a larger LDS request) measured 2 % slower at
// N = 2048 because the riders then need a second round of CUs.

// L21 tile <- A21 tile * inv(L11)'   (in place; the tile is fully read before it is written): workgroups [0, ntrsm).
// Workgroups ntrsm + 4 j + slab, j in [0, nbt) (present when the inverse is built alongside) write tile j of row
// block p of inv(L), a 16-column slab each:  j > p zero, j == p inv(L_pp), j < p  -inv(L_pp) * (sum over the K chunks of the partial products that
// potrf_diag_kernel's extra workgroups left in W, added in ascending chunk order: a fixed summation order).
// NEAR (the one-panel-at-a-time schedule, see launch_potrf): workgroup b < ntrsm also applies panel p's update to
// its tile of the NEXT block column, A[I][p+1] -= L[I][p] L[p+1][p]', which is all the next diagonal block and the
// next panel solve need; it forms L[p+1][p] itself from the inverse it has staged anyway (redundant across the
// workgroups, but a launch of its own would cost more than the 40 MFMAs per wave).
constexpr int TRSM_LDS_BYTES = 2 * G64NT::STAGE_DOUBLES * 8;
constexpr int NEAR_LDS_BYTES = (16 + 2 * NB) * DLD * 8;  // a 16-row slab, A[p+1][p] and inv(L_pp); >= the inverse-row path's 2 tiles
template <bool NEAR>
__global__ void __launch_bounds__(256)
    potrf_trsm_kernel(double *__restrict__ L, int ld, int p, const double *__restrict__ dinv, int ntrsm, int nbt,
                      double *__restrict__ Linv, const double *__restrict__ Wpart) {
  extern __shared__ __align__(16) double sm[];
  static_assert(2 * G64NT::STAGE_DOUBLES >= 2 * NB * DLD, "the inverse-row path reuses the GEMM staging area");
  if ((int)blockIdx.x >= ntrsm) {
    // one workgroup per 64 x 16 column slab of a tile: 2 x 16-byte loads per thread and chunk, so the partials of
    // 16 chunks are in flight at once and the whole sum costs one memory round trip (a launch this short cannot
    // afford one round trip per few chunks)
    const int xb = blockIdx.x - ntrsm, j = xb >> 2, slab = xb & 3, tid = threadIdx.x;
    double *out = Linv + ((int64_t)p * NB) * ld + (int64_t)j * NB + slab * 16;
    const double *dv = dinv + (int64_t)p * NB * NB;
    const int ri = tid >> 3, rc = (tid & 7) * 2;  // this thread's two double2 of a slab: rows ri and ri + 32
    if (j >= p) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        double2 v = make_double2(0.0, 0.0);
        if (j == p) v = *reinterpret_cast<const double2 *>(dv + (ri + 32 * h) * NB + slab * 16 + rc);
        *reinterpret_cast<double2 *>(out + (int64_t)(ri + 32 * h) * ld + rc) = v;
      }
      return;
    }
    constexpr int SLD = 17;  // odd stride: the B-operand reads (k = lane >> 4, n = lane & 15) stay conflict-free
    double *Dn = sm, *Ts = sm + NB * DLD;
    const int PC = (p + 1) / 2;
    {
      double2 dn[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int e = tid + 256 * t, i = e >> 5, j2 = (e & 31) * 2;
        dn[t] = *reinterpret_cast<const double2 *>(dv + i * NB + j2);
      }
      double2 acc[2] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
      const double *src0 = Wpart + (int64_t)j * NB + slab * 16 + rc;
      for (int c0 = j / 2; c0 < PC; c0 += 16) {
        double2 v[16][2];
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            v[u][h] = (c0 + u < PC) ? *reinterpret_cast<const double2 *>(
                                          src0 + ((int64_t)(c0 + u) * NB + ri + 32 * h) * ld)
                                    : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < 16; ++u)  // ascending chunk order
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            acc[h].x += v[u][h].x;
            acc[h].y += v[u][h].y;
          }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        Ts[(ri + 32 * h) * SLD + rc] = acc[h].x;
        Ts[(ri + 32 * h) * SLD + rc + 1] = acc[h].y;
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int e = tid + 256 * t, i = e >> 5, j2 = (e & 31) * 2;
        *reinterpret_cast<double2 *>(Dn + i * DLD + j2) = dn[t];
      }
    }
    __syncthreads();
    // out[i][n] = -sum_{k <= i} Dn[i][k] Ts[k][n]: wave w owns rows 16w .. 16w+15; two accumulators alternate
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
    d4_t c0a = {0.0, 0.0, 0.0, 0.0}, c1a = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      if (kq > wave) break;
#pragma unroll
      for (int s4 = 0; s4 < 4; s4 += 2) {
        c0a = mfma_f64(-Dn[(wave * 16 + lr) * DLD + kq * 16 + 4 * s4 + lq], Ts[(kq * 16 + 4 * s4 + lq) * SLD + lr], c0a);
        c1a = mfma_f64(-Dn[(wave * 16 + lr) * DLD + kq * 16 + 4 * s4 + 4 + lq],
                       Ts[(kq * 16 + 4 * s4 + 4 + lq) * SLD + lr], c1a);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) out[(int64_t)(wave * 16 + lq + 4 * rr) * ld + lr] = c0a[rr] + c1a[rr];
    return;
  }
  static_assert(2 * G64NT::STAGE_DOUBLES >= G64NT_K64::STAGE_DOUBLES, "one whole-K stage fits the staging area");
  double *tile = L + ((int64_t)(p + 1 + blockIdx.x) * NB) * ld + (int64_t)p * NB;
  if (NEAR) {
    // One workgroup per 16-row slab of a tile (4 per tile): a tile's three products on ONE CU are 144 dependent-rate
    // MFMAs per wave (3.8 us of a 9 us launch); as slabs it is 40 (L[p+1][p], still redundant) + <= 16 + 16, and a
    // block column still has fewer workgroups than the chip has CUs.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int slab = blockIdx.x & 3, r0 = slab * 16;
    tile = L + ((int64_t)(p + 1 + (blockIdx.x >> 2)) * NB) * ld + (int64_t)p * NB;
    double *Ai = sm, *Aq = sm + 16 * DLD, *Xp = sm + (16 + NB) * DLD;
    const double *tq = dinv + (int64_t)nbt * NB * NB;  // A[p+1][p]: the copy the diag launch left behind dinv
    const double *dv = dinv + (int64_t)p * NB * NB;
    double *ctile = tile + NB;  // A[I][p+1]
    {
      double2 va[2], vq[8], vx[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int e = tid + 256 * t, i = e >> 5, j2 = (e & 31) * 2;
        if (t < 2) va[t] = *reinterpret_cast<const double2 *>(tile + (int64_t)(r0 + i) * ld + j2);
        vq[t] = *reinterpret_cast<const double2 *>(tq + i * NB + j2);
        vx[t] = *reinterpret_cast<const double2 *>(dv + i * NB + j2);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int e = tid + 256 * t, i = e >> 5, j2 = (e & 31) * 2;
        if (t < 2) *reinterpret_cast<double2 *>(Ai + i * DLD + j2) = va[t];
        *reinterpret_cast<double2 *>(Aq + i * DLD + j2) = vq[t];
        *reinterpret_cast<double2 *>(Xp + i * DLD + j2) = vx[t];
      }
    }
    d4_t cin;  // wave w owns column block w of the slab's C rows; in flight under the solves
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) cin[rr] = ctile[(int64_t)(r0 + lq + 4 * rr) * ld + wave * 16 + lr];
    __syncthreads();
    // solves: out[a][j] = sum_{k <= j} A[a][k] X[j][k]   (X lower triangular: whole 16-blocks above skipped).
    // L[p+1][p]: wave w rows 16w.., all four column blocks.  The slab of L[I][p]: wave w column block w.
    d4_t lqv[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    d4_t li = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kq = 0; kq < 4; ++kq)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int ko = kq * 16 + 4 * s4 + lq;
        const double aq = Aq[(wave * 16 + lr) * DLD + ko];
#pragma unroll
        for (int jb = kq; jb < 4; ++jb) lqv[jb] = mfma_f64(aq, Xp[(jb * 16 + lr) * DLD + ko], lqv[jb]);
        if (kq <= wave) li = mfma_f64(Ai[lr * DLD + ko], Xp[(wave * 16 + lr) * DLD + ko], li);
      }
    __syncthreads();  // every wave is done reading Ai / Aq
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      Ai[(lq + 4 * rr) * DLD + wave * 16 + lr] = li[rr];
      tile[(int64_t)(r0 + lq + 4 * rr) * ld + wave * 16 + lr] = li[rr];  // L[I][p]
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) Aq[(wave * 16 + lq + 4 * rr) * DLD + jb * 16 + lr] = lqv[jb][rr];
    }
    __syncthreads();
    d4_t u = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4)
      u = mfma_f64(Ai[lr * DLD + 4 * k4 + lq], Aq[(wave * 16 + lr) * DLD + 4 * k4 + lq], u);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) ctile[(int64_t)(r0 + lq + 4 * rr) * ld + wave * 16 + lr] = cin[rr] - u[rr];
    return;
  }
  d4_t acc[2][2] = {};
  G64NT_K64::run(tile, ld, dinv + (int64_t)p * NB * NB, NB, 0, NB, acc, sm);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[(int64_t)G64NT_K64::out_row(i, r) * ld + G64NT_K64::out_col(j)] = acc[i][j][r];
}

// Trailing update: A[I][J] -= L[I][kc0 .. kc0+kb) L[J][kc0 .. kc0+kb)'   for block columns J in [j0, j0+ncols) and
// I >= J (lower tiles only; 1-D grid over exactly those tiles).  kb = 1 updates with one 64-wide panel, kb = 2
// with two at once (K = 128): the big update is issued once per PAIR of panels, which doubles its work per launch
// at the same tile count.  The C tile is fetched before the product so its latency hides under the MFMAs.
template <bool STAMP>
__global__ void __launch_bounds__(256, 2)
    potrf_syrk_kernel(double *__restrict__ L, int ld, int kc0, int kb, int j0, int ncols, int nbt,
                      unsigned long long *__restrict__ stamps) {
  if (STAMP && threadIdx.x == 0) stamps[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
  __shared__ __align__(16) double sm[2 * G64NT::STAGE_DOUBLES];
  syrk_tile<STAMP>(L, ld, kc0, kb, j0, ncols, nbt, blockIdx.x, sm, stamps);
  if (STAMP && threadIdx.x == 0) stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime();
}

// The same update for launches with fewer tiles than CUs: all of K = 64 KB staged at once (dynamic LDS).
template <int KB>
__global__ void __launch_bounds__(256)
    potrf_syrk_small_kernel(double *__restrict__ L, int ld, int kc0, int j0, int ncols, int nbt) {
  extern __shared__ __align__(16) double dsm_small[];
  using G = typename std::conditional<KB == 1, G64NT_K64, G64NT_K128>::type;
  syrk_tile<false, G>(L, ld, kc0, KB, j0, ncols, nbt, blockIdx.x, dsm_small, nullptr);
}

// Linv <- blockdiag(dinv), zero elsewhere.
__global__ void __launch_bounds__(256)
    linv_init_kernel(double *__restrict__ Linv, const double *__restrict__ dinv, int n) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n * n) return;
  int i = (int)(e / n), j = (int)(e - (int64_t)i * n);
  Linv[e] = ((i / NB) == (j / NB)) ? dinv[(int64_t)(i / NB) * NB * NB + (i % NB) * NB + (j % NB)] : 0.0;
}

// Triangular-inverse level, step 1:  W[C, A] = L[C, A] * Linv[A, A]      (Linv[A,A] lower: k >= j)
// grid (s/64, s/64, pairs); pair t covers A = [2ts, 2ts+s), C = [2ts+s, min(2ts+2s, n)).
__global__ void __launch_bounds__(256)
    trtri_step1_kernel(const double *__restrict__ L, const double *__restrict__ Linv, double *__restrict__ W, int n,
                       int s) {
  const int a0 = 2 * blockIdx.z * s, c0 = a0 + s;
  const int crow = c0 + blockIdx.y * NB;
  if (crow >= n) return;
  const int tj = blockIdx.x;
  __shared__ __align__(16) double sm[2 * G64NN::STAGE_DOUBLES];
  d4_t acc[2][2] = {};
  G64NN::run(L + (int64_t)crow * n + a0, n, Linv + (int64_t)a0 * n + a0 + tj * NB, n, tj * NB, s, acc, sm);
  double *out = W + (int64_t)crow * n + a0 + tj * NB;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(int64_t)G64NN::out_row(i, r) * n + G64NN::out_col(j)] = acc[i][j][r];
}

// Step 2:  Linv[C, A] = -Linv[C, C] * W[C, A]      (Linv[C,C] lower: k <= i)
__global__ void __launch_bounds__(256)
    trtri_step2_kernel(double *__restrict__ Linv, const double *__restrict__ W, int n, int s) {
  const int a0 = 2 * blockIdx.z * s, c0 = a0 + s;
  const int ti = blockIdx.y;
  const int crow = c0 + ti * NB;
  if (crow >= n) return;
  const int tj = blockIdx.x;
  __shared__ __align__(16) double sm[2 * G64NN::STAGE_DOUBLES];
  d4_t acc[2][2] = {};
  G64NN::run(Linv + (int64_t)crow * n + c0, n, W + (int64_t)c0 * n + a0 + tj * NB, n, 0, (ti + 1) * NB, acc, sm);
  double *out = Linv + (int64_t)crow * n + a0 + tj * NB;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(int64_t)G64NN::out_row(i, r) * n + G64NN::out_col(j)] = -acc[i][j][r];
}

// t = Linv * r  (lower triangular matrix-vector, one wave per row, ycols right-hand sides)
__global__ void __launch_bounds__(256)
    trmv_lower_kernel(const double *__restrict__ Linv, const double *__restrict__ r, double *__restrict__ t, int n,
                      int ycols, int64_t sLinv = 0, int64_t sr = 0, int64_t st = 0) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  Linv += blockIdx.z * sLinv;  // a batch of independent systems: grid.z = fit
  r += blockIdx.z * sr;
  t += blockIdx.z * st;
  for (int cidx = 0; cidx < ycols; ++cidx) {
    double s = 0.0;
    // eight loads of each operand in flight per lane; the sum runs in the same order as the plain loop (same bits)
    for (int k0 = lane; k0 <= row; k0 += 64 * 8) {
      double av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + 64 * u;
        av[u] = k <= row ? Linv[(int64_t)row * n + k] : 0.0;
        bv[u] = k <= row ? r[(int64_t)k * ycols + cidx] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + 64 * u <= row) s += av[u] * bv[u];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) t[(int64_t)row * ycols + cidx] = s;
  }
}

// e = |Linv| * |r|: the magnitude sum behind each entry of Linv * r, i.e. (up to the unit roundoff and a term count)
// the rounding-error bound of that entry.  b7_gp_append uses it to decide whether the new pivot is distinguishable
// from zero at working precision.
__global__ void __launch_bounds__(256)
    trmv_lower_abs_kernel(const double *__restrict__ Linv, const double *__restrict__ r, double *__restrict__ e,
                          int n, int nrows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  double s = 0.0;
  if (row < nrows)
    for (int k = lane; k <= row; k += 64) s += fabs(Linv[(int64_t)row * n + k]) * fabs(r[k]);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) e[row] = s;
}

// alpha = Linv' * t in two passes with a fixed summation order: partial[ib][col] over 256-row slices, then their
// sum in ascending ib.  Block = 64 columns x one 256-row slice (4 waves x 64 rows).
constexpr int TSL = 256;
__global__ void __launch_bounds__(256)
    trmv_lower_t_part_kernel(const double *__restrict__ Linv, const double *__restrict__ t,
                             double *__restrict__ part, int n, int ycols, int64_t sLinv = 0, int64_t st = 0) {
  __shared__ double red[4][64];
  Linv += blockIdx.z * sLinv;
  t += blockIdx.z * st;
  part += blockIdx.z * st;  // t and its slice partials share one block per fit: [n ycols | nslices ycols n]
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
  const int r0 = blockIdx.y * TSL + wave * 64;
  for (int cidx = 0; cidx < ycols; ++cidx) {
    double s = 0.0;
    if (r0 < n && r0 + 63 >= blockIdx.x * 64) {  // Linv[i][col] = 0 for i < col: slices above the diagonal are zero
      for (int i0 = r0; i0 < r0 + 64; i0 += 16) {  // sixteen rows in flight; summed in row order (same bits as a plain loop)
        double av[16], tv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          av[u] = Linv[(int64_t)(i0 + u) * n + col];
          tv[u] = t[(int64_t)(i0 + u) * ycols + cidx];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) s += av[u] * tv[u];
      }
    }
    red[wave][threadIdx.x & 63] = s;
    __syncthreads();
    if (wave == 0)
      part[((int64_t)blockIdx.y * ycols + cidx) * n + col] =
          ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256)
    trmv_lower_t_sum_kernel(const double *__restrict__ part, double *__restrict__ alpha, int n, int nreal, int ycols,
                            int yld, int nslices, int64_t st = 0, int64_t salpha = 0, const int *__restrict__ rep_src = nullptr,
                            int *__restrict__ rep_dst = nullptr, int rep_words = 0) {
  // the last kernel of a fit hands the factorisation's pivot reports to mapped host memory on its way (the factorisation
  // before it on the stream is complete): a 16-byte copy launch per fit less
  if (rep_dst && blockIdx.x == 0 && blockIdx.z == 0 && (int)threadIdx.x < rep_words) rep_dst[threadIdx.x] = rep_src[threadIdx.x];
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= n) return;
  part += blockIdx.z * st;
  alpha += blockIdx.z * salpha;
  for (int cidx = 0; cidx < ycols; ++cidx) {
    double s = 0.0;
    for (int ib = col / TSL; ib < nslices; ++ib) s += part[((int64_t)ib * ycols + cidx) * n + col];
    alpha[(int64_t)col * yld + cidx] = (col < nreal) ? s : 0.0;
  }
  for (int cidx = ycols; cidx < yld; ++cidx) alpha[(int64_t)col * yld + cidx] = 0.0;
}

}  // namespace

// The 64x64 factor-and-invert kernel on a stand-alone 64x64 matrix (fantasize's pending-point covariance).
int launch_fantasy_factor(b7_ctx *c, double *S, double *dinv_tmp, int *info_dev) {
  B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_diag_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
  B7_HIP(c, hipMemsetAsync(info_dev, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(potrf_diag_kernel<1>, dim3(1), dim3(256), DIAG_LDS_BYTES, c->stream, S, NB, 0, dinv_tmp,
                     info_dev, (unsigned long long *)nullptr, (const double *)nullptr, (double *)nullptr, 0, 0, 0, 0, 0,
                     1);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

// with_inverse: inv(L) is built row block by row block inside the factorisation's own launches (extra workgroups
// of the diagonal-block and panel-solve kernels, see there), so it costs no launches and almost no time of its
// own -- the factorisation is a chain of small dependent kernels that leaves most CUs idle.  Needs c->Linv and
// c->W (n x n each).  Without it (b7_chol) only L and dinv are produced and launch_trtri is the way to inv(L).
int launch_potrf(b7_ctx *c, double extra, bool with_inverse, int *report_hint, FactorNote *note) {
  if (note) *note = FactorNote();
  if (c->Npad == 64 && c->potrf_small && with_inverse && c->potrf_sched == 3 && c->inverse_inline) {
    // one 64-block: factorisation, inverse, pivot report and (one response column) alpha in one workgroup of one launch
    B7_TRY(b7_ensure(c, c->info, B7_INFO_BYTES));
    const bool one_col = c->ycols == 1 && c->yld == 1;
    B7_TRY(launch_potrf_small(c, 1, (const double *)c->K.p, (double *)c->L.p, (double *)c->Linv.p, (double *)c->dinv.p,
                              one_col ? (const double *)c->resid.p : nullptr, (double *)c->alpha.p, extra, (int *)c->info.p,
                              report_hint, 0, 0, 0, 0, 4));
    c->linv_done = true;
    if (note) note->alpha_done = one_col, note->report_written = report_hint;
    else if (one_col) { /* alpha was written as well; a caller that does not ask recomputes it: same bits */ }
    return B7_OK;
  }
  if (c->potrf_sched == 3 && c->Npad <= B7_PERSIST_NMAX && c->inverse_inline) return launch_potrf_persist(c, extra, with_inverse);
  PhaseScope ps(c, "potrf");
  const int n = c->Npad, nb = n / NB;
  double *L = (double *)c->L.p;
  // Once a row block has many more partial products than the CUs can take next to the factorisation the gain
  // fades: measured inline vs separate passes 0.84 / 1.05 ms at N = 2048, 2.5 / 2.9 at 4096, 12.2 / 12.4 at 8192.
  with_inverse = with_inverse && c->inverse_inline && (n <= 8192 || c->inverse_inline > 1);
  c->linv_done = false;
  const double *Linv = with_inverse ? (const double *)c->Linv.p : nullptr;
  double *Wp = with_inverse ? (double *)c->W.p : nullptr;
  int64_t total = (int64_t)n * n;
  hipLaunchKernelGGL(copy_lower_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                     (const double *)c->K.p, L, n, c->N, extra);
  B7_HIP(c, hipMemsetAsync(c->info.p, 0, 4 * sizeof(int), c->stream));
  if (!c->potrf_attrs_set) {  // once per context: every one of these is a host API call
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_diag_kernel<0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_diag_kernel<1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_diag_kernel<2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_syrk_small_kernel<1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G64NT_K64::STAGE_DOUBLES * 8));
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_syrk_small_kernel<2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G64NT_K128::STAGE_DOUBLES * 8));
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_trsm_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, TRSM_LDS_BYTES));
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_trsm_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, NEAR_LDS_BYTES));
    c->potrf_attrs_set = true;
  }
  // panels in pairs (a, b = a + 1): the narrow update of block column b after panel a, then ONE trailing update
  // with both panels (K = 128) for everything to the right of b.
  // deferred part of the previous group's trailing update: (kc0, kb, j0, ncols), taken by the next diag launch
  struct Deferred { int kc0, kb, j0, ncols, tiles; } pend = {0, 0, 0, 0, 0};
  auto diag = [&](int p) {
    // extra workgroups: the partial products of row block p of inv(L), sum over chunks c of min(p, 2c + 2) tiles
    int nt = 0;
    if (with_inverse)
      for (int cc = 0; cc < (p + 1) / 2; ++cc) nt += (2 * cc + 2 < p) ? 2 * cc + 2 : p;
    const int ns = pend.tiles;
    const dim3 grid(1 + (p + 1 < nb ? 1 : 0) + nt + ns);  // factor, copier of the tile below, riders
    if (c->diag_variant == 0)
      hipLaunchKernelGGL(potrf_diag_kernel<0>, grid, dim3(256), DIAG_LDS_BYTES, c->stream, L, n, p,
                         (double *)c->dinv.p, (int *)c->info.p, (unsigned long long *)nullptr, Linv, Wp, nt,
                         pend.kc0, pend.kb, pend.j0, pend.ncols, nb);
    else if (c->diag_variant == 2)
      hipLaunchKernelGGL(potrf_diag_kernel<2>, grid, dim3(256), DIAG_LDS_BYTES, c->stream, L, n, p,
                         (double *)c->dinv.p, (int *)c->info.p, (unsigned long long *)nullptr, Linv, Wp, nt,
                         pend.kc0, pend.kb, pend.j0, pend.ncols, nb);
    else
      hipLaunchKernelGGL(potrf_diag_kernel<1>, grid, dim3(256), DIAG_LDS_BYTES, c->stream, L, n, p,
                         (double *)c->dinv.p, (int *)c->info.p, (unsigned long long *)nullptr, Linv, Wp, nt,
                         pend.kc0, pend.kb, pend.j0, pend.ncols, nb);
    pend.tiles = 0;
  };
  auto trsm = [&](int p, bool near) {
    const int ntrsm = nb - p - 1, nx = with_inverse ? 4 * nb : 0;  // 4 column slabs per tile of the inverse row
    if (ntrsm + nx <= 0) return;
    if (near)  // four 16-row slabs per tile
      hipLaunchKernelGGL(potrf_trsm_kernel<true>, dim3(4 * ntrsm + nx), dim3(256), NEAR_LDS_BYTES, c->stream, L, n, p,
                         (const double *)c->dinv.p, 4 * ntrsm, nb, (double *)c->Linv.p, (const double *)Wp);
    else
      hipLaunchKernelGGL(potrf_trsm_kernel<false>, dim3(ntrsm + nx), dim3(256), TRSM_LDS_BYTES, c->stream, L, n, p,
                         (const double *)c->dinv.p, ntrsm, nb, (double *)c->Linv.p, (const double *)Wp);
  };
  auto syrk = [&](int kc0, int kb, int j0, int ncols) {
    int tiles = 0;
    for (int J = j0; J < j0 + ncols; ++J) tiles += nb - J;
    if (tiles <= 0) return;
    if (tiles <= 256 && kb == 1 && c->syrk_small)
      hipLaunchKernelGGL(potrf_syrk_small_kernel<1>, dim3(tiles), dim3(256), G64NT_K64::STAGE_DOUBLES * 8, c->stream, L,
                         n, kc0, j0, ncols, nb);
    else if (tiles <= 256 && kb == 2 && c->syrk_small)
      hipLaunchKernelGGL(potrf_syrk_small_kernel<2>, dim3(tiles), dim3(256), G64NT_K128::STAGE_DOUBLES * 8, c->stream,
                         L, n, kc0, j0, ncols, nb);
    else
      hipLaunchKernelGGL(potrf_syrk_kernel<false>, dim3(tiles), dim3(256), 0, c->stream, L, n, kc0, kb, j0, ncols, nb,
                         (unsigned long long *)nullptr);
  };
  // Panels in groups of G: inside a group every new panel first receives the updates of the group's earlier
  // panels on its own block column only (narrow, K = 64 q); the bulk of the trailing matrix is updated ONCE per
  // group with K = 64 G.  Larger G raises the bulk updates' MFMA utilisation but makes the narrow updates deep
  // and serial: measured potrf at N = 2048 is 1.07 ms for G = 1 and 2, 1.12 for 4, 1.27 for 8 (tools/potrf_ab.py).
  // measured pair schedule vs this one: 0.386 / 0.347 ms at N = 1024, 0.83 / 0.76 at 2048, 2.48 / 2.33 at 4096,
  // 12.2 / 12.6 at 8192 (K = 64 rider tiles by the thousand): one panel at a time up to Npad = 4096
  if (((c->potrf_sched == 1 || c->potrf_sched == 3) && n <= 4096) || c->potrf_sched == 2) {
    // One panel at a time, two launches each, the rest riding along:
    //   diag(p)   + riders: partial products of inverse row p, and the FAR part of panel p-1's update (block columns
    //               >= p+1 ... i.e. everything but the column its own near part already did), K = 64 per tile
    //   near(p)   panel solve of every tile row below + panel p's update of block column p+1 only (+ inverse row p)
    // The critical path is diag -> near -> diag ...; each tile still receives the panels' updates in ascending
    // order, one panel (K = 64) at a time.
    for (int p = 0; p < nb; ++p) {
      diag(p);
      trsm(p, true);
      if (p + 2 < nb) {
        pend = {p, 1, p + 2, nb - (p + 2), 0};
        for (int J = p + 2; J < nb; ++J) pend.tiles += nb - J;
      }
    }
    B7_HIP(c, hipGetLastError());
    c->linv_done = with_inverse;
    return B7_OK;
  }
  const int G = c->potrf_group;
  for (int a = 0; a < nb; a += G) {
    const int gsz = (nb - a < G) ? nb - a : G;
    for (int q = 0; q < gsz; ++q) {
      const int p = a + q;
      if (q > 0) syrk(a, q, p, 1);  // block column p <- panels a .. p-1
      diag(p);
      trsm(p, false);
    }
    // The group's trailing update, K = 64 * gsz, in two parts: the next group's own block columns now, everything
    // beyond them as extra workgroups of the next group's first diagonal-block launch (nothing reads those tiles
    // before that group's own trailing update, and that launch keeps one CU busy for 13 us).  Same products in the
    // same order on every tile: bit-identical to the one-launch form (B7_POTRF_DEFER=0).
    const int next = a + gsz;
    if (next < nb) {
      const int near = c->potrf_defer ? ((nb - next < G) ? nb - next : G) : nb - next;
      syrk(a, gsz, next, near);
      const int far0 = next + near;
      if (far0 < nb) {
        pend = {a, gsz, far0, nb - far0, 0};
        for (int J = far0; J < nb; ++J) pend.tiles += nb - J;
      }
    }
  }
  B7_HIP(c, hipGetLastError());
  c->linv_done = with_inverse;
  return B7_OK;
}

int launch_trtri(b7_ctx *c) {
  if (c->linv_done) return B7_OK;  // already built inside launch_potrf
  PhaseScope ps(c, "trtri");
  const int n = c->Npad;
  int64_t total = (int64_t)n * n;
  double *Linv = (double *)c->Linv.p;
  hipLaunchKernelGGL(linv_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, Linv,
                     (const double *)c->dinv.p, n);
  for (int s = NB; s < n; s *= 2) {
    const int pairs = (n + 2 * s - 1) / (2 * s);
    dim3 grid(s / NB, s / NB, pairs);
    hipLaunchKernelGGL(trtri_step1_kernel, grid, dim3(256), 0, c->stream, (const double *)c->L.p,
                       (const double *)Linv, (double *)c->W.p, n, s);
    hipLaunchKernelGGL(trtri_step2_kernel, grid, dim3(256), 0, c->stream, Linv, (const double *)c->W.p, n, s);
  }
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

// l = Linv[0:nrows, 0:nrows] * k  and  u = Linv[0:nrows, 0:nrows]' * l   (the two mat-vecs of b7_gp_append;
// rows >= nrows of the outputs are zeroed so that the padding row being replaced does not leak in)
__global__ void zero_tail_kernel(double *__restrict__ v, int from, int n) {
  const int i = from + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = 0.0;
}
int launch_append_vectors(b7_ctx *c, const double *krow, double *lvec, double *uvec, double *part, double *evec) {
  const int n = c->Npad, nrows = c->N;
  hipLaunchKernelGGL(trmv_lower_kernel, dim3(n / 4), dim3(256), 0, c->stream, (const double *)c->Linv.p, krow, lvec, n,
                     1);
  hipLaunchKernelGGL(trmv_lower_abs_kernel, dim3(n / 4), dim3(256), 0, c->stream, (const double *)c->Linv.p, krow,
                     evec, n, nrows);
  hipLaunchKernelGGL(zero_tail_kernel, dim3((n - nrows + 255) / 256), dim3(256), 0, c->stream, lvec, nrows, n);
  const int nslices = (n + TSL - 1) / TSL;
  hipLaunchKernelGGL(trmv_lower_t_part_kernel, dim3(n / 64, nslices), dim3(256), 0, c->stream,
                     (const double *)c->Linv.p, (const double *)lvec, part, n, 1);
  hipLaunchKernelGGL(trmv_lower_t_sum_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const double *)part,
                     uvec, n, nrows, 1, 1, nslices);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

// out[0] = sum_i log L_ii, out[1 + k] = r_k' alpha_k: the two reductions of the negative log marginal likelihood, so
// that a fit hands the host (info, nll terms) in ONE small copy instead of a strided diagonal and alpha.
namespace {
__global__ void __launch_bounds__(256)
    nll_terms_kernel(const double *__restrict__ L, const double *__restrict__ resid, const double *__restrict__ alpha,
                     int N, int ld, int ycols, int yld, double *__restrict__ out) {
  __shared__ double red[256];
  const int t = threadIdx.x;
  double s = 0.0;
  for (int i = t; i < N; i += 256) s += log(L[(int64_t)i * (ld + 1)]);
  red[t] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  if (t == 0) out[0] = red[0];
  if (ycols == 1) {
    __syncthreads();
    double q = 0.0;
    for (int i = t; i < N; i += 256) q += resid[i] * alpha[(int64_t)i * yld];
    red[t] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (t < o) red[t] += red[t + o];
      __syncthreads();
    }
    if (t == 0) out[1] = red[0];
  } else {
    for (int k = t; k < ycols; k += 256) {  // one column per thread, rows in order
      double q = 0.0;
      for (int i = 0; i < N; ++i) q += resid[(int64_t)i * ycols + k] * alpha[(int64_t)i * yld + k];
      out[1 + k] = q;
    }
  }
}
}  // namespace

int launch_nll_terms(b7_ctx *c, double *out_dev) {
  hipLaunchKernelGGL(nll_terms_kernel, dim3(1), dim3(256), 0, c->stream, (const double *)c->L.p,
                     (const double *)c->resid.p, (const double *)c->alpha.p, c->N, c->Npad, c->ycols, c->yld, out_dev);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

namespace {
__global__ void copy_report_kernel(const int *__restrict__ src, int *__restrict__ dst, int words) {
  if ((int)threadIdx.x < words) dst[threadIdx.x] = src[threadIdx.x];
}
}  // namespace

int launch_alpha(b7_ctx *c, int *report_dev, int report_words, const FactorNote &note) {
  if (note.alpha_done) {  // the one-block factorisation produced alpha already (launch_potrf); only the report may be owed
    if (report_dev && report_dev != note.report_written) {
      hipLaunchKernelGGL(copy_report_kernel, dim3(1), dim3(64), 0, c->stream, (const int *)c->info.p, report_dev, report_words);
      B7_HIP(c, hipGetLastError());
    }
    return B7_OK;
  }
  const int n = c->Npad;
  const int nslices = (n + TSL - 1) / TSL;
  // t = Linv * resid (n x ycols) and the nslices x ycols x n slice partials of Linv' * t live in a buffer of their
  // own: W (n x n) is too small for them whenever ycols * (1 + nslices) > n (e.g. 100 fantasy columns at n = 128).
  B7_TRY(b7_ensure(c, c->atmp, sizeof(double) * (size_t)n * c->ycols * (size_t)(1 + nslices)));
  PhaseScope ps(c, "alpha");
  double *t = (double *)c->atmp.p;
  hipLaunchKernelGGL(trmv_lower_kernel, dim3(n / 4), dim3(256), 0, c->stream, (const double *)c->Linv.p,
                     (const double *)c->resid.p, t, n, c->ycols);
  double *part = t + (size_t)n * c->ycols;
  hipLaunchKernelGGL(trmv_lower_t_part_kernel, dim3(n / 64, nslices), dim3(256), 0, c->stream,
                     (const double *)c->Linv.p, (const double *)t, part, n, c->ycols);
  hipLaunchKernelGGL(trmv_lower_t_sum_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const double *)part,
                     (double *)c->alpha.p, n, c->N, c->ycols, c->yld, nslices, (int64_t)0, (int64_t)0, (const int *)c->info.p,
                     report_dev, report_dev ? report_words : 0);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

// alpha_b = Linv_b' (Linv_b r_b) for B single-column fits at once (grid.z = fit); the same kernels, the same sums
int launch_alpha_batch(b7_ctx *c, int B, const double *Linv, const double *resid, double *alpha, const int *report_src,
                       int *report_dev, int report_words) {
  const int n = c->Npad;
  const int nslices = (n + TSL - 1) / TSL;
  const int64_t st = (int64_t)n * (1 + nslices), nn = (int64_t)n * n;
  B7_TRY(b7_ensure(c, c->atmp, sizeof(double) * (size_t)st * B));
  PhaseScope ps(c, "alpha");
  double *t = (double *)c->atmp.p;
  hipLaunchKernelGGL(trmv_lower_kernel, dim3(n / 4, 1, B), dim3(256), 0, c->stream, Linv, resid, t, n, 1, nn, (int64_t)n, st);
  hipLaunchKernelGGL(trmv_lower_t_part_kernel, dim3(n / 64, nslices, B), dim3(256), 0, c->stream, Linv, (const double *)t,
                     t + n, n, 1, nn, st);
  hipLaunchKernelGGL(trmv_lower_t_sum_kernel, dim3((n + 255) / 256, 1, B), dim3(256), 0, c->stream, (const double *)(t + n),
                     alpha, n, c->N, 1, 1, nslices, st, (int64_t)n, report_src, report_dev, report_dev ? report_words : 0);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

// ---- pieces of the jitter fallback that used to run on the host ------------------------------------------------------
namespace {
__global__ void __launch_bounds__(1024)
    fro_norm_sq_kernel(const double *__restrict__ A, int n, int ld, double *__restrict__ out) {
  __shared__ double red[1024];
  const int t = threadIdx.x;
  double s = 0.0;
  for (int64_t e = t; e < (int64_t)n * n; e += 1024) {
    const int i = (int)(e / n), j = (int)(e - (int64_t)i * n);
    const double v = A[(int64_t)i * ld + j];
    s += v * v;
  }
  red[t] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {  // fixed tree: the same bits every time
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  if (t == 0) out[0] = red[0];
}
__global__ void __launch_bounds__(256) set_identity_kernel(double *__restrict__ L, double *__restrict__ dinv, int n) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (int64_t)n * n) {
    const int i = (int)(e / n), j = (int)(e - (int64_t)i * n);
    L[e] = i == j ? 1.0 : 0.0;
  }
  if (e < (int64_t)n * NB) {  // dinv: n/64 blocks of 64 x 64
    const int within = (int)(e % (NB * NB));
    dinv[e] = (within / NB == within % NB) ? 1.0 : 0.0;
  }
}
}  // namespace

int launch_fro_norm_sq(b7_ctx *c, const double *A, int n, int ld, double *out_dev) {
  hipLaunchKernelGGL(fro_norm_sq_kernel, dim3(1), dim3(1024), 0, c->stream, A, n, ld, out_dev);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_set_identity(b7_ctx *c) {
  const int n = c->Npad;
  const int64_t total = (int64_t)n * n;
  hipLaunchKernelGGL(set_identity_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                     (double *)c->L.p, (double *)c->dinv.p, n);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}
