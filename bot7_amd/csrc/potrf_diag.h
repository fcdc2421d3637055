// The serial heart of the Cholesky: factor AND invert one 64x64 diagonal block that sits in LDS.  Shared by the
// per-panel launch schedule (potrf.hip: potrf_diag_kernel) and the persistent schedule (potrf_persist.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "gemm_f64.h"

namespace b7diag {

constexpr int NB = 64;

// ---- 64x64 diagonal block: factor and invert, blocked by 16 -----------------------------------------------------
// The serial part of the whole Cholesky.  Per 16-wide sub-block kb, wave 0 factors the 16x16 diagonal sub-block with
// one matrix row per lane (16 registers), pivots and multipliers broadcast through DPP (no LDS, no barriers); the
// same instruction stream inverts it (lane row 0) and solves the sub-panel rows below it (lane rows 1..3).  The
// trailing update inside the 64x64 block and the assembly of the 64x64 inverse from the four 16x16 inverses (two
// doubling levels, X = -inv(C) * (B * inv(A))) are 16x16x16 MFMA products; VAR 1 runs as many of them as the data
// dependences allow on waves 1..3 while wave 0 is already in the next factor step.  The f64 accumulator layout
// (row = (l>>4)+4r) is exactly the B-operand layout of k-step r, so T = B*inv(A) feeds the second product of a
// doubling level straight from registers.
constexpr int DLD = NB + 2;  // LDS row stride: = 2 (mod 4) doubles -> conflict-free ds_read_b64 fragments
constexpr int TLD = 34;

template <int J>
__device__ __forceinline__ double row_share(double v) {  // value of lane J of this lane's 16-lane row
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x150 + J, 0xF, 0xF, true);  // every lane of the row is a valid source
  hi = __builtin_amdgcn_mov_dpp(hi, 0x150 + J, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// 64-bit row broadcast as ONE instruction (v_mov_b64_dpp row_newbcast: the only DPP control the fp64 ALU has)
template <int J>
__device__ __forceinline__ double row_share64(double v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xF, 0xF, false);
}
// acc += (value of `src` in lane J of this lane's 16-lane row) * mul, as ONE instruction: v_fmac_f64 with its first
// operand taken through DPP (5 cycles of issue; mov_dpp + fma is 8 + 5, the old 32-bit pair 16 + 5 --
// tools/valu_probe.hip).  hipcc does not form it from mov_dpp + fma, hence the asm.  A DPP read needs two wait
// states after a VALU write of the same register and the compiler cannot see into asm: the statements are volatile
// (kept in program order) and every caller keeps at least two other instructions between the write of `src` and
// this read (see the 16x16 routine; s_nop 1 would cost 8 cycles each).
// The same broadcast for a value that an asm statement wrote (the pivot chain): straight from that register into a fresh
// one.  The builtin ties source and destination, so the compiler copies the value first and pads copy -> DPP with its own
// wait states (three issue slots per column); here the two wait states after the asm write are the caller's business, as
// for fmac_share.
template <int J>
__device__ __forceinline__ double row_share64_raw(double v) {
  double out;
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(v), "n"(J));
  return out;
}
template <int J>
__device__ __forceinline__ void fmac_share(double &acc, double src, double mul) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc)
               : "v"(src), "v"(mul), "n"(J));
}
// The same broadcast multiply-add as two compiler-scheduled instructions (v_mov_b64_dpp + v_fma_f64): the reference form
// of VAR 2, which tests/test_gpu_parity.py compares bit for bit with the asm form -- the hazard spacing around the
// inline asm is hand-kept, and a compiler bump may reorder its neighbours.
template <int J>
__device__ __forceinline__ void fmac_share_ref(double &acc, double src, double mul) {
  acc = __builtin_fma(row_share64<J>(src), mul, acc);
}
template <int VAR, int J>
__device__ __forceinline__ void fmac_bcast(double &acc, double src, double mul) {
  if constexpr (VAR == 2)
    fmac_share_ref<J>(acc, src, mul);
  else
    fmac_share<J>(acc, src, mul);
}
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// A_ij -= L_i,ko L_j,ko'  for one 16x16 tile of the 64x64 block in LDS (ko = first column of the source block column)
__device__ __forceinline__ void block16_update(double *__restrict__ A, int i, int j, int ko, int lr, int lq) {
  d4_t c;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) c[rr] = A[(i * 16 + lq + 4 * rr) * DLD + j * 16 + lr];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4)
    c = mfma_f64(-A[(i * 16 + lr) * DLD + ko + 4 * s4 + lq], A[(j * 16 + lr) * DLD + ko + 4 * s4 + lq], c);
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) A[(i * 16 + lq + 4 * rr) * DLD + j * 16 + lr] = c[rr];
}


// The dependent chain that opens column J of the 16x16 factor step, one link per call so that the caller can place the
// links between the previous column's multiply-adds (diag_core): broadcast the pivot, its reciprocal by estimate and two
// Newton steps, the multiplier -c_iJ r_J, s_J = rhs_J - sum and its scaled copy, the pivot kept for the failure test.
template <int VAR>
struct PivotChain {
  static constexpr int LINKS = 10;
  double dj, r, e, nm, xs;
  template <int I, int J>
  __device__ __forceinline__ void link(double (&a)[16], double (&x)[16], const double (&rhs)[16], const double (&sacc)[16],
                                       double &dmine, int lr) {
    if constexpr (I == 0) dj = (VAR == 2) ? row_share64<J>(a[J]) : row_share64_raw<J>(a[J]);
    if constexpr (I == 1) r = __builtin_amdgcn_rcp(dj);
    if constexpr (I == 2) x[J] = rhs[J] - sacc[J];  // independent of the reciprocal: fills its latency
    if constexpr (I == 3) e = __builtin_fma(-dj, r, 1.0);
    if constexpr (I == 4) r = __builtin_fma(e, r, r);
    if constexpr (I == 5) e = __builtin_fma(-dj, r, 1.0);
    if constexpr (I == 6) r = __builtin_fma(e, r, r);
    if constexpr (I == 7) nm = -(a[J] * r);  // -l~_iJ of the LDL' form (rows i > J)
    if constexpr (I == 8) xs = x[J] * r;
    if constexpr (I == 9) dmine = (lr == J) ? dj : dmine;
  }
};

// A: [64][DLD] the block (lower part meaningful; VAR 1 expects I_16 in rows 0..15 x columns 48..63), X: [64][DLD]
// zero on entry, T: [32][TLD] scratch.  On return (after the closing barrier) A's lower triangle is L_pp and X is
// inv(L_pp) (zero above the diagonal).  A non-positive / NaN / subnormal pivot is reported dpotrf-style through
// info[0] (1-based global row, first failure wins); the arithmetic after it is garbage the host discards.
// 256 threads, every thread calls.  stamps (nullable unless STAMP): s_memtime phase stamps, slots 2..17.
struct NoHook {
  __device__ __forceinline__ void operator()(int, int) const {}
};
// hook(kb, wave) runs on waves 1..3 (VAR 1) at the end of their share of step kb's factor phase, i.e. in time they would
// otherwise spend waiting for wave 0 at the phase's barrier (they have nothing at all to do at kb = 0, wave 3 nothing at
// kb = 2): the persistent schedule finishes its look-ahead update at kb = 0 (it may write the 16x16 sub-tiles (i, j),
// 1 <= j <= i <= 3, of A: nothing in step 0's factor phase reads or writes them) and fetches the next panel's tiles at
// kb >= 2 (kb = 4 stands for the routine's tail, where waves 1..3 wait for wave 0's last inverse pair).  Apart from that
// it must not touch A, X or T.
// nlive (VAR >= 1): rows / columns [nlive, 64) of the block are padding -- exact identity rows that no real row couples to
// (the small-problem kernels: N = 25 observations in a 64-block).  A 16-wide step that lies wholly in the padding has
// nothing to factor: the chain would reproduce L = I, inv = I (+0.0 elsewhere) bit for bit in 1.6 us of dependent
// instructions, so wave 0 writes the sixteen ones of the inverse instead and leaves the block as it is.  Everything else
// (barriers, in-block updates, inverse doubling) runs as always.
// Every thread of a 256-thread group passes exactly DIAG_CORE_BARRIERS workgroup barriers in here (VAR >= 1): waves beyond
// the four that call diag_core keep in step with diag_bystander.
constexpr int DIAG_CORE_BARRIERS = 10;
// LB: the routine's barriers order LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): a caller with global stores in flight
// (gp_small.hip writes its results while the factorisation goes on) does not wait for them at every barrier, as
// __syncthreads (vmcnt(0) as well) would.  Nothing inside the routine communicates through global memory.
template <bool LB>
__device__ __forceinline__ void diag_barrier() {
  if constexpr (LB) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  } else {
    __syncthreads();
  }
}
template <bool LB = false, class F>
__device__ __forceinline__ void diag_bystander(F &&between) {
  for (int bi = 0; bi < DIAG_CORE_BARRIERS; ++bi) {
    between(bi);
    diag_barrier<LB>();
  }
}
template <int VAR, bool STAMP, class Hook = NoHook, bool LB = false>
__device__ __forceinline__ void diag_core(double *__restrict__ A, double *__restrict__ X, double *__restrict__ T, int p,
                                          int *__restrict__ info, unsigned long long *__restrict__ stamps,
                                          const Hook &hook = Hook(), int nlive = NB) {
#define B7_DIAG_STAMP(i) \
  if (STAMP && threadIdx.x == 0) stamps[i] = __builtin_amdgcn_s_memtime()
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  // 16-blocks that hold anything (nlive): a block row i >= nl of the factor is e_i' -- zero left of its diagonal -- so every
  // product below whose left operand is such a row block adds exact zeros to what is already there (+0.0 in X, the identity
  // in A: the caller's images) and is left out; T's tiles for those rows are written as zeros (its storage is scratch).
  // nlive = 64: nl = 4, nothing changes.
  const int nl = (nlive + 15) >> 4;
  for (int kb = 0; kb < 4; ++kb) {
    const int o = kb * 16;
    if (VAR >= 1 && wave == 0 && o >= nlive) {
      if (lane < 16) X[(o + lane) * DLD + o + lane] = 1.0;  // a step wholly in the padding: inv = I, the block stays I (see nlive)
    } else if (VAR >= 1 && wave == 0) {
      // Square-root-free pivot chain, built for ISSUE cycles: one wave issues a VALU instruction every ~5 cycles
      // and this routine is bound by that, not by latency (tools/valu_probe.hip).  Lane lr keeps row lr of
      // C = L diag(sqrt(d)) (c_ij = l_ij sqrt(d_j), pivots d_j = c_jj).  Per column j: broadcast d_j, r_j = 1/d_j
      // (estimate + 2 Newton steps), one multiplier -c_ij r_j, then one fused DPP multiply-add per remaining column.
      // Row j of the UNIT-lower inverse (of C diag(r)) needs only r, so it rides along in the same block of
      // straight-line code; the 16 values y_j = 1/sqrt(d_j) that turn C and that inverse into L and inv(L) are
      // computed once, lane j doing y_j from the pivot it captured, and broadcast.  No branch inside the chain:
      // a non-positive (or NaN) pivot is found afterwards from the captured pivots; the arithmetic after it is
      // garbage that the host discards together with this attempt.
      // The recurrence that inverts the block, s_j = rhs_j - sum_{k<j} c_jk (r_k s_k), is a forward substitution
      // with the identity as right-hand side.  The four 16-lane rows of the wave replicate the factorisation
      // anyway (DPP broadcasts stay inside a row), so rows 1..3 run the SAME instructions on a different right-hand
      // side: one row each of the sub-panel blocks below, which come out solved (L_ik = A_ik inv(L_kk)') for free.
      double a[16], x[16], rhs[16], sacc[16];
      const int ib = kb + lq;  // sub-panel block of this lane row (lq >= 1); lane row 0 carries the inverse
      const bool has_sub = lq > 0 && ib < 4;
      // every lane reads its right-hand side through one pointer, no selects: the identity rows live in the
      // block's unused upper-right corner (written before the loop), lane rows without a block read zeros from X's
      // upper triangle
      const double *rsrc = (lq == 0) ? A + lr * DLD + 48 : (has_sub ? A + (ib * 16 + lr) * DLD + o : X + 48);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        a[k] = A[(o + lr) * DLD + o + k];
        rhs[k] = rsrc[k];
        sacc[k] = 0.0;
      }
      double dmine = 1.0;
      // The substitution is RIGHT-looking: as soon as s_j is known its contribution c_mj (r_j s_j) goes to the sums of all
      // later rows m, one independent multiply-add each, instead of row j collecting its j terms in one dependent chain when
      // its turn comes.  Every sum still receives its terms in ascending k from zero -- the bits are those of the left-looking
      // form -- but nothing in a column now waits on anything except the pivot chain itself (d_j, its reciprocal, the
      // multiplier, s_j: ten dependent operations, ~10 cycles of latency each against ~5 of issue), and that chain is
      // software-pipelined by hand into the PREVIOUS column's multiply-adds: the first pair of a column finishes what the
      // next column's chain needs (c_j+1,j+1 and the sum of row j + 1), the other 2 (14 - j) are independent of it and
      // one link of the chain goes between every two (from column 5 on: every one) of them.  sched_barrier pins that order;
      // the measured cost of a 16-column step fell from 4400 to the cycles tools/nll_small_stamps.py reports.
      PivotChain<VAR> pc;
      static_for<PivotChain<VAR>::LINKS>([&](auto Ic) { pc.template link<decltype(Ic)::value, 0>(a, x, rhs, sacc, dmine, lr); });
      static_for<16>([&](auto Jc) {
        constexpr int j = Jc, R = 15 - j;            // R columns k > j
        constexpr int STRIDE = (2 * (R - 2) >= 2 * PivotChain<VAR>::LINKS) ? 2 : 1;  // multiply-adds between two links of the chain
        const double nm = pc.nm, xsj = pc.xs;         // this column's multipliers: the chain below overwrites pc
        static_for<2 * (R > 0 ? R : 0)>([&](auto Fc) {
          constexpr int f = Fc, k = j + 1 + f / 2;
          if constexpr (f % 2 == 0)
            fmac_bcast<VAR, k>(a[k], a[j], nm);      // c_ik -= l~_ij c_kj; a[j] was last written >= 2 asm ago
          else
            fmac_bcast<VAR, k>(sacc[k], a[j], xsj);  // sum of row k += c_kj (r_j s_j)
          // links of column j + 1's chain: after the second pair (c_j+1,j+1 was written by the first multiply-add of
          // this list and is read through DPP: two wait states, kept here by the three asm statements in between)
          if constexpr (f >= 3 && (f - 3) % STRIDE == 0 && (f - 3) / STRIDE < PivotChain<VAR>::LINKS) {
            __builtin_amdgcn_sched_barrier(0);
            pc.template link<(f - 3) / STRIDE, j + 1>(a, x, rhs, sacc, dmine, lr);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        if constexpr (j < 15) {
          // what is left of the chain once the multiply-adds have run out (late columns: the block's serial tail)
          constexpr int done = (2 * R >= 4) ? ((2 * R - 4) / STRIDE + 1 < PivotChain<VAR>::LINKS ? (2 * R - 4) / STRIDE + 1 : PivotChain<VAR>::LINKS) : 0;
          if constexpr (2 * R < 4) asm volatile("s_nop 1");  // column 14: only one asm statement follows the write of c_15,15
          static_for<PivotChain<VAR>::LINKS - done>(
              [&](auto Ic) { pc.template link<done + decltype(Ic)::value, j + 1>(a, x, rhs, sacc, dmine, lr); });
        }
      });
      // also NaN, like dpotrf's test; a subnormal pivot counts as failed too (its reciprocal overflows)
      const unsigned long long badmask = __ballot(!(dmine >= 2.2250738585072014e-308)) & 0xFFFFull;
      if (badmask != 0 && lane == 0 && info[0] == 0) info[0] = p * NB + o + __ffsll((long long)badmask);
      double ymine = __builtin_amdgcn_rsq(dmine);  // 1/sqrt(d_lr): hardware estimate + two Newton steps
      const double hp = 0.5 * dmine;
      ymine = ymine * __builtin_fma(-hp * ymine, ymine, 1.5);
      ymine = ymine * __builtin_fma(-hp * ymine, ymine, 1.5);
      if constexpr (VAR != 2) asm volatile("s_nop 1" : "+v"(ymine));  // two wait states between the VALU write and the DPP reads
      static_for<16>([&](auto Kc) {
        constexpr int k = Kc;
        const double yk = (VAR == 2) ? row_share64<k>(ymine) : row_share64_raw<k>(ymine);
        a[k] *= yk;  // L[lr][k] = c_lr,k / sqrt(d_k)
        x[k] *= yk;  // lane row 0: inv(L)[k][lr] = s_k / sqrt(d_k); lane rows 1..3: L_ik[lr][k]
      });
      // one store loop for all lanes: lane row 0 writes its column of inv(L_kk) (stride DLD), rows 1..3 their
      // solved sub-panel row (stride 1), rows without a block into an unused upper block of A (never read)
      double *wdst = (lq == 0) ? X + o * DLD + o + lr : (has_sub ? A + (ib * 16 + lr) * DLD + o : A + lr * DLD + 32);
      const int wstride = (lq == 0) ? DLD : 1;
#pragma unroll
      for (int k = 0; k < 16; ++k) wdst[k * wstride] = x[k];
      if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) A[(o + lr) * DLD + o + k] = (k <= lr) ? a[k] : 0.0;
      }
    }
    if (VAR >= 1 && wave > 0 && kb == 0) hook(0, wave);  // waves 1..3 have nothing of their own to do in step 0
    if (VAR >= 1 && wave > 0 && kb > 0) {
      // while wave 0 factors: the previous step's update of the tiles right of block column kb, with the previous
      // step's panel (columns o - 16 ..).  kb = 1: (2,2) (3,2) (3,3); kb = 2: (3,3) again with panel 1; kb = 3: none.
      // Nothing wave 0 touches in this phase (tile (kb,kb), the rows (i,kb) below it, X) is read or written here.
      const int ko = o - 16;
      if (kb == 1) {
        const int i = wave == 1 ? 2 : 3, j = wave == 3 ? 3 : 2;
        if (i < nl) block16_update(A, i, j, ko, lr, lq);
      } else if (kb == 2 && wave == 1) {
        if (3 < nl) block16_update(A, 3, 3, ko, lr, lq);
      } else if (kb == 2 && wave == 2 && 1 < nl) {
        // inverse doubling 16 -> 32 of the pair (0,1): X_10 = -inv(L_11) * (L_10 * inv(L_00)); its inputs are final
        d4_t t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) t = mfma_f64(A[(16 + lr) * DLD + 4 * s4 + lq], X[(4 * s4 + lq) * DLD + lr], t);
        d4_t xx = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) xx = mfma_f64(-X[(16 + lr) * DLD + 16 + 4 * s4 + lq], t[s4], xx);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) X[(16 + lq + 4 * rr) * DLD + lr] = xx[rr];
      } else if (kb == 3) {
        // first half of the doubling 32 -> 64: T = L[32:64, 0:32] * X[0:32, 0:32] (both final since step 2);
        // tiles (ti, tj): wave 1 (0,0) and (0,1), wave 2 (1,0), wave 3 (1,1)
        for (int q = 0; q < (wave == 1 ? 2 : 1); ++q) {
          const int ti = wave == 1 ? 0 : 1, tj = wave == 1 ? q : wave - 2;
          d4_t t0 = {0.0, 0.0, 0.0, 0.0}, t1 = {0.0, 0.0, 0.0, 0.0};
          if (2 + ti < nl) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
              t0 = mfma_f64(A[(32 + ti * 16 + lr) * DLD + 4 * s4 + lq], X[(4 * s4 + lq) * DLD + tj * 16 + lr], t0);
              t1 = mfma_f64(A[(32 + ti * 16 + lr) * DLD + 16 + 4 * s4 + lq], X[(16 + 4 * s4 + lq) * DLD + tj * 16 + lr], t1);
            }
          }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) T[(ti * 16 + lq + 4 * rr) * TLD + tj * 16 + lr] = t0[rr] + t1[rr];
        }
      }
      hook(kb, wave);
    }
    if (VAR == 0 && wave == 0) {
      // every 16-lane row of the wave holds the same 16x16 sub-block (lane lr = matrix row lr)
      double a[16], r[16], x[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) a[k] = A[(o + lr) * DLD + o + k];
      static_for<16>([&](auto Jc) {
        constexpr int j = Jc;
        double pj = row_share<j>(a[j]);
        if (!(pj > 0.0)) {  // also catches NaN, like dpotrf's test; uniform over the wave
          if (lane == 0 && info[0] == 0) info[0] = p * NB + o + j + 1;
          pj = 1.0;  // keep the arithmetic finite; the host discards this attempt
        }
        // 1/sqrt(pj): hardware estimate + two Newton steps (quadratic: 2^-26 -> full), then dj = pj * rj.
        // LAPACK's dpotf2 likewise scales the column by the reciprocal of the pivot's root.
        double y = __builtin_amdgcn_rsq(pj);
        const double hp = 0.5 * pj;
        y = y * __builtin_fma(-hp * y, y, 1.5);
        y = y * __builtin_fma(-hp * y, y, 1.5);
        r[j] = y;
        const double dj = pj * y;
        a[j] = (lr == j) ? dj : a[j] * y;
        static_for<16>([&](auto Kc) {
          constexpr int k = Kc;
          if constexpr (k > j) {
            const double lk = row_share<k>(a[j]);  // L[k][j]
            a[k] -= a[j] * lk;                     // meaningful for rows >= k
          }
        });
      });
      // inverse: lane lr computes column lr of inv(L_kk); x[i] = X[i][lr]
      static_for<16>([&](auto Ic) {
        constexpr int i = Ic;
        double sacc = 0.0;
        static_for<16>([&](auto Kc) {
          constexpr int k = Kc;
          if constexpr (k < i) sacc += row_share<i>(a[k]) * x[k];  // L[i][k] * X[k][c]
        });
        x[i] = (i == lr) ? r[i] : ((i > lr) ? -(r[i] * sacc) : 0.0);
      });
      if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          A[(o + lr) * DLD + o + k] = (k <= lr) ? a[k] : 0.0;
          X[(o + k) * DLD + o + lr] = x[k];
        }
      }
    }
    B7_DIAG_STAMP(2 + 4 * kb);
    diag_barrier<LB>();
    B7_DIAG_STAMP(3 + 4 * kb);
    // sub-panel: L_ik = A_ik * inv(L_kk)'  for block rows ib > kb, one 16x16 block per wave (VAR 1 solved it above)
    if (VAR == 0) {
      const int ib = kb + 1 + wave;
      if (ib < 4) {
        d4_t c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          c = mfma_f64(A[(ib * 16 + lr) * DLD + o + 4 * s4 + lq], X[(o + lr) * DLD + o + 4 * s4 + lq], c);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) A[(ib * 16 + lq + 4 * rr) * DLD + o + lr] = c[rr];
      }
      diag_barrier<LB>();
    }
    B7_DIAG_STAMP(4 + 4 * kb);
    // trailing update inside the block: A_ij -= L_ik L_jk'  for kb < j <= i < 4.  VAR 1 does only block column
    // kb + 1 here (all the next factor step reads); the tiles right of it were left to waves 1..3 of the NEXT
    // step's factor phase, where they idle anyway (see there).
    if (VAR >= 1) {
      const int i = kb + 1 + wave;
      if (i < nl) block16_update(A, i, kb + 1, o, lr, lq);
    } else {
      int pidx = 0;
      for (int i = kb + 1; i < 4; ++i)
        for (int j = kb + 1; j <= i; ++j, ++pidx) {
          if ((pidx & 3) != wave) continue;
          block16_update(A, i, j, o, lr, lq);
        }
    }
    diag_barrier<LB>();
    B7_DIAG_STAMP(5 + 4 * kb);
  }

  // What is left of the inverse doubling (VAR 1 did the pair (0,1) and T = L[32:64,0:32] X[0:32,0:32] inside the
  // loop, on waves that were idle): the pair (2,3), then X[32:64, 0:32] = -X[32:64, 32:64] * T.
  if (VAR >= 1 && wave > 0) hook(4, wave);  // waves 1..3 wait for wave 0's pair (2,3) here
  if ((VAR >= 1 ? wave == 0 : wave < 2) && (VAR == 0 || 3 < nl)) {
    const int a0 = (VAR >= 1 ? 1 : wave) * 32, c0 = a0 + 16;
    d4_t t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
      t = mfma_f64(A[(c0 + lr) * DLD + a0 + 4 * s4 + lq], X[(a0 + 4 * s4 + lq) * DLD + a0 + lr], t);
    d4_t xx = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) xx = mfma_f64(-X[(c0 + lr) * DLD + c0 + 4 * s4 + lq], t[s4], xx);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) X[(c0 + lq + 4 * rr) * DLD + a0 + lr] = xx[rr];
  }
  diag_barrier<LB>();
  {
    const int ti = wave >> 1, tj = wave & 1;
    if (VAR == 0) {
      d4_t t0 = {0.0, 0.0, 0.0, 0.0}, t1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        t0 = mfma_f64(A[(32 + ti * 16 + lr) * DLD + 4 * s4 + lq], X[(4 * s4 + lq) * DLD + tj * 16 + lr], t0);
        t1 = mfma_f64(A[(32 + ti * 16 + lr) * DLD + 16 + 4 * s4 + lq], X[(16 + 4 * s4 + lq) * DLD + tj * 16 + lr], t1);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) T[(ti * 16 + lq + 4 * rr) * TLD + tj * 16 + lr] = t0[rr] + t1[rr];
      diag_barrier<LB>();
    }
    if (VAR == 0 || 2 + ti < nl) {
      d4_t x0 = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        x0 = mfma_f64(-X[(32 + ti * 16 + lr) * DLD + 32 + 4 * s4 + lq], T[(4 * s4 + lq) * TLD + tj * 16 + lr], x0);
        x1 = mfma_f64(-X[(32 + ti * 16 + lr) * DLD + 48 + 4 * s4 + lq], T[(16 + 4 * s4 + lq) * TLD + tj * 16 + lr], x1);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) X[(32 + ti * 16 + lq + 4 * rr) * DLD + tj * 16 + lr] = x0[rr] + x1[rr];
    }
  }
  diag_barrier<LB>();
#undef B7_DIAG_STAMP
}

constexpr int DIAG_LDS_BYTES = (2 * NB * DLD + 32 * TLD) * 8;

}  // namespace b7diag
