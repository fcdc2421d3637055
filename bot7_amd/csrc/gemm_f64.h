// fp64 MFMA building blocks for gfx950: v_mfma_f64_16x16x4_f64 through an LDS-staged tile loop.
//
// Fragment maps (verified on MI355X by tools/mfma_probe.hip, profiles/r01_mfma_f64_probe.txt):
//   A operand: lane l holds A[row = l & 15][k = l >> 4]          (one f64)
//   B operand: lane l holds B[k = l >> 4][col = l & 15]          (one f64)
//   C/D:       lane l, register r holds D[row = (l >> 4) + 4 r][col = l & 15]
// Measured issue rate: one 16x16x4 MFMA per 64 cycles per SIMD = 78.6 TFLOP/s chip peak; a single dependent
// accumulator chain already reaches it, and fp64 VALU FMAs share the same units (they do not add).
//
// LDS image of an operand tile: [rows][BK] doubles with row stride BK + PAD.  A fragment read is one 8-byte
// read per lane at (row l&15, k 4s + (l>>4)).  hipcc fuses the reads of k-steps s and s+1 into ds_read2_b64,
// which is serviced in 16-lane groups over 32 four-byte banks: 16 rows at one k hit distinct bank pairs only when
// the stride is ODD (2*stride*row mod 32 must take 16 values); measured on the PAD = 2 image: 40 % of all LDS
// cycles were conflict cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r01b_*).  An odd stride is
// also conflict-free for plain ds_read_b64 (64 banks, 32-lane groups: (34 r + 2 kk) mod 64 are all distinct).
// The price is that rows are only 8-byte aligned, so staging stores are ds_write_b64 pairs.
// PAD = 2 (stride = 2 mod 4) is kept for the small-GEMM users: conflict-free for ds_read_b64 only.
#pragma once
#include <hip/hip_runtime.h>

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d4_t mfma_f64(double a, double b, d4_t c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Block-level NT/NN GEMM main loop.
//   C[m][n] += sum_{k in [kbeg, kend)} A[m][k] * Bop[k][n]
//   A is row-major [m][k] (leading dimension lda), pointing at the block's first row.
//   B_KN == false: B is row-major [n][k] (Bop = B^T), pointing at the block's first n-row.
//   B_KN == true : B is row-major [k][n], pointing at column n0 of row 0.
// BM x BN block tile, BK-deep LDS stages (double buffered), WM x WN waves, each wave owning a
// (BM/WM) x (BN/WN) sub-tile as TM x TN MFMA tiles.  All extents are multiples of the tile sizes (callers
// pad), kbeg/kend multiples of BK.  256 threads.
template <int BM, int BN, int BK, int WM, int WN, bool B_KN, int PAD = 2>
struct GemmF64 {
  static constexpr int THREADS = 64 * WM * WN;
  static constexpr int TM = BM / WM / 16;
  static constexpr int TN = BN / WN / 16;
  static constexpr int STRIDE = BK + PAD;
  static constexpr bool ALIGNED16 = (STRIDE % 2) == 0;
  static constexpr int A_CHUNKS = BM * BK / 2;  // 16-byte chunks per stage
  static constexpr int B_CHUNKS = BN * BK / 2;
  static constexpr int A_PER_T = A_CHUNKS / THREADS;
  static constexpr int B_PER_T = B_CHUNKS / THREADS;
  static constexpr int STAGE_DOUBLES = (BM + BN) * STRIDE;
  static constexpr int LDS_BYTES = 2 * STAGE_DOUBLES * 8;
  static_assert(A_CHUNKS % THREADS == 0 && B_CHUNKS % THREADS == 0, "tile/threads mismatch");
  static_assert(BK % 4 == 0, "BK must be a multiple of the MFMA depth");

  struct Regs {
    d2_t a[A_PER_T];
    d2_t b[B_PER_T];
  };

  __device__ static __forceinline__ void load_global(Regs &r, const double *__restrict__ A, int64_t lda,
                                                     const double *__restrict__ B, int64_t ldb, int k) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      int c = t + i * THREADS;
      int row = c / (BK / 2), kc = c % (BK / 2);
      r.a[i] = *reinterpret_cast<const d2_t *>(A + (int64_t)row * lda + k + 2 * kc);
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      int c = t + i * THREADS;
      if (!B_KN) {
        int row = c / (BK / 2), kc = c % (BK / 2);
        r.b[i] = *reinterpret_cast<const d2_t *>(B + (int64_t)row * ldb + k + 2 * kc);
      } else {
        int kr = c / (BN / 2), nc = c % (BN / 2);
        r.b[i] = *reinterpret_cast<const d2_t *>(B + (int64_t)(k + kr) * ldb + 2 * nc);
      }
    }
  }

  __device__ static __forceinline__ void store_lds(const Regs &r, double *__restrict__ sm) {
    const int t = threadIdx.x;
    double *sa = sm, *sb = sm + BM * STRIDE;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      int c = t + i * THREADS;
      int row = c / (BK / 2), kc = c % (BK / 2);
      if (ALIGNED16) {
        *reinterpret_cast<d2_t *>(sa + row * STRIDE + 2 * kc) = r.a[i];
      } else {
        sa[row * STRIDE + 2 * kc] = r.a[i][0];
        sa[row * STRIDE + 2 * kc + 1] = r.a[i][1];
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      int c = t + i * THREADS;
      if (!B_KN) {
        int row = c / (BK / 2), kc = c % (BK / 2);
        if (ALIGNED16) {
          *reinterpret_cast<d2_t *>(sb + row * STRIDE + 2 * kc) = r.b[i];
        } else {
          sb[row * STRIDE + 2 * kc] = r.b[i][0];
          sb[row * STRIDE + 2 * kc + 1] = r.b[i][1];
        }
      } else {
        int kr = c / (BN / 2), nc = c % (BN / 2);
        sb[(2 * nc) * STRIDE + kr] = r.b[i][0];
        sb[(2 * nc + 1) * STRIDE + kr] = r.b[i][1];
      }
    }
  }

  // Same, for a stage that lies inside the diagonal block of a lower-triangular A: krel = offset of the stage's
  // first k inside that block.  A[row][k] = 0 for k > row, so the 16-row strip i of this wave contributes to
  // k-step s only while krel + 4 s <= (last row of the strip); the branch is wave-uniform.
  __device__ static __forceinline__ void compute_stage_tri(const double *__restrict__ sm, d4_t (&acc)[TM][TN],
                                                           int krel) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: the skips become s_cbranch
    const int wm = wave / WN, wn = wave % WN;
    const double *sa = sm + (wm * (BM / WM) + (lane & 15)) * STRIDE + (lane >> 4);
    const double *sb = sm + BM * STRIDE + (wn * (BN / WN) + (lane & 15)) * STRIDE + (lane >> 4);
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      const int kk = krel + 4 * s;
      if (kk > wm * (BM / WM) + 16 * (TM - 1) + 15) continue;  // nothing left for this wave
      double bf[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = sb[j * 16 * STRIDE + 4 * s];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (kk > wm * (BM / WM) + 16 * i + 15) continue;
        const double af = sa[i * 16 * STRIDE + 4 * s];
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_f64(af, bf[j], acc[i][j]);
      }
    }
  }

  // The fragments of k-step s + 1 are requested BETWEEN the MFMAs of step s, one read behind each MFMA, in the order written
  // (sched_barrier keeps the compiler from regrouping them): tools/mfma_feed_probe.hip measures 94 cycles per
  // v_mfma_f64_16x16x4 for "read the step's fragments, then its MFMAs" with one wave per SIMD, 87 with the reads of the next
  // step hoisted in front of this step's MFMAs, 75.5 with the reads interleaved like this (64 from registers).  The
  // arithmetic and its order are untouched.
  __device__ static __forceinline__ void compute_stage(const double *__restrict__ sm, d4_t (&acc)[TM][TN]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const double *sa = sm + (wm * (BM / WM) + (lane & 15)) * STRIDE + (lane >> 4);
    const double *sb = sm + BM * STRIDE + (wn * (BN / WN) + (lane & 15)) * STRIDE + (lane >> 4);
    constexpr int STEPS = BK / 4, NF = TM + TN;
    double f[2][NF];  // fragments of a step: A strips first, then B strips
#pragma unroll
    for (int q = 0; q < NF; ++q) f[0][q] = q < TM ? sa[q * 16 * STRIDE] : sb[(q - TM) * 16 * STRIDE];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      int q = 0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = mfma_f64(f[cur][i], f[cur][TM + j], acc[i][j]);
          __builtin_amdgcn_sched_barrier(0);
          if (s + 1 < STEPS && q < NF) {
            f[nxt][q] = q < TM ? sa[q * 16 * STRIDE + 4 * (s + 1)] : sb[(q - TM) * 16 * STRIDE + 4 * (s + 1)];
            ++q;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      if (s + 1 < STEPS)
        for (; q < NF; ++q) f[nxt][q] = q < TM ? sa[q * 16 * STRIDE + 4 * (s + 1)] : sb[(q - TM) * 16 * STRIDE + 4 * (s + 1)];
    }
  }

  // sm: 2 * STAGE_DOUBLES doubles of LDS.  Ends with all waves past the last LDS read (safe to reuse sm
  // after a __syncthreads() by the caller).
  // TRI: A's block-row is the diagonal block-row of a lower-triangular matrix whose diagonal block occupies
  // the last BM columns [kend - BM, kend): structural zeros there are skipped (compute_stage_tri).
  template <bool TRI = false>
  __device__ static __forceinline__ void run(const double *__restrict__ A, int64_t lda, const double *__restrict__ B,
                                             int64_t ldb, int kbeg, int kend, d4_t (&acc)[TM][TN],
                                             double *__restrict__ sm) {
    if (kbeg >= kend) return;
    Regs r;
    load_global(r, A, lda, B, ldb, kbeg);
    store_lds(r, sm);
    __syncthreads();
    int cur = 0;
    const int ksplit = TRI ? (kend - BM > kbeg ? kend - BM : kbeg) : kend;
    for (int k = kbeg; k < ksplit; k += BK) {
      const bool more = (k + BK) < kend;
      if (more) load_global(r, A, lda, B, ldb, k + BK);
      compute_stage(sm + cur * STAGE_DOUBLES, acc);
      if (more) store_lds(r, sm + (cur ^ 1) * STAGE_DOUBLES);
      __syncthreads();
      cur ^= 1;
    }
    if (TRI) {
      for (int k = ksplit; k < kend; k += BK) {
        const bool more = (k + BK) < kend;
        if (more) load_global(r, A, lda, B, ldb, k + BK);
        compute_stage_tri(sm + cur * STAGE_DOUBLES, acc, k - (kend - BM));
        if (more) store_lds(r, sm + (cur ^ 1) * STAGE_DOUBLES);
        __syncthreads();
        cur ^= 1;
      }
    }
  }

  // Output coordinates of accumulator element (i, j, r) of this lane inside the block tile.
  __device__ static __forceinline__ int out_row(int i, int r) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave / WN) * (BM / WM) + i * 16 + (lane >> 4) + 4 * r;
  }
  __device__ static __forceinline__ int out_col(int j) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave % WN) * (BN / WN) + j * 16 + (lane & 15);
  }
};
