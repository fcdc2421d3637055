// Posterior variance over a chunk of candidates:  var_j = amp - || L^-1 k*_j ||^2.
//
// Reference arithmetic replaced: the variance half of model:predict(X_obs, Y_obs, X_hid, hyp, {mean, var})
// (call sites scores/expected_improvement.lua:63, scores/confidence_bound.lua:63; gp_regressor itself is in
// the absent `gp` package): textbook GP regression, V = L^-1 K(X*,X)', var = k** - colsumsq(V).
//
// This is the dominant kernel of the whole path: M * N^2 flops (4.19 MFLOP per candidate at N = 2048), all
// of it on v_mfma_f64_16x16x4_f64.  V is never stored: a workgroup owns 256 (large grids) or 128 (small grids)
// candidates, walks the 128-row tiles of the explicit inverse factor top to bottom (k only up to the diagonal: Linv
// is lower triangular) and folds each finished tile of V into per-candidate sums of squares held in registers.
//
// The arithmetic is fixed, and stated so that a host model can reproduce it bit for bit (tools/post_probe.hip does):
//   v[n][c]  = fma chain over k ascending of Linv[n][k] * K*[c][k]     (what a chain of v_mfma_f64_16x16x4_f64 computes,
//              tools/mfma_acc_probe.hip)
//   per lane group g = 0..3, 128-row tile t and row half h:  s = fma chain of v^2 over rows 16 I + g + 4 r  (I in the
//              half ascending, r ascending), colss[h] += s over t ascending
//   ss[c]    = ((g0 + g1) + (g2 + g3))_{h=0} + (...)_{h=1}
// so results do not depend on the grid size, the shard layout or the run -- the arg-max downstream relies on that.
//
// Shape history (N = 2048, 262144 candidates per launch, one MI355X): round 1, compiler-scheduled GEMM tile loop,
// 128 x 128 tile 44.6 TF -> 2 blocks/CU 59.2 -> odd LDS stride, zero-strip skip 61.6 -> 128 x 256 tile with 8 waves and
// a static priority raise 64.7 TF.  Round 2: in-kernel stamps (tools/post_clock.py: s_memtime / s_memrealtime around
// every workgroup after >= 2 s of back-to-back launches) showed the chip holding 2.37 GHz -- not the 2.1 GHz round 1
// had inferred from counters -- and the workgroup NOT issuing MFMAs for 14 % of its cycles: the two waves of a SIMD
// run in lock step, sit in the same LDS wait at mid-stage and at the barrier, and 256 registers per wave leave no room
// to fetch fragments ahead.  The kernels below: 96 % issue efficiency at 2.31-2.33 GHz, 71.7 TF (128-row n-tiles) and
// 73 TF (256-row n-tiles) = 91-93 % of the 78.6 TF peak.
//
// Algorithmic work per launch: rows * Npad^2 flops (triangle exploited), bytes: the K* chunk is read
// (t+1)/T-weighted ~ (T+1)/2 times from L2/MALL/HBM (T = Npad/128 n-tiles), Linv once per workgroup from L2.
#include "b7_internal.h"
#include <utility>

#ifdef B7_POST_STAMPS
// Diagnostic build only (tools/post_clock.py; never defined for the shipped library): every workgroup records
// s_memtime / s_memrealtime at its start and end into a buffer nothing else reads; the in-kernel clock is
// d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).
__device__ unsigned long long b7_post_stamps[4 * 4096];
extern "C" int b7dbg_post_stamps(unsigned long long *out, int nblocks) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(b7_post_stamps), sizeof(unsigned long long) * 4 * nblocks);
}
#endif

typedef double d2_t __attribute__((ext_vector_type(2)));

// Several fits over the same candidates in one launch (b7_eval_nominate: the hyper samples of a nomination): grid.y = fit;
// L^-1, K* and the output are strided per fit, amp (and the noise term when the variance includes it) come from arrays.
// base == nullptr: a single fit, the scalar arguments apply.
struct PostBatch {
  int64_t sLinv = 0, sks = 0, svar = 0;
  const double *base = nullptr, *var_add = nullptr;
};

// ---- one wave per SIMD, one continuous MFMA stream -----------------------------------------------------------------------
// Four waves (512 registers each); a wave owns all 128 rows of the n-tile and 16 NJ candidates: 8 x NJ accumulator
// tiles of 16 x 16 (NJ = 4: 128 x 256 workgroup tile; NJ = 2: 128 x 128, for grids with fewer than one 256-candidate
// workgroup per CU).  LDS image of a stage: [128 + 64 NJ][16 + 1] doubles (odd row stride: ds_read2_b64 fragment
// reads are conflict-free), two buffers.  A wave keeps two fragment sets and the next stage's global data in flight:
//   first half of stage n  : MFMAs of k-steps 0,1 (set F0) | read F1 = k-steps 2,3 of stage n | store G (stage n+1,
//                            loaded a stage ago) into the other LDS buffer | issue the global loads of stage n+2
//   barrier (mid-stage)    : stage n+1's image is complete; nobody reads this stage's buffer any more
//   second half of stage n : MFMAs of k-steps 2,3 (set F1) | read F0 = k-steps 0,1 of stage n+1
// The side operations are placed between the MFMAs in program order (about one every second MFMA), so a wave never
// waits on LDS or on memory, only on the barrier's skew.  The stage stream runs across the n-tiles without a
// prologue per tile; in the 8 diagonal stages of a tile the 16-row strips above the diagonal are skipped.  Two LDS
// buffers suffice: every read of a buffer precedes the mid-stage barrier of the stage after the one that computes from
// it, every write to it follows that barrier.
namespace w4 {
constexpr int BM = 128, BK = 16, LD = BK + 1, A_DBL = BM * LD;

// The accumulator tiles of a wave are AGPRs addressed by NUMBER inside inline asm, never held in a C++ variable:
// tile (i, j) is a[8 (NJ i + j) : 8 (NJ i + j) + 7].  (As compiler-visible values -- the builtin, or "+a" asm
// operands -- the register allocator carried some of them across the loop back-edge in VGPRs: hundreds of v_accvgpr
// copies per stage, spills, and no hazard handling after an asm MFMA.)  Every asm statement below names all 256 AGPRs as
// clobbered, so the compiler keeps nothing of its own in them; bot7_amd/build.py checks in the ISA of every build that no
// instruction outside these asm statements names an AGPR.
#define B7_W4_ACC_REGS \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
  "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", \
  "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", \
  "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", \
  "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", \
  "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", \
  "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", \
  "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", \
  "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", \
  "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
  "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", \
  "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", \
  "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", \
  "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", \
  "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", \
  "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", \
  "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", \
  "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", \
  "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

template <int NJ, int I, int J, bool ZERO>
__device__ __forceinline__ void mfma_tile(double a, double b) {
  constexpr int lo = 8 * (NJ * I + J);
  if constexpr (ZERO)
    asm volatile("v_mfma_f64_16x16x4_f64 a[%2:%3], %0, %1, 0" ::"v"(a), "v"(b), "n"(lo), "n"(lo + 7) : B7_W4_ACC_REGS);
  else
    asm volatile("v_mfma_f64_16x16x4_f64 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(lo), "n"(lo + 7)
                 : B7_W4_ACC_REGS);
}

// accumulator tile (I, J) back into VGPRs: the caller has drained the matrix pipe (s_nop) before the first one
template <int NJ, int I, int J>
__device__ __forceinline__ void read_tile(double (&v)[4]) {
  constexpr int lo = 8 * (NJ * I + J);
  unsigned w[8];
  asm volatile(
      "v_accvgpr_read_b32 %0, a[%8]\n\tv_accvgpr_read_b32 %1, a[%9]\n\tv_accvgpr_read_b32 %2, a[%10]\n\t"
      "v_accvgpr_read_b32 %3, a[%11]\n\tv_accvgpr_read_b32 %4, a[%12]\n\tv_accvgpr_read_b32 %5, a[%13]\n\t"
      "v_accvgpr_read_b32 %6, a[%14]\n\tv_accvgpr_read_b32 %7, a[%15]"
      : "=v"(w[0]), "=v"(w[1]), "=v"(w[2]), "=v"(w[3]), "=v"(w[4]), "=v"(w[5]), "=v"(w[6]), "=v"(w[7])
      : "n"(lo), "n"(lo + 1), "n"(lo + 2), "n"(lo + 3), "n"(lo + 4), "n"(lo + 5), "n"(lo + 6), "n"(lo + 7));
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = __hiloint2double((int)w[2 * r + 1], (int)w[2 * r]);
}

template <int NJ>
struct Frags {
  double a[2][8], b[2][NJ];  // two k-steps: 8 strips of L^-1 rows, NJ strips of candidates
};
template <int NJ>
struct Stage {  // global data of one stage in flight: 4 + 2 NJ sixteen-byte chunks per thread
  d2_t c[4 + 2 * NJ];
};
struct Cursor {  // (n-tile, 16-deep k block) of a stage; clamps at the last stage
  int t, kb, ntiles;
  __device__ __forceinline__ void advance() {
    if (kb + 1 < (t + 1) * (BM / BK)) {
      ++kb;
    } else if (t + 1 < ntiles) {
      ++t;
      kb = 0;
    }
  }
};

template <int N, class F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// fragment pair P of a two-k-step set: P < NJ candidates strip P, else rows strip P - NJ (k-steps S0, S0 + 1)
template <int NJ, int P, int S0>
__device__ __forceinline__ void read_pair(Frags<NJ> &f, const double *__restrict__ sa, const double *__restrict__ sb) {
  if constexpr (P < NJ) {
    f.b[0][P] = sb[P * 16 * LD + 4 * S0];
    f.b[1][P] = sb[P * 16 * LD + 4 * S0 + 4];
  } else {
    f.a[0][P - NJ] = sa[(P - NJ) * 16 * LD + 4 * S0];
    f.a[1][P - NJ] = sa[(P - NJ) * 16 * LD + 4 * S0 + 4];
  }
}

// chunk C of a stage: C < 4 rows r0 + 32 C of the L^-1 tile, else candidates r0 + 32 (C - 4) of the K* tile
template <int NJ, int C>
__device__ __forceinline__ void load_chunk(Stage<NJ> &g, const double *__restrict__ qa, const double *__restrict__ qb,
                                           int64_t lda) {
  if constexpr (C < 4)
    g.c[C] = *reinterpret_cast<const d2_t *>(qa + (int64_t)(32 * C) * lda);
  else
    g.c[C] = *reinterpret_cast<const d2_t *>(qb + (int64_t)(32 * (C - 4)) * lda);
}
template <int NJ, int C>
__device__ __forceinline__ void store_chunk(const Stage<NJ> &g, double *__restrict__ w) {
  constexpr int o = C < 4 ? 32 * C * LD : A_DBL + 32 * (C - 4) * LD;
  w[o] = g.c[C][0];
  w[o + 1] = g.c[C][1];
}

// One stage of the stream.  FIRST = first 16-row strip with anything to do (0 away from the diagonal block); ZERO: the
// first stage of an n-tile starts the accumulators from the literal 0.  NM MFMAs per half; the side operations of a half
// (first: the fragment pairs of F1, the LDS stores of the next stage, the global loads of the one after; second: the
// fragment pairs of the next stage's F0) are spread evenly between them.
template <int NJ, int FIRST, bool ZERO>
__device__ __forceinline__ void stage(Frags<NJ> &f0, Frags<NJ> &f1, Stage<NJ> &g, double *__restrict__ cur,
                                      double *__restrict__ nxt, int fa, int fb, int wofs, const double *__restrict__ pa,
                                      const double *__restrict__ pb, int64_t lda, Cursor &ld) {
  constexpr int NI = 8 - FIRST, NM = 2 * NJ * NI, NP = 8 + NJ, NC = 4 + 2 * NJ, NS = NP + 2 * NC;
  const double *qa = pa + ((int64_t)ld.t * BM) * lda + ld.kb * BK, *qb = pb + ld.kb * BK;
  ld.advance();
  // ---- first half: k-steps 0, 1 from F0
  constexpr int K1 = (NS + NM - 1) / NM + 1, K2 = (NP + NM - 1) / NM + 1;  // side operations behind one MFMA, at most
  static_for<NM>([&](auto m_) {
    constexpr int m = decltype(m_)::value, ks = m / (NJ * NI), i = FIRST + (m % (NJ * NI)) / NJ, j = m % NJ;
    mfma_tile<NJ, i, j, ZERO && ks == 0>(f0.a[ks][i], f0.b[ks][j]);
    static_for<K1>([&](auto k_) {
      constexpr int sidx = m * NS / NM + decltype(k_)::value;
      if constexpr (sidx < (m + 1) * NS / NM) {
        if constexpr (sidx < NP) {
          if constexpr (sidx < NJ || sidx - NJ >= FIRST) read_pair<NJ, sidx, 2>(f1, cur + fa, cur + fb);
        } else if constexpr (sidx < NP + NC) {
          store_chunk<NJ, sidx - NP>(g, nxt + wofs);
        } else {
          load_chunk<NJ, sidx - NP - NC>(g, qa, qb, lda);
        }
      }
    });
  });
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's part of the next image is in LDS (the loads stay in flight)
  __builtin_amdgcn_s_barrier();
  // ---- second half: k-steps 2, 3 from F1; the next stage's F0 (all strips: its FIRST is not known here)
  static_for<NM>([&](auto m_) {
    constexpr int m = decltype(m_)::value, ks = m / (NJ * NI), i = FIRST + (m % (NJ * NI)) / NJ, j = m % NJ;
    mfma_tile<NJ, i, j, false>(f1.a[ks][i], f1.b[ks][j]);
    static_for<K2>([&](auto k_) {
      constexpr int sidx = m * NP / NM + decltype(k_)::value;
      if constexpr (sidx < (m + 1) * NP / NM) read_pair<NJ, sidx, 0>(f0, nxt + fa, nxt + fb);
    });
  });
}

template <int NJ>
__global__ void __launch_bounds__(256)
    post_kernel_w4(const double *__restrict__ Linv, const double *__restrict__ ks, int Npad, int64_t row0, int64_t Mtotal,
                   double base, double sgn, double var_add, int clamp, double var_min, double *__restrict__ var,
                   PostBatch bat) {
  constexpr int BN = 64 * NJ, STAGE_DBL = (BM + BN) * LD, NC = 4 + 2 * NJ, NP = 8 + NJ;
  extern __shared__ __align__(16) double sm[];
  if (bat.base) {
    const int64_t y = blockIdx.y;
    Linv += y * bat.sLinv;
    ks += y * bat.sks;
    var += y * bat.svar;
    base = bat.base[y];
    if (bat.var_add) var_add = bat.var_add[y];
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef B7_POST_STAMPS
  const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
  const int64_t lda = Npad;
  // staging: thread -> (row r0 + 32 i, 16-byte chunk kc) of both operand tiles
  const int r0 = threadIdx.x >> 3, kc = threadIdx.x & 7;
  const double *pa = Linv + (int64_t)r0 * lda + 2 * kc;
  const double *pb = ks + ((int64_t)blockIdx.x * BN + r0) * lda + 2 * kc;
  const int wofs = r0 * LD + 2 * kc;
  // fragments: lane -> (row or candidate lane & 15 of a strip, k = 4 s + (lane >> 4))
  const int fa = (lane & 15) * LD + (lane >> 4);
  const int fb = A_DBL + (wave * 16 * NJ + (lane & 15)) * LD + (lane >> 4);

  const int ntiles = Npad / BM;
  Cursor ld{0, 0, ntiles};
  Stage<NJ> g;
  Frags<NJ> f0, f1;
  double *const b0 = sm, *const b1 = sm + STAGE_DBL;  // every tile has an even number of stages: it starts on b0
  static_for<NC>([&](auto c_) { load_chunk<NJ, decltype(c_)::value>(g, pa, pb, lda); });  // stage 0
  ld.advance();
  static_for<NC>([&](auto c_) { store_chunk<NJ, decltype(c_)::value>(g, b0 + wofs); });
  {
    const double *qa = pa + ((int64_t)ld.t * BM) * lda + ld.kb * BK, *qb = pb + ld.kb * BK;
    static_for<NC>([&](auto c_) { load_chunk<NJ, decltype(c_)::value>(g, qa, qb, lda); });  // stage 1: stored during stage 0
    ld.advance();
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  static_for<NP>([&](auto p_) { read_pair<NJ, decltype(p_)::value, 0>(f0, b0 + fa, b0 + fb); });

  double colss[2][NJ] = {};  // rows 0..63 and 64..127 of every n-tile apart
#define B7_STAGE(F, Z, CUR, NXT) stage<NJ, F, Z>(f0, f1, g, CUR, NXT, fa, fb, wofs, pa, pb, lda, ld)
  for (int t = 0; t < ntiles; ++t) {
    // 8 t stages away from the diagonal block, then its 8 stages (in stage q the strips 0..q-1 hold zeros only); the
    // tile's first stage starts the accumulators from zero; buffers alternate, b0 first
    B7_STAGE(0, true, b0, b1);
    if (t > 0) {
      for (int p = 0; p < 4 * t - 1; ++p) {
        B7_STAGE(0, false, b1, b0);
        B7_STAGE(0, false, b0, b1);
      }
      B7_STAGE(0, false, b1, b0);
      B7_STAGE(0, false, b0, b1);
    }
    B7_STAGE(1, false, b1, b0);
    B7_STAGE(2, false, b0, b1);
    B7_STAGE(3, false, b1, b0);
    B7_STAGE(4, false, b0, b1);
    B7_STAGE(5, false, b1, b0);
    B7_STAGE(6, false, b0, b1);
    B7_STAGE(7, false, b1, b0);
    // the asm MFMAs are opaque to the compiler's hazard recogniser: drain the matrix pipe before VALU reads the tile
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    static_for<2 * NJ>([&](auto hj_) {
      constexpr int h = decltype(hj_)::value / NJ, j = decltype(hj_)::value % NJ;
      double s = 0.0;
      static_for<4>([&](auto i_) {
        double v[4];
        read_tile<NJ, 4 * h + decltype(i_)::value, j>(v);
#pragma unroll
        for (int r = 0; r < 4; ++r) s = __builtin_fma(v[r], v[r], s);  // explicit: the order is part of the result
      });
      colss[h][j] += s;
    });
  }
#undef B7_STAGE

  // lanes l, l^16, l^32, l^48 hold partial sums of the same candidate; then the two row halves
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    double v0 = colss[0][j], v1 = colss[1][j];
    v0 += __shfl_xor(v0, 16);
    v0 += __shfl_xor(v0, 32);
    v1 += __shfl_xor(v1, 16);
    v1 += __shfl_xor(v1, 32);
    const int64_t gidx = row0 + (int64_t)blockIdx.x * BN + wave * 16 * NJ + j * 16 + lane;
    if (lane < 16 && gidx < Mtotal) {
      double ss = v0;
      ss += v1;
      double v = (base + sgn * ss) + var_add;  // GP: amp - ss; Bayesian-linear head: 1/beta + ss
      if (clamp) v = (v < var_min) ? var_min : v;  // TH clamp: NaN passes through
      var[gidx] = v;
    }
  }
#ifdef B7_POST_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 4096) {
    const unsigned long long st1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    unsigned long long *o = b7_post_stamps + 4 * blockIdx.x;
    o[0] = st0, o[1] = sr0, o[2] = st1, o[3] = sr1;
  }
#endif
}

template <int NJ>
constexpr int lds_bytes() { return 2 * (BM + 64 * NJ) * LD * 8; }

// ---- the tall shape: 256 rows of L^-1 x 128 candidates per workgroup ------------------------------------------------------
// Same stream, same 32 accumulator tiles per wave (16 strips of rows x 2 strips of candidates), but an n-tile of 256 rows:
// the K* rows of a workgroup are walked Npad/256 times instead of Npad/128, which halves the K* bytes the kernel pulls
// through the L2 (the traffic the counters price at 8.6x the algorithmic bytes) at the price of reading L^-1 twice as
// often from the L2 it is resident in.  18 fragments per k-step do not fit twice over as two-k-step sets in 256 VGPRs,
// so the fragment sets hold ONE k-step each and alternate per k-step; the barrier moves to three quarters of the stage:
//   k-step 0: MFMAs(set 0) | read set 1 = k-step 1 | first half of the LDS stores of the next stage
//   k-step 1: MFMAs(set 1) | read set 0 = k-step 2 | second half of the stores
//   k-step 2: MFMAs(set 0) | read set 1 = k-step 3 | global loads of the stage after next
//   barrier
//   k-step 3: MFMAs(set 1) | read set 0 = k-step 0 of the next stage
// The fold runs over the two 128-row halves of the tile in the order of the 128-row kernels, so the bits are the same.
template <int NJ, int NR>
struct Tall {
  static constexpr int BMT = 16 * NR, BN = 64 * NJ, A_DBLT = BMT * LD, STAGE_DBL = (BMT + BN) * LD;
  static constexpr int NCA = BMT / 32, NC = NCA + 2 * NJ, NP = NR + NJ;
  static constexpr int LDS_BYTES = 2 * STAGE_DBL * 8;
  static_assert(NJ * NR == 32, "32 accumulator tiles per wave");

  struct Set {
    double a[NR], b[NJ];
  };
  struct Data {
    d2_t c[NC];
  };
  struct Cursor {
    int t, kb, ntiles;
    __device__ __forceinline__ void advance() {
      if (kb + 1 < (t + 1) * (BMT / BK)) {
        ++kb;
      } else if (t + 1 < ntiles) {
        ++t;
        kb = 0;
      }
    }
  };

  // fragment P of k-step S: P < NJ candidates strip P, else rows strip P - NJ
  template <int P, int S>
  static __device__ __forceinline__ void read_one(Set &f, const double *__restrict__ sa, const double *__restrict__ sb) {
    if constexpr (P < NJ)
      f.b[P] = sb[P * 16 * LD + 4 * S];
    else
      f.a[P - NJ] = sa[(P - NJ) * 16 * LD + 4 * S];
  }
  template <int C>
  static __device__ __forceinline__ void load_chunk(Data &g, const double *__restrict__ qa, const double *__restrict__ qb,
                                                    int64_t lda) {
    if constexpr (C < NCA)
      g.c[C] = *reinterpret_cast<const d2_t *>(qa + (int64_t)(32 * C) * lda);
    else
      g.c[C] = *reinterpret_cast<const d2_t *>(qb + (int64_t)(32 * (C - NCA)) * lda);
  }
  template <int C>
  static __device__ __forceinline__ void store_chunk(const Data &g, double *__restrict__ w) {
    constexpr int o = C < NCA ? 32 * C * LD : A_DBLT + 32 * (C - NCA) * LD;
    w[o] = g.c[C][0];
    w[o + 1] = g.c[C][1];
  }

  // one k-step: NJ * (NR - FIRST) MFMAs from set `use`, NSIDE side operations spread between them
  template <int FIRST, bool ZERO, int NSIDE, class Side>
  static __device__ __forceinline__ void kstep(const Set &use, Side &&side) {
    constexpr int NI = NR - FIRST, NM = NJ * NI;
    constexpr int KMAX = (NSIDE + NM - 1) / NM + 1;  // side operations behind one MFMA: [m NSIDE / NM, (m + 1) NSIDE / NM)
    static_for<NM>([&](auto m_) {
      constexpr int m = decltype(m_)::value, i = FIRST + m / NJ, j = m % NJ;
      mfma_tile<NJ, i, j, ZERO>(use.a[i], use.b[j]);
      static_for<KMAX>([&](auto k_) {
        constexpr int sidx = m * NSIDE / NM + decltype(k_)::value;
        if constexpr (sidx < (m + 1) * NSIDE / NM) side(std::integral_constant<int, sidx>{});
      });
    });
  }

  // FIRST / NEXT_FIRST: first strip with anything to do in this stage / in the next one
  template <int FIRST, int NEXT_FIRST, bool ZERO>
  static __device__ __forceinline__ void stage(Set &s0, Set &s1, Data &g, double *__restrict__ cur, double *__restrict__ nxt,
                                               int fa, int fb, int wofs, const double *__restrict__ pa,
                                               const double *__restrict__ pb, int64_t lda, Cursor &ld) {
    constexpr int H = NC / 2;
    const double *qa = pa + ((int64_t)ld.t * BMT) * lda + ld.kb * BK, *qb = pb + ld.kb * BK;
    ld.advance();
    kstep<FIRST, ZERO, NP + H>(s0, [&](auto s_) {
      constexpr int x = decltype(s_)::value;
      if constexpr (x < NP) {
        if constexpr (x < NJ || x - NJ >= FIRST) read_one<x, 1>(s1, cur + fa, cur + fb);
      } else {
        store_chunk<x - NP>(g, nxt + wofs);
      }
    });
    kstep<FIRST, false, NP + NC - H>(s1, [&](auto s_) {
      constexpr int x = decltype(s_)::value;
      if constexpr (x < NP) {
        if constexpr (x < NJ || x - NJ >= FIRST) read_one<x, 2>(s0, cur + fa, cur + fb);
      } else {
        store_chunk<H + x - NP>(g, nxt + wofs);
      }
    });
    kstep<FIRST, false, NP + NC>(s0, [&](auto s_) {
      constexpr int x = decltype(s_)::value;
      if constexpr (x < NP) {
        if constexpr (x < NJ || x - NJ >= FIRST) read_one<x, 3>(s1, cur + fa, cur + fb);
      } else {
        load_chunk<x - NP>(g, qa, qb, lda);
      }
    });
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's part of the next image is in LDS (the loads stay in flight)
    __builtin_amdgcn_s_barrier();
    kstep<FIRST, false, NP>(s1, [&](auto s_) {
      constexpr int x = decltype(s_)::value;
      if constexpr (x < NJ || x - NJ >= NEXT_FIRST) read_one<x, 0>(s0, nxt + fa, nxt + fb);
    });
  }
};

template <int NJ, int NR>
__global__ void __launch_bounds__(256)
    post_kernel_w4t(const double *__restrict__ Linv, const double *__restrict__ ks, int Npad, int64_t row0, int64_t Mtotal,
                    double base, double sgn, double var_add, int clamp, double var_min, double *__restrict__ var,
                    PostBatch bat) {
  using T = Tall<NJ, NR>;
  constexpr int BMT = T::BMT, BN = T::BN, NC = T::NC, NP = T::NP;
  extern __shared__ __align__(16) double sm[];
  if (bat.base) {
    const int64_t y = blockIdx.y;
    Linv += y * bat.sLinv;
    ks += y * bat.sks;
    var += y * bat.svar;
    base = bat.base[y];
    if (bat.var_add) var_add = bat.var_add[y];
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef B7_POST_STAMPS
  const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
  const int64_t lda = Npad;
  const int r0 = threadIdx.x >> 3, kc = threadIdx.x & 7;
  const double *pa = Linv + (int64_t)r0 * lda + 2 * kc;
  const double *pb = ks + ((int64_t)blockIdx.x * BN + r0) * lda + 2 * kc;
  const int wofs = r0 * LD + 2 * kc;
  const int fa = (lane & 15) * LD + (lane >> 4);
  const int fb = T::A_DBLT + (wave * 16 * NJ + (lane & 15)) * LD + (lane >> 4);

  const int ntiles = Npad / BMT;
  typename T::Cursor ld{0, 0, ntiles};
  typename T::Data g;
  typename T::Set s0, s1;
  double *const b0 = sm, *const b1 = sm + T::STAGE_DBL;
  static_for<NC>([&](auto c_) { T::template load_chunk<decltype(c_)::value>(g, pa, pb, lda); });  // stage 0
  ld.advance();
  static_for<NC>([&](auto c_) { T::template store_chunk<decltype(c_)::value>(g, b0 + wofs); });
  {
    const double *qa = pa + ((int64_t)ld.t * BMT) * lda + ld.kb * BK, *qb = pb + ld.kb * BK;
    static_for<NC>([&](auto c_) { T::template load_chunk<decltype(c_)::value>(g, qa, qb, lda); });  // stage 1
    ld.advance();
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  static_for<NP>([&](auto p_) { T::template read_one<decltype(p_)::value, 0>(s0, b0 + fa, b0 + fb); });

  double colss[2][NJ] = {};
#define B7_STAGE(F, NF, Z, CUR, NXT) T::template stage<F, NF, Z>(s0, s1, g, CUR, NXT, fa, fb, wofs, pa, pb, lda, ld)
  for (int t = 0; t < ntiles; ++t) {
    B7_STAGE(0, 0, true, b0, b1);
    if (t > 0) {
      for (int p = 0; p < (NR / 2) * t - 1; ++p) {
        B7_STAGE(0, 0, false, b1, b0);
        B7_STAGE(0, 0, false, b0, b1);
      }
      B7_STAGE(0, 0, false, b1, b0);
      B7_STAGE(0, 1, false, b0, b1);  // diagonal stage 0
    }
    // diagonal stages 1 .. NR-1 (for t == 0 the zeroing stage above was diagonal stage 0; it read the next set in full)
    static_for<NR - 1>([&](auto q_) {
      constexpr int Q = decltype(q_)::value + 1;
      if constexpr (Q % 2 == 1)
        B7_STAGE(Q, (Q + 1) % NR, false, b1, b0);
      else
        B7_STAGE(Q, (Q + 1) % NR, false, b0, b1);
    });
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    // the tile is NR/8 128-row tiles of the fold, each as (rows 0..63, rows 64..127), in ascending order
    static_for<(NR / 4) * NJ>([&](auto qj_) {
      constexpr int q = decltype(qj_)::value / NJ, j = decltype(qj_)::value % NJ;  // q: 64-row group of the tile
      double s = 0.0;
      static_for<4>([&](auto i_) {
        double v[4];
        read_tile<NJ, 4 * q + decltype(i_)::value, j>(v);
#pragma unroll
        for (int r = 0; r < 4; ++r) s = __builtin_fma(v[r], v[r], s);
      });
      colss[q & 1][j] += s;
    });
  }
#undef B7_STAGE

#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    double v0 = colss[0][j], v1 = colss[1][j];
    v0 += __shfl_xor(v0, 16);
    v0 += __shfl_xor(v0, 32);
    v1 += __shfl_xor(v1, 16);
    v1 += __shfl_xor(v1, 32);
    const int64_t gidx = row0 + (int64_t)blockIdx.x * BN + wave * 16 * NJ + j * 16 + lane;
    if (lane < 16 && gidx < Mtotal) {
      double ss = v0;
      ss += v1;
      double v = (base + sgn * ss) + var_add;
      if (clamp) v = (v < var_min) ? var_min : v;
      var[gidx] = v;
    }
  }
#ifdef B7_POST_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 4096) {
    const unsigned long long st1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    unsigned long long *o = b7_post_stamps + 4 * blockIdx.x;
    o[0] = st0, o[1] = sr0, o[2] = st1, o[3] = sr1;
  }
#endif
}
}  // namespace w4

// ---- one 64-block: N <= 64 observations (the first 64 trials of every run at the reference's defaults) or the 50 basis
// features of the DNGO head.  A persistent workgroup (two per CU) reads the A fragments of L^-1 (64 x 64) into registers once
// and walks tiles of 64 candidates -- their K* rows staged through LDS, the next tile's in flight meanwhile --, a wave 16 of
// them: v = L^-1 K*' as four accumulator tiles, every chain over k ascending up to the row tile's diagonal (what is above is
// zero), then per candidate the sum of squares over the 64 rows in a fixed order: per lane group g the fma chain over rows
// 16 I + g + 4 r (I, then r, ascending), ((g0 + g1) + (g2 + g3)).
// Nothing here depends on the grid size, the shard or the launch shape, as in the large kernels.
namespace small64 {
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int NB = 64, LLD = NB + 2, BN = 64;  // 64 candidates per tile: ~200 registers, two workgroups per CU -- one loads while the other computes
constexpr int LDS_BYTES = BN * LLD * 8;  // the K* tile; L^-1 lives in registers

__global__ void __launch_bounds__(256)
    post_small_kernel(const double *__restrict__ Linv, const double *__restrict__ ks, int64_t row0, int64_t Mtotal, int ntiles,
                      double base, double sgn, double var_add, int clamp, double var_min, double *__restrict__ var, PostBatch bat) {
  extern __shared__ __align__(16) double sm[];
  if (bat.base) {
    const int64_t y = blockIdx.y;
    Linv += y * bat.sLinv;
    ks += y * bat.sks;
    var += y * bat.svar;
    base = bat.base[y];
    if (bat.var_add) var_add = bat.var_add[y];
  }
  double *Ks = sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  // a workgroup walks tiles blockIdx.x, + gridDim.x, ...; the K* rows of the NEXT tile are on their way (16 chunks of 16 bytes
  // per thread, a wave reading two whole 512-byte rows per instruction) while this one is computed
  d2_t pv[BN / 8];
  auto fetch = [&](int tile) {
    const double *kt = ks + ((int64_t)tile * BN) * NB;
#pragma unroll
    for (int i = 0; i < BN / 8; ++i) {
      const int e = tid + 256 * i;
      pv[i] = *reinterpret_cast<const d2_t *>(kt + (int64_t)(e >> 5) * NB + 2 * (e & 31));
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  // L^-1 is the same for every tile: its A fragments -- row tile I, k-steps 0 .. 4 I + 3 (nothing above the diagonal) -- are
  // read from memory ONCE into 40 registers per lane; per tile only the candidates' fragments come from LDS, sixteen per
  // candidate strip, and the MFMA chains run from registers
  double af[40];
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int k4 = 0; k4 < 4 * (I + 1); ++k4) af[2 * I * (I + 1) + k4] = Linv[(int64_t)(I * 16 + lr) * NB + 4 * k4 + lq];
  // every load so far has landed before the loop is entered: with the 40 fragment loads still counted as outstanding at the
  // loop head, the compiler's (counter-based) waits for them sit in the middle of the MFMA section and, from the second
  // iteration on, wait for the PREFETCH instead -- load and compute times added up (41 us for 1564 tiles) instead of overlapping
  // (as empty asm statements that "use" the fragments: a plain s_waitcnt does not keep the optimiser from sinking the loads
  // below it, and only a wait the compiler inserted itself clears its bookkeeping)
#pragma unroll
  for (int g = 0; g < 4; ++g)
    asm volatile("" : "+v"(af[10 * g]), "+v"(af[10 * g + 1]), "+v"(af[10 * g + 2]), "+v"(af[10 * g + 3]), "+v"(af[10 * g + 4]),
                      "+v"(af[10 * g + 5]), "+v"(af[10 * g + 6]), "+v"(af[10 * g + 7]), "+v"(af[10 * g + 8]), "+v"(af[10 * g + 9]));
  for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int i = 0; i < BN / 8; ++i) {
      const int e = tid + 256 * i, r = e >> 5, c2 = 2 * (e & 31);
      Ks[r * LLD + c2] = pv[i][0];
      Ks[r * LLD + c2 + 1] = pv[i][1];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
    __builtin_amdgcn_sched_barrier(0);  // the loads go out HERE: left to itself the scheduler sinks them below the MFMA section
                                        // to shorten 64 registers' live range, and load and compute times add instead of overlapping
    constexpr int NJ = BN / 64;  // 16-candidate strips per wave
    double ss[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const double *bp = Ks + (wave * 16 * NJ + j * 16 + lr) * LLD + lq;
      double bf[16];
#pragma unroll
      for (int k4 = 0; k4 < 16; ++k4) bf[k4] = bp[4 * k4];
      double s = 0.0;
#pragma unroll
      for (int I = 0; I < 4; ++I) {
        d4_t c = {0.0, 0.0, 0.0, 0.0};
#ifdef B7_PS_ABLATE  // diagnostic build only (never defined for the shipped library): one MFMA per chain instead of 4 (I + 1)
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(af[2 * I * (I + 1)], bf[I], c, 0, 0, 0);
#else
#pragma unroll
        for (int k4 = 0; k4 < 4 * (I + 1); ++k4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(af[2 * I * (I + 1) + k4], bf[k4], c, 0, 0, 0);
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) s = __builtin_fma(c[r], c[r], s);
      }
      ss[j] = s;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      // lanes l, l ^ 16, l ^ 32, l ^ 48 hold the four lane groups' sums of one candidate: (g0 + g1) + (g2 + g3)
      double v = ss[j];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      const int64_t gidx = row0 + (int64_t)tile * BN + wave * 16 * NJ + j * 16 + lr;
      if (lane < 16 && gidx < Mtotal) {
        double o = (base + sgn * v) + var_add;
        if (clamp) o = (o < var_min) ? var_min : o;
        var[gidx] = o;
      }
    }
    __syncthreads();  // everybody is done with this tile's image
  }
}
}  // namespace small64

#ifndef B7_POST_NO_LAUNCHERS
namespace {
struct PostArgs {  // what one launch needs besides the kernel: S fits side by side (S = 1: the context's own fit)
  const double *Linv, *ks;
  double *var;
  double base, sgn, var_add;
  int S;
  PostBatch pb;
};

template <int NJ>
int launch_post_w4(b7_ctx *c, const PostArgs &a, int64_t row0, int64_t rows, int64_t Mtotal) {
  constexpr int lds = w4::lds_bytes<NJ>();
  B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(w4::post_kernel_w4<NJ>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(w4::post_kernel_w4<NJ>, dim3((unsigned)(rows / (64 * NJ)), a.S), dim3(256), lds, c->stream, a.Linv, a.ks,
                     c->Npad, row0, Mtotal, a.base, a.sgn, a.var_add, c->opts.var_clamp, c->opts.var_min, a.var, a.pb);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

template <int NJ, int NR>
int launch_post_tall(b7_ctx *c, const PostArgs &a, int64_t row0, int64_t rows, int64_t Mtotal) {
  using T = w4::Tall<NJ, NR>;
  auto kern = w4::post_kernel_w4t<NJ, NR>;
  B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                T::LDS_BYTES));
  hipLaunchKernelGGL(kern, dim3((unsigned)(rows / T::BN), a.S), dim3(256), T::LDS_BYTES, c->stream, a.Linv, a.ks, c->Npad, row0,
                     Mtotal, a.base, a.sgn, a.var_add, c->opts.var_clamp, c->opts.var_min, a.var, a.pb);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_post_small(b7_ctx *c, const PostArgs &a, int64_t row0, int64_t rows, int64_t Mtotal) {
  B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(small64::post_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                small64::LDS_BYTES));
  // two workgroups per CU (33 KB of LDS, ~200 registers each); with S fits side by side the CUs are split between them
  const int ntiles = (int)(rows / small64::BN), S = a.S > 0 ? a.S : 1;
  int gx = (2 * c->cus + S - 1) / S;
  if (gx > ntiles) gx = ntiles;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(small64::post_small_kernel, dim3((unsigned)gx, S), dim3(256), small64::LDS_BYTES, c->stream, a.Linv, a.ks, row0,
                     Mtotal, ntiles, a.base, a.sgn, a.var_add, c->opts.var_clamp, c->opts.var_min, a.var, a.pb);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int dispatch_post(b7_ctx *c, const PostArgs &a, int64_t row0, int64_t rows, int64_t Mtotal) {
  if (rows % B7_MROWS) return b7_fail(c, B7_ERR_INVALID, "post: rows %lld not a multiple of %d", (long long)rows, B7_MROWS);
  if (c->Npad == 64) return launch_post_small(c, a, row0, rows, Mtotal);
  const bool large = rows / 256 * a.S >= c->cus;  // at least one 256-candidate workgroup per CU
  // large grids: the tall shape (n-tiles of 256 rows: half the K* bytes through the L2) when the padded N allows it
  if (c->Npad % 256 == 0 && large) return launch_post_tall<2, 16>(c, a, row0, rows, Mtotal);
  // 256 candidates per workgroup when that still gives every CU one; otherwise 128 (same arithmetic, same bits)
  if (!large) return launch_post_w4<2>(c, a, row0, rows, Mtotal);
  return launch_post_w4<4>(c, a, row0, rows, Mtotal);
}
}  // namespace

int launch_post(b7_ctx *c, const double *ks, int64_t row0, int64_t rows, int64_t Mtotal, double *var) {
  PhaseScope ps(c, "post");
  const bool blr = c->model_kind == 1;
  PostArgs a{(const double *)c->Linv.p, ks, var, blr ? 0.0 : c->amp, blr ? 1.0 : -1.0,
             blr ? c->noise : (c->opts.var_with_noise ? c->noise : 0.0), 1, PostBatch{}};
  return dispatch_post(c, a, row0, rows, Mtotal);
}

// S GP fits over the same rows: Linv_s = Linv + s Npad^2, K*_s = ks + s sks, var_s = var + s svar; amp_s (and noise_s) on
// the device
int launch_post_batch(b7_ctx *c, int S, const double *Linv, const double *ks, int64_t sks, int64_t rows, int64_t Mtotal,
                      double *var, int64_t svar, const double *amp_dev, const double *noise_dev) {
  PhaseScope ps(c, "post");
  PostArgs a{Linv, ks, var, 0.0, -1.0, 0.0, S, PostBatch{}};
  a.pb.sLinv = (int64_t)c->Npad * c->Npad;
  a.pb.sks = sks;
  a.pb.svar = svar;
  a.pb.base = amp_dev;
  a.pb.var_add = c->opts.var_with_noise ? noise_dev : nullptr;
  return dispatch_post(c, a, 0, rows, Mtotal);
}

// S Bayesian-linear heads over the SAME feature rows (b7_blr_eval_nominate_marg): Linv_s = Linv + s Npad^2, the features are
// shared (stride 0), var_s = 1/beta_s + |Linv_s z|^2 with zero_dev[S] = 0 and invbeta_dev[S] on the device
int launch_post_heads(b7_ctx *c, int S, const double *Linv, const double *feat, int64_t rows, int64_t Mtotal, double *var, int64_t svar,
                      const double *zero_dev, const double *invbeta_dev) {
  PhaseScope ps(c, "post");
  PostArgs a{Linv, feat, var, 0.0, 1.0, 0.0, S, PostBatch{}};
  a.pb.sLinv = (int64_t)c->Npad * c->Npad;
  a.pb.sks = 0;
  a.pb.svar = svar;
  a.pb.base = zero_dev;
  a.pb.var_add = invbeta_dev;
  return dispatch_post(c, a, 0, rows, Mtotal);
}
#endif
