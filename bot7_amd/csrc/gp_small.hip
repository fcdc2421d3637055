// The GP fit for SMALL observation sets -- N <= 128, d <= 32, one response column: the reference's own regime (budget 100,
// bots/abstract.lua:64) -- in ONE workgroup of ONE launch per hyper vector.
//
// Reference arithmetic replaced, per hyper vector (bots/bayesopt.lua:68,73-76 -> model:sample_hypers / model:predict's first half):
//   K(X,X) + noise I (utils/math.lua:65-111), chol (utils/math.lua:159-218: the plain attempt; a failed pivot is reported and
//   the caller redoes that fit through the general path and its jitter schedule), and then either
//     MODE 0  the likelihood's two reductions, |L^-1 r|^2 and sum log L_ii  (one density evaluation of samplers/slice.lua:92-168)
//     MODE 1  L^-1 in full, alpha = L^-T L^-1 r and the scaled observations: everything the posterior kernels read
//             (b7_eval_nominate's S fits are S workgroups of one launch: observation scaling, K(X,X), its fix-up, two memsets,
//             the persistent factorisation and three triangular matrix-vector launches of the general schedule in one).
//
// Structure.  512 threads.  Waves 0..3 are the four waves b7diag::diag_core is written for; waves 4..7 keep in step with its
// barriers (diag_bystander).  While wave 0 runs block (0,0)'s four pivot chains, everybody who is idle -- waves 4..7, and waves
// 1..3 through the routine's hook -- assembles what does not depend on L11: the rest of K11, all of K21, the K entries of
// block (1,1) (kept in LDS until L21 L21' is there to be subtracted); while it runs block (1,1)'s, waves 4..7 do the
// likelihood's first solve and the update of the second residual.  Four 64 x 66 images in LDS change roles as the data die:
// B0 observations -> inv(L22);  B1 K11 / L11 -> K22 / L22;  B2 K21 -> L21 -> L21 inv(L11) -> inv(L)21;  B3 inv(L11).
// The kernel is latency all the way: ONE workgroup executes ~40 KB of straight-line code once, out of an instruction cache that
// is cold at every launch (a sub-tile's code costs 2 300 cycles the first time and 1 700 the second) -- hence one instance per
// block count (TWO) and one site of sub-tile code per caller (k_job) rather than one per use.
//
// Bits.  K entries: the chain of v_mfma_f64_16x16x4 over the input dimensions, the (c - xs/2) - zs/2 argument and the table
// exponential of ksx_kernel (ksx_exp.h).  L, L^-1: diag_core per 64-block; L21 = C inv(L11)' with the blocks above inv(L11)'s
// diagonal skipped; K22 - L21 L21' as the 64-deep chain from zero, then the subtraction (lower sub-tiles only);
// inv(L)21 = -inv(L22) (0 + L21 inv(L11)) with the two alternating accumulators of potrf_persist.hip's inv_job -- the
// persistent schedule's arithmetic, operation for operation.  alpha: the sums of trmv_lower_kernel / trmv_lower_t_part_kernel
// (two 64-block systems) or of potrf_small64_kernel (one).  tests/test_gpu_parity.py holds L, L^-1 and alpha against the
// general path BIT FOR BIT and the likelihood against nll_small_kernel's bits.
#include "b7_internal.h"
#include "ksx_exp.h"
#include "potrf_diag.h"

#ifdef B7_GS_STAMP
// Diagnostic build only (tools/gp_small_stamps.py; never defined for the shipped library): workgroup 0's waves 0 and 4 record
// s_memtime at their phase boundaries into a buffer nothing else reads.
__device__ unsigned long long b7_gs_stamps[2 * 32 + 32];
extern "C" int b7dbg_gs_stamps(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(b7_gs_stamps), sizeof(unsigned long long) * (2 * 32 + 32));
}
#define GS_STAMP(i)                                                                                          \
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) b7_gs_stamps[(threadIdx.x >> 8) * 32 + (i)] = __builtin_amdgcn_s_memtime()
#else
#define GS_STAMP(i)
#endif

namespace {
using namespace b7diag;  // NB = 64, DLD, TLD, diag_core, diag_bystander

__constant__ double exp2_tab_gs[128];  // b7_exp2_tab (ensure_gs_table)

constexpr int OLD = 33;         // row stride of the observation image [128][OLD] (32 columns, zero padded)
constexpr int BUF = NB * DLD;   // one 64 x 66 image
static_assert(128 * OLD <= BUF, "the observation image lives in one block image");
constexpr int GS_THREADS = 512;
constexpr int K22_STASH = 6;  // sub-tiles of block (1,1) whose K entries are formed ahead of time (all there is up to N = 112)
constexpr int GS_LDS_DOUBLES = 4 * BUF + 32 * TLD + 5 * 128 + 32 + 128 + 64 + K22_STASH * 256;
static_assert(GS_LDS_DOUBLES * 8 + 512 <= 160 * 1024, "LDS budget (the static arrays -- inf, the diagnostic build's stamps -- need < 512 B)");

struct GsInline {  // the hypers of a single evaluation, passed in the kernel arguments (no second trip over the bus)
  double v[35];
};

struct GsArgs {
  const double *xobs, *y;  // N x d raw observations, N responses
  int N, d, dpad, B;
  const double *hyp_mem;   // [B x d lengthscales | B amp | B noise | B mean], device-visible (mapped host memory is fine)
  int use_inline;
  int *info;               // 4 ints per fit (device memory), nullable
  int *report;             // the same into mapped host memory, nullable
  unsigned *done;          // completion word in mapped host memory (B == 1), nullable
  // MODE 0
  double *terms;           // 2 per fit: |L^-1 r|^2, sum log L_ii
  // MODE 1 (per-fit strides: w dpad, zsc npad dpad, zss npad, matrices npad^2, dinv npad 64, vectors npad)
  double *hyp_out;         // device copy of the pack for the kernels downstream, nullable
  double *w, *zsc, *zss, *L, *Linv, *dinv, *alpha, *resid;  // L, dinv, resid nullable
};

// every barrier of this kernel orders LDS traffic only: its global stores are results for LATER kernels and stay in flight
__device__ __forceinline__ void lds_barrier() { diag_barrier<true>(); }

// rows 0..15 x columns 48..63 of a block about to be factored: I_16 (potrf_diag.h: the right-hand side of the inversion)
__device__ __forceinline__ void identity_corner(double *A) {
  const int t = threadIdx.x;
  if (t < 256) A[(t >> 4) * DLD + 48 + (t & 15)] = ((t >> 4) == (t & 15)) ? 1.0 : 0.0;
}
__device__ __forceinline__ void zero_block(double *X) {
#pragma unroll
  for (int t = 0; t < (BUF + GS_THREADS - 1) / GS_THREADS; ++t) {
    const int e = threadIdx.x + GS_THREADS * t;
    if (e < BUF) X[e] = 0.0;
  }
}

// One 16 x 16 sub-tile (it, jt) of the 64 x 64 block (I0, J0) of K(X,X) + noise I, in the accumulator layout (v[r]: row
// 16 it + (lane >> 4) + 4 r, column 16 jt + (lane & 15)); rows / columns >= N are the identity.
// x (z .* w)' on MFMA exactly as ksx_kernel forms it: A fragments are the raw rows, B fragments the raw columns' rows times w
// (the product rounded once, as prep_obs_kernel rounds z .* w), a chain of v_mfma_f64_16x16x4 over the k-steps = the ascending
// fma chain over the (zero padded) dimensions; ks = dpad / 4 steps hold anything.  A sub-tile wholly in the padding is not
// computed.
__device__ __forceinline__ void k_tile_vals(const double *__restrict__ obs, const double (&wq)[8], const double *__restrict__ hn,
                                            const double *__restrict__ tab, int I0, int J0, int N, double noise, int it, int jt,
                                            int ks, double (&v)[4]) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const int gj = J0 + 16 * jt + lr;
  if (I0 + 16 * it >= N || J0 + 16 * jt >= N) {  // wave-uniform
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (I0 + 16 * it + lq + 4 * r == gj) ? 1.0 : 0.0;
    return;
  }
  d4_t c = {0.0, 0.0, 0.0, 0.0};
  const double *ap = obs + (I0 + 16 * it + lr) * OLD + lq, *bp = obs + gj * OLD + lq;
  // all sixteen fragment reads at once (the columns beyond dpad hold zeros); the chain runs over the k-steps that hold anything,
  // in three wave-uniform pieces (2, +2, +4) instead of a branch per step
  double af[8], bf[8];
#pragma unroll
  for (int k4 = 0; k4 < 8; ++k4) af[k4] = ap[4 * k4], bf[k4] = bp[4 * k4] * wq[k4];
  c = mfma_f64(af[0], bf[0], c);
  c = mfma_f64(af[1], bf[1], c);
  if (ks > 2) {
    c = mfma_f64(af[2], bf[2], c);
    c = mfma_f64(af[3], bf[3], c);
  }
  if (ks > 4) {
#pragma unroll
    for (int k4 = 4; k4 < 8; ++k4) c = mfma_f64(af[k4], bf[k4], c);
  }
  const double hj = hn[gj];
  double arg[4], kv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) arg[r] = (c[r] - hn[I0 + 16 * it + lq + 4 * r]) - hj;
  amp_exp_nonpos4(arg, tab, kv);
  // all four exponentials exist HERE, side by side: without this the optimiser sinks each of them into its own lane-divergent
  // "not padding" branch below and the four 14-deep chains run one after the other (a sub-tile took 2000 cycles, not 900)
  asm volatile("" : "+v"(kv[0]), "+v"(kv[1]), "+v"(kv[2]), "+v"(kv[3]));
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gi = I0 + 16 * it + lq + 4 * r;
    const bool pad = gi >= N || gj >= N, dgn = gi == gj;
    const double x = dgn ? kv[r] + noise : kv[r];
    v[r] = pad ? (dgn ? 1.0 : 0.0) : x;
  }
}
__device__ __forceinline__ void tile_put(double *__restrict__ T, int it, int jt, const double (&v)[4]) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) T[(16 * it + lq + 4 * r) * DLD + 16 * jt + lr] = v[r];
}
// the two chains of a wave's share of L21 = K21 inv(L11)' (column blocks C0 and C1 of its row strip): ap -> this lane's A
// fragments of K21 (k-step t at ap[4 t]), xp -> inv(L11)'s row lr, k = lq (column block jb's fragments at xp[16 jb DLD + 4 t]).
// All operands first, then the MFMAs: the chain never waits for LDS.
template <int C0, int C1>
__device__ __forceinline__ void l21_chains(const double *__restrict__ ap, const double *__restrict__ xp, d4_t (&l)[2]) {
  constexpr int N0 = 4 * (C0 + 1), N1 = 4 * (C1 + 1), NA = N0 > N1 ? N0 : N1;
  double aq[NA], x0[N0], x1[N1];
#pragma unroll
  for (int t = 0; t < NA; ++t) aq[t] = ap[4 * t];
#pragma unroll
  for (int t = 0; t < N0; ++t) x0[t] = xp[C0 * 16 * DLD + 4 * t];
#pragma unroll
  for (int t = 0; t < N1; ++t) x1[t] = xp[C1 * 16 * DLD + 4 * t];
  l[0] = d4_t{0.0, 0.0, 0.0, 0.0};
  l[1] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < NA; ++t) {
    if (t < N0) l[0] = mfma_f64(aq[t], x0[t], l[0]);
    if (t < N1) l[1] = mfma_f64(aq[t], x1[t], l[1]);
  }
}
// the two chains of a wave's share of L21 inv(L11) (column blocks C0, C1): ap -> A fragments of L21 (k-step t at ap[4 t]),
// xb -> inv(L11)[k = lq][column lr] (k-step t, column block jb at xb[4 t DLD + 16 jb]); k-steps t >= 4 jb only
template <int C0, int C1>
__device__ __forceinline__ void p_chains(const double *__restrict__ ap, const double *__restrict__ xb, d4_t (&pv)[2]) {
  constexpr int F0 = 4 * C0, F1 = 4 * C1, FA = F0 < F1 ? F0 : F1;
  double aq[16], x0[16], x1[16];
#pragma unroll
  for (int t = FA; t < 16; ++t) aq[t] = ap[4 * t];
#pragma unroll
  for (int t = F0; t < 16; ++t) x0[t] = xb[4 * t * DLD + 16 * C0];
#pragma unroll
  for (int t = F1; t < 16; ++t) x1[t] = xb[4 * t * DLD + 16 * C1];
#pragma unroll
  for (int t = FA; t < 16; ++t) {
    if (t >= F0) pv[0] = mfma_f64(aq[t], x0[t], pv[0]);
    if (t >= F1) pv[1] = mfma_f64(aq[t], x1[t], pv[1]);
  }
}
// q-th sub-tile of a diagonal block's lower triangle: (0,0) (1,0) (1,1) (2,0) ...
__device__ __forceinline__ void lower_tile(int q, int &it, int &jt) {
  it = (q >= 1) + (q >= 3) + (q >= 6);
  jt = q - ((it * (it + 1)) >> 1);
}

// a 64 x 64 image (row stride DLD) -> global rows of leading dimension ld; lower: entries above the diagonal become zero
__device__ __forceinline__ void store_block(const double *__restrict__ img, double *__restrict__ dst, int64_t ld, bool lower, int t0,
                                            int nt) {
  // two columns (16 bytes) per lane and instruction: a wave stores 1 KiB at a time (images and destinations are 16-byte
  // aligned: DLD and every leading dimension are even)
  for (int e = t0; e < NB * NB / 2; e += nt) {
    const int i = e >> 5, j = 2 * (e & 31);
    d2_t v = *reinterpret_cast<const d2_t *>(img + i * DLD + j);
    if (lower) {
      if (j > i) v[0] = 0.0;
      if (j + 1 > i) v[1] = 0.0;
    }
    *reinterpret_cast<d2_t *>(dst + (int64_t)i * ld + j) = v;
  }
}
__device__ __forceinline__ void store_zero_block(double *__restrict__ dst, int64_t ld, int t0, int nt) {
  const d2_t zero = {0.0, 0.0};
  for (int e = t0; e < NB * NB / 2; e += nt) *reinterpret_cast<d2_t *>(dst + (int64_t)(e >> 5) * ld + 2 * (e & 31)) = zero;
}

// TWO: N > 64 (two 64-blocks).  A template parameter, not a run-time test: the kernel runs every instruction once, out of a cold
// instruction cache, and each block of code it has to jump over is a fetch from memory (a one-block evaluation that carried
// the two-block phases as untaken branches was 0.9 us slower).
template <int MODE, bool TWO>
__global__ void __launch_bounds__(GS_THREADS) gp_small_kernel(GsArgs a, GsInline hin) {
  extern __shared__ __align__(16) double sm[];
  __shared__ int inf[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15,
            lq = lane >> 4;
  const int N = a.N, d = a.d, B = a.B;
  constexpr int npad = TWO ? 128 : 64;
  constexpr bool two = TWO;
  const bool aux = wave >= 4;
  double *B0 = sm, *B1 = B0 + BUF, *B2 = B1 + BUF, *B3 = B2 + BUF, *T = B3 + BUF;
  double *r = T + 32 * TLD;   // [128] residual y - mean
  double *hn = r + 128;       // [128] half norms
  double *z = hn + 128;       // [128] MODE 0: L^-1 r
  double *dg = z + 128;       // [128] diagonal of L
  double *tv = dg + 128;      // [128] MODE 1: t = L^-1 r in trmv_lower_kernel's order
  double *w = tv + 128;       // [32]
  double *tab = w + 32;       // [128] amp 2^(j/128)
  double *red = tab + 128;    // [64]
  double *ks22 = red + 64;    // [K22_STASH][4][64] K entries of block (1,1)'s first sub-tiles, accumulator layout
  double *obs = B0;
  const double *hyp = a.use_inline ? hin.v : a.hyp_mem;
  const double *ls = hyp + (size_t)b * d;
  const double amp = hyp[(size_t)B * d + b], noise = hyp[(size_t)B * (d + 1) + b], mean = hyp[(size_t)B * (d + 2) + b];
  GS_STAMP(0);
  if (tid < 4) inf[tid] = 0;
  double lsv = 0.0;  // this lane's lengthscale (a batch's hypers sit in mapped HOST memory: every read is a trip over PCIe)
  if (tid < 32) {
    if (tid < d) lsv = ls[tid];
    w[tid] = tid < d ? 1.0 / lsv : 0.0;  // inv_ls = ones:cdiv(lenscale), utils/math.lua:72
  }
  if (tid < 128) {
    tab[tid] = amp * exp2_tab_gs[tid];
    r[tid] = tid < N ? a.y[tid] - mean : 0.0;
  }
  {
    // the N x d observations are one contiguous block: eight coalesced loads per thread, all in flight at once, then the
    // scatter into the zero-padded [128][OLD] image
    double v[8];
    const int total = N * d;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int idx = tid + GS_THREADS * t;
      v[t] = idx < total ? a.xobs[idx] : 0.0;
    }
    for (int e = tid; e < 128 * OLD; e += GS_THREADS) obs[e] = 0.0;
    zero_block(B3);       // the inverse's image and the identity corner of block (0,0) (rows 0..15 x columns 48..63: sub-tile
    identity_corner(B1);  // (0,3), which no K sub-tile writes): nothing touches them before the factor routine -- done here,
    lds_barrier();        // while the loads are in flight
    GS_STAMP(1);
    const float rd = 1.0f / (float)d;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int idx = tid + GS_THREADS * t;
      if (idx < total) {
        int i = (int)((float)idx * rd);  // idx / d for idx < 4096, d <= 32: the estimate is off by at most one
        i += (i + 1) * d <= idx;
        i -= i * d > idx;
        obs[i * OLD + (idx - i * d)] = v[t];
      }
    }
  }
  lds_barrier();
  if (tid < 128) {
    double s = 0.0;
    for (int k = 0; k < a.dpad; k += 4) {  // the columns beyond dpad add (0 * 0) * 0; dpad is a multiple of 4: four reads in flight
      const double x0 = obs[tid * OLD + k], x1 = obs[tid * OLD + k + 1], x2 = obs[tid * OLD + k + 2], x3 = obs[tid * OLD + k + 3];
      const double w0 = w[k], w1 = w[k + 1], w2 = w[k + 2], w3 = w[k + 3];
      s += (x0 * x0) * w0;  // Z_ss = (Z.^2) * inv_ls, :79
      s += (x1 * x1) * w1;
      s += (x2 * x2) * w2;
      s += (x3 * x3) * w3;
    }
    hn[tid] = 0.5 * s;
  }
  if (MODE == 1) {
    // what the posterior kernels read of the observations, as prep_obs_kernel leaves it: z .* w (zero padded), the weights
    // (the half norms follow below, once they are in LDS), and this fit's hypers where the kernels downstream find them
    const int dpad = a.dpad;
    double *zo = a.zsc + (size_t)b * npad * dpad;
    const int dsh = __builtin_ctz(dpad);  // dpad is 4, 8, 16 or 32
    for (int e = tid; e < npad * dpad; e += GS_THREADS) {
      const int i = e >> dsh, k = e & (dpad - 1);
      zo[e] = obs[i * OLD + k] * w[k];
    }
    if (tid < dpad) a.w[(size_t)b * dpad + tid] = w[tid];
    if (a.hyp_out && tid < d + 3) {  // (from the registers that hold them already, not read a second time)
      const size_t at = tid < d ? (size_t)b * d + tid : (size_t)B * (d + (tid - d)) + b;
      a.hyp_out[at] = tid < d ? lsv : (tid == d ? amp : (tid == d + 1 ? noise : mean));
    }
  }
  lds_barrier();
  if (MODE == 1 && tid < npad) a.zss[(size_t)b * npad + tid] = tid < N ? hn[tid] : 1e300;  // padding: covariance exactly 0
  GS_STAMP(2);
  const int ks = a.dpad >> 2;  // k-steps that hold anything: the columns beyond dpad are zero (same bits with or without them)
  double wq[8];
#pragma unroll
  for (int k4 = 0; k4 < 8; ++k4) wq[k4] = w[4 * k4 + lq];
  // K11 -> B1.  The factor routine's first step reads column block 0 only and its first update the tiles (i, 1): those and
  // (2,2) are the first round, one sub-tile per wave; (3,2) and (3,3) are not touched before step 1 and are assembled by
  // waves 4, 5 while wave 0 is in step 0.
  {
    constexpr int R1_I[8] = {0, 1, 2, 3, 1, 2, 3, 2}, R1_J[8] = {0, 0, 0, 0, 1, 1, 1, 2};
    double kv[4];
    k_tile_vals(obs, wq, hn, tab, 0, 0, N, noise, R1_I[wave], R1_J[wave], ks, kv);
    tile_put(B1, R1_I[wave], R1_J[wave], kv);
  }
  GS_STAMP(3);
#ifdef B7_GS_STAMP
  {  // the same sub-tile once more: what does a K sub-tile cost when its code is already in the instruction cache?
    constexpr int R1_I[8] = {0, 1, 2, 3, 1, 2, 3, 2}, R1_J[8] = {0, 0, 0, 0, 1, 1, 1, 2};
    double kv2[4];
    GS_STAMP(14);
    k_tile_vals(obs, wq, hn, tab, 0, 0, N, noise, R1_I[wave], R1_J[wave], ks, kv2);
    tile_put(B1, R1_I[wave], R1_J[wave], kv2);
    GS_STAMP(15);
    if (blockIdx.x == 0 && lane == 0) b7_gs_stamps[16 + wave] = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);  // HW_ID.SIMD_ID
  }
#endif
  lds_barrier();
  GS_STAMP(4);
  // ---- block (0,0): factor and invert on waves 0..3.  Everything else that can be known before L11 is assembled meanwhile, a
  // sub-tile at a time, by whoever is idle: waves 4..7 during each of the routine's four 16-column pivot chains (3 900 cycles
  // of wave 0's; a sub-tile is 1 700 - 2 300), waves 1..3 in the routine's hook -- they have nothing of their own in step 0
  // and a few hundred cycles of it in steps 1..3.  What: K11's (3,2), (3,3) (first touched in step 1), the sixteen sub-tiles of
  // K21 -> B2, and the K entries of the first six sub-tiles of block (1,1) -- all that is live up to N = 112 -- which wait in
  // LDS (ks22) for the products they are held against further down.  Nothing is left for the routine's tail.
  //   slot table (t = sub-tile (t >> 2, t & 3) of K21; q = lower sub-tile q of block (1,1)):
  //     chain 0: waves 4..7: K11 (3,2), K11 (3,3), t0, t1     wave 1: t2    wave 2: q0    wave 3: q1
  //     chain 1: waves 4..7: t3 .. t6                         waves 1..3: t7, t8, t9
  //     chain 2: waves 4..7: t10 .. t13                       wave 3: t14   wave 1: t15
  //     chain 3: waves 4..7: q2 .. q5
  // ONE site of sub-tile code per caller (the kernel runs out of a cold instruction cache: every further copy is fetched from
  // memory again).  job 0..15: K21's sub-tile t -> B2; 16..21: the K entries of block (1,1)'s sub-tile q = job - 16 -> ks22;
  // 22, 23: K11's (3,2), (3,3) -> B1; < 0: nothing
  auto k_job = [&](int job) {
    if (job < 0) return;
    int I0 = 64, J0 = 0, it = job >> 2, jt = job & 3;
    if (job >= 22) {
      I0 = 0, it = 3, jt = job - 20;
    } else if (job >= 16) {
      J0 = 64;
      lower_tile(job - 16, it, jt);
    }
    double kv[4];
    k_tile_vals(obs, wq, hn, tab, I0, J0, N, noise, it, jt, ks, kv);
    if (job >= 16 && job < 22) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) ks22[(job - 16) * 256 + rr * 64 + lane] = kv[rr];
    } else {
      tile_put(job >= 22 ? B1 : B2, it, jt, kv);
    }
  };
  auto core_hook = [&](int kb, int wv) {  // waves 1..3, after their own share of step kb's factor phase (kb = 4: the tail)
    if (!two) return;
    int job = -1;
    if (kb == 0) job = wv == 1 ? 2 : 14 + wv;       // t2, q0, q1
    else if (kb == 1) job = 6 + wv;                  // t7, t8, t9
    else if (kb == 2 && wv != 2) job = wv == 3 ? 14 : 15;
    k_job(job);
  };
  if (!aux) {
#ifdef B7_GS_STAMP
    __shared__ unsigned long long dst_[24];  // the factor routine's own phase stamps (slots 2..17), block 0
    diag_core<1, true, decltype(core_hook), true>(B1, B3, T, 0, inf, dst_, core_hook, N < NB ? N : NB);
    if (blockIdx.x == 0 && tid < 24) b7_gs_stamps[32 + 8 + tid] = dst_[tid];
#else
    diag_core<1, false, decltype(core_hook), true>(B1, B3, T, 0, inf, nullptr, core_hook, N < NB ? N : NB);  // B1 -> L11 (lower), B3 = inv(L11)
#endif
  } else {
    diag_bystander<true>([&](int bi) {
      if ((bi & 1) || bi > 6) return;  // the even phases 0, 2, 4, 6 are wave 0's pivot chains
      const int g = wave - 4;
      int job;
      if (bi == 0) job = g < 2 ? 22 + g : (two ? g - 2 : -1);
      else job = !two ? -1 : bi == 2 ? 3 + g : bi == 4 ? 10 + g : 18 + g;
      k_job(job);
    });
  }
  GS_STAMP(5);
  // (diag_core ends with a barrier: everybody sees L11, inv(L11) and K21)
  double *Lb = MODE == 1 && a.L ? a.L + (size_t)b * npad * npad : nullptr;
  double *Lib = MODE == 1 ? a.Linv + (size_t)b * npad * npad : nullptr;
  double *dib = MODE == 1 && a.dinv ? a.dinv + (size_t)b * npad * NB : nullptr;
  if (!two && !aux) {  // one block: z1 = inv(L11) r1 on waves 0..3 -- four lanes per row, ascending columns within each quarter,
    const int row = tid >> 2, part = tid & 3;  // then the quarters in order; two blocks: below, off the path to block (1,1)
    if (MODE == 0) {
      double acc = 0.0;
      for (int k = 16 * part; k < 16 * part + 16; ++k) acc = __builtin_fma(B3[row * DLD + k], r[k], acc);
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      if (part == 0) z[row] = acc;
    }
    if (part == 0) dg[row] = B1[row * DLD + row];
  }
  if (two && wave == 1) dg[lane] = B1[lane * DLD + lane];  // before L11's image is given up (the barrier below)
  if (Lb) {
    store_block(B1, Lb, npad, true, tid, GS_THREADS);
    if (two) store_zero_block(Lb + NB, npad, tid, GS_THREADS);
  }
  if (two) {
    // L21 = K21 inv(L11)': 16 x 16 sub-tile (rs, jb) is the chain over k-steps 0 .. 4 jb + 3 (blocks above inv(L11)'s diagonal
    // skipped); wave (rs, 0) takes column blocks 0 and 3, wave (rs, 1) blocks 1 and 2: twenty MFMAs each
    const int rs = wave & 3;
    d4_t lv[2];
    if (wave < 4)
      l21_chains<0, 3>(B2 + (rs * 16 + lr) * DLD + lq, B3 + lr * DLD + lq, lv);
    else
      l21_chains<1, 2>(B2 + (rs * 16 + lr) * DLD + lq, B3 + lr * DLD + lq, lv);
    const int jc0 = wave < 4 ? 0 : 1, jc1 = wave < 4 ? 3 : 2;
    GS_STAMP(6);
    lds_barrier();  // every wave is done reading K21, and L11 has been taken out of B1
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      B2[(rs * 16 + lq + 4 * rr) * DLD + jc0 * 16 + lr] = lv[0][rr];
      B2[(rs * 16 + lq + 4 * rr) * DLD + jc1 * 16 + lr] = lv[1][rr];
    }
    lds_barrier();
    GS_STAMP(7);
    // K22 - L21 L21' -> B1, lower sub-tiles only, a sub-tile per wave (two for waves 0, 1): its K entries (registers), the
    // 64-deep chain of L21 L21' from zero, then the subtraction
    // sub-tile q goes to wave (q + 2) mod 8: the first six -- all there is to compute up to N = 112 -- land on waves 2..7
    for (int q = (wave + 6) & 7; q < 10; q += 8) {
      int it, jt;
      lower_tile(q, it, jt);
      if (NB + 16 * it >= N) {  // a row strip wholly in the padding: K is the identity there and L21's rows are zero (u = +0)
        const double one[4] = {it == jt && lq == lr ? 1.0 : 0.0, it == jt && lq + 4 == lr ? 1.0 : 0.0, it == jt && lq + 8 == lr ? 1.0 : 0.0,
                               it == jt && lq + 12 == lr ? 1.0 : 0.0};
        tile_put(B1, it, jt, one);
        continue;
      }
      double av[16], bv[16], kv[4];
      const double *ar = B2 + (16 * it + lr) * DLD + lq, *br = B2 + (16 * jt + lr) * DLD + lq;
#pragma unroll
      for (int k4 = 0; k4 < 16; ++k4) av[k4] = ar[4 * k4], bv[k4] = br[4 * k4];
      if (q >= K22_STASH) {
        k_tile_vals(obs, wq, hn, tab, 64, 64, N, noise, it, jt, ks, kv);  // row strip 3 (N > 112): formed here
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) kv[rr] = ks22[q * 256 + rr * 64 + lane];
      }
      d4_t u = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k4 = 0; k4 < 16; ++k4) u = mfma_f64(av[k4], bv[k4], u);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) B1[(16 * it + lq + 4 * rr) * DLD + 16 * jt + lr] = kv[rr] - u[rr];
    }
    d4_t pv[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    if (MODE == 1) {
      // L21 inv(L11): the first (and only) 64-deep chunk of inv_job's partial sums, k ascending; inv(L11) is lower triangular,
      // the k-steps above column block jb's diagonal hold zeros and are skipped (they add nothing).  Column blocks as above.
      if (wave < 4)
        p_chains<0, 3>(B2 + (rs * 16 + lr) * DLD + lq, B3 + lq * DLD + lr, pv);
      else
        p_chains<1, 2>(B2 + (rs * 16 + lr) * DLD + lq, B3 + lq * DLD + lr, pv);
      // what of block row 0 and of L21 goes to global memory (the images are read-only in this phase)
      if (Lb) store_block(B2, Lb + (size_t)NB * npad, npad, false, tid, GS_THREADS);
      store_block(B3, Lib, npad, true, tid, GS_THREADS);
      store_zero_block(Lib + NB, npad, tid, GS_THREADS);
      if (dib) store_block(B3, dib, NB, true, tid, GS_THREADS);
    }
    GS_STAMP(8);
    lds_barrier();  // nobody reads the observations or L21's image any more
    GS_STAMP(9);
    if (MODE == 1) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {  // inv_job: tot = tot + cur
        B2[(rs * 16 + lq + 4 * rr) * DLD + jc0 * 16 + lr] = 0.0 + pv[0][rr];
        B2[(rs * 16 + lq + 4 * rr) * DLD + jc1 * 16 + lr] = 0.0 + pv[1][rr];
      }
    }
    zero_block(B0);
    identity_corner(B1);  // rows 0..15 x columns 48..63: sub-tile (0,3), which nobody above wrote
    lds_barrier();
    GS_STAMP(10);
    if (!aux) {
#ifdef B7_GS_STAMP
      __shared__ unsigned long long dst2_[24];
      diag_core<1, true, NoHook, true>(B1, B0, T, 1, inf, dst2_, NoHook(), N - NB);
      if (blockIdx.x == 0 && tid < 24) b7_gs_stamps[64 + tid] = dst2_[tid];
#else
      diag_core<1, false, NoHook, true>(B1, B0, T, 1, inf, nullptr, NoHook(), N - NB);  // B1 -> L22, B0 = inv(L22)
#endif
    } else {
      // the helper waves have nothing to assemble any more.  MODE 0: what the second block's factor does not need happens
      // here, off everybody's way -- z1 = inv(L11) r1 (four lanes per row, ascending columns within each quarter, then the
      // quarters in order: the general path's sums) under the first pivot chain, r2 -= L21 z1 under the second; inv(L11)'s and
      // L21's images stay as they are until the end of a likelihood evaluation.
      diag_bystander<true>([&](int bi) {
        if (MODE != 0 || (bi != 0 && bi != 2)) return;
        const int row = (tid & 255) >> 2, part = tid & 3;
        const double *m = (bi == 0 ? B3 : B2) + row * DLD + 16 * part, *v = (bi == 0 ? r : z) + 16 * part;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = __builtin_fma(m[k], v[k], acc);
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (part == 0) {
          if (bi == 0) z[row] = acc;
          else r[64 + row] = r[64 + row] - acc;
        }
      });
    }
    GS_STAMP(11);
    if (!aux) {
      const int row = tid >> 2, part = tid & 3;
      if (MODE == 0) {
        double acc = 0.0;
        for (int k = 16 * part; k < 16 * part + 16; ++k) acc = __builtin_fma(B0[row * DLD + k], r[64 + k], acc);
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (part == 0) z[64 + row] = acc;
      }
      if (part == 0) dg[64 + row] = B1[row * DLD + row];
    }
    if (MODE == 1) {
      // inv(L)21 = -inv(L22) (L21 inv(L11)): inv_job's epilogue -- per sub-tile two accumulators that take the k-steps of
      // every 16-block alternately, blocks above inv(L22)'s diagonal skipped, their sum at the end.  Wave (rs, half) takes
      // sub-tiles (rs, 2 half) and (3 - rs, 2 half + 1): twenty MFMAs each
      const int half = wave >> 2;
      d4_t outv[2];
#pragma unroll
      for (int si = 0; si < 2; ++si) {
        const int rr2 = si == 0 ? rs : 3 - rs, s = 2 * half + si;
        d4_t c0a = {0.0, 0.0, 0.0, 0.0}, c1a = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
          if (kq > rr2) break;
#pragma unroll
          for (int s4 = 0; s4 < 4; s4 += 2) {
            c0a = mfma_f64(-B0[(rr2 * 16 + lr) * DLD + kq * 16 + 4 * s4 + lq], B2[(kq * 16 + 4 * s4 + lq) * DLD + 16 * s + lr], c0a);
            c1a = mfma_f64(-B0[(rr2 * 16 + lr) * DLD + kq * 16 + 4 * s4 + 4 + lq], B2[(kq * 16 + 4 * s4 + 4 + lq) * DLD + 16 * s + lr],
                           c1a);
          }
        }
        outv[si] = c0a + c1a;
      }
      GS_STAMP(23);
      lds_barrier();
#pragma unroll
      for (int si = 0; si < 2; ++si) {
        const int rr2 = si == 0 ? rs : 3 - rs, s = 2 * half + si;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) B2[(rr2 * 16 + lq + 4 * rr) * DLD + 16 * s + lr] = outv[si][rr];
      }
      lds_barrier();
      GS_STAMP(24);
      store_block(B2, Lib + (size_t)NB * npad, npad, false, tid, GS_THREADS);
      store_block(B0, Lib + (size_t)NB * npad + NB, npad, true, tid, GS_THREADS);
      if (dib) store_block(B0, dib + NB * NB, NB, true, tid, GS_THREADS);
      if (Lb) store_block(B1, Lb + (size_t)NB * npad + NB, npad, true, tid, GS_THREADS);
    }
  } else if (MODE == 1) {
    store_block(B3, Lib, npad, true, tid, GS_THREADS);
    if (dib) store_block(B3, dib, NB, true, tid, GS_THREADS);
  }

  GS_STAMP(12);
  if (MODE == 0) {
    lds_barrier();
    // |z|^2 and sum log L_ii in a fixed order: a butterfly inside each wave, then the waves in order (nll_small_kernel's)
    double ssq = 0.0, ld = 0.0;
    if (tid < npad) {
      ssq = z[tid] * z[tid];
      ld = log(dg[tid]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ssq += __shfl_xor(ssq, o);
      ld += __shfl_xor(ld, o);
    }
    if (lane == 0 && wave < 4) {
      red[wave] = ssq;
      red[4 + wave] = ld;
    }
    lds_barrier();
    if (tid == 0) {
      a.terms[2 * b] = (red[0] + red[1]) + (red[2] + red[3]);
      a.terms[2 * b + 1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
  } else {
    double *alb = a.alpha + (size_t)b * npad;
    if (two) {
      // t = inv(L) r with trmv_lower_kernel's sums: there a wave takes a row, lane k the columns k and k + 64 (k <= row), and a
      // butterfly (xor 32, 16, ..., 1) adds the 64 partial sums.  Here a THREAD takes a row and walks the same binary tree over
      // its 64 partial sums in registers -- level o adds element l and l + o for l < o, exactly the pairs lane 0 of the
      // butterfly sees -- so the bits are the same and nothing crosses lanes (192 ds_bpermute per wave took 6 us).
      {
        // ... and four threads take a row: thread q of the quad the columns l = q (mod 4), whose partial sums meet each other
        // at every level down to o = 4 inside that thread (l and l + o are the same residue), the last two levels -- (0, 2),
        // (1, 3), then (0, 1) -- across the quad.  The same tree, a quarter of the time, all eight waves
        const int row = tid >> 2, q = tid & 3;
        const double *l0 = (row < NB ? B3 + row * DLD : B2 + (row - NB) * DLD) + q;  // columns 0..63 of the row
        const double *l1 = B0 + (row < NB ? 0 : row - NB) * DLD + q;                  // columns 64..127 (rows >= 64)
        double p[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) p[m] = __builtin_fma(l0[4 * m], r[4 * m + q], 0.0);
        if (wave >= 4) {  // rows 64..127
#pragma unroll
          for (int m = 0; m < 16; ++m) p[m] = __builtin_fma(l1[4 * m], r[NB + 4 * m + q], p[m]);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1)
#pragma unroll
          for (int m = 0; m < o; ++m) p[m] = p[m] + p[m + o];
        double v = p[0];
        v = v + __shfl_xor(v, 2);
        v = v + __shfl_xor(v, 1);
        if (q == 0) tv[row] = v;
      }
      GS_STAMP(20);
      lds_barrier();
      GS_STAMP(21);
      // alpha = inv(L)' t as trmv_lower_t_part_kernel / _sum_kernel sum it: per column the chain over rows 0..63 and the chain
      // over rows 64..127 (each ascending from zero), ((first + second) + 0) + 0, then 0 + that
      if (tid < 128) {
        const int col = tid;
        double s0 = 0.0, s1 = 0.0;
        if (col < NB) {
#pragma unroll 8
          for (int i = 0; i < NB; ++i) {
            s0 = __builtin_fma(B3[i * DLD + col], tv[i], s0);
            s1 = __builtin_fma(B2[i * DLD + col], tv[NB + i], s1);
          }
        } else {
#pragma unroll 8
          for (int i = 0; i < NB; ++i) s1 = __builtin_fma(B0[i * DLD + (col - NB)], tv[NB + i], s1);
        }
        double part = ((s0 + s1) + 0.0) + 0.0;
        part = 0.0 + part;
        alb[col] = col < N ? part : 0.0;
      }
    } else {
      // one block: potrf_small64_kernel's sums -- t by four lanes per row (ascending quarters, then the quarters in order),
      // alpha_j = chain over rows j..63
      if (tid < 256) {
        const int row = tid >> 2, part = tid & 3;
        double acc = 0.0;
        for (int k = 16 * part; k < 16 * part + 16; ++k) acc = __builtin_fma(B3[row * DLD + k], r[k], acc);
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (part == 0) tv[row] = acc;
      }
      lds_barrier();
      if (tid < NB) {
        double acc = 0.0;
        for (int i = tid; i < NB; ++i) acc = __builtin_fma(B3[i * DLD + tid], tv[i], acc);
        alb[tid] = tid < N ? acc : 0.0;
      }
    }
    GS_STAMP(22);
    if (a.resid && tid < npad) a.resid[(size_t)b * npad + tid] = r[tid];
  }
  GS_STAMP(13);
  if (tid == 0) {
    if (a.info)
      for (int k = 0; k < 4; ++k) a.info[4 * b + k] = inf[k];
    if (a.report)
      for (int k = 0; k < 4; ++k) a.report[4 * b + k] = inf[k];
  }
  if (a.done) {
    // a single evaluation's caller spins on this word instead of waiting for the dispatch to retire (MODE 0: thread 0 wrote
    // everything the host reads; the release orders it before the flag as the host sees them)
    if (tid == 0) __hip_atomic_store(a.done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}


int ensure_gs_table(b7_ctx *c) {
  static bool done[64] = {false};
  if (c->device < 64 && done[c->device]) return B7_OK;
  B7_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(exp2_tab_gs), b7_exp2_tab, sizeof(b7_exp2_tab)));
  if (c->device < 64) done[c->device] = true;
  return B7_OK;
}

template <int MODE, bool TWO>
int gs_launch2(b7_ctx *c, const GsArgs &a, const double *hyp_host) {
  B7_TRY(ensure_gs_table(c));
  const size_t lds = sizeof(double) * GS_LDS_DOUBLES;
  // the opt-in to > 64 KiB of dynamic LDS is per device: once per process AND device (a process may hold contexts on several)
  static bool attr_done[64] = {false};
  if (c->device >= 64 || !attr_done[c->device]) {
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(gp_small_kernel<MODE, TWO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (c->device < 64) attr_done[c->device] = true;
  }
  GsInline hin = {};
  GsArgs k = a;
  k.use_inline = (a.B == 1 && hyp_host != nullptr) ? 1 : 0;
  for (int i = 0; k.use_inline && i < a.d + 3; ++i) hin.v[i] = hyp_host[i];
  hipLaunchKernelGGL((gp_small_kernel<MODE, TWO>), dim3(a.B), dim3(GS_THREADS), lds, c->stream, k, hin);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}
template <int MODE>
int gs_launch(b7_ctx *c, const GsArgs &a, const double *hyp_host) {
  return a.N > NB ? gs_launch2<MODE, true>(c, a, hyp_host) : gs_launch2<MODE, false>(c, a, hyp_host);
}

}  // namespace

// (the kernel pads N to one or two 64-blocks itself; a context told to pad small sets to 128 -- the diagnostic build's
// B7_NPAD_SMALL=0 -- keeps the general path)
bool gp_small_applies(const b7_ctx *c) { return c->Npad == (c->N > 64 ? 128 : 64) && c->dfit <= 32 && c->ycols == 1; }

// B likelihood evaluations of the resident data.  hyp_dev: [B x d lengthscales | B amp | B noise | B mean] (b7_gp_nll_batch's
// pack, device-visible); terms_dev[2 B], info_dev[4 B]; done_dev (nullable): a word the kernel sets to 1 after its results
// are visible to the host (B == 1 only).  hyp_host: the same pack in host memory; a single evaluation's hypers travel in the
// kernel arguments instead.
int launch_nll_small8(b7_ctx *c, int B, const double *hyp_dev, const double *hyp_host, double *terms_dev, int *info_dev,
                      unsigned *done_dev) {
  PhaseScope ps(c, "potrf");
  GsArgs a = {};
  a.xobs = (const double *)c->xobs.p;
  a.y = (const double *)c->ybuf.p;
  a.N = c->N, a.d = c->dfit, a.dpad = c->dpad, a.B = B;
  a.hyp_mem = hyp_dev;
  a.info = info_dev;
  a.done = B == 1 ? done_dev : nullptr;
  a.terms = terms_dev;
  return gs_launch<0>(c, a, hyp_host);
}

// B whole fits of the resident data (b7_eval_nominate's hyper samples; B = 1: the context's own fit slot).  The outputs are laid
// out as the general path's batch buffers: w [B][dpad], zsc [B][Npad][dpad], zss [B][Npad], Linv (and L, nullable) [B][Npad^2],
// dinv (nullable) [B][Npad 64], alpha (and resid, nullable) [B][Npad]; hyp_out (nullable) receives the pack for the kernels
// downstream.  info_dev / report_dev: 4 ints per fit, the second in mapped host memory (either may be null).
int launch_fit_small(b7_ctx *c, int B, const double *hyp_dev, const double *hyp_host, double *hyp_out, double *w, double *zsc,
                     double *zss, double *L, double *Linv, double *dinv, double *alpha, double *resid, int *info_dev,
                     int *report_dev) {
  PhaseScope ps(c, "potrf");
  GsArgs a = {};
  a.xobs = (const double *)c->xobs.p;
  a.y = (const double *)c->ybuf.p;
  a.N = c->N, a.d = c->dfit, a.dpad = c->dpad, a.B = B;
  a.hyp_mem = hyp_dev;
  a.info = info_dev;
  a.report = report_dev;
  a.hyp_out = hyp_out;
  a.w = w, a.zsc = zsc, a.zss = zss, a.L = L, a.Linv = Linv, a.dinv = dinv, a.alpha = alpha, a.resid = resid;
  return gs_launch<1>(c, a, hyp_host);
}
