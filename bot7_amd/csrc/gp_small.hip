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
// barriers (diag_bystander) and do what nothing on the dependent chain waits for: the K sub-tiles of block row 1 while
// block (0,0) is factored, those of block (1,1) while L21 is solved and squared, the stores to global memory.  Four 64 x 66
// images in LDS change roles as the data die:  B0 observations -> inv(L22);  B1 K11 / L11 -> K22 / L22;  B2 K21 -> L21 ->
// L21 inv(L11) -> inv(L)21;  B3 inv(L11).
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

namespace {
using namespace b7diag;  // NB = 64, DLD, TLD, diag_core, diag_bystander

__constant__ double exp2_tab_gs[128];  // B7_EXP2_TAB (ensure_gs_table)

constexpr int OLD = 33;         // row stride of the observation image [128][OLD] (32 columns, zero padded)
constexpr int BUF = NB * DLD;   // one 64 x 66 image
static_assert(128 * OLD <= BUF, "the observation image lives in one block image");
constexpr int GS_THREADS = 512;
constexpr int GS_LDS_DOUBLES = 4 * BUF + 32 * TLD + 5 * 128 + 32 + 128 + 64;

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

// One 16 x 16 sub-tile (it, jt) of the 64 x 64 block (I0, J0) of K(X,X) + noise I -> T; rows / columns >= N are the identity.
// x (z .* w)' on MFMA exactly as ksx_kernel forms it: A fragments are the raw rows, B fragments the raw columns' rows times w
// (the product rounded once, as prep_obs_kernel rounds z .* w), a chain of v_mfma_f64_16x16x4 over the eight k-steps = the
// ascending fma chain over the 32 (zero padded) dimensions.  A sub-tile wholly in the padding is written, not computed.
__device__ __forceinline__ void k_tile(const double *__restrict__ obs, const double (&wq)[8], const double *__restrict__ hn,
                                       const double *__restrict__ tab, int I0, int J0, int N, double noise, double *__restrict__ T,
                                       int it, int jt) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const int gj = J0 + 16 * jt + lr;
  if (I0 + 16 * it >= N || J0 + 16 * jt >= N) {  // wave-uniform
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * it + lq + 4 * r;
      T[i * DLD + 16 * jt + lr] = (I0 + i == gj) ? 1.0 : 0.0;
    }
    return;
  }
  d4_t c = {0.0, 0.0, 0.0, 0.0};
  const double *ap = obs + (I0 + 16 * it + lr) * OLD + lq, *bp = obs + gj * OLD + lq;
#pragma unroll
  for (int k4 = 0; k4 < 8; ++k4) c = mfma_f64(ap[4 * k4], bp[4 * k4] * wq[k4], c);
  const double hj = hn[gj];
  double arg[4], kv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) arg[r] = (c[r] - hn[I0 + 16 * it + lq + 4 * r]) - hj;
  amp_exp_nonpos4(arg, tab, kv);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 16 * it + lq + 4 * r, gi = I0 + i;
    double v = kv[r];
    if (gi >= N || gj >= N) v = (gi == gj) ? 1.0 : 0.0;
    else if (gi == gj) v = v + noise;
    T[i * DLD + 16 * jt + lr] = v;
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
  for (int e = t0; e < NB * NB; e += nt) {
    const int i = e >> 6, j = e & 63;
    dst[(int64_t)i * ld + j] = (!lower || j <= i) ? img[i * DLD + j] : 0.0;
  }
}
__device__ __forceinline__ void store_zero_block(double *__restrict__ dst, int64_t ld, int t0, int nt) {
  for (int e = t0; e < NB * NB; e += nt) dst[(int64_t)(e >> 6) * ld + (e & 63)] = 0.0;
}

template <int MODE>
__global__ void __launch_bounds__(GS_THREADS) gp_small_kernel(GsArgs a, GsInline hin) {
  extern __shared__ __align__(16) double sm[];
  __shared__ int inf[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15,
            lq = lane >> 4;
  const int N = a.N, d = a.d, B = a.B;
  const int npad = N > 64 ? 128 : 64;
  const bool two = npad == 128, aux = wave >= 4;
  double *B0 = sm, *B1 = B0 + BUF, *B2 = B1 + BUF, *B3 = B2 + BUF, *T = B3 + BUF;
  double *r = T + 32 * TLD;   // [128] residual y - mean
  double *hn = r + 128;       // [128] half norms
  double *z = hn + 128;       // [128] MODE 0: L^-1 r
  double *dg = z + 128;       // [128] diagonal of L
  double *tv = dg + 128;      // [128] MODE 1: t = L^-1 r in trmv_lower_kernel's order
  double *w = tv + 128;       // [32]
  double *tab = w + 32;       // [128] amp 2^(j/128)
  double *red = tab + 128;    // [64]
  double *obs = B0;
  const double *hyp = a.use_inline ? hin.v : a.hyp_mem;
  const double *ls = hyp + (size_t)b * d;
  const double amp = hyp[(size_t)B * d + b], noise = hyp[(size_t)B * (d + 1) + b], mean = hyp[(size_t)B * (d + 2) + b];
  if (tid < 4) inf[tid] = 0;
  if (tid < 32) w[tid] = tid < d ? 1.0 / ls[tid] : 0.0;  // inv_ls = ones:cdiv(lenscale), utils/math.lua:72
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
    __syncthreads();
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
  __syncthreads();
  if (tid < 128) {
    double s = 0.0;
    for (int k = 0; k < 32; ++k) {
      const double x = obs[tid * OLD + k];
      s += (x * x) * w[k];  // Z_ss = (Z.^2) * inv_ls, :79
    }
    hn[tid] = 0.5 * s;
  }
  if (MODE == 1) {
    // what the posterior kernels read of the observations, as prep_obs_kernel leaves it: z .* w (zero padded), the weights
    // (the half norms follow below, once they are in LDS), and this fit's hypers where the kernels downstream find them
    const int dpad = a.dpad;
    double *zo = a.zsc + (size_t)b * npad * dpad;
    for (int e = tid; e < npad * dpad; e += GS_THREADS) {
      const int i = e / dpad, k = e - i * dpad;
      zo[e] = obs[i * OLD + k] * w[k];
    }
    if (tid < dpad) a.w[(size_t)b * dpad + tid] = w[tid];
    if (a.hyp_out && tid < d + 3) {
      const size_t at = tid < d ? (size_t)b * d + tid : (size_t)B * (d + (tid - d)) + b;
      a.hyp_out[at] = hyp[at];
    }
  }
  __syncthreads();
  if (MODE == 1 && tid < npad) a.zss[(size_t)b * npad + tid] = tid < N ? hn[tid] : 1e300;  // padding: covariance exactly 0
  double wq[8];
#pragma unroll
  for (int k4 = 0; k4 < 8; ++k4) wq[k4] = w[4 * k4 + lq];
  // K11 -> B1: the ten sub-tiles on and below the diagonal, over all eight waves
  for (int q = wave; q < 10; q += 8) {
    int it, jt;
    lower_tile(q, it, jt);
    k_tile(obs, wq, hn, tab, 0, 0, N, noise, B1, it, jt);
  }
  zero_block(B3);
  __syncthreads();
  identity_corner(B1);
  __syncthreads();
  // ---- block (0,0): factor and invert on waves 0..3; waves 4..7 assemble K21 -> B2 meanwhile, a round of sub-tiles in front of
  // every other barrier of the routine
  if (!aux) {
    diag_core<1, false>(B1, B3, T, 0, inf, nullptr, NoHook(), N < NB ? N : NB);  // B1 -> L11 (lower), B3 = inv(L11)
  } else {
    diag_bystander([&](int bi) {
      if (two && !(bi & 1) && (bi >> 1) < 4) {
        const int q = (wave - 4) + 4 * (bi >> 1);
        k_tile(obs, wq, hn, tab, 64, 0, N, noise, B2, q >> 2, q & 3);
      }
    });
  }
  // (diag_core ends with a barrier: everybody sees L11, inv(L11) and K21)
  double *Lb = MODE == 1 && a.L ? a.L + (size_t)b * npad * npad : nullptr;
  double *Lib = MODE == 1 ? a.Linv + (size_t)b * npad * npad : nullptr;
  double *dib = MODE == 1 && a.dinv ? a.dinv + (size_t)b * npad * NB : nullptr;
  if (!aux) {
    const int row = tid >> 2, part = tid & 3;
    if (MODE == 0) {  // z1 = inv(L11) r1: four lanes per row, ascending columns within each quarter, then the quarters in order
      double acc = 0.0;
      for (int k = 16 * part; k < 16 * part + 16; ++k) acc = __builtin_fma(B3[row * DLD + k], r[k], acc);
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      if (part == 0) z[row] = acc;
    }
    if (part == 0) dg[row] = B1[row * DLD + row];
  } else if (Lb) {
    store_block(B1, Lb, npad, true, tid - 256, 256);
    if (two) store_zero_block(Lb + NB, npad, tid - 256, 256);
  }
  if (two) {
    // L21 = K21 inv(L11)': wave (rs, half) rows 16 rs .., column blocks 2 half, 2 half + 1; k ascending, blocks above inv(L11)'s
    // diagonal skipped
    const int rs = wave & 3, jb0 = 2 * (wave >> 2);
    d4_t lv[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    {
      const double *ap = B2 + (rs * 16 + lr) * DLD + lq, *xp = B3 + lr * DLD + lq;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const double aq = ap[4 * t];
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (jb0 + j >= (t >> 2)) lv[j] = mfma_f64(aq, xp[(jb0 + j) * 16 * DLD + 4 * t], lv[j]);
      }
    }
    __syncthreads();  // every wave is done reading K21 (and L11 has been taken out of B1)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int j = 0; j < 2; ++j) B2[(rs * 16 + lq + 4 * rr) * DLD + (jb0 + j) * 16 + lr] = lv[j][rr];
    __syncthreads();
    d4_t u[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    d4_t pv[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    if (!aux) {
      // K22 - L21 L21' on the lower sub-tiles of row strip `wave`: the 64-deep chain from zero (subtracted below, once waves
      // 4..7 have put K22 in place)
      const double *ar = B2 + (16 * wave + lr) * DLD + lq;
#pragma unroll
      for (int k4 = 0; k4 < 16; ++k4) {
        const double af = ar[4 * k4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
          if (jb <= wave) u[jb] = mfma_f64(af, B2[(16 * jb + lr) * DLD + lq + 4 * k4], u[jb]);
      }
      if (MODE == 0) {  // r2 -= L21 z1
        const int row = tid >> 2, part = tid & 3;
        double acc = 0.0;
        for (int k = 16 * part; k < 16 * part + 16; ++k) acc = __builtin_fma(B2[row * DLD + k], z[k], acc);
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (part == 0) r[64 + row] = r[64 + row] - acc;
      } else {
        // L21 inv(L11): the first (and only) 64-deep chunk of inv_job's partial sums, k ascending; inv(L11) is lower
        // triangular, the k-steps above column block jb's diagonal hold zeros and are skipped (they add nothing)
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) {
          const double af = ar[4 * k4];
#pragma unroll
          for (int jb = 0; jb < 4; ++jb)
            if (k4 >= 4 * jb) pv[jb] = mfma_f64(af, B3[(4 * k4 + lq) * DLD + 16 * jb + lr], pv[jb]);
        }
      }
    } else {
      // K22 -> B1 (lower sub-tiles), then what of the first block goes to global memory
      for (int q = wave - 4; q < 10; q += 4) {
        int it, jt;
        lower_tile(q, it, jt);
        k_tile(obs, wq, hn, tab, 64, 64, N, noise, B1, it, jt);
      }
      if (MODE == 1) {
        if (Lb) store_block(B2, Lb + (size_t)NB * npad, npad, false, tid - 256, 256);
        store_block(B3, Lib, npad, true, tid - 256, 256);
        store_zero_block(Lib + NB, npad, tid - 256, 256);
        if (dib) store_block(B3, dib, NB, true, tid - 256, 256);
      }
    }
    __syncthreads();  // K22 is in B1; nobody reads the observations or L21's image any more
    if (!aux) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
          if (jb <= wave) {
            const int e = (16 * wave + lq + 4 * rr) * DLD + 16 * jb + lr;
            B1[e] = B1[e] - u[jb][rr];
          }
      if (MODE == 1) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int jb = 0; jb < 4; ++jb) B2[(16 * wave + lq + 4 * rr) * DLD + 16 * jb + lr] = 0.0 + pv[jb][rr];  // inv_job: tot = tot + cur
      }
    }
    zero_block(B0);
    __syncthreads();
    identity_corner(B1);
    __syncthreads();
    if (!aux) {
      diag_core<1, false>(B1, B0, T, 1, inf, nullptr, NoHook(), N - NB);  // B1 -> L22, B0 = inv(L22)
    } else {
      diag_bystander([&](int) {});
    }
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
      // inv(L)21 = -inv(L22) (L21 inv(L11)): inv_job's epilogue -- per column block two accumulators that take the k-steps of
      // every 16-block alternately, blocks above inv(L22)'s diagonal skipped, their sum at the end
      const int rs2 = wave & 3, s0 = 2 * (wave >> 2);
      d4_t outv[2];
#pragma unroll
      for (int si = 0; si < 2; ++si) {
        const int s = s0 + si;
        d4_t c0a = {0.0, 0.0, 0.0, 0.0}, c1a = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
          if (kq > rs2) break;
#pragma unroll
          for (int s4 = 0; s4 < 4; s4 += 2) {
            c0a = mfma_f64(-B0[(rs2 * 16 + lr) * DLD + kq * 16 + 4 * s4 + lq], B2[(kq * 16 + 4 * s4 + lq) * DLD + 16 * s + lr], c0a);
            c1a = mfma_f64(-B0[(rs2 * 16 + lr) * DLD + kq * 16 + 4 * s4 + 4 + lq], B2[(kq * 16 + 4 * s4 + 4 + lq) * DLD + 16 * s + lr],
                           c1a);
          }
        }
        outv[si] = c0a + c1a;
      }
      __syncthreads();
#pragma unroll
      for (int si = 0; si < 2; ++si)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) B2[(rs2 * 16 + lq + 4 * rr) * DLD + 16 * (s0 + si) + lr] = outv[si][rr];
      __syncthreads();
      store_block(B2, Lib + (size_t)NB * npad, npad, false, tid, GS_THREADS);
      store_block(B0, Lib + (size_t)NB * npad + NB, npad, true, tid, GS_THREADS);
      if (dib) store_block(B0, dib + NB * NB, NB, true, tid, GS_THREADS);
      if (Lb) store_block(B1, Lb + (size_t)NB * npad + NB, npad, true, tid, GS_THREADS);
    }
  } else if (MODE == 1) {
    if (tid < 256) {
      store_block(B3, Lib, npad, true, tid, 256);
      if (dib) store_block(B3, dib, NB, true, tid, 256);
    }
  }

  if (MODE == 0) {
    __syncthreads();
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
    __syncthreads();
    if (tid == 0) {
      a.terms[2 * b] = (red[0] + red[1]) + (red[2] + red[3]);
      a.terms[2 * b + 1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
  } else {
    double *alb = a.alpha + (size_t)b * npad;
    if (two) {
      // t = inv(L) r, row by row as trmv_lower_kernel sums it: lane k takes columns k and k + 64 (k <= row), then the butterfly
      for (int i = 0; i < 16; ++i) {
        const int row = 16 * wave + i;
        const double *l0 = row < NB ? B3 + row * DLD : B2 + (row - NB) * DLD;  // columns 0..63 of the row
        double s = 0.0;
        if (lane <= row) s = __builtin_fma(l0[lane], r[lane], s);
        if (row >= NB && lane + NB <= row) s = __builtin_fma(B0[(row - NB) * DLD + lane], r[NB + lane], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) tv[row] = s;
      }
      __syncthreads();
      // alpha = inv(L)' t as trmv_lower_t_part_kernel / _sum_kernel sum it: per column the chain over rows 0..63 and the chain
      // over rows 64..127 (each ascending from zero), ((first + second) + 0) + 0, then 0 + that
      if (tid < 128) {
        const int col = tid;
        double s0 = 0.0, s1 = 0.0;
        if (col < NB) {
          for (int i = 0; i < NB; ++i) s0 = __builtin_fma(B3[i * DLD + col], tv[i], s0);
          for (int i = 0; i < NB; ++i) s1 = __builtin_fma(B2[i * DLD + col], tv[NB + i], s1);
        } else {
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
      __syncthreads();
      if (tid < NB) {
        double acc = 0.0;
        for (int i = tid; i < NB; ++i) acc = __builtin_fma(B3[i * DLD + tid], tv[i], acc);
        alb[tid] = tid < N ? acc : 0.0;
      }
    }
    if (a.resid && tid < npad) a.resid[(size_t)b * npad + tid] = r[tid];
  }
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
  B7_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(exp2_tab_gs), B7_EXP2_TAB, sizeof(B7_EXP2_TAB)));
  if (c->device < 64) done[c->device] = true;
  return B7_OK;
}

template <int MODE>
int gs_launch(b7_ctx *c, const GsArgs &a, const double *hyp_host) {
  B7_TRY(ensure_gs_table(c));
  const size_t lds = sizeof(double) * GS_LDS_DOUBLES;
  // the opt-in to > 64 KiB of dynamic LDS is per device: once per process AND device (a process may hold contexts on several)
  static bool attr_done[64] = {false};
  if (c->device >= 64 || !attr_done[c->device]) {
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(gp_small_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (c->device < 64) attr_done[c->device] = true;
  }
  GsInline hin = {};
  GsArgs k = a;
  k.use_inline = (a.B == 1 && hyp_host != nullptr) ? 1 : 0;
  for (int i = 0; k.use_inline && i < a.d + 3; ++i) hin.v[i] = hyp_host[i];
  hipLaunchKernelGGL(gp_small_kernel<MODE>, dim3(a.B), dim3(GS_THREADS), lds, c->stream, k, hin);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

}  // namespace

bool gp_small_applies(const b7_ctx *c) { return c->Npad <= 128 && c->dfit <= 32 && c->ycols == 1; }

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
