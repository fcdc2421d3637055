// The GP likelihood for SMALL observation sets in one kernel, one workgroup per evaluation.
//
// Reference: model:sample_hypers (bots/bayesopt.lua:68,73-75) drives samplers/slice.lua:92-168, and every density evaluation
// is K(X,X) + noise I (utils/math.lua:65-111), its Cholesky (utils/math.lua:159-218) and r' K^-1 r + log|K|.  With the
// reference's own defaults (budget 100: bots/abstract.lua:64, so N <= 100) that is hundreds of SEQUENTIAL evaluations per
// trial against one nomination, and at N <= 128 the general path (observation scaling, K assembly, fix-up, the persistent
// factorisation with its vector job: six launches, each at the ~4.5 us dispatch floor, around 25 us of arithmetic) is bound by
// launches, not by work.  Here a workgroup does all of it in LDS for one hyper vector: raw observations in, (|z|^2,
// sum log L_ii, first bad pivot) out; B evaluations are B workgroups of one launch with no communication between them.
//
// The arithmetic is the general path's, operation for operation where that is cheap to say: K entries are the same
// ascending fma chain over the input dimensions that a chain of v_mfma_f64_16x16x4 computes in ksx_kernel, the same
// (c - xs/2) - zs/2 argument and the same table exponential (ksx_exp.h); the 64x64 diagonal blocks go through
// b7diag::diag_core (potrf_diag.h), the block below by C inv(L11)' and the 64-deep update chain from zero followed by the
// subtraction, as the persistent schedule does.  Only the final reductions (|z|^2, log-determinant) are summed in another
// order; tests hold the result against the general path at 1e-12 relative.
//
// Limits: Npad <= 128 (one or two 64-blocks), d <= 32, one response column.  A failed pivot is reported, not repaired: the
// caller redoes that evaluation through the general path and its jitter schedule.
#include "b7_internal.h"
#include "ksx_exp.h"
#include "potrf_diag.h"

namespace {
using namespace b7diag;  // NB = 64, DLD, TLD, diag_core

__constant__ double exp2_tab_small[128];  // b7_exp2_tab (ensure_small_table)

constexpr int OLD = 33;  // row stride of the observation image [128][OLD] (32 columns, zero padded)
constexpr int NLL_SMALL_LDS_DOUBLES = 128 * OLD + 3 * NB * DLD + 32 * TLD + 128 * 4 + 32 + 128 + 512;
static_assert(128 * OLD <= NB * DLD, "the observation image is reused as the inverse's tile");

// rows 0..15 x columns 48..63 of a block about to be factored: I_16 (potrf_diag.h: the right-hand side of the inversion)
__device__ __forceinline__ void identity_corner(double *A) {
  const int t = threadIdx.x;
  if (t < 256) {
    const int i = t >> 4, j = t & 15;
    A[i * DLD + 48 + j] = (i == j) ? 1.0 : 0.0;
  }
}

// one 64x64 block of K(X,X) + noise I: entry (I0 + i, J0 + j) -> T[i][j]; rows / columns >= N are the identity.
// x (z .* w)' on MFMA exactly as ksx_kernel forms it: per 16x16 sub-tile the A fragments are the raw rows, the B fragments the
// raw columns' rows times w (the product rounded once, as prep_obs_kernel rounds z .* w), and a chain of v_mfma_f64_16x16x4
// over the eight k-steps is the ascending fma chain over the 32 (zero padded) dimensions.
// The sub-tiles are dealt round-robin to the four waves.  A diagonal block (I0 == J0) gets its ten sub-tiles on and below
// the diagonal only (diag_core reads nothing above: potrf_diag.h:87); sub-tiles that lie wholly in the padding are written,
// not computed -- early in a run (N of a few dozen) that is most of them.
__device__ __forceinline__ void k_block(const double *__restrict__ obs, const double *__restrict__ w, const double *__restrict__ hn,
                                        const double *__restrict__ tab, int I0, int J0, int N, double noise,
                                        double *__restrict__ T) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4;
  const bool diag = I0 == J0;
  const int ntiles = diag ? 10 : 16;
  double wq[8];
#pragma unroll
  for (int k4 = 0; k4 < 8; ++k4) wq[k4] = w[4 * k4 + lq];
  for (int q = wave; q < ntiles; q += 4) {
    int it, jt;
    if (diag) {
      it = (q >= 1) + (q >= 3) + (q >= 6);
      jt = q - ((it * (it + 1)) >> 1);
    } else {
      it = q >> 2;
      jt = q & 3;
    }
    const int gj = J0 + 16 * jt + lr;
    if (I0 + 16 * it >= N || J0 + 16 * jt >= N) {  // wave-uniform: all of this sub-tile is padding
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * it + lq + 4 * r;
        T[i * DLD + 16 * jt + lr] = (I0 + i == gj) ? 1.0 : 0.0;
      }
      continue;
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
#ifdef B7_NLL_POISON
      if (diag && jt == it && lr > lq + 4 * r) v = __builtin_nan("");  // test build: nothing above the diagonal is read
#endif
      T[i * DLD + 16 * jt + lr] = v;
    }
  }
#ifdef B7_NLL_POISON
  if (diag)
    for (int e = threadIdx.x; e < NB * NB; e += 256)
      if (((e & 63) >> 4) > ((e >> 6) >> 4)) T[(e >> 6) * DLD + (e & 63)] = __builtin_nan("");
#endif
}

// X = 0 (all of [64][DLD]) with a fixed trip count, so the stores go out back to back
__device__ __forceinline__ void zero_block(double *X) {
#pragma unroll
  for (int t = 0; t < (NB * DLD + 255) / 256; ++t) {
    const int e = threadIdx.x + 256 * t;
    if (e < NB * DLD) X[e] = 0.0;
  }
}

struct NllSmallInline {  // the hypers of a single evaluation, passed in the kernel arguments (B == 1: no second trip over the bus)
  double v[35];
};

__global__ void __launch_bounds__(256)
    nll_small_kernel(const double *__restrict__ xobs, const double *__restrict__ y, int N, int d, const double *__restrict__ hyp_mem,
                     int B, double *__restrict__ terms, int *__restrict__ info, unsigned *__restrict__ done, NllSmallInline hin,
                     int use_inline) {
  extern __shared__ __align__(16) double sm[];
#ifdef B7_NLL_STAMP
  unsigned long long st_[16];
  int sn_ = 0;
#define STAMP() st_[sn_++] = __builtin_amdgcn_s_memtime()
#else
#define STAMP()
#endif
  STAMP();
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  const int npad = N > 64 ? 128 : 64;
  double *obs = sm;                    // [128][OLD], later X = inv(L_pp) [64][DLD]
  double *X = sm;
  double *A11 = sm + NB * DLD;         // [64][DLD]
  double *A21 = A11 + NB * DLD;
  double *A22 = A21 + NB * DLD;
  double *T = A22 + NB * DLD;          // [32][TLD]
  double *r = T + 32 * TLD;            // [128] residual, then z
  double *hn = r + 128;                // [128] half norms
  double *z = hn + 128;                // [128]
  double *dg = z + 128;                // [128] diagonal of L
  double *w = dg + 128;                // [32]
  double *tab = w + 32;                // [128]
  double *red = tab + 128;             // [512]
  __shared__ int inf[4];               // diag_core's failure report; thread 0 hands it to the host at the end
  const double *hyp = use_inline ? hin.v : hyp_mem;
  const double *ls = hyp + (size_t)b * d;
  const double amp = hyp[(size_t)B * d + b], noise = hyp[(size_t)B * (d + 1) + b], mean = hyp[(size_t)B * (d + 2) + b];
  if (tid < 4) inf[tid] = 0;
  if (tid < 32) w[tid] = tid < d ? 1.0 / ls[tid] : 0.0;  // inv_ls = ones:cdiv(lenscale), utils/math.lua:72
  if (tid < 128) {
    tab[tid] = amp * exp2_tab_small[tid];
    r[tid] = tid < N ? y[tid] - mean : 0.0;
  }
  {
    // the N x d observations are one contiguous block: sixteen coalesced loads per thread, all in flight at once (a load per
    // loop iteration behind a condition was most of this kernel's time outside the factor routine), then the scatter
    // into the zero-padded [128][OLD] image
    double v[16];
    const int total = N * d;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int idx = tid + 256 * t;
      v[t] = idx < total ? xobs[idx] : 0.0;
    }
    for (int e = tid; e < 128 * OLD; e += 256) obs[e] = 0.0;
    __syncthreads();
    STAMP();  // 1
    const float rd = 1.0f / (float)d;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int idx = tid + 256 * t;
      if (idx < total) {
        int i = (int)((float)idx * rd);      // idx / d for idx < 4096, d <= 32: the estimate is off by at most one
        i += (i + 1) * d <= idx;
        i -= i * d > idx;
        obs[i * OLD + (idx - i * d)] = v[t];
      }
    }
  }
  __syncthreads();
  STAMP();  // 2
  if (tid < 128) {
    double s = 0.0;
    for (int k = 0; k < 32; ++k) {
      const double x = obs[tid * OLD + k];
      s += (x * x) * w[k];  // Z_ss = (Z.^2) * inv_ls, :79 (the padding's half norm is never used: those entries are set)
    }
    hn[tid] = 0.5 * s;
  }
  __syncthreads();
  STAMP();  // 3
  k_block(obs, w, hn, tab, 0, 0, N, noise, A11);
  if (npad == 128) {
    k_block(obs, w, hn, tab, 64, 0, N, noise, A21);
    k_block(obs, w, hn, tab, 64, 64, N, noise, A22);
  }
  __syncthreads();  // the observation image is dead: its place becomes X
  STAMP();  // 4
  zero_block(X);
  identity_corner(A11);
  __syncthreads();
  STAMP();  // 5
#ifdef B7_NLL_STAMP
  __shared__ unsigned long long dst_[24];  // the factor routine's own phase stamps (slots 2..17), first block
  diag_core<1, true>(A11, X, T, 0, inf, dst_);
#else
  diag_core<1, false>(A11, X, T, 0, inf, nullptr);  // A11 -> L11 (lower), X = inv(L11); ends with a barrier
#endif
  STAMP();  // 6
  // z1 = inv(L11) r1 (four lanes per row, ascending columns within each quarter, then the quarters in order)
  {
    const int row = tid >> 2, part = tid & 3;
    double acc = 0.0;
    for (int k = 16 * part; k < 16 * part + 16; ++k) acc = __builtin_fma(X[row * DLD + k], r[k], acc);
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (part == 0) {
      z[row] = acc;
      dg[row] = A11[row * DLD + row];
    }
  }
  __syncthreads();
  STAMP();  // 7
  if (npad == 128) {
    // L21 = A21 inv(L11)': wave w rows 16 w .., four column blocks, k ascending, blocks above inv(L11)'s diagonal skipped
    d4_t lv[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    {
      const double *ap = A21 + (wave * 16 + lr) * DLD + lq, *xp = X + lr * DLD + lq;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const double aq = ap[4 * t];
#pragma unroll
        for (int jb = t >> 2; jb < 4; ++jb) lv[jb] = mfma_f64(aq, xp[jb * 16 * DLD + 4 * t], lv[jb]);
      }
    }
    __syncthreads();  // every wave is done reading A21
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) A21[(wave * 16 + lq + 4 * rr) * DLD + jb * 16 + lr] = lv[jb][rr];
    __syncthreads();
    STAMP();  // 8
    // A22 -= L21 L21': per 16x16 sub-tile the 64-deep chain from zero, then the subtraction; r2 -= L21 z1
    {
      d4_t u[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
      const double *ar = A21 + (16 * wave + lr) * DLD + lq;
#pragma unroll
      for (int k4 = 0; k4 < 16; ++k4) {
        const double af = ar[4 * k4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) u[jb] = mfma_f64(af, A21[(16 * jb + lr) * DLD + lq + 4 * k4], u[jb]);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
          const int e = (16 * wave + lq + 4 * rr) * DLD + 16 * jb + lr;
          A22[e] = A22[e] - u[jb][rr];
        }
      const int row = tid >> 2, part = tid & 3;
      double acc = 0.0;
      for (int k = 16 * part; k < 16 * part + 16; ++k) acc = __builtin_fma(A21[row * DLD + k], z[k], acc);
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      if (part == 0) r[64 + row] = r[64 + row] - acc;
    }
    __syncthreads();
    zero_block(X);
    identity_corner(A22);
    __syncthreads();
    STAMP();  // 9
    diag_core<1, false>(A22, X, T, 1, inf, nullptr);
    STAMP();  // 10
    {
      const int row = tid >> 2, part = tid & 3;
      double acc = 0.0;
      for (int k = 16 * part; k < 16 * part + 16; ++k) acc = __builtin_fma(X[row * DLD + k], r[64 + k], acc);
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      if (part == 0) {
        z[64 + row] = acc;
        dg[64 + row] = A22[row * DLD + row];
      }
    }
    __syncthreads();
    STAMP();  // 11
  }
  // |z|^2 and sum log L_ii in a fixed order: a butterfly inside each wave, then the waves in order
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
  if (lane == 0) {
    red[wave] = ssq;
    red[4 + wave] = ld;
  }
  __syncthreads();
  STAMP();  // last
  if (tid == 0) {
#ifdef B7_NLL_STAMP
    // workgroup b reports stamp b; workgroups 16.. the factor routine's slots 2.. relative to its entry (experiment build only)
    terms[2 * b] = b < 16 ? (b < sn_ ? (double)(st_[b] - st_[0]) : -1.0) : (b < 32 ? (double)(dst_[b - 14] - st_[5]) : -1.0);
    terms[2 * b + 1] = 0.0;
#else
    terms[2 * b] = (red[0] + red[1]) + (red[2] + red[3]);
    terms[2 * b + 1] = (red[4] + red[5]) + (red[6] + red[7]);
#endif
    for (int k = 0; k < 4; ++k) info[4 * b + k] = inf[k];
    // a single evaluation's caller spins on this word instead of waiting for the dispatch to retire: everything above was
    // written by this thread, and the release orders it before the flag as the host sees them
    if (done) __hip_atomic_store(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int ensure_small_table(b7_ctx *c) {
  static bool done[64] = {false};
  if (c->device < 64 && done[c->device]) return B7_OK;
  B7_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(exp2_tab_small), b7_exp2_tab, sizeof(b7_exp2_tab)));
  if (c->device < 64) done[c->device] = true;
  return B7_OK;
}

}  // namespace

bool nll_small_applies(const b7_ctx *c) { return c->Npad <= 128 && c->dfit <= 32 && c->ycols == 1; }

// hyp_dev: [B x d lengthscales | B amp | B noise | B mean] (b7_gp_nll_batch's pack); terms_dev[2 B], info_dev[4 B]
// done_dev (nullable): a word the kernel sets to 1 after its results are visible to the host (B == 1 only).
// hyp_host: the same pack in host memory; a single evaluation's hypers travel in the kernel arguments instead.
int launch_nll_small(b7_ctx *c, int B, const double *hyp_dev, const double *hyp_host, double *terms_dev, int *info_dev,
                     unsigned *done_dev) {
  PhaseScope ps(c, "potrf");
  B7_TRY(ensure_small_table(c));
  const size_t lds = sizeof(double) * NLL_SMALL_LDS_DOUBLES;
  // the opt-in to > 64 KiB of dynamic LDS is per device: once per process AND device (a process may hold contexts on several)
  static bool attr_done[64] = {false};
  if (c->device >= 64 || !attr_done[c->device]) {
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(nll_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (c->device < 64) attr_done[c->device] = true;
  }
  NllSmallInline hin = {};
  const int use_inline = (B == 1 && hyp_host != nullptr) ? 1 : 0;
  for (int k = 0; use_inline && k < c->dfit + 3; ++k) hin.v[k] = hyp_host[k];
  hipLaunchKernelGGL(nll_small_kernel, dim3(B), dim3(256), lds, c->stream, (const double *)c->xobs.p, (const double *)c->ybuf.p, c->N,
                     c->dfit, hyp_dev, B, terms_dev, info_dev, B == 1 ? done_dev : nullptr, hin, use_inline);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}
