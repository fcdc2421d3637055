// ARD squared-exponential covariance assembly: K(X,X) and K(X*,X).
//
// Reference arithmetic replaced (the kernel object itself, gp.kernels.ardse, is in the absent `gp` package;
// the in-repo statement of its distance is utils.math.pdist):
//   utils/math.lua:65-111   D = X^2 w (+) (Z^2 w)' - 2 X (Z' .* w),  w = 1/lenscale,  clamp(0, huge) (:106)
//   K = amp * exp(-D/2)     (+ noise on the diagonal of K(X,X))
//   mean part of model:predict (scores/expected_improvement.lua:63): mu = m + K(X*,X) alpha
//
// Layout: the inner product X (Z' .* w) runs on v_mfma_f64_16x16x4_f64 with the query rows as the A operand
// and the pre-scaled observations (z .* w, zero-padded to the class dpad in {4,8,16,32,48,64,96}) as B.  A block
// owns 64 query rows and walks observations in slabs of 64 (32 for dpad >= 48), double-buffered in LDS with the
// next slab prefetched into registers while the current one is consumed (one barrier per slab), so it sees whole
// rows of K(X*,X): the posterior-mean dot product with alpha is accumulated in registers on the way and needs no
// second pass over K*.
// Epilogue per element: arg = (c - xs/2) - zs/2 (the reference's ((-2c + xs) + zs) scaled by the exact factor
// -1/2, same rounding points), clamp to <= 0 (NaN passes, as TH's clamp), amp * exp(arg).  Lane pairs swap one
// value (DPP quad_perm) so every lane stores 16 bytes: 2 stores per 16x16 tile instead of 4.
// Padding observations carry zs/2 = +inf, which makes their covariance exactly 0.
//
// Priced by ablation on MI355X (tools/ksx_ab.py, 262144 x 2048, d = 32) before this structure: full 2.09 ms,
// without stores 1.49, without exp 2.02, without MFMA 1.74 -> the old kernel was bound by its own staging
// (integer division per element, two barriers per slab, no prefetch), not by arithmetic or HBM.  Now 1.06 ms =
// 4.0 TB/s (51 % of HBM peak) at d = 32; 58 % at d <= 6, 40 % at d = 39, 32 % at d = 64 (tools/ksx_rate.py).
#include "b7_internal.h"
#include "gemm_f64.h"
#include "ksx_exp.h"

namespace {

__constant__ double exp2_tab_dev[128];  // b7_exp2_tab, uploaded once per process (ensure_exp_table)

// ---- observation pre-scaling: zsc = z .* w (padded), zsh = (sum z^2 w)/2 ---------------------------------------
// One wave per 64 observations: the rows are loaded and stored through an LDS tile (coalesced both ways; a thread
// walking its own row in global memory was 16 us at N = 2048), the weights 1/ls are formed once per workgroup, and
// each lane sums its row in ascending k exactly as before (same bits).
__global__ void __launch_bounds__(64) prep_obs_kernel(const double *__restrict__ xobs, const double *ls, double *w,
                                                      double *zsc, double *zsh, int N, int Npad, int d, int dpad) {
  extern __shared__ __align__(16) double psm[];
  {  // batch of fits over the same observations (b7_gp_nll_batch): blockIdx.y selects the hyper vector
    const int64_t b = blockIdx.y;
    ls += b * d;
    w += b * dpad;
    zsc += b * (int64_t)Npad * dpad;
    zsh += b * Npad;
  }
  const int tld = dpad + 1, lane = threadIdx.x, row0 = blockIdx.x * 64;
  double *tile = psm, *wl = psm + 64 * tld;
  for (int k = lane; k < dpad; k += 64) {
    const double wk = (k < d) ? 1.0 / ls[k] : 0.0;  // inv_ls = ones:cdiv(lenscale), utils/math.lua:72
    wl[k] = wk;
    if (blockIdx.x == 0) w[k] = wk;
  }
  int nrows = N - row0;
  nrows = nrows < 0 ? 0 : (nrows > 64 ? 64 : nrows);
  const double *src = xobs + (int64_t)row0 * d;
#pragma unroll 8
  for (int e = lane; e < nrows * d; e += 64) {  // unrolled: eight loads in flight (a load -> wait -> LDS store loop was 21 us)
    const int r = e / d, k = e - r * d;
    tile[r * tld + k] = src[e];
  }
  __syncthreads();
  double s = 0.0;
#pragma unroll 8
  for (int k = 0; k < dpad; ++k) {
    const double z = (lane < nrows && k < d) ? tile[lane * tld + k] : 0.0;
    const double wk = wl[k];
    tile[lane * tld + k] = z * wk;
    s += (z * z) * wk;  // Z_ss = (Z.^2) * inv_ls, :79
  }
  if (row0 + lane < Npad) zsh[row0 + lane] = (lane < nrows) ? 0.5 * s : 1e300;  // padding: covariance exactly 0 (amp_exp_nonpos)
  __syncthreads();
  double *dst = zsc + (int64_t)row0 * dpad;
#pragma unroll 8
  for (int e = lane; e < 64 * dpad; e += 64) {
    const int r = e / dpad, k = e - r * dpad;
    if (row0 + r < Npad) dst[e] = tile[r * tld + k];
  }
}

// out[0] = even lane ? a : (lane ^ 1)'s b;  out[1] = odd lane ? b : (lane ^ 1)'s a.  Select and lane exchange are ONE
// instruction per 32-bit half, v_cndmask_b32_dpp (VOP2: D = VCC ? src1 : dpp(src0)), which hipcc does not form from
// update_dpp + a select (it emitted v_mov_b32_dpp + 3 v_cndmask per half: 20 VALU instructions per 16x16 tile against
// 8 here, and the kernel is bound by VALU + MFMA issue).  The two s_mov between the producers of a / b and the first DPP
// read are the 2 wait states that hazard needs (VALU writes VGPR -> DPP reads it); VCC is rewritten, hence the clobber.
__device__ __forceinline__ void pair_pack(double a, double b, d2_t &out) {
  const int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  int o0l, o0h, o1l, o1h;
  asm volatile(
      "s_mov_b32 vcc_lo, 0x55555555\n\t"
      "s_mov_b32 vcc_hi, 0x55555555\n\t"
      "s_nop 0\n\t"
      "v_cndmask_b32_dpp %0, %6, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %1, %7, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_not_b64 vcc, vcc\n\t"
      "v_cndmask_b32_dpp %2, %4, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %3, %5, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "=&v"(o0l), "=&v"(o0h), "=&v"(o1l), "=&v"(o1h)
      : "v"(alo), "v"(ahi), "v"(blo), "v"(bhi)
      : "vcc", "scc");
  out[0] = __hiloint2double(o0h, o0l);
  out[1] = __hiloint2double(o1h, o1l);
}

// ---- K(X*,X) chunk / K(X,X) -----------------------------------------------------------------------------------
// xq: query rows, row-major with d columns; rows [row0, row0+rows) of a set with Mtotal rows (rows beyond
// Mtotal-1 are clamped: they produce values nobody reads).  out: rows x Npad (leading dimension Npad).
// mu (nullable): mean + K* alpha for ycols == 1.  blockIdx.y splits the observations into gridDim.y equal
// ranges of whole slabs (used for K(X,X), where there are few query rows; mu must then be null).
constexpr int KQ = 64;  // query rows per block (16 per wave)
// observation slab: 64 rows, 32 for the wide classes so that two blocks still fit a CU's LDS
// (measured at DPAD = 64 with 64-row slabs: 1 block/CU, 2.07 TB/s against 3.97 TB/s at DPAD = 32)
__host__ __device__ constexpr int ksx_slab(int dpad) { return dpad >= 48 ? 32 : 64; }

// DPAD = padded input dimension, one of the classes {4, 8, 16, 32, 48, 64, 96} (b7_dpad_class): compile-time so that
// the MFMA chain over DPAD/4 k-steps unrolls and the query fragments stay in registers for the whole block.
// blockIdx.z = fit index of a batch over the same observations (b7_gp_nll_batch); all zero / null for a single fit
using KBatch = KBatchDesc;
template <int DPAD>
__global__ void __launch_bounds__(256)
    ksx_kernel(const double *__restrict__ xq, int64_t row0, int64_t Mtotal, int d, int dpad_rt, const double *w,
               const double *zsc, const double *zsh, const double *__restrict__ alpha, double amp, double meanc, int Npad,
               double *out, double *__restrict__ mu, KBatch kb) {
  extern __shared__ __align__(16) double sm[];
  if (kb.amp) {
    const int64_t bz = blockIdx.z;
    w += bz * kb.s_w;
    zsc += bz * kb.s_zsc;
    zsh += bz * kb.s_zsh;
    out += bz * kb.s_out;
    amp = kb.amp[bz];
    if (kb.mean) {  // the batch carries its own alpha / mean / mu per fit
      alpha += bz * kb.s_alpha;
      mu += bz * kb.s_mu;
      meanc = kb.mean[bz];
    }
  }
  constexpr int KO = ksx_slab(DPAD);
  constexpr int TPR = 256 / KO;                       // staging threads per slab row
  constexpr int dpad = DPAD, NCH = (DPAD / 2 + TPR - 1) / TPR, KSTEPS = DPAD / 4;
  constexpr int stride = DPAD + 1;  // odd: conflict-free for the fused ds_read2_b64 fragment reads (gemm_f64.h)
  (void)dpad_rt;
  // The query tile is dead once its fragments sit in registers (before the first slab is consumed), so the two slab buffers
  // live in the SAME LDS region: 50 -> 34 KB per workgroup at DPAD 32, 67 -> 34 at 64, 99 -> 50 at 96, i.e. three workgroups
  // per CU (the register file's limit) instead of two at d = 64 and one at d = 96.  Costs two barriers in the prologue.
  constexpr int R0 = (KQ * stride > 2 * KO * stride) ? KQ * stride : 2 * KO * stride;
  double *sq = sm;                          // KQ x stride, prologue only
  double *so = sm;                          // 2 x KO x stride, from the first slab on
  double *sh = sm + R0;                     // 2 x KO   zs/2 of the slab
  double *sal = sh + 2 * KO;                // 2 x KO   alpha of the slab
  double *shq = sal + 2 * KO;               // KQ       xs/2 of the queries
  double *stab = shq + KQ;                  // 128      amp * 2^(j/128)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int64_t qbase = row0 + (int64_t)blockIdx.x * KQ;
  const int srow = tid / TPR, sq4 = tid % TPR;  // slab staging: TPR threads per row
  const int qrow = tid >> 2, qq4 = tid & 3;     // query staging: 64 rows, 4 threads per row
  constexpr int half = DPAD >> 1;            // 16-byte chunks per row

  if (tid < 128) stab[tid] = amp * exp2_tab_dev[tid];
  // query tile (zero-padded columns)
  {
    int64_t g = qbase + qrow;
    if (g > Mtotal - 1) g = Mtotal - 1;
    for (int k = qq4; k < dpad; k += 4) sq[qrow * stride + k] = (k < d) ? xq[g * d + k] : 0.0;
  }
  const int nslab_total = Npad / KO;
  const int nslab = nslab_total / gridDim.y;
  const int slab0 = blockIdx.y * nslab;

  d2_t pre[NCH];
  double pre_h = 0.0, pre_a = 0.0;
  auto load_slab = [&](int s) {
    const double *src = zsc + (int64_t)(s * KO + srow) * dpad;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int kc = i * TPR + sq4;
      if (kc < half) pre[i] = *reinterpret_cast<const d2_t *>(src + 2 * kc);
    }
    if (tid < KO) {
      pre_h = zsh[s * KO + tid];
      pre_a = alpha ? alpha[s * KO + tid] : 0.0;
    }
  };
  auto store_slab = [&](int buf) {
    double *dst = so + buf * KO * stride + srow * stride;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int kc = i * TPR + sq4;
      if (kc < half) {
        dst[2 * kc] = pre[i][0];
        dst[2 * kc + 1] = pre[i][1];
      }
    }
    if (tid < KO) {
      sh[buf * KO + tid] = pre_h;
      sal[buf * KO + tid] = pre_a;
    }
  };

  load_slab(slab0);
  __syncthreads();  // query tile visible
  if (tid < KQ) {
    double s = 0.0;
    for (int k = 0; k < dpad; ++k) {
      double x = sq[tid * stride + k];
      s += (x * x) * w[k];  // X_ss = (X.^2) * inv_ls, :78
    }
    shq[tid] = 0.5 * s;
  }
  double qf[KSTEPS];  // this lane's A fragments: query row (wave*16 + lr), k = 4 k4 + lq
#pragma unroll
  for (int k4 = 0; k4 < KSTEPS; ++k4) qf[k4] = sq[(wave * 16 + lr) * stride + lq + 4 * k4];
  __syncthreads();  // every wave holds its fragments, shq is complete: the region now belongs to the slabs
  store_slab(0);
  __syncthreads();

  double hq[4], macc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r) hq[r] = shq[wave * 16 + lq + 4 * r];
  // this lane's 16-byte store slots: even lanes write rows r = 0 and 2, odd lanes rows r = 1 and 3, two columns
  const bool odd = lane & 1;
  double *orow = out + ((int64_t)blockIdx.x * KQ + wave * 16 + lq + (odd ? 4 : 0)) * Npad + (lr & ~1);

  int cur = 0;
  for (int s = 0; s < nslab; ++s) {
    const bool more = (s + 1) < nslab;
    if (more) load_slab(slab0 + s + 1);
    const double *sob = so + cur * KO * stride;
    const int o0 = (slab0 + s) * KO;
#pragma unroll
    for (int t = 0; t < KO / 16; ++t) {
      const double *ob = sob + (t * 16 + lr) * stride + lq;
      d4_t c = {0.0, 0.0, 0.0, 0.0};
      {
        double bf[KSTEPS];
#pragma unroll
        for (int k4 = 0; k4 < KSTEPS; ++k4) bf[k4] = ob[4 * k4];
#if B7_KSX_ABLATE & 8   // one MFMA per tile instead of DPAD / 4
        c = mfma_f64(qf[0], bf[0], c);
#else
#pragma unroll
        for (int k4 = 0; k4 < KSTEPS; ++k4) c = mfma_f64(qf[k4], bf[k4], c);
#endif
      }
      const double hk = sh[cur * KO + t * 16 + lr];
      const double al = sal[cur * KO + t * 16 + lr];
      double kv[4], arg[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) arg[r] = (c[r] - hq[r]) - hk;  // = -1/2 * (((-2 c) + xs) + zs), utils/math.lua:82
#if B7_KSX_ABLATE & 4   // diagnostic builds only (tools/ksx_ablate.py): no exp
#pragma unroll
      for (int r = 0; r < 4; ++r) kv[r] = arg[r];
#else
      amp_exp_nonpos4(arg, stab, kv);  // :106 clamp(0, huge) on the distance (NaN passes), amp * exp(-D/2)
#endif
#pragma unroll
      for (int r = 0; r < 4; ++r) macc[r] = __builtin_fma(kv[r], al, macc[r]);
      // rows (lq, lq+4, lq+8, lq+12) x column lr  ->  16-byte stores of two adjacent columns: even lanes write rows r = 0
      // and 2 (their own value, then the odd neighbour's), odd lanes rows r = 1 and 3 (the even neighbour's, then their own)
      d2_t v01, v23;
      pair_pack(kv[0], kv[1], v01);
      pair_pack(kv[2], kv[3], v23);
      {
        double *p = orow + o0 + t * 16;
#if B7_KSX_ABLATE & 1   // no stores (a condition that never holds keeps the values alive)
        if (v01[0] == 1.2345e-300) {
#endif
#if B7_KSX_ABLATE & 16  // the same bytes to row-contiguous addresses (2 rows x 512 B per instruction; wrong placement)
        double *pi = out + ((int64_t)blockIdx.x * KQ + wave * 16 + 4 * t + (lane >> 5)) * Npad + o0 + (lane & 31) * 2;
        *reinterpret_cast<d2_t *>(pi) = v01;
        *reinterpret_cast<d2_t *>(pi + (int64_t)2 * Npad) = v23;
#elif B7_KSX_ABLATE & 32  // non-temporal stores
        __builtin_nontemporal_store(v01, reinterpret_cast<d2_t *>(p));
        __builtin_nontemporal_store(v23, reinterpret_cast<d2_t *>(p + (int64_t)8 * Npad));
#else
        *reinterpret_cast<d2_t *>(p) = v01;
        *reinterpret_cast<d2_t *>(p + (int64_t)8 * Npad) = v23;
#endif
#if B7_KSX_ABLATE & 1
        }
#endif
      }
    }
    if (more) store_slab(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  if (mu) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = macc[r];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      const int64_t g = qbase + wave * 16 + lq + 4 * r;
      if (lr == 0 && g < Mtotal) mu[g] = meanc + v;
    }
  }
}

// K(X,X) post-pass: padding rows/columns become identity, `diag_add` (noise) goes on the diagonal.
__global__ void __launch_bounds__(256)
    kxx_fix_kernel(double *K, int N, int Npad, double diag_add, const double *diag_arr, int64_t sK) {
  if (diag_arr) {  // batch: blockIdx.y = fit index
    K += (int64_t)blockIdx.y * sK;
    diag_add = diag_arr[blockIdx.y];
  }
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)Npad * Npad) return;
  int i = (int)(e / Npad), j = (int)(e - (int64_t)i * Npad);
  if (i >= N || j >= N)
    K[e] = (i == j) ? 1.0 : 0.0;
  else if (i == j)
    K[e] += diag_add;
}

size_t ksx_lds_bytes(int dpad) {
  const int KO = ksx_slab(dpad);
  const size_t r0 = (size_t)(KQ > 2 * KO ? KQ : 2 * KO) * (dpad + 1);  // query tile and slab buffers share a region
  size_t bytes = sizeof(double) * (r0 + 4 * KO + KQ + 128);
#if B7_KSX_ABLATE & 128   // diagnostic: pad the request so that one workgroup fewer fits a CU (sensitivity to occupancy)
  const size_t per_cu = 160 * 1024, fit = per_cu / bytes;
  if (fit > 1) bytes = per_cu / (fit - 1) - 512;
#endif
  return bytes;
}

// the 2^(j/128) table goes to constant memory once per process and device
int ensure_exp_table(b7_ctx *c) {
  static bool done[64] = {false};
  if (c->device < 64 && done[c->device]) return B7_OK;
  B7_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(exp2_tab_dev), b7_exp2_tab, sizeof(b7_exp2_tab)));
  if (c->device < 64) done[c->device] = true;
  return B7_OK;
}

template <int DPAD>
int ksx_launch(b7_ctx *c, dim3 grid, const double *xq, int64_t row0, int64_t Mtotal, int d, const ObsSet &o,
               const double *alpha, double meanc, double *out, double *mu) {
  const size_t lds = ksx_lds_bytes(DPAD);
  auto kern = ksx_kernel<DPAD>;
  B7_TRY(ensure_exp_table(c));
  // dynamic LDS above the 64 KiB default needs an explicit opt-in (gfx950 has 160 KiB per workgroup)
  B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, c->stream, xq, row0, Mtotal, d, c->dpad, o.w ? o.w : (const double *)c->w.p,
                     o.zsc, o.zsh, alpha, c->amp, meanc, o.npad, out, mu, o.batch);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int ksx_dispatch(b7_ctx *c, dim3 grid, const double *xq, int64_t row0, int64_t Mtotal, int d, const ObsSet &o,
                 const double *alpha, double meanc, double *out, double *mu) {
  switch (c->dpad) {
    case 4: return ksx_launch<4>(c, grid, xq, row0, Mtotal, d, o, alpha, meanc, out, mu);
    case 8: return ksx_launch<8>(c, grid, xq, row0, Mtotal, d, o, alpha, meanc, out, mu);
    case 16: return ksx_launch<16>(c, grid, xq, row0, Mtotal, d, o, alpha, meanc, out, mu);
    case 32: return ksx_launch<32>(c, grid, xq, row0, Mtotal, d, o, alpha, meanc, out, mu);
    case 48: return ksx_launch<48>(c, grid, xq, row0, Mtotal, d, o, alpha, meanc, out, mu);
    case 64: return ksx_launch<64>(c, grid, xq, row0, Mtotal, d, o, alpha, meanc, out, mu);
    case 96: return ksx_launch<96>(c, grid, xq, row0, Mtotal, d, o, alpha, meanc, out, mu);
    default: return b7_fail(c, B7_ERR_UNSUPPORTED, "covariance kernel: dpad %d is not a built class", c->dpad);
  }
}

}  // namespace

int launch_prep_obs(b7_ctx *c, const double *xobs, const double *ls_dev, int N, int d) {
  PhaseScope ps(c, "prep");
  const size_t lds = sizeof(double) * (64 * (size_t)(c->dpad + 1) + c->dpad);
  hipLaunchKernelGGL(prep_obs_kernel, dim3((c->Npad + 63) / 64), dim3(64), lds, c->stream, xobs, ls_dev,
                     (double *)c->w.p, (double *)c->zsc.p, (double *)c->zss.p, N, c->Npad, d, c->dpad);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

// A second observation set (the pending points of fantasize) scaled with the CURRENT lengthscales; w is rewritten
// with the same values.
int launch_prep_obs_aux(b7_ctx *c, const double *xobs, const double *ls_dev, int N, int npad, double *zsc,
                        double *zsh) {
  const size_t lds = sizeof(double) * (64 * (size_t)(c->dpad + 1) + c->dpad);
  hipLaunchKernelGGL(prep_obs_kernel, dim3((npad + 63) / 64), dim3(64), lds, c->stream, xobs, ls_dev, (double *)c->w.p,
                     zsc, zsh, N, npad, c->dfit, c->dpad);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

// K(Xq, Xobs-set) for an arbitrary observation set: rows x o.npad, no mean.
int launch_k_generic(b7_ctx *c, const double *xq, int64_t rows, int64_t Mtotal, const ObsSet &o, double *out) {
  if (rows % KQ || o.npad % 64) return b7_fail(c, B7_ERR_INVALID, "k_generic: extents not multiples of 64");
  return ksx_dispatch(c, dim3((unsigned)(rows / KQ), 1), xq, 0, Mtotal, c->dfit, o, nullptr, 0.0, out, nullptr);
}

int launch_ksx(b7_ctx *c, const double *xq, int64_t row0, int64_t rows, int64_t Mtotal, int d, double *ks,
               double *mu, int ycols) {
  PhaseScope ps(c, "ksx");
  if (rows % KQ) return b7_fail(c, B7_ERR_INVALID, "ksx: rows %lld not a multiple of %d", (long long)rows, KQ);
  if (ycols != 1) mu = nullptr;  // the multi-column mean is a GEMM of its own (launch_mean_multi)
  const ObsSet o{(const double *)c->zsc.p, (const double *)c->zss.p, c->Npad};
  return ksx_dispatch(c, dim3((unsigned)(rows / KQ), 1), xq, row0, Mtotal, d, o,
                      mu ? (const double *)c->alpha.p : nullptr, c->mean, ks, mu);
}

int launch_kxx(b7_ctx *c, double diag_add) {
  PhaseScope ps(c, "kxx");
  const int Npad = c->Npad;
  // few query rows: split the observations over blockIdx.y so that the grid covers the chip
  int ny = 1;
  while (ny < 16 && (Npad / 64) % (ny * 2) == 0 && (Npad / KQ) * ny < 2 * c->cus) ny *= 2;
  const ObsSet o{(const double *)c->zsc.p, (const double *)c->zss.p, Npad};
  B7_TRY(ksx_dispatch(c, dim3(Npad / KQ, ny), (const double *)c->xobs.p, 0, c->N, c->dfit, o, nullptr, 0.0,
                      (double *)c->K.p, nullptr));
  int64_t total = (int64_t)Npad * Npad;
  hipLaunchKernelGGL(kxx_fix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, (double *)c->K.p,
                     c->N, Npad, diag_add, (const double *)nullptr, (int64_t)0);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

// ---- B fits over the same observations: per-fit scaled observations, then K_b = amp_b exp(-D_b / 2) + noise_b I ---------
// ls: B x d, amp / noise: B (device).  zsc: B x Npad x dpad, zss: B x Npad, w: B x dpad, K: B x Npad x Npad.
// K(X*, X) and the posterior mean of S fits over the same candidate rows in one launch (grid.z = fit): the hyper samples
// of a nomination whose K* fits the workspace S times over (b7_eval_nominate)
int launch_ksx_batch(b7_ctx *c, int S, const double *xq, int64_t rows, int64_t Mtotal, const double *w, const double *zsc,
                     const double *zss, const double *amp_dev, const double *mean_dev, const double *alpha, double *ks,
                     int64_t s_out, double *mu, int64_t s_mu) {
  PhaseScope ps(c, "ksx");
  if (rows % KQ) return b7_fail(c, B7_ERR_INVALID, "ksx: rows %lld not a multiple of %d", (long long)rows, KQ);
  ObsSet o{zsc, zss, c->Npad};
  o.w = w;
  o.batch.s_w = c->dpad;
  o.batch.s_zsc = (int64_t)c->Npad * c->dpad;
  o.batch.s_zsh = c->Npad;
  o.batch.s_out = s_out;
  o.batch.amp = amp_dev;
  o.batch.s_alpha = c->Npad;
  o.batch.s_mu = s_mu;
  o.batch.mean = mean_dev;
  return ksx_dispatch(c, dim3((unsigned)(rows / KQ), 1, S), xq, 0, Mtotal, c->dfit, o, alpha, 0.0, ks, mu);
}

int launch_kxx_batch(b7_ctx *c, int B, const double *ls_dev, const double *amp_dev, const double *noise_dev, double *w,
                     double *zsc, double *zss, double *K) {
  PhaseScope ps(c, "kxx");
  const int Npad = c->Npad, d = c->dfit, dpad = c->dpad;
  const size_t lds = sizeof(double) * (64 * (size_t)(dpad + 1) + dpad);
  hipLaunchKernelGGL(prep_obs_kernel, dim3((Npad + 63) / 64, B), dim3(64), lds, c->stream, (const double *)c->xobs.p, ls_dev,
                     w, zsc, zss, c->N, Npad, d, dpad);
  ObsSet o{zsc, zss, Npad};
  o.w = w;
  o.batch.s_w = dpad;
  o.batch.s_zsc = (int64_t)Npad * dpad;
  o.batch.s_zsh = Npad;
  o.batch.s_out = (int64_t)Npad * Npad;
  o.batch.amp = amp_dev;
  // as launch_kxx: with few fits the observations are split over blockIdx.y so that the grid still covers the chip
  int ny = 1;
  while (ny < 16 && (Npad / 64) % (ny * 2) == 0 && (Npad / KQ) * ny * B < 2 * c->cus) ny *= 2;
  B7_TRY(ksx_dispatch(c, dim3(Npad / KQ, ny, B), (const double *)c->xobs.p, 0, c->N, d, o, nullptr, 0.0, K, nullptr));
  const int64_t total = (int64_t)Npad * Npad;
  hipLaunchKernelGGL(kxx_fix_kernel, dim3((unsigned)((total + 255) / 256), B), dim3(256), 0, c->stream, K, c->N, Npad, 0.0,
                     noise_dev, total);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}
