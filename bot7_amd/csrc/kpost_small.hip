// K(X*,X), posterior mean AND posterior variance in one kernel for SMALL observation sets (N <= 128, d <= 32, one response
// column: the reference's own regime, bots/abstract.lua:64): K* never leaves the chip.
//
// Reference arithmetic replaced: model:predict(X_obs, Y_obs, X_hid, hyp, {mean, var}) (scores/expected_improvement.lua:63) --
//   K* = amp exp(-pdist(X*, X, ls) / 2)  (utils/math.lua:65-111),  mu = m + K* alpha,  var = amp - colsumsq(L^-1 K*').
// The general path materialises K* (ksx_kernel: 8 Npad bytes per candidate and fit out to HBM) and reads it back in the
// variance GEMM (post_kernel_w4 / post_small_kernel); at N = 100 with ten hyper samples over 2e4 candidates that is 37 + 82 us
// of a 230 us nomination, the second kernel at half the MFMA peak on 128-row tiles of which 28 rows are padding.
//
// Here a wave owns 16-candidate strips.  Per 16-observation strip J it forms the K*' tile on MFMA with the
// OBSERVATIONS as rows -- D[obs][cand] = sum_k (z w)[obs][k] x[cand][k], the products and the ascending chain of ksx_kernel
// with the operands swapped -- applies the same (c - xs/2) - zs/2 argument and table exponential (ksx_exp.h), and the four
// accumulator registers of that tile ARE the B operands of the four k-steps of  V[I] += L^-1[I][J] K*'[J]  (register r holds
// row (lane >> 4) + 4 r, which is what k-step r wants from lane group lane >> 4): no LDS round trip, no store.  L^-1's A
// fragments sit in LDS in fragment order (one conflict-free 8-byte read per lane and MFMA), for the row strips that hold
// real observations only: NT = ceil(N / 16) of them, so the work follows N, not its padding.
//
// Bits: those of the general path.  K* entries as above; mu: per (candidate, observation mod 16) the ascending fma chain over
// the strips, then ksx_kernel's butterfly (observation index bits 0, 1 across lane groups, bits 2, 3 across registers);
// v = ascending fma chain over k per element (what an MFMA chain computes), k up to the row strip's diagonal; sum of squares per
// lane group over rows 16 I + g + 4 r (I, then r ascending) for I = 0..3 and I = 4..7 apart, ((g0 + g1) + (g2 + g3)) each, first
// half + second half -- post_kernel_w4's and post_small_kernel's statement.  Skipped padding contributes exact zeros there.
#include "b7_internal.h"
#include "gemm_f64.h"
#include "ksx_exp.h"

namespace {

__constant__ double exp2_tab_kp[128];  // b7_exp2_tab (ensure_kp_table)

constexpr int KP_THREADS = 512;  // eight waves share one LDS copy of a fit; one workgroup per CU (two waves per SIMD)

struct KpArgs {
  const double *xq;   // M x d candidates (row-major)
  int64_t M;
  int d, N, npad, S;
  const double *w, *zsc, *zss, *Linv, *alpha;  // per fit: dpad | npad dpad | npad | npad^2 | npad
  const double *hyp;  // [S x d | S amp | S noise | S mean] on the device, or null: the scalars below (S == 1)
  double amp, noise, mean;
  double *mu, *var;   // per fit: M each, stride sout
  int64_t sout;
  int var_with_noise, clamp;
  double var_min;
};

__host__ __device__ constexpr int tri(int I) { return (I * (I + 1)) >> 1; }

// LDS (doubles): LF[tri(NT)][4][64] | ZF[NT][KS][64] | zs[16 NT] | al[16 NT] | wv[DPAD] | tab[128]
template <int DPAD>
__host__ __device__ constexpr int kp_lds_doubles(int nt) {
  return tri(nt) * 256 + nt * (DPAD / 4) * 64 + 32 * nt + DPAD + 128;
}

// two workgroups per CU (four waves per SIMD: <= 128 registers) for the narrow classes up to seven row strips -- where two LDS
// copies of a fit fit as well --: twice the waves to hide the exponential's dependent chain behind, and 16-candidate strips
// dealt in units half the size
__host__ __device__ constexpr int kp_waves_per_simd(int dpad, int nt) { return (dpad <= 8 && nt <= 7) ? 4 : 2; }

template <int DPAD, int NT>
__global__ void __launch_bounds__(KP_THREADS) __attribute__((amdgpu_waves_per_eu(kp_waves_per_simd(DPAD, NT), 4)))
kpost_small_kernel(KpArgs a) {
  constexpr int KS = DPAD / 4;
  extern __shared__ __align__(16) double sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
  const int s = blockIdx.y, npad = a.npad, d = a.d;  // NT = ceil(N / 16) row strips hold observations (the launcher picks the instance)
  double *LF = sm, *ZF = LF + tri(NT) * 256, *zs = ZF + NT * KS * 64, *al = zs + 16 * NT, *wv = al + 16 * NT, *tab = wv + DPAD;
  const double amp = a.hyp ? a.hyp[(size_t)a.S * d + s] : a.amp;
  const double noise = a.hyp ? a.hyp[(size_t)a.S * (d + 1) + s] : a.noise;
  const double meanc = a.hyp ? a.hyp[(size_t)a.S * (d + 2) + s] : a.mean;
  const double *Li = a.Linv + (size_t)s * npad * npad, *zc = a.zsc + (size_t)s * npad * DPAD;
  // ---- this fit into LDS, in fragment order: tile (I, J) of L^-1, k-step q, lane l <- L^-1[16 I + (l & 15)][16 J + 4 q + (l >> 4)]
  {
    const int t2 = tid & 255, row = t2 >> 4, c = t2 & 15, dst = (c >> 2) * 64 + (c & 3) * 16 + row;
    for (int t = tid >> 8; t < tri(NT); t += KP_THREADS / 256) {  // tile t = tri(I) + J: half the workgroup per tile
      int I = 0;
      while (tri(I + 1) <= t) ++I;
      const int J = t - tri(I);
      LF[t * 256 + dst] = Li[(size_t)(16 * I + row) * npad + 16 * J + c];
    }
    for (int e = tid; e < 16 * NT * DPAD; e += KP_THREADS) {  // (z w)[16 J + r][4 k4 + g] -> ZF[J][k4][16 g + r]
      const int i = e / DPAD, k = e - i * DPAD;
      ZF[((i >> 4) * KS + (k >> 2)) * 64 + (k & 3) * 16 + (i & 15)] = zc[e];
    }
    if (tid < 16 * NT) {
      zs[tid] = a.zss[(size_t)s * npad + tid];
      al[tid] = a.alpha[(size_t)s * npad + tid];
    }
    if (tid < DPAD) wv[tid] = a.w[(size_t)s * DPAD + tid];
    if (tid < 128) tab[tid] = amp * exp2_tab_kp[tid];
  }
  __syncthreads();
  const double var_add = a.var_with_noise ? noise : 0.0;
  double *mu = a.mu + (size_t)s * a.sout, *var = a.var + (size_t)s * a.sout;
  // ---- strips of 16 candidates, dealt round-robin to the waves of this fit's workgroups.  No barrier from here on: LDS is
  // read-only, every wave runs by itself.
  const int64_t nstrips = (a.M + 15) >> 4;
  const int64_t W = (int64_t)gridDim.x * (KP_THREADS / 64);
  // (wave-major numbering: the waves that get one strip more than the others -- the remainder of nstrips / W -- are wave 0 of
  // as many workgroups, not all eight waves of a few: no SIMD gets more than one extra strip)
  for (int64_t st = (int64_t)wave * gridDim.x + blockIdx.x; st < nstrips; st += W) {
    // the candidates' B fragments (x[cand = l & 15][4 k4 + (l >> 4)], zero padded) and half norm xs/2 (ksx_kernel: one
    // ascending sum per candidate; every lane computes its candidate's, the four lane groups redundantly)
    double xf[KS], hq;
    {
      int64_t g = st * 16 + lr;
      g = g < a.M ? g : a.M - 1;
      const double *px = a.xq + g * d;
      double sx = 0.0;
#pragma unroll
      for (int k = 0; k < DPAD; ++k) {
        const double v = k < d ? px[k] : 0.0;
        sx += (v * v) * wv[k];  // X_ss = (X.^2) * inv_ls, utils/math.lua:78
        if ((k & 3) == lq) xf[k >> 2] = v;
      }
      hq = 0.5 * sx;
    }
    d4_t acc[NT];
    double mac[4] = {0.0, 0.0, 0.0, 0.0};
    double ss[2] = {0.0, 0.0};
#pragma unroll
    for (int J = 0; J < NT; ++J) {
      // K*' tile J: rows = observations 16 J .., columns = candidates
      d4_t c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k4 = 0; k4 < KS; ++k4) c = mfma_f64(ZF[(J * KS + k4) * 64 + lane], xf[k4], c);
      // (alpha's four entries: read ahead of the exponentials up to six row strips, behind them from seven on, where the
      // registers they would hold across the chain are what keeps the instance at 128 and two workgroups on a CU)
      double hk[4], alj[4], arg[4], kv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) hk[r] = zs[16 * J + lq + 4 * r];
      if (NT <= 6) {
#pragma unroll
        for (int r = 0; r < 4; ++r) alj[r] = al[16 * J + lq + 4 * r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) arg[r] = (c[r] - hq) - hk[r];  // = -1/2 (((-2 c) + xs) + zs), utils/math.lua:82
      amp_exp_nonpos4(arg, tab, kv);
      // all four exponentials exist HERE: left alone, the optimiser sinks each one in front of the k-step that consumes it
      // and the four 14-deep chains run one after the other between the MFMAs
      asm volatile("" : "+v"(kv[0]), "+v"(kv[1]), "+v"(kv[2]), "+v"(kv[3]));
#pragma unroll
      for (int r = 0; r < 4; ++r) mac[r] = __builtin_fma(kv[r], NT <= 6 ? alj[r] : al[16 * J + lq + 4 * r], mac[r]);
      // V[I] += L^-1[I][J] K*'[J] for the row strips I >= J; register q of the tile is the B operand of k-step q.  The
      // fragments of a k-step first (NT - J independent reads), then its MFMAs (NT - J independent chains)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double af[NT];
#pragma unroll
        for (int I = J; I < NT; ++I) af[I] = LF[(tri(I) + J) * 256 + 64 * q + lane];
#pragma unroll
        for (int I = J; I < NT; ++I) {
          if (J == 0 && q == 0)
            acc[I] = mfma_f64(af[I], kv[0], d4_t{0.0, 0.0, 0.0, 0.0});
          else
            acc[I] = mfma_f64(af[I], kv[q], acc[I]);
        }
      }
      // row strip J is complete: its squares join the sum of its half (rows 16 J + g + 4 r, r ascending)
#pragma unroll
      for (int r = 0; r < 4; ++r) ss[J >> 2] = __builtin_fma(acc[J][r], acc[J][r], ss[J >> 2]);
    }
    // epilogue: the butterflies of ksx_kernel (mean) and post_kernel_w4 (variance)
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = mac[r];
      v[r] += __shfl_xor(v[r], 16);
      v[r] += __shfl_xor(v[r], 32);
    }
    const double msum = (v[0] + v[1]) + (v[2] + v[3]);
    double v0 = ss[0], v1 = ss[1];
    v0 += __shfl_xor(v0, 16);
    v0 += __shfl_xor(v0, 32);
    v1 += __shfl_xor(v1, 16);
    v1 += __shfl_xor(v1, 32);
    const int64_t g = st * 16 + lr;
    if (lane < 16 && g < a.M) {
      mu[g] = meanc + msum;
      double sq = v0;
      if (NT > 4) sq += v1;
      double o = (amp + -1.0 * sq) + var_add;
      if (a.clamp) o = (o < a.var_min) ? a.var_min : o;  // TH clamp: NaN passes through
      var[g] = o;
    }
  }
}

int ensure_kp_table(b7_ctx *c) {
  static bool done[64] = {false};
  if (c->device < 64 && done[c->device]) return B7_OK;
  B7_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(exp2_tab_kp), b7_exp2_tab, sizeof(b7_exp2_tab)));
  if (c->device < 64) done[c->device] = true;
  return B7_OK;
}

template <int DPAD, int NT>
int kp_launch(b7_ctx *c, const KpArgs &a) {
  B7_TRY(ensure_kp_table(c));
  const size_t lds = sizeof(double) * (size_t)kp_lds_doubles<DPAD>(NT);
  // the opt-in to > 64 KiB of dynamic LDS is per device: once per instantiation AND device (a process may hold contexts on several)
  static bool attr_done[64] = {false};
  if (c->device >= 64 || !attr_done[c->device]) {
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kpost_small_kernel<DPAD, NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
    if (c->device < 64) attr_done[c->device] = true;
  }
  // workgroups per CU: two where the instance's registers (<= 128: four waves per SIMD) and LDS allow it, else one; the CUs'
  // slots are split between the S fits; never more workgroups of a fit than it has strips for their waves
  static int per_cu_cache = 0;  // per instantiation
  if (per_cu_cache == 0) {
    hipFuncAttributes fa;
    B7_HIP(c, hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kpost_small_kernel<DPAD, NT>)));
    per_cu_cache = (fa.numRegs <= 128 && 2 * (lds + 512) <= (size_t)160 * 1024) ? 2 : 1;
  }
  const int64_t nstrips = (a.M + 15) / 16;
  int64_t gx = (int64_t)per_cu_cache * c->cus / a.S;
  const int64_t need = (nstrips + (KP_THREADS / 64) - 1) / (KP_THREADS / 64);
  if (gx > need) gx = need;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((kpost_small_kernel<DPAD, NT>), dim3((unsigned)gx, (unsigned)a.S), dim3(KP_THREADS), lds, c->stream, a);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

// the instance for ceil(N / 16) row strips: all eight for the narrow classes (d <= 8: the regime the kernel is for), rounded up
// to 4 / 8 for the wide ones (their padding strips multiply zeros: same bits, a bounded waste, a quarter of the binary)
template <int DPAD>
int kp_dispatch(b7_ctx *c, const KpArgs &a) {
  const int nt = (a.N + 15) / 16;
  if constexpr (DPAD <= 8) {
    switch (nt) {
      case 1: return kp_launch<DPAD, 1>(c, a);
      case 2: return kp_launch<DPAD, 2>(c, a);
      case 3: return kp_launch<DPAD, 3>(c, a);
      case 4: return kp_launch<DPAD, 4>(c, a);
      case 5: return kp_launch<DPAD, 5>(c, a);
      case 6: return kp_launch<DPAD, 6>(c, a);
      case 7: return kp_launch<DPAD, 7>(c, a);
      default: return kp_launch<DPAD, 8>(c, a);
    }
  } else {
    return nt <= 4 ? kp_launch<DPAD, 4>(c, a) : kp_launch<DPAD, 8>(c, a);
  }
}

}  // namespace

bool kpost_small_applies(const b7_ctx *c) { return c->Npad <= 128 && c->dfit <= 32 && c->ycols == 1; }

// S fits (S = 1 with hyp_dev == nullptr: the scalars amp / noise / mean) over the M candidates at xq: mu[s][M], var[s][M]
// (stride sout).  Strides of the per-fit arrays: w dpad, zsc Npad dpad, zss Npad, Linv Npad^2, alpha Npad.
int launch_kpost_small(b7_ctx *c, int S, const double *xq, int64_t M, const double *w, const double *zsc, const double *zss,
                       const double *Linv, const double *alpha, const double *hyp_dev, double amp, double noise, double mean,
                       double *mu, double *var, int64_t sout) {
  PhaseScope ps(c, "kpost");
  if (M <= 0) return B7_OK;
  KpArgs a = {};
  a.xq = xq, a.M = M, a.d = c->dfit, a.N = c->N, a.npad = c->Npad, a.S = S;
  a.w = w, a.zsc = zsc, a.zss = zss, a.Linv = Linv, a.alpha = alpha, a.hyp = hyp_dev;
  a.amp = amp, a.noise = noise, a.mean = mean;
  a.mu = mu, a.var = var, a.sout = sout;
  a.var_with_noise = c->opts.var_with_noise, a.clamp = c->opts.var_clamp, a.var_min = c->opts.var_min;
  switch (c->dpad) {
    case 4: return kp_dispatch<4>(c, a);
    case 8: return kp_dispatch<8>(c, a);
    case 16: return kp_dispatch<16>(c, a);
    case 32: return kp_dispatch<32>(c, a);
    default: return b7_fail(c, B7_ERR_UNSUPPORTED, "kpost_small: dpad %d", c->dpad);
  }
}
