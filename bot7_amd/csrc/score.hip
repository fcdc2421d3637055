// Acquisition scoring and the arg-max, with the reference's operation order.
//
// Reference arithmetic replaced (one rounded op here per Torch tensor op there; contraction is off):
//   utils/math.lua:261-288  erf   A&S 7.1.26: t = 1/(1+p|x|); Horner in t; 1 - poly*exp(-x^2); sign 2*(x>=0)-1
//   utils/math.lua:305-312  norm_cdf = (erf(x/sqrt2) + 1) * 0.5       :293-300 norm_pdf = exp(-x^2/2)/sqrt(2pi)
//   scores/expected_improvement.lua:69-88  sigma = sqrt(var); imprv = (fmin - mu) - xi; z = imprv/sigma;
//                                          ei = clamp(imprv*Phi(z) + sigma*phi(z), 0, inf); row mean if c > 1
//   scores/confidence_bound.lua:70-106     LCB = mu - sqrt(var)*k, UCB = mu + sqrt(var)*k; sign flip (:89-93)
//   bots/bayesopt.lua:69-79                score:add(...) per hyper sample, score:div(nSamples)
//   bots/bayesopt.lua:96                   score:max(1): first maximum, first NaN wins (TH max)
// exp/sqrt/division are the correctly-rounded-or-1-ulp ocml double routines; everything else is exact IEEE.
#pragma clang fp contract(off)
#include "b7_internal.h"

namespace {

__device__ __forceinline__ double b7_erf(double x) {
  const double c1 = 0.254829592, c2 = -0.284496736, c3 = 1.421413741, c4 = -1.453152027, c5 = 1.061405429,
               p = 0.3275911;
  double t = 1.0 / ((fabs(x) * p) + 1.0);
  double r = t * c5;
  r = r + c4;
  r = r * t;
  r = r + c3;
  r = r * t;
  r = r + c2;
  r = r * t;
  r = r + c1;
  r = r * t;
  double e = exp((x * x) * -1.0);
  r = ((r * e) * -1.0) + 1.0;
  double s = ((x >= 0.0) ? 1.0 : 0.0) * 2.0 + -1.0;
  return r * s;
}
__device__ __forceinline__ double b7_norm_cdf(double z) {
  double u = z * 0.70710678118654746;  // 1/math.sqrt(2), utils/math.lua:13
  return (b7_erf(u) + 1.0) * 0.5;
}
__device__ __forceinline__ double b7_norm_pdf(double z) {
  return exp((z * z) * -0.5) * 0.3989422804014327;  // 1/math.sqrt(2*math.pi), utils/math.lua:15
}

__global__ void __launch_bounds__(256)
    ei_kernel(const double *__restrict__ mu, const double *__restrict__ var, const double *__restrict__ fmin,
              double xi, int64_t M, int c, double *__restrict__ out, int accumulate, double fmin0) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += stride) {
    double sigma = sqrt(var[j]);
    double acc = 0.0;
    for (int k = 0; k < c; ++k) {
      double imprv = ((fmin ? fmin[k] : fmin0) + (-mu[j * c + k])) + (-xi);  // fmin == nullptr: one column, its f_min a kernel argument
      double z = imprv / sigma;
      double ei = (imprv * b7_norm_cdf(z)) + (sigma * b7_norm_pdf(z));
      ei = (ei < 0.0) ? 0.0 : ei;
      acc = (c == 1) ? ei : acc + ei;
    }
    double v = (c == 1) ? acc : acc / (double)c;
    out[j] = accumulate == 1 ? out[j] + v : (accumulate == 2 ? 0.0 + v : v);  // 2: the first score:add onto torch.zeros
  }
}

__global__ void __launch_bounds__(256)
    cb_kernel(const double *__restrict__ mu, const double *__restrict__ var, double kappa, int upper, double sign,
              int64_t M, int c, double *__restrict__ out, int accumulate) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += stride) {
    double s = sqrt(var[j]) * kappa;
    double acc = 0.0;
    for (int k = 0; k < c; ++k) {
      double v = upper ? (mu[j * c + k] + s) : (mu[j * c + k] + (-s));
      acc = (c == 1) ? v : acc + v;
    }
    double val = (c == 1) ? acc : acc / (double)c;
    val = (sign > 0.0) ? val : -val;
    out[j] = accumulate == 1 ? out[j] + val : (accumulate == 2 ? 0.0 + val : val);
  }
}

// S hyper samples at once (one response column): acc[j] = ((acc[j] + score_0) + score_1) + ... in sample order, exactly
// what S score:add calls leave (bots/bayesopt.lua:76)
__global__ void __launch_bounds__(256)
    ei_batch_kernel(const double *__restrict__ mu, const double *__restrict__ var, int S, int64_t sstride,
                    const double *__restrict__ fmin, double xi, int64_t M, double *__restrict__ out, int fresh, double fmin0) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += stride) {
    double a = fresh ? 0.0 : out[j];  // fresh: the accumulator is torch.zeros (bots/bayesopt.lua:69), not read
    for (int s = 0; s < S; ++s) {
      double sigma = sqrt(var[s * sstride + j]);
      double imprv = ((fmin ? fmin[0] : fmin0) + (-mu[s * sstride + j])) + (-xi);
      double z = imprv / sigma;
      double ei = (imprv * b7_norm_cdf(z)) + (sigma * b7_norm_pdf(z));
      ei = (ei < 0.0) ? 0.0 : ei;
      a = a + ei;
    }
    out[j] = a;
  }
}
__global__ void __launch_bounds__(256)
    cb_batch_kernel(const double *__restrict__ mu, const double *__restrict__ var, int S, int64_t sstride, double kappa,
                    int upper, double sign, int64_t M, double *__restrict__ out, int fresh) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += stride) {
    double a = fresh ? 0.0 : out[j];
    for (int s = 0; s < S; ++s) {
      double sd = sqrt(var[s * sstride + j]) * kappa;
      double v = upper ? (mu[s * sstride + j] + sd) : (mu[s * sstride + j] + (-sd));
      v = (sign > 0.0) ? v : -v;
      a = a + v;
    }
    out[j] = a;
  }
}

__global__ void __launch_bounds__(256) fill_kernel(double *__restrict__ p, int64_t n, double v) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) p[j] = v;
}

// (value, index) ordering of TH's max: the first NaN wins; otherwise the larger value; ties -> lower index.
struct Best {
  double v;
  int64_t i;
};
__device__ __forceinline__ bool better(const Best &a, const Best &b) {  // is a strictly preferred to b
  if (b.i < 0) return a.i >= 0;
  if (a.i < 0) return false;
  const bool an = a.v != a.v, bn = b.v != b.v;
  if (an || bn) return an && (!bn || a.i < b.i);
  return (a.v > b.v) || (a.v == b.v && a.i < b.i);
}
__device__ __forceinline__ Best wave_best(Best x) {
  for (int o = 32; o > 0; o >>= 1) {
    Best y;
    y.v = __shfl_xor(x.v, o);
    y.i = __shfl_xor(x.i, o);
    if (better(y, x)) x = y;
  }
  return x;
}
__device__ __forceinline__ Best block_best(Best x, Best *sh) {
  x = wave_best(x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = x;
  __syncthreads();
  if (wave == 0) {
    Best y = (lane < (int)(blockDim.x >> 6)) ? sh[lane] : Best{0.0, -1};
    y = wave_best(y);
    if (lane == 0) sh[0] = y;
  }
  __syncthreads();
  return sh[0];
}

// acc[j] /= divisor (score:div), then per-block best.
__global__ void __launch_bounds__(256)
    finish_kernel(double *__restrict__ acc, int64_t M, double divisor, Best *__restrict__ part) {
  __shared__ Best sh[4];
  Best b{0.0, -1};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += stride) {
    double v = acc[j] / divisor;
    acc[j] = v;
    Best cnd{v, j};
    if (better(cnd, b)) b = cnd;
  }
  b = block_best(b, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = b;
}

__global__ void __launch_bounds__(256) argmax_final_kernel(const Best *__restrict__ part, int n, Best *__restrict__ out) {
  __shared__ Best sh[4];
  Best b{0.0, -1};
  for (int j = threadIdx.x; j < n; j += blockDim.x)
    if (better(part[j], b)) b = part[j];
  b = block_best(b, sh);
  if (threadIdx.x == 0) out[0] = b;
}

// The global-exchange form of argmax_final_kernel: this rank's record of the exchange table (b7_internal.h: value bits,
// 1-based GLOBAL index, status 0, shard rows, the winner's grid row).  With all_slots the records of every other rank are
// zeroed as well (comm.hip sums the tables of all ranks); a single-process group merges records on the host and needs
// only this one.
// this rank's record (and, with all_slots, zeros for every other rank's) from the local best b; every thread of the block calls
__device__ __forceinline__ void write_record(Best b, unsigned long long *__restrict__ tab, int rank, int world, long long offset,
                                             long long rows, const double *__restrict__ grid, int d, int all_slots,
                                             unsigned long long *__restrict__ host_rec, unsigned *__restrict__ host_done) {
  if (all_slots)
    for (int e = threadIdx.x; e < world * B7_TAB_W; e += blockDim.x)
      if (e / B7_TAB_W != rank) tab[e] = 0ull;
  unsigned long long *rec = tab + (size_t)rank * B7_TAB_W;
  const bool have = b.i >= 0;
  for (int e = threadIdx.x; e < B7_TAB_W; e += blockDim.x) {
    unsigned long long w = 0ull;
    if (e == B7_TAB_VAL) w = have ? (unsigned long long)__double_as_longlong(b.v) : 0ull;
    else if (e == B7_TAB_IDX) w = have ? (unsigned long long)(offset + b.i + 1) : 0ull;
    else if (e == B7_TAB_ROWS) w = (unsigned long long)rows;
    else if (e >= B7_TAB_ROW0 && e - B7_TAB_ROW0 < d && have && grid)
      w = (unsigned long long)__double_as_longlong(grid[b.i * d + (e - B7_TAB_ROW0)]);
    rec[e] = w;
    if (host_rec) host_rec[e] = w;  // the same record straight into mapped host memory: no copy launch behind this kernel
  }
  if (host_done) {
    // the host spins on this word instead of waiting for the stream: every thread's part of the record is pushed out to
    // system scope before the barrier, thread 0's release store comes after it
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void __launch_bounds__(256) argmax_slot_kernel(const Best *__restrict__ part, int n, unsigned long long *__restrict__ tab,
                                                          int rank, int world, long long offset, long long rows,
                                                          const double *__restrict__ grid, int d, int all_slots,
                                                          long long forced_local, unsigned long long *__restrict__ host_rec,
                                                          unsigned *__restrict__ host_done) {
  __shared__ Best sh[4];
  Best b{0.0, -1};
  if (n > 0) {
    for (int j = threadIdx.x; j < n; j += blockDim.x)
      if (better(part[j], b)) b = part[j];
    b = block_best(b, sh);
  } else if (forced_local >= 0) {  // no scores: the record names a given row (b7_nominate_commit's broadcast)
    b = Best{0.0, forced_local};
  }
  write_record(b, tab, rank, world, offset, rows, grid, d, all_slots, host_rec, host_done);
}

// score:add over the S hyper samples of a nomination, score:div, score:max(1) and this rank's exchange record in ONE launch
// (bots/bayesopt.lua:76-79, :96): ei_batch_kernel / cb_batch_kernel + finish_kernel + argmax_slot_kernel, the same operations in
// the same order per candidate -- acc = ((0 + s_0) + s_1) + ..., acc / divisor -- then the per-block best, and the LAST block to
// arrive (a ticket from one atomic counter; every block's partial is fenced before its ticket) reduces the partials and writes
// the record.  Three dependent launches at the ~4.5 us dispatch floor each become one.
struct ScoreArgs {
  const double *mu, *var;
  int S, kind;  // kind: B7_SCORE_EI / B7_SCORE_CB
  long long sstride;
  const double *fmin;
  double fmin0, tradeoff, sign;
  int upper, fresh;
};
__global__ void __launch_bounds__(256)
    score_finish_slot_kernel(ScoreArgs sa, double *__restrict__ acc, long long M, double divisor, Best *__restrict__ part,
                             unsigned *__restrict__ ticket, unsigned long long *__restrict__ tab, int rank, int world, long long offset,
                             const double *__restrict__ grid, int d, int all_slots, unsigned long long *__restrict__ host_rec,
                             unsigned *__restrict__ host_done) {
  __shared__ Best sh[4];
  __shared__ unsigned last;
  Best b{0.0, -1};
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += stride) {
    double a = sa.fresh ? 0.0 : acc[j];  // fresh: the accumulator is torch.zeros (bots/bayesopt.lua:69), not read
    for (int s = 0; s < sa.S; ++s) {
      const double m = sa.mu[s * sa.sstride + j], vr = sa.var[s * sa.sstride + j];
      double sc;
      if (sa.kind == B7_SCORE_EI) {
        double sigma = sqrt(vr);
        double imprv = ((sa.fmin ? sa.fmin[0] : sa.fmin0) + (-m)) + (-sa.tradeoff);
        double z = imprv / sigma;
        sc = (imprv * b7_norm_cdf(z)) + (sigma * b7_norm_pdf(z));
        sc = (sc < 0.0) ? 0.0 : sc;
      } else {
        double sd = sqrt(vr) * sa.tradeoff;
        sc = sa.upper ? (m + sd) : (m + (-sd));
        sc = (sa.sign > 0.0) ? sc : -sc;
      }
      a = a + sc;
    }
    const double v = a / divisor;
    acc[j] = v;
    Best cnd{v, j};
    if (better(cnd, b)) b = cnd;
  }
  b = block_best(b, sh);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = b;
    __threadfence();  // the partial is visible device-wide before the ticket is taken
    const unsigned t = atomicAdd(ticket, 1u);
    last = (t == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();  // the other blocks' partials, published before their tickets
  Best f{0.0, -1};
  for (int j = threadIdx.x; j < (int)gridDim.x; j += blockDim.x) {
    Best pj;  // agent-scope loads: the partials of other CUs, not a stale line of this CU's cache
    pj.v = __hip_atomic_load(&part[j].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pj.i = __hip_atomic_load(&part[j].i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (better(pj, f)) f = pj;
  }
  __syncthreads();
  f = block_best(f, sh);
  if (threadIdx.x == 0) *ticket = 0u;  // ready for the next launch (stream order: nobody else touches it meanwhile)
  write_record(f, tab, rank, world, offset, M, grid, d, all_slots, host_rec, host_done);
}

__global__ void __launch_bounds__(256) keep_record_kernel(unsigned long long *__restrict__ tab, int rank, int world) {
  for (int e = threadIdx.x; e < world * B7_TAB_W; e += blockDim.x)
    if (e / B7_TAB_W != rank) tab[e] = 0ull;
}

int nblocks(b7_ctx *c, int64_t n) {
  int64_t b = (n + 255) / 256, cap = (int64_t)c->cus * 8;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

// 0: out = score; 1: out += score; 2: out = 0.0 + score -- the context's accumulator was declared torch.zeros without being
// filled (b7_eval_nominate: bots/bayesopt.lua:69 costs no launch of its own) and this is the first score:add onto it
static int acc_mode(b7_ctx *c, const double *out, bool accumulate) {
  if (out != (const double *)c->acc.p) return accumulate ? 1 : 0;
  const bool fresh = c->acc_fresh;
  c->acc_fresh = false;
  return accumulate ? (fresh ? 2 : 1) : 0;
}

int launch_ei(b7_ctx *c, const double *mu, const double *var, const double *fmin_dev, double tradeoff, int64_t M,
              int ycols, double *out, bool accumulate) {
  PhaseScope ps(c, "score");
  if (M <= 0) return B7_OK;
  if (!fmin_dev && ycols != 1) return b7_fail(c, B7_ERR_INVALID, "ei: f_min of %d columns must be staged on the device", ycols);
  hipLaunchKernelGGL(ei_kernel, dim3(nblocks(c, M)), dim3(256), 0, c->stream, mu, var, fmin_dev, tradeoff, M, ycols,
                     out, acc_mode(c, out, accumulate), c->fmin_scalar);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_cb(b7_ctx *c, const double *mu, const double *var, double tradeoff, int upper, double sign, int64_t M,
              int ycols, double *out, bool accumulate) {
  PhaseScope ps(c, "score");
  if (M <= 0) return B7_OK;
  hipLaunchKernelGGL(cb_kernel, dim3(nblocks(c, M)), dim3(256), 0, c->stream, mu, var, tradeoff, upper, sign, M,
                     ycols, out, acc_mode(c, out, accumulate));
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_ei_batch(b7_ctx *c, int S, const double *mu, const double *var, int64_t stride, const double *fmin_dev,
                    double tradeoff, int64_t M, double *acc) {
  PhaseScope ps(c, "score");
  if (M <= 0) return B7_OK;
  hipLaunchKernelGGL(ei_batch_kernel, dim3(nblocks(c, M)), dim3(256), 0, c->stream, mu, var, S, stride, fmin_dev, tradeoff, M,
                     acc, acc_mode(c, acc, true) == 2 ? 1 : 0, c->fmin_scalar);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_cb_batch(b7_ctx *c, int S, const double *mu, const double *var, int64_t stride, double tradeoff, int upper,
                    double sign, int64_t M, double *acc) {
  PhaseScope ps(c, "score");
  if (M <= 0) return B7_OK;
  hipLaunchKernelGGL(cb_batch_kernel, dim3(nblocks(c, M)), dim3(256), 0, c->stream, mu, var, S, stride, tradeoff, upper, sign,
                     M, acc, acc_mode(c, acc, true) == 2 ? 1 : 0);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_fill(b7_ctx *c, double *p, int64_t n, double v) {
  if (n <= 0) return B7_OK;
  hipLaunchKernelGGL(fill_kernel, dim3(nblocks(c, n)), dim3(256), 0, c->stream, p, n, v);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

// zeros that were only declared (acc_fresh) and never met a score launch: write them before anybody reads the accumulator
// a batched score that b7_eval_nominate left for the fused finish (launch_score_finish_slot) and somebody else wants first
int score_flush_pending(b7_ctx *c) {
  if (!c->pend.on) return B7_OK;
  const b7_ctx::PendingScore ps = c->pend;
  c->pend.on = false;
  if (ps.kind == B7_SCORE_EI) return launch_ei_batch(c, ps.S, ps.mu, ps.var, ps.stride, ps.fd, ps.tradeoff, c->M, (double *)c->acc.p);
  return launch_cb_batch(c, ps.S, ps.mu, ps.var, ps.stride, ps.tradeoff, ps.upper, ps.sign, c->M, (double *)c->acc.p);
}

int acc_materialize(b7_ctx *c) {
  B7_TRY(score_flush_pending(c));
  if (!c->acc_fresh) return B7_OK;
  c->acc_fresh = false;
  return launch_fill(c, (double *)c->acc.p, c->M, 0.0);
}

int launch_finish(b7_ctx *c, double *acc, int64_t M, double divisor, double *best_val, int64_t *best_idx1) {
  PhaseScope ps(c, "argmax");
  if (M <= 0) return b7_fail(c, B7_ERR_INVALID, "finish: empty score vector");
  const int nb = nblocks(c, M);
  B7_TRY(b7_ensure(c, c->part, sizeof(Best) * (size_t)(nb + 1)));
  Best *part = (Best *)c->part.p;
  hipLaunchKernelGGL(finish_kernel, dim3(nb), dim3(256), 0, c->stream, acc, M, divisor, part);
  // the final (value, index) goes straight into pinned, device-mapped host memory: no copy, just the synchronisation
  Best *res_dev = reinterpret_cast<Best *>(static_cast<char *>(c->pinned_dev) + 2304);
  const Best &h = *reinterpret_cast<const Best *>(static_cast<const char *>(c->pinned) + 2304);
  hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(256), 0, c->stream, (const Best *)part, nb, res_dev);
  B7_HIP(c, hipGetLastError());
  B7_HIP(c, hipStreamSynchronize(c->stream));
  if (best_val) *best_val = h.v;
  if (best_idx1) *best_idx1 = h.i + 1;
  return B7_OK;
}

// launch_finish without the host round trip: the local result stays on the device, in this rank's record of the
// exchange table, together with the grid row it names.  M == 0 (an empty shard) writes an all-zero record.
// host_rec / host_done (nullable): device addresses of a mapped host copy of this rank's record and of the word the kernel
// sets once that copy is complete (comm.hip: exch_local with a mirror).
int launch_finish_slot(b7_ctx *c, double *acc, int64_t M, double divisor, uint64_t *tab_dev, int rank, int world,
                       int64_t offset, const double *grid, int d, bool all_slots, uint64_t *host_rec, unsigned *host_done) {
  PhaseScope ps(c, "argmax");
  const int nb = M > 0 ? nblocks(c, M) : 0;
  B7_TRY(b7_ensure(c, c->part, sizeof(Best) * (size_t)(nb + 1)));
  Best *part = (Best *)c->part.p;
  if (nb > 0) hipLaunchKernelGGL(finish_kernel, dim3(nb), dim3(256), 0, c->stream, acc, M, divisor, part);
  hipLaunchKernelGGL(argmax_slot_kernel, dim3(1), dim3(256), 0, c->stream, (const Best *)part, nb,
                     (unsigned long long *)tab_dev, rank, world, (long long)offset, (long long)M, grid, d, all_slots ? 1 : 0,
                     -1ll, (unsigned long long *)host_rec, host_done);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

// The fused form: the S-sample score, score:div, the local arg-max and the record in one launch.
int launch_score_finish_slot(b7_ctx *c, const b7_ctx::PendingScore &ps, double *acc, int64_t M, double divisor, uint64_t *tab_dev,
                             int rank, int world, int64_t offset, const double *grid, int d, bool all_slots, uint64_t *host_rec,
                             unsigned *host_done) {
  PhaseScope scope(c, "score");
  // (one-wave blocks for small grids -- 313 instead of 79 workgroups for 2e4 candidates -- measured SLOWER: 145 vs 139 us per
  // nomination at N = 100, S = 10; the last block's pass over four times as many partials costs more than the spread saves.
  // Likewise the S scores of a candidate on S threads side by side, parked in LDS and added in order by one of them, over 512
  // blocks: 137 vs 132 us.  The kernel's 15 us are launch, ticket and the last block's pass, not the ten scores in a row)
  const int threads = 256;
  const int nb = nblocks(c, M);
  B7_TRY(b7_ensure(c, c->part, sizeof(Best) * (size_t)(nb + 1)));
  Best *part = (Best *)c->part.p;
  if (!c->ticket.p) {  // the ticket counter: a word of its own (c->part is shared scratch), zero between launches
    B7_TRY(b7_ensure(c, c->ticket, 64));
    B7_HIP(c, hipMemsetAsync(c->ticket.p, 0, 64, c->stream));
  }
  unsigned *ticket = (unsigned *)c->ticket.p;
  ScoreArgs sa{ps.mu, ps.var, ps.S, ps.kind, (long long)ps.stride, ps.fd, c->fmin_scalar, ps.tradeoff, ps.sign, ps.upper,
               acc_mode(c, acc, true) == 2 ? 1 : 0};
  hipLaunchKernelGGL(score_finish_slot_kernel, dim3(nb), dim3(threads), 0, c->stream, sa, acc, (long long)M, divisor, part, ticket,
                     (unsigned long long *)tab_dev, rank, world, (long long)offset, grid, d, all_slots ? 1 : 0,
                     (unsigned long long *)host_rec, host_done);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

// The record of a row broadcast: the rank that holds global row idx1_global (local0 >= 0, 0-based within its shard) writes
// (0.0, idx1_global, 0, rows, the row); every other rank writes zeros.
int launch_row_slot(b7_ctx *c, uint64_t *tab_dev, int rank, int world, int64_t idx1_global, int64_t local0, const double *grid,
                    int d) {
  hipLaunchKernelGGL(argmax_slot_kernel, dim3(1), dim3(256), 0, c->stream, (const Best *)nullptr, 0,
                     (unsigned long long *)tab_dev, rank, world, (long long)(idx1_global - 1 - (local0 >= 0 ? local0 : 0)),
                     (long long)c->M, grid, d, 1, (long long)local0, (unsigned long long *)nullptr, (unsigned *)nullptr);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

int launch_keep_record(b7_ctx *c, uint64_t *tab_dev, int rank, int world) {
  hipLaunchKernelGGL(keep_record_kernel, dim3(1), dim3(256), 0, c->stream, (unsigned long long *)tab_dev, rank, world);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}
