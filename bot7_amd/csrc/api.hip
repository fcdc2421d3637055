// C-ABI entry points of libbot7hip.so (declared in include/bot7hip.h) and the context behind them.
// Host-side orchestration only: every number is produced by the kernels in sobol/covar/potrf/posterior/
// score.hip.  There is no CPU fallback: without a working HIP device b7_create fails and nothing else runs.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "b7_internal.h"
#include <chrono>

static thread_local std::string g_create_err;

int b7_fail(b7_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c)
    c->err = buf;
  else
    g_create_err = buf;
  return code;
}

int b7_ensure(b7_ctx *c, DevBuf &b, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (b.cap >= bytes) return B7_OK;
  if (b.p) {
    // keep stream order: nothing in flight may still use the old block
    B7_HIP(c, hipStreamSynchronize(c->stream));
    B7_HIP(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  hipError_t e = hipMalloc(&b.p, bytes);
  if (e != hipSuccess) {
    b.p = nullptr;
    return b7_fail(c, B7_ERR_NOMEM, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
  }
  b.cap = bytes;
  return B7_OK;
}

void b7_release(DevBuf &b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

static hipEvent_t phase_event(b7_ctx *c) {
  if (!c->free_events.empty()) {
    hipEvent_t e = c->free_events.back();
    c->free_events.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

// Turn the recorded event pairs into phase times (one stream synchronisation) and recycle the events.
static void resolve_phases(b7_ctx *c) {
  if (c->pending.empty()) return;
  (void)hipStreamSynchronize(c->stream);
  for (const b7_ctx::PendingPhase &p : c->pending) {
    float ms = 0.f;
    if (p.e0 && p.e1 && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      PhaseStat &s = c->phases[p.name];
      s.ms += ms;
      s.launches += 1;
    }
    if (p.e0) c->free_events.push_back(p.e0);
    if (p.e1) c->free_events.push_back(p.e1);
  }
  c->pending.clear();
}

// Phases do not nest.  Recording costs two hipEventRecord calls and no synchronisation: a profiled step keeps the
// host running ahead of the GPU exactly like an unprofiled one.
PhaseScope::PhaseScope(b7_ctx *c_, const char *name_) : c(c_), name(name_) {
  if (!c->profile) return;
  c->phase_e0 = phase_event(c);
  if (c->phase_e0) (void)hipEventRecord(c->phase_e0, c->stream);
}
PhaseScope::~PhaseScope() {
  if (!c->profile || !c->phase_e0) return;
  hipEvent_t e1 = phase_event(c);
  if (e1) (void)hipEventRecord(e1, c->stream);
  c->pending.push_back({name, c->phase_e0, e1});
  c->phase_e0 = nullptr;
  if (c->pending.size() >= 4096) resolve_phases(c);  // bound the pool in long profiled loops
}

static double *cur_grid(b7_ctx *c) { return (double *)c->grid[c->grid_cur].p; }

static void invalidate_predictions(b7_ctx *c) {
  c->pend.on = false;  // a batched score nobody collected belongs to the grid that just changed
  c->predicted = false;
  c->acc_valid = false;
  c->Mfeat = 0;  // DNGO features belong to the grid they were computed from
  c->win_valid = false;  // so does the winner's row of the last exchange
}

static int group_guard(b7_ctx *c, const char *who) {
  if (c->group && !c->group_busy)
    return b7_fail(c, B7_ERR_STATE, "%s: this context's grid is a shard of a group; use the b7_group_grid_* calls", who);
  return B7_OK;
}

// Stable deletion of candidate row local_idx1 (utils/tensor.lua:158-170), enqueued on the context's stream.  With row_out
// the removed row is copied out first and the stream is synchronised.
int grid_drop_row(b7_ctx *c, int64_t idx1, double *row_out) {
  B7_TRY(group_guard(c, "grid_remove"));
  if (idx1 < 1 || idx1 > c->M)
    return b7_fail(c, B7_ERR_INVALID, "grid_remove: index %lld outside [1, %lld]", (long long)idx1, (long long)c->M);
  B7_HIP(c, hipSetDevice(c->device));
  if (row_out)
    B7_HIP(c, hipMemcpyAsync(row_out, cur_grid(c) + (idx1 - 1) * c->d, sizeof(double) * c->d, hipMemcpyDeviceToHost,
                             c->stream));
  const int other = c->grid_cur ^ 1;
  B7_TRY(b7_ensure(c, c->grid[other], sizeof(double) * (size_t)c->M * c->d));
  B7_TRY(launch_remove_row(c, cur_grid(c), (double *)c->grid[other].p, c->M, c->d, idx1 - 1));
  c->grid_cur = other;
  c->M -= 1;
  invalidate_predictions(c);
  if (row_out) B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

extern "C" {

int b7_abi_version(void) { return B7_ABI_VERSION; }

int b7_create(b7_ctx **out, int device_id) {
  if (!out) return b7_fail(nullptr, B7_ERR_INVALID, "b7_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return b7_fail(nullptr, B7_ERR_HIP, "no HIP device available (%s); libbot7hip has no CPU path",
                   e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= ndev)
    return b7_fail(nullptr, B7_ERR_INVALID, "device_id %d out of range [0,%d)", device_id, ndev);
  e = hipSetDevice(device_id);
  if (e != hipSuccess) return b7_fail(nullptr, B7_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device_id);
  if (e != hipSuccess) return b7_fail(nullptr, B7_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return b7_fail(nullptr, B7_ERR_UNSUPPORTED, "device %d is %s; libbot7hip is built for gfx950 only", device_id,
                   prop.gcnArchName);
  b7_ctx *c = new b7_ctx();
  c->device = device_id;
  c->cus = prop.multiProcessorCount;
  b7_gp_default_opts(&c->opts);
  // the one tunable of the shipped library, read once here, never in the launch paths
  if (const char *pv = getenv("B7_SPIN_US")) c->spin_us = atoi(pv);  // 0: never spin on a completion word, always wait for the stream
#ifdef B7_DIAG
  // DIAGNOSTIC build only (tools/_build/libbot7hip_diag.so, python -m bot7_amd.build --diag; tests load it beside the shipped
  // library): the A/B arms that tests/test_gpu_parity.py holds against the default paths, stamp collection, the persistent
  // schedule's fault injector.  None of these names exists in libbot7hip.so.
  if (const char *pv = getenv("B7_DIAG_VARIANT")) c->diag_variant = atoi(pv);  // 0 rsqrt chain, 1 DPP-fused (default), 2 its mov+fma reference
  if (const char *pv = getenv("B7_NPAD_SMALL")) c->npad_small = atoi(pv) != 0;  // 0: pad N <= 64 (and <= 64 basis features) to 128 as N > 64
  if (const char *pv = getenv("B7_POTRF_SMALL")) c->potrf_small = atoi(pv) != 0;
  if (getenv("B7_POTRF_SCHED") || getenv("B7_DIAG_VARIANT") || getenv("B7_INVERSE_INLINE")) c->potrf_small = c->fit_small = false;  // an explicit schedule is an A/B arm of the general path
  if (const char *pv = getenv("B7_BLR_SMALL")) c->blr_small = atoi(pv) != 0;  // 0: the head of b7_blr_eval_nominate through the general launches
  if (const char *pv = getenv("B7_INVERSE_INLINE")) c->inverse_inline = atoi(pv);  // 0 never, 1 up to N = 8192, 2 always
  if (const char *pv = getenv("B7_POTRF_SCHED")) c->potrf_sched = atoi(pv);  // 0 pairs, 1 one panel at a time up to N = 4096, 2 always
  c->persist_stamps = getenv("B7_PERSIST_STAMPS") != nullptr;
  if (const char *pv = getenv("B7_PERSIST_HELPERS")) c->persist_helpers = atoi(pv);
  if (const char *pv = getenv("B7_PERSIST_FAULT")) c->persist_fault = atoi(pv);
  if (const char *pv = getenv("B7_NLL_SMALL")) c->nll_small = atoi(pv);  // 0: likelihoods of small sets through the general path too; 2: round 3's kernel
  if (const char *pv = getenv("B7_FIT_SMALL")) c->fit_small = atoi(pv) != 0;  // 0: small fits through the general schedule too
  if (const char *pv = getenv("B7_KPOST_SMALL")) c->kpost_small = atoi(pv) != 0;  // 0: small posteriors through ksx_kernel + post_kernel too
  if (const char *pv = getenv("B7_SYRK_SMALL")) c->syrk_small = atoi(pv) ? 1 : 0;
  if (const char *pv = getenv("B7_POTRF_DEFER")) c->potrf_defer = atoi(pv) ? 1 : 0;
  if (const char *pv = getenv("B7_POTRF_GROUP")) {
    const int g = atoi(pv);
    if (g >= 1 && g <= 8) c->potrf_group = g;
  }
#endif
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fit, hipEventDisableTiming);
  if (e == hipSuccess) e = hipHostMalloc(&c->pinned, 16384, hipHostMallocMapped);
  if (e == hipSuccess) e = hipHostGetDevicePointer(&c->pinned_dev, c->pinned, 0);
  if (e != hipSuccess) {
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return b7_fail(nullptr, B7_ERR_HIP, "stream / pinned result block creation: %s", hipGetErrorString(e));
  }
  // fixed scratch: [0,1K) lengthscales, [2K,4K) fmin, [4K,...) Sobol table + mins/maxes; never regrown
  if (b7_ensure(c, c->scratch, 64 * 1024) != B7_OK) {
    g_create_err = c->err;
    b7_destroy(c);
    return B7_ERR_NOMEM;
  }
  *out = c;
  return B7_OK;
}

void b7_destroy(b7_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)b7_comm_destroy(c);
  DevBuf *all[] = {&c->grid[0], &c->grid[1], &c->xobs, &c->w,     &c->zsc,  &c->zss,     &c->K,      &c->L,
                   &c->Linv,    &c->W,       &c->dinv, &c->alpha, &c->resid, &c->info,   &c->ybuf,   &c->mu,
                   &c->var,     &c->acc,     &c->ks,   &c->part,  &c->scratch, &c->tmpgrid, &c->tmpmu, &c->tmpvar, &c->fant, &c->feat, &c->netbuf, &c->atmp, &c->slots, &c->pstamps,
                   &c->bhyp, &c->bw, &c->bzsc, &c->bzss, &c->bK, &c->bL, &c->bdinv, &c->bflags, &c->binfo, &c->bresid, &c->bterms,
                   &c->bLinv, &c->balpha, &c->bmu, &c->bvar, &c->ticket};
  for (auto &kv : c->pjobs_cache) b7_release(kv.second.buf);
  for (DevBuf *b : all) b7_release(*b);
  if (c->tev_init)
    for (int i = 0; i < B7_MAX_TIMERS; ++i) {
      (void)hipEventDestroy(c->tev[i][0]);
      (void)hipEventDestroy(c->tev[i][1]);
    }
  resolve_phases(c);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->pin_eval) (void)hipHostFree(c->pin_eval);
  if (c->pin_blr) (void)hipHostFree(c->pin_blr);
  if (c->pin_nll) (void)hipHostFree(c->pin_nll);
  if (c->tab_host) (void)hipHostFree(c->tab_host);
  for (hipEvent_t e : c->free_events) (void)hipEventDestroy(e);
  if (c->phase_e0) (void)hipEventDestroy(c->phase_e0);
  if (c->ev_fit) (void)hipEventDestroy(c->ev_fit);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *b7_last_error(const b7_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int b7_device_info(b7_ctx *c, char *name_out, int *compute_units, int64_t *hbm_bytes) {
  if (!c) return B7_ERR_INVALID;
  hipDeviceProp_t prop;
  B7_HIP(c, hipGetDeviceProperties(&prop, c->device));
  if (name_out) snprintf(name_out, 64, "%s (%s)", prop.name, prop.gcnArchName);
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return B7_OK;
}

int b7_sync(b7_ctx *c) {
  if (!c) return B7_ERR_INVALID;
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

int b7_set_workspace(b7_ctx *c, int64_t bytes) {
  if (!c || bytes < (int64_t)(8 * 128 * 128)) return c ? b7_fail(c, B7_ERR_INVALID, "workspace too small") : B7_ERR_INVALID;
  c->ks_bytes = (size_t)bytes;
  return B7_OK;
}

// ---- grids ---------------------------------------------------------------------------------------------
static int grid_alloc(b7_ctx *c, int64_t M, int d) {
  B7_TRY(group_guard(c, "grid"));
  if (M < 0 || d < 1) return b7_fail(c, B7_ERR_INVALID, "grid: size %lld dims %d", (long long)M, d);
  if (d > B7_MAX_D) return b7_fail(c, B7_ERR_UNSUPPORTED, "grid: dims %d > %d", d, B7_MAX_D);
  B7_HIP(c, hipSetDevice(c->device));
  c->grid_cur = 0;
  B7_TRY(b7_ensure(c, c->grid[0], sizeof(double) * (size_t)M * d));
  c->M = M;
  c->d = d;
  invalidate_predictions(c);
  return B7_OK;
}

static int grid_copy_out(b7_ctx *c, double *out_host) {
  if (out_host && c->M > 0)
    B7_HIP(c, hipMemcpyAsync(out_host, cur_grid(c), sizeof(double) * (size_t)c->M * c->d, hipMemcpyDeviceToHost,
                             c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

// grids/sobol.lua:82-85, grids/random.lua:29-32: only one of mins / maxes given.  The shift / scale uses the column minima /
// maxima of the WHOLE grid: with a communicator the shards' extremes are combined by one all-reduce (min / max of d doubles).
// That makes the call COLLECTIVE, so a rank that failed before it (rc_before: a bad argument, an allocation) still enters it:
// the reduced vector carries one more element, a status that the failing rank sets to the value that wins the reduction
// (-inf under MIN, +inf under MAX), and every rank returns an error together instead of the others waiting for ever.
static int onesided(b7_ctx *c, const double *mins, const double *maxes, int rc_before) {
  const bool collective = c->comm && c->comm_world > 1;
  if (rc_before != B7_OK && !collective) return rc_before;
  const int d = (rc_before == B7_OK) ? c->d : 0;
  std::vector<double> ext(2 * (size_t)B7_MAX_D + 2);
  double *lo = ext.data(), *hi = ext.data() + B7_MAX_D + 1;
  for (int k = 0; k <= B7_MAX_D; ++k) lo[k] = INFINITY, hi[k] = -INFINITY;  // an empty shard constrains nothing
  int rc = rc_before;
  const std::string own = c->err;
  if (rc == B7_OK && c->M > 0) {
    std::vector<double> cmin(d), cmax(d);
    rc = b7_grid_colrange(c, cmin.data(), cmax.data());
    if (rc == B7_OK) {
      memcpy(lo, cmin.data(), sizeof(double) * d);
      memcpy(hi, cmax.data(), sizeof(double) * d);
    }
  }
  const std::string own2 = rc != B7_OK ? c->err : own;
  double *use = mins ? lo : hi;
  if (collective) {
    // every rank reduces the same B7_MAX_D + 1 values whatever its own d (a failed rank may not know it): the extremes, then
    // the status
    use[B7_MAX_D] = rc == B7_OK ? (mins ? INFINITY : -INFINITY) : (mins ? -INFINITY : INFINITY);
    const int rcc = b7_comm_allreduce_f64(c, use, B7_MAX_D + 1, mins ? B7_COMM_MIN : B7_COMM_MAX);
    if (rc != B7_OK) {
      c->err = own2;
      return rc;
    }
    if (rcc != B7_OK) return rcc;
    if (use[B7_MAX_D] == (mins ? -INFINITY : INFINITY))
      return b7_fail(c, B7_ERR_COMM, "one-sided grid map: another rank failed before the exchange of the column extremes; no rank maps its shard");
  }
  if (rc != B7_OK) return rc;
  if (c->M == 0) return B7_OK;
  return b7_grid_apply_onesided(c, mins, maxes, use);
}

static int grid_sobol_local(b7_ctx *c, int64_t size, int dims, int64_t skip, const double *mins, const double *maxes) {
  if (size < 0 || skip < 0) return b7_fail(c, B7_ERR_INVALID, "sobol: size %lld skip %lld", (long long)size, (long long)skip);
  if (dims < 1 || dims >= 40)  // assert(C.dims and C.dims < C.max_dims), grids/sobol.lua:36
    return b7_fail(c, B7_ERR_RANGE, "sobol: dims %d not in [1, 39] (grids/sobol.lua:36)", dims);
  // "Too many calls": lo0(seed) must stay <= 30 (grids/sobol.lua:317-324) -> seed <= 2^30 - 2
  if (size > 0 && size + skip - 1 > ((int64_t)1 << 30) - 2)
    return b7_fail(c, B7_ERR_RANGE, "sobol: point index %lld beyond 2^30-2 (grids/sobol.lua:317-324)",
                   (long long)(size + skip - 1));
  const bool both = mins && maxes;
  B7_TRY(grid_alloc(c, size, dims));
  return launch_sobol(c, cur_grid(c), size, dims, skip, both ? mins : nullptr, both ? maxes : nullptr);
}

int b7_grid_sobol(b7_ctx *c, int64_t size, int dims, int64_t skip, const double *mins, const double *maxes,
                  double *out_host) {
  if (!c) return B7_ERR_INVALID;
  int rc = grid_sobol_local(c, size, dims, skip, mins, maxes);
  if ((mins != nullptr) != (maxes != nullptr)) rc = onesided(c, mins, maxes, rc);  // collective with a communicator: entered on failure too
  B7_TRY(rc);
  return grid_copy_out(c, out_host);
}

static int grid_random_local(b7_ctx *c, int64_t size, int dims, uint64_t seed, int64_t row_offset, const double *mins,
                             const double *maxes) {
  if (size < 0 || row_offset < 0) return b7_fail(c, B7_ERR_INVALID, "random grid: size/offset negative");
  const bool both = mins && maxes;
  B7_TRY(grid_alloc(c, size, dims));
  return launch_random_grid(c, cur_grid(c), size, dims, seed, row_offset, both ? mins : nullptr, both ? maxes : nullptr);
}

int b7_grid_random(b7_ctx *c, int64_t size, int dims, uint64_t seed, int64_t row_offset, const double *mins,
                   const double *maxes, double *out_host) {
  if (!c) return B7_ERR_INVALID;
  int rc = grid_random_local(c, size, dims, seed, row_offset, mins, maxes);
  if ((mins != nullptr) != (maxes != nullptr)) rc = onesided(c, mins, maxes, rc);
  B7_TRY(rc);
  return grid_copy_out(c, out_host);
}

// ---- torch.rand's own stream (grids/random.lua:24) ---------------------------------------------------------------------
// Torch7's CPU generator is MT19937 (TH/THRandom.c) [public knowledge, not in the reference tree]: manualSeed(s) is
// init_genrand(s), THRandom_random the tempered 32-bit output, and torch.rand fills a tensor in row-major order.  Two
// generations of TH differ in how a double is made from it: `resolution` 32 is THRandom_uniform's
// random() * 2^-32 (Torch7 up to 2017, the era of bot7), 53 the later ((random64() & (2^53 - 1)) * 2^-53 with
// random64 = (random() << 32) | random().  Sequential by construction, so it runs on the host (33M draws take 0.1 s) and
// only the affine map of grids/random.lua:27-33 runs on the device.
namespace {
struct Mt19937 {
  uint32_t mt[624];
  int idx;
  explicit Mt19937(uint32_t seed) {
    mt[0] = seed;
    for (int j = 1; j < 624; ++j) mt[j] = 1812433253u * (mt[j - 1] ^ (mt[j - 1] >> 30)) + (uint32_t)j;
    idx = 624;
  }
  uint32_t next() {
    if (idx >= 624) {
      for (int k = 0; k < 624; ++k) {
        const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
};
}  // namespace

int b7_torch_rand(uint64_t seed, int64_t n, int resolution, double *out) {
  if (n < 0 || (n > 0 && !out) || (resolution != 32 && resolution != 53)) return B7_ERR_INVALID;
  Mt19937 g((uint32_t)seed);
  if (resolution == 32)
    for (int64_t i = 0; i < n; ++i) out[i] = (double)g.next() * (1.0 / 4294967296.0);
  else
    for (int64_t i = 0; i < n; ++i) {
      const uint64_t hi = g.next(), lo = g.next();
      out[i] = (double)(((hi << 32) | lo) & ((1ull << 53) - 1)) * 1.1102230246251565404e-16;
    }
  return B7_OK;
}

static int grid_random_torch_local(b7_ctx *c, int64_t size, int dims, uint64_t seed, int resolution, const double *mins,
                                   const double *maxes) {
  if (size < 0 || (resolution != 32 && resolution != 53))
    return b7_fail(c, B7_ERR_INVALID, "random grid (torch stream): size >= 0, resolution 32 or 53");
  std::vector<double> u((size_t)size * (dims > 0 ? dims : 0));
  if (dims >= 1 && b7_torch_rand(seed, (int64_t)u.size(), resolution, u.data()) != B7_OK)
    return b7_fail(c, B7_ERR_INVALID, "random grid (torch stream): bad arguments");
  B7_TRY(b7_grid_upload(c, u.data(), size, dims));
  if (mins && maxes && size > 0) {   // grids/random.lua:27-28: cmul by (maxes + -mins), then add mins: two rounded operations
    double *stage = reinterpret_cast<double *>(static_cast<char *>(c->pinned) + 8192);
    double *v_dev = (double *)((char *)c->scratch.p + 8192 + 4096);
    for (int pass = 0; pass < 2; ++pass) {
      B7_HIP(c, hipStreamSynchronize(c->stream));
      for (int k = 0; k < dims; ++k) stage[k] = pass == 0 ? maxes[k] + (-mins[k]) : mins[k];
      B7_HIP(c, hipMemcpyAsync(v_dev, stage, sizeof(double) * dims, hipMemcpyHostToDevice, c->stream));
      B7_TRY(launch_col_affine(c, cur_grid(c), c->M, dims, v_dev, pass == 0));
    }
  }
  return B7_OK;
}

int b7_grid_random_torch(b7_ctx *c, int64_t size, int dims, uint64_t seed, int resolution, const double *mins,
                         const double *maxes, double *out_host) {
  if (!c) return B7_ERR_INVALID;
  int rc = grid_random_torch_local(c, size, dims, seed, resolution, mins, maxes);
  if ((mins != nullptr) != (maxes != nullptr)) rc = onesided(c, mins, maxes, rc);  // collective with a communicator: entered on failure too
  B7_TRY(rc);
  return grid_copy_out(c, out_host);
}

int b7_grid_colrange(b7_ctx *c, double *col_min, double *col_max) {
  if (!c) return B7_ERR_INVALID;
  if (c->M <= 0) return b7_fail(c, B7_ERR_STATE, "grid_colrange: no candidate grid on this context");
  B7_HIP(c, hipSetDevice(c->device));
  const int d = c->d;
  double *out_dev = (double *)((char *)c->scratch.p + 8192 + 4096);  // 2 x 96 doubles, clear of the Sobol table
  B7_TRY(launch_colrange(c, cur_grid(c), c->M, d, out_dev));
  std::vector<double> h(2 * (size_t)d);
  B7_HIP(c, hipMemcpyAsync(h.data(), out_dev, sizeof(double) * 2 * d, hipMemcpyDeviceToHost, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  if (col_min) memcpy(col_min, h.data(), sizeof(double) * d);
  if (col_max) memcpy(col_max, h.data() + d, sizeof(double) * d);
  return B7_OK;
}

int b7_grid_apply_onesided(b7_ctx *c, const double *mins, const double *maxes, const double *col_ext) {
  if (!c) return B7_ERR_INVALID;
  if ((mins != nullptr) == (maxes != nullptr) || !col_ext)
    return b7_fail(c, B7_ERR_INVALID, "grid_apply_onesided: exactly one of mins / maxes, and the column extremes");
  B7_TRY(group_guard(c, "grid_apply_onesided"));
  B7_HIP(c, hipSetDevice(c->device));
  const int d = c->d;
  // torch.add(mins, grid:min(1)[1]) (grids/sobol.lua:83) / torch.cdiv(maxes, grid:max(1)[1]) (:85): one rounded operation per
  // column, here; then one per element on the device
  double *stage = reinterpret_cast<double *>(static_cast<char *>(c->pinned) + 8192);  // the lengthscale staging slot
  B7_HIP(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < d; ++k) stage[k] = mins ? mins[k] + col_ext[k] : maxes[k] / col_ext[k];
  double *v_dev = (double *)((char *)c->scratch.p + 8192 + 4096);
  B7_HIP(c, hipMemcpyAsync(v_dev, stage, sizeof(double) * d, hipMemcpyHostToDevice, c->stream));
  B7_TRY(launch_col_affine(c, cur_grid(c), c->M, d, v_dev, maxes != nullptr));
  invalidate_predictions(c);
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

int b7_grid_upload(b7_ctx *c, const double *X, int64_t M, int d) {
  if (!c) return B7_ERR_INVALID;
  if (!X && M > 0) return b7_fail(c, B7_ERR_INVALID, "grid_upload: X is NULL");
  B7_TRY(grid_alloc(c, M, d));
  if (M > 0)
    B7_HIP(c, hipMemcpyAsync(cur_grid(c), X, sizeof(double) * (size_t)M * d, hipMemcpyHostToDevice, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

int b7_grid_download(b7_ctx *c, int64_t row0, int64_t rows, double *out_host) {
  if (!c) return B7_ERR_INVALID;
  if (row0 < 0 || rows < 0 || row0 + rows > c->M || (!out_host && rows > 0))
    return b7_fail(c, B7_ERR_INVALID, "grid_download: rows [%lld, %lld) outside [0, %lld)", (long long)row0,
                   (long long)(row0 + rows), (long long)c->M);
  if (rows > 0)
    B7_HIP(c, hipMemcpyAsync(out_host, cur_grid(c) + row0 * c->d, sizeof(double) * (size_t)rows * c->d,
                             hipMemcpyDeviceToHost, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

int b7_grid_shape(b7_ctx *c, int64_t *M, int *d) {
  if (!c) return B7_ERR_INVALID;
  if (M) *M = c->M;
  if (d) *d = c->d;
  return B7_OK;
}

int b7_grid_remove(b7_ctx *c, int64_t idx1, double *row_out) {
  if (!c) return B7_ERR_INVALID;
  B7_TRY(grid_drop_row(c, idx1, row_out));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

int b7_grid_remove_rows(b7_ctx *c, const int64_t *idx1, int64_t n, double *rows_out) {
  if (!c) return B7_ERR_INVALID;
  if (n < 0 || (n > 0 && !idx1)) return b7_fail(c, B7_ERR_INVALID, "grid_remove_rows: bad index list");
  B7_TRY(group_guard(c, "grid_remove_rows"));
  if (n == 0) return B7_OK;
  for (int64_t i = 0; i < n; ++i)
    if (idx1[i] < 1 || idx1[i] > c->M)
      return b7_fail(c, B7_ERR_INVALID, "grid_remove_rows: index %lld outside [1, %lld]", (long long)idx1[i],
                     (long long)c->M);
  B7_HIP(c, hipSetDevice(c->device));
  // keep:indexFill(1, idx, 0) is idempotent: duplicates remove the row once (utils/tensor.lua:162)
  std::vector<int64_t> idx0(idx1, idx1 + n), cuts(idx1, idx1 + n);
  for (int64_t &v : idx0) v -= 1;
  std::sort(cuts.begin(), cuts.end());
  cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
  const int64_t ncut = (int64_t)cuts.size();
  for (int64_t i = 0; i < ncut; ++i) cuts[i] = (cuts[i] - 1) - i;
  B7_TRY(b7_ensure(c, c->tmpvar, sizeof(int64_t) * (size_t)(n + ncut)));
  int64_t *idx0_dev = (int64_t *)c->tmpvar.p, *cuts_dev = idx0_dev + n;
  B7_HIP(c, hipMemcpyAsync(idx0_dev, idx0.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice, c->stream));
  B7_HIP(c, hipMemcpyAsync(cuts_dev, cuts.data(), sizeof(int64_t) * ncut, hipMemcpyHostToDevice, c->stream));
  if (rows_out) {
    B7_TRY(b7_ensure(c, c->tmpmu, sizeof(double) * (size_t)n * c->d));
    B7_TRY(launch_gather_rows(c, cur_grid(c), (double *)c->tmpmu.p, idx0_dev, n, c->d));
    B7_HIP(c, hipMemcpyAsync(rows_out, c->tmpmu.p, sizeof(double) * (size_t)n * c->d, hipMemcpyDeviceToHost, c->stream));
  }
  const int other = c->grid_cur ^ 1;
  B7_TRY(b7_ensure(c, c->grid[other], sizeof(double) * (size_t)c->M * c->d));
  B7_TRY(launch_remove_rows(c, cur_grid(c), (double *)c->grid[other].p, c->M, c->d, cuts_dev, (int)ncut));
  c->grid_cur = other;
  c->M -= ncut;
  invalidate_predictions(c);
  B7_HIP(c, hipStreamSynchronize(c->stream));  // idx0 / cuts (pageable) are consumed, rows_out is complete
  return B7_OK;
}

// ---- model ---------------------------------------------------------------------------------------------
int b7_gp_default_opts(b7_gp_opts *o) {
  if (!o) return B7_ERR_INVALID;
  o->jitter_eps = 1e-8;     // utils/math.lua:175
  o->jitter_growth = 1.1;   // utils/math.lua:176
  o->var_with_noise = 0;
  o->var_clamp = 0;
  o->var_min = 0.0;
  return B7_OK;
}

int b7_gp_set_opts(b7_ctx *c, const b7_gp_opts *o) {
  if (!c || !o) return B7_ERR_INVALID;
  if (!(o->jitter_eps > 0.0) || !(o->jitter_growth > 1.0))
    return b7_fail(c, B7_ERR_INVALID, "gp opts: jitter_eps must be > 0 and jitter_growth > 1");
  c->opts = *o;
  return B7_OK;
}

// One factorisation attempt of K + extra*I; returns dpotrf-style info through *info.
static int try_factor(b7_ctx *c, double extra, int *info, bool with_inverse, FactorNote *note = nullptr);
void persist_gave_up(b7_ctx *c);
static int chol_with_jitter(b7_ctx *c, double *jitter_out, int *info_first_out, bool with_inverse, FactorNote *note = nullptr);
static int jitter_retries(b7_ctx *c, double *jitter_out, bool with_inverse, FactorNote *note = nullptr);

// Y - mean on the device (padding rows zero): the sampler changes only the hypers, the data stay where they are
__global__ void __launch_bounds__(256) resid_kernel(const double *__restrict__ y, double *__restrict__ r, int64_t nreal,
                                                    int64_t ntotal, double mean) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < ntotal) r[e] = e < nreal ? y[e] - mean : 0.0;
}

int b7_gp_set_data(b7_ctx *c, const double *X, const double *Y, int N, int d, int ycols) {
  if (!c) return B7_ERR_INVALID;
  if (!X || !Y) return b7_fail(c, B7_ERR_INVALID, "gp_set_data: NULL argument");
  if (N < 1 || d < 1 || ycols < 1) return b7_fail(c, B7_ERR_INVALID, "gp_set_data: N %d d %d ycols %d", N, d, ycols);
  if (d > B7_MAX_D) return b7_fail(c, B7_ERR_UNSUPPORTED, "gp_set_data: d %d > %d", d, B7_MAX_D);
  if (ycols > 256) return b7_fail(c, B7_ERR_UNSUPPORTED, "gp_set_data: ycols %d > 256", ycols);
  B7_HIP(c, hipSetDevice(c->device));
  c->fitted = false;
  c->have_data = false;
  c->predicted = false;  // the score accumulator survives: marginalisation adds across fits (bots/bayesopt.lua:73-78)
  c->N = N;
  c->Npad = npad_of(c, N);
  c->dfit = d;
  c->dpad = b7_dpad_class(d);
  c->ycols = ycols;
  c->yld = (ycols == 1) ? 1 : (int)round_up(ycols, 64);
  const size_t np = (size_t)c->Npad, nn = np * np * sizeof(double);
  B7_TRY(b7_ensure(c, c->xobs, sizeof(double) * np * d));  // room for b7_gp_append up to Npad rows
  B7_TRY(b7_ensure(c, c->ybuf, sizeof(double) * np * ycols));
  B7_TRY(b7_ensure(c, c->w, sizeof(double) * c->dpad));
  B7_TRY(b7_ensure(c, c->zsc, sizeof(double) * np * c->dpad));
  B7_TRY(b7_ensure(c, c->zss, sizeof(double) * np));
  B7_TRY(b7_ensure(c, c->K, nn));
  B7_TRY(b7_ensure(c, c->L, nn));
  B7_TRY(b7_ensure(c, c->Linv, nn));
  B7_TRY(b7_ensure(c, c->W, nn));
  B7_TRY(b7_ensure(c, c->dinv, sizeof(double) * (np + B7_PANEL) * B7_PANEL));
  B7_TRY(b7_ensure(c, c->alpha, sizeof(double) * np * c->yld));
  B7_TRY(b7_ensure(c, c->resid, sizeof(double) * np * ycols));
  B7_TRY(b7_ensure(c, c->info, B7_INFO_BYTES));
  B7_HIP(c, hipMemcpyAsync(c->xobs.p, X, sizeof(double) * (size_t)N * d, hipMemcpyHostToDevice, c->stream));
  B7_HIP(c, hipMemcpyAsync(c->ybuf.p, Y, sizeof(double) * (size_t)N * ycols, hipMemcpyHostToDevice, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));  // the caller's arrays are consumed
  c->have_data = true;
  return B7_OK;
}

static int predict_into(b7_ctx *c, const double *xq, int64_t M, double *mu, double *var);
static int check_hyp(b7_ctx *c, const b7_hyp *hyp, int d);
static int fit_front(b7_ctx *c, const b7_hyp *hyp, const double *ls_dev);

// then_predict: the posterior over the resident grid is enqueued right behind the fit, before the host has seen the
// pivot report, and the host only waits for the report (an event), not for the prediction.  If the report says the
// plain attempt failed (rare), the speculative prediction is thrown away and redone after the jitter schedule.
static int fit_hyp_core(b7_ctx *c, const b7_hyp *hyp, double *nll_out, double *jitter_used, int *info_out,
                        bool then_predict) {
  if (!c) return B7_ERR_INVALID;
  if (!c->have_data) return b7_fail(c, B7_ERR_STATE, "gp_fit_hyp: call b7_gp_set_data first");
  const int N = c->N, d = c->dfit, ycols = c->ycols;
  B7_TRY(check_hyp(c, hyp, d));
  B7_HIP(c, hipSetDevice(c->device));

  // the only upload of a fit: d lengthscales, through the pinned staging slot [8192, 8960) (free again once the
  // synchronisation that ends every fit has passed); amp / noise / mean travel as kernel arguments
  double *ls_stage = reinterpret_cast<double *>(static_cast<char *>(c->pinned) + 8192);
  double *ls_dev = (double *)c->scratch.p;  // first 4096 bytes of scratch: up to 512 doubles
  memcpy(ls_stage, hyp->lenscale_sq, sizeof(double) * d);
  // N <= 128, d <= 32, one response column: residual, observation scaling, K(X,X), factorisation, inverse and alpha are ONE
  // workgroup of ONE launch (gp_small.hip) with the d + 3 hypers in the kernel arguments; the kernel also leaves the
  // lengthscales where b7_gp_append / b7_gp_fantasize look for the current fit's.  K itself is not kept: a failed pivot
  // (rare) assembles it through the general front end before the jitter schedule runs
  const bool small = c->fit_small && c->potrf_sched == 3 && c->inverse_inline && gp_small_applies(c);
  if (small) {
    ls_stage[d] = hyp->amp, ls_stage[d + 1] = hyp->noise, ls_stage[d + 2] = hyp->mean;
    c->fitted = false;
    c->model_kind = 0;
    c->predicted = false;
    c->amp = hyp->amp;
    c->noise = hyp->noise;
    c->mean = hyp->mean;
  } else {
    B7_HIP(c, hipMemcpyAsync(ls_dev, ls_stage, sizeof(double) * d, hipMemcpyHostToDevice, c->stream));
    B7_TRY(fit_front(c, hyp, ls_dev));
  }

  // First attempt with everything that follows it enqueued BEFORE the host looks at the pivot report: when the
  // inverse came out of the factorisation itself, alpha and the likelihood terms do not need the host, and one
  // small copy (info + terms) with one synchronisation ends the fit.  A failed pivot (rare) falls back to the
  // jitter schedule and redoes the tail.
  struct FitBlock { int info[4]; double terms[257]; };
  static_assert(sizeof(FitBlock) <= 2304, "pinned block: [0, 2304) fit report, [2304, ..) arg-max result, [4096, 6144) fmin");
  FitBlock &blk = *static_cast<FitBlock *>(c->pinned);  // pinned: the copy needs no pageable staging
  double *terms_dev = reinterpret_cast<double *>(reinterpret_cast<char *>(c->info.p) + 16);
  const size_t blk_bytes = 16 + sizeof(double) * (nll_out ? 1 + ycols : 0);
  bool tail_done = false;
  FactorNote note;  // what the last factorisation already did for the launch_alpha behind it
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (small && attempt == 0) {
      B7_TRY(launch_fit_small(c, 1, nullptr, ls_stage, ls_dev, (double *)c->w.p, (double *)c->zsc.p, (double *)c->zss.p, (double *)c->L.p,
                              (double *)c->Linv.p, (double *)c->dinv.p, (double *)c->alpha.p, (double *)c->resid.p, (int *)c->info.p,
                              nullptr));
      c->linv_done = true;
      tail_done = true;
      if (nll_out) B7_TRY(launch_nll_terms(c, terms_dev));
    } else {
    B7_TRY(launch_potrf(c, 0.0, true, nullptr, &note));
    tail_done = c->linv_done;
    if (tail_done) {
      B7_TRY(launch_alpha(c, nullptr, 0, note));
      if (nll_out) B7_TRY(launch_nll_terms(c, terms_dev));
    }
    }
    B7_HIP(c, hipMemcpyAsync(&blk, c->info.p, tail_done ? blk_bytes : 16, hipMemcpyDeviceToHost, c->stream));
    if (then_predict && tail_done) {
      B7_HIP(c, hipEventRecord(c->ev_fit, c->stream));
      c->fitted = true;  // for the launchers; withdrawn below if the report says otherwise
      B7_TRY(predict_into(c, (const double *)c->grid[c->grid_cur].p, c->M, (double *)c->mu.p, (double *)c->var.p));
      B7_HIP(c, hipEventSynchronize(c->ev_fit));
    } else {
      B7_HIP(c, hipStreamSynchronize(c->stream));
    }
    if (blk.info[1] == 0) break;
    c->fitted = false;
    // the persistent schedule gave up on a hand-off (its workgroups were not all resident, e.g. the GPU is shared
    // with another process's persistent kernel): same arithmetic through the launch schedule, which cannot stall
    if (attempt == 1) return b7_fail(c, B7_ERR_HIP, "Cholesky: hand-off time-out (code %d) outside the persistent schedule", blk.info[1]);
    persist_gave_up(c);
  }
  const int info_first = blk.info[0];
  double jitter = 0.0;
  bool predicted = then_predict && tail_done;
  if (info_first != 0) {
    c->fitted = false;
    if (small) B7_TRY(fit_front(c, hyp, ls_dev));  // K(X,X) for the retries (the one-launch fit does not keep it)
    B7_TRY(jitter_retries(c, &jitter, true, &note));
    tail_done = false;
    predicted = false;
  }
  if (!tail_done) {
    B7_TRY(launch_trtri(c));
    B7_TRY(launch_alpha(c, nullptr, 0, note));
    if (nll_out) B7_TRY(launch_nll_terms(c, terms_dev));
    B7_HIP(c, hipMemcpyAsync(&blk, c->info.p, blk_bytes, hipMemcpyDeviceToHost, c->stream));
    B7_HIP(c, hipStreamSynchronize(c->stream));
  }
  if (then_predict && !predicted) {
    c->fitted = true;
    B7_TRY(predict_into(c, (const double *)c->grid[c->grid_cur].p, c->M, (double *)c->mu.p, (double *)c->var.p));
  }
  if (nll_out)
    for (int k = 0; k < ycols; ++k) nll_out[k] = 0.5 * blk.terms[1 + k] + blk.terms[0] + 0.5 * N * log(2.0 * M_PI);
  if (jitter_used) *jitter_used = jitter;
  if (info_out) *info_out = info_first;
  if (c->potrf_sched_saved == 3 && c->persist_aborts < 3) c->potrf_sched = 3;
  c->fitted = true;
  return B7_OK;
}

__global__ void __launch_bounds__(256) resid_batch_kernel(const double *__restrict__ y, double *__restrict__ r, int N,
                                                          int Npad, const double *__restrict__ mean) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i < Npad) r[(int64_t)b * Npad + i] = i < N ? y[i] - mean[b] : 0.0;
}

int b7_gp_nll_batch(b7_ctx *c, int B, const double *lenscale_sq, const double *amp, const double *noise,
                    const double *mean, double *nll_out, double *jitter_out, int *info_out) {
  if (!c) return B7_ERR_INVALID;
  if (!c->have_data) return b7_fail(c, B7_ERR_STATE, "gp_nll_batch: call b7_gp_set_data first");
  if (B < 1 || !lenscale_sq || !amp || !noise || !mean || !nll_out) return b7_fail(c, B7_ERR_INVALID, "gp_nll_batch: bad arguments");
  if (c->ycols != 1) return b7_fail(c, B7_ERR_UNSUPPORTED, "gp_nll_batch: one response column only");
  if (c->Npad > B7_PERSIST_NMAX) return b7_fail(c, B7_ERR_UNSUPPORTED, "gp_nll_batch: N > 4096 (evaluate with b7_gp_fit_hyp one by one)");
  const int N = c->N, n = c->Npad, d = c->dfit, dpad = c->dpad, nb = n / B7_PANEL;
  for (int b = 0; b < B; ++b) {
    for (int k = 0; k < d; ++k)
      if (!(lenscale_sq[(size_t)b * d + k] > 0.0)) return b7_fail(c, B7_ERR_INVALID, "gp_nll_batch: lenscale_sq[%d][%d] must be > 0", b, k);
    if (!(amp[b] > 0.0) || !(noise[b] >= 0.0)) return b7_fail(c, B7_ERR_INVALID, "gp_nll_batch: amp > 0, noise >= 0 (fit %d)", b);
  }
  B7_HIP(c, hipSetDevice(c->device));
  // hypers in and results out through ONE block of pinned, device-mapped host memory, laid out
  // [B x d lengthscales | B amp | B noise | B mean][2 B terms][4 B ints of pivot reports][completion word]
  const size_t hyp_doubles = (size_t)B * (d + 3), need = sizeof(double) * (hyp_doubles + 2 * (size_t)B) + sizeof(int) * (4 * (size_t)B + 4);
  if (c->pin_nll_bytes < need) {
    B7_HIP(c, hipStreamSynchronize(c->stream));
    if (c->pin_nll) (void)hipHostFree(c->pin_nll);
    c->pin_nll = nullptr;
    c->pin_nll_bytes = 0;
    B7_HIP(c, hipHostMalloc(&c->pin_nll, 2 * need, hipHostMallocMapped));
    B7_HIP(c, hipHostGetDevicePointer(&c->pin_nll_dev, c->pin_nll, 0));
    c->pin_nll_bytes = 2 * need;
  }
  double *pack = static_cast<double *>(c->pin_nll);
  memcpy(pack, lenscale_sq, sizeof(double) * (size_t)B * d);
  memcpy(pack + (size_t)B * d, amp, sizeof(double) * B);
  memcpy(pack + (size_t)B * (d + 1), noise, sizeof(double) * B);
  memcpy(pack + (size_t)B * (d + 2), mean, sizeof(double) * B);
  if (c->nll_small && gp_small_applies(c)) {
    // N <= 128, d <= 32: every evaluation is ONE workgroup of ONE launch (gp_small.hip), observations in, two numbers out.
    // A fit whose plain factorisation fails (rare) sends the whole batch through the general path below, jitter schedule
    // included.  The kernel reads the B x (d + 3) numbers and writes its 2 doubles + 4 ints per evaluation straight across
    // the bus -- no copy calls, one launch, one wait
    const double *terms = pack + hyp_doubles;
    const int *info = reinterpret_cast<const int *>(terms + 2 * (size_t)B);
    double *pack_dev = static_cast<double *>(c->pin_nll_dev);
    int *info_dev = reinterpret_cast<int *>(pack_dev + hyp_doubles + 2 * (size_t)B);
    // a single evaluation (every density call of the slice sampler) is waited for on a word the kernel sets after its results:
    // the host sees them as soon as they have crossed the bus instead of after the dispatch has retired and the runtime
    // has noticed.  The stream stays ordered (later launches queue behind the kernel); a kernel that has not answered after
    // 200 us is waited for the ordinary way, which also surfaces a fault.
    volatile unsigned *done = reinterpret_cast<volatile unsigned *>(const_cast<int *>(info) + 4 * (size_t)B);
    *done = 0u;
#ifdef B7_DIAG
    if (c->nll_small == 2)  // round 3's four-wave kernel (nll_small.hip, diagnostic build only): the bit-for-bit reference of the likelihood
      B7_TRY(launch_nll_small(c, B, pack_dev, pack, pack_dev + hyp_doubles, info_dev, reinterpret_cast<unsigned *>(info_dev + 4 * (size_t)B)));
    else
#endif
      B7_TRY(launch_nll_small8(c, B, pack_dev, pack, pack_dev + hyp_doubles, info_dev, reinterpret_cast<unsigned *>(info_dev + 4 * (size_t)B)));
    bool answered = false;
    if (B == 1) {
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned spins = 0; !answered && c->spin_us > 0; ++spins) {
        answered = __atomic_load_n(const_cast<const unsigned *>(done), __ATOMIC_ACQUIRE) != 0u;
        if (!answered && (spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(c->spin_us < 200 ? c->spin_us : 200)) break;
      }
    }
    if (!answered) B7_HIP(c, hipStreamSynchronize(c->stream));
    bool clean = true;
    for (int b = 0; b < B; ++b) clean = clean && info[(size_t)b * 4] == 0;
    if (clean) {
      const double c0s = 0.5 * N * log(2.0 * M_PI);
      for (int b = 0; b < B; ++b) {
        nll_out[b] = 0.5 * terms[(size_t)b * 2] + terms[(size_t)b * 2 + 1] + c0s;
        if (jitter_out) jitter_out[b] = 0.0;
        if (info_out) info_out[b] = 0;
      }
      return B7_OK;
    }
  }
  const size_t fw = persist_flag_words_host(nb), nn = (size_t)n * n;
  B7_TRY(b7_ensure(c, c->bhyp, sizeof(double) * (size_t)B * (d + 3)));
  B7_TRY(b7_ensure(c, c->bw, sizeof(double) * (size_t)B * dpad));
  B7_TRY(b7_ensure(c, c->bzsc, sizeof(double) * (size_t)B * n * dpad));
  B7_TRY(b7_ensure(c, c->bzss, sizeof(double) * (size_t)B * n));
  B7_TRY(b7_ensure(c, c->bK, sizeof(double) * B * nn));
  B7_TRY(b7_ensure(c, c->bL, sizeof(double) * B * nn));
  B7_TRY(b7_ensure(c, c->bdinv, sizeof(double) * (size_t)B * n * B7_PANEL));
  // one block: [2 B doubles of likelihood terms][4 B ints of pivot reports][B x fw flag words] -- reports and flags are zeroed
  // by one memset, terms and reports come back in one copy; the jitter schedule's norm goes into bterms
  const size_t head_bytes = sizeof(double) * 2 * (size_t)B + sizeof(int) * 4 * (size_t)B;
  B7_TRY(b7_ensure(c, c->bflags, head_bytes + sizeof(unsigned) * B * fw));
  B7_TRY(b7_ensure(c, c->bresid, sizeof(double) * (size_t)B * n));
  B7_TRY(b7_ensure(c, c->bterms, 64));
  // all hypers in one upload from the pinned block (no pageable staging)
  double *hyp_dev = (double *)c->bhyp.p;
  B7_HIP(c, hipMemcpyAsync(hyp_dev, pack, sizeof(double) * hyp_doubles, hipMemcpyHostToDevice, c->stream));
  const double *ls_dev = hyp_dev, *amp_dev = hyp_dev + (size_t)B * d, *noise_dev = amp_dev + B, *mean_dev = noise_dev + B;
  hipLaunchKernelGGL(resid_batch_kernel, dim3((n + 255) / 256, B), dim3(256), 0, c->stream, (const double *)c->ybuf.p,
                     (double *)c->bresid.p, N, n, mean_dev);
  B7_TRY(launch_kxx_batch(c, B, ls_dev, amp_dev, noise_dev, (double *)c->bw.p, (double *)c->bzsc.p, (double *)c->bzss.p,
                          (double *)c->bK.p));
  // the likelihood terms and the pivot reports of all fits sit side by side on the device ([2 B doubles][4 B ints]) and come
  // back in ONE copy into the pinned block
  double *terms_dev = (double *)c->bflags.p;
  int *info_dev = reinterpret_cast<int *>(terms_dev + 2 * (size_t)B);
  unsigned *flags_dev = reinterpret_cast<unsigned *>(info_dev + 4 * (size_t)B);
  B7_TRY(launch_nll_batch(c, B, (const double *)c->bK.p, (double *)c->bL.p, (double *)c->bdinv.p, flags_dev,
                          info_dev, (const double *)c->bresid.p, terms_dev, nullptr));
  double *terms = pack + hyp_doubles;
  int *info = reinterpret_cast<int *>(terms + 2 * (size_t)B);
  B7_HIP(c, hipMemcpyAsync(terms, terms_dev, sizeof(double) * 2 * (size_t)B + sizeof(int) * 4 * (size_t)B, hipMemcpyDeviceToHost, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  const double c0 = 0.5 * N * log(2.0 * M_PI);
  for (int b = 0; b < B; ++b) {
    double jitter = 0.0;
    const int info_first = info[(size_t)b * 4];
    int bad = info_first, aborted = info[(size_t)b * 4 + 1];
    const double *Kb = (const double *)c->bK.p + b * nn;
    double *Lb = (double *)c->bL.p + b * nn, *dib = (double *)c->bdinv.p + (size_t)b * n * B7_PANEL;
    unsigned *fb = flags_dev + b * fw;
    int *ib = info_dev + (size_t)b * 4;
    const double *rb = (const double *)c->bresid.p + (size_t)b * n;
    double *tb = terms_dev + 2 * (size_t)b;
    auto redo = [&](double extra) -> int {  // this fit alone (it has the whole chip), eps on the diagonal
      B7_TRY(launch_nll_one(c, Kb, Lb, dib, fb, ib, rb, tb, extra));
      int two[2];
      B7_HIP(c, hipMemcpyAsync(two, ib, sizeof(two), hipMemcpyDeviceToHost, c->stream));
      B7_HIP(c, hipMemcpyAsync(&terms[(size_t)b * 2], tb, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      B7_HIP(c, hipStreamSynchronize(c->stream));
      bad = two[0];
      aborted = two[1];
      return B7_OK;
    };
    if (aborted) {  // a hand-off timed out (the chip was shared): once more, alone
      c->persist_aborts += 1;
      B7_TRY(redo(0.0));
    }
    if (aborted) {
      // still no luck with the persistent schedule (the GPU is occupied by somebody else's persistent kernel): this
      // likelihood through the launch schedule, which cannot stall -- b7_gp_fit_hyp's own fallback, jitter schedule
      // included.  It runs in the context's fit slot, so the current fit is gone afterwards (the one case in which this
      // call does not leave it alone; the next predict asks for a fit with B7_ERR_STATE).
      b7_hyp h{lenscale_sq + (size_t)b * d, amp[b], noise[b], mean[b]};
      persist_gave_up(c);
      double nll1 = 0.0, jit1 = 0.0;
      int info1 = 0;
      const int rc1 = fit_hyp_core(c, &h, &nll1, &jit1, &info1, false);
      c->fitted = false;
      c->predicted = false;
      B7_TRY(rc1);
      nll_out[b] = nll1;
      if (jitter_out) jitter_out[b] = jit1;
      if (info_out) info_out[b] = info1;
      continue;
    }
    if (bad != 0) {  // the jitter schedule of utils/math.lua:174-202 for this fit
      double *fro_dev = (double *)c->bterms.p;
      B7_TRY(launch_fro_norm_sq(c, Kb, N, n, fro_dev));
      double fro = 0.0;
      B7_HIP(c, hipMemcpyAsync(&fro, fro_dev, sizeof(double), hipMemcpyDeviceToHost, c->stream));
      B7_HIP(c, hipStreamSynchronize(c->stream));
      const double max_eps = sqrt(fro);
      if (max_eps != max_eps) return b7_fail(c, B7_ERR_INVALID, "gp_nll_batch: K of fit %d contains NaN", b);
      double eps = c->opts.jitter_eps;
      while (bad != 0) {
        if (eps > max_eps) return b7_fail(c, B7_ERR_INVALID, "gp_nll_batch: fit %d cannot be factored", b);
        eps = eps * c->opts.jitter_growth;
        B7_TRY(redo(eps));
        if (aborted) return b7_fail(c, B7_ERR_HIP, "gp_nll_batch: hand-off time-out (code %d) in fit %d", aborted, b);
        jitter = eps;
      }
    }
    nll_out[b] = 0.5 * terms[(size_t)b * 2] + terms[(size_t)b * 2 + 1] + c0;
    if (jitter_out) jitter_out[b] = jitter;
    if (info_out) info_out[b] = info_first;
  }
  return B7_OK;
}

int b7_gp_fit_hyp(b7_ctx *c, const b7_hyp *hyp, double *nll_out, double *jitter_used, int *info_out) {
  return fit_hyp_core(c, hyp, nll_out, jitter_used, info_out, false);
}

int b7_gp_predict_hyp(b7_ctx *c, const b7_hyp *hyp, double *mean_host, double *var_host, double *nll_out,
                      double *jitter_used, int *info_out) {
  if (!c) return B7_ERR_INVALID;
  if (!c->have_data) return b7_fail(c, B7_ERR_STATE, "gp_predict_hyp: call b7_gp_set_data first");
  if (c->M <= 0) return b7_fail(c, B7_ERR_STATE, "gp_predict_hyp: no candidate grid on this context");
  if (c->d != c->dfit) return b7_fail(c, B7_ERR_INVALID, "gp_predict_hyp: grid dims %d != data dims %d", c->d, c->dfit);
  B7_HIP(c, hipSetDevice(c->device));
  B7_TRY(b7_ensure(c, c->mu, sizeof(double) * (size_t)c->M * c->ycols));
  B7_TRY(b7_ensure(c, c->var, sizeof(double) * (size_t)c->M));
  B7_TRY(fit_hyp_core(c, hyp, nll_out, jitter_used, info_out, true));
  c->predicted = true;
  c->Mpred = c->M;
  if (mean_host)
    B7_HIP(c, hipMemcpyAsync(mean_host, c->mu.p, sizeof(double) * (size_t)c->M * c->ycols, hipMemcpyDeviceToHost,
                             c->stream));
  if (var_host)
    B7_HIP(c, hipMemcpyAsync(var_host, c->var.p, sizeof(double) * (size_t)c->M, hipMemcpyDeviceToHost, c->stream));
  if (mean_host || var_host) B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

// ---- bayesopt:eval + nominate as one call ------------------------------------------------------------------
// bots/bayesopt.lua:56-99: score = (1/S) sum_s acq(model, hyp_s, X_obs, Y_obs, X_hid), then score:max(1).  The
// separate entry points (b7_gp_predict_hyp, b7_score_*, b7_score_finish*) cost one host round trip per hyper sample
// (the pivot report) plus one for the arg-max; at small N and M (cfg2: 0.2 ms of GPU work per sample) the round trips
// are a third of the wall time.  Here every sample's fit, posterior and score:add are enqueued back to back, each
// fit's 16-byte pivot report is copied into its own pinned slot in stream order, the arg-max follows, and the host
// synchronises ONCE.  Knowing all S samples up front also lets the S fits run SIDE BY SIDE: one persistent launch with
// grid.y = sample (launch_fit_batch), K assembly, residuals and alpha batched the same way; a fit is a dependent chain
// that leaves most of the chip idle, so ten cost little more than one (cfg2, S = 10: 1.55 -> 0.75 ms per nomination).  A report that says "pivot failed" or "hand-off timed out" (rare) throws the accumulated score
// away and redoes the whole nomination through the per-sample path, jitter schedule included, so the result is the
// one the separate calls give.
static int stage_fmin(b7_ctx *c, const double *fmin, double **fd_out) {
  if (c->ycols == 1) {  // one response column (every path but the fantasy scores): f_min travels as a kernel argument
    c->fmin_scalar = fmin[0];
    *fd_out = nullptr;
    return B7_OK;
  }
  double *fh = reinterpret_cast<double *>(static_cast<char *>(c->pinned) + 4096);
  double *fd = (double *)((char *)c->scratch.p + 2048);
  if (!c->fmin_staged || memcmp(fh, fmin, sizeof(double) * c->ycols) != 0) {
    B7_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(fh, fmin, sizeof(double) * c->ycols);
    c->fmin_staged = true;
  }
  B7_HIP(c, hipMemcpyAsync(fd, fh, sizeof(double) * c->ycols, hipMemcpyHostToDevice, c->stream));
  *fd_out = fd;
  return B7_OK;
}

static int score_add(b7_ctx *c, const b7_score_spec *sp, const double *fd) {
  if (sp->kind == B7_SCORE_EI)
    return launch_ei(c, (const double *)c->mu.p, (const double *)c->var.p, fd, sp->tradeoff, c->M, c->ycols,
                     (double *)c->acc.p, true);
  return launch_cb(c, (const double *)c->mu.p, (const double *)c->var.p, sp->tradeoff, sp->upper, sp->sign, c->M,
                   c->ycols, (double *)c->acc.p, true);
}

static int check_hyp(b7_ctx *c, const b7_hyp *hyp, int d) {
  if (!hyp || !hyp->lenscale_sq) return b7_fail(c, B7_ERR_INVALID, "gp_fit_hyp: NULL argument");
  for (int k = 0; k < d; ++k)
    if (!(hyp->lenscale_sq[k] > 0.0)) return b7_fail(c, B7_ERR_INVALID, "gp_fit: lenscale_sq[%d] must be > 0", k);
  if (!(hyp->amp > 0.0) || !(hyp->noise >= 0.0)) return b7_fail(c, B7_ERR_INVALID, "gp_fit: amp > 0, noise >= 0");
  return B7_OK;
}

// residual, observation scaling and K(X,X) of one hyper sample whose lengthscales are (on their way) at ls_dev
static int fit_front(b7_ctx *c, const b7_hyp *hyp, const double *ls_dev) {
  const int N = c->N, d = c->dfit, ycols = c->ycols;
  c->fitted = false;
  c->model_kind = 0;
  c->predicted = false;
  c->amp = hyp->amp;
  c->noise = hyp->noise;
  c->mean = hyp->mean;
  const int64_t ntotal = (int64_t)c->Npad * ycols;
  hipLaunchKernelGGL(resid_kernel, dim3((unsigned)((ntotal + 255) / 256)), dim3(256), 0, c->stream,
                     (const double *)c->ybuf.p, (double *)c->resid.p, (int64_t)N * ycols, ntotal, hyp->mean);
  B7_TRY(launch_prep_obs(c, (const double *)c->xobs.p, ls_dev, N, d));
  return launch_kxx(c, hyp->noise);
}

}  // extern "C"

int npad_of(const b7_ctx *c, int64_t n) { return (c->npad_small && n <= 64) ? 64 : (int)round_up(n, B7_NPAD); }

int eval_validate(b7_ctx *c, int S, const b7_hyp *hyps, const b7_score_spec *spec) {
  if (S < 1 || !hyps || !spec) return b7_fail(c, B7_ERR_INVALID, "eval_nominate: S >= 1, hyps and spec required");
  if (spec->kind != B7_SCORE_EI && spec->kind != B7_SCORE_CB)
    return b7_fail(c, B7_ERR_INVALID, "eval_nominate: unknown score kind %d", spec->kind);
  if (spec->kind == B7_SCORE_EI && !spec->fmin) return b7_fail(c, B7_ERR_INVALID, "eval_nominate: EI needs fmin");
  if (!c->have_data) return b7_fail(c, B7_ERR_STATE, "eval_nominate: call b7_gp_set_data first");
  if (c->M > 0 && c->d != c->dfit)
    return b7_fail(c, B7_ERR_INVALID, "eval_nominate: grid dims %d != data dims %d", c->d, c->dfit);
  for (int s = 0; s < S; ++s) B7_TRY(check_hyp(c, &hyps[s], c->dfit));
  return B7_OK;
}

// bots/bayesopt.lua:69-78 as stream work: zero the accumulator, then fit + posterior + score:add per hyper sample, each
// fit's pivot report copied to its pinned slot in stream order.  Returns without waiting for any of it.
int eval_enqueue(b7_ctx *c, int S, const b7_hyp *hyps, const b7_score_spec *spec) {
  const int d = c->dfit;
  B7_HIP(c, hipSetDevice(c->device));
  B7_TRY(b7_ensure(c, c->mu, sizeof(double) * (size_t)c->M * c->ycols));
  B7_TRY(b7_ensure(c, c->var, sizeof(double) * (size_t)c->M));
  B7_TRY(b7_ensure(c, c->acc, sizeof(double) * (size_t)c->M));
  {  // predict_into's workspace, sized now: a reallocation inside the loop would synchronise
    const size_t row_bytes = sizeof(double) * (size_t)c->Npad;
    int64_t chunk = (int64_t)(c->ks_bytes / row_bytes) / B7_MROWS * B7_MROWS;
    if (chunk < B7_MROWS) chunk = B7_MROWS;
    const int64_t Mpad = round_up(c->M, B7_MROWS);
    B7_TRY(b7_ensure(c, c->ks, (size_t)(chunk > Mpad ? Mpad : chunk) * row_bytes));
  }
  // pinned staging: [S][4] pivot reports | hypers of all S samples ([S][d] lengthscales, then S amp, S noise, S mean)
  const size_t hyp_doubles = (size_t)S * (d + 3), ls_bytes = sizeof(double) * hyp_doubles, rep_bytes = 16 * (size_t)S;
  if (c->pin_eval_bytes < ls_bytes + rep_bytes) {
    B7_HIP(c, hipStreamSynchronize(c->stream));  // nothing in flight writes reports into the block about to go
    if (c->pin_eval) (void)hipHostFree(c->pin_eval);
    c->pin_eval = nullptr;
    c->pin_eval_bytes = 0;
    B7_HIP(c, hipHostMalloc(&c->pin_eval, 2 * (ls_bytes + rep_bytes), hipHostMallocMapped));
    B7_HIP(c, hipHostGetDevicePointer(&c->pin_eval_dev, c->pin_eval, 0));
    c->pin_eval_bytes = 2 * (ls_bytes + rep_bytes);
  }
  B7_TRY(b7_ensure(c, c->bhyp, ls_bytes));
  int *reports = static_cast<int *>(c->pin_eval);                                        // [S][4]
  double *ls_host = reinterpret_cast<double *>(static_cast<char *>(c->pin_eval) + rep_bytes);  // [S][d] | amp | noise | mean
  for (int s = 0; s < S; ++s) {
    memcpy(ls_host + (size_t)s * d, hyps[s].lenscale_sq, sizeof(double) * d);
    ls_host[(size_t)S * d + s] = hyps[s].amp;
    ls_host[(size_t)S * (d + 1) + s] = hyps[s].noise;
    ls_host[(size_t)S * (d + 2) + s] = hyps[s].mean;
  }
  memset(reports, 0xff, rep_bytes);
  // the fits of all samples side by side in one persistent launch (one critical workgroup each) when that schedule serves
  // this size and the responses are a single column; otherwise one after the other
  // N <= 128, d <= 32 (the reference's own regime): the whole fit of every hyper sample is one workgroup of ONE launch
  // (gp_small.hip), for any S -- observation scaling, K(X,X), factorisation, inverse, alpha, the pivot reports into the mapped
  // block; the kernel reads the hypers straight from that block and leaves the device copy the kernels downstream read
  const bool small = c->fit_small && c->potrf_sched == 3 && c->inverse_inline && gp_small_applies(c);
  const bool batch = small || (S > 1 && c->ycols == 1 && c->potrf_sched == 3 && c->Npad <= B7_PERSIST_NMAX && c->inverse_inline);
  const int n = c->Npad;
  const size_t nn = (size_t)n * n;
  if (batch) {
    const size_t fw = persist_flag_words_host(n / B7_PANEL);
    B7_TRY(b7_ensure(c, c->bw, sizeof(double) * (size_t)S * c->dpad));
    B7_TRY(b7_ensure(c, c->bzsc, sizeof(double) * (size_t)S * n * c->dpad));
    B7_TRY(b7_ensure(c, c->bzss, sizeof(double) * (size_t)S * n));
    B7_TRY(b7_ensure(c, c->bLinv, sizeof(double) * S * nn));
    B7_TRY(b7_ensure(c, c->binfo, sizeof(int) * 4 * (size_t)S));
    B7_TRY(b7_ensure(c, c->balpha, sizeof(double) * (size_t)S * n));
    if (!small) {
      B7_TRY(b7_ensure(c, c->bK, sizeof(double) * S * nn));
      B7_TRY(b7_ensure(c, c->bL, sizeof(double) * S * nn));
      B7_TRY(b7_ensure(c, c->bdinv, sizeof(double) * (size_t)S * n * B7_PANEL));
      B7_TRY(b7_ensure(c, c->bflags, sizeof(unsigned) * S * fw));
      B7_TRY(b7_ensure(c, c->bresid, sizeof(double) * (size_t)S * n));
    }
  }
  if (!small) B7_HIP(c, hipMemcpyAsync(c->bhyp.p, ls_host, ls_bytes, hipMemcpyHostToDevice, c->stream));
  double *fd = nullptr;
  if (spec->kind == B7_SCORE_EI) B7_TRY(stage_fmin(c, spec->fmin, &fd));

  c->pend.on = false;
  c->acc_fresh = true;  // torch.zeros(X_hid:size(1)), bots/bayesopt.lua:69: declared, not launched -- the first score:add starts from 0.0
  c->acc_valid = true;
  if (batch) {
    const double *hyp_dev = (const double *)c->bhyp.p, *amp_dev = hyp_dev + (size_t)S * d, *noise_dev = amp_dev + S,
                 *mean_dev = noise_dev + S;
    int *reports_dev = static_cast<int *>(c->pin_eval_dev);
    if (small) {
      const double *hyp_map = reinterpret_cast<const double *>(static_cast<const char *>(c->pin_eval_dev) + rep_bytes);
      B7_TRY(launch_fit_small(c, S, hyp_map, ls_host, (double *)c->bhyp.p, (double *)c->bw.p, (double *)c->bzsc.p, (double *)c->bzss.p,
                              nullptr, (double *)c->bLinv.p, nullptr, (double *)c->balpha.p, nullptr, (int *)c->binfo.p, reports_dev));
    } else {
    hipLaunchKernelGGL(resid_batch_kernel, dim3((n + 255) / 256, S), dim3(256), 0, c->stream, (const double *)c->ybuf.p,
                       (double *)c->bresid.p, c->N, n, mean_dev);
    B7_TRY(launch_kxx_batch(c, S, hyp_dev, amp_dev, noise_dev, (double *)c->bw.p, (double *)c->bzsc.p, (double *)c->bzss.p,
                            (double *)c->bK.p));
    if (n == 64 && c->potrf_small) {
      // one 64-block per fit: factorisation, inverse, alpha and the pivot report (mirrored into the mapped block) of all S fits
      // in ONE launch of S workgroups
      B7_TRY(launch_potrf_small(c, S, (const double *)c->bK.p, (double *)c->bL.p, (double *)c->bLinv.p, (double *)c->bdinv.p,
                                (const double *)c->bresid.p, (double *)c->balpha.p, 0.0, (int *)c->binfo.p, reports_dev, (int64_t)nn,
                                (int64_t)nn, (int64_t)n * B7_PANEL, (int64_t)n, 4));
    } else {
    B7_TRY(launch_fit_batch(c, S, (const double *)c->bK.p, (double *)c->bL.p, (double *)c->bLinv.p, (double *)c->bdinv.p,
                            (unsigned *)c->bflags.p, (int *)c->binfo.p));
    if (4 * S <= 256) {  // the reports ride on the last kernel of the fits into the mapped block: no copy launch
      B7_TRY(launch_alpha_batch(c, S, (const double *)c->bLinv.p, (const double *)c->bresid.p, (double *)c->balpha.p,
                                (const int *)c->binfo.p, reports_dev, 4 * S));
    } else {
      B7_TRY(launch_alpha_batch(c, S, (const double *)c->bLinv.p, (const double *)c->bresid.p, (double *)c->balpha.p));
      B7_HIP(c, hipMemcpyAsync(reports, c->binfo.p, rep_bytes, hipMemcpyDeviceToHost, c->stream));
    }
    }
    }
    const size_t row_bytes = sizeof(double) * (size_t)n;
    const int64_t Mpad = round_up(c->M, B7_MROWS);
    if (c->kpost_small && kpost_small_applies(c)) {
      // N <= 128, d <= 32: K(X*,X), mean and variance of all S samples in ONE kernel that never stores K* (kpost_small.hip)
      B7_TRY(b7_ensure(c, c->bmu, sizeof(double) * (size_t)S * c->M));
      B7_TRY(b7_ensure(c, c->bvar, sizeof(double) * (size_t)S * c->M));
      B7_TRY(launch_kpost_small(c, S, (const double *)c->grid[c->grid_cur].p, c->M, (const double *)c->bw.p, (const double *)c->bzsc.p,
                                (const double *)c->bzss.p, (const double *)c->bLinv.p, (const double *)c->balpha.p, hyp_dev, 0.0, 0.0, 0.0,
                                (double *)c->bmu.p, (double *)c->bvar.p, c->M));
      // score:add of the S samples is left to the exchange step, which runs it fused with score:div, the arg-max and the record
      c->pend.on = true;
      c->pend.kind = spec->kind, c->pend.S = S, c->pend.upper = spec->upper;
      c->pend.mu = (const double *)c->bmu.p, c->pend.var = (const double *)c->bvar.p, c->pend.fd = fd;
      c->pend.stride = c->M, c->pend.tradeoff = spec->tradeoff, c->pend.sign = spec->sign;
      c->fitted = false;     // neither the context's fit slot nor its mean / variance vectors hold any of these samples
      c->predicted = false;
    } else if ((size_t)Mpad * S * row_bytes <= c->ks_bytes) {
      // K* of all S samples fits the workspace at once (the reference's default sizes: 2e4 candidates, tens to hundreds
      // of observations, 10 samples): K*, posterior and score:add of all samples in ONE launch each (grid.z / grid.y =
      // sample), the score summed over the samples in order inside the kernel
      B7_TRY(b7_ensure(c, c->ks, (size_t)Mpad * S * row_bytes));
      B7_TRY(b7_ensure(c, c->bmu, sizeof(double) * (size_t)S * c->M));
      B7_TRY(b7_ensure(c, c->bvar, sizeof(double) * (size_t)S * c->M));
      B7_TRY(launch_ksx_batch(c, S, (const double *)c->grid[c->grid_cur].p, Mpad, c->M, (const double *)c->bw.p,
                              (const double *)c->bzsc.p, (const double *)c->bzss.p, amp_dev, mean_dev,
                              (const double *)c->balpha.p, (double *)c->ks.p, (int64_t)Mpad * n, (double *)c->bmu.p, c->M));
      B7_TRY(launch_post_batch(c, S, (const double *)c->bLinv.p, (const double *)c->ks.p, (int64_t)Mpad * n, Mpad, c->M,
                               (double *)c->bvar.p, c->M, amp_dev, noise_dev));
      if (spec->kind == B7_SCORE_EI)
        B7_TRY(launch_ei_batch(c, S, (const double *)c->bmu.p, (const double *)c->bvar.p, c->M, fd, spec->tradeoff, c->M,
                               (double *)c->acc.p));
      else
        B7_TRY(launch_cb_batch(c, S, (const double *)c->bmu.p, (const double *)c->bvar.p, c->M, spec->tradeoff, spec->upper,
                               spec->sign, c->M, (double *)c->acc.p));
      c->fitted = false;     // neither the context's fit slot nor its mean / variance vectors hold any of these samples
      c->predicted = false;
    } else {
      // the posterior kernels read the fit through the context: point it at one sample's slot after the other
      void *const zsc0 = c->zsc.p, *const zss0 = c->zss.p, *const w0 = c->w.p, *const alpha0 = c->alpha.p, *const linv0 = c->Linv.p;
      int rc = B7_OK;
      for (int s = 0; s < S && rc == B7_OK; ++s) {
        c->zsc.p = (double *)c->bzsc.p + (size_t)s * n * c->dpad;
        c->zss.p = (double *)c->bzss.p + (size_t)s * n;
        c->w.p = (double *)c->bw.p + (size_t)s * c->dpad;
        c->alpha.p = (double *)c->balpha.p + (size_t)s * n;
        c->Linv.p = (double *)c->bLinv.p + s * nn;
        c->amp = hyps[s].amp;
        c->noise = hyps[s].noise;
        c->mean = hyps[s].mean;
        c->model_kind = 0;
        c->fitted = true;
        rc = predict_into(c, (const double *)c->grid[c->grid_cur].p, c->M, (double *)c->mu.p, (double *)c->var.p);
        if (rc == B7_OK) rc = score_add(c, spec, fd);
      }
      c->zsc.p = zsc0, c->zss.p = zss0, c->w.p = w0, c->alpha.p = alpha0, c->Linv.p = linv0;
      c->fitted = false;  // the context's own fit slot does not hold any of these fits
      c->predicted = true;
      c->Mpred = c->M;
      B7_TRY(rc);
    }
  } else {
    for (int s = 0; s < S; ++s) {
      B7_TRY(fit_front(c, &hyps[s], (const double *)c->bhyp.p + (size_t)s * d));
      int *report = static_cast<int *>(c->pin_eval_dev) + 4 * s;  // a one-block factorisation mirrors its report itself
      FactorNote note;
      B7_TRY(launch_potrf(c, 0.0, true, report, &note));
      if (!c->linv_done) B7_TRY(launch_trtri(c));  // on a failed factor this inverts rubbish; the report discards it
      B7_TRY(launch_alpha(c, report, 4, note));  // + this fit's pivot report, no copy launch
      c->fitted = true;
      B7_TRY(predict_into(c, (const double *)c->grid[c->grid_cur].p, c->M, (double *)c->mu.p, (double *)c->var.p));
      c->predicted = true;
      c->Mpred = c->M;
      B7_TRY(score_add(c, spec, fd));
    }
    // this fit's lengthscales sit in the batch staging block, not where b7_gp_append / b7_gp_fantasize look for the
    // current fit's: the slot is declared empty, as the header says
    c->fitted = false;
  }
  return B7_OK;
}

// after the stream has drained: did every fit of the last eval_enqueue factor at the first attempt, without a hand-off
// time-out?
bool eval_reports_clean(b7_ctx *c, int S) {
  const int *reports = static_cast<const int *>(c->pin_eval);
  bool clean = true, aborted = false;
  for (int s = 0; s < S; ++s) {
    clean = clean && reports[4 * s] == 0 && reports[4 * s + 1] == 0;
    aborted = aborted || reports[4 * s + 1] != 0;
  }
  if (aborted) persist_gave_up(c);
  return clean;
}

// the same nomination through the per-sample path, jitter schedule (utils/math.lua:159-218) included; synchronous
int eval_redo(b7_ctx *c, int S, const b7_hyp *hyps, const b7_score_spec *spec, double *jitter_out, int *info_out) {
  B7_HIP(c, hipSetDevice(c->device));
  B7_TRY(launch_fill(c, (double *)c->acc.p, c->M, 0.0));
  c->acc_fresh = false;
  double *fd = nullptr;
  for (int s = 0; s < S; ++s) {
    B7_TRY(fit_hyp_core(c, &hyps[s], nullptr, jitter_out ? jitter_out + s : nullptr, info_out ? info_out + s : nullptr, true));
    c->predicted = true;
    c->Mpred = c->M;
    if (spec->kind == B7_SCORE_EI) B7_TRY(stage_fmin(c, spec->fmin, &fd));
    B7_TRY(score_add(c, spec, fd));
  }
  return B7_OK;
}

extern "C" {

int b7_eval_nominate(b7_ctx *c, int S, const b7_hyp *hyps, const b7_score_spec *spec, int64_t global_row_offset,
                     double *best_val, int64_t *best_idx1, double *jitter_out, int *info_out) {
  if (!c) return B7_ERR_INVALID;
  if (c->group) return b7_fail(c, B7_ERR_STATE, "eval_nominate: this context belongs to a group (b7_group_eval_nominate)");
  const bool exchange = c->comm && c->comm_world > 1;
  const int world = c->comm ? c->comm_world : 1, rank = c->comm ? c->comm_rank : 0;
  if (jitter_out && S > 0) std::fill(jitter_out, jitter_out + S, 0.0);
  if (info_out && S > 0) std::fill(info_out, info_out + S, 0);
  int rc = eval_validate(c, S, hyps, spec);
  if (rc == B7_OK && global_row_offset < 0) rc = b7_fail(c, B7_ERR_INVALID, "eval_nominate: negative row offset");
  if (rc == B7_OK && c->M == 0 && !exchange) rc = b7_fail(c, B7_ERR_STATE, "eval_nominate: no candidate grid on this context");
  if (!exchange) {
    // the arg-max and the copy of its record are enqueued before the host has seen any report: one synchronisation
    B7_TRY(rc);
    B7_TRY(eval_enqueue(c, S, hyps, spec));
    B7_TRY(exch_local(c, (double)S, global_row_offset, rank, world, true, true));  // record mirrored into mapped host memory
    B7_TRY(exch_wait_mirror(c));
    if (!eval_reports_clean(c, S)) {
      B7_TRY(eval_redo(c, S, hyps, spec, jitter_out, info_out));
      B7_TRY(exch_local(c, (double)S, global_row_offset, rank, world, true, true));
      B7_TRY(exch_wait_mirror(c));
    }
    return exch_conclude(c, c->tab_host, world, best_val, best_idx1);
  }
  // with a communicator the collective comes after the report check (a rank that redoes its nomination must not
  // issue one collective too many), and a rank that fails locally still reaches it, with a failure record
  if (rc == B7_OK && c->M > 0) {
    rc = eval_enqueue(c, S, hyps, spec);
    if (rc == B7_OK) rc = hipStreamSynchronize(c->stream) == hipSuccess ? B7_OK : b7_fail(c, B7_ERR_HIP, "eval_nominate: stream failed");
    if (rc == B7_OK && !eval_reports_clean(c, S)) rc = eval_redo(c, S, hyps, spec, jitter_out, info_out);
  }
  if (rc == B7_OK) rc = exch_local(c, (double)S, global_row_offset, rank, world, true);
  const std::string own = c->err;
  if (rc != B7_OK) B7_TRY(exch_fail_record(c, rank, world, rc));
  B7_TRY(exch_allreduce(c));
  B7_TRY(exch_fetch(c, 0, world));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  if (rc != B7_OK) {
    exch_forget(c);
    c->err = own;
    return rc;
  }
  return exch_conclude(c, c->tab_host, world, best_val, best_idx1);
}

int b7_gp_fit(b7_ctx *c, const double *X, const double *Y, int N, int d, int ycols, const b7_hyp *hyp,
              double *nll_out, double *jitter_used, int *info_out) {
  if (!c) return B7_ERR_INVALID;
  if (!hyp || !hyp->lenscale_sq) return b7_fail(c, B7_ERR_INVALID, "gp_fit: NULL argument");
  B7_TRY(b7_gp_set_data(c, X, Y, N, d, ycols));
  return b7_gp_fit_hyp(c, hyp, nll_out, jitter_used, info_out);
}

}  // extern "C"

// A persistent factorisation timed out on a hand-off: count it and use the launch schedule from here on (three strikes
// switch this context over for good; the arithmetic is the same either way).
void persist_gave_up(b7_ctx *c) {
  c->persist_aborts += 1;
  if (c->potrf_sched == 3) c->potrf_sched_saved = 3;
  c->potrf_sched = 1;
}
static void persist_restore(b7_ctx *c) {
  if (c->potrf_sched_saved == 3 && c->persist_aborts < 3) c->potrf_sched = 3;
}

static int try_factor(b7_ctx *c, double extra, int *info, bool with_inverse, FactorNote *note) {
  int two[2] = {0, 0};
  for (int attempt = 0; attempt < 2; ++attempt) {
    B7_TRY(launch_potrf(c, extra, with_inverse, nullptr, note));  // factors (K + extra*I): eps goes on the ORIGINAL matrix, utils/math.lua:190
    B7_HIP(c, hipMemcpyAsync(two, c->info.p, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    B7_HIP(c, hipStreamSynchronize(c->stream));
    if (two[1] == 0) break;
    if (attempt == 1) return b7_fail(c, B7_ERR_HIP, "Cholesky: hand-off time-out (code %d) outside the persistent schedule", two[1]);
    persist_gave_up(c);
  }
  persist_restore(c);
  *info = two[0];
  return B7_OK;
}

// utils/math.lua:159-218 on c->K (N x N inside Npad x Npad): plain attempt, then the growing-jitter retries on
// the ORIGINAL matrix; leaves L and dinv on the device.
static int chol_with_jitter(b7_ctx *c, double *jitter_out, int *info_first_out, bool with_inverse, FactorNote *note) {
  int info = 0;
  B7_TRY(try_factor(c, 0.0, &info, with_inverse, note));
  *info_first_out = info;
  *jitter_out = 0.0;
  if (info != 0) B7_TRY(jitter_retries(c, jitter_out, with_inverse, note));
  return B7_OK;
}

// The retries of utils/math.lua:174-202 after a failed plain attempt.
static int jitter_retries(b7_ctx *c, double *jitter_out, bool with_inverse, FactorNote *note) {
  const int N = c->N;
  int info = 1;
  double jitter = 0.0;
  {
    // max_eps = src:norm() (Frobenius) of the N x N matrix that was handed to chol (:174), reduced on the device in
    // a fixed order (it only gates the chol(I) fallback; copying K to the host cost 32 MiB of PCIe at N = 2048)
    double *fro_dev = reinterpret_cast<double *>(reinterpret_cast<char *>(c->info.p) + 16);
    B7_TRY(launch_fro_norm_sq(c, (const double *)c->K.p, N, c->Npad, fro_dev));
    double fro = 0.0;
    B7_HIP(c, hipMemcpyAsync(&fro, fro_dev, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    B7_HIP(c, hipStreamSynchronize(c->stream));
    const double max_eps = sqrt(fro);
    if (max_eps != max_eps)  // the reference's while-loop never ends here (eps > NaN is false); fail instead
      return b7_fail(c, B7_ERR_INVALID, "chol: the matrix contains NaN (check X_obs and the hyper-parameters)");
    double eps = c->opts.jitter_eps;
    for (;;) {
      if (eps > max_eps) {  // :184-186 chol(I)
        jitter = -1.0;
        B7_TRY(launch_set_identity(c));  // L = I, dinv = identity blocks
        c->linv_done = false;  // launch_trtri rebuilds inv(L) from this L and dinv
        if (note) *note = FactorNote();  // ... and launch_alpha computes alpha from that inverse
        break;
      }
      eps = eps * c->opts.jitter_growth;  // :188
      B7_TRY(try_factor(c, eps, &info, with_inverse, note));
      if (info == 0) {
        jitter = eps;
        break;
      }
    }
  }
  *jitter_out = jitter;
  return B7_OK;
}


extern "C" {

static int predict_into(b7_ctx *c, const double *xq, int64_t M, double *mu, double *var) {
  if (M == 0) return B7_OK;
  if (c->kpost_small && c->model_kind == 0 && kpost_small_applies(c))  // small fits: one kernel, K* never stored
    return launch_kpost_small(c, 1, xq, M, (const double *)c->w.p, (const double *)c->zsc.p, (const double *)c->zss.p,
                              (const double *)c->Linv.p, (const double *)c->alpha.p, nullptr, c->amp, c->noise, c->mean, mu, var, M);
  const size_t row_bytes = sizeof(double) * (size_t)c->Npad;
  int64_t chunk = (int64_t)(c->ks_bytes / row_bytes) / B7_MROWS * B7_MROWS;
  if (chunk < B7_MROWS) chunk = B7_MROWS;
  const int64_t Mpad = round_up(M, B7_MROWS);
  if (chunk > Mpad) chunk = Mpad;
  B7_TRY(b7_ensure(c, c->ks, (size_t)chunk * row_bytes));
  for (int64_t row0 = 0; row0 < M; row0 += chunk) {
    int64_t rows = Mpad - row0 < chunk ? Mpad - row0 : chunk;
    B7_TRY(launch_ksx(c, xq, row0, rows, M, c->dfit, (double *)c->ks.p, mu, c->ycols));
    if (c->ycols > 1) B7_TRY(launch_mean_multi(c, (const double *)c->ks.p, row0, rows, M, mu));
    B7_TRY(launch_post(c, (const double *)c->ks.p, row0, rows, M, var));
  }
  return B7_OK;
}

int b7_chol(b7_ctx *c, const double *src, int n, double *res, double *jitter_used, int *info_out) {
  if (!c) return B7_ERR_INVALID;
  if (!src || !res || n < 1) return b7_fail(c, B7_ERR_INVALID, "chol: bad arguments");
  B7_HIP(c, hipSetDevice(c->device));
  c->fitted = false;
  c->have_data = false;
  c->predicted = false;
  c->N = n;
  c->Npad = npad_of(c, n);
  const size_t np = (size_t)c->Npad, nn = np * np * sizeof(double);
  B7_TRY(b7_ensure(c, c->K, nn));
  B7_TRY(b7_ensure(c, c->L, nn));
  B7_TRY(b7_ensure(c, c->dinv, sizeof(double) * (np + B7_PANEL) * B7_PANEL));
  B7_TRY(b7_ensure(c, c->info, B7_INFO_BYTES));
  std::vector<double> Kp(np * np, 0.0);
  for (size_t i = 0; i < np; ++i) {
    if (i < (size_t)n)
      memcpy(&Kp[i * np], src + i * (size_t)n, sizeof(double) * n);
    else
      Kp[i * np + i] = 1.0;
  }
  B7_HIP(c, hipMemcpy(c->K.p, Kp.data(), nn, hipMemcpyHostToDevice));
  int info_first = 0;
  double jitter = 0.0;
  B7_TRY(chol_with_jitter(c, &jitter, &info_first, false));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  B7_HIP(c, hipMemcpy2D(res, sizeof(double) * n, c->L.p, sizeof(double) * np, sizeof(double) * n, n,
                        hipMemcpyDeviceToHost));
  if (jitter_used) *jitter_used = jitter;
  if (info_out) *info_out = info_first;
  return B7_OK;
}

int b7_gp_predict(b7_ctx *c, double *mean_host, double *var_host) {
  if (!c) return B7_ERR_INVALID;
  if (!c->fitted || c->model_kind != 0) return b7_fail(c, B7_ERR_STATE, "gp_predict: no GP fit on this context");
  if (c->M <= 0) return b7_fail(c, B7_ERR_STATE, "gp_predict: no candidate grid on this context");
  if (c->d != c->dfit) return b7_fail(c, B7_ERR_INVALID, "gp_predict: grid dims %d != fit dims %d", c->d, c->dfit);
  B7_HIP(c, hipSetDevice(c->device));
  B7_TRY(b7_ensure(c, c->mu, sizeof(double) * (size_t)c->M * c->ycols));
  B7_TRY(b7_ensure(c, c->var, sizeof(double) * (size_t)c->M));
  B7_TRY(predict_into(c, cur_grid(c), c->M, (double *)c->mu.p, (double *)c->var.p));
  c->predicted = true;
  c->Mpred = c->M;
  if (mean_host)
    B7_HIP(c, hipMemcpyAsync(mean_host, c->mu.p, sizeof(double) * (size_t)c->M * c->ycols, hipMemcpyDeviceToHost,
                             c->stream));
  if (var_host)
    B7_HIP(c, hipMemcpyAsync(var_host, c->var.p, sizeof(double) * (size_t)c->M, hipMemcpyDeviceToHost, c->stream));
  if (mean_host || var_host) B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

int b7_gp_predict_at(b7_ctx *c, const double *X1, int64_t M1, double *mean_host, double *var_host) {
  if (!c) return B7_ERR_INVALID;
  if (!c->fitted || c->model_kind != 0) return b7_fail(c, B7_ERR_STATE, "gp_predict_at: no GP fit on this context");
  if (M1 < 0 || (!X1 && M1 > 0)) return b7_fail(c, B7_ERR_INVALID, "gp_predict_at: bad X1/M1");
  if (M1 == 0) return B7_OK;
  B7_HIP(c, hipSetDevice(c->device));
  B7_TRY(b7_ensure(c, c->tmpgrid, sizeof(double) * (size_t)M1 * c->dfit));
  B7_TRY(b7_ensure(c, c->tmpmu, sizeof(double) * (size_t)M1 * c->ycols));
  B7_TRY(b7_ensure(c, c->tmpvar, sizeof(double) * (size_t)M1));
  B7_HIP(c, hipMemcpyAsync(c->tmpgrid.p, X1, sizeof(double) * (size_t)M1 * c->dfit, hipMemcpyHostToDevice, c->stream));
  B7_TRY(predict_into(c, (const double *)c->tmpgrid.p, M1, (double *)c->tmpmu.p, (double *)c->tmpvar.p));
  if (mean_host)
    B7_HIP(c, hipMemcpyAsync(mean_host, c->tmpmu.p, sizeof(double) * (size_t)M1 * c->ycols, hipMemcpyDeviceToHost,
                             c->stream));
  if (var_host)
    B7_HIP(c, hipMemcpyAsync(var_host, c->tmpvar.p, sizeof(double) * (size_t)M1, hipMemcpyDeviceToHost, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

int b7_gp_fantasize(b7_ctx *c, const double *X_pend, int P, int n, uint64_t seed, double *Y_out, double *mean_out,
                    double *cov_out) {
  if (!c) return B7_ERR_INVALID;
  if (!c->fitted || c->model_kind != 0) return b7_fail(c, B7_ERR_STATE, "gp_fantasize: no GP fit on this context");
  if (c->ycols != 1) return b7_fail(c, B7_ERR_UNSUPPORTED, "gp_fantasize: the fit must have one response column");
  if (!X_pend || P < 1 || n < 1 || !Y_out) return b7_fail(c, B7_ERR_INVALID, "gp_fantasize: bad arguments");
  if (P > 64) return b7_fail(c, B7_ERR_UNSUPPORTED, "gp_fantasize: %d pending points > 64", P);
  B7_HIP(c, hipSetDevice(c->device));
  const int np = c->Npad, d = c->dfit, dp = c->dpad;
  // workspace carve-up (doubles): xp 64*d | zsc_p 64*dp | zsh_p 64 | kp 64*np | vt 64*np | g 4096 | kpp 4096 |
  // S 4096 | dinv 4096 | mu 64 | out P*n | info
  size_t need = (size_t)64 * d + (size_t)64 * dp + 64 + (size_t)2 * 64 * np + 4 * 4096 + 64 + (size_t)P * n + 16;
  B7_TRY(b7_ensure(c, c->fant, need * sizeof(double)));
  double *xp = (double *)c->fant.p, *zscp = xp + (size_t)64 * d, *zshp = zscp + (size_t)64 * dp, *kp = zshp + 64;
  double *vt = kp + (size_t)64 * np, *g = vt + (size_t)64 * np, *kpp = g + 4096, *S = kpp + 4096, *dv = S + 4096;
  double *mu = dv + 4096, *out = mu + 64;
  int *info_dev = (int *)(out + (size_t)P * n + 2);
  B7_HIP(c, hipMemcpyAsync(xp, X_pend, sizeof(double) * (size_t)P * d, hipMemcpyHostToDevice, c->stream));
  // K(Xp, X) with the fused mean, then V' = K(Xp,X) L^-T and G = V'V
  B7_TRY(launch_ksx(c, xp, 0, 64, P, d, kp, mu, 1));
  B7_TRY(launch_gemm_nt(c, kp, np, (const double *)c->Linv.p, np, vt, np, 64, np, np));
  B7_TRY(launch_gemm_nt(c, vt, np, vt, np, g, 64, 64, 64, np));
  // K(Xp, Xp): the pending points as their own observation set, scaled with the fit's lengthscales
  double *ls_dev = (double *)c->scratch.p;  // still holds the lengthscales of the current fit
  B7_TRY(launch_prep_obs_aux(c, xp, ls_dev, P, 64, zscp, zshp));
  const ObsSet op{zscp, zshp, 64};
  B7_TRY(launch_k_generic(c, xp, 64, P, op, kpp));
  B7_TRY(launch_fantasy_cov(c, kpp, g, S, P, c->opts.var_with_noise ? c->noise : 0.0));
  if (cov_out) {
    B7_HIP(c, hipStreamSynchronize(c->stream));
    B7_HIP(c, hipMemcpy2D(cov_out, sizeof(double) * P, S, sizeof(double) * 64, sizeof(double) * P, P,
                          hipMemcpyDeviceToHost));
  }
  // factor with the same jitter schedule as utils.math.chol (eps on the ORIGINAL matrix: keep a copy in kpp)
  B7_HIP(c, hipMemcpyAsync(kpp, S, sizeof(double) * 4096, hipMemcpyDeviceToDevice, c->stream));
  double eps = c->opts.jitter_eps;
  int info = 0;
  for (int attempt = 0;; ++attempt) {
    B7_TRY(launch_fantasy_factor(c, S, dv, info_dev));
    B7_HIP(c, hipMemcpyAsync(&info, info_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    B7_HIP(c, hipStreamSynchronize(c->stream));
    if (info == 0) break;
    if (attempt > 4000) return b7_fail(c, B7_ERR_INVALID, "gp_fantasize: posterior covariance cannot be factored");
    eps = eps * c->opts.jitter_growth;
    B7_HIP(c, hipMemcpyAsync(S, kpp, sizeof(double) * 4096, hipMemcpyDeviceToDevice, c->stream));
    B7_TRY(launch_add_diag(c, S, 64, P, eps));
  }
  B7_TRY(launch_fantasy_sample(c, S, mu, P, n, seed, out));
  B7_HIP(c, hipMemcpyAsync(Y_out, out, sizeof(double) * (size_t)P * n, hipMemcpyDeviceToHost, c->stream));
  if (mean_out) B7_HIP(c, hipMemcpyAsync(mean_out, mu, sizeof(double) * P, hipMemcpyDeviceToHost, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

int b7_gp_append(b7_ctx *c, const double *x_new, const double *y_new) {
  if (!c) return B7_ERR_INVALID;
  if (!c->fitted || c->model_kind != 0) return b7_fail(c, B7_ERR_STATE, "gp_append: no GP fit on this context");
  if (!x_new || !y_new) return b7_fail(c, B7_ERR_INVALID, "gp_append: NULL argument");
  if (c->N + 1 > c->Npad)
    return b7_fail(c, B7_ERR_STATE, "gp_append: the padded factor is full (N = %d); refit with b7_gp_fit", c->N);
  B7_HIP(c, hipSetDevice(c->device));
  const int N = c->N, np = c->Npad, d = c->dfit, yc = c->ycols;
  c->predicted = false;
  // the new observation joins the observation set (raw row, scaled row, half norm, residual row)
  B7_HIP(c, hipMemcpyAsync((double *)c->xobs.p + (size_t)N * d, x_new, sizeof(double) * d, hipMemcpyHostToDevice,
                           c->stream));
  std::vector<double> r(yc);
  for (int k = 0; k < yc; ++k) r[k] = y_new[k] - c->mean;
  B7_HIP(c, hipMemcpyAsync((double *)c->resid.p + (size_t)N * yc, r.data(), sizeof(double) * yc, hipMemcpyHostToDevice,
                           c->stream));
  if (c->have_data)  // the resident data set grows with the fit, so a later b7_gp_fit_hyp sees the new row too
    B7_HIP(c, hipMemcpyAsync((double *)c->ybuf.p + (size_t)N * yc, y_new, sizeof(double) * yc, hipMemcpyHostToDevice,
                             c->stream));
  double *ls_dev = (double *)c->scratch.p;  // lengthscales of the current fit
  B7_TRY(launch_prep_obs(c, (const double *)c->xobs.p, ls_dev, N + 1, d));
  // k = K(x_new, [X; x_new]) through the covariance kernel (row 0 of a 64-row launch)
  const size_t nslices = ((size_t)np + 255) / 256;  // slice partials of launch_append_vectors: nslices x np
  B7_TRY(b7_ensure(c, c->fant, sizeof(double) * ((size_t)64 * np + 4 * (size_t)np + (size_t)np * nslices + 64)));
  double *krows = (double *)c->fant.p, *lvec = krows + (size_t)64 * np, *uvec = lvec + np, *evec = uvec + np;
  double *part = evec + np;
  int *status_dev = (int *)(part + (size_t)np * nslices);
  B7_TRY(launch_ksx(c, (const double *)c->xobs.p + (size_t)N * d, 0, 64, 1, d, krows, nullptr, 1));
  B7_TRY(launch_append_vectors(c, krows, lvec, uvec, part, evec));
  B7_TRY(launch_append_finalize(c, krows, lvec, uvec, evec, status_dev));
  int status = 0;
  B7_HIP(c, hipMemcpyAsync(&status, status_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  if (status != 0) {
    // roll the observation set back; the caller refits from scratch (jitter schedule included)
    B7_TRY(launch_prep_obs(c, (const double *)c->xobs.p, ls_dev, N, d));
    B7_HIP(c, hipStreamSynchronize(c->stream));
    return b7_fail(c, B7_ERR_STATE, "gp_append: the extended matrix is not positive definite; refit with b7_gp_fit");
  }
  c->N = N + 1;
  B7_TRY(launch_alpha(c));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

int b7_gp_download(b7_ctx *c, double *L_host, double *alpha_host, double *Linv_host) {
  if (!c) return B7_ERR_INVALID;
  if (!c->fitted) return b7_fail(c, B7_ERR_STATE, "gp_download: no fit on this context");
  const size_t N = c->N, np = c->Npad;
  B7_HIP(c, hipStreamSynchronize(c->stream));
  if (L_host)
    B7_HIP(c, hipMemcpy2D(L_host, sizeof(double) * N, c->L.p, sizeof(double) * np, sizeof(double) * N, N,
                          hipMemcpyDeviceToHost));
  if (Linv_host)
    B7_HIP(c, hipMemcpy2D(Linv_host, sizeof(double) * N, c->Linv.p, sizeof(double) * np, sizeof(double) * N, N,
                          hipMemcpyDeviceToHost));
  if (alpha_host)
    B7_HIP(c, hipMemcpy2D(alpha_host, sizeof(double) * c->ycols, c->alpha.p, sizeof(double) * c->yld,
                          sizeof(double) * c->ycols, N, hipMemcpyDeviceToHost));
  return B7_OK;
}

// ---- DNGO: basis features + Bayesian linear head --------------------------------------------------------------
static int upload_net(b7_ctx *c, const b7_mlp *net, int *z_out) {
  if (!net || !net->dims || !net->W || !net->b || net->n_layers < 1 || net->n_layers > 4)
    return b7_fail(c, B7_ERR_INVALID, "mlp: 1..4 layers with dims, W and b required");
  size_t total = 0;
  for (int l = 0; l < net->n_layers; ++l) total += (size_t)net->dims[l + 1] * net->dims[l] + net->dims[l + 1];
  std::vector<double> pack(total);
  size_t off = 0;
  for (int l = 0; l < net->n_layers; ++l) {
    const size_t nw = (size_t)net->dims[l + 1] * net->dims[l];
    memcpy(&pack[off], net->W[l], nw * sizeof(double));
    off += nw;
    memcpy(&pack[off], net->b[l], (size_t)net->dims[l + 1] * sizeof(double));
    off += net->dims[l + 1];
  }
  *z_out = net->dims[net->n_layers];
  // the same network as last time (models/dngo.lua:155-171 runs it over X_obs and then over the candidates): already there
  if (c->net_host.size() == total && c->netbuf.cap >= total * sizeof(double) &&
      memcmp(c->net_host.data(), pack.data(), total * sizeof(double)) == 0)
    return B7_OK;
  B7_TRY(b7_ensure(c, c->netbuf, total * sizeof(double)));
  B7_HIP(c, hipStreamSynchronize(c->stream));  // nobody is reading the old weights any more
  B7_HIP(c, hipMemcpy(c->netbuf.p, pack.data(), total * sizeof(double), hipMemcpyHostToDevice));
  c->net_host.swap(pack);
  return B7_OK;
}

// The feature matrix: round_up(M, 256) rows x zpad columns, of which the basis kernels write the first z of the first M
// rows.  Everything else must be zero (the variance GEMM reads whole padded rows) and is zeroed when the buffer is
// (re)allocated or its column layout changes -- not on every call: a 67 MB memset per nomination was a sixth of a cfg5 step.
static int feat_alloc(b7_ctx *c, int64_t M, int z) {
  const int zpad = npad_of(c, z);
  const size_t bytes = sizeof(double) * (size_t)round_up(M, B7_MROWS) * zpad;
  const bool grown = c->feat.cap < bytes;
  B7_TRY(b7_ensure(c, c->feat, bytes));
  if (grown || c->feat_z != z || c->feat_zeroed_bytes < bytes) {
    B7_HIP(c, hipMemsetAsync(c->feat.p, 0, c->feat.cap, c->stream));
    c->feat_zeroed_bytes = c->feat.cap;
    c->feat_z = z;
  }
  c->Mfeat = M;
  c->zdim = z;
  c->predicted = false;
  return B7_OK;
}

int b7_blr_basis(b7_ctx *c, const b7_mlp *net, const double *X, int64_t M, double *Z_host) {
  if (!c) return B7_ERR_INVALID;
  B7_HIP(c, hipSetDevice(c->device));
  int z = 0;
  B7_TRY(upload_net(c, net, &z));
  if (z > 256) return b7_fail(c, B7_ERR_UNSUPPORTED, "blr: basis width %d > 256", z);
  const int zpad = npad_of(c, z);
  if (!X) {  // resident grid -> resident features
    if (c->M <= 0 || c->d <= 0) return b7_fail(c, B7_ERR_STATE, "blr_basis: no candidate grid on this context");
    B7_TRY(feat_alloc(c, c->M, z));
    B7_TRY(launch_mlp_forward(c, (const double *)c->grid[c->grid_cur].p, c->M, c->d, (const double *)c->netbuf.p,
                              net->dims, net->n_layers, net->activation, (double *)c->feat.p, zpad));
    if (Z_host)
      B7_HIP(c, hipMemcpy2DAsync(Z_host, sizeof(double) * z, c->feat.p, sizeof(double) * zpad, sizeof(double) * z,
                                 c->M, hipMemcpyDeviceToHost, c->stream));
    B7_HIP(c, hipStreamSynchronize(c->stream));
    return B7_OK;
  }
  if (M < 1 || !Z_host) return b7_fail(c, B7_ERR_INVALID, "blr_basis: M >= 1 and Z_host required with X");
  const int d = net->dims[0];
  B7_TRY(b7_ensure(c, c->tmpgrid, sizeof(double) * (size_t)M * d));
  B7_TRY(b7_ensure(c, c->tmpmu, sizeof(double) * (size_t)M * z));
  B7_HIP(c, hipMemcpyAsync(c->tmpgrid.p, X, sizeof(double) * (size_t)M * d, hipMemcpyHostToDevice, c->stream));
  B7_TRY(launch_mlp_forward(c, (const double *)c->tmpgrid.p, M, d, (const double *)c->netbuf.p, net->dims,
                            net->n_layers, net->activation, (double *)c->tmpmu.p, z));
  B7_HIP(c, hipMemcpyAsync(Z_host, c->tmpmu.p, sizeof(double) * (size_t)M * z, hipMemcpyDeviceToHost, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

int b7_blr_features(b7_ctx *c, const double *Z1, int64_t M, int z) {
  if (!c) return B7_ERR_INVALID;
  if (!Z1 || M < 1 || z < 1 || z > 256) return b7_fail(c, B7_ERR_INVALID, "blr_features: bad arguments");
  B7_HIP(c, hipSetDevice(c->device));
  const int zpad = npad_of(c, z);
  B7_TRY(feat_alloc(c, M, z));
  B7_HIP(c, hipMemcpy2DAsync(c->feat.p, sizeof(double) * zpad, Z1, sizeof(double) * z, sizeof(double) * z, M,
                             hipMemcpyHostToDevice, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  if (c->M != M) {  // features stand in for a grid: the score calls size themselves by M
    c->M = M;
    c->d = 0;
    c->acc_valid = false;
  }
  return B7_OK;
}

// Shared tail of the two fit entry points: c->tmpgrid holds Z0' (zpad x nk, zero padded) on the device.
static int blr_fit_core(b7_ctx *c, const double *Y0, int N, int z, double alpha_prec, double beta, double mean,
                        double *nll_out) {
  const int zpad = npad_of(c, z), nk = (int)round_up(N, 16);
  const size_t np = (size_t)zpad, nn = np * np * sizeof(double);
  c->have_data = false;
  c->N = z;
  c->Npad = zpad;
  c->ycols = 1;
  c->yld = 1;
  c->mean = mean;
  c->noise = 1.0 / beta;
  c->amp = 0.0;
  B7_TRY(b7_ensure(c, c->K, nn));
  B7_TRY(b7_ensure(c, c->L, nn));
  B7_TRY(b7_ensure(c, c->Linv, nn));
  B7_TRY(b7_ensure(c, c->W, nn));
  B7_TRY(b7_ensure(c, c->dinv, sizeof(double) * (np + B7_PANEL) * B7_PANEL));
  B7_TRY(b7_ensure(c, c->alpha, sizeof(double) * np));
  B7_TRY(b7_ensure(c, c->resid, sizeof(double) * np));
  B7_TRY(b7_ensure(c, c->tmpvar, sizeof(double) * (size_t)nk));
  B7_TRY(b7_ensure(c, c->info, B7_INFO_BYTES));
  std::vector<double> rb((size_t)nk, 0.0);
  for (int i = 0; i < N; ++i) rb[i] = beta * (Y0[i] - mean);
  B7_HIP(c, hipMemcpyAsync(c->tmpvar.p, rb.data(), sizeof(double) * nk, hipMemcpyHostToDevice, c->stream));
  // G = Z0'Z0 (MFMA), K = beta G + alpha I, q = Z0' beta r
  B7_TRY(launch_gemm_nt(c, (const double *)c->tmpgrid.p, nk, (const double *)c->tmpgrid.p, nk, (double *)c->W.p, zpad,
                        zpad, zpad, nk));
  B7_TRY(launch_blr_assemble(c, (const double *)c->W.p, (double *)c->K.p, z, zpad, alpha_prec, beta));
  B7_TRY(launch_gemv_rows(c, (const double *)c->tmpgrid.p, nk, (const double *)c->tmpvar.p, nk, 0.0, 0, zpad, zpad,
                          (double *)c->resid.p));
  int info_first = 0;
  double jitter = 0.0;
  FactorNote note;
  B7_TRY(chol_with_jitter(c, &jitter, &info_first, true, &note));
  B7_TRY(launch_trtri(c));
  B7_TRY(launch_alpha(c, nullptr, 0, note));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  if (nll_out) {
    // -log p(y | alpha, beta) = -[ z/2 log alpha + N/2 log beta - E(m) - 1/2 log|K| - N/2 log 2 pi ],
    // E(m) = beta/2 |r|^2 - 1/2 (beta q)' m     (Bishop 3.82 / 3.86)
    std::vector<double> diag(z), m(z), bq(z);
    B7_HIP(c, hipMemcpy2D(diag.data(), sizeof(double), c->L.p, sizeof(double) * (np + 1), sizeof(double), z,
                          hipMemcpyDeviceToHost));
    B7_HIP(c, hipMemcpy(m.data(), c->alpha.p, sizeof(double) * z, hipMemcpyDeviceToHost));
    B7_HIP(c, hipMemcpy(bq.data(), c->resid.p, sizeof(double) * z, hipMemcpyDeviceToHost));
    double logdet = 0.0, rr = 0.0, qm = 0.0;
    for (int k = 0; k < z; ++k) {
      logdet += 2.0 * log(diag[k]);
      qm += bq[k] * m[k];
    }
    for (int i = 0; i < N; ++i) rr += (Y0[i] - mean) * (Y0[i] - mean);
    const double Em = 0.5 * beta * rr - 0.5 * qm;
    *nll_out = -(0.5 * z * log(alpha_prec) + 0.5 * N * log(beta) - Em - 0.5 * logdet - 0.5 * N * log(2.0 * M_PI));
  }
  c->model_kind = 1;
  c->fitted = true;
  return B7_OK;
}

int b7_blr_fit(b7_ctx *c, const double *Z0, const double *Y0, int N, int z, double alpha_prec, double beta,
               double mean, double *nll_out) {
  if (!c) return B7_ERR_INVALID;
  if (!Z0 || !Y0 || N < 1 || z < 1 || z > 256) return b7_fail(c, B7_ERR_INVALID, "blr_fit: bad arguments");
  if (!(alpha_prec > 0.0) || !(beta > 0.0)) return b7_fail(c, B7_ERR_INVALID, "blr_fit: precisions must be > 0");
  B7_HIP(c, hipSetDevice(c->device));
  c->fitted = false;
  c->predicted = false;
  const int zpad = npad_of(c, z), nk = (int)round_up(N, 16);
  B7_TRY(b7_ensure(c, c->tmpgrid, sizeof(double) * (size_t)zpad * nk));
  std::vector<double> zt((size_t)zpad * nk, 0.0);  // Z0' (a layout change, no arithmetic)
  for (int i = 0; i < N; ++i)
    for (int k = 0; k < z; ++k) zt[(size_t)k * nk + i] = Z0[(size_t)i * z + k];
  B7_HIP(c, hipMemcpyAsync(c->tmpgrid.p, zt.data(), sizeof(double) * (size_t)zpad * nk, hipMemcpyHostToDevice,
                           c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return blr_fit_core(c, Y0, N, z, alpha_prec, beta, mean, nll_out);
}

int b7_blr_fit_x(b7_ctx *c, const b7_mlp *net, const double *X0, const double *Y0, int N, double alpha_prec,
                 double beta, double mean, double *nll_out) {
  if (!c) return B7_ERR_INVALID;
  if (!X0 || !Y0 || N < 1) return b7_fail(c, B7_ERR_INVALID, "blr_fit_x: bad arguments");
  if (!(alpha_prec > 0.0) || !(beta > 0.0)) return b7_fail(c, B7_ERR_INVALID, "blr_fit_x: precisions must be > 0");
  B7_HIP(c, hipSetDevice(c->device));
  c->fitted = false;
  c->predicted = false;
  int z = 0;
  B7_TRY(upload_net(c, net, &z));
  if (z > 256) return b7_fail(c, B7_ERR_UNSUPPORTED, "blr: basis width %d > 256", z);
  const int d = net->dims[0], zpad = npad_of(c, z), nk = (int)round_up(N, 16);
  B7_TRY(b7_ensure(c, c->tmpmu, sizeof(double) * ((size_t)N * d + (size_t)N * z)));
  B7_TRY(b7_ensure(c, c->tmpgrid, sizeof(double) * (size_t)zpad * nk));
  double *xdev = (double *)c->tmpmu.p, *zdev = xdev + (size_t)N * d;
  B7_HIP(c, hipMemcpyAsync(xdev, X0, sizeof(double) * (size_t)N * d, hipMemcpyHostToDevice, c->stream));
  B7_TRY(launch_mlp_forward(c, xdev, N, d, (const double *)c->netbuf.p, net->dims, net->n_layers, net->activation,
                            zdev, z));
  B7_TRY(launch_transpose_pad(c, zdev, N, z, z, (double *)c->tmpgrid.p, zpad, nk));
  return blr_fit_core(c, Y0, N, z, alpha_prec, beta, mean, nll_out);
}

int b7_blr_predict(b7_ctx *c, double *mean_host, double *var_host) {
  if (!c) return B7_ERR_INVALID;
  if (!c->fitted || c->model_kind != 1) return b7_fail(c, B7_ERR_STATE, "blr_predict: no Bayesian-linear fit");
  if (c->Mfeat <= 0)
    return b7_fail(c, B7_ERR_STATE, "blr_predict: no current features (call b7_blr_basis / b7_blr_features after the "
                                    "last change of the candidate grid)");
  if (c->zdim != c->N) return b7_fail(c, B7_ERR_INVALID, "blr_predict: feature width %d != fit width %d", c->zdim, c->N);
  if (c->Mfeat != c->M) return b7_fail(c, B7_ERR_STATE, "blr_predict: features are stale (the grid changed)");
  B7_HIP(c, hipSetDevice(c->device));
  const int64_t M = c->Mfeat;
  B7_TRY(b7_ensure(c, c->mu, sizeof(double) * (size_t)M));
  B7_TRY(b7_ensure(c, c->var, sizeof(double) * (size_t)M));
  const int64_t Mpad = round_up(M, B7_MROWS);
  B7_TRY(launch_gemv_rows(c, (const double *)c->feat.p, c->Npad, (const double *)c->alpha.p, c->Npad, c->mean, 0, M, M,
                          (double *)c->mu.p));
  B7_TRY(launch_post(c, (const double *)c->feat.p, 0, Mpad, M, (double *)c->var.p));
  c->predicted = true;
  c->Mpred = M;
  if (mean_host) B7_HIP(c, hipMemcpyAsync(mean_host, c->mu.p, sizeof(double) * (size_t)M, hipMemcpyDeviceToHost, c->stream));
  if (var_host) B7_HIP(c, hipMemcpyAsync(var_host, c->var.p, sizeof(double) * (size_t)M, hipMemcpyDeviceToHost, c->stream));
  if (mean_host || var_host) B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

// ---- models/dngo.lua:155-175 + bots/bayesopt.lua:65-66 + :96 as ONE call ----------------------------------------------
// The DNGO branch of bayesopt:eval scores once (no hyper marginalisation): features of the observations, the Bayesian
// linear head, features of every candidate, predictive mean / variance, the acquisition, score:max(1).  Through the
// separate entry points that is six calls and four host synchronisations around 0.2 ms of GPU work; here everything is
// enqueued back to back (the head's Cholesky reports its pivot status into pinned memory in stream order), the arg-max
// record follows, and the host waits once.  A failed pivot (rare: K = beta Z'Z + alpha I is positive definite by
// construction) redoes the fit through b7_blr_fit_x's jitter schedule and scores again.
static int blr_enqueue_fit(b7_ctx *c, const b7_mlp *net, const double *X0, const double *Y0, int N, int z, double alpha_prec,
                           double beta, double mean, std::vector<double> &rb) {
  const int d = net->dims[0], zpad = npad_of(c, z), nk = (int)round_up(N, 16);
  const size_t np = (size_t)zpad, nn = np * np * sizeof(double);
  B7_TRY(b7_ensure(c, c->tmpmu, sizeof(double) * ((size_t)N * d + (size_t)N * z)));
  B7_TRY(b7_ensure(c, c->tmpgrid, sizeof(double) * (size_t)zpad * nk));
  B7_TRY(b7_ensure(c, c->K, nn));
  B7_TRY(b7_ensure(c, c->L, nn));
  B7_TRY(b7_ensure(c, c->Linv, nn));
  B7_TRY(b7_ensure(c, c->W, nn));
  B7_TRY(b7_ensure(c, c->dinv, sizeof(double) * (np + B7_PANEL) * B7_PANEL));
  B7_TRY(b7_ensure(c, c->alpha, sizeof(double) * np));
  B7_TRY(b7_ensure(c, c->resid, sizeof(double) * np));
  B7_TRY(b7_ensure(c, c->tmpvar, sizeof(double) * (size_t)nk));
  B7_TRY(b7_ensure(c, c->info, B7_INFO_BYTES));
  c->have_data = false;
  c->fitted = false;
  c->predicted = false;
  c->N = z;
  c->Npad = zpad;
  c->ycols = 1;
  c->yld = 1;
  c->mean = mean;
  c->noise = 1.0 / beta;
  c->amp = 0.0;
  c->model_kind = 1;
  // the observations and beta (y - mean) go up in ONE copy from pinned staging, laid out [X0 | beta (y - mean) | features]:
  // the caller's arrays are pageable, and an "asynchronous" copy from pageable memory makes the host wait for the stream to
  // reach it -- two of them per nomination serialised the host's enqueueing with the GPU's work
  (void)rb;
  const size_t up_doubles = (size_t)N * d + (size_t)nk;
  B7_TRY(b7_ensure(c, c->tmpmu, sizeof(double) * (up_doubles + (size_t)N * z)));
  if (c->pin_blr_bytes < sizeof(double) * up_doubles) {
    B7_HIP(c, hipStreamSynchronize(c->stream));
    if (c->pin_blr) (void)hipHostFree(c->pin_blr);
    c->pin_blr = nullptr;
    c->pin_blr_bytes = 0;
    B7_HIP(c, hipHostMalloc(&c->pin_blr, 2 * sizeof(double) * up_doubles, hipHostMallocDefault));
    c->pin_blr_bytes = 2 * sizeof(double) * up_doubles;
  }
  double *stage = static_cast<double *>(c->pin_blr);
  memcpy(stage, X0, sizeof(double) * (size_t)N * d);
  for (int i = 0; i < nk; ++i) stage[(size_t)N * d + i] = i < N ? beta * (Y0[i] - mean) : 0.0;
  double *xdev = (double *)c->tmpmu.p, *yvdev = xdev + (size_t)N * d, *zdev = yvdev + nk;
  B7_HIP(c, hipMemcpyAsync(xdev, stage, sizeof(double) * up_doubles, hipMemcpyHostToDevice, c->stream));
  B7_TRY(launch_mlp_forward(c, xdev, N, d, (const double *)c->netbuf.p, net->dims, net->n_layers, net->activation, zdev, z));
  if (zpad == 64 && c->blr_small) {
    // z <= 64 features: Z'Z, the assembly, b, the factorisation, its inverse and the head's weights in ONE workgroup of ONE
    // launch (blr_small.hip) instead of nine dispatches; a failed pivot is reported and redone through b7_blr_fit_x
    B7_TRY(launch_blr_head_small(c, zdev, N, z, z, yvdev, alpha_prec, beta, nullptr));
    B7_HIP(c, hipMemcpyAsync(c->pinned, c->info.p, 16, hipMemcpyDeviceToHost, c->stream));
    c->fitted = true;
    return B7_OK;
  }
  B7_TRY(launch_transpose_pad(c, zdev, N, z, z, (double *)c->tmpgrid.p, zpad, nk));
  B7_TRY(launch_gemm_nt(c, (const double *)c->tmpgrid.p, nk, (const double *)c->tmpgrid.p, nk, (double *)c->W.p, zpad, zpad,
                        zpad, nk));
  B7_TRY(launch_blr_assemble(c, (const double *)c->W.p, (double *)c->K.p, z, zpad, alpha_prec, beta));
  B7_TRY(launch_gemv_rows(c, (const double *)c->tmpgrid.p, nk, (const double *)yvdev, nk, 0.0, 0, zpad, zpad,
                          (double *)c->resid.p));
  FactorNote note;
  B7_TRY(launch_potrf(c, 0.0, true, nullptr, &note));
  if (!c->linv_done) B7_TRY(launch_trtri(c));  // on a failed factor this inverts rubbish; the report discards it
  B7_TRY(launch_alpha(c, nullptr, 0, note));
  // the pivot report of the plain attempt, into the pinned fit-report block in stream order
  B7_HIP(c, hipMemcpyAsync(c->pinned, c->info.p, 16, hipMemcpyDeviceToHost, c->stream));
  c->fitted = true;
  return B7_OK;
}

// features of the resident grid (recomputed, as models/dngo.lua:155-171 does on every predict), mean, variance, score
static int blr_enqueue_score(b7_ctx *c, const b7_mlp *net, int z, const b7_score_spec *spec) {
  const int zpad = npad_of(c, z);
  B7_TRY(feat_alloc(c, c->M, z));
  B7_TRY(b7_ensure(c, c->mu, sizeof(double) * (size_t)c->M));
  B7_TRY(b7_ensure(c, c->var, sizeof(double) * (size_t)c->M));
  B7_TRY(b7_ensure(c, c->acc, sizeof(double) * (size_t)c->M));
  bool mean_done = false;
  B7_TRY(launch_mlp_forward_mean(c, (const double *)c->grid[c->grid_cur].p, c->M, c->d, (const double *)c->netbuf.p, net->dims,
                                 net->n_layers, net->activation, (double *)c->feat.p, zpad, (const double *)c->alpha.p, c->mean,
                                 (double *)c->mu.p, &mean_done));
  if (!mean_done)
    B7_TRY(launch_gemv_rows(c, (const double *)c->feat.p, c->Npad, (const double *)c->alpha.p, c->Npad, c->mean, 0, c->M, c->M,
                            (double *)c->mu.p));
  B7_TRY(launch_post(c, (const double *)c->feat.p, 0, round_up(c->M, B7_MROWS), c->M, (double *)c->var.p));
  c->predicted = true;
  c->Mpred = c->M;
  // bots/bayesopt.lua:65-66: the score of the one model, no accumulation over samples -> written, not added
  double *fd = nullptr;
  if (spec->kind == B7_SCORE_EI) {
    B7_TRY(stage_fmin(c, spec->fmin, &fd));
    B7_TRY(launch_ei(c, (const double *)c->mu.p, (const double *)c->var.p, fd, spec->tradeoff, c->M, 1, (double *)c->acc.p, false));
  } else {
    B7_TRY(launch_cb(c, (const double *)c->mu.p, (const double *)c->var.p, spec->tradeoff, spec->upper, spec->sign, c->M, 1,
                     (double *)c->acc.p, false));
  }
  c->acc_valid = true;
  return B7_OK;
}

int b7_blr_eval_nominate(b7_ctx *c, const b7_mlp *net, const double *X0, const double *Y0, int N, double alpha_prec,
                         double beta, double mean, const b7_score_spec *spec, int64_t global_row_offset, double *best_val,
                         int64_t *best_idx1, double *jitter_used) {
  if (!c) return B7_ERR_INVALID;
  if (c->group) return b7_fail(c, B7_ERR_STATE, "blr_eval_nominate: this context belongs to a group");
  const bool exchange = c->comm && c->comm_world > 1;
  const int world = c->comm ? c->comm_world : 1, rank = c->comm ? c->comm_rank : 0;
  if (jitter_used) *jitter_used = 0.0;
  int rc = B7_OK, z = 0;
  if (!X0 || !Y0 || N < 1 || !spec) rc = b7_fail(c, B7_ERR_INVALID, "blr_eval_nominate: bad arguments");
  else if (!(alpha_prec > 0.0) || !(beta > 0.0)) rc = b7_fail(c, B7_ERR_INVALID, "blr_eval_nominate: precisions must be > 0");
  else if (spec->kind != B7_SCORE_EI && spec->kind != B7_SCORE_CB) rc = b7_fail(c, B7_ERR_INVALID, "blr_eval_nominate: unknown score kind %d", spec->kind);
  else if (spec->kind == B7_SCORE_EI && !spec->fmin) rc = b7_fail(c, B7_ERR_INVALID, "blr_eval_nominate: EI needs fmin");
  else if (global_row_offset < 0) rc = b7_fail(c, B7_ERR_INVALID, "blr_eval_nominate: negative row offset");
  else if (c->M == 0 && !exchange) rc = b7_fail(c, B7_ERR_STATE, "blr_eval_nominate: no candidate grid on this context");
  if (rc == B7_OK) rc = hipSetDevice(c->device) == hipSuccess ? B7_OK : b7_fail(c, B7_ERR_HIP, "hipSetDevice failed");
  if (rc == B7_OK) rc = upload_net(c, net, &z);
  if (rc == B7_OK && z > 256) rc = b7_fail(c, B7_ERR_UNSUPPORTED, "blr: basis width %d > 256", z);
  if (rc == B7_OK && c->M > 0 && net->dims[0] != c->d)
    rc = b7_fail(c, B7_ERR_INVALID, "blr_eval_nominate: network input width %d != grid dims %d", net->dims[0], c->d);
  std::vector<double> rb;  // beta (Y - mean): the source of an asynchronous copy, alive until the synchronisation below
  auto local = [&]() -> int {
    if (c->M == 0) return B7_OK;   // an empty shard: nothing to score, but the exchange is collective
    B7_TRY(blr_enqueue_fit(c, net, X0, Y0, N, z, alpha_prec, beta, mean, rb));
    return blr_enqueue_score(c, net, z, spec);
  };
  auto redo = [&]() -> int {        // the jitter schedule of utils/math.lua:159-218, through the synchronous fit
    B7_TRY(b7_blr_fit_x(c, net, X0, Y0, N, alpha_prec, beta, mean, nullptr));
    if (jitter_used) *jitter_used = -2.0;  // "a jitter was needed" (its size is the fit's business; see b7_blr_fit_x)
    return blr_enqueue_score(c, net, z, spec);
  };
  const int *report = static_cast<const int *>(c->pinned);
  if (!exchange) {
    B7_TRY(rc);
    B7_TRY(local());
    B7_TRY(exch_local(c, 1.0, global_row_offset, rank, world, true, true));
    B7_TRY(exch_wait_mirror(c));
    if (report[0] != 0 || report[1] != 0) {
      if (report[1] != 0) persist_gave_up(c);
      B7_TRY(redo());
      B7_TRY(exch_local(c, 1.0, global_row_offset, rank, world, true, true));
      B7_TRY(exch_wait_mirror(c));
    }
    return exch_conclude(c, c->tab_host, world, best_val, best_idx1);
  }
  if (rc == B7_OK) rc = local();
  if (rc == B7_OK) rc = hipStreamSynchronize(c->stream) == hipSuccess ? B7_OK : b7_fail(c, B7_ERR_HIP, "blr_eval_nominate: stream failed");
  if (rc == B7_OK && c->M > 0 && (report[0] != 0 || report[1] != 0)) {
    if (report[1] != 0) persist_gave_up(c);
    rc = redo();
  }
  if (rc == B7_OK) rc = exch_local(c, 1.0, global_row_offset, rank, world, true);
  const std::string own = c->err;
  if (rc != B7_OK) B7_TRY(exch_fail_record(c, rank, world, rc));
  B7_TRY(exch_allreduce(c));
  B7_TRY(exch_fetch(c, 0, world));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  if (rc != B7_OK) {
    exch_forget(c);
    c->err = own;
    return rc;
  }
  return exch_conclude(c, c->tab_host, world, best_val, best_idx1);
}

// ---- the same with the head's hypers marginalised (models/dngo.lua:109,174) -------------------------------------------------
// S heads over the same features.  z <= 64 features (the usual DNGO head): everything is enqueued without a host wait -- features
// of the observations, the S heads as S workgroups of ONE launch (blr_small.hip), features of the candidates, the S means (one
// batched matrix-vector launch), the S variances (one launch of the posterior kernel over the shared features), and score:add x S +
// div + arg-max + record fused in one launch.  Wider heads, or a failed pivot: head by head through b7_blr_fit_x.
static int blr_marg_slow(b7_ctx *c, const b7_mlp *net, const double *X0, const double *Y0, int N, int S, const double *ap,
                         const double *bt, const double *mn, int z, const b7_score_spec *spec, double *nll_out) {
  const int zpad = npad_of(c, z);
  c->pend.on = false;
  B7_TRY(b7_ensure(c, c->acc, sizeof(double) * (size_t)c->M));
  B7_TRY(launch_fill(c, (double *)c->acc.p, c->M, 0.0));
  c->acc_fresh = false;
  c->acc_valid = true;
  for (int s = 0; s < S; ++s) {
    B7_TRY(b7_blr_fit_x(c, net, X0, Y0, N, ap[s], bt[s], mn[s], nll_out ? nll_out + s : nullptr));   // synchronous, jitter schedule included
    if (s == 0) {
      B7_TRY(feat_alloc(c, c->M, z));
      B7_TRY(launch_mlp_forward(c, (const double *)c->grid[c->grid_cur].p, c->M, c->d, (const double *)c->netbuf.p, net->dims,
                                net->n_layers, net->activation, (double *)c->feat.p, zpad));
    }
    c->Mfeat = c->M;
    B7_TRY(b7_ensure(c, c->mu, sizeof(double) * (size_t)c->M));
    B7_TRY(b7_ensure(c, c->var, sizeof(double) * (size_t)c->M));
    B7_TRY(launch_gemv_rows(c, (const double *)c->feat.p, c->Npad, (const double *)c->alpha.p, c->Npad, c->mean, 0, c->M, c->M,
                            (double *)c->mu.p));
    B7_TRY(launch_post(c, (const double *)c->feat.p, 0, round_up(c->M, B7_MROWS), c->M, (double *)c->var.p));
    double *fd = nullptr;
    if (spec->kind == B7_SCORE_EI) {
      B7_TRY(stage_fmin(c, spec->fmin, &fd));
      B7_TRY(launch_ei(c, (const double *)c->mu.p, (const double *)c->var.p, fd, spec->tradeoff, c->M, 1, (double *)c->acc.p, true));
    } else {
      B7_TRY(launch_cb(c, (const double *)c->mu.p, (const double *)c->var.p, spec->tradeoff, spec->upper, spec->sign, c->M, 1,
                       (double *)c->acc.p, true));
    }
  }
  c->predicted = true;
  c->Mpred = c->M;
  return B7_OK;
}

static int blr_marg_fast(b7_ctx *c, const b7_mlp *net, const double *X0, const double *Y0, int N, int S, const double *ap,
                         const double *bt, const double *mn, int z, const b7_score_spec *spec, bool want_terms) {
  const int d = net->dims[0];
  const size_t up_doubles = (size_t)N * d + (size_t)N + 5 * (size_t)S;   // [X0 | y | S alpha | S beta | S mean | S zeros | S 1/beta]
  B7_TRY(b7_ensure(c, c->tmpmu, sizeof(double) * (up_doubles + (size_t)N * z)));
  B7_TRY(b7_ensure(c, c->bL, sizeof(double) * (size_t)S * 64 * 64));
  B7_TRY(b7_ensure(c, c->bLinv, sizeof(double) * (size_t)S * 64 * 64));
  B7_TRY(b7_ensure(c, c->balpha, sizeof(double) * (size_t)S * 64));
  B7_TRY(b7_ensure(c, c->bresid, sizeof(double) * (size_t)S * 64));
  B7_TRY(b7_ensure(c, c->binfo, sizeof(int) * 4 * (size_t)S));
  B7_TRY(b7_ensure(c, c->bterms, sizeof(double) * 3 * (size_t)S + 64));
  B7_TRY(b7_ensure(c, c->bmu, sizeof(double) * (size_t)S * c->M));
  B7_TRY(b7_ensure(c, c->bvar, sizeof(double) * (size_t)S * c->M));
  B7_TRY(b7_ensure(c, c->acc, sizeof(double) * (size_t)c->M));
  B7_TRY(b7_ensure(c, c->info, B7_INFO_BYTES));
  // pinned, device-mapped block for the reports and the evidence's terms: [S][4] ints | [S][3] doubles
  const size_t rep_bytes = 16 * (size_t)S, term_bytes = sizeof(double) * 3 * (size_t)S;
  if (c->pin_eval_bytes < rep_bytes + term_bytes) {
    B7_HIP(c, hipStreamSynchronize(c->stream));
    if (c->pin_eval) (void)hipHostFree(c->pin_eval);
    c->pin_eval = nullptr;
    c->pin_eval_bytes = 0;
    B7_HIP(c, hipHostMalloc(&c->pin_eval, 2 * (rep_bytes + term_bytes), hipHostMallocMapped));
    B7_HIP(c, hipHostGetDevicePointer(&c->pin_eval_dev, c->pin_eval, 0));
    c->pin_eval_bytes = 2 * (rep_bytes + term_bytes);
  }
  if (c->pin_blr_bytes < sizeof(double) * up_doubles) {
    B7_HIP(c, hipStreamSynchronize(c->stream));
    if (c->pin_blr) (void)hipHostFree(c->pin_blr);
    c->pin_blr = nullptr;
    c->pin_blr_bytes = 0;
    B7_HIP(c, hipHostMalloc(&c->pin_blr, 2 * sizeof(double) * up_doubles, hipHostMallocDefault));
    c->pin_blr_bytes = 2 * sizeof(double) * up_doubles;
  }
  c->have_data = false, c->fitted = false, c->predicted = false;
  c->N = z, c->Npad = 64, c->ycols = 1, c->yld = 1, c->model_kind = 1, c->amp = 0.0;
  double *stage = static_cast<double *>(c->pin_blr);
  memcpy(stage, X0, sizeof(double) * (size_t)N * d);
  memcpy(stage + (size_t)N * d, Y0, sizeof(double) * N);
  double *hs = stage + (size_t)N * d + N;
  for (int s = 0; s < S; ++s) hs[s] = ap[s], hs[S + s] = bt[s], hs[2 * S + s] = mn[s], hs[3 * S + s] = 0.0, hs[4 * S + s] = 1.0 / bt[s];
  double *xdev = (double *)c->tmpmu.p, *ydev = xdev + (size_t)N * d, *hdev = ydev + N, *zdev = hdev + 5 * (size_t)S;
  B7_HIP(c, hipMemcpyAsync(xdev, stage, sizeof(double) * up_doubles, hipMemcpyHostToDevice, c->stream));
  B7_TRY(launch_mlp_forward(c, xdev, N, d, (const double *)c->netbuf.p, net->dims, net->n_layers, net->activation, zdev, z));
  int *reports = static_cast<int *>(c->pin_eval);
  memset(reports, 0xff, rep_bytes);
  B7_TRY(launch_blr_heads_small(c, S, zdev, N, z, z, ydev, hdev, (double *)c->bL.p, (double *)c->bLinv.p, (double *)c->balpha.p,
                                (double *)c->bresid.p, (int *)c->binfo.p, static_cast<int *>(c->pin_eval_dev), (double *)c->bterms.p));
  if (want_terms)
    B7_HIP(c, hipMemcpyAsync(static_cast<char *>(c->pin_eval) + rep_bytes, c->bterms.p, term_bytes, hipMemcpyDeviceToHost, c->stream));
  B7_TRY(feat_alloc(c, c->M, z));
  B7_TRY(launch_mlp_forward(c, (const double *)c->grid[c->grid_cur].p, c->M, c->d, (const double *)c->netbuf.p, net->dims,
                            net->n_layers, net->activation, (double *)c->feat.p, 64));
  B7_TRY(launch_gemv_rows_batch(c, S, (const double *)c->feat.p, 64, (const double *)c->balpha.p, 64, 64, hdev + 2 * (size_t)S, c->M,
                                (double *)c->bmu.p, c->M));
  B7_TRY(launch_post_heads(c, S, (const double *)c->bLinv.p, (const double *)c->feat.p, round_up(c->M, B7_MROWS), c->M,
                           (double *)c->bvar.p, c->M, hdev + 3 * (size_t)S, hdev + 4 * (size_t)S));
  double *fd = nullptr;
  if (spec->kind == B7_SCORE_EI) B7_TRY(stage_fmin(c, spec->fmin, &fd));
  c->acc_fresh = true;
  c->acc_valid = true;
  c->pend.on = true;
  c->pend.kind = spec->kind, c->pend.S = S, c->pend.upper = spec->upper;
  c->pend.mu = (const double *)c->bmu.p, c->pend.var = (const double *)c->bvar.p, c->pend.fd = fd;
  c->pend.stride = c->M, c->pend.tradeoff = spec->tradeoff, c->pend.sign = spec->sign;
  return B7_OK;
}

int b7_blr_eval_nominate_marg(b7_ctx *c, const b7_mlp *net, const double *X0, const double *Y0, int N, int S, const double *ap,
                              const double *bt, const double *mn, const b7_score_spec *spec, int64_t global_row_offset,
                              double *best_val, int64_t *best_idx1, double *nll_out, double *jitter_used) {
  if (!c) return B7_ERR_INVALID;
  if (c->group) return b7_fail(c, B7_ERR_STATE, "blr_eval_nominate_marg: this context belongs to a group");
  const bool exchange = c->comm && c->comm_world > 1;
  const int world = c->comm ? c->comm_world : 1, rank = c->comm ? c->comm_rank : 0;
  if (jitter_used) *jitter_used = 0.0;
  int rc = B7_OK, z = 0;
  if (!X0 || !Y0 || N < 1 || S < 1 || !ap || !bt || !mn || !spec) rc = b7_fail(c, B7_ERR_INVALID, "blr_eval_nominate_marg: bad arguments");
  else if (spec->kind != B7_SCORE_EI && spec->kind != B7_SCORE_CB) rc = b7_fail(c, B7_ERR_INVALID, "blr_eval_nominate_marg: unknown score kind %d", spec->kind);
  else if (spec->kind == B7_SCORE_EI && !spec->fmin) rc = b7_fail(c, B7_ERR_INVALID, "blr_eval_nominate_marg: EI needs fmin");
  else if (global_row_offset < 0) rc = b7_fail(c, B7_ERR_INVALID, "blr_eval_nominate_marg: negative row offset");
  else if (c->M == 0 && !exchange) rc = b7_fail(c, B7_ERR_STATE, "blr_eval_nominate_marg: no candidate grid on this context");
  for (int s = 0; rc == B7_OK && s < S; ++s)
    if (!(ap[s] > 0.0) || !(bt[s] > 0.0)) rc = b7_fail(c, B7_ERR_INVALID, "blr_eval_nominate_marg: precisions must be > 0 (sample %d)", s);
  if (rc == B7_OK) rc = hipSetDevice(c->device) == hipSuccess ? B7_OK : b7_fail(c, B7_ERR_HIP, "hipSetDevice failed");
  if (rc == B7_OK) rc = upload_net(c, net, &z);
  if (rc == B7_OK && z > 256) rc = b7_fail(c, B7_ERR_UNSUPPORTED, "blr: basis width %d > 256", z);
  if (rc == B7_OK && c->M > 0 && net->dims[0] != c->d)
    rc = b7_fail(c, B7_ERR_INVALID, "blr_eval_nominate_marg: network input width %d != grid dims %d", net->dims[0], c->d);
  const bool fast = rc == B7_OK && z <= 64 && c->blr_small && c->npad_small;
  auto finish_terms = [&]() {   // the evidence of every head from the kernel's three sums (Bishop 3.82 / 3.86, as blr_fit_core)
    if (!nll_out) return;
    const double *t = reinterpret_cast<const double *>(static_cast<const char *>(c->pin_eval) + 16 * (size_t)S);
    for (int s = 0; s < S; ++s) {
      const double Em = 0.5 * bt[s] * t[3 * s + 2] - 0.5 * t[3 * s + 1];
      nll_out[s] = -(0.5 * z * log(ap[s]) + 0.5 * N * log(bt[s]) - Em - t[3 * s] - 0.5 * N * log(2.0 * M_PI));
    }
  };
  auto reports_clean = [&]() {
    const int *rep = static_cast<const int *>(c->pin_eval);
    for (int s = 0; s < S; ++s)
      if (rep[4 * s] != 0 || rep[4 * s + 1] != 0) return false;
    return true;
  };
  if (!exchange) {
    B7_TRY(rc);
    if (fast) {
      B7_TRY(blr_marg_fast(c, net, X0, Y0, N, S, ap, bt, mn, z, spec, nll_out != nullptr));
      B7_TRY(exch_local(c, (double)S, global_row_offset, rank, world, true, true));
      B7_TRY(exch_wait_mirror(c));
      if (reports_clean()) {
        finish_terms();
        c->fitted = false;   // the context's own fit slot holds none of the S heads
        return exch_conclude(c, c->tab_host, world, best_val, best_idx1);
      }
      if (jitter_used) *jitter_used = -2.0;
    }
    B7_TRY(blr_marg_slow(c, net, X0, Y0, N, S, ap, bt, mn, z, spec, nll_out));
    B7_TRY(exch_local(c, (double)S, global_row_offset, rank, world, true, true));
    B7_TRY(exch_wait_mirror(c));
    return exch_conclude(c, c->tab_host, world, best_val, best_idx1);
  }
  // with a communicator: the local part first (a rank that fails still reaches the collective with a failure record)
  if (rc == B7_OK && c->M > 0) {
    bool done = false;
    if (fast) {
      rc = blr_marg_fast(c, net, X0, Y0, N, S, ap, bt, mn, z, spec, nll_out != nullptr);
      if (rc == B7_OK) rc = score_flush_pending(c);
      if (rc == B7_OK) rc = hipStreamSynchronize(c->stream) == hipSuccess ? B7_OK : b7_fail(c, B7_ERR_HIP, "blr_eval_nominate_marg: stream failed");
      if (rc == B7_OK && reports_clean()) {
        finish_terms();
        done = true;
      } else if (rc == B7_OK && jitter_used) {
        *jitter_used = -2.0;
      }
    }
    if (rc == B7_OK && !done) rc = blr_marg_slow(c, net, X0, Y0, N, S, ap, bt, mn, z, spec, nll_out);
  }
  if (rc == B7_OK) rc = exch_local(c, (double)S, global_row_offset, rank, world, true);
  const std::string own = c->err;
  if (rc != B7_OK) B7_TRY(exch_fail_record(c, rank, world, rc));
  B7_TRY(exch_allreduce(c));
  B7_TRY(exch_fetch(c, 0, world));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  if (rc != B7_OK) {
    exch_forget(c);
    c->err = own;
    return rc;
  }
  return exch_conclude(c, c->tab_host, world, best_val, best_idx1);
}

// ---- scores --------------------------------------------------------------------------------------------
int b7_score_reset(b7_ctx *c) {
  if (!c) return B7_ERR_INVALID;
  if (c->M <= 0) return b7_fail(c, B7_ERR_STATE, "score_reset: no candidate grid");
  B7_HIP(c, hipSetDevice(c->device));
  B7_TRY(b7_ensure(c, c->acc, sizeof(double) * (size_t)c->M));
  B7_TRY(launch_fill(c, (double *)c->acc.p, c->M, 0.0));  // torch.zeros(X_hid:size(1)), bots/bayesopt.lua:69
  c->acc_fresh = false;
  c->acc_valid = true;
  return B7_OK;
}

static int score_ready(b7_ctx *c, const char *who) {
  if (!c->predicted || c->Mpred != c->M) return b7_fail(c, B7_ERR_STATE, "%s: call b7_gp_predict first", who);
  if (!c->acc_valid) return b7_fail(c, B7_ERR_STATE, "%s: call b7_score_reset first", who);
  return B7_OK;
}

int b7_score_ei(b7_ctx *c, const double *fmin, double tradeoff) {
  if (!c) return B7_ERR_INVALID;
  if (!fmin) return b7_fail(c, B7_ERR_INVALID, "score_ei: fmin is NULL");
  B7_TRY(score_ready(c, "score_ei"));
  B7_HIP(c, hipSetDevice(c->device));
  // fmin goes pageable -> pinned staging ([4096, 6144) of the pinned block) -> device: the caller's array need not
  // outlive this call and the copy is a true asynchronous one.  The staging slot may still be the source of an
  // earlier copy in flight, hence the wait when the values change (once per nomination: fmin is the same for every
  // hyper sample of a marginalisation loop).
  double *fd = nullptr;
  B7_TRY(stage_fmin(c, fmin, &fd));
  return launch_ei(c, (const double *)c->mu.p, (const double *)c->var.p, fd, tradeoff, c->M, c->ycols,
                   (double *)c->acc.p, true);
}

int b7_score_cb(b7_ctx *c, double tradeoff, int upper, double sign) {
  if (!c) return B7_ERR_INVALID;
  B7_TRY(score_ready(c, "score_cb"));
  B7_HIP(c, hipSetDevice(c->device));
  return launch_cb(c, (const double *)c->mu.p, (const double *)c->var.p, tradeoff, upper, sign, c->M, c->ycols,
                   (double *)c->acc.p, true);
}

int b7_score_finish(b7_ctx *c, double divisor, double *best_val, int64_t *best_idx1, double *scores_host) {
  if (!c) return B7_ERR_INVALID;
  if (!c->acc_valid) return b7_fail(c, B7_ERR_STATE, "score_finish: call b7_score_reset first");
  B7_HIP(c, hipSetDevice(c->device));
  B7_TRY(acc_materialize(c));
  B7_TRY(launch_finish(c, (double *)c->acc.p, c->M, divisor, best_val, best_idx1));
  if (scores_host) {
    B7_HIP(c, hipMemcpyAsync(scores_host, c->acc.p, sizeof(double) * (size_t)c->M, hipMemcpyDeviceToHost, c->stream));
    B7_HIP(c, hipStreamSynchronize(c->stream));
  }
  return B7_OK;
}

static int upload_mv(b7_ctx *c, const double *mean, const double *var, int64_t M, int cc) {
  B7_TRY(b7_ensure(c, c->tmpmu, sizeof(double) * (size_t)M * cc));
  B7_TRY(b7_ensure(c, c->tmpvar, sizeof(double) * (size_t)M));
  B7_TRY(b7_ensure(c, c->tmpgrid, sizeof(double) * (size_t)M));
  B7_HIP(c, hipMemcpyAsync(c->tmpmu.p, mean, sizeof(double) * (size_t)M * cc, hipMemcpyHostToDevice, c->stream));
  B7_HIP(c, hipMemcpyAsync(c->tmpvar.p, var, sizeof(double) * (size_t)M, hipMemcpyHostToDevice, c->stream));
  return B7_OK;
}

int b7_ei_compute(b7_ctx *c, const double *mean, const double *var, const double *fmin, double tradeoff, int64_t M,
                  int cc, double *out) {
  if (!c) return B7_ERR_INVALID;
  if (M < 0 || cc < 1 || cc > 256 || (M > 0 && (!mean || !var || !fmin || !out)))
    return b7_fail(c, B7_ERR_INVALID, "ei_compute: bad arguments");
  if (M == 0) return B7_OK;
  B7_HIP(c, hipSetDevice(c->device));
  B7_TRY(upload_mv(c, mean, var, M, cc));
  double *fd = (double *)((char *)c->scratch.p + 2048);
  B7_HIP(c, hipMemcpyAsync(fd, fmin, sizeof(double) * cc, hipMemcpyHostToDevice, c->stream));
  B7_TRY(launch_ei(c, (const double *)c->tmpmu.p, (const double *)c->tmpvar.p, fd, tradeoff, M, cc,
                   (double *)c->tmpgrid.p, false));
  B7_HIP(c, hipMemcpyAsync(out, c->tmpgrid.p, sizeof(double) * (size_t)M, hipMemcpyDeviceToHost, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

int b7_cb_compute(b7_ctx *c, const double *mean, const double *var, double tradeoff, int upper, double sign, int64_t M,
                  int cc, double *out) {
  if (!c) return B7_ERR_INVALID;
  if (M < 0 || cc < 1 || (M > 0 && (!mean || !var || !out))) return b7_fail(c, B7_ERR_INVALID, "cb_compute: bad arguments");
  if (M == 0) return B7_OK;
  B7_HIP(c, hipSetDevice(c->device));
  B7_TRY(upload_mv(c, mean, var, M, cc));
  B7_TRY(launch_cb(c, (const double *)c->tmpmu.p, (const double *)c->tmpvar.p, tradeoff, upper, sign, M, cc,
                   (double *)c->tmpgrid.p, false));
  B7_HIP(c, hipMemcpyAsync(out, c->tmpgrid.p, sizeof(double) * (size_t)M, hipMemcpyDeviceToHost, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

int b7_argmax(b7_ctx *c, const double *scores, int64_t M, double *best_val, int64_t *best_idx1) {
  if (!c) return B7_ERR_INVALID;
  if (M < 1 || !scores) return b7_fail(c, B7_ERR_INVALID, "argmax: empty input");
  B7_HIP(c, hipSetDevice(c->device));
  B7_TRY(b7_ensure(c, c->tmpgrid, sizeof(double) * (size_t)M));
  B7_HIP(c, hipMemcpyAsync(c->tmpgrid.p, scores, sizeof(double) * (size_t)M, hipMemcpyHostToDevice, c->stream));
  return launch_finish(c, (double *)c->tmpgrid.p, M, 1.0, best_val, best_idx1);
}

// ---- measurement ---------------------------------------------------------------------------------------
static int timers_init(b7_ctx *c) {
  if (c->tev_init) return B7_OK;
  for (int i = 0; i < B7_MAX_TIMERS; ++i) {
    B7_HIP(c, hipEventCreate(&c->tev[i][0]));
    B7_HIP(c, hipEventCreate(&c->tev[i][1]));
  }
  c->tev_init = true;
  return B7_OK;
}

int b7_timer_start(b7_ctx *c, int slot) {
  if (!c || slot < 0 || slot >= B7_MAX_TIMERS) return B7_ERR_INVALID;
  B7_TRY(timers_init(c));
  B7_HIP(c, hipEventRecord(c->tev[slot][0], c->stream));
  return B7_OK;
}

int b7_timer_stop(b7_ctx *c, int slot) {
  if (!c || slot < 0 || slot >= B7_MAX_TIMERS) return B7_ERR_INVALID;
  B7_TRY(timers_init(c));
  B7_HIP(c, hipEventRecord(c->tev[slot][1], c->stream));
  return B7_OK;
}

int b7_timer_ms(b7_ctx *c, int slot, float *ms_out) {
  if (!c || slot < 0 || slot >= B7_MAX_TIMERS || !ms_out) return B7_ERR_INVALID;
  B7_TRY(timers_init(c));
  B7_HIP(c, hipEventSynchronize(c->tev[slot][1]));
  B7_HIP(c, hipEventElapsedTime(ms_out, c->tev[slot][0], c->tev[slot][1]));
  return B7_OK;
}

int b7_profile_enable(b7_ctx *c, int on) {
  if (!c) return B7_ERR_INVALID;
  c->profile = on != 0;
  return B7_OK;
}

int b7_profile_reset(b7_ctx *c) {
  if (!c) return B7_ERR_INVALID;
  resolve_phases(c);
  c->phases.clear();
  return B7_OK;
}

int b7_profile_get(b7_ctx *c, const char *phase, double *ms_total, int64_t *launches) {
  if (!c || !phase) return B7_ERR_INVALID;
  resolve_phases(c);
  auto it = c->phases.find(phase);
  if (ms_total) *ms_total = it == c->phases.end() ? 0.0 : it->second.ms;
  if (launches) *launches = it == c->phases.end() ? 0 : it->second.launches;
  return B7_OK;
}

}  // extern "C"
