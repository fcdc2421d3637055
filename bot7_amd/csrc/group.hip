// One host process driving several GPUs: the reference is ONE LuaJIT process (bots/abstract.lua:155-169 run_experiment),
// so the drop-in that keeps its trial loop, its RNG stream and its single call of the user's black-box objective per trial
// (bots/abstract.lua:124) is a group of contexts behind one handle, not eight processes in lock step.
//
// A group shards the candidate grid over its members by contiguous row ranges, replicates the observations,
// and runs bayesopt:eval + nominate (bots/bayesopt.lua:56-99) on all members at once: the fits, posteriors and score:adds
// of every member are enqueued on that member's stream without a host wait in between, then the records of the exchange
// table (b7_internal.h) are combined --
//   * members on distinct devices: one grouped ncclAllReduce (ncclCommInitAll communicators, ncclGroupStart/End) over
//     xGMI, exactly the collective of the one-process-per-GPU layout;
//   * members that share a device ("virtual ranks": RCCL refuses two ranks on one device; this is how the sharding, index
//     and winner logic is exercised on a one-GPU box), or B7_GROUP_EXCHANGE=host: every member copies its own record to
//     pinned host memory and the host merges them --
// and the host waits once per member stream.  The winner rule, the index arithmetic and the commit (stable deletion on the
// union of the shards) are the functions the per-process path uses (comm.hip).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <set>
#include <vector>

#include "b7_internal.h"
#include "comm_rccl.h"

struct b7_group {
  std::vector<b7_ctx *> ctx;
  std::vector<ncclComm_t> comm;  // one per member when the exchange runs over RCCL
  bool use_rccl = false;
  std::string err;
  uint64_t table[B7_TAB_W * B7_MAX_WORLD];  // the merged records of the last exchange
  bool win_valid = false;
  int64_t win_idx1 = 0;
  int win_rank = -1;
  double win_row[B7_MAX_D];
};

namespace {

int gfail(b7_group *g, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (g) g->err = buf;
  return code;
}

// a member's failure becomes the group's, with the rank in front
int member(b7_group *g, int r, int rc) {
  if (rc != B7_OK) g->err = "member " + std::to_string(r) + ": " + g->ctx[r]->err;
  return rc;
}
#define G_TRY(g, r, expr)                     \
  do {                                        \
    int rc__ = member((g), (r), (expr));      \
    if (rc__ != B7_OK) return rc__;           \
  } while (0)

struct Busy {  // lets the group call the grid mutators of its members (b7_internal.h: group_busy)
  b7_group *g;
  explicit Busy(b7_group *g_) : g(g_) { for (b7_ctx *c : g->ctx) c->group_busy = true; }
  ~Busy() { for (b7_ctx *c : g->ctx) c->group_busy = false; }
};

void shard_range(int64_t M, int r, int n, int64_t *lo, int64_t *hi) {  // the first M % n members get one extra row
  const int64_t base = M / n, extra = M % n;
  *lo = r * base + std::min<int64_t>(r, extra);
  *hi = *lo + base + (r < extra ? 1 : 0);
}

int64_t offset_of(const b7_group *g, int r) {
  int64_t off = 0;
  for (int i = 0; i < r; ++i) off += g->ctx[i]->M;
  return off;
}

// the combine step of the exchange, enqueued on every member's stream
int exchange(b7_group *g) {
  const int n = (int)g->ctx.size();
  if (g->use_rccl) {
    Rccl &r = rccl();
    if (r.GroupStart() != ncclSuccess) return gfail(g, B7_ERR_COMM, "ncclGroupStart failed");
    for (int i = 0; i < n; ++i) {
      b7_ctx *c = g->ctx[i];
      if (hipSetDevice(c->device) != hipSuccess) return gfail(g, B7_ERR_HIP, "hipSetDevice(%d) failed", c->device);
      ncclResult_t e = r.AllReduce(c->slots.p, c->slots.p, (size_t)B7_TAB_W * n, ncclUint64, ncclSum, g->comm[i], c->stream);
      if (e != ncclSuccess) {
        (void)r.GroupEnd();
        return gfail(g, B7_ERR_COMM, "ncclAllReduce (member %d): %s", i, r.GetErrorString(e));
      }
    }
    ncclResult_t e = r.GroupEnd();
    if (e != ncclSuccess) return gfail(g, B7_ERR_COMM, "ncclGroupEnd: %s", r.GetErrorString(e));
    if (hipSetDevice(g->ctx[0]->device) != hipSuccess) return gfail(g, B7_ERR_HIP, "hipSetDevice(%d) failed", g->ctx[0]->device);
    G_TRY(g, 0, exch_fetch(g->ctx[0], 0, n));  // every member holds the whole table now: member 0's copy is read
  } else {
    for (int i = 0; i < n; ++i) {
      if (hipSetDevice(g->ctx[i]->device) != hipSuccess) return gfail(g, B7_ERR_HIP, "hipSetDevice(%d) failed", g->ctx[i]->device);
      G_TRY(g, i, exch_fetch(g->ctx[i], i, 1));
    }
  }
  return B7_OK;
}

int gather_table(b7_group *g) {  // after the streams have drained
  const int n = (int)g->ctx.size();
  if (g->use_rccl) memcpy(g->table, g->ctx[0]->tab_host, sizeof(uint64_t) * B7_TAB_W * n);
  else
    for (int i = 0; i < n; ++i)
      memcpy(g->table + (size_t)i * B7_TAB_W, g->ctx[i]->tab_host + (size_t)i * B7_TAB_W, sizeof(uint64_t) * B7_TAB_W);
  return B7_OK;
}

int sync_all(b7_group *g) {
  for (size_t i = 0; i < g->ctx.size(); ++i) {
    b7_ctx *c = g->ctx[i];
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
      return gfail(g, B7_ERR_HIP, "member %zu: stream synchronisation failed", i);
  }
  return B7_OK;
}

}  // namespace

extern "C" {

int b7_group_create(b7_group **out, int n, const int *device_ids) {
  if (!out) return B7_ERR_INVALID;
  *out = nullptr;
  if (n < 1 || n > B7_MAX_WORLD || !device_ids) return b7_fail(nullptr, B7_ERR_INVALID, "group_create: n %d (1..%d) and device_ids required", n, B7_MAX_WORLD);
  b7_group *g = new b7_group();
  for (int i = 0; i < n; ++i) {
    b7_ctx *c = nullptr;
    const int rc = b7_create(&c, device_ids[i]);
    if (rc != B7_OK) {  // b7_create left its message where b7_last_error(NULL) finds it
      for (b7_ctx *m : g->ctx) { m->group = nullptr; b7_destroy(m); }
      delete g;
      return rc;
    }
    c->group = g;
    g->ctx.push_back(c);
  }
  const std::set<int> distinct(device_ids, device_ids + n);
  // B7_GROUP_EXCHANGE: "host" merges on the host even across distinct devices; in the DIAGNOSTIC build "rccl" issues the
  // grouped all-reduce even when members share a device (real RCCL refuses that at ncclCommInitAll; the tests' in-process
  // double accepts it, which is how the grouped branch runs with n > 1 on a one-GPU box)
  const char *mode = getenv("B7_GROUP_EXCHANGE");
  g->use_rccl = (int)distinct.size() == n && !(mode && strcmp(mode, "host") == 0);
#ifdef B7_DIAG
  if (mode && strcmp(mode, "rccl") == 0) g->use_rccl = true;
#endif
  if (g->use_rccl) {
    Rccl &r = rccl();
    ncclResult_t e = ncclSuccess;
    if (r.handle) {
      g->comm.assign(n, nullptr);
      e = r.CommInitAll(g->comm.data(), n, device_ids);
    }
    if (!r.handle || e != ncclSuccess) {
      b7_fail(nullptr, B7_ERR_COMM, "group_create: %s", r.handle ? r.GetErrorString(e) : r.err.c_str());
      g->comm.clear();
      for (b7_ctx *m : g->ctx) { m->group = nullptr; b7_destroy(m); }
      delete g;
      return B7_ERR_COMM;
    }
  }
  for (int i = 0; i < n; ++i)
    if (exch_table_ensure(g->ctx[i], n) != B7_OK) {
      b7_fail(nullptr, B7_ERR_NOMEM, "group_create: member %d: %s", i, g->ctx[i]->err.c_str());
      b7_group_destroy(g);
      return B7_ERR_NOMEM;
    }
  *out = g;
  return B7_OK;
}

void b7_group_destroy(b7_group *g) {
  if (!g) return;
  (void)sync_all(g);
  if (g->use_rccl && rccl().handle)
    for (ncclComm_t cm : g->comm)
      if (cm) (void)rccl().CommDestroy(cm);
  for (b7_ctx *c : g->ctx) {
    c->group = nullptr;
    b7_destroy(c);
  }
  delete g;
}

const char *b7_group_last_error(const b7_group *g) { return g ? g->err.c_str() : b7_last_error(nullptr); }

int b7_group_info(b7_group *g, int *n, int *uses_rccl) {
  if (!g) return B7_ERR_INVALID;
  if (n) *n = (int)g->ctx.size();
  if (uses_rccl) *uses_rccl = g->use_rccl ? 1 : 0;
  return B7_OK;
}

b7_ctx *b7_group_ctx(b7_group *g, int rank) {
  if (!g || rank < 0 || rank >= (int)g->ctx.size()) return nullptr;
  return g->ctx[rank];
}

int b7_group_set_workspace(b7_group *g, int64_t bytes) {
  if (!g) return B7_ERR_INVALID;
  for (size_t i = 0; i < g->ctx.size(); ++i) G_TRY(g, (int)i, b7_set_workspace(g->ctx[i], bytes));
  return B7_OK;
}

int b7_group_gp_set_opts(b7_group *g, const b7_gp_opts *opts) {
  if (!g) return B7_ERR_INVALID;
  for (size_t i = 0; i < g->ctx.size(); ++i) G_TRY(g, (int)i, b7_gp_set_opts(g->ctx[i], opts));
  return B7_OK;
}

// ---- the sharded candidate grid -------------------------------------------------------------------------------------
int b7_group_grid_sobol(b7_group *g, int64_t size, int dims, int64_t skip, const double *mins, const double *maxes) {
  if (!g) return B7_ERR_INVALID;
  if (size < 0) return gfail(g, B7_ERR_INVALID, "group_grid_sobol: size %lld", (long long)size);
  Busy busy(g);
  const int n = (int)g->ctx.size();
  g->win_valid = false;
  if (mins && maxes) {
    for (int i = 0; i < n; ++i) {
      int64_t lo, hi;
      shard_range(size, i, n, &lo, &hi);
      G_TRY(g, i, b7_grid_sobol(g->ctx[i], hi - lo, dims, skip + lo, mins, maxes, nullptr));
    }
    return B7_OK;
  }
  // no map, or a one-sided one (grids/sobol.lua:82-85): that shifts / scales by the column minima / maxima of the WHOLE
  // grid, so the members generate the raw points, report their column ranges, and apply the map with the merged ones
  for (int i = 0; i < n; ++i) {
    int64_t lo, hi;
    shard_range(size, i, n, &lo, &hi);
    G_TRY(g, i, b7_grid_sobol(g->ctx[i], hi - lo, dims, skip + lo, nullptr, nullptr, nullptr));
  }
  if (!mins && !maxes) return B7_OK;
  return b7_group_grid_onesided(g, mins, maxes);
}

int b7_group_grid_random(b7_group *g, int64_t size, int dims, uint64_t seed, const double *mins, const double *maxes) {
  if (!g) return B7_ERR_INVALID;
  if (size < 0) return gfail(g, B7_ERR_INVALID, "group_grid_random: size %lld", (long long)size);
  Busy busy(g);
  const int n = (int)g->ctx.size();
  g->win_valid = false;
  const bool both = mins && maxes;
  for (int i = 0; i < n; ++i) {
    int64_t lo, hi;
    shard_range(size, i, n, &lo, &hi);
    G_TRY(g, i, b7_grid_random(g->ctx[i], hi - lo, dims, seed, lo, both ? mins : nullptr, both ? maxes : nullptr, nullptr));
  }
  if (both || (!mins && !maxes)) return B7_OK;
  return b7_group_grid_onesided(g, mins, maxes);
}

// grids/sobol.lua:82-85, grids/random.lua:29-32 over the union of the members' shards: column minima / maxima of every
// shard, merged on the host (min / max are exact in any order), then the map on every member
int b7_group_grid_onesided(b7_group *g, const double *mins, const double *maxes) {
  if (!g) return B7_ERR_INVALID;
  if ((mins != nullptr) == (maxes != nullptr)) return gfail(g, B7_ERR_INVALID, "group_grid_onesided: exactly one of mins / maxes");
  Busy busy(g);
  const int n = (int)g->ctx.size(), d = g->ctx[0]->d;
  std::vector<double> ext(d), part(2 * (size_t)d);
  bool have = false;
  for (int i = 0; i < n; ++i) {
    if (g->ctx[i]->M == 0) continue;
    G_TRY(g, i, b7_grid_colrange(g->ctx[i], part.data(), part.data() + d));
    const double *src = mins ? part.data() : part.data() + d;
    for (int k = 0; k < d; ++k) ext[k] = !have ? src[k] : (mins ? std::min(ext[k], src[k]) : std::max(ext[k], src[k]));
    have = true;
  }
  if (!have) return B7_OK;
  for (int i = 0; i < n; ++i)
    if (g->ctx[i]->M > 0) G_TRY(g, i, b7_grid_apply_onesided(g->ctx[i], mins, maxes, ext.data()));
  return B7_OK;
}

int b7_group_grid_upload(b7_group *g, const double *X, int64_t M, int d) {
  if (!g) return B7_ERR_INVALID;
  if (M < 0 || (!X && M > 0)) return gfail(g, B7_ERR_INVALID, "group_grid_upload: bad arguments");
  Busy busy(g);
  const int n = (int)g->ctx.size();
  g->win_valid = false;
  for (int i = 0; i < n; ++i) {
    int64_t lo, hi;
    shard_range(M, i, n, &lo, &hi);
    G_TRY(g, i, b7_grid_upload(g->ctx[i], X ? X + lo * d : nullptr, hi - lo, d));
  }
  return B7_OK;
}

int b7_group_grid_shape(b7_group *g, int64_t *M_global, int *d, int64_t *offsets) {
  if (!g) return B7_ERR_INVALID;
  const int n = (int)g->ctx.size();
  if (M_global) *M_global = offset_of(g, n);
  if (d) *d = g->ctx[0]->d;
  if (offsets)
    for (int i = 0; i <= n; ++i) offsets[i] = offset_of(g, i);
  return B7_OK;
}

int b7_group_grid_download(b7_group *g, int64_t row0, int64_t rows, double *out) {
  if (!g) return B7_ERR_INVALID;
  const int n = (int)g->ctx.size();
  const int64_t Mg = offset_of(g, n);
  if (row0 < 0 || rows < 0 || row0 + rows > Mg || (!out && rows > 0))
    return gfail(g, B7_ERR_INVALID, "group_grid_download: rows [%lld, %lld) outside [0, %lld)", (long long)row0, (long long)(row0 + rows), (long long)Mg);
  const int d = g->ctx[0]->d;
  int64_t off = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t Mi = g->ctx[i]->M, a = std::max(row0, off), b = std::min(row0 + rows, off + Mi);
    if (b > a) G_TRY(g, i, b7_grid_download(g->ctx[i], a - off, b - a, out + (a - row0) * d));
    off += Mi;
  }
  return B7_OK;
}

// utils.tensor.remove / steal with an index tensor (utils/tensor.lua:158-193) on the union: indices are 1-based against the
// union BEFORE the call; every member deletes its own in one stable pass
int b7_group_grid_remove_rows(b7_group *g, const int64_t *idx1, int64_t cnt, double *rows_out) {
  if (!g) return B7_ERR_INVALID;
  if (cnt < 0 || (cnt > 0 && !idx1)) return gfail(g, B7_ERR_INVALID, "group_grid_remove_rows: bad index list");
  Busy busy(g);
  const int n = (int)g->ctx.size(), d = g->ctx[0]->d;
  const int64_t Mg = offset_of(g, n);
  for (int64_t k = 0; k < cnt; ++k)
    if (idx1[k] < 1 || idx1[k] > Mg)
      return gfail(g, B7_ERR_INVALID, "group_grid_remove_rows: index %lld outside [1, %lld]", (long long)idx1[k], (long long)Mg);
  g->win_valid = false;
  std::vector<int64_t> offs(n + 1);
  for (int i = 0; i <= n; ++i) offs[i] = offset_of(g, i);
  std::vector<int64_t> local, where;
  std::vector<double> rows;
  for (int i = 0; i < n; ++i) {
    local.clear();
    where.clear();
    for (int64_t k = 0; k < cnt; ++k)
      if (idx1[k] > offs[i] && idx1[k] <= offs[i + 1]) {
        local.push_back(idx1[k] - offs[i]);
        where.push_back(k);
      }
    if (local.empty()) continue;
    rows.resize(local.size() * (size_t)d);
    G_TRY(g, i, b7_grid_remove_rows(g->ctx[i], local.data(), (int64_t)local.size(), rows_out ? rows.data() : nullptr));
    if (rows_out)
      for (size_t j = 0; j < local.size(); ++j) memcpy(rows_out + where[j] * d, rows.data() + j * d, sizeof(double) * d);
  }
  return B7_OK;
}

// ---- the model: the observations go to every member -------------------------------------------------------------------
int b7_group_gp_set_data(b7_group *g, const double *X, const double *Y, int N, int d, int ycols) {
  if (!g) return B7_ERR_INVALID;
  for (size_t i = 0; i < g->ctx.size(); ++i) G_TRY(g, (int)i, b7_gp_set_data(g->ctx[i], X, Y, N, d, ycols));
  return B7_OK;
}

// ---- bayesopt:eval + nominate over all members (bots/bayesopt.lua:56-99) ---------------------------------------------
int b7_group_eval_nominate(b7_group *g, int S, const b7_hyp *hyps, const b7_score_spec *spec, double *best_val,
                           int64_t *best_idx1, double *jitter_out, int *info_out) {
  if (!g) return B7_ERR_INVALID;
  const int n = (int)g->ctx.size();
  if (jitter_out && S > 0) std::fill(jitter_out, jitter_out + S, 0.0);
  if (info_out && S > 0) std::fill(info_out, info_out + S, 0);
  g->win_valid = false;
  for (int i = 0; i < n; ++i) G_TRY(g, i, eval_validate(g->ctx[i], S, hyps, spec));
  if (offset_of(g, n) == 0) return gfail(g, B7_ERR_STATE, "group_eval_nominate: no candidate grid on this group");
  // everything a member has to do, enqueued on its stream; the host moves on to the next member without waiting
  for (int i = 0; i < n; ++i) {
    b7_ctx *c = g->ctx[i];
    if (c->M > 0) G_TRY(g, i, eval_enqueue(c, S, hyps, spec));
    if (hipSetDevice(c->device) != hipSuccess) return gfail(g, B7_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    G_TRY(g, i, exch_local(c, (double)S, offset_of(g, i), i, n, g->use_rccl));
  }
  int rc = exchange(g);
  if (rc != B7_OK) return rc;
  rc = sync_all(g);
  if (rc != B7_OK) return rc;
  // a fit that needed the jitter schedule (or a hand-off that timed out) redoes that member's nomination through the
  // per-sample path; the members fit the same matrices, so the first one's report speaks for all
  bool redone = false;
  std::vector<char> member_redone(n, 0);
  for (int i = 0; i < n; ++i) {
    b7_ctx *c = g->ctx[i];
    if (c->M == 0 || eval_reports_clean(c, S)) continue;
    if (hipSetDevice(c->device) != hipSuccess) return gfail(g, B7_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    G_TRY(g, i, eval_redo(c, S, hyps, spec, redone ? nullptr : jitter_out, redone ? nullptr : info_out));
    G_TRY(g, i, exch_local(c, (double)S, offset_of(g, i), i, n, g->use_rccl));
    redone = true;
    member_redone[i] = 1;
  }
  if (redone) {
    // After the first in-place all-reduce EVERY member's slots hold the whole summed table.  A redone member has just rewritten
    // all of its slots (its new record, zeros elsewhere); every other member -- an empty shard included: its own slot is a
    // valid zero record -- must go back to "own record, zeros elsewhere" too, or the second sum adds a stale copy of every
    // record to the fresh ones
    if (g->use_rccl)
      for (int i = 0; i < n; ++i) {
        b7_ctx *c = g->ctx[i];
        if (!member_redone[i]) {
          if (hipSetDevice(c->device) != hipSuccess) return gfail(g, B7_ERR_HIP, "hipSetDevice(%d) failed", c->device);
          G_TRY(g, i, exch_rewrite_record(c, i, n));
        }
      }
    rc = exchange(g);
    if (rc != B7_OK) return rc;
    rc = sync_all(g);
    if (rc != B7_OK) return rc;
  }
  gather_table(g);
  b7_ctx *c0 = g->ctx[0];
  const int rcc = exch_conclude(c0, g->table, n, best_val, best_idx1);
  if (rcc != B7_OK) {
    g->err = c0->err;
    return rcc;
  }
  g->win_valid = true;
  g->win_idx1 = c0->win_idx1;
  g->win_rank = c0->win_rank;
  memcpy(g->win_row, c0->win_row, sizeof(g->win_row));
  c0->win_valid = false;  // the cache is the group's; a member alone cannot commit
  return B7_OK;
}

// bots/abstract.lua:118 steal(pending, candidates, idx) on the sharded candidate set: the nominee's coordinates come from
// the record of the last exchange when idx is its winner (nothing is copied), else from the owner's grid; the owner deletes
// the row stably on its device (enqueued: the next nomination runs behind it), the shards behind it move up by one
int b7_group_nominate_commit(b7_group *g, int64_t idx1_global, double *row_out) {
  if (!g) return B7_ERR_INVALID;
  Busy busy(g);
  const int n = (int)g->ctx.size();
  int owner = -1;
  int64_t local = 0, off = 0;
  for (int i = 0; i < n && owner < 0; ++i) {
    int64_t loc = 0;
    if (b7_shard_commit_rule(idx1_global, off, g->ctx[i]->M, &loc, nullptr) != B7_OK)
      return gfail(g, B7_ERR_INVALID, "group_nominate_commit: index %lld", (long long)idx1_global);
    if (loc > 0) owner = i, local = loc;
    off += g->ctx[i]->M;
  }
  if (owner < 0) return gfail(g, B7_ERR_INVALID, "group_nominate_commit: index %lld outside [1, %lld]", (long long)idx1_global, (long long)off);
  b7_ctx *c = g->ctx[owner];
  const int d = c->d;
  if (g->win_valid && g->win_idx1 == idx1_global) {
    if (g->win_rank != owner) return gfail(g, B7_ERR_STATE, "group_nominate_commit: the exchange named member %d, the shards say %d", g->win_rank, owner);
    if (row_out) memcpy(row_out, g->win_row, sizeof(double) * d);
    G_TRY(g, owner, grid_drop_row(c, local, nullptr));
  } else {
    G_TRY(g, owner, grid_drop_row(c, local, row_out));
  }
  g->win_valid = false;
  return B7_OK;
}

}  // extern "C"
