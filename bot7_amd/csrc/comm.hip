// The one collective of the path, behind the C ABI: the global arg-max exchange of a candidate-sharded nomination.
//
// Reference: bots/bayesopt.lua:95-96 takes score:max(1) over ALL candidates.  With the grid sharded one process per
// GPU every rank owns rows [offset, offset + M_local) and the same (value, 1-based global index) must
// come out on every rank.  RCCL has no MAXLOC: each rank writes its own record -- (value bits, global index, status, rows
// in its shard, the grid row the index names: B7_TAB_W 64-bit words, b7_internal.h) -- into a zero-initialised
// [world][B7_TAB_W] table and ONE ncclAllReduce(sum, uint64) over xGMI turns the table into an all-gather (adding zeros is
// exact for every bit pattern: NaN payloads and -0.0 survive, indices are exact to 2^63); every rank then applies TH's max
// rule on the host: the first NaN wins, else the largest value, ties to the lowest global index.  The row rides along so
// that the trial loop's `steal` (b7_nominate_commit) needs no second collective; a rank that failed locally sends a
// status record instead of staying away.  800 B per rank, latency-bound: ring per-link bandwidth is irrelevant here.
//
// librccl is resolved lazily (dlopen at the first b7_comm_* call): a single-GPU user never maps its 570 MB, and a
// host that already carries an RCCL (e.g. the one torch bundles, same SONAME) shares that copy.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <chrono>

#include "b7_internal.h"
#include "comm_rccl.h"

Rccl &rccl() {
  static Rccl r;
  if (r.handle || !r.err.empty()) return r;
  // B7_RCCL_LIB (diagnostic build only): the tests' double for RCCL's transport, or a name that does not exist, to see the
  // failure path without a host that lacks the library.  The shipped library resolves librccl by its SONAME and nothing else.
#ifdef B7_DIAG
  const char *forced = getenv("B7_RCCL_LIB");
#else
  const char *forced = nullptr;
#endif
  const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  if (forced) r.handle = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
  else
    for (const char *n : names) {
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
  if (!r.handle) {
    const char *e = dlerror();  // one call: dlerror() clears the message it returns
    r.err = std::string("dlopen(") + (forced ? forced : "librccl.so.1") + "): " + (e ? e : "not found");
    return r;
  }
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
  r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.handle, "ncclCommInitAll"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.handle, "ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.handle, "ncclGroupEnd"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GroupStart || !r.GroupEnd ||
      !r.GetErrorString) {
    r.err = "librccl is missing one of ncclGetUniqueId/CommInitRank/CommInitAll/CommDestroy/AllReduce/GroupStart/GroupEnd/GetErrorString";
    r.handle = nullptr;
  }
  return r;
}

// ---- the exchange, in pieces (b7_eval_nominate, b7_score_finish_global and group.hip assemble them) --------------------

int exch_table_ensure(b7_ctx *c, int world) {
  if (world < 1 || world > B7_MAX_WORLD) return b7_fail(c, B7_ERR_UNSUPPORTED, "exchange: world %d not in [1, %d]", world, B7_MAX_WORLD);
  B7_TRY(b7_ensure(c, c->slots, sizeof(uint64_t) * B7_TAB_W * (size_t)world + sizeof(double) * 128 + 64));
  if (!c->tab_host) {  // + one staging record + the record whose first word is the mirror's completion word
    B7_HIP(c, hipHostMalloc((void **)&c->tab_host, sizeof(uint64_t) * B7_TAB_W * (B7_MAX_WORLD + 2), hipHostMallocMapped));
    B7_HIP(c, hipHostGetDevicePointer((void **)&c->tab_host_dev, c->tab_host, 0));
  }
  return B7_OK;
}

static volatile unsigned *mirror_done_word(b7_ctx *c) {
  return reinterpret_cast<volatile unsigned *>(c->tab_host + (size_t)(B7_MAX_WORLD + 1) * B7_TAB_W);
}

void exch_forget(b7_ctx *c) { c->win_valid = false; }

// score:div + local score:max(1) on the device; this rank's record lands in the table with the index already global
// and the grid row it names beside it (an empty shard contributes a zero record)
// mirror: the kernel also writes this rank's record into the mapped host table and raises the completion word behind it
// (a context that nominates by itself: no copy launch and no stream wait between the arg-max and the answer; exch_wait_mirror)
int exch_local(b7_ctx *c, double divisor, int64_t offset, int rank, int world, bool all_slots, bool mirror) {
  B7_TRY(exch_table_ensure(c, world));
  uint64_t *host_rec = nullptr;
  unsigned *host_done = nullptr;
  if (mirror) {
    *mirror_done_word(c) = 0u;
    host_rec = c->tab_host_dev + (size_t)rank * B7_TAB_W;
    host_done = reinterpret_cast<unsigned *>(c->tab_host_dev + (size_t)(B7_MAX_WORLD + 1) * B7_TAB_W);
  }
  if (c->M > 0 && c->pend.on) {  // the nomination's batched score is still owed: score:add x S, div, arg-max, record in ONE launch
    const b7_ctx::PendingScore ps = c->pend;
    c->pend.on = false;
    return launch_score_finish_slot(c, ps, (double *)c->acc.p, c->M, divisor, (uint64_t *)c->slots.p, rank, world, offset,
                                    (const double *)c->grid[c->grid_cur].p, c->d, all_slots, host_rec, host_done);
  }
  if (c->M > 0) B7_TRY(acc_materialize(c));
  return launch_finish_slot(c, c->M > 0 ? (double *)c->acc.p : nullptr, c->M, divisor, (uint64_t *)c->slots.p, rank, world,
                            offset, c->M > 0 ? (const double *)c->grid[c->grid_cur].p : nullptr, c->d, all_slots, host_rec,
                            host_done);
}

// Everything enqueued before the mirrored arg-max has completed once its word is up (one stream).  Short nominations are
// answered while the host is still spinning; after 500 us the ordinary stream wait takes over (it also surfaces faults).
int exch_wait_mirror(b7_ctx *c) {
  volatile unsigned *done = mirror_done_word(c);
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; c->spin_us > 0; ++spins) {
    if (__atomic_load_n(const_cast<const unsigned *>(done), __ATOMIC_ACQUIRE) != 0u) return B7_OK;
    if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(c->spin_us)) break;
  }
  B7_HIP(c, hipStreamSynchronize(c->stream));
  if (__atomic_load_n(const_cast<const unsigned *>(done), __ATOMIC_ACQUIRE) == 0u)
    return b7_fail(c, B7_ERR_HIP, "nomination: the arg-max kernel finished without raising its completion word");
  return B7_OK;
}

// A rank that could not score its shard still takes part in the collective: its record carries the error code, and
// every rank returns an error together instead of the others waiting in ncclAllReduce for ever.
int exch_fail_record(b7_ctx *c, int rank, int world, int code) {
  B7_TRY(exch_table_ensure(c, world));
  uint64_t *stage = c->tab_host + (size_t)B7_MAX_WORLD * B7_TAB_W;  // the record behind the table's host copy: staging
  B7_HIP(c, hipStreamSynchronize(c->stream));
  memset(stage, 0, sizeof(uint64_t) * B7_TAB_W);
  stage[B7_TAB_STATUS] = (uint64_t)(int64_t)(code < 0 ? -code : 1);
  B7_HIP(c, hipMemsetAsync(c->slots.p, 0, sizeof(uint64_t) * B7_TAB_W * (size_t)world, c->stream));
  B7_HIP(c, hipMemcpyAsync((uint64_t *)c->slots.p + (size_t)rank * B7_TAB_W, stage, sizeof(uint64_t) * B7_TAB_W,
                           hipMemcpyHostToDevice, c->stream));
  return B7_OK;
}

int exch_allreduce(b7_ctx *c) {
  if (!c->comm) return B7_OK;
  Rccl &r = rccl();
  PhaseScope ps(c, "exchange");
  B7_NCCL(c, r, r.AllReduce(c->slots.p, c->slots.p, (size_t)B7_TAB_W * c->comm_world, ncclUint64, ncclSum,
                            static_cast<ncclComm_t>(c->comm), c->stream));
  return B7_OK;
}

int exch_rewrite_record(b7_ctx *c, int rank, int world) { return launch_keep_record(c, (uint64_t *)c->slots.p, rank, world); }

int exch_fetch(b7_ctx *c, int first, int n) {
  B7_HIP(c, hipMemcpyAsync(c->tab_host + (size_t)first * B7_TAB_W, (const uint64_t *)c->slots.p + (size_t)first * B7_TAB_W,
                           sizeof(uint64_t) * B7_TAB_W * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  return B7_OK;
}

// TH max over the gathered records; idx <= 0 marks an empty shard.
bool exch_pick(const uint64_t *tab, int world, int stride, double *val, int64_t *idx1, int *rank_out) {
  bool have = false;
  double bv = 0.0;
  int64_t bi = 0;
  int br = -1;
  for (int r = 0; r < world; ++r) {
    double v;
    memcpy(&v, &tab[(size_t)stride * r], sizeof(double));
    const int64_t i = (int64_t)tab[(size_t)stride * r + 1];
    if (i <= 0) continue;
    if (!have) {
      have = true, bv = v, bi = i, br = r;
      continue;
    }
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) {
      if (vn && (!bn || i < bi)) bv = v, bi = i, br = r;
    } else if (v > bv || (v == bv && i < bi)) {
      bv = v, bi = i, br = r;
    }
  }
  *val = bv;
  *idx1 = bi;
  if (rank_out) *rank_out = br;
  return have;
}

int exch_conclude(b7_ctx *c, const uint64_t *tab, int world, double *best_val, int64_t *best_idx1) {
  exch_forget(c);
  for (int r = 0; r < world; ++r)
    if (tab[(size_t)r * B7_TAB_W + B7_TAB_STATUS] != 0)
      return b7_fail(c, B7_ERR_COMM, "nomination: rank %d could not score its shard (error %lld); no rank nominates", r,
                     -(long long)tab[(size_t)r * B7_TAB_W + B7_TAB_STATUS]);
  double v = 0.0;
  int64_t i = 0;
  int wr = -1;
  if (!exch_pick(tab, world, B7_TAB_W, &v, &i, &wr)) return b7_fail(c, B7_ERR_STATE, "score_finish_global: every shard is empty");
  c->win_valid = true;
  c->win_idx1 = i;
  c->win_rank = wr;
  c->win_d = c->d > 0 ? c->d : c->dfit;
  memcpy(c->win_row, &tab[(size_t)wr * B7_TAB_W + B7_TAB_ROW0], sizeof(double) * B7_MAX_D);
  c->shard_world = world;
  for (int r = 0; r < world; ++r) c->shard_rows[r] = (int64_t)tab[(size_t)r * B7_TAB_W + B7_TAB_ROWS];
  if (best_val) *best_val = v;
  if (best_idx1) *best_idx1 = i;
  return B7_OK;
}

extern "C" {

int b7_comm_pick_winner(const uint64_t *table, int world, double *best_val, int64_t *best_idx1) {
  if (!table || world < 1) return B7_ERR_INVALID;
  double v = 0.0;
  int64_t i = 0;
  if (!exch_pick(table, world, 2, &v, &i, nullptr)) return B7_ERR_STATE;  // every shard empty
  if (best_val) *best_val = v;
  if (best_idx1) *best_idx1 = i;
  return B7_OK;
}

int b7_shard_commit_rule(int64_t idx1_global, int64_t offset, int64_t M_local, int64_t *local_idx1, int64_t *new_offset) {
  if (idx1_global < 1 || offset < 0 || M_local < 0) return B7_ERR_INVALID;
  const bool mine = idx1_global > offset && idx1_global <= offset + M_local;
  if (local_idx1) *local_idx1 = mine ? idx1_global - offset : 0;
  // the deleted row lies in a shard before this one: every row of this shard moves up by one in the union
  if (new_offset) *new_offset = (idx1_global <= offset) ? offset - 1 : offset;
  return B7_OK;
}

int b7_comm_unique_id(void *id_out) {
  if (!id_out) return B7_ERR_INVALID;
  Rccl &r = rccl();
  if (!r.handle) return b7_fail(nullptr, B7_ERR_COMM, "%s", r.err.c_str());
  static_assert(B7_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is RCCL's ncclUniqueId");
  ncclUniqueId id;
  B7_NCCL(nullptr, r, r.GetUniqueId(&id));
  memcpy(id_out, &id, B7_COMM_ID_BYTES);
  return B7_OK;
}

int b7_comm_init(b7_ctx *c, int rank, int world, const void *id_bytes) {
  if (!c) return B7_ERR_INVALID;
  if (!id_bytes || world < 1 || rank < 0 || rank >= world)
    return b7_fail(c, B7_ERR_INVALID, "comm_init: rank %d of %d", rank, world);
  if (world > B7_MAX_WORLD) return b7_fail(c, B7_ERR_UNSUPPORTED, "comm_init: world %d > %d", world, B7_MAX_WORLD);
  if (c->comm) return b7_fail(c, B7_ERR_STATE, "comm_init: this context already has a communicator");
  if (c->group) return b7_fail(c, B7_ERR_STATE, "comm_init: this context belongs to a single-process group");
  Rccl &r = rccl();
  if (!r.handle) return b7_fail(c, B7_ERR_COMM, "%s", r.err.c_str());
  B7_HIP(c, hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(&id, id_bytes, B7_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  B7_NCCL(c, r, r.CommInitRank(&comm, world, id, rank));
  c->comm = comm;
  c->comm_rank = rank;
  c->comm_world = world;
  exch_forget(c);
  return exch_table_ensure(c, world);
}

int b7_comm_info(b7_ctx *c, int *rank, int *world) {
  if (!c) return B7_ERR_INVALID;
  if (rank) *rank = c->comm ? c->comm_rank : 0;
  if (world) *world = c->comm ? c->comm_world : 1;
  return B7_OK;
}

int b7_comm_destroy(b7_ctx *c) {
  if (!c) return B7_ERR_INVALID;
  if (!c->comm) return B7_OK;
  Rccl &r = rccl();
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  ncclComm_t comm = static_cast<ncclComm_t>(c->comm);
  c->comm = nullptr;
  c->comm_world = 1;
  c->comm_rank = 0;
  exch_forget(c);
  if (r.handle) B7_NCCL(c, r, r.CommDestroy(comm));
  return B7_OK;
}

int b7_comm_allreduce_f64(b7_ctx *c, double *inout, int n, int op) {
  if (!c) return B7_ERR_INVALID;
  if (n < 0 || (n > 0 && !inout) || (op != B7_COMM_SUM && op != B7_COMM_MAX && op != B7_COMM_MIN))
    return b7_fail(c, B7_ERR_INVALID, "comm_allreduce_f64: bad arguments");
  if (n > 128) return b7_fail(c, B7_ERR_UNSUPPORTED, "comm_allreduce_f64: n %d > 128 (control-plane values only)", n);
  B7_HIP(c, hipSetDevice(c->device));
  if (!c->comm || n == 0) {  // a context without a communicator is a world of one
    B7_HIP(c, hipStreamSynchronize(c->stream));
    return B7_OK;
  }
  Rccl &r = rccl();
  B7_TRY(exch_table_ensure(c, c->comm_world));
  double *stage = (double *)((uint64_t *)c->slots.p + (size_t)B7_TAB_W * c->comm_world);  // behind the table
  B7_HIP(c, hipMemcpyAsync(stage, inout, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  const ncclRedOp_t rop = op == B7_COMM_SUM ? ncclSum : op == B7_COMM_MAX ? ncclMax : ncclMin;
  B7_NCCL(c, r, r.AllReduce(stage, stage, (size_t)n, ncclDouble, rop, static_cast<ncclComm_t>(c->comm), c->stream));
  B7_HIP(c, hipMemcpyAsync(inout, stage, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

int b7_score_finish_global(b7_ctx *c, double divisor, int64_t global_row_offset, double *best_val,
                           int64_t *best_idx1) {
  if (!c) return B7_ERR_INVALID;
  if (c->group) return b7_fail(c, B7_ERR_STATE, "score_finish_global: this context belongs to a group (b7_group_eval_nominate)");
  const int world = c->comm ? c->comm_world : 1, rank = c->comm ? c->comm_rank : 0;
  int rc = B7_OK;
  if (global_row_offset < 0) rc = b7_fail(c, B7_ERR_INVALID, "score_finish_global: negative row offset");
  else if (c->M > 0 && !c->acc_valid) rc = b7_fail(c, B7_ERR_STATE, "score_finish_global: call b7_score_reset first");
  if (rc == B7_OK) rc = hipSetDevice(c->device) == hipSuccess ? B7_OK : b7_fail(c, B7_ERR_HIP, "hipSetDevice failed");
  if (rc == B7_OK) rc = exch_local(c, divisor, global_row_offset, rank, world, true, !c->comm);
  if (rc != B7_OK) {
    if (world == 1) return rc;
    const std::string own = c->err;
    B7_TRY(exch_fail_record(c, rank, world, rc));  // reach the collective all the same
    B7_TRY(exch_allreduce(c));
    B7_HIP(c, hipStreamSynchronize(c->stream));
    exch_forget(c);
    c->err = own;
    return rc;
  }
  if (!c->comm) {  // no communicator, nobody to exchange with: the record went straight to the mapped host table
    B7_TRY(exch_wait_mirror(c));
    return exch_conclude(c, c->tab_host, world, best_val, best_idx1);
  }
  B7_TRY(exch_allreduce(c));
  B7_TRY(exch_fetch(c, 0, world));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return exch_conclude(c, c->tab_host, world, best_val, best_idx1);
}

int b7_exchange_info(b7_ctx *c, int *world_out, int64_t *rows_per_rank, int64_t *winner_idx1, int *winner_rank,
                     double *winner_row) {
  if (!c) return B7_ERR_INVALID;
  if (!c->win_valid) return b7_fail(c, B7_ERR_STATE, "exchange_info: no nomination since the grid last changed");
  if (world_out) *world_out = c->shard_world;
  if (rows_per_rank) memcpy(rows_per_rank, c->shard_rows, sizeof(int64_t) * c->shard_world);
  if (winner_idx1) *winner_idx1 = c->win_idx1;
  if (winner_rank) *winner_rank = c->win_rank;
  if (winner_row) memcpy(winner_row, c->win_row, sizeof(double) * c->win_d);
  return B7_OK;
}

// bots/abstract.lua:118 on a sharded candidate set: `steal(pending, candidates, idx)` = the winner's row for every rank
// + utils.tensor.remove's stable deletion on the union of the shards (utils/tensor.lua:158-170): the owner compacts its
// shard on the device, every shard behind it moves up by one global index, shards before it are untouched.
int b7_nominate_commit(b7_ctx *c, int64_t idx1_global, int64_t *global_row_offset, double *row_out) {
  if (!c) return B7_ERR_INVALID;
  if (!global_row_offset) return b7_fail(c, B7_ERR_INVALID, "nominate_commit: global_row_offset is NULL");
  if (c->group) return b7_fail(c, B7_ERR_STATE, "nominate_commit: this context belongs to a group (b7_group_nominate_commit)");
  const int world = c->comm ? c->comm_world : 1, rank = c->comm ? c->comm_rank : 0;
  int64_t local = 0, new_off = *global_row_offset;
  if (b7_shard_commit_rule(idx1_global, *global_row_offset, c->M, &local, &new_off) != B7_OK)
    return b7_fail(c, B7_ERR_INVALID, "nominate_commit: index %lld, offset %lld", (long long)idx1_global, (long long)*global_row_offset);
  B7_HIP(c, hipSetDevice(c->device));
  const int d = c->d > 0 ? c->d : c->dfit;  // a rank that never made its (empty) shard still learns the row's width from the data
  // the row: every rank already holds it when idx is the winner of the last exchange (the model-based trials);
  // otherwise (the random initial trials, bots/bayesopt.lua:90-91) the owner broadcasts it with one more all-reduce.
  // Every rank takes the same branch: they all saw the same exchange and are all told the same idx.
  const bool cached = c->win_valid && c->win_idx1 == idx1_global;
  double row[B7_MAX_D];
  if (cached) {
    if (world > 1 && (c->win_rank == rank) != (local > 0))
      return b7_fail(c, B7_ERR_INVALID, "nominate_commit: offset %lld does not agree with the exchange (winner on rank %d)",
                     (long long)*global_row_offset, c->win_rank);
    if (world == 1 && local == 0) return b7_fail(c, B7_ERR_INVALID, "nominate_commit: index %lld outside the grid", (long long)idx1_global);
    memcpy(row, c->win_row, sizeof(double) * B7_MAX_D);
  } else if (world == 1) {
    if (local == 0)
      return b7_fail(c, B7_ERR_INVALID, "nominate_commit: index %lld outside (%lld, %lld]", (long long)idx1_global,
                     (long long)*global_row_offset, (long long)(*global_row_offset + c->M));
    B7_HIP(c, hipMemcpyAsync(row, (const double *)c->grid[c->grid_cur].p + (local - 1) * d, sizeof(double) * d,
                             hipMemcpyDeviceToHost, c->stream));
    B7_HIP(c, hipStreamSynchronize(c->stream));
  } else {
    int rc = exch_table_ensure(c, world);
    if (rc == B7_OK)
      rc = launch_row_slot(c, (uint64_t *)c->slots.p, rank, world, idx1_global, local > 0 ? local - 1 : -1,
                           c->M > 0 ? (const double *)c->grid[c->grid_cur].p : nullptr, d);
    if (rc != B7_OK) B7_TRY(exch_fail_record(c, rank, world, rc));
    B7_TRY(exch_allreduce(c));
    B7_TRY(exch_fetch(c, 0, world));
    B7_HIP(c, hipStreamSynchronize(c->stream));
    B7_TRY(rc);
    int owner = -1;
    for (int r = 0; r < world; ++r) {
      const uint64_t *rec = c->tab_host + (size_t)r * B7_TAB_W;
      if (rec[B7_TAB_STATUS] != 0) return b7_fail(c, B7_ERR_COMM, "nominate_commit: rank %d failed (error %lld)", r, -(long long)rec[B7_TAB_STATUS]);
      if ((int64_t)rec[B7_TAB_IDX] == idx1_global) {
        if (owner >= 0) return b7_fail(c, B7_ERR_INVALID, "nominate_commit: ranks %d and %d both claim index %lld (overlapping offsets)", owner, r, (long long)idx1_global);
        owner = r;
      }
    }
    if (owner < 0) return b7_fail(c, B7_ERR_INVALID, "nominate_commit: index %lld lies in no rank's shard", (long long)idx1_global);
    memcpy(row, c->tab_host + (size_t)owner * B7_TAB_W + B7_TAB_ROW0, sizeof(double) * B7_MAX_D);
  }
  if (local > 0) B7_TRY(grid_drop_row(c, local, nullptr));  // enqueued: the next nomination runs behind it on the same stream
  *global_row_offset = new_off;
  if (row_out) memcpy(row_out, row, sizeof(double) * d);
  exch_forget(c);
  return B7_OK;
}

}  // extern "C"
