// The one collective of the path, behind the C ABI: the global arg-max exchange of a candidate-sharded nomination.
//
// Reference: bots/bayesopt.lua:95-96 takes score:max(1) over ALL candidates.  With the grid sharded one process per
// GPU every rank owns rows [offset, offset + M_local) and the same (value, 1-based global index) must
// come out on every rank.  RCCL has no MAXLOC: each rank writes its own (value bits, global index) pair into a
// zero-initialised [world, 2] table of 64-bit words and ONE ncclAllReduce(sum, uint64) over xGMI turns the table into
// an all-gather (adding zeros is exact for every bit pattern: NaN payloads and -0.0 survive, indices are exact to
// 2^63); every rank then applies TH's max rule on the host: the first NaN wins, else the largest value, ties to the
// lowest global index.  32 B per rank, latency-bound: ring per-link bandwidth is irrelevant here.
//
// librccl is resolved lazily (dlopen at the first b7_comm_* call): a single-GPU user never maps its 570 MB, and a
// host that already carries an RCCL (e.g. the one torch bundles, same SONAME) shares that copy.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "b7_internal.h"

namespace {

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

Rccl &rccl() {
  static Rccl r;
  if (r.handle || !r.err.empty()) return r;
  const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  for (const char *n : names) {
    r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) break;
  }
  if (!r.handle) {
    r.err = std::string("dlopen(librccl.so.1): ") + (dlerror() ? dlerror() : "not found");
    return r;
  }
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) {
    r.err = "librccl is missing one of ncclGetUniqueId/CommInitRank/CommDestroy/AllReduce/GetErrorString";
    r.handle = nullptr;
  }
  return r;
}

#define B7_NCCL(c, r, expr)                                                                       \
  do {                                                                                            \
    ncclResult_t e__ = (expr);                                                                    \
    if (e__ != ncclSuccess) return b7_fail((c), B7_ERR_COMM, "%s: %s", #expr, (r).GetErrorString(e__)); \
  } while (0)

// Pinned block layout (b7_internal.h): [6144, 8192) is the host copy of the exchange table (<= 64 ranks).
constexpr int SLOT_OFF = 6144;
constexpr int MAX_WORLD = (8192 - SLOT_OFF) / 16;

// TH max over the gathered pairs; idx <= 0 marks an empty shard.
bool pick_winner(const uint64_t *tab, int world, double *val, int64_t *idx1) {
  bool have = false;
  double bv = 0.0;
  int64_t bi = 0;
  for (int r = 0; r < world; ++r) {
    double v;
    memcpy(&v, &tab[2 * r], sizeof(double));
    const int64_t i = (int64_t)tab[2 * r + 1];
    if (i <= 0) continue;
    if (!have) {
      have = true, bv = v, bi = i;
      continue;
    }
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) {
      if (vn && (!bn || i < bi)) bv = v, bi = i;
    } else if (v > bv || (v == bv && i < bi)) {
      bv = v, bi = i;
    }
  }
  *val = bv;
  *idx1 = bi;
  return have;
}

}  // namespace

extern "C" {

int b7_comm_pick_winner(const uint64_t *table, int world, double *best_val, int64_t *best_idx1) {
  if (!table || world < 1) return B7_ERR_INVALID;
  double v = 0.0;
  int64_t i = 0;
  if (!pick_winner(table, world, &v, &i)) return B7_ERR_STATE;  // every shard empty
  if (best_val) *best_val = v;
  if (best_idx1) *best_idx1 = i;
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
  if (world > MAX_WORLD) return b7_fail(c, B7_ERR_UNSUPPORTED, "comm_init: world %d > %d", world, MAX_WORLD);
  if (c->comm) return b7_fail(c, B7_ERR_STATE, "comm_init: this context already has a communicator");
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
  B7_TRY(b7_ensure(c, c->slots, sizeof(uint64_t) * 2 * (size_t)world + 64));
  return B7_OK;
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
  B7_TRY(b7_ensure(c, c->slots, sizeof(double) * 128 + 64));
  B7_HIP(c, hipMemcpyAsync(c->slots.p, inout, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  const ncclRedOp_t rop = op == B7_COMM_SUM ? ncclSum : op == B7_COMM_MAX ? ncclMax : ncclMin;
  B7_NCCL(c, r, r.AllReduce(c->slots.p, c->slots.p, (size_t)n, ncclDouble, rop, static_cast<ncclComm_t>(c->comm),
                            c->stream));
  B7_HIP(c, hipMemcpyAsync(inout, c->slots.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  return B7_OK;
}

int b7_score_finish_global(b7_ctx *c, double divisor, int64_t global_row_offset, double *best_val,
                           int64_t *best_idx1) {
  if (!c) return B7_ERR_INVALID;
  if (global_row_offset < 0) return b7_fail(c, B7_ERR_INVALID, "score_finish_global: negative row offset");
  if (c->M > 0 && !c->acc_valid) return b7_fail(c, B7_ERR_STATE, "score_finish_global: call b7_score_reset first");
  B7_HIP(c, hipSetDevice(c->device));
  const int world = c->comm ? c->comm_world : 1, rank = c->comm ? c->comm_rank : 0;
  B7_TRY(b7_ensure(c, c->slots, sizeof(uint64_t) * 2 * (size_t)world + 64));
  uint64_t *tab_dev = static_cast<uint64_t *>(c->slots.p);
  // score:div + local score:max(1) on the device; the pair lands in this rank's slot of the zeroed table with the
  // index already global (an empty shard contributes (0, 0))
  B7_TRY(launch_finish_slot(c, c->M > 0 ? (double *)c->acc.p : nullptr, c->M, divisor, tab_dev, rank, world,
                            global_row_offset));
  if (c->comm) {
    Rccl &r = rccl();
    PhaseScope ps(c, "exchange");
    B7_NCCL(c, r, r.AllReduce(tab_dev, tab_dev, 2 * (size_t)world, ncclUint64, ncclSum,
                              static_cast<ncclComm_t>(c->comm), c->stream));
  }
  uint64_t *tab = reinterpret_cast<uint64_t *>(static_cast<char *>(c->pinned) + SLOT_OFF);
  B7_HIP(c, hipMemcpyAsync(tab, tab_dev, sizeof(uint64_t) * 2 * world, hipMemcpyDeviceToHost, c->stream));
  B7_HIP(c, hipStreamSynchronize(c->stream));
  double v = 0.0;
  int64_t i = 0;
  if (!pick_winner(tab, world, &v, &i)) return b7_fail(c, B7_ERR_STATE, "score_finish_global: every shard is empty");
  if (best_val) *best_val = v;
  if (best_idx1) *best_idx1 = i;
  return B7_OK;
}

}  // extern "C"
