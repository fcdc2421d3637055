// TEST DOUBLE, not product: the eight RCCL entry points libbot7hip resolves with dlsym (csrc/comm_rccl.h), implemented over
// POSIX shared memory so that several PROCESSES can form a world on ONE GPU (RCCL itself refuses two ranks on one device).
// Loaded through B7_RCCL_LIB by tests/test_sharded_loop.py; it lets b7_eval_nominate / b7_nominate_commit run their
// world > 1 branches for real -- everything except the collective's transport.  All-reduce = every rank copies its buffer to
// its slot, barrier, every rank reduces all slots in rank order, barrier, copy back.  Every wait is bounded.
// ncclCommInitAll (one process, n communicators: b7_group_*) is served IN the process: the all-reduces issued between
// ncclGroupStart and ncclGroupEnd are queued and carried out together at ncclGroupEnd (drain every member's stream, sum the
// members' device buffers on the host, write the sum back to each) -- device ids may repeat, which is what lets the grouped
// branch of csrc/group.hip run with n > 1 on a one-GPU box.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <vector>

namespace {
constexpr int MAX_RANKS = 16, SLOT_WORDS = 8192;
struct Shm {
  std::atomic<unsigned> arrived, generation, attached;
  unsigned pad[13];
  unsigned long long slots[MAX_RANKS][SLOT_WORDS];
};
}  // namespace
struct LocalWorld {  // ncclCommInitAll: the n communicators of one process
  int n, alive;
};
struct ncclComm {
  int rank, n;
  Shm *shm;
  char name[80];
  LocalWorld *local;  // non-null: an in-process communicator
  int device;
};

namespace {
struct Pending {
  const void *send;
  void *recv;
  size_t count;
  ncclDataType_t dt;
  ncclRedOp_t op;
  ncclComm *comm;
  hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Pending> g_pending;

double combine(double acc, double v, ncclRedOp_t op) { return op == ncclSum ? acc + v : op == ncclMax ? (v > acc ? v : acc) : (v < acc ? v : acc); }

// the queued all-reduces of one in-process world, all at once (every member must have issued exactly one, same count / type)
ncclResult_t run_local(std::vector<Pending> &q) {
  if (q.empty()) return ncclSuccess;
  const LocalWorld *w = q[0].comm->local;
  if ((int)q.size() != w->n) return ncclInvalidUsage;
  const size_t count = q[0].count;
  std::vector<std::vector<unsigned long long>> h(q.size(), std::vector<unsigned long long>(count));
  for (size_t i = 0; i < q.size(); ++i) {
    if (q[i].comm->local != w || q[i].count != count || q[i].dt != q[0].dt) return ncclInvalidUsage;
    if (hipSetDevice(q[i].comm->device) != hipSuccess || hipStreamSynchronize(q[i].stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(h[i].data(), q[i].send, 8 * count, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  }
  std::vector<unsigned long long> out(count);
  for (size_t k = 0; k < count; ++k) {
    if (q[0].dt == ncclUint64) {
      unsigned long long acc = 0;
      for (size_t i = 0; i < q.size(); ++i) acc += h[i][k];
      out[k] = acc;
    } else {
      double acc = 0;
      for (size_t i = 0; i < q.size(); ++i) {
        double v;
        memcpy(&v, &h[i][k], 8);
        acc = i == 0 ? v : combine(acc, v, q[0].op);
      }
      memcpy(&out[k], &acc, 8);
    }
  }
  for (size_t i = 0; i < q.size(); ++i) {
    if (hipSetDevice(q[i].comm->device) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(q[i].recv, out.data(), 8 * count, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}

bool barrier(ncclComm *c) {
  Shm *s = c->shm;
  const unsigned gen = s->generation.load();
  if (s->arrived.fetch_add(1) + 1 == (unsigned)c->n) {
    s->arrived.store(0);
    s->generation.fetch_add(1);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (s->generation.load() == gen) {
    usleep(20);
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return false;
  }
  return true;
}
}  // namespace

extern "C" {
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "rccl_shm_stub: failure or time-out"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "b7stub_%d_%ld", (int)getpid(), (long)std::chrono::steady_clock::now().time_since_epoch().count());
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  ncclComm *c = new ncclComm();
  c->rank = rank, c->n = nranks;
  c->local = nullptr;
  (void)hipGetDevice(&c->device);
  char clean[64];
  int k = 0;
  for (int i = 0; i < 48 && id.internal[i]; ++i) {
    const char ch = id.internal[i];
    clean[k++] = ((ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || (ch >= '0' && ch <= '9') || ch == '_') ? ch : 'x';
  }
  clean[k] = 0;
  snprintf(c->name, sizeof(c->name), "/%s", k ? clean : "b7stub");
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Shm)) != 0) return ncclSystemError;
  c->shm = static_cast<Shm *>(mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
  close(fd);
  if (c->shm == MAP_FAILED) return ncclSystemError;
  c->shm->attached.fetch_add(1);   // a fresh object is zero-filled
  const auto t0 = std::chrono::steady_clock::now();
  while (c->shm->attached.load() < (unsigned)nranks) {
    usleep(50);
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return ncclSystemError;
  }
  *out = c;
  return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t *out, int n, const int *devs) {
  if (!out || n < 1 || n > 64) return ncclInvalidArgument;
  LocalWorld *w = new LocalWorld{n, n};
  for (int i = 0; i < n; ++i) {
    ncclComm *c = new ncclComm();
    c->rank = i, c->n = n, c->shm = nullptr, c->local = w, c->name[0] = 0;
    c->device = devs ? devs[i] : i;
    out[i] = c;
  }
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->local) {
    if (--c->local->alive == 0) delete c->local;
    delete c;
    return ncclSuccess;
  }
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->shm, sizeof(Shm));
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
  if (g_depth > 0 && --g_depth > 0) return ncclSuccess;
  std::vector<Pending> q;
  q.swap(g_pending);
  return run_local(q);
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t stream) {
  if (count > (size_t)SLOT_WORDS || (dt != ncclUint64 && dt != ncclDouble)) return ncclInvalidArgument;
  if (c->local) {  // an in-process world: queued until the group closes (alone: a world of one, or a usage error)
    g_pending.push_back(Pending{send, recv, count, dt, op, c, stream});
    if (g_depth > 0) return ncclSuccess;
    std::vector<Pending> q;
    q.swap(g_pending);
    return run_local(q);
  }
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  std::vector<unsigned long long> h(count);
  if (hipMemcpy(h.data(), send, 8 * count, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  memcpy(c->shm->slots[c->rank], h.data(), 8 * count);
  if (!barrier(c)) return ncclSystemError;
  for (size_t i = 0; i < count; ++i) {
    if (dt == ncclUint64) {
      unsigned long long acc = 0;
      for (int r = 0; r < c->n; ++r) acc += c->shm->slots[r][i];
      h[i] = acc;
    } else {
      double acc = 0;
      for (int r = 0; r < c->n; ++r) {
        double v;
        memcpy(&v, &c->shm->slots[r][i], 8);
        acc = r == 0 ? v : (op == ncclSum ? acc + v : op == ncclMax ? (v > acc ? v : acc) : (v < acc ? v : acc));
      }
      memcpy(&h[i], &acc, 8);
    }
  }
  if (!barrier(c)) return ncclSystemError;   // nobody rewrites a slot before everyone has read it
  if (hipMemcpy(recv, h.data(), 8 * count, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}
}
