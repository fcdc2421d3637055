// TEST DOUBLE, not product: the eight RCCL entry points libbot7hip resolves with dlsym (csrc/comm_rccl.h), implemented over
// POSIX shared memory so that several PROCESSES can form a world on ONE GPU (RCCL itself refuses two ranks on one device).
// Loaded through B7_RCCL_LIB by tests/test_sharded_loop.py; it lets b7_eval_nominate / b7_nominate_commit run their
// world > 1 branches for real -- everything except the collective's transport.  All-reduce = every rank copies its buffer to
// its slot, barrier, every rank reduces all slots in rank order, barrier, copy back.  Every wait is bounded.
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
struct ncclComm {
  int rank, n;
  Shm *shm;
  char name[80];
};

namespace {
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
ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { return ncclInvalidUsage; }
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->shm, sizeof(Shm));
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t stream) {
  if (count > (size_t)SLOT_WORDS || (dt != ncclUint64 && dt != ncclDouble)) return ncclInvalidArgument;
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
