// librccl, resolved lazily (dlopen at the first use): shared by the per-process communicator (comm.hip) and the
// single-process group (group.hip).
#pragma once
#include <rccl/rccl.h>

#include <string>

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};
Rccl &rccl();

#define B7_NCCL(c, r, expr)                                                                       \
  do {                                                                                            \
    ncclResult_t e__ = (expr);                                                                    \
    if (e__ != ncclSuccess) return b7_fail((c), B7_ERR_COMM, "RCCL call at %s:%d: %s", __FILE_NAME__, __LINE__, (r).GetErrorString(e__)); \
  } while (0)
