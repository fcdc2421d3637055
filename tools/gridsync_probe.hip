// Probe: cost of a cooperative-groups grid barrier on this GPU (one cooperative launch, R barriers), to price a
// single-launch Cholesky against its 64 dependent launches.  Diagnostic tool.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__global__ void __launch_bounds__(256) sync_kernel(double *buf, int rounds) {
  cg::grid_group grid = cg::this_grid();
  double v = buf[blockIdx.x * 256 + threadIdx.x];
  for (int r = 0; r < rounds; ++r) {
    v = v * 1.0000001 + 1.0;
    buf[((blockIdx.x + r) % gridDim.x) * 256 + threadIdx.x] = v;   // something another workgroup reads next round
    grid.sync();
    v += buf[((blockIdx.x + r + 1) % gridDim.x) * 256 + threadIdx.x];
  }
  buf[blockIdx.x * 256 + threadIdx.x] = v;
}

// Hand-rolled barrier: one arrival counter + a generation word, agent-scope atomics, thread 0 of each workgroup
// spins (bounded) on the generation.  Launched cooperatively so that every workgroup is resident.
__device__ __forceinline__ void gbar(unsigned *ctr, unsigned *gen, unsigned nblocks, int *timeout) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
      __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      int spins = 0;
      while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {
        if (++spins > (1 << 22)) { *timeout = 1; break; }
      }
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) gbar_kernel(double *buf, int rounds, unsigned *ctr, unsigned *gen, int *timeout) {
  double v = buf[blockIdx.x * 256 + threadIdx.x];
  for (int r = 0; r < rounds; ++r) {
    v = v * 1.0000001 + 1.0;
    __hip_atomic_store(&buf[((blockIdx.x + r) % gridDim.x) * 256 + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    gbar(ctr, gen, gridDim.x, timeout);
    v += __hip_atomic_load(&buf[((blockIdx.x + r + 1) % gridDim.x) * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  buf[blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
  for (int blocks : {32, 128, 256, 512}) {
    double *buf; unsigned *sync2; int *timeout, hto = 0;
    CK(hipMalloc(&buf, sizeof(double) * 512 * 256));
    CK(hipMalloc(&sync2, 256)); CK(hipMalloc(&timeout, 4));
    CK(hipMemset(buf, 0, sizeof(double) * 512 * 256)); CK(hipMemset(sync2, 0, 256)); CK(hipMemset(timeout, 0, 4));
    unsigned *ctr = sync2, *gen = sync2 + 32;
    for (int rounds : {1, 201}) {
      void *args[] = {&buf, &rounds, &ctr, &gen, &timeout};
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipError_t e = hipLaunchCooperativeKernel((void *)gbar_kernel, dim3(blocks), dim3(256), args, 0, 0);
        if (e != hipSuccess) { printf("blocks %d: cooperative launch refused: %s\n", blocks, hipGetErrorString(e)); break; }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      CK(hipMemcpy(&hto, timeout, 4, hipMemcpyDeviceToHost));
      printf("hand-rolled barrier: blocks %3d rounds %3d: %.1f us%s\n", blocks, rounds, best * 1e3, hto ? "  (SPIN TIMEOUT)" : "");
    }
    CK(hipFree(buf)); CK(hipFree(sync2)); CK(hipFree(timeout));
  }
  printf("cooperative_groups grid.sync():\n");
  for (int blocks : {32, 128, 256, 512}) {
    double *buf;
    CK(hipMalloc(&buf, sizeof(double) * 512 * 256));
    CK(hipMemset(buf, 0, sizeof(double) * 512 * 256));
    for (int rounds : {1, 201}) {
      void *args[] = {&buf, &rounds};
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipError_t e = hipLaunchCooperativeKernel((void *)sync_kernel, dim3(blocks), dim3(256), args, 0, 0);
        if (e != hipSuccess) { printf("blocks %d: cooperative launch refused: %s\n", blocks, hipGetErrorString(e)); break; }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("blocks %3d rounds %3d: %.1f us\n", blocks, rounds, best * 1e3);
    }
    CK(hipFree(buf));
  }
  return 0;
}
