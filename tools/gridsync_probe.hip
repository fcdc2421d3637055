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

int main() {
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
