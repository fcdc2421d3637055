// Probe: issue rate of v_mfma_i32_16x16x64_i8 on gfx950 (what an int8 / Ozaki-split variance product would have to beat the fp64 MFMA
// by; diagnostic tool, not product; the idea was dropped: VERDICT r3).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void rate(int *out, int iters, int seed) {
  v4i acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4i){0, 0, 0, 0};
  v4i a = {seed + (int)threadIdx.x, seed * 3, seed * 5, seed * 7}, b = {seed * 11, seed + 1, (int)threadIdx.x, seed};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
  }
  int s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  int *out;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<8>, dim3(256 * 8), dim3(256), 0, 0, out, iters, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = 2.0 * 16 * 16 * 64 * 8.0 * iters * (256.0 * 8 * 4);
    printf("i8 16x16x64 MFMA: %.3f ms -> %.1f TOPS\n", ms, ops / ms / 1e9);
  }
  return 0;
}
