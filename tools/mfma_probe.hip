// Probe: v_mfma_f64_16x16x4_f64 fragment layout and issue rate on gfx950 (diagnostic tool, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__global__ void layout_kernel(const double* A, const double* B, double* D) {
  // A: 16x4 row-major, B: 4x16 row-major, D: 16x16 row-major
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];  // raw dump: lane-major
}

template <int NACC>
__global__ void rate_kernel(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MFMA wave + VALU fp64 FMA wave co-issue test: even waves MFMA, odd waves v_fma_f64
__global__ void mix_kernel(double* out, int iters, double a0, double b0, int mode) {
  int wave = threadIdx.x >> 6;
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  double s = 0;
  bool do_mfma = (mode == 0) || (mode == 2 && (wave & 4) == 0);
  bool do_valu = (mode == 1) || (mode == 2 && (wave & 4) != 0);
  if (do_mfma) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  }
  if (do_valu) {
    double x[16];
    for (int i = 0; i < 16; ++i) x[i] = i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(a, x[i], b);
    }
    for (int i = 0; i < 16; ++i) s += x[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz arch %s\n", p.name, p.multiProcessorCount, p.clockRate, p.gcnArchName);
  // layout
  std::vector<double> A(64), B(64), D(256);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = (i + 1) * 100 + k;        // asymmetric
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (k == 0) ? (j + 1) * 0.001 : (k==1? 1e-7*(j+1):0);
  double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  layout_kernel<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
  CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
  // check hypothesis: lane l reg r -> row=(l>>4)+4*r, col=l&15
  int bad1 = 0, bad2 = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    double got = D[l * 4 + r];
    auto ref = [&](int row, int col) { double s = 0; for (int k = 0; k < 4; ++k) s += A[row * 4 + k] * B[k * 16 + col]; return s; };
    if (got != ref((l >> 4) + 4 * r, l & 15)) bad1++;
    if (got != ref((l >> 4) * 4 + r, l & 15)) bad2++;
  }
  printf("layout: f64-form(row=(l>>4)+4r) mismatches=%d ; f32-form(row=4(l>>4)+r) mismatches=%d\n", bad1, bad2);

  // rate
  int nblk = p.multiProcessorCount * 4; double* out; CK(hipMalloc(&out, sizeof(double) * nblk * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](auto launch, const char* name, double flops) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-40s %8.3f ms  %8.2f TFLOP/s\n", name, ms, flops / ms * 1e-9);
  };
  int iters = 20000;
  for (int wpb : {4, 8}) {
    double fl = (double)p.multiProcessorCount * wpb * iters * 4 * 2048.0;
    char nm[64]; snprintf(nm, 64, "mfma f64 4acc %d waves/CU", wpb);
    timeit([&] { rate_kernel<4><<<p.multiProcessorCount, wpb * 64>>>(out, iters, 1.0, 2.0); }, nm, fl);
  }
  {
    double fl = (double)p.multiProcessorCount * 4 * iters * 1 * 2048.0;
    timeit([&] { rate_kernel<1><<<p.multiProcessorCount, 256>>>(out, iters, 1.0, 2.0); }, "mfma f64 1acc 4 waves/CU (dep chain)", fl);
    fl = (double)p.multiProcessorCount * 4 * iters * 16 * 2048.0;
    timeit([&] { rate_kernel<16><<<p.multiProcessorCount, 256>>>(out, iters/4, 1.0, 2.0); }, "mfma f64 16acc 4 waves/CU", fl/4);
  }
  {
    // mode0: 8 waves all MFMA; mode1: 8 waves all VALU fma; mode2: 4 MFMA + 4 VALU
    double fm = (double)p.multiProcessorCount * 8 * iters * 4 * 2048.0;
    double fv = (double)p.multiProcessorCount * 8 * iters * 64.0 * 64 * 2;
    timeit([&] { mix_kernel<<<p.multiProcessorCount, 512>>>(out, iters, 1.0, 2.0, 0); }, "mix mode0 (8 waves mfma)", fm);
    timeit([&] { mix_kernel<<<p.multiProcessorCount, 512>>>(out, iters, 1.0, 2.0, 1); }, "mix mode1 (8 waves valu fma64)", fv);
    timeit([&] { mix_kernel<<<p.multiProcessorCount, 512>>>(out, iters, 1.0, 2.0, 2); }, "mix mode2 (4 mfma + 4 valu)", fm / 2 + fv / 2);
  }
  return 0;
}
