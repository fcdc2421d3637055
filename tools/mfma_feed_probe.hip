// Probe: how fast can a wave feed v_mfma_f64_16x16x4_f64 from LDS?  (diagnostic tool, not product)
// One workgroup per CU, WAVES waves (4 = one per SIMD, 8 = two); every wave runs `steps` k-steps of 4 independent MFMAs
// (one A fragment, four B fragments, all five read from LDS as 8-byte reads with the odd row stride the product kernels
// use).  Variants of the loop:
//   0  operands stay in registers (no LDS reads): the pipe's own rate
//   1  read the step's five fragments, then its MFMAs (what a plain `for k: c = mfma(lds[..], lds[..], c)` compiles to)
//   2  the next step's fragments are requested before this step's MFMAs (two register sets, loop unrolled by two)
//   3  as 2, with the reads of step k + 1 placed BETWEEN the MFMAs of step k (one read after each MFMA)
// Reports cycles per MFMA from s_memtime around the loop (wave 0 of every workgroup; median over workgroups).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int STRIDE = 53, ROWS = 80;  // [ROWS][STRIDE] doubles: rows 0..15 the A strip, 16..79 four B tiles

template <int VARIANT>
__global__ void __launch_bounds__(512) feed_kernel(double *out, unsigned long long *cyc, int steps, int reps) {
  extern __shared__ double sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  double *mine = sm;  // every wave reads the same image (the probe is about issue and latency, not capacity)
  for (int e = tid; e < ROWS * STRIDE; e += blockDim.x) mine[e] = 1e-3 * (e % 97) - 0.04;
  __syncthreads();
  const double *pa = mine + lr * STRIDE + lq;
  const double *pb0 = mine + (16 + lr) * STRIDE + lq, *pb1 = pb0 + 16 * STRIDE, *pb2 = pb1 + 16 * STRIDE, *pb3 = pb2 + 16 * STRIDE;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
    if (VARIANT == 0) {
      const double a = pa[0], b0 = pb0[0], b1 = pb1[0], b2 = pb2[0], b3 = pb3[0];
      for (int k = 0; k < steps; ++k) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, c3, 0, 0, 0);
      }
    } else if (VARIANT == 1) {
      for (int k = 0; k < steps; ++k) {
        const double a = pa[4 * k], b0 = pb0[4 * k], b1 = pb1[4 * k], b2 = pb2[4 * k], b3 = pb3[4 * k];
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, c3, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // two register sets; steps is even
      double a = pa[0], b0 = pb0[0], b1 = pb1[0], b2 = pb2[0], b3 = pb3[0];
      for (int k = 0; k < steps; k += 2) {
        double xa, x0, x1, x2, x3;
        if (VARIANT == 2) {
          xa = pa[4 * (k + 1)], x0 = pb0[4 * (k + 1)], x1 = pb1[4 * (k + 1)], x2 = pb2[4 * (k + 1)], x3 = pb3[4 * (k + 1)];
          __builtin_amdgcn_sched_barrier(0);
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, c3, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          const int kn = (k + 2 < steps) ? k + 2 : k;
          a = pa[4 * kn], b0 = pb0[4 * kn], b1 = pb1[4 * kn], b2 = pb2[4 * kn], b3 = pb3[4 * kn];
          __builtin_amdgcn_sched_barrier(0);
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, x0, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, x1, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, x2, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, x3, c3, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          xa = pa[4 * (k + 1)];
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, c0, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          x0 = pb0[4 * (k + 1)];
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, c1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          x1 = pb1[4 * (k + 1)];
          c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, c2, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          x2 = pb2[4 * (k + 1)];
          x3 = pb3[4 * (k + 1)];
          c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, c3, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          const int kn = (k + 2 < steps) ? k + 2 : k;
          a = pa[4 * kn];
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, x0, c0, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          b0 = pb0[4 * kn];
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, x1, c1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          b1 = pb1[4 * kn];
          c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, x2, c2, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          b2 = pb2[4 * kn];
          b3 = pb3[4 * kn];
          c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, x3, c3, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int V>
void run(int waves, int steps, int reps, double clock_ratio) {
  const int blocks = 256;
  double *out;
  unsigned long long *cyc;
  CK(hipMalloc(&out, sizeof(double) * blocks * 512));
  CK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 8));
  const size_t lds = sizeof(double) * ROWS * STRIDE;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(feed_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL(feed_kernel<V>, dim3(blocks), dim3(64 * waves), lds, 0, out, cyc, steps, reps);
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> h(blocks * 8);
  CK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
  std::vector<double> per;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < waves; ++w) per.push_back((double)h[b * 8 + w] * clock_ratio / ((double)steps * reps * 4));
  std::sort(per.begin(), per.end());
  // s_memtime counts core clocks on this chip (s_memrealtime is the 100 MHz one)
  printf("variant %d, %d waves per CU (%d per SIMD), %d steps: %.1f core cycles per MFMA per wave (median), i.e. %.1f per SIMD slot\n", V,
         waves, waves / 4, steps, per[per.size() / 2], per[per.size() / 2] / (waves / 4));
  CK(hipFree(out));
  CK(hipFree(cyc));
}

int main() {
  const double ratio = 1.0;
  for (int waves : {4, 8}) {
    run<0>(waves, 12, 200, ratio);
    run<1>(waves, 12, 200, ratio);
    run<2>(waves, 12, 200, ratio);
    run<3>(waves, 12, 200, ratio);
  }
  return 0;
}
