// Probe: the posterior kernels of the library (post_kernel_w4<4>, <2>) and the eight-wave kernel they replaced
// (tools/post_kernel_w8.h) on synthetic operands against the host model of the arithmetic stated at the top of
// bot7_amd/csrc/posterior.hip:
//   v[n][c] = ascending fma chain over k of Linv[n][k] * ks[c][k]   (v_mfma_f64_16x16x4_f64, tools/mfma_acc_probe.hip)
//   ss[c]   = the fixed-order sum of squares
// Every kernel must reproduce the model bit for bit.  (diagnostic tool, not product; includes the kernel source directly)
#include "../bot7_amd/csrc/posterior.hip"
#include "post_kernel_w8.h"
#include <cstdio>
#include <cstring>
#include <vector>

int main(int argc, char **argv) {
  const int Npad = argc > 1 ? atoi(argv[1]) : 128, nnz = argc > 2 ? atoi(argv[2]) : 5, row_lo = argc > 3 ? atoi(argv[3]) : 0;
  const int M = 256;
  std::vector<double> L((size_t)Npad * Npad, 0.0), K((size_t)M * Npad, 0.0), V4(M), V8(M);
  srand(3);
  for (int n = row_lo; n < nnz; ++n)
    for (int k = 0; k <= n; ++k) L[(size_t)n * Npad + k] = rand() / (double)RAND_MAX - 0.5;
  for (int c = 0; c < M; ++c)
    for (int k = 0; k < nnz; ++k) K[(size_t)c * Npad + k] = rand() / (double)RAND_MAX - 0.5;
  double *dL, *dK, *dV;
  hipMalloc(&dL, L.size() * 8); hipMalloc(&dK, K.size() * 8); hipMalloc(&dV, M * 8);
  hipMemcpy(dL, L.data(), L.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dK, K.data(), K.size() * 8, hipMemcpyHostToDevice);
  std::vector<double> V2(M);
  {
    auto kern = w4::post_kernel_w4<4>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, w4::lds_bytes<4>());
    hipLaunchKernelGGL(kern, dim3(M / 256), dim3(256), w4::lds_bytes<4>(), 0, dL, dK, Npad, (int64_t)0, (int64_t)M, 0.0, 1.0, 0.0, 0, 0.0, dV, PostBatch{});
    hipDeviceSynchronize();
    hipMemcpy(V4.data(), dV, M * 8, hipMemcpyDeviceToHost);
  }
  {
    auto kern = w4::post_kernel_w4<2>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, w4::lds_bytes<2>());
    hipLaunchKernelGGL(kern, dim3(M / 128), dim3(256), w4::lds_bytes<2>(), 0, dL, dK, Npad, (int64_t)0, (int64_t)M, 0.0, 1.0, 0.0, 0, 0.0, dV, PostBatch{});
    hipDeviceSynchronize();
    hipMemcpy(V2.data(), dV, M * 8, hipMemcpyDeviceToHost);
  }
  std::vector<double> VT(M, 0.0);
  if (Npad % 256 == 0) {
    using T = w4::Tall<2, 16>;
    auto kern = w4::post_kernel_w4t<2, 16>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3(M / T::BN), dim3(256), T::LDS_BYTES, 0, dL, dK, Npad, (int64_t)0, (int64_t)M, 0.0, 1.0, 0.0, 0, 0.0, dV, PostBatch{});
    hipDeviceSynchronize();
    hipMemcpy(VT.data(), dV, M * 8, hipMemcpyDeviceToHost);
  }
  {
    using GP = GemmF64<128, 256, 16, 2, 4, false, 1>;
    auto kern = w8::post_kernel<128, 256, 2, 4, 2, 1, true, true>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GP::LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3(M / 256), dim3(512), GP::LDS_BYTES, 0, dL, dK, Npad, (int64_t)0, (int64_t)M, 0.0, 1.0, 0.0, 0, 0.0, dV);
    hipDeviceSynchronize();
    hipMemcpy(V8.data(), dV, M * 8, hipMemcpyDeviceToHost);
  }
  // host model: per candidate c, rows n: v = ascending fma chain over k; squares folded in the kernels' order:
  // lane group g = 0..3 holds rows 16 I + g + 4 r; per 128-row tile and half h: s = sum_{I in half} sum_r v^2 (fma chain),
  // colss[h] += s; then (g0 + g1) + (g2 + g3) per half, then half0 + half1
  int bad4 = 0, bad2 = 0, bad8 = 0, badt = 0;
  for (int c = 0; c < M; ++c) {
    double colss[2][4] = {};
    for (int t = 0; t < Npad / 128; ++t)
      for (int h = 0; h < 2; ++h)
        for (int g = 0; g < 4; ++g) {
          double s = 0.0;
          for (int I = 4 * h; I < 4 * h + 4; ++I)
            for (int r = 0; r < 4; ++r) {
              const int n = t * 128 + 16 * I + g + 4 * r;
              double v = 0.0;
              for (int k = 0; k < Npad; ++k) v = __builtin_fma(L[(size_t)n * Npad + k], K[(size_t)c * Npad + k], v);
              s = __builtin_fma(v, v, s);
            }
          colss[h][g] += s;
        }
    double hv[2];
    for (int h = 0; h < 2; ++h) {
      double a = colss[h][0] + colss[h][1], b = colss[h][2] + colss[h][3];
      hv[h] = a + b;
    }
    double ss = hv[0];
    ss += hv[1];
    bad4 += memcmp(&ss, &V4[c], 8) != 0;
    bad2 += memcmp(&ss, &V2[c], 8) != 0;
    bad8 += memcmp(&ss, &V8[c], 8) != 0;
    badt += Npad % 256 == 0 && memcmp(&ss, &VT[c], 8) != 0;
  }
  printf("Npad %d, rows [%d, %d) nonzero: of %d candidates, differing from the host model: w4<4> %d, w4<2> %d, w8 %d, tall %d\n", Npad, row_lo, nnz, M, bad4, bad2, bad8, badt);
}
