// Probe: does a chain of v_mfma_f64_16x16x4_f64 give the same bits with the accumulator in AGPRs (first step from the
// inline constant 0) as with the accumulator in VGPRs starting from a zeroed register?  (diagnostic tool, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__global__ void chain(const double* A, const double* B, double* outV, double* outA, double* outS, int steps) {
  int l = threadIdx.x;
  d4 c = {0, 0, 0, 0};
  for (int s = 0; s < steps; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[s * 64 + l], B[s * 64 + l], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) outV[l * 4 + r] = c[r];
  // AGPR chain: first step from the literal 0
  double a = A[l], b = B[l];
  asm volatile("v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, 0" ::"v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
  for (int s = 1; s < steps; ++s) {
    a = A[s * 64 + l], b = B[s * 64 + l];
    asm volatile("s_nop 15\n\ts_nop 15\n\tv_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]" ::"v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
  }
  unsigned w[8];
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\t"
               "v_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\tv_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7"
               : "=v"(w[0]), "=v"(w[1]), "=v"(w[2]), "=v"(w[3]), "=v"(w[4]), "=v"(w[5]), "=v"(w[6]), "=v"(w[7]));
  for (int r = 0; r < 4; ++r) outA[l * 4 + r] = __hiloint2double((int)w[2 * r + 1], (int)w[2 * r]);
  // scalar reference orders: fma chain k ascending on top of c
  (void)outS;
}

int main() {
  const int steps = 4;
  std::vector<double> A(steps * 64), B(steps * 64), V(256), G(256);
  srand(1);
  for (auto& x : A) x = (rand() / (double)RAND_MAX) * 2 - 1;
  for (auto& x : B) x = (rand() / (double)RAND_MAX) * 2 - 1;
  double *dA, *dB, *dV, *dG;
  CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dB, B.size() * 8)); CK(hipMalloc(&dV, 2048)); CK(hipMalloc(&dG, 2048));
  CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
  chain<<<1, 64>>>(dA, dB, dV, dG, nullptr, steps); CK(hipDeviceSynchronize());
  CK(hipMemcpy(V.data(), dV, 2048, hipMemcpyDeviceToHost)); CK(hipMemcpy(G.data(), dG, 2048, hipMemcpyDeviceToHost));
  int diff = 0; for (int i = 0; i < 256; ++i) diff += memcmp(&V[i], &G[i], 8) != 0;
  printf("VGPR chain vs AGPR chain (first step C = literal 0): %d of 256 elements differ\n", diff);
  // host references: D[row][col], row=(l>>4)+4r, col=l&15 ; A lane l: row l&15, k l>>4 ; B lane l: k l>>4, col l&15
  int d_seq = 0, d_pair = 0, d_tree = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) + 4 * r, col = l & 15;
    double c1 = 0, c2 = 0, c3 = 0;
    for (int s = 0; s < steps; ++s) {
      double p[4];
      for (int k = 0; k < 4; ++k) p[k] = 0;
      double a[4], b[4];
      for (int k = 0; k < 4; ++k) { a[k] = A[s * 64 + k * 16 + row]; b[k] = B[s * 64 + k * 16 + col]; }
      // (1) sequential fma chain on top of c
      for (int k = 0; k < 4; ++k) c1 = __builtin_fma(a[k], b[k], c1);
      // (2) dot product first (fma chain from 0), then add
      double dsum = 0; for (int k = 0; k < 4; ++k) dsum = __builtin_fma(a[k], b[k], dsum);
      c2 = c2 + dsum;
      // (3) descending chain
      for (int k = 3; k >= 0; --k) c3 = __builtin_fma(a[k], b[k], c3);
    }
    d_seq += memcmp(&c1, &V[l * 4 + r], 8) != 0; d_pair += memcmp(&c2, &V[l * 4 + r], 8) != 0; d_tree += memcmp(&c3, &V[l * 4 + r], 8) != 0;
  }
  printf("host models vs VGPR chain: ascending fma chain %d, dot-then-add %d, descending chain %d differ\n", d_seq, d_pair, d_tree);
  return 0;
}
