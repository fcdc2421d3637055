// Probe: cost of the instructions the 16x16 pivot chain is made of, one wave alone on its SIMD (diagnostic tool).
// Each test runs REP copies of an instruction pattern between two s_memtime reads; "dep" = each instruction
// consumes the previous result (latency), "ind" = 8 independent streams (issue rate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
constexpr int REP = 64;

template <int MODE>
__global__ void probe(double *out, unsigned long long *cyc, double seed) {
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + 1e-3 * threadIdx.x + i;
  double m = 1.0 + 1e-9 * threadIdx.x;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int r = 0; r < REP; ++r) {
    if (MODE == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[0]) : "v"(m));
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[i]) : "v"(m));
    }
    if (MODE == 2) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(v[0]));
    if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(v[i]) : "v"(m));
    }
    if (MODE == 4) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(v[0]) : "v"(v[0]), "v"(m));
    if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[(i + 4) & 7]), "v"(m));  // dpp source written 4 instructions earlier, no nop
    }
    if (MODE == 6) asm volatile("v_rcp_f64 %0, %0\n\ts_nop 0" : "+v"(v[0]));
    if (MODE == 7) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f64 %0, %0" : "+v"(v[i]));
    }
    if (MODE == 8) asm volatile("v_rsq_f64 %0, %0\n\ts_nop 0" : "+v"(v[0]));
    if (MODE == 9) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(m), "v"(m));
    }
    if (MODE == 10) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("s_nop 0");
    }
    if (MODE == 11) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(((int *)&v[i])[0]) : "v"(((int *)&m)[0]));
    }
    if (MODE == 12) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[0]) : "v"(m));
    if (MODE == 13) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(((int *)&v[i])[0]) : "v"(((int *)&m)[0]));
    }
    if (MODE == 14) {  // fma chain feeding a dpp broadcast and back: the pivot hand-off (mul -> dpp mov -> fma)
      asm volatile("v_mul_f64 %0, %0, %1\n\ts_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %0, %0, %1, %1" : "+v"(v[0]) : "v"(m));
    }
    if (MODE == 15) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(m));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  double s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char *name, int per_rep) {
  double *out; unsigned long long *cyc, h = 0;
  CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8));
  unsigned long long best = ~0ull;
  for (int t = 0; t < 5; ++t) {
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, 1.25);
    CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    if (h < best) best = h;
  }
  printf("%-44s %7llu ticks / %d instr = %.2f ticks per instruction\n", name, best, REP * per_rep, (double)best / (REP * per_rep));
  CK(hipFree(out)); CK(hipFree(cyc));
}

__global__ void spin(unsigned long long *cyc, long long wall_ticks) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  long long w0 = wall_clock64();
  while (wall_clock64() - w0 < wall_ticks) {}
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  cyc[0] = t1 - t0;
}

int main() {
  unsigned long long *cyc, h;
  CK(hipMalloc(&cyc, 8));
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, cyc, 100000);   // 1 ms of the 100 MHz wall clock
  CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
  printf("s_memtime ticks per ms of wall_clock64: %llu  (=> %.1f MHz)\n", h, h / 1000.0);
  run<0>("v_fma_f64 dependent", 1);
  run<1>("v_fma_f64 x8 independent", 8);
  run<12>("v_mul_f64 dependent", 1);
  run<15>("v_mul_f64 x8 independent", 8);
  run<2>("s_nop1 + v_mov_b64_dpp dependent", 1);
  run<3>("v_mov_b64_dpp x8 independent", 8);
  run<11>("v_mov_b32_dpp x8 independent", 8);
  run<4>("s_nop1 + v_fmac_f64_dpp dependent", 1);
  run<5>("v_fmac_f64_dpp x8, dpp src written 4 instr earlier", 8);
  run<9>("v_fmac_f64_dpp x8 (const src)", 8);
  run<10>("s_nop 0 x8", 8);
  run<6>("v_rcp_f64 dependent (+s_nop 0)", 1);
  run<7>("v_rcp_f64 x8 independent", 8);
  run<8>("v_rsq_f64 dependent (+s_nop 0)", 1);
  run<13>("v_cndmask_b32 x8", 8);
  run<14>("mul -> nop1 -> mov_dpp -> fma dependent (per group)", 1);
  return 0;
}
