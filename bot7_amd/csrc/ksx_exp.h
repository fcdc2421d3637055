// The exponential of the ARD-SE covariance, shared by covar.hip (K(X*,X), K(X,X)) and nll_small.hip (the fused small-N
// likelihood): identical code, so identical bits.
#pragma once
#include <hip/hip_runtime.h>

#include "exp_table.h"
#ifndef B7_KSX_ABLATE
#define B7_KSX_ABLATE 0
#endif

// amp * exp(min(arg, 0)) with NaN passing (utils/math.lua:106's clamp of the distance at 0 is TH's: it compares, it does not
// sanitise), in 16 VALU instructions (the Taylor-13 form it replaces took 27: 36 -> 27 per output of the kernel, which is
// bound by VALU + MFMA issue on the one fp64 pipe):
//   a  = max(min(arg, 0), -1000)               v_min / v_max drop a NaN; it is put back below
//   nb = fma(a, 128/ln2, 1.5 * 2^52)           the low mantissa bits of nb ARE n = rint(a * 128/ln2) (two's complement)
//   r  = a - n * ln2/128 in two pieces         HEAD has 35 bits, so n * HEAD is exact for |n| < 2^18; |r| <= ln2/256
//   q  = r * (1 + r/2 + r^2/6 + r^3/24 + r^4/120)        e^r - 1, truncation r^6/720 <= 5.5e-19
//   q  = fma(arg, 0.0, q)                      NaN (or inf) in arg -> NaN; otherwise adds a signed zero
//   amp * exp(a) = 2^(n >> 7) * T[n & 127] * (1 + q),  T[j] = amp * 2^(j/128) in LDS, v_ldexp for the power of two
// (gradual underflow; a = -1000 gives exactly 0).  Measured against long-double exp over 2e7 arguments in [-40, 0]:
// relative error <= 1.85 * 2^-53 with amp = 1 (tools/gen_exp_table.py writes the table and the constants).
// Padding observations carry zs/2 = 1e300 (not +inf: inf * 0 would make the NaN carrier fire): arg = -1e300 -> exactly 0.
// Four arguments at a time, stage by stage: one such chain is 14 dependent fp64 instructions (8.6 cycles each when the next
// one waits for it, 4.8 when it does not), and the compiler, left to itself, ran the four chains of a 16x16 tile nearly one
// after the other -- the epilogue was bound by latency, not by issue (tools/ksx_ablate.py: a third fewer instructions
// changed nothing).  Written as stages over r = 0..3 the four chains interleave and each instruction's latency is covered
// by the other three.
// keeps the instruction scheduler from moving anything across: without it the four Horner chains are emitted one after the
// other again (it minimises live registers; there are plenty here)
#define B7_STAGE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void amp_exp_nonpos4(const double (&arg)[4], const double *__restrict__ tab, double (&out)[4]) {
  double a[4], nb[4], nf[4], r[4], p[4], q[4], t[4];
  int n[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = __builtin_fmin(arg[i], 0.0);
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = __builtin_fmax(a[i], -1000.0);
#pragma unroll
  for (int i = 0; i < 4; ++i) nb[i] = __builtin_fma(a[i], B7_EXP_INV, B7_EXP_MAGIC);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    n[i] = __double2loint(nb[i]);
#if B7_KSX_ABLATE & 2    // every lane reads the same table entry: no LDS bank conflicts
    t[i] = tab[(n[i] >> 20) & 1];
#else
    t[i] = tab[n[i] & 127];
#endif
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) nf[i] = nb[i] - B7_EXP_MAGIC;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __builtin_fma(nf[i], -B7_EXP_HEAD, a[i]);
B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __builtin_fma(nf[i], -B7_EXP_TAIL, r[i]);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = __builtin_fma(r[i], 1.0 / 120.0, 1.0 / 24.0);
B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = __builtin_fma(p[i], r[i], 1.0 / 6.0);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = __builtin_fma(p[i], r[i], 0.5);
B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = __builtin_fma(p[i], r[i], 1.0);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = r[i] * p[i];
B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = __builtin_fma(arg[i], 0.0, q[i]);
  B7_STAGE();
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = __builtin_ldexp(__builtin_fma(t[i], q[i], t[i]), n[i] >> 7);
}

