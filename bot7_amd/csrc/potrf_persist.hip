// Blocked Cholesky + explicit inverse as ONE persistent launch with point-to-point hand-offs (Npad <= 4096).
//
// Same arithmetic as the per-panel launch schedule of potrf.hip, element for element (utils/math.lua:165 torch.potrf +
// the inline inverse): every 64x64 tile (I, J) of the lower triangle is
//     C_0 = K_IJ (+ eps on the diagonal),   C_{q+1} = C_q - L_Iq L_Jq'  for q = 0 .. J-1  (a 64-deep MFMA chain from
//     zero, subtracted),   L_JJ = chol(C_J),   L_IJ = C_J inv(L_JJ)'  (the triangular chain of potrf_trsm_kernel<NEAR>),
// and row block p of inv(L) is  -inv(L_pp) * sum over 128-deep K chunks (ascending) of L[p, kc] inv(L)[kc, j].
// Only the SCHEDULE differs.  The launch schedule pays two dependent launches per panel (>= 4.2 us each before doing
// anything); here the dependent chain lives inside one workgroup and everything else is pulled by
// workgroups that wait on flags:
//
//   workgroup 0 ("critical"): for p = 0 .. nb-1: factor + invert the diagonal block in LDS (potrf_diag.h), publish
//     inv(L_pp) at once (every helper of the panel waits for it); then the look-ahead that the next diagonal block waits
//     for, without leaving the CU: L[p+1][p] = C(p+1, p) inv(L_pp)'  and  C(p+1, p+1) -= L[p+1][p] L[p+1][p]' -- of the
//     update only column block 0 before the next factorisation starts: the other sub-tiles, the store of L[p+1][p] and
//     its flag are done by waves 1..3 inside that factorisation's first 16x16 step, where they would idle, and the two
//     tiles of the panel after that are fetched into LDS by the same waves in its later steps (diag_core's hook).
//   workgroups 1 .. H ("helpers"): pull tile jobs from one queue (an atomic counter over a list sorted by the panel at
//     which a job can finish).  A job owns its tile in REGISTERS from C_0 to the end: it waits for the two L tiles of
//     the next panel update (flags), applies it, and after the last one waits for inv(L_JJ) and publishes L_IJ.  Tiles
//     (p, p-1) and (p, p) are brought up to panel p-2 by a helper and handed to the critical workgroup.  The tiles of
//     inv(L) are jobs of the same kind; the structurally zero upper tiles are filled by the first jobs of the queue,
//     while the helpers would otherwise idle until the first panel is factored.
//
// Hand-offs follow cdna_hip_programming.md Guideline 16 in its write-through form: every byte another workgroup reads
// inside the launch is stored with sc1 (buffer_store ... sc1), every storing wave drains (s_waitcnt vmcnt(0)), the
// workgroup's barrier, then ONE lane stores the flag (agent-scope relaxed atomic = sc1 store); a consumer polls that
// one word from one lane (relaxed agent loads, s_sleep between polls), joins its workgroup's barrier and reads the
// bytes with sc1 loads only (they bypass the CU's L1, so no acquire fence).  One workgroup per CU (the LDS request
// sees to that).  Every spin is bounded: a poller that gives up raises the launch's abort word, every other wait sees
// it, all workgroups drain, and the host redoes the factorisation with the launch schedule.  Flags are zeroed by a
// memset ahead of every launch.
//
// Every dependency of a queued job is either produced by the critical workgroup or by a job EARLIER in the queue, and
// a popped job is held by a resident workgroup, so the queue cannot deadlock while workgroup 0 makes progress; workgroup
// 0 in turn only waits for jobs whose inputs it has already published.
#include <algorithm>
#include <vector>

#include "b7_internal.h"
#include "gemm_f64.h"
#include "potrf_diag.h"

namespace {

using namespace b7diag;
using GT = GemmF64<64, 64, 64, 2, 2, false>;  // NT, all of K = 64 in one LDS stage (row stride 66 = DLD)
using GN = GemmF64<64, 64, 64, 2, 2, true>;   // NN: the inverse's partial products (potrf.hip's G64NN runs the same chain
                                              // in 32-deep stages)
static_assert(GT::STRIDE == DLD, "the trsm / look-ahead code reads GT's LDS image as [64][DLD]");

typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

constexpr int FLAG_HDR = 16;                 // words: [0] queue head, [1] abort code, rest padding
constexpr unsigned SPIN_LIMIT = 1u << 21;    // polls before a waiter gives up (>= 0.5 s)
constexpr int PERSIST_LDS_BYTES = (4 * NB * DLD + 32 * TLD) * 8;  // A, X, T, S1, S2 of the critical workgroup: 141 KiB
static_assert(PERSIST_LDS_BYTES >= 2 * GT::STAGE_DOUBLES * 8 && PERSIST_LDS_BYTES > 80 * 1024,
              "two update stages of a tile job; one workgroup per CU");

enum JobType { JOB_TILE = 0, JOB_PRE_SUB = 1, JOB_PRE_DIAG = 2, JOB_INV = 3, JOB_INV_DIAG = 4, JOB_ZERO = 5, JOB_VEC = 6 };

struct PArgs {
  const double *K;
  double *L, *Linv, *dinv;
  unsigned *flags;
  int *info;
  unsigned long long *stamps;  // nullable: [nb][8] critical-path stamps, then [njobs][4]: job start, end, cycles spent in
                               // trailing-update products (load + stage + MFMA + subtract), number of such products
  const int4 *jobs;
  int njobs, n, nb, nreal, with_inverse;
  double extra;
  // a batch of independent factorisations in one launch (blockIdx.y = fit): strides per fit, 0 for a single one
  int64_t sK, sL, sdinv, sLinv;
  int sflags, sinfo;
  // likelihood mode (b7_gp_nll_batch): no inverse; one JOB_VEC per fit solves L z = r alongside and leaves
  // terms[0] = |z|^2, terms[1] = sum log L_ii
  const double *resid;
  double *terms;
  int fault_panel;  // tests only (B7_PERSIST_FAULT=p): workgroup 0 never raises ready(p, p); -1 = off
};

// ---- flags ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ld_flag(const unsigned *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_flag(unsigned *p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct Flags {
  unsigned *base;
  int nb;
  __device__ unsigned *head() const { return base; }
  __device__ unsigned *abort_word() const { return base + 1; }
  __device__ unsigned *ready(int I, int J) const { return base + FLAG_HDR + I * nb + J; }
  __device__ unsigned *iready(int p, int j) const { return base + FLAG_HDR + nb * nb + p * nb + j; }
  __device__ unsigned *pre_sub(int p) const { return base + FLAG_HDR + 2 * nb * nb + p; }
  __device__ unsigned *pre_diag(int p) const { return base + FLAG_HDR + 2 * nb * nb + nb + p; }
};

// Lanes 0 and 1 of wave 0 poll one word each (pass the same word twice to wait for one), the workgroup learns the
// outcome through LDS.  false = abort (uniform): the caller returns.  ONE barrier per wait: the result goes to one of
// two LDS slots used alternately (a slot is rewritten two waits later, and every thread has read it before it arrives
// at the wait in between).  sh_ok: int[2]; `turn` is a per-thread counter that all threads advance together.
__device__ __forceinline__ bool wg_wait2(unsigned *f0, unsigned *f1, const Flags &F, int *info, int *sh_ok, int &turn,
                                         int code) {
  const int slot = turn & 1;
  turn += 1;
  if (threadIdx.x < 64) {
    unsigned *mine = (threadIdx.x & 1) ? f1 : f0;
    int ok = 1;
    unsigned spins = 0;
    for (;;) {
      const bool set = threadIdx.x > 1 || ld_flag(mine) != 0u;
      if (__all(set)) break;
      __builtin_amdgcn_s_sleep(1);
      ++spins;
      if ((spins & 255u) == 0u && ld_flag(F.abort_word()) != 0u) {
        ok = 0;
        break;
      }
      if (spins > SPIN_LIMIT) {  // give up: everybody drains, the host falls back to the launch schedule
        if (threadIdx.x == 0) {
          st_flag(F.abort_word(), (unsigned)code);
          __hip_atomic_store(info + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ok = 0;
        break;
      }
    }
    if (threadIdx.x == 0) sh_ok[slot] = ok;
  }
  __syncthreads();
  return sh_ok[slot] != 0;
}
// One look at two flags, no waiting: true when both are up.  Ends with a barrier, like wg_wait2.
__device__ __forceinline__ bool wg_test2(unsigned *f0, unsigned *f1, int *sh_ok, int &turn) {
  const int slot = turn & 1;
  turn += 1;
  if (threadIdx.x < 64) {
    unsigned *mine = (threadIdx.x & 1) ? f1 : f0;
    const bool set = threadIdx.x > 1 || ld_flag(mine) != 0u;
    const bool all = __all(set);
    if (threadIdx.x == 0) sh_ok[slot] = all ? 1 : 0;
  }
  __syncthreads();
  return sh_ok[slot] != 0;
}
__device__ __forceinline__ bool wg_wait(unsigned *flag, const Flags &F, int *info, int *sh_ok, int &turn, int code) {
  return wg_wait2(flag, flag, F, info, sh_ok, turn, code);
}

// Every storing wave drains, the workgroup meets, ONE lane raises the flag (Guideline 16, R1).
__device__ __forceinline__ void wg_publish(unsigned *flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) st_flag(flag, 1u);
}

// ---- 64x64 tiles: thread t owns the 16-byte chunks c = t + 256 i (row c >> 5, columns 2 (c & 31) ..) -----------------
struct Tile8 {
  d2_t v[8];
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const double *base) {
  const unsigned long long a = (unsigned long long)base;  // wave-uniform by construction: make that provable
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ d2_t ld16_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off) {
  const u4_t x = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);  // aux 16 = sc1: bypasses this CU's L1
  d2_t r;
  r[0] = __hiloint2double((int)x[1], (int)x[0]);
  r[1] = __hiloint2double((int)x[3], (int)x[2]);
  return r;
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off, d2_t v) {
  u4_t x;
  x[0] = (unsigned)__double2loint(v[0]);
  x[1] = (unsigned)__double2hiint(v[0]);
  x[2] = (unsigned)__double2loint(v[1]);
  x[3] = (unsigned)__double2hiint(v[1]);
  __builtin_amdgcn_raw_buffer_store_b128(x, rs, byte_off, 0, 16);  // write-through
}
__device__ __forceinline__ void tile_load_sc1(d2_t (&v)[8], const double *base, int ld) {
  const __amdgpu_buffer_rsrc_t rs = tile_rsrc(base);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + 256 * i;
    v[i] = ld16_sc1(rs, ((c >> 5) * ld + 2 * (c & 31)) * 8);
  }
}
__device__ __forceinline__ void tile_load_plain(d2_t (&v)[8], const double *base, int ld) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + 256 * i;
    v[i] = *reinterpret_cast<const d2_t *>(base + (int64_t)(c >> 5) * ld + 2 * (c & 31));
  }
}
__device__ __forceinline__ void tile_store_sc1(const d2_t (&v)[8], double *base, int ld) {
  const __amdgpu_buffer_rsrc_t rs = tile_rsrc(base);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + 256 * i;
    st16_sc1(rs, ((c >> 5) * ld + 2 * (c & 31)) * 8, v[i]);
  }
}
__device__ __forceinline__ void tile_store_plain(const d2_t (&v)[8], double *base, int ld) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + 256 * i;
    *reinterpret_cast<d2_t *>(base + (int64_t)(c >> 5) * ld + 2 * (c & 31)) = v[i];
  }
}
// registers <-> an LDS image [64][DLD] (16-byte aligned rows: DLD is even)
__device__ __forceinline__ void tile_to_lds(const d2_t (&v)[8], double *img) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + 256 * i;
    *reinterpret_cast<d2_t *>(img + (c >> 5) * DLD + 2 * (c & 31)) = v[i];
  }
}
__device__ __forceinline__ void tile_from_lds(d2_t (&v)[8], const double *img) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + 256 * i;
    v[i] = *reinterpret_cast<const d2_t *>(img + (c >> 5) * DLD + 2 * (c & 31));
  }
}

// the accumulator layout of GT / GN (both 2 x 2 waves of 2 x 2 MFMA tiles): element (i, j, r) of this lane
__device__ __forceinline__ int acc_row(int i, int r) { return GT::out_row(i, r); }
__device__ __forceinline__ int acc_col(int j) { return GT::out_col(j); }

// ---- the critical workgroup -------------------------------------------------------------------------------------------
// copy_lower_kernel's view of a diagonal tile of K: the tile as assembled, eps added to the first nreal diagonal entries
__device__ __forceinline__ void add_extra(d2_t (&v)[8], int tile, int nreal, double extra) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + 256 * i, row = c >> 5, col = 2 * (c & 31);
    if (tile * NB + row < nreal) {
      if (col == row) v[i][0] = v[i][0] + extra;
      if (col + 1 == row) v[i][1] = v[i][1] + extra;
    }
  }
}
// VAR 1 of the factor routine keeps I_16 in rows 0..15 x columns 48..63 of the block image
__device__ __forceinline__ void put_identity_corner(double *A) {
  const int t = threadIdx.x;
  A[(t >> 4) * DLD + 48 + (t & 15)] = ((t >> 4) == (t & 15)) ? 1.0 : 0.0;
}

// What waves 1..3 do inside the factor routine (diag_core's hook) while wave 0 is in its serial 16x16 steps.
// Step 0: the rest of the look-ahead into THIS block (finish_lookahead).  Steps 2, 3 and the routine's tail: fetching the
// NEXT panel's two tiles -- wave 3 brings tile (p+1, p) into S1 and waves 1 / 2 one half each of tile (p+1, p+1) into S2,
// if the helpers have already handed them over (one non-blocking look at the flag per attempt; the hand-over usually
// lands around step 3).  What was not fetched here is fetched after the factorisation the blocking way.
struct Prefetch {
  const PArgs &a;
  const Flags &F;
  double *S1, *S2;
  int *done;  // LDS: [0] tile below in S1, [1] / [2] upper / lower half of the next diagonal tile in S2
  int p;
  double *A;     // the block being factored: step 0's hook completes its lower sub-tiles right of column block 0
  int *pub_cnt;  // LDS: waves that have drained their stores of this round; the fourth raises ready(p, p-1)

  // Step 0 of panel p's factorisation, waves 1..3 (wave 0 is factoring sub-block 0 and solving column block 0):
  // the part of the look-ahead that wave 0 does not need yet.  (1) C(p, p) -= L[p][p-1] L[p][p-1]' on the six sub-tiles
  // (i, j), 1 <= j <= i <= 3, two per wave as interleaved 64-deep chains from zero, then the subtraction; (2) L[p][p-1]
  // goes out write-through; (3) drained, counted in LDS, and the last of the four waves raises its flag.
  __device__ __forceinline__ void finish_lookahead(int wave) const {
    const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4, n = a.n;
    // L[p][p-1] from its LDS image first (192 lanes): the write-through stores drain under the MFMAs below
    {
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(a.L + ((int64_t)p * NB) * n + (int64_t)(p - 1) * NB);
      const int t = threadIdx.x - 64;
#pragma unroll
      for (int i = 0; i < 11; ++i) {
        const int c = t + 192 * i;
        if (c < 2048) {
          const int row = c >> 5, col = 2 * (c & 31);
          st16_sc1(rs, (row * n + col) * 8, *reinterpret_cast<const d2_t *>(S1 + row * DLD + col));
        }
      }
    }
    // wave 1: (1,1) (2,1)   wave 2: (3,1) (2,2)   wave 3: (3,2) (3,3)
    const int s0 = wave == 1 ? 1 : 3, c0 = wave == 3 ? 2 : 1, s1 = wave == 3 ? 3 : 2, c1 = wave == 1 ? 1 : (wave == 2 ? 2 : 3);
    d4_t u0 = {0.0, 0.0, 0.0, 0.0}, u1 = {0.0, 0.0, 0.0, 0.0};
    const double *a0 = S1 + (16 * s0 + lr) * DLD + lq, *b0 = S1 + (16 * c0 + lr) * DLD + lq;
    const double *a1 = S1 + (16 * s1 + lr) * DLD + lq, *b1 = S1 + (16 * c1 + lr) * DLD + lq;
    {
      // the fragments of step k + 1 are requested between the MFMAs of step k (tools/mfma_feed_probe.hip: 94 -> 75 cycles per
      // MFMA with one wave per SIMD); same products, same order
      double f[2][4] = {{a0[0], b0[0], a1[0], b1[0]}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
      for (int k4 = 0; k4 < 16; ++k4) {
        const int cur = k4 & 1, nxt = cur ^ 1;
        u0 = mfma_f64(f[cur][0], f[cur][1], u0);
        __builtin_amdgcn_sched_barrier(0);
        if (k4 + 1 < 16) {
          f[nxt][0] = a0[4 * (k4 + 1)];
          f[nxt][1] = b0[4 * (k4 + 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
        u1 = mfma_f64(f[cur][2], f[cur][3], u1);
        __builtin_amdgcn_sched_barrier(0);
        if (k4 + 1 < 16) {
          f[nxt][2] = a1[4 * (k4 + 1)];
          f[nxt][3] = b1[4 * (k4 + 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int e0 = (16 * s0 + lq + 4 * rr) * DLD + 16 * c0 + lr, e1 = (16 * s1 + lq + 4 * rr) * DLD + 16 * c1 + lr;
      A[e0] = S2[e0] - u0[rr];
      A[e1] = S2[e1] - u1[rr];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // issued ~1 us ago
    if (lane == 0) {
      const int before = __hip_atomic_fetch_add(pub_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (before == 3) st_flag(F.ready(p, p - 1), 1u);
    }
  }
  __device__ __forceinline__ void rows(const double *src, bool plain, double *img, int row0, int nrows, int tile,
                                       bool diag) const {
    const int lane = threadIdx.x & 63, n = a.n;
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(src);
    for (int i0 = 0; i0 < nrows / 2; i0 += 16) {  // 16 chunks of 16 bytes in flight per lane
      d2_t v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * (i0 + i), row = row0 + (c >> 5), col = 2 * (c & 31);
        v[i] = plain ? *reinterpret_cast<const d2_t *>(src + (int64_t)row * n + col) : ld16_sc1(rs, (row * n + col) * 8);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * (i0 + i), row = row0 + (c >> 5), col = 2 * (c & 31);
        if (diag && plain && tile * NB + row < a.nreal) {  // C_0 of a diagonal tile read straight from K: + eps
          if (col == row) v[i][0] = v[i][0] + a.extra;
          if (col + 1 == row) v[i][1] = v[i][1] + a.extra;
        }
        *reinterpret_cast<d2_t *>(img + row * DLD + col) = v[i];
      }
    }
  }
  __device__ __forceinline__ void operator()(int kb, int wave) const {
    if (kb == 0) {
      if (p > 0) finish_lookahead(wave);
      return;
    }
    // panel 0's neighbours come straight from K (cold in HBM: the blocking path after the factorisation is faster)
    if (kb < 2 || p == 0 || p + 1 >= a.nb) return;  // kb = 2, 3 and the tail (4)
    const int n = a.n, lane = threadIdx.x & 63;
    if (wave == 3) {
      if (done[0] != 0 || ld_flag(F.pre_sub(p + 1)) == 0u) return;
      rows(a.L + ((int64_t)(p + 1) * NB) * n + (int64_t)p * NB, false, S1, 0, 64, p + 1, false);
      if (lane == 0) done[0] = 1;
    } else {
      if (done[wave] != 0 || ld_flag(F.pre_diag(p + 1)) == 0u) return;
      rows(a.L + ((int64_t)(p + 1) * NB) * n + (int64_t)(p + 1) * NB, false, S2, wave == 1 ? 0 : 32, 32, p + 1, true);
      if (lane == 0) done[wave] = 1;
    }
  }
};

__device__ void critical_path(const PArgs &a, const Flags &F, double *dsm, int *sh_ok, int *pf_done, int *pub_cnt) {
  int turn = 0;
  double *A = dsm, *X = dsm + NB * DLD, *T = X + NB * DLD, *S1 = T + 32 * TLD, *S2 = S1 + NB * DLD;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15,
            lq = lane >> 4;
  const int n = a.n, nb = a.nb;
  unsigned long long *st = a.stamps;
#define PST(p, i) \
  if (st && tid == 0) st[(p) * 8 + (i)] = __builtin_amdgcn_s_memtime()
  PST(nb - 1, 6);  // entry (the last panel has no look-ahead: its slots 3..7 are free)
  {
    d2_t v[8];
    tile_load_plain(v, a.K, n);
    add_extra(v, 0, a.nreal, a.extra);
    tile_to_lds(v, A);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i][0] = v[i][1] = 0.0;
    tile_to_lds(v, X);  // zero once: every round rewrites X's lower triangle in full and never touches the upper one
  }
  if (tid < 4) pf_done[tid] = 0;
  __syncthreads();
  put_identity_corner(A);
  __syncthreads();
  for (int p = 0; p < nb; ++p) {
    PST(p, 0);
    diag_core<1, false>(A, X, T, p, a.info, nullptr, Prefetch{a, F, S1, S2, pf_done, p, A, pub_cnt});
    PST(p, 1);
    const bool last = p + 1 == nb;
    const bool have_d1 = pf_done[0] != 0, have_d2 = pf_done[1] != 0 && pf_done[2] != 0;
    // inv(L_pp) first and by itself: it is what every helper of this panel waits for (nobody inside the launch reads L_pp),
    // and the sooner its flag is up the sooner the next panel's tiles come back.  The drain costs this workgroup ~0.5 us.
    {
      d2_t vx[8];
      tile_from_lds(vx, X);
      tile_store_sc1(vx, a.dinv + (int64_t)p * NB * NB, NB);
    }
#ifdef B7_DIAG
    if (p != a.fault_panel)  // tests only (B7_PERSIST_FAULT): withhold one flag to exercise the time-out path
#endif
      wg_publish(F.ready(p, p));
    PST(p, 2);
    d2_t d1[8], d2[8];
    // what the prefetch did not get (its latency overlaps the store of L_pp issued behind it)
    if (!last && p == 0) {
      tile_load_plain(d1, a.K + (int64_t)NB * n, n);
      tile_load_plain(d2, a.K + (int64_t)NB * n + NB, n);
      add_extra(d2, 1, a.nreal, a.extra);
    } else if (!last && (!have_d1 || !have_d2)) {
      unsigned *f1 = F.pre_sub(p + 1), *f2 = F.pre_diag(p + 1);
      if (!wg_wait2(have_d1 ? f2 : f1, have_d2 ? f1 : f2, F, a.info, sh_ok, turn, 100 + p)) return;
      if (!have_d1) tile_load_sc1(d1, a.L + ((int64_t)(p + 1) * NB) * n + (int64_t)p * NB, n);
      if (!have_d2) tile_load_sc1(d2, a.L + ((int64_t)(p + 1) * NB) * n + (int64_t)(p + 1) * NB, n);
    }
    {  // L_pp, upper triangle zeroed
      d2_t va[8];
      tile_from_lds(va, A);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i, row = c >> 5, col = 2 * (c & 31);
        if (col > row) va[i][0] = 0.0;
        if (col + 1 > row) va[i][1] = 0.0;
      }
      tile_store_sc1(va, a.L + ((int64_t)p * NB) * n + (int64_t)p * NB, n);
    }
    if (last) {
      PST(p, 7);  // exit
      if (st && tid == 0) st[p * 8 + 5] = __builtin_amdgcn_s_memrealtime();  // 100 MHz, the same counter on every CU
      break;
    }
    if (!have_d1) tile_to_lds(d1, S1);
    if (!have_d2) tile_to_lds(d2, S2);
    __syncthreads();  // also: everybody has read pf_done
    if (tid < 4) pf_done[tid] = 0;
    if (tid == 4) *pub_cnt = 0;
    PST(p, 3);
    // L[p+1][p] = C inv(L_pp)': wave w rows 16 w .., all four column blocks; the chain of potrf_trsm_kernel<NEAR>
    // (k ascending, whole 16-blocks above the diagonal of inv(L_pp) skipped)
    d4_t lqv[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    {
      // operands of step t+1 are read from LDS while step t's MFMAs run (the compiler otherwise waits for each
      // ds_read right in front of its MFMA: ~100 cycles per 64-cycle MFMA)
      const double *ap = S1 + (wave * 16 + lr) * DLD + lq, *xp = X + lr * DLD + lq;
      double aq = ap[0], xb[4];
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) xb[jb] = xp[jb * 16 * DLD];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int kq = t >> 2;
        double naq = 0.0, nxb[4] = {0.0, 0.0, 0.0, 0.0};
        if (t + 1 < 16) {
          naq = ap[4 * (t + 1)];
#pragma unroll
          for (int jb = (t + 1) >> 2; jb < 4; ++jb) nxb[jb] = xp[jb * 16 * DLD + 4 * (t + 1)];
        }
#pragma unroll
        for (int jb = kq; jb < 4; ++jb) lqv[jb] = mfma_f64(aq, xb[jb], lqv[jb]);
        aq = naq;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) xb[jb] = nxb[jb];
      }
    }
    __syncthreads();  // every wave is done reading S1
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) S1[(wave * 16 + lq + 4 * rr) * DLD + jb * 16 + lr] = lqv[jb][rr];
    __syncthreads();
    PST(p, 4);
    PST(p, 5);  // (slot kept for the stamp reader's layout)
    // The next factorisation's first step needs column block 0 of C(p+1, p+1) only (wave 0 factors sub-block (0, 0) and
    // solves the three below it in the same instruction stream): wave w applies the update to sub-tile (w, 0) now -- the
    // 64-deep chain from zero, then the subtraction (syrk_tile / NEAR update) -- and waves 1..3 do the other six lower
    // sub-tiles, the store of L[p+1][p] and its flag inside that first step, where they would otherwise idle
    // (Prefetch::finish_lookahead).  The six sub-tiles above the diagonal are never read by the factor routine.
    {
      d4_t u = {0.0, 0.0, 0.0, 0.0};
      const double *ar = S1 + (16 * wave + lr) * DLD + lq, *br = S1 + lr * DLD + lq;
      double f[2][2] = {{ar[0], br[0]}, {0.0, 0.0}};
#pragma unroll
      for (int k4 = 0; k4 < 16; ++k4) {  // the next step's two fragments requested behind this step's MFMA
        const int cur = k4 & 1, nxt = cur ^ 1;
        u = mfma_f64(f[cur][0], f[cur][1], u);
        __builtin_amdgcn_sched_barrier(0);
        if (k4 + 1 < 16) {
          f[nxt][0] = ar[4 * (k4 + 1)];
          f[nxt][1] = br[4 * (k4 + 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int e = (16 * wave + lq + 4 * rr) * DLD + lr;
        A[e] = S2[e] - u[rr];
      }
    }
    put_identity_corner(A);  // rows 0..15 x columns 48..63: above the diagonal, nobody else writes there
    PST(p, 6);
    if (wave == 0) {  // this wave's stores of L_pp / inv(L_pp) have long drained; it is the fourth party to the flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        const int before = __hip_atomic_fetch_add(pub_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (before == 3) st_flag(F.ready(p + 1, p), 1u);
      }
    }
    __syncthreads();  // column block 0 and the corner are in place
    PST(p, 7);
  }
#undef PST
}


__device__ __forceinline__ void acc_to_lds(const d4_t (&C)[2][2], double *img, int stride) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) img[acc_row(i, r) * stride + acc_col(j)] = C[i][j][r];
}

// tile (I, J) of the lower triangle, I > J (type JOB_TILE / JOB_PRE_SUB) or I == J (JOB_PRE_DIAG)
__device__ bool tile_job(const PArgs &a, const Flags &F, int type, int I, int J, double *sm, int *sh_ok, int &turn, int jid) {
  const int n = a.n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  d4_t C[2][2];
  {
    const double *k0 = a.K + ((int64_t)I * NB) * n + (int64_t)J * NB;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = acc_row(i, r), col = acc_col(j);
          double v = k0[(int64_t)row * n + col];
          if (I == J && row == col && I * NB + row < a.nreal) v = v + a.extra;
          C[i][j][r] = v;
        }
  }
  const int nupd = (type == JOB_PRE_DIAG) ? J - 1 : J;  // the critical workgroup applies panel J-1 to its own diagonal tile
  unsigned long long busy = 0;
  // C_{q+1} = C_q - L_Iq L_Jq', q ascending.  The two tiles of update q + 1 are requested BEFORE update q is computed whenever
  // their flags are already up (one look, no waiting: away from the front of the factorisation they always are), into the
  // other half of the staging area: the load latency (~1 us of a 2.9 us update) hides under the 64 MFMAs per wave.  At
  // the front the old order remains: wait, load, compute.  Same products, same subtractions, same order.
  GT::Regs r;
  const double *Li = a.L + ((int64_t)I * NB) * n, *Lj = a.L + ((int64_t)J * NB) * n;
  if (nupd > 0) {
    if (!wg_wait2(F.ready(I, 0), F.ready(J, 0), F, a.info, sh_ok, turn, 1000 + I * 64)) return false;
    tile_load_sc1(r.a, Li, n);
    tile_load_sc1(r.b, Lj, n);
  }
  for (int q = 0; q < nupd; ++q) {
    const unsigned long long t0 = a.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    double *st = sm + (q & 1) * GT::STAGE_DOUBLES;  // rewritten at q + 2, behind the barrier of q + 1
    GT::store_lds(r, st);
    bool ahead = false;
    if (q + 1 < nupd) {
      ahead = wg_test2(F.ready(I, q + 1), F.ready(J, q + 1), sh_ok, turn);  // its barrier also publishes the stage
      if (ahead) {
        tile_load_sc1(r.a, Li + (int64_t)(q + 1) * NB, n);
        tile_load_sc1(r.b, Lj + (int64_t)(q + 1) * NB, n);
      }
    } else {
      __syncthreads();
    }
    d4_t P[2][2] = {};
    GT::compute_stage(st, P);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) C[i][j] = C[i][j] - P[i][j];
    if (a.stamps) busy += __builtin_amdgcn_s_memtime() - t0;
    if (q + 1 < nupd && !ahead) {
      if (!wg_wait2(F.ready(I, q + 1), F.ready(J, q + 1), F, a.info, sh_ok, turn, 1000 + I * 64 + q + 1)) return false;
      tile_load_sc1(r.a, Li + (int64_t)(q + 1) * NB, n);
      tile_load_sc1(r.b, Lj + (int64_t)(q + 1) * NB, n);
    }
  }
  __syncthreads();  // everybody is done with the staging area: the epilogue below reuses it
  if (a.stamps && tid == 0) {
    a.stamps[a.nb * 8 + jid * 4 + 2] = busy;
    a.stamps[a.nb * 8 + jid * 4 + 3] = (unsigned long long)nupd;
  }
  double *Ai = sm, *Xp = sm + NB * DLD;
  double *dst = a.L + ((int64_t)I * NB) * n + (int64_t)J * NB;
  acc_to_lds(C, Ai, DLD);
  if (type == JOB_TILE) {
    if (!wg_wait(F.ready(J, J), F, a.info, sh_ok, turn, 5000 + J)) return false;
    d2_t vx[8];
    tile_load_sc1(vx, a.dinv + (int64_t)J * NB * NB, NB);
    tile_to_lds(vx, Xp);
    __syncthreads();
    // L_IJ = C inv(L_JJ)': sub-tile (slab s, column block cb) is a chain over the k-blocks 0 .. cb (those above the
    // diagonal of inv(L_JJ) are skipped), i.e. 4 (cb + 1) MFMAs; the sixteen sub-tiles are dealt so that every wave
    // issues 40:  wave 0: (0,3) (1,3) (0,1)   wave 1: (2,3) (3,3) (1,1)   wave 2: (0,2) (1,2) (2,2) (0,0)
    // wave 3: (3,2) (2,1) (3,1) (1,0) (2,0) (3,0)
    constexpr int TS_N[4] = {3, 3, 4, 6};
    constexpr int TS_S[4][6] = {{0, 1, 0, 0, 0, 0}, {2, 3, 1, 0, 0, 0}, {0, 1, 2, 0, 0, 0}, {3, 2, 3, 1, 2, 3}};
    constexpr int TS_C[4][6] = {{3, 3, 1, 0, 0, 0}, {3, 3, 1, 0, 0, 0}, {2, 2, 2, 0, 0, 0}, {2, 1, 1, 0, 0, 0}};
    d4_t li[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      li[t] = d4_t{0.0, 0.0, 0.0, 0.0};
      if (t >= TS_N[wave]) continue;
      const int sl = TS_S[wave][t], cb = TS_C[wave][t];
      const double *ar = Ai + (16 * sl + lr) * DLD + lq, *xr = Xp + (16 * cb + lr) * DLD + lq;
      for (int k4 = 0; k4 < 4 * (cb + 1); ++k4) li[t] = mfma_f64(ar[4 * k4], xr[4 * k4], li[t]);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      if (t >= TS_N[wave]) continue;
      const int sl = TS_S[wave][t], cb = TS_C[wave][t];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Ai[(16 * sl + lq + 4 * rr) * DLD + cb * 16 + lr] = li[t][rr];
    }
  }
  __syncthreads();
  d2_t vo[8];
  tile_from_lds(vo, Ai);
  tile_store_sc1(vo, dst, n);
  wg_publish(type == JOB_TILE ? F.ready(I, J) : (type == JOB_PRE_SUB ? F.pre_sub(I) : F.pre_diag(I)));
  __syncthreads();  // the staging area is free for the next job
  return true;
}

// tile (p, j), j < p, of inv(L)
__device__ bool inv_job(const PArgs &a, const Flags &F, int p, int j, double *sm, int *sh_ok, int &turn) {
  const int n = a.n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  d4_t tot[2][2] = {}, cur[2][2] = {};
  // Products L[p][t] * inv(L)[t][j], t = j .. p-1, in 128-deep chunks summed in ascending order.  As in tile_job, the pair of
  // tiles of step t + 1 is requested before step t is computed whenever its flags are already up (one look, no waiting),
  // into the other half of the staging area.
  GN::Regs r;
  auto flags_of = [&](int t, unsigned *&f0, unsigned *&f1) {
    f0 = F.ready(p, t);
    f1 = t == j ? F.ready(j, j) : F.iready(t, j);
  };
  auto request = [&](int t) {
    const double *Lpt = a.L + ((int64_t)p * NB) * n + (int64_t)t * NB;
    // rows [64 t, 64 t + 64) x columns [64 j, ..) of inv(L); the diagonal tile is inv(L_jj) itself
    const double *Btile = (t == j) ? a.dinv + (int64_t)j * NB * NB : a.Linv + ((int64_t)t * NB) * n + (int64_t)j * NB;
    const int ldb = (t == j) ? NB : n;
    const __amdgpu_buffer_rsrc_t ra = tile_rsrc(Lpt), rb = tile_rsrc(Btile);
#pragma unroll
    for (int i = 0; i < GN::A_PER_T; ++i) {
      const int c = tid + i * 256, row = c / 32, kc = c % 32;
      r.a[i] = ld16_sc1(ra, (row * n + 2 * kc) * 8);
    }
#pragma unroll
    for (int i = 0; i < GN::B_PER_T; ++i) {
      const int c = tid + i * 256, kr = c / 32, nc = c % 32;
      r.b[i] = ld16_sc1(rb, (kr * ldb + 2 * nc) * 8);
    }
  };
  if (j < p) {
    unsigned *f0, *f1;
    flags_of(j, f0, f1);
    if (!wg_wait2(f0, f1, F, a.info, sh_ok, turn, 10000 + p * 64 + j)) return false;
    request(j);
  }
  for (int t = j; t < p; ++t) {
    double *st = sm + ((t - j) & 1) * GN::STAGE_DOUBLES;  // rewritten two steps on, behind the barrier of the next step
    GN::store_lds(r, st);  // the whole 64-deep tile pair in one LDS stage: the same k-ascending chain, one round trip
    bool ahead = false;
    unsigned *f0 = nullptr, *f1 = nullptr;
    if (t + 1 < p) {
      flags_of(t + 1, f0, f1);
      ahead = wg_test2(f0, f1, sh_ok, turn);  // its barrier also publishes the stage
      if (ahead) request(t + 1);
    } else {
      __syncthreads();
    }
    GN::compute_stage(st, cur);  // the chain runs on through a chunk's second tile
    if ((t & 1) == 1 || t == p - 1) {  // end of the 128-deep chunk t / 2: partials are summed in ascending chunk order
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          tot[i][jj] = tot[i][jj] + cur[i][jj];
          cur[i][jj] = d4_t{0.0, 0.0, 0.0, 0.0};
        }
    }
    if (t + 1 < p && !ahead) {
      if (!wg_wait2(f0, f1, F, a.info, sh_ok, turn, 10000 + p * 64 + t + 1)) return false;
      request(t + 1);
    }
  }
  __syncthreads();  // everybody is done with the staging area: the epilogue below reuses it
  if (!wg_wait(F.ready(p, p), F, a.info, sh_ok, turn, 30000 + p)) return false;
  constexpr int SLD = 65;  // odd stride: the B-operand reads (k = lane >> 4, n = lane & 15) stay conflict-free
  double *Dn = sm, *Ts = sm + NB * DLD;
  {
    d2_t vd[8];
    tile_load_sc1(vd, a.dinv + (int64_t)p * NB * NB, NB);
    tile_to_lds(vd, Dn);
  }
  acc_to_lds(tot, Ts, SLD);
  __syncthreads();
  // out = -inv(L_pp) * sum: wave w owns rows 16 w ..; two accumulators alternate (potrf_trsm_kernel's inverse rows)
  d4_t outv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    d4_t c0a = {0.0, 0.0, 0.0, 0.0}, c1a = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      if (kq > wave) break;
#pragma unroll
      for (int s4 = 0; s4 < 4; s4 += 2) {
        c0a = mfma_f64(-Dn[(wave * 16 + lr) * DLD + kq * 16 + 4 * s4 + lq], Ts[(kq * 16 + 4 * s4 + lq) * SLD + 16 * s + lr], c0a);
        c1a = mfma_f64(-Dn[(wave * 16 + lr) * DLD + kq * 16 + 4 * s4 + 4 + lq],
                       Ts[(kq * 16 + 4 * s4 + 4 + lq) * SLD + 16 * s + lr], c1a);
      }
    }
    outv[s] = c0a + c1a;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Dn[(wave * 16 + lq + 4 * rr) * DLD + 16 * s + lr] = outv[s][rr];
  __syncthreads();
  d2_t vo[8];
  tile_from_lds(vo, Dn);
  tile_store_sc1(vo, a.Linv + ((int64_t)p * NB) * n + (int64_t)j * NB, n);
  wg_publish(F.iready(p, j));
  __syncthreads();
  return true;
}

// L z = r by forward substitution, block row by block row as the tiles of L are published; |z|^2 and sum log L_ii are
// all the likelihood needs (no inverse, no alpha).  One workgroup per fit; thread t: row t >> 2, a quarter of the columns.
__device__ bool vec_job(const PArgs &a, const Flags &F, double *sm, int *sh_ok, int &turn, unsigned long long *st) {
  const int n = a.n, nb = a.nb, tid = threadIdx.x, row = tid >> 2, part = tid & 3;
  unsigned long long t_wait = 0, t_work = 0, t0 = 0;  // diagnostics only (st != nullptr)
  double *z = sm, *accs = sm + 64 * 64, *red = accs + 64, *img0 = red + 512, *img1 = img0 + NB * DLD, *dsave = img1 + NB * DLD;
  static_assert((2 * 64 * 64 + 64 + 512 + 2 * NB * DLD) * 8 <= PERSIST_LDS_BYTES, "vec_job: z, accs, red and two tile images");
  double ssq = 0.0, logdet = 0.0;
  for (int p = 0; p < nb; ++p) {
    double acc = 0.0;
    // Row p's tiles L[p][0 .. p-1], one after the other.  Each is loaded the way the tile jobs load (one 16-byte chunk
    // per lane, a wave reads two whole rows: the earlier form had every lane read its own 16 columns, 64 scattered
    // 16-byte requests per instruction, 25 GB/s, and the chain trailed the factorisation by 0.45 ms at N = 2048),
    // staged through an LDS image, and summed from there in the order it always had.  The next tile's loads are issued
    // before this one is consumed; a flag is awaited for up to four tiles at once (tile (p, q + 1) is computed from
    // tile (p, q), so its flag implies the earlier ones), the last tile of the row -- the late one -- by itself.
    d2_t r0[8], r1[8], r2[8];  // a ring of three tiles in flight: the loads of tile q + 2 are issued before tile q is consumed
    int seen = -1;             // tiles <= seen have had their flag observed
    auto await = [&](int q) -> bool {
      if (q <= seen) return true;
      const int last = q >= p - 1 ? p - 1 : (q + 3 < p - 2 ? q + 3 : p - 2);
      if (st) t0 = __builtin_amdgcn_s_memtime();
      if (!wg_wait(F.ready(p, last), F, a.info, sh_ok, turn, 50000 + p * 64 + q)) return false;
      if (st) t_wait += __builtin_amdgcn_s_memtime() - t0;
      seen = last;
      return true;
    };
    auto issue = [&](int q, d2_t (&v)[8]) -> bool {
      if (q >= p) return true;
      if (!await(q)) return false;
      tile_load_sc1(v, a.L + ((int64_t)p * NB) * n + (int64_t)q * NB, n);
      return true;
    };
    auto consume = [&](int q, const d2_t (&v)[8]) {
      if (st) t0 = __builtin_amdgcn_s_memtime();
      double *img = (q & 1) ? img1 : img0;  // two images: the barrier of tile q + 1 is what frees the image of tile q
      tile_to_lds(v, img);
      __syncthreads();
      const double *tr = img + row * DLD + 16 * part, *zq = z + q * 64 + 16 * part;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc += tr[2 * i] * zq[2 * i];
        acc += tr[2 * i + 1] * zq[2 * i + 1];
      }
      if (st) t_work += __builtin_amdgcn_s_memtime() - t0;
    };
    if (!issue(0, r0) || !issue(1, r1)) return false;
    for (int q = 0; q < p; q += 3) {
      if (!issue(q + 2, r2)) return false;
      consume(q, r0);
      if (q + 1 < p) {
        if (!issue(q + 3, r0)) return false;
        consume(q + 1, r1);
      }
      if (q + 2 < p) {
        if (!issue(q + 4, r1)) return false;
        consume(q + 2, r2);
      }
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (part == 0) accs[row] = a.resid[p * 64 + row] - acc;
    if (st) t0 = __builtin_amdgcn_s_memtime();
    if (!wg_wait(F.ready(p, p), F, a.info, sh_ok, turn, 60000 + p)) return false;  // its barrier publishes accs
    if (st) t_wait += __builtin_amdgcn_s_memtime() - t0;
    double sz = 0.0, dii = 0.0;
    {
      d2_t v[8];
      tile_load_sc1(v, a.dinv + (int64_t)p * NB * NB, NB);  // coalesced and through an image, as the row's tiles (the
      tile_to_lds(v, img0);                                  // barrier inside the wait above has freed both images)
      __syncthreads();
      const double *tr = img0 + row * DLD + 16 * part;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sz += tr[2 * i] * accs[16 * part + 2 * i];
        sz += tr[2 * i + 1] * accs[16 * part + 2 * i + 1];
      }
      dii = img0[row * DLD + row];
    }
    sz += __shfl_xor(sz, 1);
    sz += __shfl_xor(sz, 2);
    if (part == 0) {
      z[p * 64 + row] = sz;
      ssq += sz * sz;
      dsave[p * 64 + row] = dii;  // log L_ii = -log inv(L_pp)_ii (the diagonal of a triangular inverse is the reciprocal
                                  // diagonal); the logarithms are taken after the chain, all at once
    }
    __syncthreads();
  }
  for (int e = tid; e < nb * 64; e += 256) logdet -= log(dsave[e]);  // fixed assignment of terms to threads: same bits every run
  red[tid] = part == 0 ? ssq : 0.0;
  red[256 + tid] = logdet;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      red[tid] += red[tid + o];
      red[256 + tid] += red[256 + tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    a.terms[0] = red[0];
    a.terms[1] = red[256];
    if (st) st[2] = t_wait, st[3] = t_work;
  }
  return true;
}

__global__ void __launch_bounds__(256) potrf_persist_kernel(PArgs a) {
  extern __shared__ __align__(16) double dsm[];
  __shared__ int sh_ok[2], sh_job, pf_done[4], pub_cnt;
  {
    const int64_t b = blockIdx.y;  // one factorisation per grid row
    a.K += b * a.sK;
    a.L += b * a.sL;
    a.dinv += b * a.sdinv;
    if (a.Linv) a.Linv += b * a.sLinv;
    a.flags += b * a.sflags;
    a.info += b * a.sinfo;
    if (a.resid) {
      a.resid += b * a.n;
      a.terms += 2 * b;
    }
  }
  const Flags F{a.flags, a.nb};
  if (blockIdx.x == 0) {
    critical_path(a, F, dsm, sh_ok, pf_done, &pub_cnt);
    return;
  }
  const int n = a.n;
  int turn = 0;
  for (;;) {
    if (threadIdx.x == 0)
      sh_job = (int)__hip_atomic_fetch_add(F.head(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int jid = __builtin_amdgcn_readfirstlane(sh_job);
    __syncthreads();
    if (jid >= a.njobs) return;
    if (ld_flag(F.abort_word()) != 0u) return;  // uniform: every lane reads the same word
    const int4 job = a.jobs[jid];
    const int type = __builtin_amdgcn_readfirstlane(job.x), I = __builtin_amdgcn_readfirstlane(job.y),
              J = __builtin_amdgcn_readfirstlane(job.z);
    if (a.stamps && threadIdx.x == 0) a.stamps[a.nb * 8 + jid * 4] = __builtin_amdgcn_s_memtime();
    bool ok = true;
    if (type == JOB_TILE || type == JOB_PRE_SUB || type == JOB_PRE_DIAG) {
      ok = tile_job(a, F, type, I, J, dsm, sh_ok, turn, jid);
    } else if (type == JOB_INV) {
      ok = inv_job(a, F, I, J, dsm, sh_ok, turn);
    } else if (type == JOB_VEC) {
      ok = vec_job(a, F, dsm, sh_ok, turn, a.stamps ? a.stamps + a.nb * 8 + jid * 4 : nullptr);
    } else if (type == JOB_INV_DIAG) {  // inv(L)[p][p] = inv(L_pp): nobody inside the launch reads it
      ok = wg_wait(F.ready(I, I), F, a.info, sh_ok, turn, 40000 + I);
      if (ok) {
        d2_t v[8];
        tile_load_sc1(v, a.dinv + (int64_t)I * NB * NB, NB);
        tile_store_plain(v, a.Linv + ((int64_t)I * NB) * n + (int64_t)I * NB, n);
      }
    } else {  // JOB_ZERO: a structurally zero tile above the diagonal, in L and in inv(L)
      d2_t z[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) z[i][0] = z[i][1] = 0.0;
      tile_store_plain(z, a.L + ((int64_t)I * NB) * n + (int64_t)J * NB, n);
      if (a.with_inverse) tile_store_plain(z, a.Linv + ((int64_t)I * NB) * n + (int64_t)J * NB, n);
    }
    if (a.stamps && threadIdx.x == 0) a.stamps[a.nb * 8 + jid * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    if (!ok) return;
  }
}

struct HostJob {
  int type, I, J;
  double key;
};

}  // namespace

// mode 0: L, dinv; 1: + inv(L); 2: likelihood terms only (JOB_VEC), no inverse and no zero fill
static int persist_jobs(b7_ctx *c, int nb, int mode, const int4 **jobs_dev, int *njobs) {
  const int key = nb * 4 + mode;
  auto it = c->pjobs_cache.find(key);
  if (it == c->pjobs_cache.end()) {
    // the queue: sorted by the panel at whose end a job can finish; every dependency of a job is produced by workgroup 0
    // or sits earlier in this order
    std::vector<HostJob> jobs;
    const double lead = nb <= 32 ? 0.045 : 1.4 / (nb - 1);
    for (int J = 0; J < nb; ++J)
      for (int I = J + 2; I < nb; ++I) jobs.push_back({JOB_TILE, I, J, J + 0.001 * (I - J)});
    for (int p = 2; p < nb; ++p) jobs.push_back({JOB_PRE_SUB, p, p - 1, (p - 1) - 0.6});
    for (int p = 2; p < nb; ++p) jobs.push_back({JOB_PRE_DIAG, p, p, (p - 1) - 0.55});
    if (mode == 1) {
      for (int p = 1; p < nb; ++p)
        // a tile of row p with many products (small j) is popped up to ~1.4 panels ahead of the light ones: popped late
        // it would still be catching up on its p - j products when the factorisation is over.  lead (p - j) < 1.5 keeps
        // every L[p][t], t < p (key < p - 1 + 0.001 nb) ahead of it in the queue: 0.045 up to 32 panels, less beyond
        for (int j = 0; j < p; ++j) jobs.push_back({JOB_INV, p, j, p + 0.5 - lead * (p - j)});
      for (int p = 0; p < nb; ++p) jobs.push_back({JOB_INV_DIAG, p, p, p + 0.9});
    }
    if (mode == 2) {
      jobs.push_back({JOB_VEC, 0, 0, -2.0});  // first: it is resident from the start and follows the panels
    } else {
      for (int I = 0; I < nb; ++I)
        for (int J = I + 1; J < nb; ++J) jobs.push_back({JOB_ZERO, I, J, -1.0});
    }
    std::stable_sort(jobs.begin(), jobs.end(), [](const HostJob &x, const HostJob &y) { return x.key < y.key; });
    std::vector<int4> packed(jobs.size());
    for (size_t i = 0; i < jobs.size(); ++i) packed[i] = make_int4(jobs[i].type, jobs[i].I, jobs[i].J, 0);
    b7_ctx::JobList jl;
    B7_TRY(b7_ensure(c, jl.buf, sizeof(int4) * (packed.size() + 1)));
    B7_HIP(c, hipMemcpy(jl.buf.p, packed.data(), sizeof(int4) * packed.size(), hipMemcpyHostToDevice));
    jl.n = (int)packed.size();
    it = c->pjobs_cache.emplace(key, jl).first;
  }
  *jobs_dev = (const int4 *)it->second.buf.p;
  *njobs = it->second.n;
  return B7_OK;
}

static size_t persist_flag_words(int nb) { return (size_t)round_up(FLAG_HDR + 2 * nb * nb + 2 * nb, 4); }

// B independent factorisations of n x n matrices (strides in PArgs) in ONE launch: grid (1 + helpers, B).  Flags and
// info of all B fits are zeroed here.  The caller guarantees B * (1 + helpers) <= CUs (every workgroup resident).
static int persist_launch(b7_ctx *c, PArgs a, int B, int mode, int helpers) {
  if (!c->persist_attr_set) {
    B7_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_persist_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, PERSIST_LDS_BYTES));
    c->persist_attr_set = true;
  }
  B7_TRY(persist_jobs(c, a.nb, mode, &a.jobs, &a.njobs));
  if (reinterpret_cast<char *>(a.flags) == reinterpret_cast<char *>(a.info) + B7_INFO_HEAD_BYTES && B == 1) {
    // the single fit: info, likelihood terms and flags are one block (b7_internal.h) -> one memset node
    B7_HIP(c, hipMemsetAsync(a.info, 0, B7_INFO_HEAD_BYTES + sizeof(unsigned) * persist_flag_words(a.nb), c->stream));
  } else {
    const size_t flag_bytes = sizeof(unsigned) * (B > 1 ? (size_t)a.sflags * B : persist_flag_words(a.nb));
    const size_t info_bytes = sizeof(int) * (B > 1 ? (size_t)a.sinfo * B : 4);
    if (reinterpret_cast<char *>(a.flags) == reinterpret_cast<char *>(a.info) + info_bytes) {
      // the caller laid the reports of this launch's fits right in front of their flags (b7_gp_nll_batch): one memset node
      B7_HIP(c, hipMemsetAsync(a.info, 0, info_bytes + flag_bytes, c->stream));
    } else {
      B7_HIP(c, hipMemsetAsync(a.flags, 0, flag_bytes, c->stream));
      B7_HIP(c, hipMemsetAsync(a.info, 0, info_bytes, c->stream));
    }
  }
  a.with_inverse = mode == 1 ? 1 : 0;
  hipLaunchKernelGGL(potrf_persist_kernel, dim3(1 + helpers, B), dim3(256), PERSIST_LDS_BYTES, c->stream, a);
  B7_HIP(c, hipGetLastError());
  return B7_OK;
}

// K + extra*I -> L, dinv, info (+ Linv) in one persistent launch.  The caller has checked Npad <= B7_PERSIST_NMAX.
int launch_potrf_persist(b7_ctx *c, double extra, bool with_inverse) {
  PhaseScope ps(c, "potrf");
  const int n = c->Npad, nb = n / NB, mode = with_inverse ? 1 : 0;
  c->linv_done = false;
  static_assert(FLAG_HDR + 2 * 64 * 64 + 2 * 64 == B7_PERSIST_FLAG_WORDS_MAX && B7_PERSIST_NMAX == 64 * NB, "flag block of the largest persistent shape");
  B7_TRY(b7_ensure(c, c->info, B7_INFO_BYTES));
  PArgs a = {};
  a.fault_panel = c->persist_fault;
  a.K = (const double *)c->K.p;
  a.L = (double *)c->L.p;
  a.Linv = with_inverse ? (double *)c->Linv.p : nullptr;
  a.dinv = (double *)c->dinv.p;
  a.flags = reinterpret_cast<unsigned *>(static_cast<char *>(c->info.p) + B7_INFO_HEAD_BYTES);
  a.info = (int *)c->info.p;
  a.n = n;
  a.nb = nb;
  a.nreal = c->N;
  a.extra = extra;
  const int4 *jobs = nullptr;
  int njobs = 0;
  B7_TRY(persist_jobs(c, nb, mode, &jobs, &njobs));
  c->pjobs_nb = nb;
  c->pjobs_n = njobs;
  if (c->persist_stamps) {
    const size_t words = (size_t)nb * 8 + (size_t)njobs * 4;
    B7_TRY(b7_ensure(c, c->pstamps, sizeof(unsigned long long) * words));
    B7_HIP(c, hipMemsetAsync(c->pstamps.p, 0, sizeof(unsigned long long) * words, c->stream));
    a.stamps = (unsigned long long *)c->pstamps.p;
  }
  // one workgroup per CU (the LDS request guarantees it); all of them must be resident at once, so never more than CUs
  int helpers = njobs < c->cus - 1 ? njobs : c->cus - 1;
  if (c->persist_helpers > 0 && c->persist_helpers < helpers) helpers = c->persist_helpers;
  B7_TRY(persist_launch(c, a, 1, mode, helpers));
  c->linv_done = with_inverse;
  return B7_OK;
}

// How many fits to run side by side in one persistent launch.  A fit takes max(its dependent chain, its trailing work / its
// helpers), the chain grows with the number of panels nb and the work with nb^3, so the helpers a fit needs to stay on its
// chain grow with nb.  Swept on MI355X (16 fits, all in one launch vs 8 / 4 / 2 / 1 per launch): with the inverse nb helpers
// per fit are plenty (N = 2048: 8 fits per launch 3.24 ms, 16: 3.38, 4: 3.57; N = 1024: 16 per launch 0.50 ms, 8: 0.61),
// without it nb / 2 (N = 2048, 16 likelihoods: 2.11 ms in one launch, 2.51 in two); never fewer than three.  The rule
// below asks for a little less so that the common batch sizes (10, 16) are not split 15 + 1; batches that do not fit are
// cut into equal launches.
static int fits_per_launch(const b7_ctx *c, int nb, bool with_inverse) {
  int h = with_inverse ? 3 * nb / 4 : nb / 2 - 1;
  if (h < 3) h = 3;
  const int p = c->cus / (1 + h);
  return p < 1 ? 1 : p;
}

int launch_nll_batch(b7_ctx *c, int B, const double *K, double *L, double *dinv, unsigned *flags, int *info,
                     const double *resid, double *terms, const double *extra_per_fit_unused) {
  (void)extra_per_fit_unused;
  PhaseScope ps(c, "potrf");
  const int n = c->Npad, nb = n / NB;
  const int4 *jobs = nullptr;
  int njobs = 0;
  B7_TRY(persist_jobs(c, nb, 2, &jobs, &njobs));
  const int64_t nn = (int64_t)n * n;
  const int fw = (int)persist_flag_words(nb);
  // fits side by side per launch (fits_per_launch): sixteen likelihoods at N = 1024 are one launch, not sixteen
  const int max_per_launch = fits_per_launch(c, nb, false);
  const int nlaunch = (B + max_per_launch - 1) / max_per_launch, even = (B + nlaunch - 1) / nlaunch;
  for (int b0 = 0, nb_here = 0; b0 < B; b0 += nb_here) {
    nb_here = B - b0 < even ? B - b0 : even;
    int helpers = c->cus / nb_here - 1;
    if (helpers > njobs) helpers = njobs;
    PArgs a = {};
    a.fault_panel = c->persist_fault;  // -1 outside the diagnostic build
    a.K = K + b0 * nn;
    a.L = L + b0 * nn;
    a.dinv = dinv + (int64_t)b0 * n * NB;
    a.flags = flags + (int64_t)b0 * fw;
    a.info = info + b0 * 4;
    a.n = n;
    a.nb = nb;
    a.nreal = c->N;
    a.extra = 0.0;
    a.sK = nn;
    a.sL = nn;
    a.sdinv = (int64_t)n * NB;
    a.sflags = fw;
    a.sinfo = 4;
    a.resid = resid + (int64_t)b0 * n;
    a.terms = terms + 2 * b0;
    if (c->persist_stamps && B == 1) {  // diagnostics (tools/persist_stamps.py N nll)
      const size_t words = (size_t)nb * 8 + (size_t)njobs * 4;
      B7_TRY(b7_ensure(c, c->pstamps, sizeof(unsigned long long) * words));
      B7_HIP(c, hipMemsetAsync(c->pstamps.p, 0, sizeof(unsigned long long) * words, c->stream));
      a.stamps = (unsigned long long *)c->pstamps.p;
      c->pjobs_nb = nb;
      c->pjobs_n = njobs;
    }
    B7_TRY(persist_launch(c, a, nb_here, 2, helpers));
  }
  return B7_OK;
}

// B complete fits (Cholesky + inverse) of B matrices in as few persistent launches as the chip holds: the hyper samples of one
// nomination (b7_eval_nominate).  Every fit gets its own critical workgroup and an equal share of helpers; with B fits
// side by side the helpers of one fit are few, but a single fit leaves most of them idle anyway (its dependent chain is what
// takes the time), so B fits cost little more than one.  Same arithmetic per tile as the single launch, whatever the helper count.
int launch_fit_batch(b7_ctx *c, int B, const double *K, double *L, double *Linv, double *dinv, unsigned *flags, int *info) {
  PhaseScope ps(c, "potrf");
  const int n = c->Npad, nb = n / NB;
  const int4 *jobs = nullptr;
  int njobs = 0;
  B7_TRY(persist_jobs(c, nb, 1, &jobs, &njobs));
  const int64_t nn = (int64_t)n * n;
  const int fw = (int)persist_flag_words(nb);
  const int max_per_launch = fits_per_launch(c, nb, true);
  const int nlaunch = (B + max_per_launch - 1) / max_per_launch, even = (B + nlaunch - 1) / nlaunch;
  for (int b0 = 0; b0 < B;) {
    const int left = B - b0, nb_here = left < even ? left : even;
    int helpers = c->cus / nb_here - 1;
    if (helpers > njobs) helpers = njobs;
    PArgs a = {};
    a.fault_panel = c->persist_fault;
    a.K = K + b0 * nn;
    a.L = L + b0 * nn;
    a.Linv = Linv + b0 * nn;
    a.dinv = dinv + (int64_t)b0 * n * NB;
    a.flags = flags + (int64_t)b0 * fw;
    a.info = info + b0 * 4;
    a.n = n;
    a.nb = nb;
    a.nreal = c->N;
    a.extra = 0.0;
    a.sK = nn;
    a.sL = nn;
    a.sLinv = nn;
    a.sdinv = (int64_t)n * NB;
    a.sflags = fw;
    a.sinfo = 4;
    B7_TRY(persist_launch(c, a, nb_here, 1, helpers));
    b0 += nb_here;
  }
  return B7_OK;
}

// one fit of the batch again with eps on the diagonal (the jitter schedule of a fit whose plain attempt failed)
int launch_nll_one(b7_ctx *c, const double *K, double *L, double *dinv, unsigned *flags, int *info, const double *resid,
                   double *terms, double extra) {
  const int n = c->Npad, nb = n / NB;
  const int4 *jobs = nullptr;
  int njobs = 0;
  B7_TRY(persist_jobs(c, nb, 2, &jobs, &njobs));
  PArgs a = {};
  a.fault_panel = c->persist_fault;  // -1 outside the diagnostic build
  a.K = K;
  a.L = L;
  a.dinv = dinv;
  a.flags = flags;
  a.info = info;
  a.n = n;
  a.nb = nb;
  a.nreal = c->N;
  a.extra = extra;
  a.resid = resid;
  a.terms = terms;
  return persist_launch(c, a, 1, 2, njobs < c->cus - 1 ? njobs : c->cus - 1);
}

size_t persist_flag_words_host(int nb) { return persist_flag_words(nb); }

// diagnostics (tools/persist_stamps.py; diagnostic build only): the stamps of the last persistent launch, [nb][8] then [njobs][4]
#ifdef B7_DIAG
extern "C" int b7dbg_persist_stamps(b7_ctx *c, unsigned long long *out, int max_words, int *nb_out, int *njobs_out) {
  if (!c->pstamps.p) return B7_ERR_STATE;
  const int words = c->pjobs_nb * 8 + c->pjobs_n * 4;
  if (nb_out) *nb_out = c->pjobs_nb;
  if (njobs_out) *njobs_out = c->pjobs_n;
  B7_HIP(c, hipStreamSynchronize(c->stream));
  B7_HIP(c, hipMemcpy(out, c->pstamps.p, sizeof(unsigned long long) * (words < max_words ? words : max_words),
                      hipMemcpyDeviceToHost));
  return B7_OK;
}
#endif

// persistent launches of this context that timed out on a hand-off and were redone by the launch schedule (include/bot7hip.h)
extern "C" int b7_persist_fallbacks(b7_ctx *c) { return c ? c->persist_aborts : -1; }
