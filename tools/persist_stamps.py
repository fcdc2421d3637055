"""Where the persistent Cholesky's critical workgroup spends its time, and when the helper jobs run (s_memtime stamps,
B7_PERSIST_STAMPS=1).   python tools/persist_stamps.py [N] [nll]      (nll: the likelihood mode, one fit of b7_gp_nll_batch)
Critical path per panel p (cycles of the 100 MHz... no: s_memtime ticks = shader cycles at ~2.4 GHz under light load):
  0 factor start | 1 factor end (waves 1..3 finish the previous look-ahead inside its first step) | 2 both next tiles
  waited for / issued | 3 L_pp, inv(L_pp) stored, tiles in LDS | 4 triangular solve done | 5 inv(L_pp) published |
  6 column block 0 of the next diagonal block updated | 7 barrier"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["B7_POTRF_SCHED"] = "3"
os.environ["B7_PERSIST_STAMPS"] = "1"
import bot7_amd  # noqa: E402
from bot7_amd import _lib
from harness import benchmarks  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
d = 32
c = bot7_amd.Context(0, lib="diag")  # the switches live in the diagnostic build (python -m bot7_amd.build --diag)
X = c.grid_sobol(N, d, 2)
Y = benchmarks.ackley(X)
amp = float(np.var(Y))
hyp = (np.full(d, d / 8.0), amp, 1e-4 * amp, float(np.mean(Y)))
NLL = len(sys.argv) > 2 and sys.argv[2] == "nll"
for _ in range(3):
    if NLL:
        c.gp_set_data(X, Y)
        c.gp_nll_batch(hyp[0][None, :], hyp[1], hyp[2], hyp[3])
    else:
        c.gp_fit(X, Y, *hyp)
L = c._L
buf = np.zeros(1 << 16, dtype=np.uint64)
nb, nj = C.c_int(), C.c_int()
rc = L.b7dbg_persist_stamps(c._h, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(nb), C.byref(nj))
assert rc == 0, rc
nb, nj = nb.value, nj.value
crit = buf[:nb * 8].reshape(nb, 8).astype(np.int64)
jobs = buf[nb * 8:nb * 8 + nj * 4].reshape(nj, 4).astype(np.int64)
t0 = crit[0, 0]
GHZ = 2.4
names = ["factor(+rest)", "publish inv", "tiles+store L", "trsm", "-", "col-0 update", "barrier"]
print("panel  start_us  " + "  ".join("%-14s" % s for s in names) + "  total_us")
tot = np.zeros(7)
for p in range(nb):
    row = crit[p]
    if p + 1 < nb:
        dd = np.diff(row) / GHZ / 1e3
        tot += dd
        if p < 4 or p % 8 == 0 or p >= nb - 3:
            print("%5d  %8.2f  " % (p, (row[0] - t0) / GHZ / 1e3) + "  ".join("%-14.2f" % v for v in dd) + "  %.2f" % ((crit[p + 1, 0] - row[0]) / GHZ / 1e3))
print("mean   %8s  " % "" + "  ".join("%-14.2f" % v for v in tot / (nb - 1)))
print("critical path total %.1f us (first factor start -> last factor end)" % ((crit[nb - 1, 1] - t0) / GHZ / 1e3))
print("workgroup 0: entry -> first factor %.1f us; last factor end -> exit %.1f us; entry -> exit %.1f us" % (
    (t0 - crit[nb - 1, 6]) / GHZ / 1e3, (crit[nb - 1, 7] - crit[nb - 1, 1]) / GHZ / 1e3, (crit[nb - 1, 7] - crit[nb - 1, 6]) / GHZ / 1e3))
# job ends and workgroup 0's exit on the 100 MHz real-time counter (comparable across CUs)
if NLL:
    print("likelihood mode: job 0 is the vector job (L z = r); it ends %.1f us after workgroup 0 left; %.1f us in flag waits, "
          "%.1f us loading and accumulating tiles" % ((jobs[0, 1] - crit[nb - 1, 5]) / 100.0, jobs[0, 2] / GHZ / 1e3, jobs[0, 3] / GHZ / 1e3))
    jobs[0, 2:] = 0
lag = (jobs[:, 1] - crit[nb - 1, 5]) / 100.0
late = np.argsort(-lag)[:8]
print("helper jobs still running after workgroup 0 left: %d; the last ends %.1f us later; latest job ids %s (lag us %s)" % (
    int((lag > 0).sum()), lag.max(), late.tolist(), np.round(lag[late], 1).tolist()))
upd = jobs[jobs[:, 3] > 0]
flop = float(upd[:, 3].sum()) * 2 * 64 ** 3
secs = float(upd[:, 2].sum()) / (GHZ * 1e9)
cu_peak = 78.6e12 / 256
print("trailing updates in helper jobs: %d products of 64x64x64 in %d jobs, %.2f us each (wait for the two tiles excluded: sc1 "
      "loads, LDS stage, 64-deep MFMA chains, subtraction) = %.3f TFLOP/s per CU = %.1f %% of one CU's fp64-MFMA peak (%.3f)"
      % (upd[:, 3].sum(), len(upd), secs / upd[:, 3].sum() * 1e6, flop / secs / 1e12, 100 * flop / secs / cu_peak, cu_peak / 1e12))
print("whole launch: N^3/3 = %.2f GFLOP of trailing updates over %.1f us of critical path = %.1f %% of the chip's fp64-MFMA peak "
      "(the schedule is bound by the one-CU factorisation chain, not by the updates)"
      % (N ** 3 / 3 / 1e9, (crit[nb - 1, 1] - t0) / GHZ / 1e3, 100 * (N ** 3 / 3) / ((crit[nb - 1, 1] - t0) / GHZ / 1e9) / 78.6e12))
