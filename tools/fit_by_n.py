"""The GP fit by N (2048 ... 8192, d = 32): wall time, the factorisation phase (Cholesky + inverse, one persistent launch up to
N = 4096, panel launches above) and what fraction of the fp64-MFMA peak its 2N^3/3 flop make.  usage (GPU box): python tools/fit_by_n.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bot7_amd
from harness import benchmarks
c = bot7_amd.Context(0)
c.profile_enable(True)
for N in (2048, 3072, 4096, 6144, 8192):
    d = 32
    X = c.grid_sobol(N, d, 2)
    Y = benchmarks.ackley(X)
    amp = float(np.var(Y))
    c.gp_set_data(X, Y)
    base = dict(lenscale_sq=np.full(d, d / 8.0), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y)))
    for _ in range(3):
        c.gp_fit_hyp(**base)
    c.profile_reset()
    t0 = time.perf_counter()
    for _ in range(8):
        c.gp_fit_hyp(**base)
    wall = (time.perf_counter() - t0) / 8
    ms, n = c.profile_get("potrf")
    tt, tn = c.profile_get("trtri")
    p = ms / n
    print("N %5d: fit wall %.3f ms, potrf phase %.3f ms (%d launches/fit), trtri %.3f ms: Cholesky+inverse 2N^3/3 at %.1f TFLOP/s = %.1f %% of peak; trailing updates N^3/3 %.1f %%" % (N, wall * 1e3, ms / 8, n // 8, tt / 8, 2 * N**3 / 3 / (ms / 8 * 1e-3) / 1e12, 2 * N**3 / 3 / (ms / 8 * 1e-3) / 78.6e12 * 100, N**3 / 3 / (ms / 8 * 1e-3) / 78.6e12 * 100), flush=True)
