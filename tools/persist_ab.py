"""Persistent Cholesky schedule (B7_POTRF_SCHED=3) against the launch schedule (1): bit-for-bit comparison of L, dinv-
derived inv(L), alpha and NLL, and phase times.   python tools/persist_ab.py [N ...]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
from harness import benchmarks  # noqa: E402

ctxs = {}
for sched in ("1", "3"):
    os.environ["B7_POTRF_SCHED"] = sched
    ctxs[sched] = bot7_amd.Context(0, lib="diag")  # the switches live in the diagnostic build (python -m bot7_amd.build --diag)
del os.environ["B7_POTRF_SCHED"]
d = 32
for N in [int(a) for a in sys.argv[1:]] or [100, 256, 700, 1024, 2048]:
    X = ctxs["1"].grid_sobol(N, d, 2)
    Y = benchmarks.ackley(X)
    amp = float(np.var(Y))
    hyp = (np.full(d, d / 8.0), amp, 1e-4 * amp, float(np.mean(Y)))
    out, t = {}, {}
    for g, c in ctxs.items():
        r = c.gp_fit(X, Y, *hyp, want_nll=True)
        out[g] = c.gp_download(N) + (r["nll"],)
        ts = []
        for _ in range(6):
            c.profile_enable(True)
            c.profile_reset()
            c.gp_fit(X, Y, *hyp, want_nll=True)
            ts.append(c.profile_get("potrf")[0])
        c.profile_enable(False)
        t[g] = float(np.median(ts[1:]))
    same = [bool(np.array_equal(a, b)) for a, b in zip(out["1"], out["3"])]
    dmax = [float(np.abs(a - b).max()) for a, b in zip(out["1"], out["3"])]
    print("N %5d  identical L/alpha/Linv/nll: %s  max diff %s   potrf ms: launch %.3f  persistent %.3f"
          % (N, same, ["%.1e" % v for v in dmax], t["1"], t["3"]), flush=True)
