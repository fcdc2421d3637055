"""Wall time of each C-ABI call of one small scoring step (hartmann6, N = 256, M = 32768): where the host time goes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd
from harness import benchmarks
c = bot7_amd.Context(0)
d, N, M = 6, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 32768
X = c.grid_sobol(N, d, 1 + M)
c.grid_sobol(M, d, 1, download=False)
Y = benchmarks.hartmann6(X)
amp = float(np.var(Y))
hyp = (np.full(d, d / 8.0), amp, 1e-4 * amp, float(np.mean(Y)))
acc = {}
def timed(name, fn):
    t0 = time.perf_counter(); r = fn(); acc.setdefault(name, []).append(time.perf_counter() - t0); return r
for it in range(60):
    c.sync()
    t0 = time.perf_counter()
    timed("gp_fit", lambda: c.gp_fit(X, Y, *hyp))
    timed("gp_predict", lambda: c.gp_predict(download=False))
    timed("score_reset", lambda: c.score_reset())
    timed("score_cb", lambda: c.score_cb())
    timed("score_finish", lambda: c.score_finish(1.0))
    acc.setdefault("step", []).append(time.perf_counter() - t0)
for k, v in acc.items():
    print("%-14s median %.1f us" % (k, 1e6 * float(np.median(v[10:]))))
c.profile_enable(True); c.profile_reset()
c.gp_fit(X, Y, *hyp); c.gp_predict(download=False); c.score_reset(); c.score_cb(); c.score_finish(1.0)
print({ph: round(c.profile_get(ph)[0] * 1e3, 1) for ph in ("kxx", "potrf", "alpha", "ksx", "post", "score", "argmax")}, "us (GPU, events)")
