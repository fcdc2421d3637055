"""Per-launch time of the K* assembly and of the posterior kernel at a one-block problem (N = 25, d = 2, 2e5 candidates): A/B of two
builds through BOT7HIP_LIB.  usage (GPU box): [BOT7HIP_LIB=...] python tools/post_small_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bot7_amd
from harness import benchmarks
ctx = bot7_amd.Context(0)
ctx.profile_enable(True)
d, N = 2, 25
X = ctx.grid_sobol(N, d, 2); Y = benchmarks.braninhoo(X)
ctx.grid_sobol(200000, d, 1000)
amp = float(np.var(Y))
ctx.gp_fit(X, Y, np.full(d, 0.25), amp, 1e-4 * amp, float(np.mean(Y)))
for _ in range(5):
    ctx.gp_predict(download=False)
ctx.profile_reset()
for _ in range(20):
    ctx.gp_predict(download=False)
ctx.sync()
for ph in ("ksx", "post"):
    ms, n = ctx.profile_get(ph)
    print(os.environ.get("BOT7HIP_LIB", "shipped")[-24:], ph, "%.1f us per launch" % (ms / n * 1e3))
