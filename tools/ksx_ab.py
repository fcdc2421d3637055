"""Prices the parts of ksx_kernel (stores / exp / MFMA) by ablation, interleaved in one process.  Diagnostic."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
from bot7_amd import benchmarks  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
modes = [int(v) for v in os.environ.get("MODES", "0,1,2,3").split(",")]
d, N = 32, 2048
ctxs = {}
for v in modes:
    os.environ["B7_KSX_ABLATE"] = str(v)
    ctxs[v] = bot7_amd.Context(0)
c0 = ctxs[modes[0]]
X_obs = c0.grid_sobol(N, d, 1 + M)
Y = benchmarks.ackley(X_obs)
amp = float(np.var(Y))
hyp = dict(lenscale_sq=np.full(d, d / 8.0), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y)))
for v, c in ctxs.items():
    c.grid_sobol(M, d, 1, download=False)
    c.gp_fit(X_obs, Y, **hyp)
    c.gp_predict(download=False)
    c.profile_enable(True)
res = {v: [] for v in modes}
for r in range(rounds):
    for v, c in ctxs.items():
        c.profile_reset()
        c.gp_predict(download=False)
        c.sync()
        res[v].append(c.profile_get("ksx")[0])
names = {0: "full", 1: "no stores", 2: "no exp", 3: "no mfma"}
for v in modes:
    ms = np.median(res[v])
    print("ksx %-10s %.3f ms  (%.0f GB/s equivalent)" % (names.get(v, v), ms, M * 8.0 * N / (ms * 1e-3) / 1e9), flush=True)
