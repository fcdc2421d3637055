"""A/B of post_kernel variants in ONE process (interleaved rounds): B7_POST_VARIANT is read at context creation.
Diagnostic tool, not product.  Usage: python tools/post_ab.py [M] [rounds]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
from bot7_amd import benchmarks  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
variants = [int(v) for v in os.environ.get("VARIANTS", "0,1,2,3").split(",")]
d, N = 32, 2048
ctxs = {}
for v in variants:
    os.environ["B7_POST_VARIANT"] = str(v)
    ctxs[v] = bot7_amd.Context(0)
c0 = ctxs[variants[0]]
X_obs = c0.grid_sobol(N, d, 1 + M)
Y = benchmarks.ackley(X_obs)
amp = float(np.var(Y))
hyp = dict(lenscale_sq=np.full(d, d / 8.0), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y)))
ref = None
for v, c in ctxs.items():
    c.grid_sobol(M, d, 1, download=False)
    c.gp_fit(X_obs, Y, **hyp)
    mu, var = c.gp_predict()
    if ref is None:
        ref = var
    print("variant", v, "max |var - var_v0| / var =", float(np.max(np.abs(var - ref) / ref)), flush=True)
    c.profile_enable(True)
res = {v: [] for v in variants}
for r in range(rounds):
    for v, c in ctxs.items():
        c.profile_reset()
        c.gp_predict(download=False)
        c.sync()
        ms, n = c.profile_get("post")
        ks, _ = c.profile_get("ksx")
        res[v].append((ms, ks))
for v in variants:
    ms = np.array([a for a, _ in res[v]])
    ks = np.array([b for _, b in res[v]])
    tf = M * float(N) * N / (ms * 1e-3) / 1e12
    print("variant %d: post ms median %.3f min %.3f  -> %.2f TFLOP/s (%.1f%% of 78.6) | ksx ms %.3f -> %.0f GB/s"
          % (v, np.median(ms), ms.min(), np.median(tf), 100 * np.median(tf) / 78.6, np.median(ks),
             M * 8.0 * N / (np.median(ks) * 1e-3) / 1e9), flush=True)
