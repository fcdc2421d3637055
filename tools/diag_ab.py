"""A/B of the 64x64 diagonal-block kernel variants (B7_DIAG_VARIANT) in one process.  Diagnostic."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
from bot7_amd import benchmarks  # noqa: E402
variants = [int(g) for g in os.environ.get("VARIANTS", "0,1").split(",")]
ctxs = {}
for g in variants:
    os.environ["B7_DIAG_VARIANT"] = str(g)
    ctxs[g] = bot7_amd.Context(0)
d, N = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
X = ctxs[variants[0]].grid_sobol(N, d, 1)
Y = benchmarks.ackley(X)
amp = float(np.var(Y))
hyp = (np.full(d, d / 8.0), amp, 1e-4 * amp, float(np.mean(Y)))
ref = None
res = {g: [] for g in variants}
for r in range(8):
    for g, c in ctxs.items():
        c.profile_enable(True)
        c.profile_reset()
        c.gp_fit(X, Y, *hyp)
        res[g].append(c.profile_get("potrf")[0])
        if r == 0:
            L, _, Li = c.gp_download(N)
            if ref is None:
                ref = (L, Li)
            print("variant", g, "max |L - L_ref| =", float(np.abs(L - ref[0]).max()), " max |Linv - ref| rel =",
                  float(np.abs(Li - ref[1]).max() / np.abs(ref[1]).max()), flush=True)
for g in variants:
    print("variant %d: potrf median %.3f ms  min %.3f" % (g, np.median(res[g][1:]), min(res[g][1:])), flush=True)
