"""A/B of the Cholesky panel-group size (B7_POTRF_GROUP) in one process.  Diagnostic."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
from harness import benchmarks  # noqa: E402
groups = [int(g) for g in os.environ.get("GROUPS", "1,2,4,8").split(",")]
ctxs = {}
for g in groups:
    os.environ["B7_POTRF_GROUP"] = str(g)
    ctxs[g] = bot7_amd.Context(0, lib="diag")  # the switches live in the diagnostic build (python -m bot7_amd.build --diag)
d, N = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
X = ctxs[groups[0]].grid_sobol(N, d, 1)
Y = benchmarks.ackley(X)
amp = float(np.var(Y))
hyp = (np.full(d, d / 8.0), amp, 1e-4 * amp, float(np.mean(Y)))
ref = None
res = {g: [] for g in groups}
for r in range(6):
    for g, c in ctxs.items():
        c.profile_enable(True)
        c.profile_reset()
        c.gp_fit(X, Y, *hyp)
        res[g].append(c.profile_get("potrf")[0])
        if r == 0:
            L, _, _ = c.gp_download(N)
            ref = L if ref is None else ref
            print("group", g, "max |L - L_ref| =", float(np.abs(L - ref).max()), flush=True)
for g in groups:
    print("group %d: potrf median %.3f ms  min %.3f" % (g, np.median(res[g][1:]), min(res[g][1:])), flush=True)
