"""A/B of Cholesky settings (environment variables read at context creation) in one process.  Diagnostic.
SETTINGS="B7_POTRF_FUSED=0|B7_POTRF_FUSED=1" python tools/diag_ab.py [N]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
from harness import benchmarks  # noqa: E402
settings = os.environ.get("SETTINGS", "B7_DIAG_VARIANT=0|B7_DIAG_VARIANT=1").split("|")
ctxs = {}
for st in settings:
    kv = dict(item.split("=") for item in st.split(",") if item)
    os.environ.update(kv)
    ctxs[st] = bot7_amd.Context(0, lib="diag")  # the switches live in the diagnostic build (python -m bot7_amd.build --diag)
    for k in kv:
        del os.environ[k]
d, N = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
X = ctxs[settings[0]].grid_sobol(N, d, 1)
Y = benchmarks.ackley(X)
amp = float(np.var(Y))
hyp = (np.full(d, d / 8.0), amp, 1e-4 * amp, float(np.mean(Y)))
ref = None
res = {g: [] for g in settings}
for r in range(8):
    for g, c in ctxs.items():
        c.profile_enable(True)
        c.profile_reset()
        c.gp_fit(X, Y, *hyp)
        res[g].append([c.profile_get(ph)[0] for ph in ("potrf", "trtri", "alpha", "kxx")])
        if r == 0:
            L, _, Li = c.gp_download(N)
            if ref is None:
                ref = (L, Li)
            print(g, ": max |L - L_ref| =", float(np.abs(L - ref[0]).max()), " max |Linv - ref| rel =",
                  float(np.abs(Li - ref[1]).max() / np.abs(ref[1]).max()), flush=True)
for g in settings:
    m = np.median(np.array(res[g][1:]), axis=0)
    print("%-40s potrf %.3f  trtri %.3f  alpha %.3f  kxx %.3f ms (medians)" % (g, m[0], m[1], m[2], m[3]), flush=True)
