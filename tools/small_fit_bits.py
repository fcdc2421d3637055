"""The one-launch small-problem kernels (gp_small.hip, kpost_small.hip) against the general schedule, BIT FOR BIT: L, L^-1, alpha
of gp_fit, the likelihoods of gp_nll_batch (against round 3's four-wave kernel), posterior mean / variance and the nomination of
eval_nominate.  The switches are read at b7_create, so the two contexts live side by side in one process.
usage (GPU box): python tools/small_fit_bits.py"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
from harness import benchmarks  # noqa: E402


def make(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return bot7_amd.Context(0, lib="diag" if env else None)   # the switches exist in the diagnostic build only
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


new = make({})
ref = make({"B7_FIT_SMALL": "0", "B7_NLL_SMALL": "2", "B7_KPOST_SMALL": "0"})
rng = np.random.default_rng(7)
bad = 0
for N, d in ((2, 2), (5, 1), (16, 3), (17, 6), (25, 2), (48, 6), (63, 6), (64, 6), (65, 6), (70, 2), (80, 6), (81, 5), (96, 6), (97, 7),
             (100, 6), (112, 6), (113, 9), (127, 16), (128, 32), (100, 32), (33, 31)):
    X = rng.random((N, d))
    Y = np.sin(X.sum(1, keepdims=True) * 3.0) + 0.01 * rng.normal(size=(N, 1))
    ls = np.full(d, d / 8.0) * (0.5 + rng.random(d))
    outs = []
    for ctx in (new, ref):
        o = ctx.gp_fit(X, Y, ls, 1.3, 1e-3, 0.1, want_nll=True)
        L, al, Li = ctx.gp_download(N)
        ctx.gp_set_data(X, Y)
        nll = ctx.gp_nll_batch(np.outer(0.5 + 0.1 * np.arange(5), ls), 1.3, 1e-3, 0.1)
        nll1 = ctx.gp_nll_batch(ls, 1.3, 1e-3, 0.1)
        ctx.grid_sobol(3000 + N, d, 5, download=False)
        hyps = [{"lenscale_sq": ls * (1 + 0.05 * s), "amp": 1.3, "noise": 1e-3, "mean": 0.1} for s in range(10)]
        p = ctx.gp_predict_hyp(ls, 1.3, 1e-3, 0.1, download=True)
        b1 = ctx.eval_nominate(hyps[:1], score="ei", fmin=[float(Y.min())])
        s1 = ctx.score_finish(1.0, download=True)[2]
        b10 = ctx.eval_nominate(hyps, score="ei", fmin=[float(Y.min())])
        s10 = ctx.score_finish(1.0, download=True)[2]
        b3 = ctx.eval_nominate(hyps[:3], score="cb")
        outs.append({"L": L, "alpha": al, "Linv": Li, "fit_nll": np.asarray(o["nll"]), "nll5": nll, "nll1": nll1, "mean": p["mean"],
                     "var": p["var"], "b1": np.array(b1), "s1": s1, "b10": np.array(b10), "s10": s10, "b3": np.array(b3)})
    line = []
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        same = a.tobytes() == b.tobytes()
        if not same:
            bad += 1
            with np.errstate(all="ignore"):
                rel = float(np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
            line.append("%s DIFF(%.1e)" % (k, rel))
    print("N %3d d %2d  %s" % (N, d, "all bits equal" if not line else "  ".join(line)), flush=True)
print("mismatching items:", bad)
sys.exit(1 if bad else 0)
