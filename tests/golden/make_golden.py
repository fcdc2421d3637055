"""Regenerates the committed fixtures from the CPU oracle (oracle/).  Run from the repo root:

    python tests/golden/make_golden.py

Provenance: "restatement-derived, not Torch7-derived" -- the reference is Lua/Torch7 and cannot run in this
pipeline (no interpreter, no `gp` package), and it holds no golden outputs of its own.  The Sobol rows are
produced by oracle/b7_oracle.c's stateful restatement of grids/sobol.lua:216-335 (which reproduces the traced
KATs of SURVEY 8a-8); the GP rows by oracle/gp.py (textbook Cholesky-form regression over the in-repo
pdist/chol numerics; PARITY UNPINNED, see oracle/gp.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cport, gp  # noqa: E402
from harness import benchmarks as B  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def sobol_cases():
    cases = []
    for size, dims, skip, row0, nrows, mm in [(1024, 39, 1, 0, 8, None), (1024, 39, 1, 1016, 8, None),
                                              (300, 6, 1, 100, 6, None), (64, 32, 4097, 0, 4, None),
                                              (50, 3, 1, 40, 5, ([-1.5, 0.0, 2.0], [2.5, 1.0, 10.0]))]:
        mins, maxes = (mm if mm else (None, None))
        pts = cport.sobol(size, dims, skip, mins, maxes)
        case = {"size": size, "dims": dims, "skip": skip, "row0": row0,
                "rows_hex": [[float(v).hex() for v in row] for row in pts[row0:row0 + nrows]]}
        if mm:
            case["mins"], case["maxes"] = mins, maxes
        cases.append(case)
    return {"provenance": "oracle/b7_oracle.c orc_sobol_generate (restatement of grids/sobol.lua)", "cases": cases}


def gp_small():
    d, N, M = 6, 16, 64
    pool = cport.sobol(M + N, d, 1)
    step = (M + N) // N
    obs = np.arange(N) * step
    mask = np.ones(M + N, dtype=bool)
    mask[obs] = False
    X_obs, X_hid = pool[obs], pool[mask]
    Y = B.hartmann6(X_obs)
    amp = float(np.var(Y))
    hyp = dict(lenscale_sq=np.full(d, d / 8.0), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y)))
    f = gp.fit(X_obs, Y, **hyp)
    mu, var = gp.predict(f, X_hid)
    ei = cport.ei(mu, var, [float(Y.min())])
    cb = cport.cb(mu, var)
    np.savez(os.path.join(HERE, "gp_small.npz"), X_obs=X_obs, Y_obs=Y, X_hid=X_hid, mu=mu, var=var, ei=ei, cb=cb,
             ei_argmax1=cport.argmax_first(ei)[0], cb_argmax1=cport.argmax_first(cb)[0], L=f.L, alpha=f.alpha,
             nll=f.nll, **hyp)


if __name__ == "__main__":
    with open(os.path.join(HERE, "sobol_kat.json"), "w") as fh:
        json.dump(sobol_cases(), fh, indent=0)
    gp_small()
    print("fixtures written to", HERE)
