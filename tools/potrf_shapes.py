"""The persistent Cholesky in the three shapes the path runs it in, as a workload for rocprofv3:

    single   one GP fit at N = 2048 (b7_gp_fit_hyp: Cholesky + inverse, one critical workgroup + 255 helpers)
    s10      the reference's nSamples = 10 marginalisation (bots/abstract.lua:67): ten fits side by side in one launch
             (b7_eval_nominate over a 256-row grid, so the launch of interest is the fit batch)
    nll16    sixteen likelihood evaluations in one launch (b7_gp_nll_batch: no inverse)

    python tools/potrf_shapes.py single|s10|nll16 [N] [reps]

Prints wall-clock per call; under `rocprofv3 --kernel-trace --pmc ...` the potrf_persist_kernel rows are what
tools/potrf_pmc_summary.py reads."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
from harness import benchmarks  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "single"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
d = 32
c = bot7_amd.Context(0)
pool = c.grid_sobol(N + 256, d, 2)
X, grid = pool[:N].copy(), pool[N:].copy()
Y = benchmarks.ackley(X)
amp = float(np.var(Y))
base = dict(lenscale_sq=np.full(d, d / 8.0), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y)))
c.gp_set_data(X, Y)
c.grid_upload(grid)


def once():
    if shape == "single":
        c.gp_fit_hyp(**base)
    elif shape == "s10":
        hyps = [dict(base, lenscale_sq=base["lenscale_sq"] * (0.8 + 0.05 * s), amp=amp * (0.9 + 0.02 * s)) for s in range(10)]
        c.eval_nominate(hyps, score="ei", fmin=[float(Y.min())])
    elif shape == "nll16":
        c.gp_nll_batch(np.outer(0.8 + 0.03 * np.arange(16), base["lenscale_sq"]), amp, base["noise"], base["mean"])
    else:
        raise SystemExit("unknown shape " + shape)


once()
c.sync()
t0 = time.perf_counter()
for _ in range(reps):
    once()
c.sync()
print("%s N %d: %.3f ms per call (wall, %d calls)" % (shape, N, (time.perf_counter() - t0) / reps * 1e3, reps), flush=True)
