"""Same inputs, same bits, call after call: 3000 nominations and 6000 likelihood evaluations at the reference's default sizes
(the paths that answer through a completion word in mapped host memory instead of a stream wait), interleaved with grid
mutations, every result compared with the first one of its kind.   usage (GPU box): python tools/determinism_soak.py"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
from harness import benchmarks  # noqa: E402

ctx = bot7_amd.Context(0)
d, N = 6, 100
X = ctx.grid_sobol(N, d, 2)
Y = benchmarks.hartmann6(X)
amp = float(np.var(Y))
hyps = [{"lenscale_sq": np.full(d, d / 8.0) * (1 + 0.05 * s), "amp": amp, "noise": 1e-4 * amp, "mean": float(np.mean(Y))} for s in range(10)]
ctx.grid_sobol(20000, d, 1000)
ctx.gp_set_data(X, Y)
ls = np.outer(0.6 + 0.1 * np.arange(4), np.full(d, d / 8.0))
first = {}
bad = 0
t0 = time.perf_counter()
for it in range(3000):
    S = (1, 10, 3)[it % 3]
    r = ctx.eval_nominate(hyps[:S], score="ei", fmin=[float(Y.min())])
    k = ("nom", S)
    bad += first.setdefault(k, r) != r
    for B in (1, 4):
        v = ctx.gp_nll_batch(ls[:B], amp, 1e-4 * amp, float(np.mean(Y))).tobytes()
        bad += first.setdefault(("nll", B), v) != v
    if it % 500 == 499:   # a grid mutation and its undo: the nominations must come back to the same bits
        row = ctx.grid_remove(7)
        g = ctx.grid_download()
        ctx.grid_upload(np.vstack([g[:6], row.reshape(1, -1), g[6:]]))
        print("iteration %d, %.1f s, mismatches so far %d" % (it + 1, time.perf_counter() - t0, bad), flush=True)
print("done: %d mismatches in 3000 nominations + 6000 likelihood calls" % bad)
sys.exit(1 if bad else 0)
