"""What one round of hyper sampling costs at the reference's own sizes (N <= 100 observations, nSamples = 10, slice sampler over
d + 3 hypers: bots/bayesopt.lua:68,73-75 + samplers/slice.lua): the model mirror's sample_hypers with the one-workgroup
likelihood kernel (default) and with the general path (B7_NLL_SMALL=0).   python tools/sampler_latency.py"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
import harness  # noqa: E402,F401  (registers the slice sampler)
from harness import benchmarks  # noqa: E402

for N, d, fn in ((25, 2, benchmarks.braninhoo), (100, 6, benchmarks.hartmann6)):
    for flag in ("1", "0"):
        os.environ["B7_NLL_SMALL"] = flag
        c = bot7_amd.Context(0, lib="diag")  # the switches live in the diagnostic build (python -m bot7_amd.build --diag)
        X = c.grid_sobol(N, d, 2)
        Y = fn(X)
        m = bot7_amd.models.gp_regressor({"sample": True, "nBurnin": 5, "seed": 3}, context=c)
        m.init(X, Y)
        m.sample_hypers(X, Y)                      # burn-in (5 updates)
        for _ in range(3):
            m.sample_hypers(X, Y, None, None, True)
        c.sync()
        e0, t0 = m.nEvals, time.perf_counter()
        for _ in range(10):                        # the ten per-sample updates of one nomination
            m.sample_hypers(X, Y, None, None, True)
        dt, ev = time.perf_counter() - t0, m.nEvals - e0
        print("N %3d d %d  %s: ten hyper samples = %d likelihood evaluations in %.2f ms (%.1f us each, sampler's Python included)"
              % (N, d, "one-workgroup kernel" if flag == "1" else "general path       ", ev, dt * 1e3, dt / ev * 1e6), flush=True)
        c.close()
