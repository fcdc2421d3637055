"""Per-block phase timing of the first K=128 trailing update of the Cholesky (diagnostic instantiation)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["B7_SYRK_STAMPS"] = "1"
os.environ["B7_POTRF_SCHED"] = "0"   # the pair schedule with stand-alone K = 128 update launches
os.environ["B7_POTRF_DEFER"] = "0"
import bot7_amd  # noqa: E402
from bot7_amd import benchmarks  # noqa: E402
c = bot7_amd.Context(0)
d, N = 32, 2048
X = c.grid_sobol(N, d, 1)
Y = benchmarks.ackley(X)
amp = float(np.var(Y))
for _ in range(3):
    c.gp_fit(X, Y, np.full(d, d / 8.0), amp, 1e-4 * amp, float(np.mean(Y)))
