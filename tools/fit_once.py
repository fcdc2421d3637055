"""Five GP fits at N (default 2048), d = 32: the workload for a rocprofv3 --kernel-trace timeline.  Diagnostic."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
from harness import benchmarks  # noqa: E402
c = bot7_amd.Context(0)
d, N = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
X = c.grid_sobol(N, d, 1)
Y = benchmarks.ackley(X)
amp = float(np.var(Y))
for r in range(5):
    c.gp_fit(X, Y, np.full(d, d / 8.0), amp, 1e-4 * amp, float(np.mean(Y)))
c.sync()
