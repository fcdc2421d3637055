"""K(X*,X) assembly rate for several input dimensions (N = 2048, 262144 candidates per launch).  Diagnostic.
usage: ksx_rate.py [d ...]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
c = bot7_amd.Context(0)
N, M = 2048, 262144
for d in ([int(a) for a in sys.argv[1:]] or (2, 6, 16, 32, 39, 64, 96)):
    X = c.grid_random(N, d, seed=3, row_offset=10 * M)
    Y = np.sin(X.sum(1, keepdims=True))
    c.grid_random(M, d, seed=3, download=False)
    c.gp_fit(X, Y, np.full(d, d / 8.0), 1.0, 1e-4, 0.0)
    c.gp_predict(download=False)
    c.profile_enable(True)
    ts = []
    for _ in range(5):
        c.profile_reset()
        c.gp_predict(download=False)
        c.sync()
        ts.append(c.profile_get("ksx")[0])
    c.profile_enable(False)
    ms = float(np.median(ts))
    print("d = %2d: ksx %.3f ms -> %.0f GB/s (%.1f%% of 8 TB/s)" % (d, ms, M * 8.0 * N / (ms * 1e-3) / 1e9,
                                                                   100 * M * 8.0 * N / (ms * 1e-3) / 8e12), flush=True)
