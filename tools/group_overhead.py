"""What does sharding itself cost?  The headline nomination (N = 2048, d = 32, 2^20 Sobol candidates, EI, one hyper sample)
through a single-process group of n VIRTUAL ranks on one GPU (b7_group_eval_nominate; the members time-share the device, so
the ideal is the unsharded time: every member refits the GP -- n fits instead of one -- and the records are merged on the
host).  Not a scaling measurement.   python tools/group_overhead.py [n ...]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
from harness import benchmarks  # noqa: E402

d, N, M = 32, 2048, 1 << 20
c = bot7_amd.Context(0)
pool = c.grid_sobol(N, d, 2)
X, Y = pool, benchmarks.ackley(pool)
amp = float(np.var(Y))
hyp = [dict(lenscale_sq=np.full(d, d / 8.0), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y)))]
spec = dict(score="ei", fmin=[float(Y.min())])
c.grid_sobol(M, d, 2 + N, download=False)
c.gp_set_data(X, Y)
want = c.eval_nominate(hyp, **spec)
c.sync()
t0 = time.perf_counter()
for _ in range(3):
    c.eval_nominate(hyp, **spec)
t1 = (time.perf_counter() - t0) / 3
print("one context: %.2f ms per nomination, winner %r" % (t1 * 1e3, want), flush=True)
c.close()
for n in [int(a) for a in sys.argv[1:]] or (1, 2, 4, 8):
    g = bot7_amd.Group([0] * n)
    g.grid_sobol(M, d, 2 + N)
    g.gp_set_data(X, Y)
    got = g.eval_nominate(hyp, **spec)
    t0 = time.perf_counter()
    for _ in range(3):
        g.eval_nominate(hyp, **spec)
    t = (time.perf_counter() - t0) / 3
    row = g.nominate_commit(got[1])
    print("group of %d virtual ranks (exchange: %s): %.2f ms per nomination (+%.2f ms = %d extra fits + exchange), same winner: %s"
          % (n, "RCCL" if g.info()["uses_rccl"] else "host merge", t * 1e3, (t - t1) * 1e3, n - 1, got == want), flush=True)
    g.close()
