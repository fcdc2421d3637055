"""Prints the achieved relative errors of posterior mean/variance against the oracle (margin vs the 1e-5 bar)."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bot7_amd  # noqa: E402
from harness import benchmarks as B  # noqa: E402
from oracle import cport, gp  # noqa: E402
from conftest import make_problem  # noqa: E402


class O(object):
    c = cport
    gp = gp


ctx = bot7_amd.Context(0)
for d, N, M, obj in [(2, 24, 256, B.braninhoo), (6, 256, 4096, B.hartmann6), (32, 129, 515, B.ackley),
                     (5, 300, 2049, B.rastrigin), (39, 64, 128, B.rastrigin), (32, 1024, 4096, B.ackley),
                     (32, 2048, 4096, B.ackley), (64, 2048, 2048, B.rastrigin)]:
    X_obs, Y, X_hid, hyp = make_problem(ctx, O, d, N, M, obj) if d < 40 else (None,) * 4
    if d >= 40:
        X_obs = ctx.grid_random(N, d, seed=1, row_offset=10 ** 6)
        X_hid = ctx.grid_random(M, d, seed=1)
        Y = obj(X_obs)
        amp = float(np.var(Y))
        hyp = dict(lenscale_sq=np.full(d, d / 8.0), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y)))
    f = gp.fit(X_obs, Y, **hyp)
    ctx.gp_fit(X_obs, Y, **hyp)
    ctx.grid_upload(X_hid)
    mu, var = ctx.gp_predict()
    mu_o, var_o = gp.predict(f, X_hid)
    em = np.max(np.abs(mu - mu_o)) / np.abs(mu_o).max()
    ev = np.max(np.abs(var - var_o) / var_o)
    print("d=%2d N=%4d M=%5d: mean err %.2e (of max|mu|), var rel err %.2e, min var/amp %.2e, cond(K) %.1e"
          % (d, N, M, em, ev, var_o.min() / hyp["amp"], np.linalg.cond(f.L) ** 2), flush=True)
