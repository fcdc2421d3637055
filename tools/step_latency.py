"""Wall time of one nomination at a small configuration (cfg2 by default) with and without the per-phase HIP events
bench.py records, for the fused call (b7_eval_nominate) and for the separate entry points.
usage: python tools/step_latency.py [d N M S]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import bot7_amd  # noqa: E402
from harness import benchmarks  # noqa: E402

d, N, M, S = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (6, 256, 32768, 1)
ctx = bot7_amd.Context(0)
X_obs = bench.make_inputs(ctx, d, N, M, 0, M)
Y = benchmarks.registry["hartmann6" if d == 6 else "ackley"](X_obs)
amp = float(np.var(Y))
hyp = {"lenscale_sq": np.full(d, d / 8.0), "amp": amp, "noise": 1e-4 * amp, "mean": float(np.mean(Y))}
hyps = [dict(hyp, lenscale_sq=hyp["lenscale_sq"] * (1 + 0.05 * s)) for s in range(S)]
ctx.gp_set_data(X_obs, Y)


def fused():
    return ctx.eval_nominate(hyps, score="cb")


def separate():
    for s, h in enumerate(hyps):
        ctx.gp_predict_hyp(h["lenscale_sq"], h["amp"], h["noise"], h["mean"])
        if s == 0:
            ctx.score_reset()
        ctx.score_cb()
    return ctx.score_finish_global(float(S), 0)


for name, fn in (("fused", fused), ("separate", separate)) * 2:
    for events in (False, True):
        ctx.profile_enable(events)
        ctx.profile_reset()
        for _ in range(50):
            r = fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(500):
            r = fn()
        ctx.sync()
        el = (time.perf_counter() - t0) / 500
        ctx.timer_start(0)
        for _ in range(500):
            r = fn()
        ctx.timer_stop(0)
        print("%-9s phase events %-5s wall %.4f ms   stream time %.4f ms   %s" % (name, events, el * 1e3, ctx.timer_ms(0) / 500, r))
ctx.profile_enable(False)
ctx.close()
