"""One nomination at the reference's own default sizes (bots/abstract.lua:60-67: 2e4 candidates, budget 100, ten hyper
samples): wall time per b7_eval_nominate, for a kernel trace of its launches.
usage (GPU box): python tools/nominate_default_trace.py   (or under rocprofv3 --kernel-trace)"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
from harness import benchmarks  # noqa: E402

ctx = bot7_amd.Context(0)
CASES = ((2, 25, benchmarks.braninhoo), (6, 100, benchmarks.hartmann6))
if os.environ.get("B7_TRACE_CASE"):
    CASES = CASES[int(os.environ["B7_TRACE_CASE"]):][:1]
for d, N, fn in CASES:
    X = ctx.grid_sobol(N, d, 2)
    Y = fn(X)
    ctx.grid_sobol(20000, d, 1000)
    ctx.gp_set_data(X, Y)
    amp = float(np.var(Y))
    hyps = [{"lenscale_sq": np.full(d, d / 8.0) * (1 + 0.05 * s), "amp": amp, "noise": 1e-4 * amp, "mean": float(np.mean(Y))} for s in range(10)]
    spec = {"score": "ei", "fmin": [float(Y.min())], "tradeoff": 0.0}
    for S in (1, 10, 1, 10):
        for _ in range(5):
            best = ctx.eval_nominate(hyps[:S], **spec)
        ts = []
        for _ in range(100):
            t0 = time.perf_counter()
            best = ctx.eval_nominate(hyps[:S], **spec)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6
        # the C call alone (what a Lua / C host pays): arguments packed once
        from bot7_amd import _lib
        import ctypes as C
        L = _lib.load()
        arr, keep = ctx._pack_hyps(hyps[:S], d)
        sp, fm = ctx._pack_spec("ei", spec["fmin"], 0.0, False, -1.0)
        v, i = C.c_double(), C.c_int64()
        args = (ctx._h, S, arr, C.byref(sp), 0, C.byref(v), C.byref(i), None, None)
        for _ in range(5):
            L.b7_eval_nominate(*args)
        t0 = time.perf_counter()
        for _ in range(200):
            L.b7_eval_nominate(*args)
        ccall = (time.perf_counter() - t0) / 200 * 1e6
        print("d %d N %3d, 20000 candidates, %2d hyper samples: mean %.1f us per nomination, median %.1f, max %.0f at call %d (winner %s); C call alone %.1f us"
              % (d, N, S, ts.mean(), np.median(ts), ts.max(), int(ts.argmax()), best[1], ccall), flush=True)
