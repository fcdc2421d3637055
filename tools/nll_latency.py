"""Latency of ONE likelihood evaluation (what a slice-sampler step costs) and of a batch, small observation sets, through
b7_gp_nll_batch with the one-workgroup kernel (default) and through the general path (B7_NLL_SMALL=0), against the oracle.
usage: nll_latency.py   (GPU box)"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
from bot7_amd import _lib  # noqa: E402
from harness import benchmarks  # noqa: E402
from oracle import gp  # noqa: E402

ctxs = {}
for flag in ("1", "0"):
    os.environ["B7_NLL_SMALL"] = flag
    ctxs[flag] = bot7_amd.Context(0, lib="diag")  # the switches live in the diagnostic build (python -m bot7_amd.build --diag)
del os.environ["B7_NLL_SMALL"]
for d, N, fn in ((2, 24, benchmarks.braninhoo), (6, 64, benchmarks.hartmann6), (6, 100, benchmarks.hartmann6), (32, 128, benchmarks.ackley)):
    X = ctxs["1"].grid_sobol(N, d, 2)
    Y = fn(X)
    amp = float(np.var(Y))
    ls = np.outer(0.6 + 0.1 * np.arange(16), np.full(d, d / 8.0))
    want = [float(gp.fit(X, Y, ls[b], amp, 1e-4 * amp, float(np.mean(Y))).nll[0]) for b in range(16)]
    line = "d %2d N %3d:" % (d, N)
    for flag, c in ctxs.items():
        c.gp_set_data(X, Y)
        got = c.gp_nll_batch(ls, amp, 1e-4 * amp, float(np.mean(Y)))
        err = float(np.max(np.abs(got - want) / np.abs(want)))
        for B in (1, 16):
            c.gp_nll_batch(ls[:B], amp, 1e-4 * amp, float(np.mean(Y)))
            c.sync()
            t0 = time.perf_counter()
            for _ in range(200):
                c.gp_nll_batch(ls[:B], amp, 1e-4 * amp, float(np.mean(Y)))
            t = (time.perf_counter() - t0) / 200
            line += "  %s B=%2d %.1f us" % ("small" if flag == "1" else "general", B, t * 1e6)
            if B == 1:
                # the C call alone (what a Lua / C host pays): arguments converted once, ctypes' own call cost (~1 us) included
                L = _lib.load()
                a = [np.ascontiguousarray(v, dtype=np.float64) for v in (ls[:1], [amp], [1e-4 * amp], [float(np.mean(Y))])]
                out = np.empty(1)
                args = [c._h, 1] + [_lib._ptr(v) for v in a] + [_lib._ptr(out), None, None]
                t0 = time.perf_counter()
                for _ in range(500):
                    L.b7_gp_nll_batch(*args)
                line += " (C call %.1f us)" % ((time.perf_counter() - t0) / 500 * 1e6)
        line += " (rel err vs oracle %.1e)" % err
    print(line, flush=True)

# above the one-workgroup kernel's range: the general path alone (observation scaling, K, fix-up, one persistent launch with the
# vector job; hypers up and terms + reports back through the pinned block, one copy each way)
c = ctxs["1"]
for d, N in ((6, 200), (6, 500), (16, 1000), (32, 2048)):
    X = c.grid_sobol(N, d, 2)
    Y = benchmarks.hartmann6(X[:, :6]) if d >= 6 else benchmarks.braninhoo(X)
    amp = float(np.var(Y))
    c.gp_set_data(X, Y)
    L = _lib.load()
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (np.full((1, d), d / 8.0), [amp], [1e-4 * amp], [float(np.mean(Y))])]
    out = np.empty(1)
    args = [c._h, 1] + [_lib._ptr(v) for v in a] + [_lib._ptr(out), None, None]
    for _ in range(5):
        L.b7_gp_nll_batch(*args)
    t0 = time.perf_counter()
    for _ in range(200):
        L.b7_gp_nll_batch(*args)
    t = (time.perf_counter() - t0) / 200
    want = float(gp.fit(X, Y, np.full(d, d / 8.0), amp, 1e-4 * amp, float(np.mean(Y))).nll[0])
    print("d %2d N %4d:  general B= 1 C call %.1f us (rel err vs oracle %.1e)" % (d, N, t * 1e6, abs(out[0] - want) / abs(want)), flush=True)
