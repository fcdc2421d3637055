"""Same-box A/B of one likelihood evaluation's C-call latency between builds of the library (clocks differ by a few per cent
between boxes of the pool, so two builds are only comparable within one run).
usage (GPU box): python3 tools/nll_ab.py libA.so libB.so [...]      (paths relative to the repo root; each library is measured
in its own child process, rounds interleaved)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ((2, 25), (6, 64), (6, 80), (6, 100), (6, 112), (32, 128))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    import bot7_amd
    from bot7_amd import _lib
    ctx = bot7_amd.Context(0)
    L = _lib.load()
    rng = np.random.default_rng(3)
    out = []
    for d, N in CASES:
        X = rng.random((N, d))
        Y = np.sin(3.0 * X.sum(1, keepdims=True)) + 0.01 * rng.normal(size=(N, 1))
        ctx.gp_set_data(X, Y)
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (np.full((1, d), d / 8.0), [1.3], [1e-3], [0.1])]
        res = np.empty(1)
        args = [ctx._h, 1] + [_lib._ptr(v) for v in a] + [_lib._ptr(res), None, None]
        for _ in range(200):
            L.b7_gp_nll_batch(*args)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(1000):
                L.b7_gp_nll_batch(*args)
            best = min(best, (time.perf_counter() - t0) / 1000 * 1e6)
        out.append("%.2f" % best)
    print(" ".join(out), flush=True)


if __name__ == "__main__":
    if sys.argv[1:] == ["--child"]:
        child()
        sys.exit(0)
    libs = sys.argv[1:]
    print("C call of b7_gp_nll_batch, one hyper vector, best of 5 x 1000, us; (d, N) =", CASES)
    for rnd in range(3):
        for lib in libs:
            env = dict(os.environ, BOT7HIP_LIB=os.path.join(ROOT, lib))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
            print("round %d  %-40s %s" % (rnd, lib, r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)
