"""Fingerprint of the factorisation's bits: SHA-256 over L, L^-1, alpha of GP fits at a few sizes, chol() of a random SPD
matrix, and small-set likelihoods.  Run with BOT7HIP_LIB pointing at two builds to see that a re-scheduling of the factor
routine changed no result bit.   usage (GPU box): python tools/bits_fingerprint.py"""
import hashlib
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402

ctx = bot7_amd.Context(0)
rng = np.random.default_rng(5)
h = hashlib.sha256()
for N, d in ((37, 3), (64, 6), (100, 6), (128, 32), (300, 8), (1024, 16), (2048, 32)):
    X = rng.random((N, d))
    Y = np.sin(X.sum(1, keepdims=True) * 3.0) + 0.01 * rng.normal(size=(N, 1))
    out = ctx.gp_fit(X, Y, np.full(d, d / 8.0), 1.3, 1e-3, 0.1, want_nll=True)
    L, al, Li = ctx.gp_download(N)
    hh = hashlib.sha256(L.tobytes() + al.tobytes() + Li.tobytes() + np.asarray(out["nll"]).tobytes()).hexdigest()
    h.update(hh.encode())
    ctx.gp_set_data(X, Y)
    nll = ctx.gp_nll_batch(np.outer(0.5 + 0.1 * np.arange(5), np.full(d, d / 8.0)), 1.3, 1e-3, 0.1)
    h.update(nll.tobytes())
    print("N %4d d %2d  fit %s  nll batch %s" % (N, d, hh[:16], hashlib.sha256(nll.tobytes()).hexdigest()[:16]))
A = rng.normal(size=(200, 200))
A = A @ A.T + 200 * np.eye(200)
Lc, jit, info = ctx.chol(A)
hh = hashlib.sha256(Lc.tobytes()).hexdigest()
h.update(hh.encode())
print("chol 200  %s  jitter %g info %d" % (hh[:16], jit, info))
print("fingerprint", h.hexdigest())
