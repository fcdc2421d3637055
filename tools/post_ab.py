"""A/B of the two large-grid posterior kernels (B7_POST_SHAPE=4 default vs 8) on one problem: bitwise comparison of the
variances, where they differ, and HIP-event times.   usage: python tools/post_ab.py [N d M]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import bot7_amd  # noqa: E402

N, d, M = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (700, 6, 70000)
NOISE = float(sys.argv[4]) if len(sys.argv) >= 5 else 1e-4
out = {}
for shape in ("4", "8"):
    os.environ["B7_POST_SHAPE"] = shape
    ctx = bot7_amd.Context(0)
    X_obs = bench.make_inputs(ctx, d, N, M, 0, M)
    Y = np.sin(3.0 * X_obs).sum(axis=1, keepdims=True)
    amp = float(np.var(Y))
    ctx.gp_fit(X_obs, Y, np.full(d, d / 8.0), amp, NOISE * amp, float(np.mean(Y)))
    mu, var = ctx.gp_predict()
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(5):
        ctx.gp_predict(download=False)
    ctx.sync()
    ms, n = ctx.profile_get("post")
    ctx.profile_enable(False)
    ctx.grid_upload(ctx.grid_download(0, 4096))
    small = ctx.gp_predict()[1]
    out[shape] = (var, ms / n, small)
    ctx.close()
a, b = out["4"][0], out["8"][0]
print("small-grid shape vs w4: %d of 4096 differ; vs w8: %d" % (int((out["4"][2] != a[:4096]).sum()), int((out["8"][2] != b[:4096]).sum())))
bad = np.nonzero(a != b)[0]
print("post ms: w4 %.4f  w8 %.4f   speed-up %.3f" % (out["4"][1], out["8"][1], out["8"][1] / out["4"][1]))
print("candidates that differ: %d of %d" % (bad.size, M))
if bad.size:
    rel = np.abs(a[bad] - b[bad]) / np.abs(b[bad])
    print("max rel diff %.3e, median %.3e; first indices %s" % (rel.max(), np.median(rel), bad[:20]))
    print("by (index %% 256) // 16:", np.bincount((bad % 256) // 16, minlength=16))
    print("by index // 256 (first 20 blocks):", np.bincount(bad // 256)[:20])
    print("nan in w4:", int(np.isnan(a).sum()), " example pairs:", list(zip(a[bad[:5]], b[bad[:5]])))
