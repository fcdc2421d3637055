"""Every export with a LIVE context (and a live group) but NULL pointers / zero sizes everywhere else, in the states a host
can get them wrong in: fresh context, after data + grid + fit.  An error code or a harmless success is fine; a crash is not.
usage (GPU box): python tools/null_sweep_gpu.py"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot7_amd  # noqa: E402
from bot7_amd import _lib  # noqa: E402

L = _lib.load()
SKIP = {"b7_destroy", "b7_group_destroy", "b7_create", "b7_group_create",
        "b7_comm_unique_id", "b7_comm_pick_winner"}  # their first pointer is an output buffer / a table, not a handle


def sweep(handle, prefix_group, label):
    n = 0
    for name in _lib.SYMBOLS:
        if name in SKIP:
            continue
        fn = getattr(L, name)
        if not fn.argtypes or fn.argtypes[0] is not C.c_void_p:
            continue
        if name.startswith("b7_group_") != prefix_group:
            continue
        args = [handle] + [0 if t in (C.c_int, C.c_int64, C.c_uint64) else (0.0 if t is C.c_double else None) for t in fn.argtypes[1:]]
        print("%s %s" % (label, name), flush=True)
        rc = fn(*args)
        print("    -> %s" % (rc,), flush=True)
        n += 1
    return n


ctx = bot7_amd.Context(0)
n = sweep(ctx._h, False, "fresh")
rng = np.random.default_rng(0)
X = rng.random((40, 3))
Y = rng.normal(size=(40, 1))
ctx.grid_sobol(500, 3, 1)
ctx.gp_fit(X, Y, np.full(3, 0.4), 1.0, 1e-3, 0.0)
ctx.gp_predict(download=False)
ctx.score_reset()
n += sweep(ctx._h, False, "fitted")
g = bot7_amd.Group([0, 0])
n += sweep(g._h, True, "group")
print("swept %d calls without a crash" % n)
