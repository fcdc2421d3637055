"""Worker of tests/test_sharded_loop.py::test_grouped_rccl_branch_runs_with_virtual_ranks: ONE process, a group of n members on
cuda:0 whose exchange is forced through the grouped all-reduce (B7_GROUP_EXCHANGE=rccl) with RCCL's entry points served by the
in-process test double (B7_RCCL_LIB = tests/stub).  Every line of csrc/group.hip's use_rccl branch -- ncclGroupStart / one
ncclAllReduce per member / ncclGroupEnd, the table read from member 0, the redo with rewritten records -- runs with n > 1.
usage: python tests/_group_worker.py out.json n [n ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bot7_amd  # noqa: E402
from harness import benchmarks  # noqa: E402
from oracle import cport  # noqa: E402

out_path, sizes = sys.argv[1], [int(v) for v in sys.argv[2:]]
res = {"cases": []}
ctx = bot7_amd.Context(0)
rng = np.random.default_rng(3)
for (d, N, M) in ((6, 100, 5001), (6, 300, 20001), (2, 40, 5)):
    pool = cport.sobol(M + N, d, 2)
    X_obs, X_hid = pool[:N].copy(), pool[N:].copy()
    Y = benchmarks.hartmann6(X_obs) if d == 6 else benchmarks.braninhoo(X_obs)
    amp = float(np.var(Y))
    hyps = [dict(lenscale_sq=np.full(d, d / 8.0) * (1 + 0.3 * s), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y))) for s in range(3)]
    Xdup, Ydup = np.concatenate([X_obs, X_obs[:7]]), np.concatenate([Y, Y[:7]])   # duplicated rows + no noise: the plain attempt fails
    hard = [dict(h, noise=0.0) for h in hyps[:2]]
    spec = {"score": "ei", "fmin": [float(Y.min())]}
    ctx.grid_upload(X_hid)
    ctx.gp_set_data(X_obs, Y)
    want = ctx.eval_nominate(hyps, **spec)
    want_cb = ctx.eval_nominate(hyps, score="cb")
    ctx.gp_set_data(Xdup, Ydup)
    wv, wi, rep = ctx.eval_nominate(hard, want_report=True, **spec)
    for n in sizes:
        g = bot7_amd.Group([0] * n)
        case = {"shape": [d, N, M], "n": n, "uses_rccl": bool(g.info()["uses_rccl"])}
        g.grid_upload(X_hid)
        g.gp_set_data(X_obs, Y)
        got = g.eval_nominate(hyps, **spec)
        got_cb = g.eval_nominate(hyps, score="cb")
        case["easy"] = tuple(got) == tuple(want) and tuple(got_cb) == tuple(want_cb)
        row = g.nominate_commit(got_cb[1])
        case["commit_row"] = bool(np.array_equal(row, X_hid[got_cb[1] - 1]))
        case["commit_set"] = bool(np.array_equal(g.grid_download(), np.delete(X_hid, got_cb[1] - 1, axis=0)))
        g.grid_upload(X_hid)
        g.gp_set_data(Xdup, Ydup)
        v, i, r = g.eval_nominate(hard, want_report=True, **spec)          # every member with rows redoes; the empty ones must not
        case["jitter_needed"] = bool((rep["jitter"] > 0).all())             # leave a stale table in the second sum
        case["hard"] = (v, i) == (wv, wi) and bool(np.array_equal(r["jitter"], rep["jitter"]))
        case["hard_values"] = [float(v), int(i), float(wv), int(wi)]
        g.close()
        res["cases"].append(case)
ctx.close()
with open(out_path, "w") as f:
    json.dump(res, f)
