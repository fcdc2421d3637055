"""One rank of tests/test_sharded_loop.py's multi-process run: several PROCESSES share cuda:0, each with its own context and a
communicator (b7_comm_init) whose transport is the shared-memory test double tests/stub/rccl_shm_stub.cpp (B7_RCCL_LIB) --
the world > 1 branches of b7_eval_nominate and b7_nominate_commit for real.
usage: _comm_worker.py rank world idhex trials out.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

rank, world, idhex, trials, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
import bot7_amd  # noqa: E402
from harness import dist  # noqa: E402
from test_sharded_loop import _cfg1_grid, _run_bot  # noqa: E402

ctx = bot7_amd.Context(0)
ident = bytes.fromhex(idhex).ljust(128, b"\0")
ctx.comm_init(rank, world, ident)
assert ctx.comm_info() == (rank, world)
grid = _cfg1_grid()
lo, hi = dist.shard_range(grid.shape[0], rank, world)
ctx.grid_upload(grid[lo:hi])
shard = dist.ShardedScorer(ctx, grid.shape[0], rank, world)
res = {"rank": rank}
xs, ys, best = _run_bot(shard, ctx, trials, nSamples=3, sample=(os.environ.get("B7_TEST_SAMPLE") == "1"))
res.update(nominees=xs.tolist(), responses=ys.tolist(), best=[best[0], best[1].tolist(), best[2]], lo=shard.lo,
           rows=ctx.grid_download().tolist(), info_world=None)
# a failure on ONE rank must come back as an error on EVERY rank, not as a hang: rank 1 asks for EI without fmin ... no:
# arguments are checked before anything is enqueued, so break the state instead -- rank (world - 1) drops its data
ctx.gp_set_data(np.asarray(xs), np.asarray(ys))
hyp = dict(lenscale_sq=np.full(2, 0.3), amp=1.0, noise=1e-3, mean=0.0)
if rank == world - 1:
    ctx.grid_upload(np.zeros((3, 5)))       # wrong dims for the data: eval_validate fails on this rank alone
try:
    ctx.eval_nominate([hyp], score="cb", global_row_offset=shard.lo)
    res["failure"] = "no error"
except bot7_amd.Bot7HipError as e:
    res["failure"] = [e.code, str(e)]
ctx.comm_destroy()
ctx.close()
json.dump(res, open(out, "w"))
