"""Worker of tests/test_sharded_loop.py::test_one_sided_grid_map_is_a_collective_every_rank_leaves: rank `r` of a world of `n` on
cuda:0 (communicator over the shared-memory RCCL double).  Round 1: every rank makes its shard of a mins-only Sobol grid (the
column minima are combined across ranks) and reports it.  Round 2: the LAST rank asks for 40 dims (refused: grids/sobol.lua:36)
-- it must return its own error, every other rank B7_ERR_COMM, nobody may hang in the all-reduce.
usage: python tests/_onesided_worker.py rank world id_hex out.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bot7_amd  # noqa: E402
from bot7_amd import _lib  # noqa: E402
from harness import dist  # noqa: E402

rank, world, ident, out = int(sys.argv[1]), int(sys.argv[2]), bytes.fromhex(sys.argv[3]), sys.argv[4]
ctx = bot7_amd.Context(0)
ctx.comm_init(rank, world, ident.ljust(128, b"\0"))
size, dims = 1001, 5
lo, hi = dist.shard_range(size, rank, world)
mins = np.array([-1.0, 0.5, 2.0, 0.0, 1e-3])
shard = ctx.grid_sobol(hi - lo, dims, 3 + lo, mins=mins)
res = {"rows": shard.tolist(), "lo": lo}
try:
    bad = 40 if rank == world - 1 else dims
    ctx.grid_sobol(hi - lo, bad, 3 + lo, mins=np.zeros(bad))
    res["failure"] = None
except _lib.Bot7HipError as e:
    res["failure"] = [e.code, str(e)]
ctx.comm_allreduce([0.0])      # the communicator still works afterwards
with open(out, "w") as f:
    json.dump(res, f)
