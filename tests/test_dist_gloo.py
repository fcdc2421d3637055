"""The N>1 path on CPU: world_size-2 (and 3) gloo groups run the sharding arithmetic and the single
arg-max exchange of bot7_amd.dist against the unsharded oracle result."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as td
    from harness import dist
    from oracle import cport
    td.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    rng = np.random.default_rng(123)
    cases = []
    s = rng.normal(size=10007)
    cases.append(s.copy())
    s2 = s.copy()
    s2[[5000, 9000, 20]] = s.max() + 2.0  # ties across shards -> lowest global index
    cases.append(s2)
    s3 = s2.copy()
    s3[[7000, 3000]] = np.nan             # NaN in two shards -> first NaN
    cases.append(s3)
    cases.append(rng.normal(size=1))      # more ranks than rows: some shards are empty
    for sc in cases:
        lo, hi = dist.shard_range(sc.size, rank, world)
        if hi > lo:
            i, v = cport.argmax_first(sc[lo:hi])
        else:
            i, v = 0, 0.0
        gv, gi = dist.exchange_best(v, i, lo, device="cpu")
        wi, wv = cport.argmax_first(sc)
        out.append((gi == wi) and ((gv == wv) or (gv != gv and wv != wv)))
    td.barrier()
    td.destroy_process_group()
    q.put((rank, out))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_argmax_exchange_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, out in results:
        assert all(out), "rank %d disagreed with the unsharded arg-max: %s" % (rank, out)


def _OracleCtx(X_obs, Y, X_shard):
    """The device's stand-in on a machine without a GPU: oracle/hostctx.py (the calls ShardedScorer and the harness bot make on a
    Context, answered by the oracle over this rank's rows)."""
    sys.path.insert(0, ROOT)
    from oracle.hostctx import OracleContext
    return OracleContext(X_obs, Y, X_shard)


def _nominate_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as td
    from harness import benchmarks, dist
    from oracle import cport, gp
    td.init_process_group("gloo", rank=rank, world_size=world)
    d, N, M = 6, 40, 1501
    pool = cport.sobol(M + N, d, 2)
    X_obs, X_hid = pool[:N], pool[N:]
    Y = benchmarks.hartmann6(X_obs)
    amp = float(np.var(Y))
    hyps = [{"lenscale_sq": np.full(d, d / 8.0) * (1.0 + 0.2 * s), "amp": amp, "noise": 1e-4 * amp, "mean": float(np.mean(Y))}
            for s in range(3)]
    ok = []
    for spec in ({"score": "ei", "fmin": [float(Y.min())], "tradeoff": 0.0}, {"score": "cb"}):
        ctx = _OracleCtx(X_obs, Y, None)
        shard = dist.ShardedScorer(ctx, M, rank, world)
        ctx.X = X_hid[shard.lo:shard.hi]
        val, idx = shard.eval_nominate(hyps, spec, device="cpu")
        whole = _OracleCtx(X_obs, Y, X_hid)           # the unsharded nomination
        for s, h in enumerate(hyps):
            whole.gp_predict_hyp(h["lenscale_sq"], h["amp"], h["noise"], h["mean"])
            if s == 0:
                whole.score_reset()
            whole.score_ei(spec["fmin"], 0.0) if spec["score"] == "ei" else whole.score_cb(1.0, False, -1.0)
        wv, wi, _ = whole.score_finish(3.0)
        ok.append(idx == wi and val == wv)
    td.barrier()
    td.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_eval_nominate_gloo(world):
    """bayesopt:eval + nominate over a grid sharded across ranks (the gloo path of ShardedScorer.eval_nominate: per-sample
    calls on each shard, one (value, index) exchange): every rank gets the unsharded winner."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nominate_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok in results:
        assert all(ok), "rank %d: sharded nomination differs from the unsharded one: %s" % (rank, ok)
