"""The reference's trial loop (bots/abstract.lua:112-152) on a candidate set sharded across GPUs (last bullet):
nomination over the union, the nominee's coordinates on every rank, stable deletion on the owner, offset shift behind it.

CPU (`-m "not gpu"`): the library's host-only bookkeeping rule against np.delete on the union; world-2 / world-3 gloo runs
of >= 6 trials of BASELINE cfg1 (braninhoo, 256-point grid) with the oracle standing in for the device, which must reproduce
the unsharded run's nominees, candidate sets and best.
GPU (`-m gpu`): the same through the product -- a single-process group of 2 and 3 virtual ranks on the one GPU
(b7_group_*: the exchange records are merged on the host, RCCL refuses two ranks on one device), a group of one over RCCL,
and b7_nominate_commit on a plain context."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class H(object):
    def __init__(self, name):
        self.name, self.min, self.max, self.size = name, 0.0, 1.0, 1


# ---- the bookkeeping rule (host-only export of the library) ----------------------------------------------------------
def test_shard_commit_rule_is_stable_deletion_on_the_union():
    from bot7_amd import _lib
    from harness import dist
    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 5, 8):
        M = 97
        union = np.arange(M)                      # row ids; the union is the concatenation of the shards in rank order
        shards = [union[slice(*dist.shard_range(M, r, world))].copy() for r in range(world)]
        offs = [dist.shard_range(M, r, world)[0] for r in range(world)]
        for _ in range(60):
            idx = int(rng.integers(1, union.size + 1))
            owners = 0
            for r in range(world):
                loc, new_off = _lib.shard_commit_rule(idx, offs[r], shards[r].size)
                if loc:
                    owners += 1
                    assert shards[r][loc - 1] == union[idx - 1]
                    shards[r] = np.delete(shards[r], loc - 1)
                offs[r] = new_off
            assert owners == 1
            union = np.delete(union, idx - 1)     # utils.tensor.remove on the unsharded tensor (utils/tensor.lua:158-170)
            assert np.array_equal(np.concatenate(shards), union)
            assert offs == list(np.cumsum([0] + [s.size for s in shards[:-1]]))
    with pytest.raises(_lib.Bot7HipError):
        _lib.shard_commit_rule(0, 0, 5)


# ---- gloo rehearsal on CPU: the oracle stands in for the device ---------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg1_grid():
    """BASELINE cfg1's 256-point uniform grid on [0,1]^2 (Torch's MT stream is not in the tree: a fixed numpy stream)."""
    return np.random.default_rng(11).random((256, 2))


def _run_bot(cand, model_ctx, trials, nSamples=2, sample=False):
    """`trials` trials of the harness bot (harness/bots: the reference's run_trial, statement for statement) over the given
    candidate set -> (nominees, responses, best)."""
    import bot7_amd
    from harness import benchmarks, bots
    cfg = {"bot": {"verbose": 0, "budget": trials, "nInitial": 2, "nSamples": nSamples, "seed": 4},
           "grid": {"type": "random", "size": 256, "dims": 2}, "score": {"type": "expected_improvement"}}
    model = bot7_amd.models.gp_regressor({"sample": True, "nBurnin": 1, "seed": 5} if sample else {}, context=model_ctx)
    bot = bots.bayesopt(benchmarks.braninhoo, [H("x1"), H("x2")], cfg, cache={"candidates": cand, "model": model})
    xs = []
    for _ in range(trials):
        x, y = bot.run_trial()
        bot.update_best(x, y)
        xs.append(np.array(x, dtype=np.float64).ravel())
    return np.stack(xs), bot.responses.copy(), (bot.best["t"], np.array(bot.best["x"]).ravel(), float(np.ravel(bot.best["y"])[0]))


def _loop_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as td
    from harness import dist
    from test_dist_gloo import _OracleCtx
    td.init_process_group("gloo", rank=rank, world_size=world)
    grid, trials = _cfg1_grid(), 8
    # the unsharded run (every rank computes it for itself)
    whole = _OracleCtx(None, None, grid.copy())
    one = dist.ShardedScorer(whole, grid.shape[0], 0, 1)
    xs1, ys1, best1 = _run_bot(one, whole, trials)
    # the sharded run: this rank owns rows [lo, hi)
    lo, hi = dist.shard_range(grid.shape[0], rank, world)
    mine = _OracleCtx(None, None, grid[lo:hi].copy())
    shard = dist.ShardedScorer(mine, grid.shape[0], rank, world)
    xs, ys, best = _run_bot(shard, mine, trials)
    # the union of the shards after the run, in rank order, against the unsharded candidate set
    import torch
    parts = [None] * world
    td.all_gather_object(parts, mine.X)
    ok = {"nominees": bool(np.array_equal(xs, xs1)), "responses": bool(np.array_equal(ys, ys1)),
          "best": best[0] == best1[0] and np.array_equal(best[1], best1[1]) and best[2] == best1[2],
          "candidates": bool(np.array_equal(np.concatenate(parts), whole.X)),
          "offset": shard.lo == sum(p.shape[0] for p in parts[:rank]), "rows": whole.X.shape[0] == grid.shape[0] - trials}
    td.barrier()
    td.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_trial_loop_gloo_reproduces_the_unsharded_run(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok in results:
        assert all(ok.values()), "rank %d: sharded trial loop differs from the unsharded one: %s" % (rank, ok)


# ---- the product: single-process groups of virtual ranks on the one GPU -------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("sample", [False, True])
def test_group_trial_loop_reproduces_the_single_context_run(ctx, sample):
    """>= 6 trials of cfg1 (braninhoo, 256-point grid, nInitial = 2) through b7_group_eval_nominate + b7_group_nominate_commit
    with the grid sharded over 2 and 3 members, against the same loop on one context: nominees, responses, best and the
    candidate set after every trial are identical, bit for bit.  With sample = True the hypers come from the slice sampler
    (three per nomination), so the marginalisation runs sharded too."""
    import bot7_amd
    from harness import benchmarks, bots, dist
    grid, trials = _cfg1_grid(), 9
    mcfg = {"sample": True, "nBurnin": 1, "seed": 5} if sample else {}

    def run(cand, model_ctx, after_trial):
        cfg = {"bot": {"verbose": 0, "budget": trials, "nInitial": 2, "nSamples": 3, "seed": 4},
               "grid": {"type": "random", "size": 256, "dims": 2}, "score": {"type": "expected_improvement"}}
        model = bot7_amd.models.gp_regressor(dict(mcfg), context=model_ctx)
        bot = bots.bayesopt(benchmarks.braninhoo, [H("x1"), H("x2")], cfg, cache={"candidates": cand, "model": model})
        xs, sets = [], []
        for _ in range(trials):
            x, y = bot.run_trial()
            bot.update_best(x, y)
            xs.append(np.array(x, dtype=np.float64).ravel())
            sets.append(after_trial())
        return np.stack(xs), bot.responses.copy(), (bot.best["t"], float(np.ravel(bot.best["y"])[0])), sets

    ctx.grid_upload(grid)
    one = dist.ShardedScorer(ctx, grid.shape[0], 0, 1)          # a world of one: b7_eval_nominate + b7_nominate_commit
    xs1, ys1, best1, sets1 = run(one, ctx, ctx.grid_download)
    host = grid.copy()
    for t, x in enumerate(xs1):                                  # the unsharded bookkeeping, on the host
        row = np.where((host == x).all(axis=1))[0][0]
        host = np.delete(host, row, axis=0)
        assert np.array_equal(sets1[t], host)
    for n in (2, 3):
        g = bot7_amd.Group([0] * n)
        assert g.info() == {"n": n, "uses_rccl": False}
        g.grid_upload(grid)
        xs, ys, best, sets = run(dist.GroupCandidates(g), g.members[0], g.grid_download)
        assert np.array_equal(xs, xs1), "group of %d nominated differently" % n
        assert np.array_equal(ys, ys1) and best == best1
        for a, b in zip(sets, sets1):
            assert np.array_equal(a, b)
        M, d, offs = g.grid_shape()
        assert M == grid.shape[0] - trials and offs[-1] == M and list(offs) == sorted(offs)
        g.close()


@pytest.mark.gpu
def test_group_nomination_matches_one_context_bit_for_bit(ctx, orc):
    """hartmann6, N = 300, 50 001 Sobol candidates, S = 3, EI and -LCB: the winner (value bits and index) of a group of 1 (the
    exchange goes through RCCL: ncclCommInitAll + grouped all-reduce), 2, 3 and 5 virtual ranks equals the single context's;
    the committed row is the grid row; an observation set that needs the jitter schedule takes the per-sample redo on every
    member and still agrees."""
    import bot7_amd
    from harness import benchmarks
    d, N, M = 6, 300, 50001
    pool = orc.c.sobol(M + N, d, 2)
    X_obs, X_hid = pool[:N].copy(), pool[N:].copy()
    Y = benchmarks.hartmann6(X_obs)
    amp = float(np.var(Y))
    hyps = [dict(lenscale_sq=np.full(d, d / 8.0) * (1 + 0.3 * s), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y))) for s in range(3)]
    Xdup = np.concatenate([X_obs, X_obs[:7]])                    # duplicated rows + no noise: the plain attempt fails
    Ydup = np.concatenate([Y, Y[:7]])
    hard = [dict(h, noise=0.0) for h in hyps[:2]]
    specs = ({"score": "ei", "fmin": [float(Y.min())]}, {"score": "cb"})
    ctx.grid_upload(X_hid)
    want = []
    ctx.gp_set_data(X_obs, Y)
    for sp in specs:
        want.append(ctx.eval_nominate(hyps, **sp))
    ctx.gp_set_data(Xdup, Ydup)
    wv, wi, rep = ctx.eval_nominate(hard, want_report=True, **specs[0])
    assert (rep["jitter"] > 0).all(), "the hard case was meant to need the jitter schedule"
    for n in (1, 2, 3, 5):
        g = bot7_amd.Group([0] * n)
        assert g.info()["uses_rccl"] == (n == 1)
        g.grid_sobol(M, d, 2 + N)                                # every member generates its own shard
        assert np.array_equal(g.grid_download(), X_hid)
        g.gp_set_data(X_obs, Y)
        for sp, w in zip(specs, want):
            v, i = g.eval_nominate(hyps, **sp)
            assert (v, i) == w, "group of %d: %r != %r" % (n, (v, i), w)
        row = g.nominate_commit(i)                               # the last winner: the row comes out of the exchange record
        assert np.array_equal(row, X_hid[i - 1])
        assert np.array_equal(g.grid_download(), np.delete(X_hid, i - 1, axis=0))
        row2 = g.nominate_commit(17)                             # not a winner: read from the owner's grid
        assert np.array_equal(row2, np.delete(X_hid, i - 1, axis=0)[16])
        with pytest.raises(bot7_amd.Bot7HipError):
            g.nominate_commit(M)                                 # M - 2 rows are left
        with pytest.raises(bot7_amd.Bot7HipError):
            g.members[0].grid_upload(X_hid[:10])                 # a member's grid only changes through the group
        g.grid_upload(X_hid)
        g.gp_set_data(Xdup, Ydup)
        v, i, r = g.eval_nominate(hard, want_report=True, **specs[0])
        assert (v, i) == (wv, wi) and np.array_equal(r["jitter"], rep["jitter"]) and np.array_equal(r["info"], rep["info"])
        g.close()


@pytest.mark.gpu
def test_nominate_commit_on_a_plain_context_and_a_world_of_one(ctx, orc):
    """b7_nominate_commit without a communicator and with a world-of-one RCCL communicator: the winner's row comes from the
    exchange record (equal to the grid row), any other index is read from the grid; the row is deleted stably; offsets
    follow the rule; the exchange record reports the shard's rows."""
    import bot7_amd
    from bot7_amd import _lib
    from harness import benchmarks
    d, N, M = 6, 64, 3000
    pool = orc.c.sobol(M + N, d, 2)
    X_obs, X_hid = pool[:N].copy(), pool[N:].copy()
    Y = benchmarks.hartmann6(X_obs)
    amp = float(np.var(Y))
    hyp = dict(lenscale_sq=np.full(d, d / 8.0), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y)))
    for with_comm in (False, True):
        c = bot7_amd.Context(0)
        if with_comm:
            c.comm_init(0, 1, _lib.comm_unique_id())
        c.grid_upload(X_hid)
        c.gp_set_data(X_obs, Y)
        off = 1000                                              # pretend 1000 rows of other shards lie before this one
        v, i = c.eval_nominate([hyp], score="ei", fmin=[float(Y.min())], global_row_offset=off)
        info = c.exchange_info()
        assert info["world"] == 1 and list(info["rows"]) == [M] and info["winner_idx1"] == i and info["winner_rank"] == 0
        assert np.array_equal(info["winner_row"], X_hid[i - off - 1])
        row, new_off = c.nominate_commit(i, off)
        assert new_off == off and np.array_equal(row, X_hid[i - off - 1])
        left = np.delete(X_hid, i - off - 1, axis=0)
        assert np.array_equal(c.grid_download(), left)
        with pytest.raises(bot7_amd.Bot7HipError):
            c.exchange_info()                                   # the grid has changed since
        row, new_off = c.nominate_commit(off + 5, off)          # a random initial pick: no exchange record to take it from
        assert np.array_equal(row, left[4]) and new_off == off
        left = np.delete(left, 4, axis=0)
        assert np.array_equal(c.grid_download(), left)
        if not with_comm:
            row, new_off = (None, None)
            with pytest.raises(bot7_amd.Bot7HipError):
                c.nominate_commit(off, off)                     # the row before this shard: not here, and nobody else to ask
        c.close()


# ---- one-sided affine maps of the grid generators (grids/sobol.lua:82-85, grids/random.lua:29-32) ----------------------
def test_oracle_one_sided_maps_follow_the_lua_lines(orc):
    """mins only: grid:add(torch.add(mins, grid:min(1)[1])); maxes only: grid:cmul(torch.cdiv(maxes, grid:max(1)[1])) --
    restated here with numpy straight from those two lines (one rounding for the per-column term, one per element)."""
    g = orc.c.sobol(257, 5, 3)
    mins, maxes = np.array([-1.0, 0.5, 2.0, 0.0, 1e-3]), np.array([2.0, 3.0, 0.25, 1.0, 7.0])
    assert np.array_equal(orc.c.sobol(257, 5, 3, mins=mins), g + (mins + g.min(axis=0)))
    assert np.array_equal(orc.c.sobol(257, 5, 3, maxes=maxes), g * (maxes / g.max(axis=0)))
    assert np.array_equal(orc.c.sobol(257, 5, 3, mins=mins, maxes=maxes), g * (maxes + (-mins)) + mins)
    assert np.array_equal(orc.c.affine(g), g)


@pytest.mark.gpu
def test_one_sided_grid_maps_match_the_oracle_on_a_context_and_on_a_group(ctx, orc):
    import bot7_amd
    rng = np.random.default_rng(5)
    for dims, size, skip in ((2, 300, 1), (7, 4097, 5), (39, 1000, 2)):
        mins, maxes = rng.normal(size=dims), rng.random(dims) + 0.5
        for kw in ({"mins": mins}, {"maxes": maxes}):
            want = orc.c.sobol(size, dims, skip, **kw)
            assert np.array_equal(ctx.grid_sobol(size, dims, skip, **kw), want), (dims, list(kw))
            assert np.array_equal(ctx.grid_download(), want)
            raw = ctx.grid_random(size, dims, seed=9)
            assert np.array_equal(ctx.grid_random(size, dims, seed=9, **kw), orc.c.affine(raw, **kw))
            lo, hi = ctx.grid_colrange()
            got = ctx.grid_download()
            assert np.array_equal(lo, got.min(axis=0)) and np.array_equal(hi, got.max(axis=0))
            for n in (2, 3):
                g = bot7_amd.Group([0] * n)     # the column extremes are those of the UNION of the members' shards
                g.grid_sobol(size, dims, skip, **kw)
                assert np.array_equal(g.grid_download(), want), (dims, list(kw), n)
                g.grid_random(size, dims, seed=9, **kw)
                assert np.array_equal(g.grid_download(), orc.c.affine(raw, **kw))
                g.close()


def test_missing_rccl_is_an_error_code_not_a_crash():
    """the dlopen failure path called dlerror() twice and built a std::string from NULL.  In a fresh process with
    the library name forced to miss, b7_comm_unique_id must come back with B7_ERR_COMM and a message."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from bot7_amd import _lib\n"
            "try:\n    _lib.comm_unique_id()\n    print('no error')\n"
            "except _lib.Bot7HipError as e:\n    print('code', e.code, str(e))\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, B7_RCCL_LIB="/nonexistent/librccl-not-here.so", BOT7HIP_LIB=_diag_lib()))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "code -7" in out.stdout and "librccl-not-here" in out.stdout, out.stdout


# ---- torch.rand's own stream for grids/random.lua (MT19937) ---------------------------------------------------------------
def test_torch_rand_stream_known_answers():
    """MT19937 known answers [public knowledge]: with the reference seed 5489 the first outputs are 3499211612, 581869302,
    3890346734, 3586334585, 545404204 and the 10000th is 4123659995 (Matsumoto & Nishimura's mt19937ar.c / C++11 [rand.predef]);
    numpy's legacy seeding is the same init_genrand, so RandomState(seed).randint / random_sample give independent checks of
    both resolutions."""
    from bot7_amd import _lib
    u = _lib.torch_rand(5489, 10000, 32)
    first = (u[:5] * 4294967296.0).astype(np.uint64)
    assert list(first) == [3499211612, 581869302, 3890346734, 3586334585, 545404204]
    assert int(u[9999] * 4294967296.0) == 4123659995
    for seed in (0, 1, 123456789):
        rs = np.random.RandomState(seed)
        raw = rs.randint(0, 2 ** 32, size=64, dtype=np.uint64)
        assert np.array_equal(_lib.torch_rand(seed, 64, 32), raw.astype(np.float64) / 4294967296.0)
        hi, lo = raw[0::2], raw[1::2]
        want53 = (((hi << np.uint64(32)) | lo) & np.uint64((1 << 53) - 1)).astype(np.float64) * 2.0 ** -53
        assert np.array_equal(_lib.torch_rand(seed, 32, 53), want53)
    with pytest.raises(_lib.Bot7HipError):
        _lib.torch_rand(1, 4, 24)


@pytest.mark.gpu
def test_grid_random_with_the_torch_stream(ctx, orc):
    from bot7_amd import _lib
    size, dims = 257, 7
    u = _lib.torch_rand(42, size * dims, 32).reshape(size, dims)
    mins, maxes = np.linspace(-1, 1, dims), np.linspace(2, 5, dims)
    assert np.array_equal(ctx.grid_random_torch(size, dims, seed=42), u)
    assert np.array_equal(ctx.grid_random_torch(size, dims, seed=42, mins=mins, maxes=maxes), orc.c.affine(u, mins, maxes))
    assert np.array_equal(ctx.grid_download(), orc.c.affine(u, mins, maxes))
    assert np.array_equal(ctx.grid_random_torch(size, dims, seed=42, mins=mins), orc.c.affine(u, mins=mins))
    assert np.array_equal(ctx.grid_random_torch(size, dims, seed=42, maxes=maxes), orc.c.affine(u, maxes=maxes))


@pytest.mark.gpu
def test_group_with_more_members_than_rows_and_shards_that_run_empty(ctx, orc):
    """Five members, seven candidates: members 2..4 hold one row each, nominations keep working while shards empty out one by
    one, down to the last candidate; every step equals the single context's; an empty group refuses to nominate."""
    import bot7_amd
    from harness import benchmarks
    d, N, M = 2, 12, 7
    rng = np.random.default_rng(21)
    X_obs, X_hid = rng.random((N, d)), rng.random((M, d))
    Y = benchmarks.braninhoo(X_obs)
    amp = float(np.var(Y))
    hyps = [dict(lenscale_sq=np.full(d, 0.3), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y)))]
    g = bot7_amd.Group([0] * 5)
    g.grid_upload(X_hid)
    assert list(g.grid_shape()[2]) == [0, 2, 4, 5, 6, 7]
    g.gp_set_data(X_obs, Y)
    ctx.grid_upload(X_hid)
    ctx.gp_set_data(X_obs, Y)
    left = X_hid.copy()
    for t in range(M):
        want = ctx.eval_nominate(hyps, score="cb")
        got = g.eval_nominate(hyps, score="cb")
        assert got == want, "step %d" % t
        row = g.nominate_commit(got[1])
        assert np.array_equal(row, left[got[1] - 1]) and np.array_equal(ctx.nominate_commit(got[1], 0)[0], row)
        left = np.delete(left, got[1] - 1, axis=0)
        assert np.array_equal(g.grid_download(), left) if left.shape[0] else g.grid_shape()[0] == 0
    with pytest.raises(bot7_amd.Bot7HipError) as e:
        g.eval_nominate(hyps, score="cb")
    assert e.value.code == -4
    g.close()


# ---- several PROCESSES, one GPU: the per-process communicator path with a test double for RCCL's transport ----------------
def _diag_lib():
    """The DIAGNOSTIC build of the library (-DB7_DIAG: B7_RCCL_LIB, B7_GROUP_EXCHANGE=rccl and the A/B switches exist there only);
    the worker processes of this file use it as THE library (BOT7HIP_LIB)."""
    from bot7_amd import build as B
    return B.build_diag()


def _stub_lib():
    """tests/stub/rccl_shm_stub.cpp -> tests/stub/_build/librccl_shm_stub.so (tools/build_stub.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from build_stub import stub_lib
    return stub_lib()


@pytest.mark.gpu
@pytest.mark.parametrize("world,sample", [(2, False), (3, True)])
def test_processes_sharing_the_gpu_run_the_trial_loop_through_the_communicator_path(ctx, world, sample, tmp_path):
    """The one-process-per-GPU layout for real, as far as one GPU allows: `world` processes, each with its own context on cuda:0
    and a communicator from b7_comm_init, run 7 trials of cfg1 in lock step through b7_eval_nominate (exchange branch: the
    collective after the pivot check, the winner's row in the record) and b7_nominate_commit (record row for model-based trials,
    broadcast all-reduce for the random initial ones, offset shift).  RCCL refuses two ranks on one device, so its eight entry
    points are served by a shared-memory test double loaded through B7_RCCL_LIB; every line of libbot7hip is the product's.
    Every rank must reproduce the single-context run, the shards must concatenate to its candidate set, and a rank that fails
    locally must make EVERY rank return B7_ERR_COMM (nobody hangs in the collective)."""
    import json
    import subprocess
    from harness import dist
    trials, grid = 7, _cfg1_grid()
    ctx.grid_upload(grid)
    one = dist.ShardedScorer(ctx, grid.shape[0], 0, 1)
    xs1, ys1, best1 = _run_bot(one, ctx, trials, nSamples=3, sample=sample)
    left1 = ctx.grid_download()
    env = dict(os.environ, B7_RCCL_LIB=_stub_lib(), BOT7HIP_LIB=_diag_lib(), PYTHONPATH=ROOT, B7_TEST_SAMPLE="1" if sample else "0")
    ident = ("b7stub_%d_%d" % (os.getpid(), world)).encode().hex()
    outs = [str(tmp_path / ("rank%d.json" % r)) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_comm_worker.py"), str(r), str(world), ident, str(trials),
                               outs[r]], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append((p.returncode, e[-1500:]))
    assert all(rc == 0 for rc, _ in logs), logs
    res = [json.load(open(o)) for o in outs]
    rows = []
    for r, d in enumerate(res):
        assert np.array_equal(np.array(d["nominees"]), xs1), "rank %d nominated differently" % r
        assert np.array_equal(np.array(d["responses"]), ys1)
        assert d["best"][0] == best1[0] and np.array_equal(np.array(d["best"][1]), best1[1]) and d["best"][2] == best1[2]
        assert d["lo"] == sum(len(x["rows"]) for x in res[:r]), "rank %d: offset after %d commits" % (r, trials)
        rows += d["rows"]
        code = d["failure"][0] if isinstance(d["failure"], list) else None
        assert code == (-1 if r == world - 1 else -7), "rank %d: %r" % (r, d["failure"])   # its own error / B7_ERR_COMM
    assert np.array_equal(np.array(rows).reshape(-1, 2), left1)


# ---- the boundary from plain C -----------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_one_sided_grid_map_is_a_collective_every_rank_leaves(orc, tmp_path):
    """grids/sobol.lua:82-83 (mins only) on a grid sharded over ranks with a communicator: the column minima of the WHOLE grid are
    combined by one all-reduce, so the call is collective -- and a rank that fails BEFORE it (here: 40 dims, refused as
    grids/sobol.lua:36 refuses them) must still enter it with a failure status, so that every rank returns (its own error /
    B7_ERR_COMM) instead of the others waiting in ncclAllReduce for ever.  Two processes on cuda:0 over the
    shared-memory RCCL double."""
    import json
    import subprocess
    world = 2
    env = dict(os.environ, B7_RCCL_LIB=_stub_lib(), BOT7HIP_LIB=_diag_lib(), PYTHONPATH=ROOT)
    ident = ("b7one_%d" % os.getpid()).encode().hex()
    outs = [str(tmp_path / ("r%d.json" % r)) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_onesided_worker.py"), str(r), str(world), ident, outs[r]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    for p in procs:
        try:
            o, e = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-2000:]
    res = [json.load(open(o)) for o in outs]
    mins = np.array([-1.0, 0.5, 2.0, 0.0, 1e-3])
    want = orc.c.sobol(1001, 5, 3, mins=mins)
    assert np.array_equal(np.concatenate([np.array(r["rows"]) for r in res]), want)
    assert res[1]["failure"][0] == -6, res[1]["failure"]          # its own B7_ERR_RANGE
    assert res[0]["failure"][0] == -7, res[0]["failure"]          # B7_ERR_COMM: another rank failed


@pytest.mark.gpu
def test_grouped_rccl_branch_runs_with_virtual_ranks(tmp_path):
    """csrc/group.hip's grouped-RCCL branch (ncclCommInitAll, ncclGroupStart / one all-reduce per member / ncclGroupEnd, the
    table read from member 0, the redo with rewritten records) with n = 2, 3, 8 members on ONE GPU: B7_GROUP_EXCHANGE=rccl
    forces the branch for repeated device ids and the in-process test double (tests/stub) serves RCCL's entry points.  The
    nomination must equal the single context's bit for bit -- the small-fit path (N = 100), the general one (N = 300), and a
    5-candidate grid on 8 members (three shards empty) whose data need the jitter schedule: the redo must not sum a stale
    table.  Real RCCL's transport across devices stays unmeasured (one GPU per box)."""
    import json
    import subprocess
    out = str(tmp_path / "group.json")
    env = dict(os.environ, B7_RCCL_LIB=_stub_lib(), B7_GROUP_EXCHANGE="rccl", BOT7HIP_LIB=_diag_lib(), PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_group_worker.py"), out, "2", "3", "8"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.load(open(out))
    assert len(res["cases"]) == 9
    for c in res["cases"]:
        assert c["uses_rccl"], c
        assert c["easy"] and c["commit_row"] and c["commit_set"], c
        assert c["jitter_needed"] and c["hard"], c


def test_header_is_c99_and_cxx11():
    """include/bot7hip.h is the drop-in boundary: it must compile as C (what cgo / LuaJIT's cdef / ctypes users assume) and as
    C++, warning-free and pedantic."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "bot7hip.h")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                ["g++", "-std=c++11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c++", hdr]):
        out = subprocess.run(cmd, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr


@pytest.mark.gpu
def test_c_driver_runs_the_trial_loop_through_the_abi_alone(orc, tmp_path):
    """examples/c_driver.c: a bot7 trial loop in C99 against the header and the .so, nothing else (grid from torch.rand's
    stream, two random picks, GP + EI nominations, b7_nominate_commit).  Every model-based nomination must be the oracle's
    arg-max on the data the driver had at that point, and the same loop over a single-process group of three virtual ranks
    must print the same lines."""
    import re
    import subprocess
    from bot7_amd import _lib
    exe = str(tmp_path / "c_driver")
    lib_dir = os.path.join(ROOT, "bot7_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_driver.c"), "-o", exe, "-L" + lib_dir, "-lbot7hip", "-lm",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    trials = 9
    one = subprocess.run([exe, str(trials), "1"], capture_output=True, text=True, timeout=120)
    assert one.returncode == 0, one.stderr
    three = subprocess.run([exe, str(trials), "3"], capture_output=True, text=True, timeout=120)
    assert three.returncode == 0, three.stderr
    assert one.stdout == three.stdout, "the group of three virtual ranks printed a different run"
    rows = re.findall(r"trial +(\d+) +idx +(\d+) +x = \(([^,]+), ([^)]+)\) +y = (\S+)", one.stdout)
    assert len(rows) == trials
    u = _lib.torch_rand(7, 256 * 2 + 2, 32)
    cand = u[:512].reshape(256, 2).copy()
    X, Y = [], []
    for t, (tt, idx, x0, x1, y) in enumerate(rows, 1):
        idx, x, y = int(idx), np.array([float(x0), float(x1)]), float(y)
        if t <= 2:
            assert idx == int(np.floor(u[512 + t - 1] * cand.shape[0])) + 1
        else:
            Xa, Ya = np.array(X), np.array(Y).reshape(-1, 1)
            amp = float(np.var(Ya)) or 1.0
            f = orc.gp.fit(Xa, Ya, np.full(2, 0.25), amp, 1e-4 * amp, float(np.mean(Ya)))
            mu, var = orc.gp.predict(f, cand)
            assert idx == orc.c.argmax_first(orc.c.ei(mu, var, [float(Ya.min())]))[0], "trial %d" % t
        assert np.array_equal(x, cand[idx - 1])
        from harness import benchmarks
        assert y == pytest.approx(float(benchmarks.braninhoo(x)[0, 0]), rel=1e-13)
        cand = np.delete(cand, idx - 1, axis=0)
        X.append(x)
        Y.append(y)
