"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Needs an MI355X: run with -m gpu.

Bars (BASELINE.json north_star): bit-exact for integer/index/byte work (Sobol, row removal, arg-max, CB given
identical inputs); posterior mean/var within 1e-5 relative (fp64); EI values within a few ulp of the oracle's
(exp() differs between glibc and ocml by <= 1 ulp, everything else is the same rounded op sequence)."""
import os

import numpy as np
import pytest

from harness import benchmarks as B
from harness import bots
from conftest import make_problem

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
REL = 1e-5  # north_star tolerance for posterior mean / variance


def relerr(a, b, floor=0.0):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def test_native_library_is_loaded(ctx):
    import bot7_amd
    info = ctx.device_info()
    assert "gfx950" in info["name"] and info["compute_units"] >= 200
    shipped = os.path.join(os.path.dirname(bot7_amd.__file__), "libbot7hip.so")
    override = os.environ.get("BOT7HIP_LIB")      # the whole suite against another build (the diagnostic one: DESIGN section 1)
    with open("/proc/self/maps") as f:
        assert os.path.basename(override or shipped) in f.read()
    assert os.path.samefile(bot7_amd.lib_path(), override or shipped)


# ---- grids ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("size,dims,skip", [(1, 1, 1), (8, 3, 1), (1000, 6, 1), (4096, 39, 1), (777, 32, 12345),
                                            (64, 5, 0), (300, 17, 2 ** 20 - 100)])
def test_sobol_bit_exact(ctx, orc, size, dims, skip):
    got = ctx.grid_sobol(size, dims, skip)
    assert np.array_equal(got, orc.c.sobol(size, dims, skip))
    assert np.array_equal(ctx.grid_download(), got)


def test_sobol_affine_bit_exact_and_errors(ctx, orc):
    mins, maxes = np.linspace(-7.3, 2.1, 9), np.linspace(3.3, 11.9, 9)
    assert np.array_equal(ctx.grid_sobol(500, 9, 1, mins, maxes), orc.c.sobol(500, 9, 1, mins, maxes))
    import bot7_amd
    with pytest.raises(bot7_amd.Bot7HipError) as e:
        ctx.grid_sobol(4, 40)                      # assert(dims < 40), grids/sobol.lua:36
    assert e.value.code == -6
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.grid_sobol(10, 3, 2 ** 30 - 5)         # "Too many calls", grids/sobol.lua:317-324
    assert ctx.grid_sobol(0, 3).shape == (0, 3)    # empty grid


def test_sobol_golden_fixture(ctx):
    import json
    with open(os.path.join(GOLD, "sobol_kat.json")) as f:
        g = json.load(f)
    for case in g["cases"]:
        got = ctx.grid_sobol(case["size"], case["dims"], case["skip"], case.get("mins"), case.get("maxes"))
        want = np.array([[float.fromhex(h) for h in row] for row in case["rows_hex"]])
        assert np.array_equal(got[case["row0"]:case["row0"] + want.shape[0]], want)


def test_sobol_sharding_is_a_slice(ctx):
    whole = ctx.grid_sobol(4000, 32, 1)
    for lo, hi in [(0, 1000), (1000, 2500), (2500, 4000)]:
        assert np.array_equal(ctx.grid_sobol(hi - lo, 32, 1 + lo), whole[lo:hi])


def _splitmix_grid(size, dims, seed, row_offset):
    with np.errstate(over="ignore"):
        ctr = (np.arange(size, dtype=np.uint64)[:, None] + np.uint64(row_offset)) * np.uint64(dims) + \
            np.arange(dims, dtype=np.uint64)[None, :]
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (ctr + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def test_random_grid_counter_based(ctx, orc):
    got = ctx.grid_random(513, 64, seed=42, row_offset=7)
    assert np.array_equal(got, _splitmix_grid(513, 64, 42, 7))
    assert got.min() >= 0.0 and got.max() < 1.0
    mins, maxes = np.full(64, -2.0), np.linspace(1, 5, 64)
    assert np.array_equal(ctx.grid_random(100, 64, 3, 0, mins, maxes),
                          orc.c.affine(_splitmix_grid(100, 64, 3, 0), mins, maxes))
    # shards of one global grid
    whole = ctx.grid_random(300, 5, seed=9)
    assert np.array_equal(ctx.grid_random(100, 5, seed=9, row_offset=200), whole[200:])


def test_grid_remove_stable(ctx, orc):
    X = orc.c.sobol(1000, 7)
    ctx.grid_upload(X)
    for idx in (1, 500, 998, 1):  # first, middle, last, first again
        row = ctx.grid_remove(idx)
        assert np.array_equal(row, X[idx - 1])
        X = orc.c.remove_row(X, idx)
        assert ctx.grid_shape() == (X.shape[0], 7)
        assert np.array_equal(ctx.grid_download(), X)
    import bot7_amd
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.grid_remove(X.shape[0] + 1)


# ---- scores ---------------------------------------------------------------------------------------------------
def test_cb_bit_exact(ctx, orc):
    rng = np.random.default_rng(0)
    mu, var = rng.normal(size=5000), rng.random(5000) * 3
    var[:3] = [0.0, 1e-300, np.nan]
    for args in [(1.0, False, -1.0), (2.5, True, 1.0), (0.0, False, 1.0), (0.3, True, -1.0)]:
        a, b = ctx.cb_compute(mu, var, *args), orc.c.cb(mu, var, *args)
        assert np.array_equal(a, b, equal_nan=True)


def test_ei_matches_oracle_and_edge_cases(ctx, orc):
    rng = np.random.default_rng(1)
    mu, var = rng.normal(size=20000), rng.random(20000) * 2 + 1e-6
    got, want = ctx.ei_compute(mu, var, [-0.3], 0.01), orc.c.ei(mu, var, [-0.3], 0.01)
    # same op sequence; only exp() may differ by an ulp, which moves EI by a few ulp of its terms
    assert np.max(np.abs(got - want)) <= 8 * np.finfo(float).eps * (np.abs(mu).max() + 1.5)
    assert np.mean(got == want) > 0.5
    # reference edge cases: sigma = 0 / negative variance / NaN propagate identically
    mu_e = np.array([1.0, -1.0, 0.0, 0.0, np.nan, 0.2])
    var_e = np.array([0.0, 0.0, 0.0, -1.0, 1.0, np.inf])
    ge, we = ctx.ei_compute(mu_e, var_e, [0.0]), orc.c.ei(mu_e, var_e, [0.0])
    assert np.array_equal(np.isnan(ge), np.isnan(we)) and np.array_equal(ge[~np.isnan(ge)], we[~np.isnan(we)])
    assert ge[0] == 0.0 and ge[1] == 1.0
    # multi-column means (fantasy columns) take the row mean
    m2 = rng.normal(size=(300, 4))
    v2 = rng.random(300) + 0.1
    assert np.allclose(ctx.ei_compute(m2, v2, [0.1, 0.0, -0.2, 0.3]), orc.c.ei(m2, v2, [0.1, 0.0, -0.2, 0.3]),
                       rtol=0, atol=1e-15)


def test_argmax_semantics(ctx, orc):
    rng = np.random.default_rng(2)
    for n in (1, 2, 63, 64, 65, 257, 100003):
        s = rng.normal(size=n)
        assert ctx.argmax(s)[1] == orc.c.argmax_first(s)[0]
    s = rng.normal(size=70000)
    s[[123, 45000, 69999]] = s.max() + 1.0            # ties -> first occurrence
    assert ctx.argmax(s) == (s[123], 124)
    s[[50000, 60000]] = np.nan                        # first NaN wins
    v, i = ctx.argmax(s)
    assert np.isnan(v) and i == 50001 == orc.c.argmax_first(s)[0]
    assert ctx.argmax(np.full(1000, -np.inf))[1] == 1


# ---- GP fit / predict -----------------------------------------------------------------------------------------
CASES = [  # (d, N, M, objective) -- ragged on purpose: N not a multiple of 64/128, M not a multiple of 128
    (2, 2, 7, B.braninhoo), (2, 24, 256, B.braninhoo), (6, 65, 1000, B.hartmann6), (6, 256, 4096, B.hartmann6),
    (32, 129, 515, B.ackley), (5, 300, 2049, B.rastrigin), (39, 64, 128, B.rastrigin),
    (8, 2300, 600, B.rastrigin),   # 36 panels: more than 16 K-chunks per row block of the inline inverse
    (4, 4200, 300, B.rastrigin),   # Npad > 4096: the pair schedule of the Cholesky takes over
]


@pytest.mark.parametrize("d,N,M,obj", CASES)
def test_fit_and_predict_match_oracle(ctx, orc, d, N, M, obj):
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, d, N, M, obj)
    f = orc.gp.fit(X_obs, Y, **hyp)
    out = ctx.gp_fit(X_obs, Y, hyp["lenscale_sq"], hyp["amp"], hyp["noise"], hyp["mean"], want_nll=True)
    assert out["info"] == 0 and out["jitter"] == 0.0 and f.jitter == 0.0
    L, alpha, Linv = ctx.gp_download(N)
    assert np.allclose(L, f.L, rtol=1e-9, atol=1e-12 * np.sqrt(hyp["amp"]))
    assert np.array_equal(np.triu(L, 1), np.zeros_like(L))
    assert np.allclose(Linv @ f.L, np.eye(N), atol=1e-7)
    assert relerr(alpha, f.alpha, floor=np.abs(f.alpha).max() * 1e-3) < 1e-6
    assert out["nll"][0] == pytest.approx(float(f.nll[0]), rel=1e-9, abs=1e-7)
    ctx.grid_upload(X_hid)
    mu, var = ctx.gp_predict()
    mu_o, var_o = orc.gp.predict(f, X_hid)
    assert relerr(mu, mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL
    assert relerr(var, var_o) < REL, "posterior variance beyond 1e-5 relative"
    assert (var > 0).all() and (var <= hyp["amp"] * (1 + 1e-12)).all()
    # predict_at (X1 that is not the resident grid) gives the same numbers and leaves the grid alone
    mu2, var2 = ctx.gp_predict_at(X_hid[: min(M, 200)])
    assert np.array_equal(mu2, mu[: min(M, 200)]) and np.array_equal(var2, var[: min(M, 200)])
    assert ctx.grid_shape() == (M, d)


@pytest.mark.parametrize("N", [2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 383, 384, 385, 511, 513])
def test_fit_at_panel_boundaries(ctx, orc, N):
    """Every N around the 64-wide panel and 128-row padding boundaries: factor, inverse, alpha and a few predictions."""
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, 3, N, 40, B.rastrigin)
    f = orc.gp.fit(X_obs, Y, **hyp)
    out = ctx.gp_fit(X_obs, Y, **hyp)
    assert out["info"] == 0
    L, alpha, Linv = ctx.gp_download(N)
    assert np.allclose(L, f.L, rtol=1e-9, atol=1e-12 * np.sqrt(hyp["amp"]))
    assert np.allclose(Linv @ f.L, np.eye(N), atol=1e-7)
    assert relerr(alpha, f.alpha, floor=np.abs(f.alpha).max() * 1e-3) < 1e-6
    ctx.grid_upload(X_hid)
    mu, var = ctx.gp_predict()
    mu_o, var_o = orc.gp.predict(f, X_hid)
    assert relerr(mu, mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(var, var_o) < REL


def test_inline_inverse_matches_separate_pass(ctx, orc, monkeypatch):
    """inv(L) built inside the Cholesky launches (default) against the stand-alone recursive-doubling pass."""
    import bot7_amd
    X_obs, Y, _, hyp = make_problem(ctx, orc, 6, 1100, 64, B.hartmann6)
    monkeypatch.setenv("B7_INVERSE_INLINE", "0")
    ref = bot7_amd.Context(0, lib="diag")      # the switches exist in the diagnostic build only (tools/_build/libbot7hip_diag.so)
    monkeypatch.delenv("B7_INVERSE_INLINE")
    for c in (ctx, ref):
        c.profile_enable(True)
        c.profile_reset()
        c.gp_fit(X_obs, Y, **hyp)
    assert ctx.profile_get("trtri")[1] == 0 and ref.profile_get("trtri")[1] == 1     # (ms, count): the pass is gone
    (L0, a0, Li0), (L1, a1, Li1) = ctx.gp_download(1100), ref.gp_download(1100)
    assert np.array_equal(L0, L1)
    assert np.allclose(Li0, Li1, rtol=0, atol=1e-11 * np.abs(Li1).max())
    assert np.array_equal(np.triu(Li0, 1), np.zeros_like(Li0))
    assert relerr(a0, a1, floor=1e-3 * np.abs(a1).max()) < 1e-9
    ref.close() if hasattr(ref, "close") else None


@pytest.mark.parametrize("env", [{"B7_POTRF_SCHED": "0"}, {"B7_POTRF_SCHED": "0", "B7_POTRF_DEFER": "0"},
                                 {"B7_POTRF_SCHED": "0", "B7_SYRK_SMALL": "0", "B7_POTRF_GROUP": "4"},
                                 {"B7_DIAG_VARIANT": "0", "B7_POTRF_SCHED": "0", "B7_INVERSE_INLINE": "0"}])
def test_cholesky_schedules_agree(ctx, orc, monkeypatch, env):
    """The alternative Cholesky schedules / kernels kept behind environment switches give the same factor."""
    import bot7_amd
    X_obs, Y, _, hyp = make_problem(ctx, orc, 6, 700, 64, B.hartmann6)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    alt = bot7_amd.Context(0, lib="diag")
    for k in env:
        monkeypatch.delenv(k)
    ctx.gp_fit(X_obs, Y, **hyp)
    alt.gp_fit(X_obs, Y, **hyp)
    (L0, a0, Li0), (L1, a1, Li1) = ctx.gp_download(700), alt.gp_download(700)
    assert np.allclose(L0, L1, rtol=0, atol=1e-12 * np.abs(L1).max())
    assert np.allclose(Li0, Li1, rtol=0, atol=1e-10 * np.abs(Li1).max())
    assert relerr(a0, a1, floor=1e-3 * np.abs(a1).max()) < 1e-6   # alpha = K^-1 r: ulp-level L differences x cond(K)


def test_golden_gp_fixture(ctx, orc):
    g = np.load(os.path.join(GOLD, "gp_small.npz"))
    ctx.gp_fit(g["X_obs"], g["Y_obs"], g["lenscale_sq"], float(g["amp"]), float(g["noise"]), float(g["mean"]))
    ctx.grid_upload(g["X_hid"])
    mu, var = ctx.gp_predict()
    assert relerr(mu, g["mu"], floor=1e-3 * np.abs(g["mu"]).max()) < REL and relerr(var, g["var"]) < REL
    ctx.score_reset()
    ctx.score_ei([float(g["Y_obs"].min())], 0.0)
    v, idx, ei = ctx.score_finish(1.0, download=True)
    assert idx == int(g["ei_argmax1"]) and np.allclose(ei, g["ei"], rtol=1e-6, atol=1e-12)
    ctx.score_reset()
    ctx.score_cb()
    v, idx, cb = ctx.score_finish(1.0, download=True)
    assert idx == int(g["cb_argmax1"]) and np.allclose(cb, g["cb"], rtol=1e-6, atol=1e-12)


def test_chol_jitter_schedule_matches_reference(ctx, orc):
    """b7_chol = utils.math.chol (utils/math.lua:159-218).  Matrices that are NOT positive definite by a wide
    margin make the failing pivot and the eps schedule deterministic: first retry 1.1e-8, eps on the original
    matrix, first success once eps exceeds |lambda_min|."""
    rng = np.random.default_rng(5)
    for n in (3, 64, 65, 200):
        # PD case: no jitter, factor matches LAPACK
        B_ = rng.normal(size=(n, n))
        A = B_ @ B_.T + n * np.eye(n)
        L, jit, info = ctx.chol(A)
        Lo, jo, io = orc.gp.chol_jitter(A)
        assert (jit, info) == (0.0, 0) == (jo, io) and np.allclose(L, Lo, rtol=1e-10, atol=1e-12)
        assert np.array_equal(np.triu(L, 1), np.zeros_like(L))
        # indefinite by a fixed margin: shift a PD matrix down by more than its smallest eigenvalue
        w = np.linalg.eigvalsh(A)
        shift = w[0] + 0.37
        Ai = A - shift * np.eye(n)
        L, jit, info = ctx.chol(Ai)
        Lo, jo, io = orc.gp.chol_jitter(Ai)
        assert info > 0 and io > 0
        assert jit == jo, "jitter schedule diverged from the reference arithmetic"
        k = np.log(jit / 1e-8) / np.log(1.1)
        assert abs(k - round(k)) < 1e-6 and 0.37 < jit <= 0.37 * 1.1 + 1e-12
        assert np.allclose(L @ L.T, Ai + jit * np.eye(n), atol=1e-8 * n)
    # rank-1: dpotrf stops at pivot 2; both report it
    v = np.arange(1.0, 9.0)[None, :]
    L, jit, info = ctx.chol(v.T @ v)
    Lo, jo, io = orc.gp.chol_jitter(v.T @ v)
    assert info == io == 2 and jit > 0
    import bot7_amd
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.chol(np.full((4, 4), np.nan))   # the reference would loop forever (eps > NaN is false)


def test_gp_fit_survives_duplicate_observations(ctx, orc):
    """Duplicate observations with zero noise make K exactly singular; whether a pivot lands at +1e-17 or -1e-17
    is rounding noise (LAPACK's blocking decides it on the CPU as ours does here), so only the contract is
    checked: either no failure and no jitter, or a reported pivot and an eps from the schedule."""
    X = orc.c.sobol(40, 3)
    X[17] = X[5]
    X[33] = X[5]
    Y = B.rastrigin(X)
    ls = np.full(3, 0.5)
    out = ctx.gp_fit(X, Y, ls, 1.0, 0.0, 0.0)
    if out["info"] > 0:
        k = np.log(out["jitter"] / 1e-8) / np.log(1.1)
        assert out["jitter"] > 0 and abs(k - round(k)) < 1e-6
        L, _, _ = ctx.gp_download(40)
        assert np.allclose(L @ L.T, orc.gp.ardse(X, None, ls, 1.0) + out["jitter"] * np.eye(40), atol=1e-9)
    else:
        assert out["jitter"] == 0.0


def test_error_conventions(ctx, orc):
    import bot7_amd
    c2 = bot7_amd.Context(0)
    with pytest.raises(bot7_amd.Bot7HipError) as e:
        c2.gp_predict()
    assert e.value.code == -4                                   # predict before fit
    X = orc.c.sobol(10, 3)
    with pytest.raises(bot7_amd.Bot7HipError):
        c2.gp_fit(X, np.zeros(10), [1.0, -1.0, 1.0], 1.0, 0.0, 0.0)   # lenscale must be positive
    c2.gp_fit(X, np.arange(10.0), np.ones(3), 1.0, 1e-3, 0.0)
    c2.grid_upload(orc.c.sobol(20, 4))
    with pytest.raises(bot7_amd.Bot7HipError):
        c2.gp_predict()                                          # dims mismatch
    c2.grid_upload(orc.c.sobol(20, 3))
    with pytest.raises(bot7_amd.Bot7HipError) as e:
        c2.score_ei([0.0])
    assert e.value.code == -4                                    # score before predict
    c2.close()


# ---- end to end: marginalised score + arg-max --------------------------------------------------------------------
def _oracle_nominate(orc, X_obs, Y, X_hid, hyps, kind):
    acc = np.zeros(X_hid.shape[0])
    for h in hyps:
        f = orc.gp.fit(X_obs, Y, **h)
        mu, var = orc.gp.predict(f, X_hid)
        s = orc.c.ei(mu, var, [float(Y.min())]) if kind == "ei" else orc.c.cb(mu, var)
        orc.c.accumulate(acc, s)
    orc.c.divide(acc, float(len(hyps)))
    return acc


def _hip_nominate(ctx, X_obs, Y, hyps, kind):
    first = True
    for h in hyps:
        ctx.gp_fit(X_obs, Y, h["lenscale_sq"], h["amp"], h["noise"], h["mean"])
        ctx.gp_predict(download=False)
        if first:
            ctx.score_reset()
            first = False
        if kind == "ei":
            ctx.score_ei([float(Y.min())], 0.0)
        else:
            ctx.score_cb()
    return ctx.score_finish(float(len(hyps)), download=True)


@pytest.mark.parametrize("kind,d,N,M,obj,S", [("ei", 2, 24, 256, B.braninhoo, 3), ("cb", 6, 256, 32768, B.hartmann6, 1),
                                              ("ei", 32, 192, 8192, B.ackley, 2)])
def test_marginalised_argmax_matches_oracle(ctx, orc, kind, d, N, M, obj, S):
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, d, N, M, obj)
    hyps = []
    for s in range(S):  # S different hyper samples, as the slice sampler would hand over
        h = dict(hyp)
        h["lenscale_sq"] = hyp["lenscale_sq"] * (1.0 + 0.25 * s)
        h["amp"] = hyp["amp"] * (1.0 + 0.1 * s)
        hyps.append(h)
    ctx.grid_upload(X_hid)
    val, idx, scores = _hip_nominate(ctx, X_obs, Y, hyps, kind)
    want = _oracle_nominate(orc, X_obs, Y, X_hid, hyps, kind)
    widx, wval = orc.c.argmax_first(want)
    top2 = np.sort(want)[-2:]
    gap = float(top2[1] - top2[0])
    err = float(np.max(np.abs(scores - want)))
    print("top-2 gap %.3e, max |score diff| %.3e" % (gap, err))
    assert err < 1e-6 * max(1.0, np.abs(want).max())
    assert gap > 4 * err, "synthetic case is degenerate: arg-max not separable at the achieved accuracy"
    assert idx == widx, "arg-max index differs from the CPU path"
    assert val == scores[idx - 1]


@pytest.mark.parametrize("N,d,M", [(700, 6, 70000), (100, 3, 66000), (2048, 32, 65536 + 256), (2, 2, 65536), (129, 3, 65537),
                                   (300, 6, 66000), (513, 4, 65700), (1000, 16, 65536), (1500, 6, 66001), (4096, 8, 65536), (2304, 5, 65600)])
def test_posterior_grid_shapes_are_bit_identical(ctx, orc, N, d, M):
    """The large-grid shapes (n-tiles of 256 rows when the padded N is a multiple of 256 -- N = 700, 2048 --, otherwise
    256-candidate workgroups on 128-row n-tiles -- N = 100) and the small-grid shape (128-candidate workgroups) give the
    same variance bits for the same candidate, and the oracle's values on a sample."""
    X_obs, Y, X_hid, hyp = make_problem(None, orc, d, N, M, lambda X: np.sin(3.0 * X).sum(axis=1, keepdims=True))
    ctx.grid_upload(X_hid)
    ctx.gp_fit(X_obs, Y, **hyp)
    mu, var = ctx.gp_predict()
    ctx.grid_upload(X_hid[:4096])           # fewer than one 256-candidate workgroup per CU
    mu_s, var_s = ctx.gp_predict()
    assert np.array_equal(var[:4096], var_s) and np.array_equal(mu[:4096], mu_s)
    f = orc.gp.fit(X_obs, Y, **hyp)
    idx = np.linspace(0, M - 1, 300).astype(int)
    mu_o, var_o = orc.gp.predict(f, X_hid[idx])
    assert relerr(var[idx], var_o) < REL


def test_posterior_kernels_reproduce_the_host_model_bit_for_bit():
    """tools/post_probe.hip: both workgroup shapes of the library's kernel, and the eight-wave kernel of earlier
    rounds (tools/post_kernel_w8.h, not shipped), launched directly on synthetic L^-1 / K* operands against the host
    model of their arithmetic -- v[n] = ascending fma chain over k (what v_mfma_f64_16x16x4_f64 does, tools/
    mfma_acc_probe.hip), squares folded in the documented order -- for one tile, a ragged N and several tiles."""
    import subprocess
    from bot7_amd import build
    exe = build.build_post_probe()
    for args in (("128", "5"), ("128", "100"), ("256", "200"), ("768", "700"), ("2048", "2048")):
        out = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        assert "differing from the host model: w4<4> 0, w4<2> 0, w8 0, tall 0" in out.stdout, out.stdout


def test_posterior_and_ei_against_50_digit_arithmetic(ctx):
    """The HIP path against textbook GP regression and closed-form EI evaluated in 50-digit arithmetic (mpmath), with no
    oracle in between: posterior mean and variance of a small well-conditioned problem, the likelihood, and EI with the
    reference's A&S erf replaced by the exact one (so EI agrees to the polynomial's 1.5e-7, utils/math.lua:261-288)."""
    import mpmath
    mpmath.mp.dps = 50
    try:
        rng = np.random.default_rng(11)
        N, M, d = 24, 40, 3
        X, Xs = rng.random((N, d)), rng.random((M, d))
        Y = np.sin(3 * X.sum(1)).reshape(-1, 1)
        ls, amp, noise, mean = np.array([0.4, 0.7, 0.3]), 1.3, 1e-3, 0.2
        ctx.grid_upload(Xs)
        rep = ctx.gp_fit(X, Y, ls, amp, noise, mean, want_nll=True)
        mu, var = ctx.gp_predict()
        ctx.score_reset()
        ctx.score_ei([float(Y.min())], 0.0)
        _, _, ei = ctx.score_finish(1.0, download=True)

        def k(a, b):
            return mpmath.mpf(amp) * mpmath.exp(-sum((mpmath.mpf(a[i]) - mpmath.mpf(b[i])) ** 2 / mpmath.mpf(ls[i])
                                                     for i in range(d)) / 2)
        K = mpmath.matrix(N, N)
        for i in range(N):
            for j in range(N):
                K[i, j] = k(X[i], X[j]) + (mpmath.mpf(noise) if i == j else 0)
        Ki = K ** -1
        r = mpmath.matrix([mpmath.mpf(float(y)) - mean for y in Y[:, 0]])
        nll = (r.T * Ki * r)[0] / 2 + mpmath.log(mpmath.det(K)) / 2 + N * mpmath.log(2 * mpmath.pi) / 2
        assert float(abs(rep["nll"][0] - nll) / abs(nll)) < 1e-10
        fmin = mpmath.mpf(float(Y.min()))
        for j in range(M):
            ks = mpmath.matrix([k(Xs[j], X[i]) for i in range(N)])
            mu_j = mean + (ks.T * Ki * r)[0]
            var_j = mpmath.mpf(amp) - (ks.T * Ki * ks)[0]
            assert float(abs(mu[j, 0] - mu_j)) < 1e-9
            assert float(abs(var[j] - var_j) / var_j) < 1e-8
            sd = mpmath.sqrt(var_j)
            z = (fmin - mu_j) / sd
            ei_j = max(mpmath.mpf(0), (fmin - mu_j) * mpmath.ncdf(z) + sd * mpmath.npdf(z))
            assert float(abs(ei[j] - ei_j)) < 4e-7 * max(1.0, float(abs(fmin - mu_j)))
    finally:
        mpmath.mp.dps = 15


def _marg_hyps(hyp, S):
    hyps = []
    for s in range(S):
        h = dict(hyp)
        h["lenscale_sq"] = hyp["lenscale_sq"] * (1.0 + 0.25 * s)
        h["amp"] = hyp["amp"] * (1.0 + 0.1 * s)
        hyps.append(h)
    return hyps


@pytest.mark.parametrize("kind,d,N,M,obj,S", [("ei", 2, 24, 256, B.braninhoo, 3), ("cb", 6, 256, 32768, B.hartmann6, 4),
                                              ("ei", 32, 192, 8192, B.ackley, 10), ("ei", 6, 2100, 1500, B.hartmann6, 2)])
def test_eval_nominate_is_the_per_sample_loop_in_one_call(ctx, orc, kind, d, N, M, obj, S):
    """b7_eval_nominate (bots/bayesopt.lua:56-99 as one call, one host synchronisation) against the separate entry
    points: same winner, same value, the same accumulator bit for bit -- and the oracle's winner.  N = 2100 runs the
    launch schedule (Npad > 2048), the others the persistent one."""
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, d, N, M, obj)
    hyps = _marg_hyps(hyp, S)
    ctx.grid_upload(X_hid)
    val0, idx0, scores0 = _hip_nominate(ctx, X_obs, Y, hyps, kind)
    ctx.gp_set_data(X_obs, Y)
    kw = {"score": "ei", "fmin": [float(Y.min())]} if kind == "ei" else {"score": "cb"}
    val1, idx1, rep = ctx.eval_nominate(hyps, want_report=True, **kw)
    _, _, scores1 = ctx.score_finish(1.0, download=True)       # the accumulator holds score / S
    assert (val1, idx1) == (val0, idx0)
    assert np.array_equal(scores1, scores0)
    assert not rep["jitter"].any() and not rep["info"].any()
    if N <= 300:
        want = _oracle_nominate(orc, X_obs, Y, X_hid, hyps, kind)
        assert idx1 == orc.c.argmax_first(want)[0]
    # a row offset only shifts the index (what a shard passes)
    assert ctx.eval_nominate(hyps, global_row_offset=1000, **kw) == (val0, idx0 + 1000)


def test_eval_nominate_more_hyper_samples_than_fit_side_by_side(ctx, orc):
    """S = 70 hyper samples: the fits run side by side in persistent launches of at most CUs/4 = 64 fits (one critical
    workgroup and three or more helpers each), so this takes two launches; same bits as seventy separate fits."""
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, 3, 100, 700, lambda X: np.sin(2.0 * X).sum(axis=1, keepdims=True))
    hyps = [dict(hyp, lenscale_sq=hyp["lenscale_sq"] * (0.7 + 0.01 * s), amp=hyp["amp"] * (1.0 + 0.003 * s)) for s in range(70)]
    ctx.grid_upload(X_hid)
    val0, idx0, scores0 = _hip_nominate(ctx, X_obs, Y, hyps, "ei")
    ctx.gp_set_data(X_obs, Y)
    val1, idx1 = ctx.eval_nominate(hyps, score="ei", fmin=[float(Y.min())])
    _, _, scores1 = ctx.score_finish(1.0, download=True)
    assert (val1, idx1) == (val0, idx0) and np.array_equal(scores1, scores0)
    # the context's own fit slot holds none of the batch's fits: a posterior needs a new fit
    import bot7_amd
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.gp_predict()


@pytest.mark.parametrize("seed", range(8))
def test_eval_nominate_random_shapes(ctx, orc, seed):
    """Seeded random (N, d, M, S, score): ragged N incl. panel boundaries, tiny and mid-sized grids, one to seven hyper
    samples -- the one-call path (fits side by side) against the separate calls, bit for bit."""
    rng = np.random.default_rng(100 + seed)
    N = int(rng.choice([2, 3, 17, 63, 64, 65, 127, 128, 129, 200, 257, 300]))
    d = int(rng.integers(1, 9))
    M = int(rng.choice([64, 100, 257, 1000, 4097]))
    S = int(rng.integers(1, 8))
    kind = "ei" if rng.random() < 0.5 else "cb"
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, d, N, M, lambda X: np.cos(2.5 * X).sum(axis=1, keepdims=True) + X[:, :1] ** 2)
    hyps = [dict(hyp, lenscale_sq=hyp["lenscale_sq"] * float(rng.uniform(0.6, 1.6)), amp=hyp["amp"] * float(rng.uniform(0.8, 1.3)),
                 noise=hyp["noise"] * float(rng.uniform(0.5, 20.0)), mean=hyp["mean"] + float(rng.normal(scale=0.05)))
            for _ in range(S)]
    ctx.grid_upload(X_hid)
    val0, idx0, scores0 = _hip_nominate(ctx, X_obs, Y, hyps, kind)
    ctx.gp_set_data(X_obs, Y)
    kw = {"score": "ei", "fmin": [float(Y.min())]} if kind == "ei" else {"score": "cb"}
    val1, idx1 = ctx.eval_nominate(hyps, **kw)
    _, _, scores1 = ctx.score_finish(1.0, download=True)
    assert (val1, idx1) == (val0, idx0), (N, d, M, S, kind)
    assert np.array_equal(scores1, scores0), (N, d, M, S, kind)


def test_eval_nominate_with_fantasy_columns(ctx, orc):
    """c > 1 response columns (the fantasies of scores/expected_improvement.lua:51-60): EI is the row mean over the
    columns (:83-85); one call against the separate entry points and against the oracle."""
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, 4, 150, 3000, lambda X: np.cos(2.0 * X).sum(axis=1, keepdims=True))
    rng = np.random.default_rng(4)
    Y3 = np.hstack([Y, Y + 0.05 * rng.normal(size=Y.shape), Y - 0.05 * rng.normal(size=Y.shape)])
    hyps = _marg_hyps(hyp, 3)
    fmin = Y3.min(axis=0)
    ctx.grid_upload(X_hid)
    first = True
    for h in hyps:
        ctx.gp_fit(X_obs, Y3, h["lenscale_sq"], h["amp"], h["noise"], h["mean"])
        ctx.gp_predict(download=False)
        if first:
            ctx.score_reset()
            first = False
        ctx.score_ei(fmin, 0.0)
    val0, idx0, scores0 = ctx.score_finish(3.0, download=True)
    ctx.gp_set_data(X_obs, Y3)
    val1, idx1 = ctx.eval_nominate(hyps, score="ei", fmin=fmin)
    _, _, scores1 = ctx.score_finish(1.0, download=True)
    assert (val1, idx1) == (val0, idx0) and np.array_equal(scores1, scores0)
    acc = np.zeros(X_hid.shape[0])
    for h in hyps:
        mu, var = orc.gp.predict(orc.gp.fit(X_obs, Y3, **h), X_hid)
        orc.c.accumulate(acc, orc.c.ei(mu, var, fmin))
    orc.c.divide(acc, 3.0)
    assert np.max(np.abs(scores1 - acc)) < 1e-6 * max(1.0, np.abs(acc).max())
    assert idx1 == orc.c.argmax_first(acc)[0]


def test_eval_nominate_redoes_the_nomination_when_a_pivot_fails(ctx, orc):
    """One of the S hyper samples makes K singular (duplicate rows, no noise): its report says so after the fact, the
    speculative scores are thrown away and the nomination is redone with utils/math.lua:159-218's jitter schedule --
    the result the separate calls give."""
    X = orc.c.sobol(300, 3, 1)
    X[7] = X[3]
    X[250] = X[100]
    Y = np.sin(3.0 * X).sum(axis=1, keepdims=True)
    X_hid = orc.c.sobol(4000, 3, 400)
    good = dict(lenscale_sq=np.full(3, 0.4), amp=1.0, noise=1e-3, mean=0.1)
    bad = dict(lenscale_sq=np.full(3, 0.4), amp=1.0, noise=0.0, mean=0.0)
    hyps = [good, bad, dict(good, amp=1.3)]
    ctx.grid_upload(X_hid)
    for kind in ("ei", "cb"):
        val0, idx0, scores0 = _hip_nominate(ctx, X, Y, hyps, kind)
        ctx.gp_set_data(X, Y)
        kw = {"score": "ei", "fmin": [float(Y.min())]} if kind == "ei" else {"score": "cb"}
        val1, idx1, rep = ctx.eval_nominate(hyps, want_report=True, **kw)
        _, _, scores1 = ctx.score_finish(1.0, download=True)
        assert (val1, idx1) == (val0, idx0) and np.array_equal(scores1, scores0)
        assert rep["jitter"][1] > 0 and rep["info"][1] > 0 and rep["jitter"][0] == 0 == rep["jitter"][2]


def test_eval_nominate_survives_a_hand_off_time_out(orc, monkeypatch):
    """Fault injection as in test_persistent_cholesky_times_out_into_the_launch_schedule, through the one-call path:
    the speculative pass sees the abort in the reports, the redo runs the launch schedule, the result is unchanged."""
    monkeypatch.setenv("B7_PERSIST_FAULT", "2")
    launch, persist = _two_schedules(monkeypatch)
    monkeypatch.delenv("B7_PERSIST_FAULT")
    try:
        X_obs, Y, X_hid, hyp = make_problem(None, orc, 6, 500, 3000, B.hartmann6)
        hyps = _marg_hyps(hyp, 3)
        out = []
        for c in (launch, persist):
            c.grid_upload(X_hid)
            c.gp_set_data(X_obs, Y)
            v, i = c.eval_nominate(hyps, score="ei", fmin=[float(Y.min())])
            out.append((v, i, c.score_finish(1.0, download=True)[2]))
        assert out[0][:2] == out[1][:2] and np.array_equal(out[0][2], out[1][2])
        assert _aborts(persist) >= 1 and _aborts(launch) == 0
    finally:
        launch.close()
        persist.close()


def test_eval_nominate_argument_errors(ctx, orc):
    import bot7_amd
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, 3, 40, 100, lambda X: np.sin(X).sum(axis=1, keepdims=True))
    ctx.grid_upload(X_hid)
    ctx.gp_set_data(X_obs, Y)
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.eval_nominate([], score="cb")                                    # S >= 1
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.eval_nominate([hyp], score="ei")                                 # EI needs fmin
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.eval_nominate([dict(hyp, amp=-1.0)], score="cb")                 # amp > 0
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.eval_nominate([hyp], score="cb", global_row_offset=-1)
    ctx.grid_upload(orc.c.sobol(50, 5))
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.eval_nominate([hyp], score="cb")                                 # grid dims != data dims
    ctx.grid_upload(X_hid)
    v, i = ctx.eval_nominate([hyp], score="cb")
    assert 1 <= i <= 100 and np.isfinite(v)


def test_bayesopt_driver_cfg1_plumbing(ctx, orc):
    """BASELINE config 1: braninhoo 2-D, GP+EI, 256-point grid, 25 trials, through the bots.bayesopt mirror;
    every nomination is re-derived with the oracle from the same observed set."""
    import bot7_amd
    from harness import bots

    class H(object):
        def __init__(self, name):
            self.name, self.min, self.max, self.size = name, 0.0, 1.0, 1

    grid = bot7_amd.grids.random({"size": 256, "dims": 2, "seed": 5, "mins": np.zeros(2), "maxes": np.ones(2)},
                                 context=ctx)()
    cfg = {"bot": {"verbose": 0, "budget": 25, "nInitial": 2, "nSamples": 1, "seed": 1},
           "grid": {"type": "random", "size": 256, "dims": 2}, "score": {"type": "expected_improvement"}}
    model = bot7_amd.models.gp_regressor({}, context=ctx)
    bot = bots.bayesopt(B.braninhoo, [H("x1"), H("x2")], cfg, cache={"candidates": grid, "model": model})
    host_cand = np.asarray(grid).copy()
    for t in range(1, 26):
        cand_before = np.asarray(bot.candidates).copy()
        obs_before = None if bot.observed is None else bot.observed.copy()
        resp_before = None if bot.responses is None else bot.responses.copy()
        x, y = bot.run_trial()
        bot.update_best(x, y)
        assert np.array_equal(cand_before, host_cand)
        if t > 2:  # model-based nomination: same pick as the oracle on the same data
            h = model.hyp
            f = orc.gp.fit(obs_before, resp_before, h["lenscale_sq"], h["amp"], h["noise"], h["mean"])
            mu, var = orc.gp.predict(f, cand_before)
            ei = orc.c.ei(mu, var, [float(resp_before.min())])
            widx = orc.c.argmax_first(ei)[0]
            assert np.array_equal(x, cand_before[widx - 1]), "trial %d nominated a different candidate" % t
        # stable deletion keeps host and device candidate sets identical
        row = np.where((host_cand == x).all(axis=1))[0][0]
        host_cand = np.delete(host_cand, row, axis=0)
        assert np.array_equal(ctx.grid_download(), host_cand)
    assert bot.observed.shape == (25, 2) and bot.responses.shape == (25, 1)
    assert float(bot.best["y"].ravel()[0]) == float(bot.responses.min())


# ---- full BASELINE sizes: size-independent properties ----------------------------------------------------------
def test_full_size_metric_config_properties(ctx, orc):
    """N = 2048, d = 32 (the metric's configuration), M = 65536 + ragged tail.
    (1) bit-reproducible; (2) chunking-independent (a checksum of checksums over shards);
    (3) 0 < var <= amp; (4) a random sample of candidates agrees with the oracle to 1e-5;
    (5) arg-max equals the arg-max of the downloaded scores with TH semantics, and -- the pool starts at bench.SOBOL_SKIP, so
    (0.5, ..., 0.5), ackley's exact minimum, is NOT an observation and EI is not identically zero -- the winner is a
    candidate with positive EI that the oracle re-derives over a window of rows around it."""
    import bench
    d, N, M = 32, 2048, 65536 + 77
    pool = ctx.grid_sobol(M + N, d, bench.SOBOL_SKIP)
    step = (M + N) // N
    obs_idx = np.arange(N) * step
    mask = np.ones(M + N, dtype=bool)
    mask[obs_idx] = False
    X_obs, X_hid = pool[obs_idx].copy(), pool[mask].copy()
    Y = B.ackley(X_obs)
    amp = float(np.var(Y))
    hyp = dict(lenscale_sq=np.full(d, d / 8.0), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y)))
    ctx.gp_fit(X_obs, Y, **hyp)
    ctx.grid_upload(X_hid)
    mu, var = ctx.gp_predict()
    ctx.score_reset()
    ctx.score_ei([float(Y.min())], 0.0)
    val, idx, ei = ctx.score_finish(1.0, download=True)
    # (1)
    ctx.gp_fit(X_obs, Y, **hyp)
    mu_b, var_b = ctx.gp_predict()
    assert np.array_equal(mu, mu_b) and np.array_equal(var, var_b)
    # (2) small workspace -> many chunks; and two half-grids
    ctx.set_workspace(8 * 2048 * 1024)
    mu_c, var_c = ctx.gp_predict()
    ctx.set_workspace(4 << 30)
    assert np.array_equal(mu, mu_c) and np.array_equal(var, var_c)
    half = 33000  # not a multiple of the tile
    ctx.grid_upload(X_hid[:half])
    mu_1, var_1 = ctx.gp_predict()
    ctx.grid_upload(X_hid[half:])
    mu_2, var_2 = ctx.gp_predict()
    assert np.array_equal(np.concatenate([var_1, var_2]), var) and np.array_equal(np.concatenate([mu_1, mu_2]), mu)
    # (3)
    assert (var > 0).all() and (var <= amp * (1 + 1e-12)).all() and np.isfinite(mu).all()
    # (4)
    f = orc.gp.fit(X_obs, Y, **hyp)
    sample = np.random.default_rng(0).choice(M, 512, replace=False)
    mu_o, var_o = orc.gp.predict(f, X_hid[sample])
    assert relerr(mu[sample], mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL
    assert relerr(var[sample], var_o) < REL
    # (5)
    widx, wval = orc.c.argmax_first(ei)
    assert idx == widx and val == wval
    assert val > 0.0 and np.count_nonzero(ei == val) == 1, "the arg-max is a tie-break, not a winner"
    w0 = int(min(max(0, idx - 1 - 2048), M - 4096))
    mu_w, var_w = orc.gp.predict(f, X_hid[w0:w0 + 4096])
    ei_w = orc.c.ei(mu_w, var_w, [float(Y.min())])
    oidx, oval = orc.c.argmax_first(ei_w)
    top2 = np.partition(ei_w, -2)[-2:]
    print("full-size winner %d, EI %.3e, oracle top-2 gap in its window %.3e, |EI - oracle| %.1e" % (idx, val, top2[1] - top2[0], abs(val - oval)))
    assert oidx + w0 == idx and abs(val - oval) <= 1e-7 * max(abs(oval), 1e-300) + 1e-3 * (top2[1] - top2[0])


# ---- hyper sampling: slice sampler on the host, every density evaluation a device fit ---------------
def test_nll_and_slice_sampled_hypers(ctx, orc):
    import bot7_amd
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, 6, 48, 512, B.hartmann6)
    model = bot7_amd.models.gp_regressor({"sample": True, "nBurnin": 3, "seed": 11}, context=ctx)
    # the density the sampler walks is -NLL of the oracle (flat prior inside the bounds)
    for scale in (0.5, 1.0, 3.0):
        h = dict(hyp, lenscale_sq=hyp["lenscale_sq"] * scale)
        f = orc.gp.fit(X_obs, Y, **h)
        lp = model.log_posterior(model._to_theta(h), X_obs, Y)
        assert lp == pytest.approx(-float(f.nll[0]), rel=1e-9, abs=1e-7)
    lo, hi = model._bounds(X_obs, Y)
    assert model.log_posterior(lo - 1.0, X_obs, Y) == -np.inf
    # a deliberately poor start; burn-in then 6 per-sample updates as bots/bayesopt.lua:68,73-75 issues them
    model.hyp = dict(hyp, lenscale_sq=hyp["lenscale_sq"] * 40.0, noise=hyp["amp"] * 0.5)
    lp0 = model.log_posterior(model._to_theta(model.hyp), X_obs, Y)
    model.sample_hypers(X_obs, Y)
    lps, thetas = [], []
    for _ in range(6):
        v = model.sample_hypers(X_obs, Y, None, None, True)
        h = model.parse_hypers(v)
        t = model._to_theta(h)
        assert (t >= lo).all() and (t <= hi).all()
        thetas.append(t)
        lps.append(model.log_posterior(t, X_obs, Y))
    assert max(lps) > lp0, "nine slice updates never left a start chosen to be poor"
    assert len({tuple(np.round(t, 12)) for t in thetas}) > 1, "chain did not move"
    assert model.nEvals > 20
    # same seed, same chain
    m2 = bot7_amd.models.gp_regressor({"sample": True, "nBurnin": 3, "seed": 11}, context=ctx)
    m2.hyp = dict(hyp, lenscale_sq=hyp["lenscale_sq"] * 40.0, noise=hyp["amp"] * 0.5)
    m2.sample_hypers(X_obs, Y)
    v2 = m2.sample_hypers(X_obs, Y, None, None, True)
    assert np.allclose(m2._to_theta(m2.parse_hypers(v2)), thetas[0], rtol=1e-9, atol=1e-9)


def test_bayesopt_with_sampled_hypers_runs(ctx, orc):
    """The reference-faithful loop: nSamples hyper draws per nomination, scores marginalised on the device."""
    import bot7_amd

    class H(object):
        def __init__(self, name):
            self.name, self.min, self.max, self.size = name, 0.0, 1.0, 1

    cfg = {"bot": {"verbose": 0, "budget": 8, "nInitial": 3, "nSamples": 3, "seed": 2},
           "grid": {"type": "sobol", "size": 400, "dims": 2}, "score": {"type": "confidence_bound"},
           "model": {"type": "gp_regressor", "sample": True, "nBurnin": 2, "seed": 5}}
    bot = bots.bayesopt(B.braninhoo, [H("x1"), H("x2")], cfg)
    bot.model._ctx = ctx
    bot.candidates = bot7_amd.grids.sobol(bot.config["grid"], context=ctx)()
    best = bot.run_experiment()
    assert bot.observed.shape == (8, 2) and bot.candidates.shape == (392, 2)
    assert np.isfinite(best["y"]).all() and float(best["y"].ravel()[0]) == float(bot.responses.min())
    assert np.array_equal(ctx.grid_download(), np.asarray(bot.candidates))


@pytest.mark.gpu
@pytest.mark.parametrize("n_initial, budget", [(60, 69), (124, 132)])
def test_sampled_hyper_loop_follows_the_oracle_driven_sampler_across_the_likelihood_paths(ctx, orc, n_initial, budget):
    """model:sample_hypers (bots/bayesopt.lua:68,73-75) over samplers/slice.lua, whole trial loops whose observation count
    crosses 64 (the one-workgroup likelihood kernel goes from one 64-block to two) and 128 (it hands over to the general
    path): the same loop with every density evaluation answered by the ORACLE's likelihood instead must draw the same hypers
    (to 1e-7: a slice sampler is continuous in its density away from accept / reject ties) and nominate the same candidates.
    Oracle: parity unpinned (no reference fixture for the GP algebra)."""
    import bot7_amd

    class H(object):
        def __init__(self, name):
            self.name, self.min, self.max, self.size = name, 0.0, 1.0, 1

    def run(oracle_density):
        cfg = {"bot": {"verbose": 0, "budget": budget, "nInitial": n_initial, "nSamples": 2, "seed": 4},
               "grid": {"type": "sobol", "size": 600, "dims": 2}, "score": {"type": "expected_improvement"},
               "model": {"type": "gp_regressor", "sample": True, "nBurnin": 1, "seed": 6}}
        bot = bots.bayesopt(B.braninhoo, [H("x1"), H("x2")], cfg)
        bot.model._ctx = ctx
        draws = []
        if oracle_density:
            bot.model.nll = lambda X, Y, hyp=None: orc.gp.fit(np.atleast_2d(X), np.asarray(Y).reshape(len(X), -1),
                                                              **(hyp or bot.model.hyp)).nll
        inner = bot.model.sample_hypers

        def logged(*a, **k):
            v = inner(*a, **k)
            draws.append(np.array(v, dtype=np.float64))
            return v
        bot.model.sample_hypers = logged
        bot.candidates = bot7_amd.grids.sobol(bot.config["grid"], context=ctx)()
        bot.run_experiment()
        return np.asarray(bot.observed).copy(), np.asarray(bot.responses).copy(), draws

    obs_dev, resp_dev, draws_dev = run(False)
    obs_orc, resp_orc, draws_orc = run(True)
    assert obs_dev.shape == (budget, 2) and len(draws_dev) == len(draws_orc) and len(draws_dev) >= 3 * (budget - n_initial)
    for k, (a, b) in enumerate(zip(draws_dev, draws_orc)):
        assert np.allclose(a, b, rtol=1e-7, atol=1e-12), "hyper draw %d left the oracle-driven chain" % k
    assert np.array_equal(obs_dev, obs_orc) and np.array_equal(resp_dev, resp_orc)


# ---- fantasy columns and pending points ----------------------------------------------------------------
def test_multi_column_fit_predict_and_ei(ctx, orc):
    """Y with c columns: K, L shared, alpha N x c, mean M x c, EI row-averaged (scores/expected_improvement.lua:83-85)."""
    X_obs, Y1, X_hid, hyp = make_problem(ctx, orc, 6, 70, 900, B.hartmann6)
    rng = np.random.default_rng(3)
    for c in (2, 7, 100):
        Y = Y1 + 0.1 * rng.normal(size=(70, c))
        f = orc.gp.fit(X_obs, Y, **hyp)
        out = ctx.gp_fit(X_obs, Y, hyp["lenscale_sq"], hyp["amp"], hyp["noise"], hyp["mean"], want_nll=True)
        assert np.allclose(out["nll"], f.nll, rtol=1e-9, atol=1e-7)
        _, alpha, _ = ctx.gp_download(70, c)
        assert np.allclose(alpha, f.alpha, rtol=1e-6, atol=1e-6 * np.abs(f.alpha).max())
        ctx.grid_upload(X_hid)
        mu, var = ctx.gp_predict()
        mu_o, var_o = orc.gp.predict(f, X_hid)
        assert mu.shape == (900, c)
        assert relerr(mu, mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(var, var_o) < REL
        fmin = Y.min(axis=0)
        ctx.score_reset()
        ctx.score_ei(fmin, 0.0)
        _, idx, ei = ctx.score_finish(1.0, download=True)
        want = orc.c.ei(mu_o, var_o, fmin)
        assert np.allclose(ei, want, rtol=1e-6, atol=1e-9) and idx == orc.c.argmax_first(want)[0]
    ctx.gp_fit(X_obs, Y1, hyp["lenscale_sq"], hyp["amp"], hyp["noise"], hyp["mean"])  # back to one column


def test_fantasize_moments_and_pending_ei(ctx, orc):
    import bot7_amd
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, 6, 60, 700, B.hartmann6)
    X_pend = X_hid[[5, 300, 650]]                      # three pending points
    ctx.gp_fit(X_obs, Y, hyp["lenscale_sq"], hyp["amp"], hyp["noise"], hyp["mean"])
    Yp, mu_p, cov_p = ctx.gp_fantasize(X_pend, 20000, seed=7, want_moments=True)
    f = orc.gp.fit(X_obs, Y, **hyp)
    mu_o, _ = orc.gp.predict(f, X_pend)
    Ks = orc.gp.ardse(X_pend, X_obs, hyp["lenscale_sq"], hyp["amp"])
    from scipy.linalg import solve_triangular
    V = solve_triangular(f.L, Ks.T, lower=True)
    cov_o = orc.gp.ardse(X_pend, None, hyp["lenscale_sq"], hyp["amp"]) - V.T @ V
    assert np.allclose(mu_p, mu_o[:, 0], rtol=1e-7, atol=1e-9)
    assert np.allclose(cov_p, cov_o, rtol=1e-6, atol=1e-8 * hyp["amp"])
    # the draws have those moments (20000 samples: 4-sigma bands) and are reproducible per seed
    sd = np.sqrt(np.diag(cov_o))
    assert np.all(np.abs(Yp.mean(axis=1) - mu_p) < 4 * sd / np.sqrt(20000))
    emp = np.cov(Yp)
    assert np.allclose(emp, cov_o, atol=0.05 * np.outer(sd, sd).max())
    assert np.array_equal(ctx.gp_fantasize(X_pend, 50, seed=7), ctx.gp_fantasize(X_pend, 50, seed=7))
    assert not np.array_equal(ctx.gp_fantasize(X_pend, 50, seed=8), ctx.gp_fantasize(X_pend, 50, seed=7))
    # EI with pending points, end to end through the score class; the oracle scores the SAME fantasies
    model = bot7_amd.models.gp_regressor({"seed": 4}, context=ctx)
    model.hyp = hyp
    score = bot7_amd.scores.expected_improvement({"nFantasies": 16})
    got = score(model, hyp, X_obs, Y, X_hid, X_pend)
    model._fcalls = 0                                  # replay the same fantasy draw
    Y_pend = model.fantasize(16, X_obs, Y, X_pend, hyp)
    X2 = np.concatenate([X_obs, X_pend])
    Y2 = np.concatenate([np.tile(Y, (1, 16)), Y_pend])
    f2 = orc.gp.fit(X2, Y2, **hyp)
    mu2, var2 = orc.gp.predict(f2, X_hid)
    want = orc.c.ei(mu2, var2, Y2.min(axis=0))
    assert got.shape == (700,) and np.allclose(got, want, rtol=1e-5, atol=1e-9)
    assert int(np.argmax(got)) == int(np.argmax(want))


# ---- DNGO: basis network + Bayesian linear head (BASELINE config 5) -----------------------------
def test_dngo_basis_and_blr_head_match_oracle(ctx, orc):
    from oracle import blr
    from conftest import make_network
    import bot7_amd
    for d, widths, act, N, M in [(6, (50, 50, 50), "Tanh", 80, 3000), (2, (100,), "ReLU", 24, 257),
                                 (32, (64, 100), "Sigmoid", 300, 1000)]:
        W, b = make_network(d, widths, seed=d)
        X_obs, Y, X_hid, _ = make_problem(ctx, orc, d, N, M, {6: B.hartmann6, 2: B.braninhoo, 32: B.ackley}[d])
        Z0 = ctx.blr_basis(W, b, act, X=X_obs)
        Z0_o = blr.basis(X_obs, W, b, act)
        assert np.allclose(Z0, Z0_o, rtol=1e-12, atol=1e-13)
        alpha_p, beta, mean = 2.0, 1.0 / (1e-2 * float(np.var(Y))), float(np.mean(Y))
        f = blr.fit(Z0_o, Y, alpha_p, beta, mean)
        nll = ctx.blr_fit(Z0, Y, alpha_p, beta, mean, want_nll=True)
        assert nll == pytest.approx(float(f["nll"]), rel=1e-9, abs=1e-7)
        nll_x = ctx.blr_fit_x(W, b, act, X_obs, Y, alpha_p, beta, mean, want_nll=True)   # basis + fit on the device
        assert nll_x == pytest.approx(float(f["nll"]), rel=1e-9, abs=1e-7)
        ctx.grid_upload(X_hid)
        Z1 = ctx.blr_basis(W, b, act, download=True)       # features of the resident grid, kept on the device
        assert np.allclose(Z1, blr.basis(X_hid, W, b, act), rtol=1e-12, atol=1e-13)
        mu, var = ctx.blr_predict()
        mu_o, var_o = blr.predict(f, blr.basis(X_hid, W, b, act))
        assert relerr(mu, mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(var, var_o) < REL
        assert (var >= 1.0 / beta).all()
        # scores reuse the GP path's accumulator and arg-max
        ctx.score_reset()
        ctx.score_ei([float(Y.min())], 0.0)
        _, idx, ei = ctx.score_finish(1.0, download=True)
        want = orc.c.ei(mu_o, var_o, [float(Y.min())])
        assert np.allclose(ei, want, rtol=1e-5, atol=1e-10) and idx == orc.c.argmax_first(want)[0]
        # caller-made features instead of grid + network
        ctx.blr_features(blr.basis(X_hid, W, b, act))
        mu2, var2 = ctx.blr_predict()
        assert np.allclose(mu2, mu, rtol=1e-12, atol=1e-13) and np.allclose(var2, var, rtol=1e-11, atol=0)
    # stale features are refused after the grid changes
    ctx.grid_upload(X_hid)
    ctx.blr_basis(W, b, act)
    ctx.grid_remove(3)
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.blr_predict()


def test_dngo_model_in_bayesopt_loop(ctx, orc):
    """config 5 plumbing: models.dngo through bots.bayesopt (no marginalisation loop, bots/bayesopt.lua:65-66)."""
    from oracle import blr
    from conftest import make_network
    import bot7_amd

    class H(object):
        def __init__(self, name):
            self.name, self.min, self.max, self.size = name, 0.0, 1.0, 1

    W, b = make_network(6, (50, 50, 50), seed=1)
    cfg = {"bot": {"verbose": 0, "budget": 7, "nInitial": 3, "nSamples": 10, "seed": 3},
           "grid": {"type": "sobol", "size": 2000, "dims": 6}, "score": {"type": "expected_improvement"},
           "model": {"type": "dngo", "network": {"weights": W, "biases": b, "activation": "Tanh"}, "alpha": 1.0}}
    grid = bot7_amd.grids.sobol(dict(cfg["grid"], mins=np.zeros(6), maxes=np.ones(6)), context=ctx)()
    model = bot7_amd.models.dngo(cfg["model"], context=ctx)
    bot = bots.bayesopt(B.hartmann6, [H("x%d" % i) for i in range(6)], cfg,
                                 cache={"candidates": grid, "model": model})
    for t in range(1, 8):
        cand = np.asarray(bot.candidates).copy()
        obs = None if bot.observed is None else bot.observed.copy()
        resp = None if bot.responses is None else bot.responses.copy()
        x, y = bot.run_trial()
        if t > 3:
            h = model.hyp
            f = blr.fit(blr.basis(obs, W, b, "Tanh"), resp, h["alpha"], h["beta"], h["mean"])
            mu, var = blr.predict(f, blr.basis(cand, W, b, "Tanh"))
            widx = orc.c.argmax_first(orc.c.ei(mu, var, [float(resp.min())]))[0]
            assert np.array_equal(x, cand[widx - 1]), "trial %d nominated a different candidate" % t
    assert bot.observed.shape == (7, 6)


def test_default_regime_device_loop_follows_the_oracle_loop(ctx):
    """The reference's default experiment (hartmann6, 2e4 Sobol candidates, nInitial 2, S = 10 slice-sampled hypers, EI) as
    bench.py --workload default runs it, 14 trials: the device loop (b7_gp_nll_batch per density evaluation, b7_eval_nominate,
    b7_nominate_commit) and the oracle loop (oracle/hostctx.py behind the same harness code and seeds) must nominate the same
    candidates and draw the same hypers -- the sampler sees the densities only through comparisons.  Then the bench line itself
    at a budget of 10.  (GP algebra: parity unpinned.)"""
    import json
    import subprocess
    import sys
    from harness import default_regime as dr
    from oracle.hostctx import OracleContext
    g = dr.run(ctx, budget=14)
    o = dr.run(OracleContext(), budget=14)
    assert g["nominees"] == o["nominees"] and len(g["nominees"]) == 14
    same, worst = dr.agreement(g, o)
    assert same == 14 and worst < 1e-9
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "default", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=root, B7_DEFAULT_BUDGET="10"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["unit"] == "trials/s" and line["n_gpus"] == 1 and line["value"] > 0 and line["vs_baseline"] is None
    assert line["roofline"]["bound"] == "dependent chain" and line["roofline_nominate"]["frac_of_pipe"] > 0
    assert line["cpu_baseline"]["kind"] == "port" and line["parity"]["leading_trials_with_the_same_nominee"] == 10
    assert line["config"]["workload"].startswith("default")


def test_two_ranks_share_the_gpu_and_agree_with_one(ctx):
    """The N>1 path end to end on real device contexts: two ranks (gloo exchange, both on cuda:0) each own half of
    a 131072-candidate Sobol grid; the exchanged winner must be the single-process winner over the whole grid."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    common = ["--workload", "cfg3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--candidates", "131072"]
                         + common, capture_output=True, text=True, env=env, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
                          "--gpus", "2", "--backend", "gloo", "--candidates", "65536"] + common,
                         capture_output=True, text=True, env=env, timeout=300)
    assert two.returncode == 0, two.stderr[-2000:]
    b1 = json.loads(one.stdout.strip().splitlines()[-1])
    b2 = json.loads(two.stdout.strip().splitlines()[-1])
    assert b2["n_gpus"] == 2 and b2["config"]["candidates_total"] == 131072 == b1["config"]["candidates_total"]
    assert b1["best"] == b2["best"], "sharded winner differs from the unsharded arg-max"
    # the same two ranks through bench.py's PRODUCT path (--backend rccl: b7_comm_init + b7_eval_nominate's exchange branch),
    # with RCCL's transport served by the shared-memory test double of tests/stub (RCCL refuses two ranks on one device)
    sys.path.insert(0, os.path.join(root, "tests"))
    from test_sharded_loop import _diag_lib, _stub_lib
    three = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", "29534", os.path.join(root, "bench.py"),
                            "--gpus", "2", "--candidates", "65536"] + common,
                           capture_output=True, text=True, env=dict(env, B7_RCCL_LIB=_stub_lib(), BOT7HIP_LIB=_diag_lib()), timeout=300)   # the override lives in the diagnostic build
    assert three.returncode == 0, three.stderr[-2000:]
    b3 = json.loads(three.stdout.strip().splitlines()[-1])
    assert b3["n_gpus"] == 2 and "b7_eval_nominate" in b3["step_api"] and "ncclAllReduce" in b3["config"]["parallelism"]
    assert b3["best"] == b1["best"], "winner through the communicator path differs from the unsharded arg-max"
    # `python bench.py --gpus 2` with NO launcher: the single-process group (b7_group_*), here with both members on cuda:0 --
    # records merged on the host, then the grouped ncclAllReduce forced through the in-process test double
    for extra_env, rccl_ranks in (({}, 0), ({"B7_RCCL_LIB": _stub_lib(), "B7_GROUP_EXCHANGE": "rccl"}, 2)):
        four = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--virtual-ranks", "--candidates", "65536"]
                              + common, capture_output=True, text=True, env=dict(env, **extra_env), timeout=300)
        assert four.returncode == 0, four.stderr[-2000:]
        b4 = json.loads(four.stdout.strip().splitlines()[-1])
        assert b4["n_gpus"] == 2 and b4["config"]["layout"] == "group" and b4["config"]["rccl_ranks"] == rccl_ranks
        assert b4["best"] == b1["best"], "the group's winner differs from the unsharded arg-max"


# ---- edge cases through the C ABI ---------------------------------------------------------------------------------
def test_edge_shapes_and_options(ctx, orc):
    import bot7_amd
    rng = np.random.default_rng(17)
    # N = 1, d = 1; M = 1
    X = np.array([[0.3]])
    Y = np.array([[1.5]])
    ctx.gp_fit(X, Y, [0.25], 2.0, 1e-3, 0.5)
    ctx.grid_upload(np.array([[0.31]]))
    mu, var = ctx.gp_predict()
    f = orc.gp.fit(X, Y, [0.25], 2.0, 1e-3, 0.5)
    mu_o, var_o = orc.gp.predict(f, np.array([[0.31]]))
    assert np.allclose(mu, mu_o, rtol=1e-9) and np.allclose(var, var_o, rtol=1e-6)
    # every padded-dimension class boundary: d = 4|5, 8|9, 16|17, 32|33, 48|49, 64|65, 96
    for d in (4, 5, 8, 9, 16, 17, 32, 33, 48, 49, 64, 65, 96):
        N, M = 37, 130
        Xo, Xh = rng.random((N, d)), rng.random((M, d))
        Yo = np.sin(Xo.sum(1, keepdims=True))
        ls = rng.random(d) * d + 0.1 * d
        f = orc.gp.fit(Xo, Yo, ls, 1.3, 1e-3, 0.1)
        ctx.gp_fit(Xo, Yo, ls, 1.3, 1e-3, 0.1)
        ctx.grid_upload(Xh)
        mu, var = ctx.gp_predict()
        mu_o, var_o = orc.gp.predict(f, Xh)
        assert relerr(mu, mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(var, var_o) < REL, "d = %d" % d
    with pytest.raises(bot7_amd.Bot7HipError) as e:
        ctx.gp_fit(rng.random((5, 97)), np.zeros(5), np.ones(97), 1.0, 1e-3, 0.0)
    assert e.value.code == -5                                    # d > 96: documented limit
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.gp_fit(rng.random((5, 3)), np.zeros((5, 257)), np.ones(3), 1.0, 1e-3, 0.0)  # ycols > 256
    # options the `gp` package leaves open: noise in the predictive variance, clamping
    Xo, Xh = rng.random((40, 3)), rng.random((500, 3))
    Yo = np.cos(3 * Xo.sum(1, keepdims=True))
    f = orc.gp.fit(Xo, Yo, [0.3] * 3, 1.0, 1e-2, 0.0)
    ctx.gp_fit(Xo, Yo, [0.3] * 3, 1.0, 1e-2, 0.0)
    ctx.grid_upload(Xh)
    ctx.gp_set_opts(var_with_noise=1)
    _, var_n = ctx.gp_predict()
    _, var_on = orc.gp.predict(f, Xh, var_with_noise=True)
    assert relerr(var_n, var_on) < REL
    ctx.gp_set_opts(var_clamp=1, var_min=0.5)
    _, var_c = ctx.gp_predict()
    _, var_oc = orc.gp.predict(f, Xh, var_min=0.5)
    assert np.allclose(var_c, var_oc, rtol=1e-5) and var_c.min() == 0.5
    ctx.gp_set_opts()                                            # back to defaults
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.gp_set_opts(jitter_growth=1.0)
    # NaN in a candidate propagates to its own mean/variance only (TH clamp passes NaN)
    Xh2 = Xh.copy()
    Xh2[7, 1] = np.nan
    ctx.grid_upload(Xh2)
    mu, var = ctx.gp_predict()
    assert np.isnan(mu[7, 0]) and np.isnan(var[7]) and np.isfinite(np.delete(var, 7)).all()
    ctx.score_reset()
    ctx.score_ei([float(Yo.min())], 0.0)
    _, idx, _ = ctx.score_finish(1.0)
    assert idx == 8                                              # the first NaN wins score:max(1), as in TH


def test_workspace_chunking_is_invisible(ctx, orc):
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, 6, 200, 5000, B.hartmann6)
    ctx.gp_fit(X_obs, Y, **hyp)
    ctx.grid_upload(X_hid)
    mu, var = ctx.gp_predict()
    for nbytes in (8 * 256 * 256, 8 * 256 * 300, 8 * 256 * 1024):   # 1, 1 and 4 tiles of 256 rows per chunk
        ctx.set_workspace(nbytes)
        mu2, var2 = ctx.gp_predict()
        assert np.array_equal(mu, mu2) and np.array_equal(var, var2)
    ctx.set_workspace(4 << 30)
    import bot7_amd
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.set_workspace(100)


def test_sobol_random_shapes_property(ctx, orc):
    """hypothesis: any (size, dims, skip) the reference accepts is reproduced bit for bit."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=30, deadline=None, derandomize=True)
    @given(st.integers(1, 700), st.integers(1, 39), st.integers(0, 2 ** 13))  # the oracle walks the recurrence from 0
    def check(size, dims, skip):
        assert np.array_equal(ctx.grid_sobol(size, dims, skip), orc.c.sobol(size, dims, skip))

    check()


def test_full_size_cfg4_shape(ctx, orc):
    """BASELINE config 4's per-GPU shape: rastrigin d = 64, N = 2048, counter-based uniform grid (Sobol stops at
    39 dims in the reference), one 65536-row shard.  Sampled candidates against the oracle, determinism, bounds."""
    d, N, M = 64, 2048, 65536
    X_obs = ctx.grid_random(N, d, seed=1, row_offset=8 * M)
    X_hid = ctx.grid_random(M, d, seed=1, row_offset=3 * M)        # shard 3 of 8
    Y = B.rastrigin(X_obs)
    amp = float(np.var(Y))
    hyp = dict(lenscale_sq=np.full(d, d / 8.0), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y)))
    ctx.gp_fit(X_obs, Y, **hyp)
    mu, var = ctx.gp_predict()
    mu_b, var_b = ctx.gp_predict()
    assert np.array_equal(mu, mu_b) and np.array_equal(var, var_b)
    assert (var > 0).all() and (var <= amp * (1 + 1e-12)).all()
    f = orc.gp.fit(X_obs, Y, **hyp)
    sample = np.random.default_rng(4).choice(M, 400, replace=False)
    mu_o, var_o = orc.gp.predict(f, X_hid[sample])
    assert relerr(mu[sample], mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(var[sample], var_o) < REL
    ctx.score_reset()
    ctx.score_ei([float(Y.min())], 0.0)
    val, idx, ei = ctx.score_finish(1.0, download=True)
    assert (idx, val) == orc.c.argmax_first(ei)


def test_bench_contract():
    """bench.py prints exactly one JSON line with the fields the driver and the judge read."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1",
                        "--candidates", "262144", "--cpu-sample", "4096"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "EI candidates scored/sec at N=2048,d=32" and d["unit"] == "candidates/s"
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 78.6
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.3 < rf["frac"] < 1.0
    assert rf["traffic"] is not None and rf["traffic"] > rf["algorithmic_bytes_per_launch"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["value"] > 1e6, "below the north-star target of 1e6 candidates/s"
    # SURVEY 8(d) inputs: Sobol point 1 = (0.5, ..., 0.5) is an observation, so the winner is not row 1, and the
    # GPU arg-max over the CPU leg's rows is the oracle's
    assert d["best"]["index1"] != 1
    bs = d["best_in_cpu_sample"]
    assert bs["matches_cpu_argmax"] and bs["index1"] == cb["argmax1"] != 1
    assert bs["max_abs_score_diff_vs_cpu"] < 1e-9 and bs["cpu_top2_gap"] > 10 * bs["max_abs_score_diff_vs_cpu"]
    mg = d["marginalised"]
    assert mg["samples"] == 10 and mg["marginalised_candidates_per_s"] > 1e5
    assert set(d["gp_fit_ms_by_N"]) == {"256", "1024", "2048"} and all(0 < v < 50 for v in d["gp_fit_ms_by_N"].values())
    assert abs(d["value"] - d["config"]["candidates_total"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9


# ---- incremental refit ---------------------------------------------------------------------------------
def test_gp_append_matches_full_refit(ctx, orc):
    import bot7_amd
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, 6, 125, 900, B.hartmann6)
    ctx.gp_fit(X_obs[:120], Y[:120], **hyp)
    ctx.grid_upload(X_hid)
    for n in range(120, 125):                                    # 120 -> 125 one observation at a time
        ctx.gp_append(X_obs[n], Y[n])
        f = orc.gp.fit(X_obs[:n + 1], Y[:n + 1], **hyp)
        L, alpha, Linv = ctx.gp_download(n + 1)
        assert np.allclose(L, f.L, rtol=1e-9, atol=1e-12)
        assert np.allclose(Linv @ f.L, np.eye(n + 1), atol=1e-8)
        assert relerr(alpha, f.alpha, floor=1e-3 * np.abs(f.alpha).max()) < 1e-6
        mu, var = ctx.gp_predict()
        mu_o, var_o = orc.gp.predict(f, X_hid)
        assert relerr(mu, mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(var, var_o) < REL
    # the padded factor is full at a multiple of 128: the call refuses and a rebuild is required
    ctx.gp_fit(X_obs[:3], Y[:3], **hyp)
    Xm = orc.c.sobol(130, 6, 5000)
    Ym = B.hartmann6(Xm)
    ctx.gp_fit(Xm[:128], Ym[:128], **hyp)
    with pytest.raises(bot7_amd.Bot7HipError) as e:
        ctx.gp_append(Xm[128], Ym[128])
    assert e.value.code == -4
    # a duplicate of an existing observation with tiny noise is numerically singular: refused, fit left intact
    ctx.gp_fit(X_obs[:50], Y[:50], hyp["lenscale_sq"], hyp["amp"], 0.0, hyp["mean"])
    L0, a0, _ = ctx.gp_download(50)
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.gp_append(X_obs[7], Y[7])
    L1, a1, _ = ctx.gp_download(50)
    assert np.array_equal(L0, L1) and np.array_equal(a0, a1)
    mu, var = ctx.gp_predict()
    f = orc.gp.fit(X_obs[:50], Y[:50], hyp["lenscale_sq"], hyp["amp"], 0.0, hyp["mean"])
    assert relerr(var, orc.gp.predict(f, X_hid)[1], floor=1e-9) < 1e-3   # noiseless: variance itself is tiny near data
    # the model mirror uses it transparently under a point estimate
    model = bot7_amd.models.gp_regressor({}, context=ctx)
    model.hyp = hyp
    model.fit(X_obs[:60], Y[:60])
    assert not model.last_fit.get("incremental")
    model.fit(X_obs[:61], Y[:61])
    assert model.last_fit.get("incremental")
    model.fit(X_obs[:62], Y[:62], dict(hyp, amp=hyp["amp"] * 1.1))   # hypers changed -> rebuild
    assert not model.last_fit.get("incremental")


# ---- round-2 regressions ---------------------------------------------------------------------
@pytest.mark.parametrize("N,c", [(70, 100), (70, 256), (128, 65), (300, 200)])
def test_multi_column_alpha_on_a_fresh_context(orc, N, c):
    """launch_alpha's intermediates (n*c*(1 + ceil(n/256)) doubles) used to live in W (n*n doubles): too small
    whenever c*(1+ceil(n/256)) > n.  A FRESH context has no buffer grown by an earlier big fit to hide behind."""
    import bot7_amd
    fresh = bot7_amd.Context(0)
    try:
        X_obs, Y1, X_hid, hyp = make_problem(None, orc, 6, N, 300, B.hartmann6)
        Y = Y1 + 0.1 * np.random.default_rng(c).normal(size=(N, c))
        out = fresh.gp_fit(X_obs, Y, hyp["lenscale_sq"], hyp["amp"], hyp["noise"], hyp["mean"], want_nll=True)
        f = orc.gp.fit(X_obs, Y, **hyp)
        assert np.allclose(out["nll"], f.nll, rtol=1e-9, atol=1e-7)
        L, alpha, Linv = fresh.gp_download(N, c)
        assert np.allclose(alpha, f.alpha, rtol=1e-6, atol=1e-6 * np.abs(f.alpha).max())
        assert np.allclose(L, f.L, rtol=1e-9, atol=1e-12)          # neighbours of the old scratch are intact
        assert np.allclose(Linv @ f.L, np.eye(N), atol=1e-8)
        fresh.grid_upload(X_hid)
        mu, var = fresh.gp_predict()
        mu_o, var_o = orc.gp.predict(f, X_hid)
        assert relerr(mu, mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(var, var_o) < REL
    finally:
        fresh.close()


def test_derived_device_grid_is_uploaded_not_mistaken_for_resident(ctx, orc):
    """A transformed / permuted view of the resident grid has other rows than the device holds: predict must
    treat it like a plain ndarray."""
    import bot7_amd
    X_obs, Y, _, hyp = make_problem(None, orc, 6, 40, 64, B.hartmann6)
    grid = bot7_amd.grids.sobol({"size": 500, "dims": 6}, context=ctx)()
    model = bot7_amd.models.gp_regressor({}, context=ctx)
    model.hyp = hyp
    assert model._is_resident(grid)
    for derived in (grid * 0.5, grid[::-1], grid * 2 - 0.3, grid[:500]):
        assert not model._is_resident(derived)
        a = model.predict(X_obs, Y, derived, hyp)
        b = model.predict(X_obs, Y, np.array(derived), hyp)
        assert np.array_equal(a["mean"], b["mean"]) and np.array_equal(a["var"], b["var"])
    mu_o, var_o = orc.gp.predict(orc.gp.fit(X_obs, Y, **hyp), np.asarray(grid) * 0.5)
    a = model.predict(X_obs, Y, grid * 0.5, hyp)
    assert relerr(a["mean"], mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(a["var"], var_o) < REL


def test_incremental_append_is_not_used_after_a_jittered_fit(ctx, orc):
    """utils/math.lua:159-218 refactors K + eps*I in full every trial; extending a jittered factor by an un-jittered
    row would diverge from that exactly where the arg-max is most sensitive."""
    import bot7_amd
    X = orc.c.sobol(40, 3, 1)
    X[7] = X[3]                                   # duplicate observation, no noise -> singular K -> jitter
    Y = np.sin(3.0 * X).sum(axis=1, keepdims=True)
    hyp = dict(lenscale_sq=np.full(3, 0.4), amp=1.0, noise=0.0, mean=0.0)
    model = bot7_amd.models.gp_regressor({}, context=ctx)
    model.hyp = hyp
    model.fit(X[:30], Y[:30])
    assert model.last_fit["jitter"] > 0.0 and model.last_fit["info"] > 0
    model.fit(X[:31], Y[:31])
    assert not model.last_fit.get("incremental") and model.last_fit["jitter"] > 0.0


# ---- round 2: multi-index removal, resident data, the exchange behind the C ABI, full BASELINE shapes ---------------
def test_grid_remove_rows_matches_repeated_single_removal(ctx, orc):
    """utils.tensor.remove with an index tensor (utils/tensor.lua:158-170): one stable pass == the oracle's deletion of
    the same rows; steal's gathered rows come back in the order given; duplicates count once."""
    import bot7_amd
    pool = ctx.grid_sobol(5000, 7, 1)
    for idx in ([1], [5000], [3, 1, 4999, 77, 78, 79], list(range(1, 5000, 13)), [10, 10, 11]):
        ctx.grid_upload(pool)
        rows = ctx.grid_remove_rows(idx)
        assert np.array_equal(rows, pool[np.asarray(idx) - 1])
        want = pool
        for i in sorted(set(idx), reverse=True):
            want = orc.c.remove_row(want, i)
        assert np.array_equal(ctx.grid_download(), want)
    ctx.grid_upload(pool)
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.grid_remove_rows([0])
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.grid_remove_rows([5001])
    assert ctx.grid_shape()[0] == 5000


def test_set_data_fit_hyp_is_gp_fit_bit_for_bit(ctx, orc):
    """b7_gp_set_data + b7_gp_fit_hyp == b7_gp_fit (same bits), refits under new hypers see the same data, and an
    appended observation joins the resident data set."""
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, 6, 200, 700, B.hartmann6)
    ctx.grid_upload(X_hid)
    a = ctx.gp_fit(X_obs, Y, want_nll=True, **hyp)
    La, aa, Lia = ctx.gp_download(200)
    mu_a, var_a = ctx.gp_predict()
    ctx.gp_set_data(X_obs, Y)
    for scale in (1.7, 1.0):
        h = dict(hyp, lenscale_sq=hyp["lenscale_sq"] * scale, mean=hyp["mean"] + (scale - 1.0))
        b = ctx.gp_fit_hyp(want_nll=True, **h)
        f = orc.gp.fit(X_obs, Y, **h)
        assert np.allclose(b["nll"], f.nll, rtol=1e-9, atol=1e-7)
    Lb, ab, Lib = ctx.gp_download(200)
    mu_b, var_b = ctx.gp_predict()
    assert a["nll"][0] == b["nll"][0] and np.array_equal(La, Lb) and np.array_equal(aa, ab) and np.array_equal(Lia, Lib)
    assert np.array_equal(mu_a, mu_b) and np.array_equal(var_a, var_b)
    # append, then refit from the resident data under other hypers
    x_new, y_new = X_hid[5], B.hartmann6(X_hid[5:6])[0]
    ctx.gp_append(x_new, y_new)
    h2 = dict(hyp, amp=hyp["amp"] * 1.3)
    out = ctx.gp_fit_hyp(want_nll=True, **h2)
    f2 = orc.gp.fit(np.vstack([X_obs, x_new]), np.vstack([Y, y_new.reshape(1, -1)]), **h2)
    assert np.allclose(out["nll"], f2.nll, rtol=1e-9, atol=1e-7)
    import bot7_amd
    fresh = bot7_amd.Context(0)
    try:
        fresh._data_d, fresh.ycols = 6, 1                 # pretend the wrapper saw data: the LIBRARY must refuse
        with pytest.raises(bot7_amd.Bot7HipError) as e:
            fresh.gp_fit_hyp(**hyp)
        assert e.value.code == -4
    finally:
        fresh.close()


def test_jitter_schedule_gate_is_computed_on_the_device(ctx, orc):
    """The Frobenius-norm gate of utils/math.lua:174 comes from a device reduction now (no host copy of K).  The
    chol(I) branch behind it (:184-186) cannot be reached with finite input -- a failed attempt needs eps <= -lambda_min
    <= ||A||_F, the branch needs eps > ||A||_F -- so what is checked is the retry path right up to the gate:
    A = -(1/n) 11' has lambda_min = -1 = -||A||_F and is fixed by the first eps above 1."""
    n = 70
    for A in (-np.ones((n, n)) / n, np.eye(n) - (1.0 + 1e-3) * np.ones((n, n)) / n):
        L, jit, info = ctx.chol(A)
        Lo, jit_o, _ = orc.c.chol_jitter(A)
        assert jit == jit_o and jit > 0 and info > 0 and np.allclose(L, Lo, rtol=1e-6, atol=1e-9)
    assert ctx.chol(-np.ones((n, n)) / n)[1] == pytest.approx(1e-8 * 1.1 ** 194, rel=1e-12)


def test_comm_world_of_one_really_calls_rccl(orc):
    """b7_comm_* through ctypes with world = 1: librccl gets mapped, ncclCommInitRank / ncclAllReduce really run, and
    the global finish equals the local one with the row offset applied (ties, NaN and -0.0 included)."""
    import bot7_amd
    from bot7_amd import _lib
    c = bot7_amd.Context(0)
    try:
        assert c.comm_info() == (0, 1)
        uid = _lib.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        c.comm_init(0, 1, uid)
        with open("/proc/self/maps") as f:
            assert "librccl" in f.read()
        assert c.comm_info() == (0, 1)
        with pytest.raises(bot7_amd.Bot7HipError) as e:
            c.comm_init(0, 1, uid)
        assert e.value.code == -4
        assert np.array_equal(c.comm_allreduce([1.5, -2.0, np.inf], "max"), [1.5, -2.0, np.inf])
        assert np.array_equal(c.comm_allreduce([1.5, -2.0], "sum"), [1.5, -2.0])
        X_obs, Y, X_hid, hyp = make_problem(None, orc, 6, 64, 3000, B.hartmann6)
        off = 123456789012
        for case in ("plain", "nan", "ties"):
            Xh = X_hid.copy()
            if case == "nan":
                Xh[[700, 300]] = np.nan                  # NaN candidates -> NaN scores: the first NaN wins
            c.gp_fit(X_obs, Y, **hyp)
            c.grid_upload(Xh)
            c.gp_predict(download=False)
            c.score_reset()
            if case != "ties":                           # "ties": the untouched accumulator, all zeros -> index 1
                c.score_ei([float(Y.min())], 0.0)
            v2, i2 = c.score_finish_global(1.0, off)
            v1, i1, s = c.score_finish(1.0, download=True)
            wi, wv = orc.c.argmax_first(s)
            assert i1 == wi and i2 == wi + off and (v2 == wv or (v2 != v2 and wv != wv))
            assert {"plain": wi > 1, "nan": wi == 301, "ties": wi == 1}[case]
        c.comm_destroy()
        assert c.comm_info() == (0, 1)
        assert c.score_finish_global(1.0, 7)[1] == 1 + 7     # without a communicator: a world of one
    finally:
        c.close()


def test_score_finish_global_matches_th_rule_on_real_scores(ctx, orc):
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, 6, 64, 3000, B.hartmann6)
    ctx.gp_fit(X_obs, Y, **hyp)
    ctx.grid_upload(X_hid)
    ctx.gp_predict(download=False)
    ctx.score_reset()
    ctx.score_ei([float(Y.min())], 0.0)
    ctx.score_ei([float(Y.min())], 0.0)
    v, i = ctx.score_finish_global(2.0, 1000)
    _, _, s = ctx.score_finish(1.0, download=True)          # already divided
    wi, wv = orc.c.argmax_first(s)
    assert (v, i) == (wv, wi + 1000)


def _full_shape(ctx, orc, d, N, M, objective, sobol=True, seed_rows=7):
    """SURVEY 8(d) inputs at a BASELINE shape, generated on the device exactly as bench.py does."""
    import bench
    X_obs = bench.make_inputs(ctx, d, N, M, 0, M)
    X_hid = ctx.grid_download()
    s = (M + N) // N
    if sobol:   # the construction itself against the oracle's pool (strided pick + stable deletion)
        pool = orc.c.sobol(M + N, d, bench.SOBOL_SKIP)
        mask = np.ones(M + N, dtype=bool)
        mask[np.arange(N) * s] = False
        assert np.array_equal(X_obs, pool[np.arange(N) * s]) and np.array_equal(X_hid, pool[mask])
    Y = objective(X_obs)
    amp = float(np.var(Y))
    hyp = dict(lenscale_sq=np.full(d, d / 8.0), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y)))
    return X_obs, Y, X_hid, hyp


def test_full_shape_cfg3(ctx, orc):
    """BASELINE config 3 at full shape: ackley d = 32, N = 1024 (Npad = 1024: 16 panels), M = 262144 Sobol
    candidates, EI.  Oracle on a 1024-row sample + the whole-grid arg-max against TH's rule on the downloaded scores
    + the oracle's arg-max over a 16384-row prefix."""
    d, N, M = 32, 1024, 262144
    X_obs, Y, X_hid, hyp = _full_shape(ctx, orc, d, N, M, B.ackley)
    out = ctx.gp_fit(X_obs, Y, want_nll=True, **hyp)
    f = orc.gp.fit(X_obs, Y, **hyp)
    assert out["info"] == 0 and np.allclose(out["nll"], f.nll, rtol=1e-9)
    mu, var = ctx.gp_predict()
    assert (var > 0).all() and (var <= hyp["amp"] * (1 + 1e-12)).all() and np.isfinite(mu).all()
    sample = np.random.default_rng(3).choice(M, 1024, replace=False)
    mu_o, var_o = orc.gp.predict(f, X_hid[sample])
    assert relerr(mu[sample], mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(var[sample], var_o) < REL
    ctx.score_reset()
    ctx.score_ei([float(Y.min())], 0.0)
    val, idx, ei = ctx.score_finish(1.0, download=True)
    assert (idx, val) == orc.c.argmax_first(ei) and idx != 1
    pre = 16384
    mu_p, var_p = orc.gp.predict(f, X_hid[:pre])
    want = orc.c.ei(mu_p, var_p, [float(Y.min())])
    gv, gi = ctx.argmax(ei[:pre])
    top2 = np.partition(want, -2)[-2:]
    print("cfg3 prefix arg-max %d, top-2 gap %.3e, max |EI - oracle| %.3e" % (gi, top2[1] - top2[0],
                                                                           np.max(np.abs(ei[:pre] - want))))
    assert gi == orc.c.argmax_first(want)[0]


def test_full_shape_cfg4_per_gpu(ctx, orc):
    """BASELINE config 4's per-GPU shape at full size: rastrigin d = 64, N = 2048, 262144 counter-based uniform
    candidates (one of eight shards of the 2M grid), EI."""
    import bench
    d, N, M8, M = 64, 2048, 8 * 262144, 262144
    X_obs = bench.make_inputs(ctx, d, N, M8, 3 * M, 4 * M)      # shard 3 of 8
    X_hid = ctx.grid_download()
    assert X_hid.shape == (M, d)
    s = (M8 + N) // N
    first = bench.pool_index(3 * M, N, s)
    assert np.array_equal(X_hid[0], ctx_row(ctx, d, first)) and (X_obs >= 0).all() and (X_obs < 1).all()
    ctx.grid_upload(X_hid)
    Y = B.rastrigin(X_obs)
    amp = float(np.var(Y))
    hyp = dict(lenscale_sq=np.full(d, d / 8.0), amp=amp, noise=1e-4 * amp, mean=float(np.mean(Y)))
    ctx.gp_fit(X_obs, Y, **hyp)
    mu, var = ctx.gp_predict()
    assert (var > 0).all() and (var <= amp * (1 + 1e-12)).all()
    f = orc.gp.fit(X_obs, Y, **hyp)
    sample = np.random.default_rng(4).choice(M, 512, replace=False)
    mu_o, var_o = orc.gp.predict(f, X_hid[sample])
    assert relerr(mu[sample], mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(var[sample], var_o) < REL
    ctx.score_reset()
    ctx.score_ei([float(Y.min())], 0.0)
    val, idx = ctx.score_finish_global(1.0, 3 * M)
    _, _, ei = ctx.score_finish(1.0, download=True)
    wi, wv = orc.c.argmax_first(ei)
    assert (val, idx) == (wv, wi + 3 * M)
    pre = 8192
    want = orc.c.ei(*orc.gp.predict(f, X_hid[:pre]), [float(Y.min())])
    assert ctx.argmax(ei[:pre])[1] == orc.c.argmax_first(want)[0]


def ctx_row(ctx, d, pool_row):
    """One row of the counter-based uniform pool (seed 1), regenerated on a scratch context."""
    import bot7_amd
    c = bot7_amd.Context(0)
    try:
        return c.grid_random(1, d, seed=1, row_offset=pool_row)[0]
    finally:
        c.close()


def test_full_shape_cfg5(ctx, orc):
    """BASELINE config 5 at full shape: DNGO head (3 x 50 tanh basis) over 65536 Sobol candidates, N = 256, with
    responses that are linear in the features (EI of order 1e-1 at the winners, not an underflow): the Bayesian linear
    head against oracle/blr.py on a 512-row sample, and the arg-max over the whole grid against the oracle."""
    from oracle import blr
    import bench
    d, N, M = 5, 256, 65536
    X_obs = bench.make_inputs(ctx, d, N, M, 0, M)
    X_hid = ctx.grid_download()
    rng = np.random.default_rng(0)
    dims = [d, 50, 50, 50]
    W = [rng.normal(scale=1.0 / np.sqrt(dims[i]), size=(dims[i + 1], dims[i])) for i in range(3)]
    b = [rng.normal(scale=0.1, size=dims[i + 1]) for i in range(3)]
    Z0 = ctx.blr_basis(W, b, "Tanh", X=X_obs)
    Y = Z0 @ rng.normal(size=(50, 1)) + 0.1 * rng.normal(size=(N, 1))
    alpha_p, beta, ymean = 1.0, 100.0, float(np.mean(Y))
    nll = ctx.blr_fit_x(W, b, "Tanh", X_obs, Y, alpha_p, beta, ymean, want_nll=True)
    f = blr.fit(blr.basis(X_obs, W, b, "Tanh"), Y, alpha_p, beta, ymean)
    assert nll == pytest.approx(float(f["nll"]), rel=1e-9, abs=1e-7)
    ctx.blr_basis(W, b, "Tanh")
    mu, var = ctx.blr_predict()
    sample = np.random.default_rng(1).choice(M, 512, replace=False)
    mu_o, var_o = blr.predict(f, blr.basis(X_hid[sample], W, b, "Tanh"))
    assert relerr(mu[sample], mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(var[sample], var_o) < REL
    ctx.score_reset()
    ctx.score_ei([float(Y.min())], 0.0)
    val, idx, ei = ctx.score_finish(1.0, download=True)
    mu_all, var_all = blr.predict(f, blr.basis(X_hid, W, b, "Tanh"))
    want = orc.c.ei(mu_all, var_all, [float(Y.min())])
    wi, wv = orc.c.argmax_first(want)
    top2 = np.partition(want, -2)[-2:]
    print("cfg5 EI max %.4g (index %d), top-2 gap %.3e, max |EI - oracle| %.3e" % (wv, wi, top2[1] - top2[0],
                                                                                 np.max(np.abs(ei - want))))
    assert wv > 1e-3, "EI at the winner should be O(1e-2 .. 1), not an underflow"
    assert idx == wi and val == pytest.approx(wv, rel=1e-6)
    # the same nomination as ONE call (b7_blr_eval_nominate): the oracle's winner, the separate calls' value; its scores
    # (left in the accumulator) against the oracle over the whole grid
    v1, i1, jit = ctx.blr_eval_nominate(W, b, "Tanh", X_obs, Y, alpha_p, beta, ymean, score="ei", fmin=[float(Y.min())],
                                        want_jitter=True)
    assert i1 == wi and v1 == pytest.approx(wv, rel=1e-6) and jit == 0.0
    _, _, ei1 = ctx.score_finish(1.0, download=True)
    assert np.max(np.abs(ei1 - want)) < 1e-9 * max(1.0, np.abs(want).max())
    v2, i2 = ctx.blr_eval_nominate(W, b, "Tanh", X_obs, Y, alpha_p, beta, ymean, score="cb")
    cbw = orc.c.cb(mu_all, var_all)
    assert i2 == orc.c.argmax_first(cbw)[0]


def test_blr_eval_nominate_small_shapes_and_activations(ctx, orc):
    """b7_blr_eval_nominate over the shapes of test_dngo_basis_and_blr_head_match_oracle (Tanh with resident weights, ReLU
    with one wide layer, Sigmoid at d = 32) and a ragged grid: winner and scores equal the oracle's."""
    from oracle import blr
    from conftest import make_network
    for d, widths, act, N, M in [(6, (50, 50, 50), "Tanh", 80, 3000), (2, (100,), "ReLU", 24, 257),
                                 (32, (64, 100), "Sigmoid", 300, 1000), (3, (20, 20), "Tanh", 10, 17)]:
        W, b = make_network(d, widths, seed=d)
        X_obs, Y, X_hid, _ = make_problem(ctx, orc, d, N, M, lambda X: np.sin(3 * X.sum(axis=1, keepdims=True)))
        alpha_p, beta, mean = 2.0, 1.0 / (1e-2 * float(np.var(Y))), float(np.mean(Y))
        ctx.grid_upload(X_hid)
        Z1 = ctx.blr_basis(W, b, act, download=True)
        assert np.allclose(Z1, blr.basis(X_hid, W, b, act), rtol=1e-12, atol=1e-13), (d, act)
        f = blr.fit(blr.basis(X_obs, W, b, act), Y, alpha_p, beta, mean)
        mu_o, var_o = blr.predict(f, blr.basis(X_hid, W, b, act))
        for kw, want in (({"score": "ei", "fmin": [float(Y.min())]}, orc.c.ei(mu_o, var_o, [float(Y.min())])),
                         ({"score": "cb"}, orc.c.cb(mu_o, var_o))):
            v, i = ctx.blr_eval_nominate(W, b, act, X_obs, Y, alpha_p, beta, mean, **kw)
            wi, wv = orc.c.argmax_first(want)
            _, _, got = ctx.score_finish(1.0, download=True)
            assert np.allclose(got, want, rtol=1e-5, atol=1e-10), (d, act, kw["score"])
            assert i == wi and v == pytest.approx(wv, rel=1e-6, abs=1e-12), (d, act, kw["score"])


# ---- the persistent Cholesky schedule (potrf_persist.hip): same bits as the launch schedule, also under load ----------
def _two_schedules(monkeypatch):
    import bot7_amd
    out = []
    for sched in ("1", "3"):
        monkeypatch.setenv("B7_POTRF_SCHED", sched)
        out.append(bot7_amd.Context(0, lib="diag"))
    monkeypatch.delenv("B7_POTRF_SCHED")
    return out


def _aborts(c):
    return c._L.b7_persist_fallbacks(c._h)


@pytest.mark.parametrize("N,d,cols", [(2, 3, 1), (64, 6, 1), (100, 6, 1), (129, 6, 1), (256, 6, 7), (700, 32, 1),
                                      (1024, 32, 1), (1500, 6, 1), (2048, 32, 1), (2100, 6, 1), (3000, 6, 1), (4096, 32, 1)])
def test_persistent_cholesky_is_bit_identical_to_the_launch_schedule(orc, monkeypatch, N, d, cols):
    """One persistent launch with flag hand-offs (default up to Npad = 4096) against two launches per panel: L, inv(L),
    alpha and the likelihood must agree bit for bit -- only the schedule differs."""
    launch, persist = _two_schedules(monkeypatch)
    try:
        obj = {3: lambda X: np.sin(3.0 * X).sum(axis=1, keepdims=True), 6: B.hartmann6, 32: B.ackley}[d]
        X_obs, Y1, _, hyp = make_problem(None, orc, d, N, 64, obj)
        Y = Y1 if cols == 1 else Y1 + 0.1 * np.random.default_rng(N).normal(size=(N, cols))
        res = []
        for c in (launch, persist):
            r = c.gp_fit(X_obs, Y, want_nll=True, **hyp)
            res.append(c.gp_download(N, cols) + (r["nll"], np.array([r["jitter"], r["info"]])))
        for a, b in zip(*res):
            assert np.array_equal(a, b)
        f = orc.gp.fit(X_obs, Y, **hyp)
        assert np.allclose(res[1][3], f.nll, rtol=1e-9, atol=1e-7)
        assert _aborts(persist) == 0
    finally:
        launch.close()
        persist.close()


def test_persistent_cholesky_jitter_retries_and_plain_chol(orc, monkeypatch):
    """The jitter schedule (eps on the diagonal of C_0 inside the persistent kernel) and b7_chol (no inverse)."""
    launch, persist = _two_schedules(monkeypatch)
    try:
        X = orc.c.sobol(300, 3, 1)
        X[7] = X[3]
        X[250] = X[100]                                  # duplicates, no noise: singular K -> retries
        Y = np.sin(3.0 * X).sum(axis=1, keepdims=True)
        hyp = dict(lenscale_sq=np.full(3, 0.4), amp=1.0, noise=0.0, mean=0.0)
        out = []
        for c in (launch, persist):
            r = c.gp_fit(X, Y, want_nll=True, **hyp)
            out.append(c.gp_download(300) + (r["nll"], np.array([r["jitter"], r["info"]])))
        assert out[0][4][0] > 0 and out[0][4][1] > 0
        for a, b in zip(*out):
            assert np.array_equal(a, b)
        rng = np.random.default_rng(2)
        A = rng.normal(size=(333, 333))
        S = A @ A.T + 333 * np.eye(333)
        (L0, j0, i0), (L1, j1, i1) = launch.chol(S), persist.chol(S)
        assert np.array_equal(L0, L1) and (j0, i0) == (j1, i1) == (0.0, 0)
        assert np.allclose(L1 @ L1.T, S, rtol=1e-12, atol=1e-9)
        assert _aborts(persist) == 0
    finally:
        launch.close()
        persist.close()


def test_persistent_cholesky_under_uneven_load(orc, monkeypatch):
    """cdna_hip_programming.md Guideline 16: a hand-off protocol is tested under UNEVEN load.  While a second context keeps
    the chip busy with posterior GEMMs of changing size, ten persistent fits in a row must reproduce the launch
    schedule's bits every time (a stale or torn tile would change L; a stalled hand-off would time out into the
    launch schedule, which is counted)."""
    import threading
    import bot7_amd
    launch, persist = _two_schedules(monkeypatch)
    noise_ctx = bot7_amd.Context(0)
    try:
        X_obs, Y, _, hyp = make_problem(None, orc, 32, 2048, 64, B.ackley)
        launch.gp_fit(X_obs, Y, **hyp)
        ref = launch.gp_download(2048)
        Xn, Yn, _, hn = make_problem(None, orc, 6, 512, 64, B.hartmann6)
        noise_ctx.gp_fit(Xn, Yn, **hn)
        stop = threading.Event()

        def load():
            k = 0
            while not stop.is_set():
                noise_ctx.grid_sobol(20000 + 37000 * (k % 5), 6, 1, download=False)
                noise_ctx.gp_predict(download=False)
                k += 1
            noise_ctx.sync()

        th = threading.Thread(target=load)
        th.start()
        try:
            for rep in range(10):
                persist.gp_fit(X_obs, Y, **hyp)
                got = persist.gp_download(2048)
                for a, b in zip(ref, got):
                    assert np.array_equal(a, b), "repetition %d differs" % rep
        finally:
            stop.set()
            th.join()
        print("persistent fits redone by the launch schedule under load:", _aborts(persist))
    finally:
        for c in (launch, persist, noise_ctx):
            c.close()


# ---- batched likelihoods for the sampler ----------------------------------------------------------------
@pytest.mark.parametrize("d,N,B", [(6, 256, 16), (6, 100, 5), (32, 500, 9), (6, 1100, 3)])
def test_nll_batch_matches_single_fits_and_oracle(ctx, orc, d, N, B):
    """b7_gp_nll_batch: B hyper vectors, one persistent launch (or a few), each NLL against b7_gp_fit_hyp and the oracle to
    1e-9; the context's current fit is left alone."""
    obj = B_OBJ[d]
    X_obs, Y, X_hid, hyp = make_problem(None, orc, d, N, 300, obj)
    ctx.gp_fit(X_obs, Y, **hyp)                       # the current fit, with predictions
    ctx.grid_upload(X_hid)
    mu0, var0 = ctx.gp_predict()
    rng = np.random.default_rng(B)
    ls = hyp["lenscale_sq"] * np.exp(rng.normal(scale=0.5, size=(B, d)))
    amp = hyp["amp"] * np.exp(rng.normal(scale=0.3, size=B))
    noise = hyp["noise"] * np.exp(rng.normal(scale=1.0, size=B))
    mean = hyp["mean"] + rng.normal(scale=0.1, size=B)
    nll, jit, info = ctx.gp_nll_batch(ls, amp, noise, mean, want_info=True)
    assert (jit == 0).all() and (info == 0).all()
    mu1, var1 = ctx.gp_predict()
    assert np.array_equal(mu0, mu1) and np.array_equal(var0, var1)      # untouched
    for b in range(B):
        f = orc.gp.fit(X_obs, Y, ls[b], amp[b], noise[b], mean[b])
        assert nll[b] == pytest.approx(float(f.nll[0]), rel=1e-9, abs=1e-7)
    single = np.array([ctx.gp_fit_hyp(ls[b], amp[b], noise[b], mean[b], want_nll=True)["nll"][0] for b in range(B)])
    assert np.allclose(nll, single, rtol=1e-11, atol=1e-9)


B_OBJ = {6: B.hartmann6, 32: B.ackley}


def test_nll_batch_jitter_and_errors(ctx, orc):
    import bot7_amd
    X = orc.c.sobol(200, 3, 1)
    X[7] = X[3]
    Y = np.sin(3.0 * X).sum(axis=1, keepdims=True)
    ctx.gp_set_data(X, Y)
    ls = np.tile(np.full(3, 0.4), (3, 1))
    nll, jit, info = ctx.gp_nll_batch(ls, [1.0, 1.0, 1.0], [0.0, 1e-3, 0.0], [0.0, 0.0, 0.1], want_info=True)
    for b in range(3):
        r = ctx.gp_fit_hyp(ls[b], 1.0, [0.0, 1e-3, 0.0][b], [0.0, 0.0, 0.1][b], want_nll=True)
        assert jit[b] == r["jitter"] and info[b] == r["info"]
        assert nll[b] == pytest.approx(float(r["nll"][0]), rel=1e-9, abs=1e-7)
    assert jit[0] > 0 and jit[1] == 0 and jit[2] > 0
    with pytest.raises(bot7_amd.Bot7HipError):
        ctx.gp_nll_batch(ls, [1.0, -1.0, 1.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0])
    ctx.gp_set_data(X, np.hstack([Y, Y]))
    with pytest.raises(bot7_amd.Bot7HipError) as e:
        ctx.gp_nll_batch(ls, 1.0, 1e-3, 0.0)
    assert e.value.code == -5


def test_nll_batch_sixteen_fits_cost_less_than_two(ctx, orc):
    """B = 16 at N = 256 within 2x one fit."""
    import time
    X_obs, Y, _, hyp = make_problem(None, orc, 6, 256, 64, B.hartmann6)
    ctx.gp_set_data(X_obs, Y)
    ls = np.tile(hyp["lenscale_sq"], (16, 1)) * np.linspace(0.8, 1.25, 16)[:, None]
    args = (ls, hyp["amp"], hyp["noise"], hyp["mean"])
    ctx.gp_nll_batch(*args)
    ctx.gp_fit_hyp(ls[0], hyp["amp"], hyp["noise"], hyp["mean"], want_nll=True)
    ctx.sync()

    def best_of(fn, reps=30):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts)

    t1 = best_of(lambda: ctx.gp_fit_hyp(ls[0], hyp["amp"], hyp["noise"], hyp["mean"], want_nll=True))
    t16 = best_of(lambda: ctx.gp_nll_batch(*args))
    print("N = 256: one fit %.3f ms, 16 likelihoods in one batch %.3f ms" % (t1 * 1e3, t16 * 1e3))
    assert t16 < 2.0 * t1


@pytest.mark.parametrize("N", [64, 700, 2048])
def test_dpp_fused_multiply_add_equals_its_two_instruction_form(orc, monkeypatch, N):
    """potrf_diag.h's fmac_share is inline asm (v_fmac_f64_dpp row_newbcast) whose read-after-write spacing is kept by
    hand; B7_DIAG_VARIANT=2 builds the same routine from v_mov_b64_dpp + v_fma_f64, which the compiler schedules and
    pads itself.  Same arithmetic, so L, inv(L) and alpha must agree bit for bit (launch schedule on both sides: the
    persistent one always uses the fused form)."""
    import bot7_amd
    ctxs = []
    for var in ("1", "2"):
        monkeypatch.setenv("B7_DIAG_VARIANT", var)
        monkeypatch.setenv("B7_POTRF_SCHED", "1")
        ctxs.append(bot7_amd.Context(0, lib="diag"))
    monkeypatch.delenv("B7_DIAG_VARIANT")
    monkeypatch.delenv("B7_POTRF_SCHED")
    try:
        d = 6 if N < 2048 else 32
        X_obs, Y, _, hyp = make_problem(None, orc, d, N, 64, B.hartmann6 if d == 6 else B.ackley)
        out = []
        for c in ctxs:
            c.gp_fit(X_obs, Y, **hyp)
            out.append(c.gp_download(N))
        for a, b in zip(*out):
            assert np.array_equal(a, b)
        f = orc.gp.fit(X_obs, Y, **hyp)
        assert np.allclose(out[0][0], f.L, rtol=1e-9, atol=1e-12)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("helpers", [1, 3, 17])
def test_persistent_cholesky_cannot_deadlock_with_few_helpers(orc, monkeypatch, helpers):
    """The job queue is a topological order of the tile dependencies, so any number of helper workgroups >= 1 must get
    through it (only slower): one, three and seventeen helpers reproduce the launch schedule's bits without a time-out."""
    import bot7_amd
    monkeypatch.setenv("B7_PERSIST_HELPERS", str(helpers))
    launch, persist = _two_schedules(monkeypatch)
    monkeypatch.delenv("B7_PERSIST_HELPERS")
    try:
        for N, d, obj in ((300, 6, B.hartmann6), (1100, 6, B.hartmann6), (2500, 6, B.hartmann6)):
            if N > 2000 and helpers == 1:
                continue                                  # 40 panels through one helper: minutes, and the order is the same
            X_obs, Y, _, hyp = make_problem(None, orc, d, N, 64, obj)
            out = []
            for c in (launch, persist):
                r = c.gp_fit(X_obs, Y, want_nll=True, **hyp)
                out.append(c.gp_download(N) + (r["nll"],))
            for a, b in zip(*out):
                assert np.array_equal(a, b)
        assert _aborts(persist) == 0
    finally:
        launch.close()
        persist.close()


def test_persistent_cholesky_times_out_into_the_launch_schedule(orc, monkeypatch):
    """Fault injection: workgroup 0 withholds one hand-off flag (B7_PERSIST_FAULT).  Its consumers must run into their
    bounded spin, raise the abort word, every workgroup must drain (the launch ends, no hang), and the host must redo
    the factorisation with the launch schedule -- same bits, one abort counted, and the context keeps working."""
    import bot7_amd
    monkeypatch.setenv("B7_PERSIST_FAULT", "2")
    launch, persist = _two_schedules(monkeypatch)
    monkeypatch.delenv("B7_PERSIST_FAULT")
    try:
        X_obs, Y, X_hid, hyp = make_problem(None, orc, 6, 500, 700, B.hartmann6)
        out = []
        for c in (launch, persist):
            r = c.gp_fit(X_obs, Y, want_nll=True, **hyp)
            out.append(c.gp_download(500) + (r["nll"],))
        for a, b in zip(*out):
            assert np.array_equal(a, b)
        assert _aborts(persist) == 1
        persist.grid_upload(X_hid)
        mu, var = persist.gp_predict()
        mu_o, var_o = orc.gp.predict(orc.gp.fit(X_obs, Y, **hyp), X_hid)
        assert relerr(mu, mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(var, var_o) < REL
        # a matrix with fewer panels than the faulty one is unaffected
        r = persist.gp_fit(X_obs[:100], Y[:100], want_nll=True, **hyp)
        assert _aborts(persist) == 1 and np.isfinite(r["nll"]).all()
        # the likelihood-only call: twice timed out -> evaluated in the context's fit slot by the launch schedule (the header's
        # exception to "leaves the current fit alone"): same value, b7_persist_fallbacks moved, the binding bumped fit_token, and
        # the next predict asks for a new fit
        persist.gp_fit(X_obs, Y, **hyp)
        launch.gp_set_data(X_obs, Y)
        persist.gp_set_data(X_obs, Y)
        tok, n0 = persist.fit_token, _aborts(persist)
        a = launch.gp_nll_batch(hyp["lenscale_sq"], hyp["amp"], hyp["noise"], hyp["mean"])
        b = persist.gp_nll_batch(hyp["lenscale_sq"], hyp["amp"], hyp["noise"], hyp["mean"])
        assert np.array_equal(a, b) and _aborts(persist) > n0 and persist.fit_token > tok
        with pytest.raises(bot7_amd.Bot7HipError):
            persist.gp_predict()
    finally:
        launch.close()
        persist.close()


def test_nll_batch_more_fits_than_fit_on_the_chip(ctx, orc):
    X_obs, Y, _, hyp = make_problem(None, orc, 6, 200, 64, B.hartmann6)
    ctx.gp_set_data(X_obs, Y)
    Bn = 70                                           # 28 fits per launch at Npad = 256: three launches
    scale = np.linspace(0.6, 1.6, Bn)
    nll = ctx.gp_nll_batch(hyp["lenscale_sq"][None, :] * scale[:, None], hyp["amp"], hyp["noise"], hyp["mean"])
    for b in (0, 27, 28, 55, 56, 69):
        f = orc.gp.fit(X_obs, Y, hyp["lenscale_sq"] * scale[b], hyp["amp"], hyp["noise"], hyp["mean"])
        assert nll[b] == pytest.approx(float(f.nll[0]), rel=1e-9, abs=1e-7)
    assert np.all(np.diff(nll) != 0)


def test_predict_hyp_is_fit_hyp_plus_predict(ctx, orc):
    """b7_gp_predict_hyp (fit and posterior enqueued back to back, the host waits for the pivot report only) against the
    two separate calls: same bits, with and without the jitter schedule kicking in after the speculative prediction."""
    X_obs, Y, X_hid, hyp = make_problem(None, orc, 6, 300, 5000, B.hartmann6)
    ctx.grid_upload(X_hid)
    ctx.gp_set_data(X_obs, Y)
    for h in (hyp, dict(hyp, lenscale_sq=hyp["lenscale_sq"] * 0.5, mean=hyp["mean"] + 0.2)):
        a = ctx.gp_fit_hyp(want_nll=True, **h)
        mu_a, var_a = ctx.gp_predict()
        b = ctx.gp_predict_hyp(download=True, want_nll=True, **h)
        assert np.array_equal(mu_a, b["mean"]) and np.array_equal(var_a, b["var"]) and a["nll"][0] == b["nll"][0]
        assert (a["jitter"], a["info"]) == (b["jitter"], b["info"]) == (0.0, 0)
        ctx.score_reset()
        ctx.score_ei([float(Y.min())], 0.0)                 # scores run on the fused prediction
        _, idx, ei = ctx.score_finish(1.0, download=True)
        assert idx == orc.c.argmax_first(orc.c.ei(*orc.gp.predict(orc.gp.fit(X_obs, Y, **h), X_hid), [float(Y.min())]))[0]
    Xd = orc.c.sobol(300, 3, 1)
    Xd[7] = Xd[3]                                            # a duplicate without noise: the plain attempt fails
    Yd = np.sin(3.0 * Xd).sum(axis=1, keepdims=True)
    ctx.grid_upload(orc.c.sobol(3000, 3, 500))
    ctx.gp_set_data(Xd, Yd)
    hj = dict(lenscale_sq=np.full(3, 0.4), amp=1.0, noise=0.0, mean=0.0)
    a = ctx.gp_fit_hyp(**hj)
    mu_a, var_a = ctx.gp_predict()
    b = ctx.gp_predict_hyp(download=True, **hj)
    assert a["jitter"] > 0 and (a["jitter"], a["info"]) == (b["jitter"], b["info"])
    assert np.array_equal(mu_a, b["mean"]) and np.array_equal(var_a, b["var"])


def test_lockstep_chains_follow_the_single_chain_sampler_and_batch_their_densities(ctx, orc):
    """config.chains = C: every chain is the reference's slice sampler with its own random stream; their density requests
    go to the device as b7_gp_nll_batch calls.  Each chain must trace what the plain sampler traces when it is given
    the same stream and start and evaluates its densities one b7_gp_fit_hyp at a time; and the C chains must need far
    fewer device round trips than C times one chain."""
    import time
    import bot7_amd
    X_obs, Y, _, hyp = make_problem(None, orc, 6, 200, 64, B.hartmann6)
    C = 8
    model = bot7_amd.models.gp_regressor({"sample": True, "chains": C, "nBurnin": 2, "seed": 3}, context=ctx)
    model.hyp = dict(hyp)
    model.sample_hypers(X_obs, Y)                                         # burn-in: two lock-step updates
    t0 = time.perf_counter()
    drawn = [model.parse_hypers(model.sample_hypers(X_obs, Y, None, None, True)) for _ in range(C)]
    t_lock = time.perf_counter() - t0
    evals, batches = model.nEvals, model.nBatches
    assert batches < evals / 3, "the chains' requests were not batched (%d batches for %d evaluations)" % (batches, evals)
    # the same update, chain by chain, through single fits
    ref = bot7_amd.models.gp_regressor({"sample": True, "seed": 3}, context=ctx)
    from harness.samplers import slice_sampler
    S = slice_sampler()
    lo, hi = model._bounds(X_obs, Y)
    t0 = time.perf_counter()
    for c in range(C):
        # replay the chain's start, burn-in and update on a fresh stream seeded like the chain's own
        rng = np.random.default_rng([3, c])
        theta = np.clip(model._to_theta(hyp) + (0.1 * rng.standard_normal(9) if c else 0.0), lo, hi)   # d + 3 = 9
        opt = dict(S.configure({"width": 0.5}), nSamples=1, rng=rng)
        f = lambda t, _a: ref.log_posterior(t, X_obs, Y)  # noqa: E731
        for _ in range(3):                                                # 2 burn-in updates + the sampled one
            theta = S.sample(f, theta.reshape(1, -1), opt, None)[0]
        got = model._to_theta(drawn[c])
        assert np.allclose(got, theta, rtol=1e-7, atol=1e-9), "chain %d left the single-chain trajectory" % c
    t_seq = time.perf_counter() - t0
    print("8 chains, one update each: lock-step %.2f ms (%d likelihoods in %d batches); chain by chain incl. burn-in replay %.2f ms"
          % (t_lock * 1e3, evals, batches, t_seq * 1e3))


# ---- ADVICE r2 ---------------------------------------------------------------------------------------------------------
def test_eval_nominate_leaves_no_half_described_fit_behind(ctx, orc):
    """After b7_eval_nominate the context's fit slot holds none of the samples (include/bot7hip.h): with one sample the fit's
    lengthscales sat in the batch staging block, and b7_gp_append / b7_gp_fantasize would have rescaled the observations with
    whatever an earlier fit left in the slot's own place.  Both must be refused until the caller fits again, and a fit +
    append afterwards equals the full refit."""
    import bot7_amd
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, 6, 50, 700, B.hartmann6)
    ctx.grid_upload(X_hid)
    ctx.gp_fit(X_obs, Y, hyp["lenscale_sq"] * 3.0, hyp["amp"], hyp["noise"], hyp["mean"])   # other lengthscales in the slot
    ctx.gp_set_data(X_obs, Y)
    for S in (1, 2):
        ctx.eval_nominate([dict(hyp, lenscale_sq=hyp["lenscale_sq"] * (1 + 0.1 * s)) for s in range(S)], score="ei", fmin=[float(Y.min())])
        with pytest.raises(bot7_amd.Bot7HipError) as e:
            ctx.gp_append(X_hid[0], B.hartmann6(X_hid[:1])[0])
        assert e.value.code == -4
        with pytest.raises(bot7_amd.Bot7HipError) as e:
            ctx.gp_fantasize(X_hid[:3], 4)
        assert e.value.code == -4
    ctx.gp_fit_hyp(**hyp)
    ctx.gp_append(X_hid[0], B.hartmann6(X_hid[:1])[0])
    L1, a1, _ = ctx.gp_download(51)
    ctx.gp_fit(np.concatenate([X_obs, X_hid[:1]]), np.concatenate([Y, B.hartmann6(X_hid[:1])]), **hyp)
    L2, a2, _ = ctx.gp_download(51)
    assert np.allclose(L1, L2, rtol=1e-9, atol=1e-12) and np.allclose(a1, a2, rtol=1e-6, atol=1e-9)


def test_stage_keys_on_the_content_of_the_observations(ctx, orc):
    """models.gp_regressor.stage decides whether the resident data are current: shapes and sums are not enough (Y and -Y
    with zero sum; two responses swapped).  The nomination after such a change must be the oracle's for the NEW data."""
    import bot7_amd
    rng = np.random.default_rng(8)
    X_obs, X_hid = rng.random((40, 3)), rng.random((900, 3))
    Y = rng.normal(size=(40, 1))
    Y -= Y.mean()
    Y[0, 0] -= Y.sum()                                # exactly zero sum for Y and -Y alike? make it so to the last bit
    model = bot7_amd.models.gp_regressor({}, context=ctx)
    grid = bot7_amd.grids.DeviceGrid(X_hid, ctx, -1)
    hyp = {"lenscale_sq": np.full(3, 0.4), "amp": 1.0, "noise": 1e-3, "mean": 0.0}
    Ys = Y.copy()
    Ys[[3, 7]] = Ys[[7, 3]]                           # same shape, same sum, different data
    for Yk in (Y, -Y, Ys):
        model.stage(X_obs, Yk, grid)
        v, i = ctx.eval_nominate([hyp], score="ei", fmin=[float(Yk.min())])
        mu, var = orc.gp.predict(orc.gp.fit(X_obs, Yk, **hyp), X_hid)
        wi, wv = orc.c.argmax_first(orc.c.ei(mu, var, [float(Yk.min())]))
        assert i == wi, "stale observations were scored"


# ---- the one-workgroup likelihood of small observation sets (csrc/nll_small.hip) ---------------------------------------
def test_small_set_likelihood_kernel_equals_the_general_path_and_the_oracle(orc, monkeypatch):
    """b7_gp_nll_batch at N <= 128, d <= 32 runs nll_small_kernel (one workgroup per evaluation, one launch); B7_NLL_SMALL=0
    sends the same call through the general path (scaling, K assembly, persistent factorisation with its vector job).  Both
    against the oracle and against each other (1e-12: same K entries, same factor routine, other summation order at the very
    end) over ragged N (1 ... 128), d (1 ... 32), batch sizes 1 ... 300; a set that needs the jitter schedule (duplicated rows,
    no noise) must fall back and report the same jitter; N = 129 and d = 33 take the general path on their own."""
    import bot7_amd
    monkeypatch.setenv("B7_NLL_SMALL", "0")
    general = bot7_amd.Context(0, lib="diag")
    monkeypatch.delenv("B7_NLL_SMALL")
    small = bot7_amd.Context(0)
    rng = np.random.default_rng(31)
    for N, d, Bn in ((1, 1, 1), (2, 3, 4), (17, 2, 7), (63, 6, 16), (64, 32, 3), (65, 5, 16), (100, 6, 40), (127, 31, 2),
                     (128, 32, 300), (129, 4, 5), (50, 33, 5)):
        X = rng.random((N, d))
        Y = np.sin(3.0 * X.sum(axis=1, keepdims=True)) + 0.05 * rng.normal(size=(N, 1))
        ls = rng.random((Bn, d)) * d * 0.3 + 0.05 * d
        amp = rng.random(Bn) + 0.5
        noise = 10.0 ** rng.uniform(-5, -2, Bn)
        mean = rng.normal(size=Bn) * 0.1
        small.gp_set_data(X, Y)
        general.gp_set_data(X, Y)
        a, ja, ia = small.gp_nll_batch(ls, amp, noise, mean, want_info=True)
        b, jb, ib = general.gp_nll_batch(ls, amp, noise, mean, want_info=True)
        assert np.max(np.abs(a - b) / np.abs(b)) < 1e-12, (N, d)
        assert np.array_equal(ja, jb) and np.array_equal(ia, ib)
        for k in range(min(Bn, 4)):
            want = float(orc.gp.fit(X, Y, ls[k], float(amp[k]), float(noise[k]), float(mean[k])).nll[0])
            assert a[k] == pytest.approx(want, rel=1e-9, abs=1e-9), (N, d, k)
        if N > 1:    # the single fit's likelihood is the same number too
            one = small.gp_fit_hyp(ls[0], float(amp[0]), float(noise[0]), float(mean[0]), want_nll=True)
            assert a[0] == pytest.approx(float(one["nll"][0]), rel=1e-12)
    # a failing plain attempt: duplicated observations without noise
    X = rng.random((40, 3))
    X = np.concatenate([X, X[:5]])
    Y = np.cos(X.sum(axis=1, keepdims=True))
    ls = np.full((3, 3), 0.4)
    for c in (small, general):
        c.gp_set_data(X, Y)
    a, ja, ia = small.gp_nll_batch(ls, 1.0, 0.0, 0.0, want_info=True)
    b, jb, ib = general.gp_nll_batch(ls, 1.0, 0.0, 0.0, want_info=True)
    assert (ja > 0).all() and np.array_equal(ja, jb) and np.array_equal(ia, ib) and np.array_equal(a, b)
    # the sampler's density goes through it: model.nll keeps the data resident and evaluates the likelihood alone
    model = bot7_amd.models.gp_regressor({}, context=small)
    Xs, Ys = rng.random((30, 4)), rng.normal(size=(30, 1))
    h = {"lenscale_sq": np.full(4, 0.7), "amp": 1.3, "noise": 1e-3, "mean": 0.2}
    v1 = float(model.nll(Xs, Ys, h)[0])
    assert v1 == pytest.approx(float(orc.gp.fit(Xs, Ys, **h).nll[0]), rel=1e-9)
    assert float(model.nll(Xs, -Ys, h)[0]) != v1     # new data of the same shape are noticed (content key)
    small.close()
    general.close()


def test_live_handles_with_null_arguments_never_crash():
    """tools/null_sweep_gpu.py: every export called with a live context (fresh, then fitted and predicted) or a live group and
    NULL pointers / zero sizes everywhere else.  Error codes or harmless successes, never a fault (child process)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools",
                                                        "null_sweep_gpu.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.returncode, out.stdout[-400:], out.stderr[-400:])
    assert "calls without a crash" in out.stdout


@pytest.mark.parametrize("d,N,M", [(1, 1, 1), (1, 1, 5), (3, 1, 300), (1, 2, 1), (96, 3, 2), (96, 130, 257), (64, 1, 129),
                                   (2, 128, 1), (39, 1, 1)])
def test_extreme_shapes_match_oracle(ctx, orc, d, N, M):
    """One observation, one candidate, one dimension, the widest input (96), and their mixes: fit, likelihood (both
    entry points), posterior and the nomination against the oracle (parity unpinned: no reference fixture for the GP algebra)."""
    rng = np.random.default_rng(1000 * d + 10 * N + M)
    X, Y, Xh = rng.random((N, d)), rng.normal(size=(N, 1)), rng.random((M, d))
    hyp = dict(lenscale_sq=np.full(d, max(d / 8.0, 0.2)), amp=1.3, noise=1e-3, mean=0.1)
    f = orc.gp.fit(X, Y, **hyp)
    out = ctx.gp_fit(X, Y, hyp["lenscale_sq"], hyp["amp"], hyp["noise"], hyp["mean"], want_nll=True)
    assert out["info"] == 0 and out["nll"][0] == pytest.approx(float(f.nll[0]), rel=1e-9, abs=1e-9)
    ctx.grid_upload(Xh)
    mu, var = ctx.gp_predict()
    mu_o, var_o = orc.gp.predict(f, Xh)
    assert relerr(mu, mu_o, floor=1e-3 * max(1e-300, np.abs(mu_o).max())) < REL and relerr(var, var_o) < REL
    ctx.gp_set_data(X, Y)
    assert ctx.gp_nll_batch(hyp["lenscale_sq"], hyp["amp"], hyp["noise"], hyp["mean"])[0] == pytest.approx(float(f.nll[0]), rel=1e-9, abs=1e-9)
    v, i = ctx.eval_nominate([hyp], score="ei", fmin=[float(Y.min())])
    acc = np.zeros(M)
    orc.c.accumulate(acc, orc.c.ei(mu_o, var_o, [float(Y.min())], 0.0))
    orc.c.divide(acc, 1.0)
    io, vo = orc.c.argmax_first(acc)
    top2 = np.sort(acc)[-2:] if M > 1 else None
    if M == 1 or top2[1] - top2[0] > 1e-9 * max(abs(top2[1]), 1e-300):
        assert i == io


def test_one_block_padding_appends_and_crosses_to_two_blocks(ctx, orc):
    """N <= 64 observations live in ONE 64-block (npad_of, post_small_kernel).  Appends inside it equal the full refit; the
    padded factor is full at 64 (refused, B7_ERR_STATE), and the model mirror's fit at N = 65 -- a rebuild in the 128
    padding -- continues the same posterior; the same sequence with B7_NPAD_SMALL=0 (128 padding throughout) agrees to 1e-9."""
    import bot7_amd
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, 4, 70, 700, B.rastrigin)
    ctx.grid_upload(X_hid)
    ctx.gp_fit(X_obs[:58], Y[:58], **hyp)
    for n in range(58, 64):
        ctx.gp_append(X_obs[n], Y[n])
        f = orc.gp.fit(X_obs[:n + 1], Y[:n + 1], **hyp)
        L, alpha, Linv = ctx.gp_download(n + 1)
        assert np.allclose(L, f.L, rtol=1e-9, atol=1e-12) and np.allclose(Linv @ f.L, np.eye(n + 1), atol=1e-8)
        mu, var = ctx.gp_predict()
        mu_o, var_o = orc.gp.predict(f, X_hid)
        assert relerr(mu, mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(var, var_o) < REL
    with pytest.raises(bot7_amd.Bot7HipError) as e:
        ctx.gp_append(X_obs[64], Y[64])
    assert e.value.code == -4
    model = bot7_amd.models.gp_regressor({}, context=ctx)
    model.hyp = hyp
    seen = []
    for n in range(62, 68):                       # the mirror extends the factor when it can and rebuilds when it must
        out = model.predict(X_obs[:n], Y[:n], X_hid, hyp)
        f = orc.gp.fit(X_obs[:n], Y[:n], **hyp)
        mu_o, var_o = orc.gp.predict(f, X_hid)
        assert relerr(out["mean"], mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(out["var"], var_o) < REL
        seen.append((np.asarray(out["mean"]).copy(), np.asarray(out["var"]).copy()))
    # the 128 padding on a context of its own
    os.environ["B7_NPAD_SMALL"] = "0"
    try:
        import subprocess
        import sys
        code = ("import os, sys, numpy as np; sys.path.insert(0, %r); import bot7_amd; c = bot7_amd.Context(0, lib='diag');"
                "d = np.load(sys.argv[1]); c.grid_upload(d['Xh']); c.gp_fit(d['X'], d['Y'], d['ls'], float(d['amp']), float(d['noise']), float(d['mean']));"
                "mu, var = c.gp_predict(); np.savez(sys.argv[2], mu=mu, var=var)") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            np.savez(os.path.join(td, "in.npz"), X=X_obs[:63], Y=Y[:63], Xh=X_hid, ls=hyp["lenscale_sq"], amp=hyp["amp"],
                     noise=hyp["noise"], mean=hyp["mean"])
            out = subprocess.run([sys.executable, "-c", code, os.path.join(td, "in.npz"), os.path.join(td, "out.npz")],
                                 capture_output=True, text=True, timeout=300)
            assert out.returncode == 0, out.stderr[-600:]
            ref = np.load(os.path.join(td, "out.npz"))
            assert np.allclose(ref["mu"].ravel(), seen[1][0].ravel(), rtol=1e-9, atol=1e-12)
            assert np.allclose(ref["var"].ravel(), seen[1][1].ravel(), rtol=1e-9, atol=1e-13)
    finally:
        del os.environ["B7_NPAD_SMALL"]


def test_completion_word_and_stream_wait_give_the_same_answers(ctx, orc):
    """b7_eval_nominate and a single b7_gp_nll_batch evaluation are answered through a word the last kernel raises in mapped host
    memory; B7_SPIN_US=0 makes the same calls wait for the stream instead (the fallback a slow answer takes): same bits."""
    import bot7_amd
    X_obs, Y, X_hid, hyp = make_problem(ctx, orc, 3, 40, 900, B.rastrigin)
    os.environ["B7_SPIN_US"] = "0"
    try:
        slow = bot7_amd.Context(0)
    finally:
        del os.environ["B7_SPIN_US"]
    got = []
    for c in (ctx, slow):
        c.grid_upload(X_hid)
        c.gp_set_data(X_obs, Y)
        hyps = [dict(hyp, lenscale_sq=hyp["lenscale_sq"] * (1 + 0.1 * s)) for s in range(3)]
        r = [c.eval_nominate(hyps[:S], score="ei", fmin=[float(Y.min())]) for S in (1, 3)]
        r.append(c.gp_nll_batch(hyp["lenscale_sq"], hyp["amp"], hyp["noise"], hyp["mean"]).tobytes())
        c.score_reset()
        got.append(r)
    assert got[0] == got[1]


# ---- round 4: the reference's own regime (N <= 128, d <= 32) in three launches -------------------------------------------
def _diag_context(monkeypatch, **env):
    """A context of the DIAGNOSTIC build (tools/_build/libbot7hip_diag.so) under the given switches (read at b7_create)."""
    import bot7_amd
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    try:
        return bot7_amd.Context(0, lib="diag")
    finally:
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.parametrize("N,d", [(2, 2), (5, 1), (16, 3), (25, 2), (48, 6), (63, 6), (64, 6), (65, 6), (80, 6), (81, 5), (96, 6), (100, 6),
                                 (112, 6), (113, 9), (127, 16), (128, 32), (100, 32), (33, 31)])
def test_small_problem_kernels_equal_the_general_schedule_bit_for_bit(ctx, orc, monkeypatch, N, d):
    """gp_small_kernel (the whole fit, or the likelihood, of a hyper vector in one eight-wave workgroup), kpost_small_kernel
    (K(X*,X) + mean + variance, K* never stored) and score_finish_slot_kernel (score:add x S + div + arg-max + record) against the
    general schedule -- observation scaling, ksx_kernel, persistent Cholesky, trmv launches, post_kernel, ei_batch / finish /
    argmax_slot -- selected in the diagnostic build by B7_FIT_SMALL=0 B7_KPOST_SMALL=0 B7_NLL_SMALL=2 (round 3's four-wave
    likelihood kernel): L, L^-1, alpha, the likelihoods, posterior mean and variance, the marginalised scores and the nominations
    must agree BIT FOR BIT, across one / two 64-blocks, partly and wholly padded 16-strips, every width class."""
    ref = _diag_context(monkeypatch, B7_FIT_SMALL="0", B7_NLL_SMALL="2", B7_KPOST_SMALL="0")
    try:
        rng = np.random.default_rng(1000 * N + d)
        X = rng.random((N, d))
        Y = np.sin(X.sum(1, keepdims=True) * 3.0) + 0.01 * rng.normal(size=(N, 1))
        ls = np.full(d, d / 8.0) * (0.5 + rng.random(d))
        hyps = [{"lenscale_sq": ls * (1 + 0.05 * s), "amp": 1.3, "noise": 1e-3, "mean": 0.1} for s in range(10)]
        outs = []
        for c in (ctx, ref):
            o = c.gp_fit(X, Y, ls, 1.3, 1e-3, 0.1, want_nll=True)
            L, al, Li = c.gp_download(N)
            c.gp_set_data(X, Y)
            nll5 = c.gp_nll_batch(np.outer(0.5 + 0.1 * np.arange(5), ls), 1.3, 1e-3, 0.1)
            nll1 = c.gp_nll_batch(ls, 1.3, 1e-3, 0.1)
            c.grid_sobol(3000 + N, d, 5, download=False)
            p = c.gp_predict_hyp(ls, 1.3, 1e-3, 0.1, download=True)
            b1 = c.eval_nominate(hyps[:1], score="ei", fmin=[float(Y.min())])
            s1 = c.score_finish(1.0, download=True)[2]
            b10 = c.eval_nominate(hyps, score="ei", fmin=[float(Y.min())])
            s10 = c.score_finish(1.0, download=True)[2]
            b3 = c.eval_nominate(hyps[:3], score="cb")
            s3 = c.score_finish(1.0, download=True)[2]
            outs.append({"L": L, "alpha": al, "Linv": Li, "fit_nll": np.asarray(o["nll"]), "nll5": nll5, "nll1": nll1, "mean": p["mean"],
                         "var": p["var"], "b1": np.array(b1), "s1": s1, "b10": np.array(b10), "s10": s10, "b3": np.array(b3), "s3": s3})
        for k in outs[0]:
            assert outs[0][k].tobytes() == outs[1][k].tobytes(), "N %d d %d: %s differs from the general schedule" % (N, d, k)
        # and against the oracle (parity unpinned: no reference fixture for the GP algebra)
        f = orc.gp.fit(X, Y, ls, 1.3, 1e-3, 0.1)
        assert relerr(outs[0]["fit_nll"], f.nll, floor=1e-9) < 1e-9
        mu_o, var_o = orc.gp.predict(f, orc.c.sobol(3000 + N, d, 5))
        assert relerr(outs[0]["mean"], mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(outs[0]["var"], var_o) < REL
    finally:
        ref.close()


def test_small_problem_kernels_on_ragged_and_tiny_grids(ctx, orc, monkeypatch):
    """The one-launch path at the edges of its shapes: one observation, a 64-block filled to the last row, one row into the second;
    grids of 1, 2, 15, 16, 17 and 31 candidates (less than a 16-candidate strip, exactly one, one and a bit); one and three hyper
    samples; EI and the confidence bound.  Bits of the general schedule, nominee of the oracle.  What the path does NOT take --
    two response columns, d = 33 -- must come out of the general path unchanged (same call, same answer as the oracle)."""
    ref = _diag_context(monkeypatch, B7_FIT_SMALL="0", B7_NLL_SMALL="2", B7_KPOST_SMALL="0")
    try:
        rng = np.random.default_rng(99)
        for N, d in ((1, 1), (3, 2), (64, 4), (65, 4)):
            X = rng.random((N, d))
            Y = np.cos(2.0 * X.sum(1, keepdims=True)) + 0.05 * rng.normal(size=(N, 1))
            ls = np.full(d, 0.4)
            hyps = [{"lenscale_sq": ls * (1 + 0.1 * s), "amp": 0.9, "noise": 1e-2, "mean": 0.2} for s in range(3)]
            for M in (1, 2, 15, 16, 17, 31):
                Xh = rng.random((M, d))
                outs = []
                for c in (ctx, ref):
                    c.gp_set_data(X, Y)
                    c.grid_upload(Xh)
                    p = c.gp_predict_hyp(ls, 0.9, 1e-2, 0.2, download=True)
                    b1 = c.eval_nominate(hyps[:1], score="ei", fmin=[float(Y.min())])
                    b3 = c.eval_nominate(hyps, score="cb")
                    s3 = c.score_finish(1.0, download=True)[2]
                    outs.append({"mean": p["mean"], "var": p["var"], "b1": np.array(b1), "b3": np.array(b3), "s3": s3})
                for k in outs[0]:
                    assert outs[0][k].tobytes() == outs[1][k].tobytes(), "N %d M %d: %s" % (N, M, k)
                acc = np.zeros(M)
                for h in hyps:
                    m_o, v_o = orc.gp.predict(orc.gp.fit(X, Y, **h), Xh)
                    orc.c.accumulate(acc, orc.c.cb(m_o, v_o))
                orc.c.divide(acc, 3.0)
                assert relerr(outs[0]["s3"], acc, floor=1e-6) < REL
                top = np.sort(acc)[-2:] if M > 1 else np.array([-np.inf, acc[0]])
                if top[1] - top[0] > 1e-6 * abs(top[1]):
                    assert int(outs[0]["b3"][1]) == orc.c.argmax_first(acc)[0]
        # outside the path's range: the same calls, through the general schedule
        X, Xh = rng.random((20, 33)), rng.random((50, 33))
        Y = np.sin(X.sum(1, keepdims=True))
        ctx.gp_fit(X, Y, np.full(33, 4.0), 1.0, 1e-3, 0.0)
        ctx.grid_upload(Xh)
        mu, var = ctx.gp_predict()
        mu_o, var_o = orc.gp.predict(orc.gp.fit(X, Y, np.full(33, 4.0), 1.0, 1e-3, 0.0), Xh)
        assert relerr(mu, mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(var, var_o) < REL
        X, Xh = rng.random((30, 3)), rng.random((40, 3))
        Y2 = np.stack([np.sin(X.sum(1)), np.cos(X.sum(1))], axis=1)
        ctx.gp_fit(X, Y2, [0.3] * 3, 1.0, 1e-3, 0.0)
        ctx.grid_upload(Xh)
        mu, var = ctx.gp_predict()
        f2 = orc.gp.fit(X, Y2, [0.3] * 3, 1.0, 1e-3, 0.0)
        mu_o, var_o = orc.gp.predict(f2, Xh)
        assert mu.shape == (40, 2) and relerr(mu, mu_o, floor=1e-3 * np.abs(mu_o).max()) < REL and relerr(var, var_o) < REL
    finally:
        ref.close()


def test_small_fit_kernel_hands_a_failed_pivot_to_the_jitter_schedule(ctx, orc):
    """Duplicated observations and no noise: gp_small_kernel reports the pivot, b7_gp_fit assembles K through the general front
    end and runs utils/math.lua:159-218's retries; b7_eval_nominate redoes the nomination per sample.  Jitter, first failing
    pivot and the posterior follow the oracle's schedule (N = 40 + 7 and N = 90 + 7: one and two blocks)."""
    for n0 in (40, 90):
        X_obs, Y, X_hid, hyp = make_problem(ctx, orc, 6, n0, 700, B.hartmann6)
        Xd, Yd = np.concatenate([X_obs, X_obs[:7]]), np.concatenate([Y, Y[:7]])
        h = dict(hyp, noise=0.0)
        out = ctx.gp_fit(Xd, Yd, want_nll=True, **h)
        f = orc.gp.fit(Xd, Yd, **h)
        # (the first failing pivot itself is a rounding matter on a singular matrix: LAPACK's and the kernel's need not coincide)
        assert out["jitter"] == f.jitter and out["jitter"] > 0 and 40 <= out["info"] <= n0 + 7 and f.info > 0
        ctx.grid_upload(X_hid)
        mu, var = ctx.gp_predict()
        mu_o, var_o = orc.gp.predict(f, X_hid)
        assert relerr(mu, mu_o, floor=1e-3 * np.abs(mu_o).max()) < 1e-4      # cond(K + eps I) ~ 1e8 here
        ctx.gp_set_data(Xd, Yd)
        v, i, rep = ctx.eval_nominate([h, dict(h, amp=h["amp"] * 1.1)], score="cb", want_report=True)
        assert (rep["jitter"] > 0).all() and (rep["info"] > 0).all()
        acc = np.zeros(X_hid.shape[0])
        for hh in (h, dict(h, amp=h["amp"] * 1.1)):
            m_o, v_o = orc.gp.predict(orc.gp.fit(Xd, Yd, **hh), X_hid)
            orc.c.accumulate(acc, orc.c.cb(m_o, v_o))
        orc.c.divide(acc, 2.0)
        top2 = np.partition(acc, -2)[-2:]
        if top2[1] - top2[0] > 1e-6 * abs(top2[1]):
            assert i == orc.c.argmax_first(acc)[0]


@pytest.mark.parametrize("S", [1, 4, 10])
def test_dngo_head_marginalised_over_hyper_samples(ctx, orc, S):
    """models/dngo.lua:109,174 (hyp = 'marginalize') as b7_blr_eval_nominate_marg at BASELINE config 5's full shape (3 x 50 tanh
    basis, N = 256, 65536 candidates): S samples (alpha, beta, mean), S heads over the same features, score:add per sample,
    score:div(S), score:max(1) -- against the oracle's loop (oracle/blr.py per sample, cport.accumulate / divide / argmax_first) over
    the WHOLE grid: the winner, its value, the marginalised scores left in the accumulator, the evidence of every head; S = 1
    must be b7_blr_eval_nominate.  A 70-feature head (two 64-blocks) takes the head-by-head path and must agree as well.
    PARITY UNPINNED: gp.models.bayes_linear's own marginalisation lives in the absent `gp` package."""
    from oracle import blr
    import bench
    d, N, M = 5, 256, 65536
    X_obs = bench.make_inputs(ctx, d, N, M, 0, M)
    X_hid = ctx.grid_download()
    rng = np.random.default_rng(0)
    for width in ((50, 50, 50), (50, 70)):
        dims = [d] + list(width)
        W = [rng.normal(scale=1.0 / np.sqrt(dims[i]), size=(dims[i + 1], dims[i])) for i in range(len(width))]
        b = [rng.normal(scale=0.1, size=dims[i + 1]) for i in range(len(width))]
        Z0 = blr.basis(X_obs, W, b, "Tanh")
        Y = Z0 @ rng.normal(size=(width[-1], 1)) + 0.1 * rng.normal(size=(N, 1))
        al = 0.5 + rng.random(S) * 2.0
        be = 50.0 + rng.random(S) * 100.0
        mn = float(np.mean(Y)) + 0.05 * rng.normal(size=S)
        Z1 = blr.basis(X_hid, W, b, "Tanh")
        for kind in ("ei", "cb"):
            acc = np.zeros(M)
            nll_o = []
            for s_ in range(S):
                f = blr.fit(Z0, Y, al[s_], be[s_], mn[s_])
                nll_o.append(float(f["nll"]))
                m_o, v_o = blr.predict(f, Z1)
                orc.c.accumulate(acc, orc.c.ei(m_o, v_o, [float(Y.min())]) if kind == "ei" else orc.c.cb(m_o, v_o))
            orc.c.divide(acc, float(S))
            wi, wv = orc.c.argmax_first(acc)
            v, i, jit, nll = ctx.blr_eval_nominate_marg(W, b, "Tanh", X_obs, Y, al, be, mn, score=kind, fmin=[float(Y.min())],
                                                        want_nll=True)
            _, _, got = ctx.score_finish(1.0, download=True)
            assert jit == 0.0
            assert np.max(np.abs(got - acc)) < 1e-8 * max(1.0, np.abs(acc).max())
            top2 = np.partition(acc, -2)[-2:]
            if top2[1] - top2[0] > 1e-7 * max(1.0, abs(top2[1])):
                assert i == wi
            assert v == pytest.approx(wv, rel=1e-6, abs=1e-9)
            assert np.allclose(nll, nll_o, rtol=1e-9, atol=1e-7)
            if S == 1:
                v1, i1 = ctx.blr_eval_nominate(W, b, "Tanh", X_obs, Y, al[0], be[0], mn[0], score=kind, fmin=[float(Y.min())])
                assert (i1, v1) == (i, pytest.approx(v, rel=1e-12))
