"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares, fails loudly without
a GPU, and the host-side mirrors keep the reference's bookkeeping semantics.  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

import bot7_amd
from bot7_amd import _lib
from harness import dist
from harness import tensor as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "bot7hip.h")).read()
    declared = sorted(set(re.findall(r"\b(b7_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    lib = ctypes.CDLL(bot7_amd.lib_path())
    for name in declared:
        assert hasattr(lib, name), "libbot7hip.so does not export %s" % name
    assert sorted(_lib.SYMBOLS) == declared, "python binding list and header disagree"
    L = _lib.load()
    assert L.b7_abi_version() == 1


def test_no_cpu_fallback_without_gpu():
    if _has_gpu():
        pytest.skip("GPU present: the failure path is exercised on the CPU box")
    with pytest.raises(bot7_amd.Bot7HipError) as e:
        bot7_amd.Context(0)
    assert e.value.code == -2 and "no CPU path" in str(e.value)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bot7_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "b7oracle" not in src, f
                # nor the test / bench harness (Python stand-ins for the reference's Lua host code) or the tests' RCCL double
                assert not re.search(r"^\s*(from|import)\s+harness\b", src, re.M), f
                assert "rccl_shm_stub" not in src, f


def test_sobol_direction_numbers_match_oracle(orc):
    for d in (1, 2, 7, 20, 39):
        assert np.array_equal(_lib.sobol_direction_numbers(d).astype(np.float64), orc.c.sobol_bank(d))
    with pytest.raises(bot7_amd.Bot7HipError):
        _lib.sobol_direction_numbers(40)


def test_tensor_steal_remove_append(orc):
    src = np.arange(24.0).reshape(6, 4)
    res, rest = T.steal(None, src, [3])
    assert np.array_equal(res, src[2:3]) and np.array_equal(rest, orc.c.remove_row(src, 3))
    res, rest = T.steal(res, rest, [1])
    assert np.array_equal(res, src[[2, 0]]) and np.array_equal(rest, src[[1, 3, 4, 5]])
    assert T.remove(src[:1], [1]) is None                      # utils/tensor.lua:165-169
    assert T.append(None, np.array([1.0, 2.0])).shape == (1, 2)  # :148-150 row-vector promotion


def test_score_and_bot_defaults():
    ei = bot7_amd.scores.expected_improvement()
    assert ei.config == {"tradeoff": 0.0, "nFantasies": 100}     # scores/expected_improvement.lua:30-31
    cb = bot7_amd.scores.confidence_bound()
    assert cb.config == {"tradeoff": 1.0, "nFantasies": 100, "bound": "lower", "sign": -1.0}  # :31-34
    assert bot7_amd.scores.confidence_bound({"tradeoff": 0.0}).config["tradeoff"] == 0.0  # Lua: 0 is truthy
    m = bot7_amd.models.gp_regressor({})
    assert m.class_() == "bot7.models.gp_regressor"
    h = m.parse_hypers([0.5, 0.25, 2.0, 1e-3, -1.0])
    assert list(h["lenscale_sq"]) == [0.5, 0.25] and (h["amp"], h["noise"], h["mean"]) == (2.0, 1e-3, -1.0)
    with pytest.raises(AssertionError):
        bot7_amd.grids.sobol({"size": 10, "dims": 40})           # grids/sobol.lua:36


def test_shard_ranges_cover_without_overlap():
    for M, G in [(10, 3), (2 ** 20, 8), (7, 8), (0, 4), (2097152, 8)]:
        rs = [dist.shard_range(M, r, G) for r in range(G)]
        assert rs[0][0] == 0 and rs[-1][1] == M and all(rs[i][1] == rs[i + 1][0] for i in range(G - 1))
        assert max(hi - lo for lo, hi in rs) - min(hi - lo for lo, hi in rs) <= 1


def test_pick_winner_th_semantics(orc):
    rng = np.random.default_rng(0)
    s = rng.normal(size=1000)
    s[[100, 700]] = s.max() + 1
    for G in (1, 2, 3, 8):
        pairs = []
        for r in range(G):
            lo, hi = dist.shard_range(s.size, r, G)
            i, v = orc.c.argmax_first(s[lo:hi])
            pairs.append((v, lo + i))
        assert dist.pick_winner(pairs) == (s[100], 101)
    s[[900, 300]] = np.nan
    pairs = []
    for r in range(4):
        lo, hi = dist.shard_range(s.size, r, 4)
        i, v = orc.c.argmax_first(s[lo:hi])
        pairs.append((v, lo + i))
    v, i = dist.pick_winner(pairs)
    assert np.isnan(v) and i == 301 == orc.c.argmax_first(s)[0]
    assert dist.pick_winner([(1.0, 0), (0.5, 7)]) == (0.5, 7)  # empty shard ignored


def test_slice_sampler_defaults_and_distribution():
    """samplers/slice.lua: defaults (:32-48) and the update itself on densities with known moments."""
    from harness.samplers import slice_sampler
    S = slice_sampler()
    opt = S.configure({})
    assert opt["max_step"] == 1e3 and opt["nSamples"] == 1 and opt["step_out"] is True and opt["logspace"] is True
    assert S.configure({"step_out": False, "logspace": False})["step_out"] is False
    # 2-D Gaussian, log-space, random directions: chain of 4000 updates
    mu, sd = np.array([1.0, -2.0]), np.array([0.5, 2.0])
    logp = lambda x, _a: float(-0.5 * np.sum(((x.ravel() - mu) / sd) ** 2))  # noqa: E731
    opt = S.configure({"seed": 3, "width": 1.0})
    x = np.zeros((1, 2))
    chain = []
    for _ in range(4000):
        x = S.sample(logp, x, opt)
        chain.append(x[0].copy())
    chain = np.array(chain[500:])
    assert np.allclose(chain.mean(0), mu, atol=0.15) and np.allclose(chain.std(0), sd, rtol=0.15)
    # Gibbs variant, linear space, bounded support: uniform on [0,1]^2
    p = lambda x, _a: 1.0 if ((x >= 0) & (x <= 1)).all() else 0.0  # noqa: E731
    opt = S.configure({"seed": 4, "width": 0.3, "gibbs": True, "logspace": False})
    x = np.full((1, 2), 0.5)
    chain = []
    for _ in range(3000):
        x = S.sample(p, x, opt)
        chain.append(x[0].copy())
    chain = np.array(chain)
    assert chain.min() >= 0 and chain.max() <= 1 and np.allclose(chain.mean(0), 0.5, atol=0.05)
    # deterministic for a fixed seed; nSamples rows all start from X0
    a = S(logp, np.zeros((1, 2)), {"seed": 9, "nSamples": 3})
    b = S(logp, np.zeros((1, 2)), {"seed": 9, "nSamples": 3})
    assert a.shape == (3, 2) and np.array_equal(a, b)


def test_missing_library_fails_loudly(monkeypatch):
    """No silent fallback: without the shared library the loader raises instead of computing elsewhere."""
    monkeypatch.setattr(_lib, "_libs", {})
    monkeypatch.setattr(_lib, "_SO", os.path.join(ROOT, "bot7_amd", "no_such_libbot7hip.so"))
    with pytest.raises(bot7_amd.Bot7HipError) as e:
        _lib.load()
    assert "no CPU fallback" in str(e.value)


def test_device_grid_tag_does_not_propagate_to_derived_arrays():
    """Only the explicit constructor marks an array as "resident on ctx" (__array_finalize__ used to copy
    ctx/version onto g*2+0.1 and g[::-1], and predict then skipped the upload)."""
    from bot7_amd.grids.abstract import DeviceGrid

    class FakeCtx(object):
        grid_version = 5

    g = DeviceGrid(np.arange(12.0).reshape(4, 3), FakeCtx(), 5)
    assert g.ctx is not None and g.version == 5
    for derived in (g * 2 + 0.1, g[::-1], g[1:], g.copy(), np.sqrt(g), g.T):
        assert getattr(derived, "ctx", None) is None and getattr(derived, "version", -1) == -1


def test_sobol_skip_zero_is_kept():
    """`config.skip or 1` (grids/sobol.lua:70): in Lua 0 is truthy, so skip = 0 stays 0."""
    calls = []

    class FakeCtx(object):
        grid_version = 0

        def grid_sobol(self, size, dims, skip, mins, maxes):
            calls.append(skip)
            return np.zeros((size, dims))

    for skip, want in ((None, 1), (0, 0), (7, 7)):
        cfg = {"size": 4, "dims": 2}
        if skip is not None:
            cfg["skip"] = skip
        bot7_amd.grids.sobol(cfg, context=FakeCtx())()
        assert calls[-1] == want


def test_library_winner_rule_matches_th_max_for_any_world_size(orc):
    """csrc/comm.hip's pick_winner (what b7_score_finish_global applies to the all-reduced table) on tables of 1..64
    ranks built from sharded score vectors: the same (value, index) as TH's max on the unsharded vector -- ties to the
    lowest global index, the first NaN wins, -0.0 and NaN payloads intact, empty shards ignored.  Host-only."""
    rng = np.random.default_rng(7)
    for case in range(6):
        s = rng.normal(size=5003)
        if case >= 1:
            s[[100, 700, 4000]] = s.max() + 1.0
        if case >= 3:
            s[[900, 300]] = np.nan
        if case == 5:
            s[:] = -0.0
        wi, wv = orc.c.argmax_first(s)
        for G in (1, 2, 3, 8, 64):
            pairs = []
            for r in range(G):
                lo, hi = dist.shard_range(s.size, r, G)
                if hi > lo:
                    i, v = orc.c.argmax_first(s[lo:hi])
                    pairs.append((v, lo + i))
                else:
                    pairs.append((0.0, 0))
            v, i = _lib.comm_pick_winner(pairs)
            assert i == wi and (v == wv or (v != v and wv != wv)), (case, G)
            assert np.signbit(v) == np.signbit(wv) or v != v
            assert dist.pick_winner(pairs)[1] == wi                 # the Python restatement used by the gloo rehearsal
    assert _lib.comm_pick_winner([(1.0, 0), (0.5, 7), (9.0, 0)]) == (0.5, 7)
    with pytest.raises(bot7_amd.Bot7HipError):
        _lib.comm_pick_winner([(1.0, 0), (2.0, 0)])


def test_posterior_accumulators_stay_out_of_the_compilers_hands():
    """post_kernel_w4 addresses its accumulators as a0..a255 by number inside inline asm.  In the ISA that ships no
    compiler-generated instruction may name an AGPR, the kernel has no scratch, each of the 32 tiles is started once
    from the literal 0 and all 256 registers are read back (bot7_amd/build.py runs the same check on every build)."""
    from bot7_amd import build
    stats = build.check_agpr_discipline()
    for nj, tiles in ((2, 16), (4, 32), (16, 32)):   # 128- and 256-candidate workgroups on 128-row n-tiles; the tall shape
        st = stats[nj]
        assert st["mfma"] >= 34 * tiles and st["mfma_from_zero"] == tiles and st["acc_reads"] == 8 * tiles and st["scratch"] == 0


def test_every_entry_point_refuses_null_arguments_without_crashing():
    """The boundary is plain C: a host that passes a NULL handle / NULL pointers / zero sizes (a LuaJIT cdata that was never
    filled in) must get an error code back from every one of the exports, not a segmentation fault.  Run in a child
    process so that a crash would fail this test instead of ending the run."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
from bot7_amd import _lib
L = _lib.load()
bad = []
for name in _lib.SYMBOLS:
    fn = getattr(L, name)
    args = [0 if t in (C.c_int, C.c_int64, C.c_uint64) else (0.0 if t is C.c_double else None) for t in fn.argtypes]
    rc = fn(*args)
    if fn.restype is C.c_int and name != "b7_abi_version" and not rc < 0:
        bad.append((name, rc))
assert not bad, bad
print("swept", len(_lib.SYMBOLS))
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stderr[-800:])
    assert "swept %d" % len(_lib.SYMBOLS) in out.stdout


def test_default_regime_loop_on_the_oracle_context_is_deterministic():
    """harness/default_regime.py (the trial loop bench.py --workload default times) on oracle/hostctx.py, the device's stand-in:
    six trials of the reference's default experiment on a 2000-candidate grid, twice -- same nominees, same responses, same
    hyper draws; every nominee a distinct 1-based index into the ORIGINAL grid; the per-trial clock split is there."""
    from harness import default_regime as dr
    from oracle.hostctx import OracleContext
    runs = [dr.run(OracleContext(), budget=6, grid_size=2000) for _ in range(2)]
    a, b = runs
    assert a["nominees"] == b["nominees"] and len(a["nominees"]) == 6 and len(set(a["nominees"])) == 6
    assert all(1 <= i <= 2000 for i in a["nominees"])
    assert np.array_equal(a["Y"], b["Y"]) and a["X"].shape == (6, 6)
    assert len(a["draws"]) == 6 and all(np.array_equal(x, y) for x, y in zip(a["draws"], b["draws"]))
    assert [len(v) for v in a["draws"]] == [0, 0, 10, 10, 10, 10]      # nInitial = 2 random picks first, then ten hyper vectors a trial
    assert dr.agreement(a, b) == (6, 0.0)
    split = dr.summarise(a["per_trial"])
    assert split["model_based_trials"] == 4 and split["nll_calls"] > 40


def test_density_memo_follows_the_data_and_the_point():
    """bot7_amd.models.gp_regressor's sampler density remembers its last evaluation (a slice update starts where the previous one
    ended): the same point under the same data is answered without a library call, the same point under NEW data is not, and a
    hyper table put in from outside restarts the chain from it."""
    import bot7_amd
    from harness import default_regime  # noqa: F401  (registers the sampler the model mirror looks up)

    class Ctx(object):
        fit_token, calls, sets = 0, 0, 0

        def gp_set_data(self, X, Y):
            self.sets += 1
            self._data_d = X.shape[1]
            self._bias = float(Y.sum())

        def gp_nll1(self, ls, amp, noise, mean):
            self.calls += 1
            return (float(np.sum(np.log(ls))) + self._bias, 0.0, 0)

    ctx = Ctx()
    m = bot7_amd.models.gp_regressor({"sample": True, "nBurnin": 0, "seed": 3}, context=ctx)
    rng = np.random.default_rng(5)
    X, Y = rng.random((20, 3)), rng.normal(size=(20, 1))
    m.init(X, Y)
    f = m._density(X, Y)
    t = m._to_theta(m.hyp)
    a = f(t, None)
    assert ctx.calls == 1 and f(t.copy(), None) == a and ctx.calls == 1          # same point, same data: no call
    a2 = f(t + 1e-3, None)
    assert a2 != a and ctx.calls == 2                                            # another point: a call
    X2, Y2 = np.vstack([X, rng.random((1, 3))]), np.vstack([Y, [[0.7]]])
    f2 = m._density(X2, Y2)
    b = f2(t + 1e-3, None)                                                        # the memo's point, but new data: a call
    assert ctx.calls == 3 and ctx.sets == 2 and b != a2
    # the chain continues from the vector it returned, bit for bit, unless somebody replaced the hyper table
    v1 = m.sample_hypers(X2, Y2, None, None, True)
    kept = m._chain_state[0].copy()
    n0 = ctx.calls
    m.sample_hypers(X2, Y2, None, None, True)
    assert ctx.calls > n0
    m.hyp = m.parse_hypers(v1)
    m.sample_hypers(X2, Y2, None, None, True)                                     # starts from log(exp(.)) of v1 again: fine, just not the memo
    assert np.all(np.isfinite(m._chain_state[0])) and kept.shape == m._chain_state[0].shape
