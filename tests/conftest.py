import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): C restatement + numpy/scipy GP."""
    from oracle import cport, gp
    cport.lib()

    class O(object):
        pass

    o = O()
    o.c = cport
    o.gp = gp
    return o


@pytest.fixture(scope="session", autouse=True)
def _diag_build():
    """The diagnostic build of the library (A/B switches, fault injector, RCCL override: -DB7_DIAG) beside the shipped one; a
    no-op when it is up to date (__graft_entry__.build() makes it)."""
    from bot7_amd import build as B
    return B.build_diag()


@pytest.fixture(scope="session")
def ctx():
    """A libbot7hip context on cuda:0.  Fails loudly (no skip, no fallback) when the GPU or the .so is missing."""
    import bot7_amd
    c = bot7_amd.Context(0)
    yield c
    c.close()


def make_problem(ctx_or_none, orc, d, N, M, objective, seed_skip=1):
    """SURVEY 8(d) synthetic inputs: Sobol pool of M+N points, strided pick of N observations removed in
    ascending order, candidates = the remaining M rows in order.  Built from the ORACLE's Sobol."""
    pool = orc.c.sobol(M + N, d, seed_skip)
    step = (M + N) // N
    obs_idx = np.arange(N) * step
    mask = np.ones(M + N, dtype=bool)
    mask[obs_idx] = False
    X_obs, X_hid = pool[obs_idx].copy(), pool[mask].copy()
    Y = objective(X_obs)
    amp = float(np.var(Y))
    hyp = {"lenscale_sq": np.full(d, d / 8.0), "amp": amp, "noise": 1e-4 * amp, "mean": float(np.mean(Y))}
    return X_obs, Y, X_hid, hyp


def make_network(d, widths, seed=0):
    """A fixed random 'trained' basis network: weights for nn.Linear(d, w1), nn.Linear(w1, w2), ..."""
    rng = np.random.default_rng(seed)
    dims = [d] + list(widths)
    weights = [rng.normal(scale=1.0 / np.sqrt(dims[i]), size=(dims[i + 1], dims[i])) for i in range(len(widths))]
    biases = [rng.normal(scale=0.1, size=dims[i + 1]) for i in range(len(widths))]
    return weights, biases
