"""Synthetic objectives on [0,1]^d (benchmarks/*.lua), vectorised over rows.  Host-side data generators for
the harness (the objective is the user's black box, evaluated on one point per trial: bots/abstract.lua:124);
nothing here is on the accelerated path."""
import numpy as np


def _rows(X, d=None):
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X.reshape(1, -1)
    if d is not None:
        assert X.shape[1] == d
    return X


def braninhoo(X):
    """benchmarks/braninhoo.lua:24-44; minima ~0.397887 at (0.124,0.818), (0.543,0.152), (0.962,0.165) (:12-14)."""
    X = _rows(X, 2)
    c1, c2, c3 = -5.1 / (4.0 * np.pi * np.pi), 5.0 / np.pi, 10.0 - 10.0 / (8.0 * np.pi)
    z1, z2 = X[:, 0] * 15.0 - 5.0, X[:, 1] * 15.0
    return ((z2 + z1 ** 2 * c1 + z1 * c2 - 6.0) ** 2 + (np.cos(z1) * c3 + 10.0)).reshape(-1, 1)


_H6_A = np.array([[10.0, 3.00, 17.0, 3.50, 1.70, 8.00], [0.05, 10.0, 17.0, 0.10, 8.00, 14.0],
                  [3.00, 3.50, 1.70, 10.0, 17.0, 8.00], [17.0, 8.00, 0.05, 10.0, 0.10, 14.0]])
_H6_P = np.array([[.1312, .1696, .5569, .0124, .8283, .5886], [.2329, .4135, .8307, .3736, .1004, .9991],
                  [.2348, .1451, .3522, .2883, .3047, .6650], [.4047, .8828, .8732, .5743, .1091, .0381]])
_H6_a = np.array([1.0, 1.2, 3.0, 3.2])


def hartmann6(X):
    """benchmarks/hartmann6.lua:36-63; f* = -3.32237 at (.201690,.150011,.476874,.275332,.311652,.657300) (:24-26)."""
    X = _rows(X, 6)
    inner = -np.einsum("kj,nkj->nk", _H6_A, (X[:, None, :] - _H6_P[None]) ** 2)
    return (-(np.exp(inner) @ _H6_a)).reshape(-1, 1)


def ackley(X):
    """benchmarks/ackley.lua:28-49; 0 at x = 0.5 (:16-17)."""
    Z = (_rows(X) - 0.5) * 65.536
    a, b, c, d = 20.0, -0.2, 2.0 * np.pi, np.exp(1.0)
    return (np.exp(np.sqrt((Z ** 2).mean(1)) * b) * -a - np.exp(np.cos(Z * c).mean(1)) + (a + d)).reshape(-1, 1)


def rastrigin(X):
    """benchmarks/rastrigin.lua:26-45."""
    Z = (_rows(X) - 0.5) * 10.24
    a, b = 10.0, 2.0 * np.pi
    return ((Z ** 2 + np.cos(Z * b) * -a).sum(1) + a * Z.shape[1]).reshape(-1, 1)


registry = {"braninhoo": braninhoo, "hartmann6": hartmann6, "ackley": ackley, "rastrigin": rastrigin}
