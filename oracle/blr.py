"""Basis-network forward pass and Bayesian linear regression head on the CPU.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED for the head: models/dngo.lua:174 calls gp.models.bayes_linear:predict(Z0, Y0, Z1, nil,
'marginalize', req) from the absent, unversioned `gp` package; the reference holds no fixture for it.  Restated as
the standard model of the paper the file cites (Snoek et al. 2015, "Scalable Bayesian Optimization Using Deep
Neural Networks"; Bishop PRML 3.3): w ~ N(0, alpha^-1 I), y = m0 + phi(x)'w + eps, eps ~ N(0, beta^-1).
The forward pass follows models/dngo.lua:155-171 with the layer stack nnTools/builder.lua:118-151 builds
(nn.Linear then activation, repeated)."""
import numpy as np

ACT = {None: lambda v: v, "Identity": lambda v: v, "Tanh": np.tanh, "ReLU": lambda v: np.maximum(v, 0.0),
       "Sigmoid": lambda v: 1.0 / (1.0 + np.exp(-v))}


def basis(X, weights, biases, activation="Tanh"):
    Z = np.asarray(X, dtype=np.float64)
    for W, b in zip(weights, biases):
        Z = ACT[activation](Z @ np.asarray(W, dtype=np.float64).T + np.asarray(b, dtype=np.float64))
    return Z


def fit(Z0, Y0, alpha_prec, beta, mean=0.0):
    Z0 = np.asarray(Z0, dtype=np.float64)
    r = np.asarray(Y0, dtype=np.float64).ravel() - mean
    z = Z0.shape[1]
    K = beta * (Z0.T @ Z0) + alpha_prec * np.eye(z)
    L = np.linalg.cholesky(K)
    q = beta * (Z0.T @ r)
    m = np.linalg.solve(K, q)
    N = Z0.shape[0]
    Em = 0.5 * beta * (r @ r) - 0.5 * (q @ m)
    nll = -(0.5 * z * np.log(alpha_prec) + 0.5 * N * np.log(beta) - Em - np.sum(np.log(np.diag(L)))
            - 0.5 * N * np.log(2.0 * np.pi))
    return {"K": K, "L": L, "m": m, "beta": beta, "mean": mean, "nll": nll}


def predict(f, Z1):
    Z1 = np.asarray(Z1, dtype=np.float64)
    mu = f["mean"] + Z1 @ f["m"]
    V = np.linalg.solve(f["L"], Z1.T)
    var = 1.0 / f["beta"] + np.einsum("ij,ij->j", V, V)
    return mu.reshape(-1, 1), var
