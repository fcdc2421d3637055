"""GP regression (Cholesky form) on the CPU with BLAS/LAPACK.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  ``bot7.models`` is ``require('gp.models')`` (models/init.lua:15): the GP regressor,
ARD-SE kernel, iso Gaussian noise model and constant mean live in the luarocks package ``gp`` (gpTorch7,
README.md:7), which the rockspec lists without a version (bot7-scm-1.rockspec:16-19) and which is not
under /root/reference.  The reference holds no test, fixture or golden output at that boundary.  What is
restated here is therefore the published textbook algorithm (Rasmussen & Williams, Alg. 2.1) anchored on
the reference's call sites:
    scores/expected_improvement.lua:63   pred = model:predict(X_obs, Y_obs, X_hid, hyp, {mean=true, var=true})
    scores/confidence_bound.lua:63       (same)
    bots/bayesopt.lua:68,74-75           sample_hypers / parse_hypers
and on the in-repo numerics the GP is built from:
    utils/math.lua:65-111    pdist: D = X^2 w (+) (Z^2 w)' - 2 X (Z' .* w), w = 1/lenscale, clamp >= 0
    utils/math.lua:159-218   chol: potrf with the growing-jitter retry schedule

Free choices that the reference does not pin (all explicit parameters, mirrored by include/bot7hip.h):
    lenscale_sq   the vector handed to pdist as `lenscale` (it divides squared differences)
    kernel        amp * exp(-0.5 * D);  K(X,X) gets `noise` added on the diagonal
    mean          constant m; alpha = K^-1 (y - m);  mu = m + K* alpha
    variance      latent: amp - colsumsq(L^-1 K*');  `var_with_noise` adds `noise`; `var_min` clamps below
"""
import numpy as np
from scipy.linalg import lapack, solve_triangular


def pdist(X, Z, lenscale_sq):
    """utils/math.lua:65-111, GEMM form (dgemm does the inner products, as torch.mm does)."""
    X = np.asarray(X, dtype=np.float64)
    w = 1.0 / np.asarray(lenscale_sq, dtype=np.float64).ravel()
    xss = (X * X) @ w
    if Z is None:
        Zw = X.T * w[:, None]
        zss = xss
    else:
        Z = np.asarray(Z, dtype=np.float64)
        Zw = Z.T * w[:, None]
        zss = (Z * Z) @ w
    D = X @ Zw
    D *= -2.0
    D += xss[:, None]
    D += zss[None, :]
    np.maximum(D, 0.0, out=D)
    return D


def ardse(X, Z, lenscale_sq, amp):
    D = pdist(X, Z, lenscale_sq)
    D *= -0.5
    np.exp(D, out=D)
    D *= amp
    return D


def chol_jitter(K, eps=1e-8, growth=1.1):
    """utils/math.lua:159-218 around LAPACK dpotrf (what torch.potrf calls, :165).

    Returns (L, jitter, info_first) where jitter = 0 if the first attempt succeeded, the eps used
    otherwise, -1 if the schedule fell through to chol(I); info_first = dpotrf's info of attempt 1."""
    L, info = lapack.dpotrf(K, lower=1, clean=1)
    if info == 0:
        return L, 0.0, 0
    info_first = int(info)
    max_eps = np.linalg.norm(K)  # src:norm() = Frobenius, :174
    if np.isnan(max_eps):
        raise ValueError("K contains NaN (the reference's loop would never terminate: eps > NaN is false)")
    n = K.shape[0]
    while True:
        if eps > max_eps:  # :184-186
            return np.eye(n), -1.0, info_first
        eps = eps * growth  # :188
        Kj = K.copy()
        Kj[np.diag_indices(n)] += eps  # :189-190 eps added to the ORIGINAL matrix
        L, info = lapack.dpotrf(Kj, lower=1, clean=1)
        if info == 0:
            return L, eps, info_first


class Fit(object):
    __slots__ = ("X", "L", "alpha", "jitter", "info", "nll", "lenscale_sq", "amp", "noise", "mean")


def fit(X, Y, lenscale_sq, amp, noise, mean):
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    if Y.ndim == 1:
        Y = Y[:, None]
    K = ardse(X, None, lenscale_sq, amp)
    K[np.diag_indices(K.shape[0])] += noise
    L, jitter, info = chol_jitter(K)
    r = Y - mean
    t = solve_triangular(L, r, lower=True)
    alpha = solve_triangular(L, t, lower=True, trans="T")
    f = Fit()
    f.X, f.L, f.alpha, f.jitter, f.info = X, L, alpha, jitter, info
    f.lenscale_sq, f.amp, f.noise, f.mean = np.asarray(lenscale_sq, dtype=np.float64).ravel(), amp, noise, mean
    n = X.shape[0]
    f.nll = 0.5 * np.sum(r * alpha, axis=0) + np.sum(np.log(np.diag(L))) + 0.5 * n * np.log(2.0 * np.pi)
    return f


def predict(f, X1, var_with_noise=False, var_min=None, chunk=8192):
    """Posterior mean (M x c) and variance (M) at X1."""
    X1 = np.asarray(X1, dtype=np.float64)
    M = X1.shape[0]
    mu = np.empty((M, f.alpha.shape[1]))
    var = np.empty(M)
    for lo in range(0, M, chunk):
        hi = min(M, lo + chunk)
        Ks = ardse(X1[lo:hi], f.X, f.lenscale_sq, f.amp)
        mu[lo:hi] = f.mean + Ks @ f.alpha
        V = solve_triangular(f.L, Ks.T, lower=True, check_finite=False)
        var[lo:hi] = f.amp - np.einsum("ij,ij->j", V, V)
    if var_with_noise:
        var += f.noise
    if var_min is not None:
        var = np.where(var < var_min, var_min, var)  # TH clamp: NaN passes through
    return mu, var
