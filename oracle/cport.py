"""ctypes binding of oracle/b7_oracle.c (test infrastructure; see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libb7oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "b7_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


_lib = None
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_bitwise_xor.restype = C.c_double
        L.orc_bitwise_xor.argtypes = [C.c_double, C.c_double]
        L.orc_i4_bit_hi1.argtypes = [C.c_double]
        L.orc_i4_bit_lo0.argtypes = [C.c_double]
        L.orc_sobol_generate.argtypes = [C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, _dp]
        L.orc_sobol_bank.argtypes = [C.c_int, _dp]
        L.orc_affine.argtypes = [_dp, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        for f in (L.orc_erf, L.orc_norm_cdf, L.orc_norm_pdf):
            f.restype = C.c_double
            f.argtypes = [C.c_double]
        L.orc_ei.argtypes = [_dp, _dp, _dp, C.c_double, C.c_int64, C.c_int, _dp]
        L.orc_cb.argtypes = [_dp, _dp, C.c_double, C.c_int, C.c_double, C.c_int64, C.c_int, _dp]
        L.orc_accumulate.argtypes = [_dp, _dp, C.c_int64]
        L.orc_divide.argtypes = [_dp, C.c_double, C.c_int64]
        L.orc_argmax_first.restype = C.c_int64
        L.orc_argmax_first.argtypes = [_dp, C.c_int64, C.POINTER(C.c_double)]
        L.orc_remove_row.argtypes = [_dp, C.c_int64, C.c_int, C.c_int64, _dp]
        L.orc_pdist.argtypes = [_dp, C.c_int64, C.c_void_p, C.c_int64, C.c_int, _dp, _dp]
        L.orc_potrf_lower.argtypes = [_dp, C.c_int]
        L.orc_chol_jitter.argtypes = [_dp, C.c_int, _dp, C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def sobol(size, dims, skip=1, mins=None, maxes=None):
    """grids/sobol.lua:58-90 via the stateful i4_sobol recurrence."""
    out = np.empty((size, dims), dtype=np.float64)
    mn = mx = None
    if mins is not None and maxes is not None:
        mn, mx = _f64(mins).ravel(), _f64(maxes).ravel()
    rc = lib().orc_sobol_generate(size, dims, skip, None if mn is None else mn.ctypes.data,
                                  None if mx is None else mx.ctypes.data, out)
    if rc != 0:
        raise ValueError("orc_sobol_generate failed rc=%d" % rc)
    if (mins is None) != (maxes is None):        # grids/sobol.lua:82-85: one of the two alone
        out = affine(out, mins, maxes)
    return out


def sobol_bank(dims):
    out = np.empty((dims, 30), dtype=np.float64)
    if lib().orc_sobol_bank(dims, out) != 0:
        raise ValueError("dims out of range")
    return out


def affine(grid, mins=None, maxes=None):
    """grids/sobol.lua:79-85 / grids/random.lua:27-33 on a copy of grid: both, mins only, maxes only, or neither."""
    g = _f64(grid).copy()
    mn = None if mins is None else _f64(mins).ravel()
    mx = None if maxes is None else _f64(maxes).ravel()
    lib().orc_affine(g, g.shape[0], g.shape[1], None if mn is None else mn.ctypes.data, None if mx is None else mx.ctypes.data)
    return g


def erf(x):
    f = lib().orc_erf
    return np.array([f(float(v)) for v in np.ravel(x)]).reshape(np.shape(x))


def norm_cdf(x):
    f = lib().orc_norm_cdf
    return np.array([f(float(v)) for v in np.ravel(x)]).reshape(np.shape(x))


def norm_pdf(x):
    f = lib().orc_norm_pdf
    return np.array([f(float(v)) for v in np.ravel(x)]).reshape(np.shape(x))


def ei(mean, var, fmin, tradeoff=0.0):
    mean = _f64(mean)
    M = mean.shape[0]
    c = 1 if mean.ndim == 1 else mean.shape[1]
    out = np.empty(M)
    lib().orc_ei(mean.reshape(M, c), _f64(var).ravel(), _f64(fmin).ravel(), float(tradeoff), M, c, out)
    return out


def cb(mean, var, tradeoff=1.0, upper=False, sign=-1.0):
    mean = _f64(mean)
    M = mean.shape[0]
    c = 1 if mean.ndim == 1 else mean.shape[1]
    out = np.empty(M)
    lib().orc_cb(mean.reshape(M, c), _f64(var).ravel(), float(tradeoff), int(bool(upper)), float(sign), M, c, out)
    return out


def accumulate(acc, score):
    lib().orc_accumulate(acc, _f64(score), acc.shape[0])
    return acc


def divide(acc, divisor):
    lib().orc_divide(acc, float(divisor), acc.shape[0])
    return acc


def argmax_first(scores):
    s = _f64(scores).ravel()
    v = C.c_double()
    idx = lib().orc_argmax_first(s, s.shape[0], C.byref(v))
    return idx, v.value


def remove_row(X, idx1):
    X = _f64(X)
    out = np.empty((X.shape[0] - 1, X.shape[1]))
    lib().orc_remove_row(X, X.shape[0], X.shape[1], idx1, out)
    return out


def pdist(X, Z, lenscale):
    X = _f64(X)
    N = X.shape[0] if Z is None else np.shape(Z)[0]
    out = np.empty((X.shape[0], N))
    Zc = None if Z is None else _f64(Z)
    lib().orc_pdist(X, X.shape[0], None if Zc is None else Zc.ctypes.data, N, X.shape[1], _f64(lenscale).ravel(), out)
    return out


def potrf_lower(A):
    A = _f64(A).copy()
    info = lib().orc_potrf_lower(A, A.shape[0])
    return A, info


def chol_jitter(K):
    K = _f64(K)
    res = np.empty_like(K)
    j = C.c_double()
    itr = lib().orc_chol_jitter(K, K.shape[0], res, C.byref(j))
    return res, j.value, itr
