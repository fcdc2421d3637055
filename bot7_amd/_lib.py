"""ctypes binding of libbot7hip.so (include/bot7hip.h) and a thin ``Context`` wrapper.

No fallbacks: if the shared library is missing or no gfx950 device is present, the first use raises
``Bot7HipError``.  Nothing in this package computes on the CPU.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BOT7HIP_LIB: the same override the LuaJIT binding honours (lua/bot7hip_ffi.lua); diagnostic builds use it (tools/)
_SO = os.environ.get("BOT7HIP_LIB") or os.path.join(_HERE, "libbot7hip.so")

B7_OK = 0
ERR_NAMES = {-1: "B7_ERR_INVALID", -2: "B7_ERR_HIP", -3: "B7_ERR_NOMEM", -4: "B7_ERR_STATE",
             -5: "B7_ERR_UNSUPPORTED", -6: "B7_ERR_RANGE", -7: "B7_ERR_COMM"}

# Every symbol include/bot7hip.h declares (tests check the library exports each of them).
SYMBOLS = [
    "b7_abi_version", "b7_create", "b7_destroy", "b7_last_error", "b7_device_info", "b7_sync", "b7_set_workspace",
    "b7_sobol_direction_numbers", "b7_grid_sobol", "b7_grid_random", "b7_grid_upload", "b7_grid_download", "b7_grid_shape", "b7_grid_remove", "b7_grid_remove_rows",
    "b7_grid_colrange", "b7_grid_apply_onesided", "b7_grid_random_torch", "b7_torch_rand",
    "b7_gp_default_opts", "b7_gp_set_opts", "b7_gp_fit", "b7_gp_set_data", "b7_gp_fit_hyp", "b7_gp_predict_hyp", "b7_gp_nll_batch", "b7_chol", "b7_gp_predict", "b7_gp_predict_at", "b7_gp_fantasize", "b7_gp_append", "b7_gp_download",
    "b7_blr_basis", "b7_blr_features", "b7_blr_fit", "b7_blr_fit_x", "b7_blr_predict", "b7_score_reset", "b7_score_ei", "b7_score_cb", "b7_score_finish",
    "b7_comm_pick_winner", "b7_comm_unique_id", "b7_comm_init", "b7_comm_info", "b7_comm_destroy", "b7_comm_allreduce_f64", "b7_score_finish_global", "b7_eval_nominate", "b7_blr_eval_nominate", "b7_blr_eval_nominate_marg",
    "b7_nominate_commit", "b7_shard_commit_rule", "b7_exchange_info",
    "b7_group_create", "b7_group_destroy", "b7_group_last_error", "b7_group_info", "b7_group_ctx", "b7_group_set_workspace", "b7_group_gp_set_opts",
    "b7_group_grid_sobol", "b7_group_grid_random", "b7_group_grid_onesided", "b7_group_grid_upload", "b7_group_grid_shape", "b7_group_grid_download",
    "b7_group_grid_remove_rows", "b7_group_gp_set_data", "b7_group_eval_nominate", "b7_group_nominate_commit",
    "b7_ei_compute", "b7_cb_compute", "b7_argmax",
    "b7_timer_start", "b7_timer_stop", "b7_timer_ms", "b7_profile_enable", "b7_profile_reset", "b7_profile_get", "b7_persist_fallbacks",
]


class Bot7HipError(RuntimeError):
    def __init__(self, code, message):
        self.code = code
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "B7_ERR"), code, message))


class Hyp(C.Structure):
    _fields_ = [("lenscale_sq", C.POINTER(C.c_double)), ("amp", C.c_double), ("noise", C.c_double),
                ("mean", C.c_double)]


class ScoreSpec(C.Structure):
    _fields_ = [("kind", C.c_int), ("tradeoff", C.c_double), ("upper", C.c_int), ("sign", C.c_double),
                ("fmin", C.POINTER(C.c_double))]


SCORE_EI, SCORE_CB = 1, 2


class Mlp(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("dims", C.POINTER(C.c_int)), ("W", C.POINTER(C.POINTER(C.c_double))),
                ("b", C.POINTER(C.POINTER(C.c_double))), ("activation", C.c_int)]


ACTIVATIONS = {None: 0, "Identity": 0, "Tanh": 1, "ReLU": 2, "Sigmoid": 3}
_MLP_CACHE = {}


class GpOpts(C.Structure):
    _fields_ = [("jitter_eps", C.c_double), ("jitter_growth", C.c_double), ("var_with_noise", C.c_int),
                ("var_clamp", C.c_int), ("var_min", C.c_double)]


_libs = {}
# the DIAGNOSTIC build (python -m bot7_amd.build --diag: the A/B switches, the fault injector and the RCCL override compiled in);
# test infrastructure, loaded BESIDE the shipped library by Context(..., lib="diag")
_DIAG_SO = os.path.join(os.path.dirname(_HERE), "tools", "_build", "libbot7hip_diag.so")


def lib_path(which=None):
    return _DIAG_SO if which == "diag" else _SO


def load(which=None):
    """Load libbot7hip.so (or, which="diag", the diagnostic build beside it) once.  torch (when installed) is imported first so
    that both share ONE HIP runtime: torch bundles its own libamdhip64 with the same SONAME, and whichever is mapped first
    serves both."""
    path = lib_path(which)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise Bot7HipError(-2, "%s not found: build it with `python -m bot7_amd.build%s` "
                               "(there is no CPU fallback)" % (path, " --diag" if which == "diag" else ""))
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path, mode=C.RTLD_GLOBAL if which != "diag" else C.RTLD_LOCAL)
    dp, vp, i64, i32, dbl = C.POINTER(C.c_double), C.c_void_p, C.c_int64, C.c_int, C.c_double
    sig = {
        "b7_abi_version": (i32, []),
        "b7_create": (i32, [C.POINTER(vp), i32]),
        "b7_destroy": (None, [vp]),
        "b7_last_error": (C.c_char_p, [vp]),
        "b7_device_info": (i32, [vp, C.c_char_p, C.POINTER(i32), C.POINTER(i64)]),
        "b7_sync": (i32, [vp]),
        "b7_set_workspace": (i32, [vp, i64]),
        "b7_sobol_direction_numbers": (i32, [i32, vp]),
        "b7_grid_sobol": (i32, [vp, i64, i32, i64, vp, vp, vp]),
        "b7_grid_random": (i32, [vp, i64, i32, C.c_uint64, i64, vp, vp, vp]),
        "b7_grid_upload": (i32, [vp, vp, i64, i32]),
        "b7_grid_download": (i32, [vp, i64, i64, vp]),
        "b7_grid_shape": (i32, [vp, C.POINTER(i64), C.POINTER(i32)]),
        "b7_grid_remove": (i32, [vp, i64, vp]),
        "b7_grid_remove_rows": (i32, [vp, vp, i64, vp]),
        "b7_gp_default_opts": (i32, [C.POINTER(GpOpts)]),
        "b7_gp_set_opts": (i32, [vp, C.POINTER(GpOpts)]),
        "b7_gp_fit": (i32, [vp, vp, vp, i32, i32, i32, C.POINTER(Hyp), vp, C.POINTER(dbl), C.POINTER(i32)]),
        "b7_gp_set_data": (i32, [vp, vp, vp, i32, i32, i32]),
        "b7_gp_fit_hyp": (i32, [vp, C.POINTER(Hyp), vp, C.POINTER(dbl), C.POINTER(i32)]),
        "b7_gp_predict_hyp": (i32, [vp, C.POINTER(Hyp), vp, vp, vp, C.POINTER(dbl), C.POINTER(i32)]),
        "b7_gp_nll_batch": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp]),
        "b7_chol": (i32, [vp, vp, i32, vp, C.POINTER(dbl), C.POINTER(i32)]),
        "b7_gp_predict": (i32, [vp, vp, vp]),
        "b7_gp_predict_at": (i32, [vp, vp, i64, vp, vp]),
        "b7_gp_fantasize": (i32, [vp, vp, i32, i32, C.c_uint64, vp, vp, vp]),
        "b7_gp_append": (i32, [vp, vp, vp]),
        "b7_gp_download": (i32, [vp, vp, vp, vp]),
        "b7_blr_basis": (i32, [vp, C.POINTER(Mlp), vp, i64, vp]),
        "b7_blr_features": (i32, [vp, vp, i64, i32]),
        "b7_blr_fit": (i32, [vp, vp, vp, i32, i32, dbl, dbl, dbl, C.POINTER(dbl)]),
        "b7_blr_fit_x": (i32, [vp, C.POINTER(Mlp), vp, vp, i32, dbl, dbl, dbl, C.POINTER(dbl)]),
        "b7_blr_predict": (i32, [vp, vp, vp]),
        "b7_score_reset": (i32, [vp]),
        "b7_score_ei": (i32, [vp, vp, dbl]),
        "b7_score_cb": (i32, [vp, dbl, i32, dbl]),
        "b7_score_finish": (i32, [vp, dbl, C.POINTER(dbl), C.POINTER(i64), vp]),
        "b7_comm_pick_winner": (i32, [vp, i32, C.POINTER(dbl), C.POINTER(i64)]),
        "b7_comm_unique_id": (i32, [vp]),
        "b7_comm_init": (i32, [vp, i32, i32, vp]),
        "b7_comm_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "b7_comm_destroy": (i32, [vp]),
        "b7_comm_allreduce_f64": (i32, [vp, vp, i32, i32]),
        "b7_score_finish_global": (i32, [vp, dbl, i64, C.POINTER(dbl), C.POINTER(i64)]),
        "b7_eval_nominate": (i32, [vp, i32, C.POINTER(Hyp), C.POINTER(ScoreSpec), i64, C.POINTER(dbl), C.POINTER(i64),
                                   vp, vp]),
        "b7_blr_eval_nominate": (i32, [vp, C.POINTER(Mlp), vp, vp, i32, dbl, dbl, dbl, C.POINTER(ScoreSpec), i64, C.POINTER(dbl),
                                       C.POINTER(i64), C.POINTER(dbl)]),
        "b7_blr_eval_nominate_marg": (i32, [vp, C.POINTER(Mlp), vp, vp, i32, i32, vp, vp, vp, C.POINTER(ScoreSpec), i64, C.POINTER(dbl),
                                            C.POINTER(i64), vp, C.POINTER(dbl)]),
        "b7_grid_random_torch": (i32, [vp, i64, i32, C.c_uint64, i32, vp, vp, vp]),
        "b7_torch_rand": (i32, [C.c_uint64, i64, i32, vp]),
        "b7_grid_colrange": (i32, [vp, vp, vp]),
        "b7_grid_apply_onesided": (i32, [vp, vp, vp, vp]),
        "b7_nominate_commit": (i32, [vp, i64, C.POINTER(i64), vp]),
        "b7_shard_commit_rule": (i32, [i64, i64, i64, C.POINTER(i64), C.POINTER(i64)]),
        "b7_exchange_info": (i32, [vp, C.POINTER(i32), vp, C.POINTER(i64), C.POINTER(i32), vp]),
        "b7_group_create": (i32, [C.POINTER(vp), i32, vp]),
        "b7_group_destroy": (None, [vp]),
        "b7_group_last_error": (C.c_char_p, [vp]),
        "b7_group_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "b7_group_ctx": (vp, [vp, i32]),
        "b7_group_set_workspace": (i32, [vp, i64]),
        "b7_group_gp_set_opts": (i32, [vp, C.POINTER(GpOpts)]),
        "b7_group_grid_sobol": (i32, [vp, i64, i32, i64, vp, vp]),
        "b7_group_grid_random": (i32, [vp, i64, i32, C.c_uint64, vp, vp]),
        "b7_group_grid_onesided": (i32, [vp, vp, vp]),
        "b7_group_grid_upload": (i32, [vp, vp, i64, i32]),
        "b7_group_grid_shape": (i32, [vp, C.POINTER(i64), C.POINTER(i32), vp]),
        "b7_group_grid_download": (i32, [vp, i64, i64, vp]),
        "b7_group_grid_remove_rows": (i32, [vp, vp, i64, vp]),
        "b7_group_gp_set_data": (i32, [vp, vp, vp, i32, i32, i32]),
        "b7_group_eval_nominate": (i32, [vp, i32, C.POINTER(Hyp), C.POINTER(ScoreSpec), C.POINTER(dbl), C.POINTER(i64), vp, vp]),
        "b7_group_nominate_commit": (i32, [vp, i64, vp]),
        "b7_ei_compute": (i32, [vp, vp, vp, vp, dbl, i64, i32, vp]),
        "b7_cb_compute": (i32, [vp, vp, vp, dbl, i32, dbl, i64, i32, vp]),
        "b7_argmax": (i32, [vp, vp, i64, C.POINTER(dbl), C.POINTER(i64)]),
        "b7_timer_start": (i32, [vp, i32]),
        "b7_timer_stop": (i32, [vp, i32]),
        "b7_timer_ms": (i32, [vp, i32, C.POINTER(C.c_float)]),
        "b7_profile_enable": (i32, [vp, i32]),
        "b7_profile_reset": (i32, [vp]),
        "b7_profile_get": (i32, [vp, C.c_char_p, C.POINTER(dbl), C.POINTER(i64)]),
        "b7_persist_fallbacks": (i32, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _libs[path] = L
    return L


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context(object):
    """One GPU's worth of state: the resident candidate grid, the current GP fit, the score accumulator."""

    def __init__(self, device_id=0, _borrowed=None, lib=None):
        self._L = load(lib)
        self._lib_name = lib
        if _borrowed is not None:   # a member of a Group: the handle belongs to the group
            self._h, self._owned = C.c_void_p(_borrowed), False
        else:
            h = C.c_void_p()
            rc = self._L.b7_create(C.byref(h), int(device_id))
            if rc != B7_OK:
                raise Bot7HipError(rc, (self._L.b7_last_error(None) or b"").decode())
            self._h, self._owned = h, True
        self.device_id = int(device_id)
        self.grid_version = 0  # bumped whenever the resident grid changes (DeviceGrid views compare against it)
        self.fit_token = 0     # bumped by every call that replaces the fit (models check it before gp_append)

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                self._L.b7_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != B7_OK:
            raise Bot7HipError(rc, (self._L.b7_last_error(self._h) or b"").decode())

    # ---- context
    def device_info(self):
        name = C.create_string_buffer(64)
        cus, mem = C.c_int(), C.c_int64()
        self._ck(self._L.b7_device_info(self._h, name, C.byref(cus), C.byref(mem)))
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": mem.value}

    def sync(self):
        self._ck(self._L.b7_sync(self._h))

    def set_workspace(self, nbytes):
        self._ck(self._L.b7_set_workspace(self._h, int(nbytes)))

    # ---- grids
    @staticmethod
    def _minmax(mins, maxes, dims):
        if mins is None and maxes is None:
            return None, None
        # one of the two alone selects the one-sided maps of grids/sobol.lua:82-85
        mn = None if mins is None else _f64(mins).ravel()
        mx = None if maxes is None else _f64(maxes).ravel()
        if (mn is not None and mn.size != dims) or (mx is not None and mx.size != dims):
            raise Bot7HipError(-1, "mins/maxes must have `dims` entries")
        return mn, mx

    def grid_sobol(self, size, dims, skip=1, mins=None, maxes=None, download=True):
        mn, mx = self._minmax(mins, maxes, dims)
        out = np.empty((size, dims), dtype=np.float64) if download else None
        self._ck(self._L.b7_grid_sobol(self._h, int(size), int(dims), int(skip), _ptr(mn), _ptr(mx), _ptr(out)))
        self.grid_version += 1
        return out

    def grid_random(self, size, dims, seed=0, row_offset=0, mins=None, maxes=None, download=True):
        mn, mx = self._minmax(mins, maxes, dims)
        out = np.empty((size, dims), dtype=np.float64) if download else None
        self._ck(self._L.b7_grid_random(self._h, int(size), int(dims), int(seed) & (2 ** 64 - 1), int(row_offset),
                                        _ptr(mn), _ptr(mx), _ptr(out)))
        self.grid_version += 1
        return out

    def grid_random_torch(self, size, dims, seed=0, resolution=32, mins=None, maxes=None, download=True):
        """grids/random.lua with torch.rand's own MT19937 stream (torch.manualSeed(seed))."""
        mn, mx = self._minmax(mins, maxes, dims)
        out = np.empty((size, dims), dtype=np.float64) if download else None
        self._ck(self._L.b7_grid_random_torch(self._h, int(size), int(dims), int(seed), int(resolution), _ptr(mn), _ptr(mx),
                                              _ptr(out)))
        self.grid_version += 1
        return out

    def grid_upload(self, X):
        X = _f64(X)
        if X.ndim != 2:
            raise Bot7HipError(-1, "grid must be 2-D")
        self._ck(self._L.b7_grid_upload(self._h, _ptr(X), X.shape[0], X.shape[1]))
        self.grid_version += 1

    def grid_shape(self):
        M, d = C.c_int64(), C.c_int()
        self._ck(self._L.b7_grid_shape(self._h, C.byref(M), C.byref(d)))
        return M.value, d.value

    def grid_download(self, row0=0, rows=None):
        M, d = self.grid_shape()
        rows = M - row0 if rows is None else rows
        out = np.empty((rows, d), dtype=np.float64)
        self._ck(self._L.b7_grid_download(self._h, int(row0), int(rows), _ptr(out)))
        return out

    def grid_remove(self, idx1):
        _, d = self.grid_shape()
        row = np.empty(d, dtype=np.float64)
        self._ck(self._L.b7_grid_remove(self._h, int(idx1), _ptr(row)))
        self.grid_version += 1
        return row

    def grid_remove_rows(self, idx1, want_rows=True):
        """Stable deletion of several rows (1-based indices against the grid before the call)."""
        idx = np.ascontiguousarray(np.asarray(idx1, dtype=np.int64).ravel())
        _, d = self.grid_shape()
        rows = np.empty((idx.size, d), dtype=np.float64) if want_rows else None
        self._ck(self._L.b7_grid_remove_rows(self._h, _ptr(idx), idx.size, _ptr(rows)))
        self.grid_version += 1
        return rows

    def grid_colrange(self):
        """grid:min(1), grid:max(1) of the resident grid."""
        _, d = self.grid_shape()
        lo, hi = np.empty(d), np.empty(d)
        self._ck(self._L.b7_grid_colrange(self._h, _ptr(lo), _ptr(hi)))
        return lo, hi

    def grid_apply_onesided(self, mins=None, maxes=None, col_ext=None):
        mn = None if mins is None else _f64(mins).ravel()
        mx = None if maxes is None else _f64(maxes).ravel()
        ext = _f64(col_ext).ravel()
        self._ck(self._L.b7_grid_apply_onesided(self._h, _ptr(mn), _ptr(mx), _ptr(ext)))
        self.grid_version += 1

    # ---- model
    def gp_set_opts(self, **kw):
        o = GpOpts()
        self._ck(self._L.b7_gp_default_opts(C.byref(o)))
        for k, v in kw.items():
            if not hasattr(o, k):
                raise Bot7HipError(-1, "unknown gp option %r" % k)
            setattr(o, k, v)
        self._ck(self._L.b7_gp_set_opts(self._h, C.byref(o)))

    def gp_fit(self, X_obs, Y_obs, lenscale_sq, amp, noise, mean, want_nll=False):
        X = _f64(X_obs)
        if X.ndim == 1:
            X = X.reshape(1, -1)
        N, d = X.shape
        Y = _f64(Y_obs).reshape(N, -1)
        ls = _f64(lenscale_sq).ravel()
        if ls.size != d:
            raise Bot7HipError(-1, "lenscale_sq must have d entries")
        hyp = Hyp(ls.ctypes.data_as(C.POINTER(C.c_double)), float(amp), float(noise), float(mean))
        nll = np.empty(Y.shape[1], dtype=np.float64) if want_nll else None
        jit, info = C.c_double(), C.c_int()
        self._ck(self._L.b7_gp_fit(self._h, _ptr(X), _ptr(Y), N, d, Y.shape[1], C.byref(hyp), _ptr(nll),
                                   C.byref(jit), C.byref(info)))
        self.ycols = Y.shape[1]
        self._data_d = d
        self.fit_token += 1
        return {"nll": nll, "jitter": jit.value, "info": info.value}

    def gp_set_data(self, X_obs, Y_obs):
        """Put (X_obs, Y_obs) on the device once; gp_fit_hyp then refits them under new hypers (the sampler's loop)."""
        X = _f64(X_obs)
        if X.ndim == 1:
            X = X.reshape(1, -1)
        N, d = X.shape
        Y = _f64(Y_obs).reshape(N, -1)
        self._ck(self._L.b7_gp_set_data(self._h, _ptr(X), _ptr(Y), N, d, Y.shape[1]))
        self.ycols = Y.shape[1]
        self._data_d = d
        self.fit_token += 1

    def gp_fit_hyp(self, lenscale_sq, amp, noise, mean, want_nll=False):
        ls = _f64(lenscale_sq).ravel()
        if ls.size != getattr(self, "_data_d", -1):
            raise Bot7HipError(-1, "lenscale_sq must have d entries (call gp_set_data first)")
        hyp = Hyp(ls.ctypes.data_as(C.POINTER(C.c_double)), float(amp), float(noise), float(mean))
        nll = np.empty(self.ycols, dtype=np.float64) if want_nll else None
        jit, info = C.c_double(), C.c_int()
        self._ck(self._L.b7_gp_fit_hyp(self._h, C.byref(hyp), _ptr(nll), C.byref(jit), C.byref(info)))
        self.fit_token += 1
        return {"nll": nll, "jitter": jit.value, "info": info.value}

    def gp_predict_hyp(self, lenscale_sq, amp, noise, mean, download=False, want_nll=False):
        """Fit the resident data under the hypers AND predict over the resident grid in one call (no host round trip
        between the two).  Returns the fit report (+ mean, var when download)."""
        ls = _f64(lenscale_sq).ravel()
        if ls.size != getattr(self, "_data_d", -1):
            raise Bot7HipError(-1, "lenscale_sq must have d entries (call gp_set_data first)")
        hyp = Hyp(ls.ctypes.data_as(C.POINTER(C.c_double)), float(amp), float(noise), float(mean))
        nll = np.empty(self.ycols, dtype=np.float64) if want_nll else None
        jit, info = C.c_double(), C.c_int()
        mu = var = None
        if download:
            M, _ = self.grid_shape()
            mu, var = np.empty((M, self.ycols), dtype=np.float64), np.empty(M, dtype=np.float64)
        self._ck(self._L.b7_gp_predict_hyp(self._h, C.byref(hyp), _ptr(mu), _ptr(var), _ptr(nll), C.byref(jit),
                                           C.byref(info)))
        self.fit_token += 1
        out = {"nll": nll, "jitter": jit.value, "info": info.value}
        if download:
            out["mean"], out["var"] = mu, var
        return out

    def gp_nll_batch(self, lenscale_sq, amp, noise, mean, want_info=False):
        """Negative log marginal likelihoods of the resident data (gp_set_data) under B hyper vectors at once."""
        ls = _f64(lenscale_sq)
        ls = ls.reshape(1, -1) if ls.ndim == 1 else ls
        B = ls.shape[0]
        if ls.shape[1] != getattr(self, "_data_d", -1):
            raise Bot7HipError(-1, "lenscale_sq must be B x d (call gp_set_data first)")
        a, nz, m = (np.ascontiguousarray(np.broadcast_to(_f64(v).ravel(), (B,))) for v in (amp, noise, mean))
        nll = np.empty(B, dtype=np.float64)
        jit = np.empty(B, dtype=np.float64)
        info = np.empty(B, dtype=np.int32)
        before = self._L.b7_persist_fallbacks(self._h)
        self._ck(self._L.b7_gp_nll_batch(self._h, B, _ptr(ls), _ptr(a), _ptr(nz), _ptr(m), _ptr(nll), _ptr(jit), _ptr(info)))
        if self._L.b7_persist_fallbacks(self._h) != before:
            self.fit_token += 1      # a timed-out hand-off was redone in the context's fit slot: whoever cached "fit is current" must refit
        return (nll, jit, info) if want_info else nll

    def gp_nll1(self, lenscale_sq, amp, noise, mean):
        """One likelihood evaluation (B = 1) through preallocated argument buffers: what a slice-sampler step costs, without the
        per-call array conversions of gp_nll_batch (the C call itself is ~20-35 us; those conversions were another ~15).
        Returns (nll, jitter, info)."""
        d = getattr(self, "_data_d", -1)
        buf = getattr(self, "_nll1_buf", None)
        if buf is None or buf[0].size != d:
            arr = [np.empty(d), np.empty(1), np.empty(1), np.empty(1), np.empty(1), np.empty(1), np.empty(1, dtype=np.int32)]
            buf = self._nll1_buf = arr + [[_ptr(a) for a in arr]]
        buf[0][:] = lenscale_sq
        buf[1][0], buf[2][0], buf[3][0] = amp, noise, mean
        p = buf[7]
        before = self._L.b7_persist_fallbacks(self._h)
        self._ck(self._L.b7_gp_nll_batch(self._h, 1, p[0], p[1], p[2], p[3], p[4], p[5], p[6]))
        if self._L.b7_persist_fallbacks(self._h) != before:
            self.fit_token += 1
        return float(buf[4][0]), float(buf[5][0]), int(buf[6][0])

    def chol(self, src):
        """utils.math.chol(src, 'L') with the jitter schedule; returns (L, jitter_used, info_first)."""
        A = _f64(src)
        n = A.shape[0]
        res = np.empty((n, n), dtype=np.float64)
        jit, info = C.c_double(), C.c_int()
        self._ck(self._L.b7_chol(self._h, _ptr(A), n, _ptr(res), C.byref(jit), C.byref(info)))
        self.fit_token += 1
        return res, jit.value, info.value

    def gp_predict(self, download=True):
        M, _ = self.grid_shape()
        if not download:
            self._ck(self._L.b7_gp_predict(self._h, None, None))
            return None, None
        mean = np.empty((M, getattr(self, "ycols", 1)), dtype=np.float64)
        var = np.empty(M, dtype=np.float64)
        self._ck(self._L.b7_gp_predict(self._h, _ptr(mean), _ptr(var)))
        return mean, var

    def gp_predict_at(self, X1):
        X1 = _f64(X1)
        if X1.ndim == 1:
            X1 = X1.reshape(1, -1)
        mean = np.empty((X1.shape[0], getattr(self, "ycols", 1)), dtype=np.float64)
        var = np.empty(X1.shape[0], dtype=np.float64)
        self._ck(self._L.b7_gp_predict_at(self._h, _ptr(X1), X1.shape[0], _ptr(mean), _ptr(var)))
        return mean, var

    def gp_fantasize(self, X_pend, nFantasies, seed=0, want_moments=False):
        """nFantasies joint posterior draws at the pending points: P x nFantasies (+ mu_P, Sigma_P)."""
        Xp = _f64(X_pend)
        if Xp.ndim == 1:
            Xp = Xp.reshape(1, -1)
        P = Xp.shape[0]
        out = np.empty((P, int(nFantasies)), dtype=np.float64)
        mu = np.empty(P, dtype=np.float64) if want_moments else None
        cov = np.empty((P, P), dtype=np.float64) if want_moments else None
        self._ck(self._L.b7_gp_fantasize(self._h, _ptr(Xp), P, int(nFantasies), int(seed) & (2 ** 64 - 1), _ptr(out),
                                         _ptr(mu), _ptr(cov)))
        return (out, mu, cov) if want_moments else out

    def gp_append(self, x_new, y_new):
        """Extend the current fit by one observation under the same hypers (O(N^2))."""
        x = _f64(x_new).ravel()
        y = _f64(y_new).ravel()
        self._ck(self._L.b7_gp_append(self._h, _ptr(x), _ptr(y)))

    def gp_download(self, N, ycols=1):
        Lh = np.empty((N, N), dtype=np.float64)
        al = np.empty((N, ycols), dtype=np.float64)
        Li = np.empty((N, N), dtype=np.float64)
        self._ck(self._L.b7_gp_download(self._h, _ptr(Lh), _ptr(al), _ptr(Li)))
        return Lh, al, Li

    # ---- DNGO: basis network + Bayesian linear head
    @staticmethod
    def _mlp(weights, biases, activation):
        Ws = [_f64(w) for w in weights]
        bs = [_f64(b).ravel() for b in biases]
        # the same arrays as last time (a trial loop hands the same network to every call until it retrains it): the ctypes
        # description is reused -- it only holds pointers, the library reads the weights themselves on every call
        key = (activation,) + tuple((w.__array_interface__["data"][0], w.shape) for w in Ws) + tuple(
            (b.__array_interface__["data"][0], b.shape) for b in bs)
        hit = _MLP_CACHE.get(key)
        if hit is not None:
            return hit
        n = len(Ws)
        dims = (C.c_int * (n + 1))(*([Ws[0].shape[1]] + [w.shape[0] for w in Ws]))
        Wp = (C.POINTER(C.c_double) * n)(*[w.ctypes.data_as(C.POINTER(C.c_double)) for w in Ws])
        bp = (C.POINTER(C.c_double) * n)(*[b.ctypes.data_as(C.POINTER(C.c_double)) for b in bs])
        m = Mlp(n, dims, Wp, bp, ACTIVATIONS[activation])
        m._keep = (Ws, bs, dims, Wp, bp)      # the arrays stay alive as long as the description does
        if len(_MLP_CACHE) >= 8:
            _MLP_CACHE.clear()
        _MLP_CACHE[key] = m
        return m

    def blr_basis(self, weights, biases, activation="Tanh", X=None, download=False):
        """Features of X (host rows) or, with X=None, of the resident grid (kept on the device)."""
        net = self._mlp(weights, biases, activation)
        z = net._keep[0][-1].shape[0]
        if X is None:
            M, _ = self.grid_shape()
            out = np.empty((M, z), dtype=np.float64) if download else None
            self._ck(self._L.b7_blr_basis(self._h, C.byref(net), None, 0, _ptr(out)))
            return out
        X = _f64(X)
        if X.ndim == 1:
            X = X.reshape(1, -1)
        out = np.empty((X.shape[0], z), dtype=np.float64)
        self._ck(self._L.b7_blr_basis(self._h, C.byref(net), _ptr(X), X.shape[0], _ptr(out)))
        return out

    def blr_features(self, Z1):
        Z1 = _f64(Z1)
        self._ck(self._L.b7_blr_features(self._h, _ptr(Z1), Z1.shape[0], Z1.shape[1]))
        self.grid_version += 1

    def blr_fit(self, Z0, Y0, alpha_prec, beta, mean=0.0, want_nll=False):
        Z0 = _f64(Z0)
        Y0 = _f64(Y0).ravel()
        nll = C.c_double()
        self._ck(self._L.b7_blr_fit(self._h, _ptr(Z0), _ptr(Y0), Z0.shape[0], Z0.shape[1], float(alpha_prec),
                                    float(beta), float(mean), C.byref(nll) if want_nll else None))
        self.ycols = 1
        self.fit_token += 1
        return nll.value if want_nll else None

    def blr_fit_x(self, weights, biases, activation, X0, Y0, alpha_prec, beta, mean=0.0, want_nll=False):
        """Basis features of X0 and the Bayesian-linear fit in one device-side call."""
        net = self._mlp(weights, biases, activation)
        X0 = _f64(X0)
        Y0 = _f64(Y0).ravel()
        nll = C.c_double()
        self._ck(self._L.b7_blr_fit_x(self._h, C.byref(net), _ptr(X0), _ptr(Y0), X0.shape[0], float(alpha_prec),
                                      float(beta), float(mean), C.byref(nll) if want_nll else None))
        self.ycols = 1
        self.fit_token += 1
        return nll.value if want_nll else None

    def blr_eval_nominate(self, weights, biases, activation, X0, Y0, alpha_prec, beta, mean=0.0, score="ei", fmin=None,
                          tradeoff=None, upper=False, sign=-1.0, global_row_offset=0, want_jitter=False):
        """bayesopt:eval's DNGO branch + nominate in one call -> (value, 1-based global index[, jitter flag])."""
        net = self._mlp(weights, biases, activation)
        X0 = _f64(X0)
        Y0 = _f64(Y0).ravel()
        spec, fm = self._pack_spec(score, fmin, tradeoff, upper, sign)
        v, i, j = C.c_double(), C.c_int64(), C.c_double()
        self._ck(self._L.b7_blr_eval_nominate(self._h, C.byref(net), _ptr(X0), _ptr(Y0), X0.shape[0], float(alpha_prec),
                                              float(beta), float(mean), C.byref(spec), int(global_row_offset), C.byref(v),
                                              C.byref(i), C.byref(j)))
        self.fit_token += 1
        return (v.value, i.value, j.value) if want_jitter else (v.value, i.value)

    def blr_eval_nominate_marg(self, weights, biases, activation, X0, Y0, alphas, betas, means, score="ei", fmin=None,
                               tradeoff=None, upper=False, sign=-1.0, global_row_offset=0, want_nll=False):
        """bayesopt:eval's DNGO branch with the head's hypers marginalised (models/dngo.lua:109,174): S samples (alpha, beta, mean),
        S heads over the same features, score:add per sample, score:div(S), score:max(1).  Returns (value, 1-based global index,
        jitter flag[, nll per sample])."""
        net = self._mlp(weights, biases, activation)
        X = _f64(X0)
        Y = _f64(Y0).ravel()
        a, b, m = (np.ascontiguousarray(_f64(v).ravel()) for v in (alphas, betas, means))
        S = a.size
        if b.size != S or m.size != S:
            raise Bot7HipError(-1, "alphas, betas and means must have the same length")
        spec, fm = self._pack_spec(score, fmin, tradeoff, upper, sign)
        nll = np.empty(S, dtype=np.float64) if want_nll else None
        v, i, jit = C.c_double(), C.c_int64(), C.c_double()
        self._ck(self._L.b7_blr_eval_nominate_marg(self._h, C.byref(net), _ptr(X), _ptr(Y), X.shape[0], S, _ptr(a), _ptr(b), _ptr(m),
                                                   C.byref(spec), int(global_row_offset), C.byref(v), C.byref(i), _ptr(nll),
                                                   C.byref(jit)))
        self.fit_token += 1
        return (v.value, i.value, jit.value, nll) if want_nll else (v.value, i.value, jit.value)

    def blr_predict(self, download=True):
        M, _ = self.grid_shape()
        if not download:
            self._ck(self._L.b7_blr_predict(self._h, None, None))
            return None, None
        mean = np.empty((M, 1), dtype=np.float64)
        var = np.empty(M, dtype=np.float64)
        self._ck(self._L.b7_blr_predict(self._h, _ptr(mean), _ptr(var)))
        return mean, var

    # ---- scores
    def score_reset(self):
        self._ck(self._L.b7_score_reset(self._h))

    def score_ei(self, fmin, tradeoff=0.0):
        f = _f64(fmin).ravel()
        self._ck(self._L.b7_score_ei(self._h, _ptr(f), float(tradeoff)))

    def score_cb(self, tradeoff=1.0, upper=False, sign=-1.0):
        self._ck(self._L.b7_score_cb(self._h, float(tradeoff), int(bool(upper)), float(sign)))

    def score_finish(self, divisor=1.0, download=False):
        v, i = C.c_double(), C.c_int64()
        out = None
        if download:
            M, _ = self.grid_shape()
            out = np.empty(M, dtype=np.float64)
        self._ck(self._L.b7_score_finish(self._h, float(divisor), C.byref(v), C.byref(i), _ptr(out)))
        return v.value, i.value, out

    # ---- multi-GPU: the arg-max exchange over RCCL (one process per GPU, one context each)
    def comm_init(self, rank, world, id_bytes):
        """ncclCommInitRank on this context's GPU; id_bytes = comm_unique_id() of rank 0 (collective call)."""
        buf = C.create_string_buffer(bytes(id_bytes), COMM_ID_BYTES)
        self._ck(self._L.b7_comm_init(self._h, int(rank), int(world), C.cast(buf, C.c_void_p)))

    def comm_info(self):
        r, w = C.c_int(), C.c_int()
        self._ck(self._L.b7_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def comm_destroy(self):
        self._ck(self._L.b7_comm_destroy(self._h))

    def comm_allreduce(self, values, op="sum"):
        """A few host doubles reduced over the communicator (control plane; doubles as a barrier)."""
        v = np.ascontiguousarray(np.asarray(values, dtype=np.float64).ravel()).copy()
        self._ck(self._L.b7_comm_allreduce_f64(self._h, _ptr(v), v.size, {"sum": 0, "max": 1, "min": 2}[op]))
        return v

    def score_finish_global(self, divisor=1.0, global_row_offset=0):
        """score:div + score:max(1) over the candidates of ALL ranks -> (value, 1-based global index)."""
        v, i = C.c_double(), C.c_int64()
        self._ck(self._L.b7_score_finish_global(self._h, float(divisor), int(global_row_offset), C.byref(v),
                                                C.byref(i)))
        return v.value, i.value

    def nominate_commit(self, idx1_global, global_row_offset=0):
        """bots/abstract.lua:118 on a sharded candidate set -> (nominee's row, this rank's new row offset)."""
        _, d = self.grid_shape()
        row, off = np.empty(d, dtype=np.float64), C.c_int64(int(global_row_offset))
        self._ck(self._L.b7_nominate_commit(self._h, int(idx1_global), C.byref(off), _ptr(row)))
        self.grid_version += 1
        return row, off.value

    def exchange_info(self):
        """The last exchange as this rank saw it: rows per rank, winner index / rank / row."""
        w, wi, wr = C.c_int(), C.c_int64(), C.c_int()
        rows, row = np.zeros(64, dtype=np.int64), np.zeros(96, dtype=np.float64)
        self._ck(self._L.b7_exchange_info(self._h, C.byref(w), _ptr(rows), C.byref(wi), C.byref(wr), _ptr(row)))
        _, d = self.grid_shape()
        return {"world": w.value, "rows": rows[:w.value].copy(), "winner_idx1": wi.value, "winner_rank": wr.value,
                "winner_row": row[:d].copy()}

    @staticmethod
    def _pack_hyps(hyps, d):
        S = len(hyps)
        arr = (Hyp * S)()
        keep = []
        for s, h in enumerate(hyps):
            if isinstance(h, dict):
                h = (h["lenscale_sq"], h["amp"], h["noise"], h["mean"])
            ls = _f64(h[0]).ravel()
            if ls.size != d:
                raise Bot7HipError(-1, "lenscale_sq must have d entries (call gp_set_data first)")
            keep.append(ls)
            arr[s] = Hyp(ls.ctypes.data_as(C.POINTER(C.c_double)), float(h[1]), float(h[2]), float(h[3]))
        return arr, keep

    @staticmethod
    def _pack_spec(score, fmin, tradeoff, upper, sign):
        if score == "ei":
            if fmin is None:
                raise Bot7HipError(-1, "EI needs fmin")
            fm = _f64(fmin).ravel()
            return ScoreSpec(SCORE_EI, 0.0 if tradeoff is None else float(tradeoff), 0, 0.0,
                             fm.ctypes.data_as(C.POINTER(C.c_double))), fm
        if score == "cb":
            return ScoreSpec(SCORE_CB, 1.0 if tradeoff is None else float(tradeoff), int(bool(upper)), float(sign),
                             None), None
        raise Bot7HipError(-1, "score must be 'ei' or 'cb'")

    def eval_nominate(self, hyps, score="ei", fmin=None, tradeoff=None, upper=False, sign=-1.0,
                      global_row_offset=0, want_report=False):
        """bayesopt:eval + nominate in one call: hyps is a sequence of (lenscale_sq, amp, noise, mean) or dicts
        with those keys; score "ei" (needs fmin) or "cb".  Returns (value, 1-based global index[, report])."""
        S = len(hyps)
        arr, keep = self._pack_hyps(hyps, getattr(self, "_data_d", -1))
        spec, fm = self._pack_spec(score, fmin, tradeoff, upper, sign)
        jit = np.zeros(S, dtype=np.float64) if want_report else None
        info = np.zeros(S, dtype=np.int32) if want_report else None
        v, i = C.c_double(), C.c_int64()
        self._ck(self._L.b7_eval_nominate(self._h, S, arr, C.byref(spec), int(global_row_offset), C.byref(v),
                                          C.byref(i), _ptr(jit), _ptr(info)))
        self.fit_token += 1
        if want_report:
            return v.value, i.value, {"jitter": jit, "info": info}
        return v.value, i.value

    def ei_compute(self, mean, var, fmin, tradeoff=0.0):
        mean = _f64(mean)
        M = mean.shape[0]
        c = 1 if mean.ndim == 1 else mean.shape[1]
        var, fmin = _f64(var).ravel(), _f64(fmin).ravel()
        out = np.empty(M, dtype=np.float64)
        self._ck(self._L.b7_ei_compute(self._h, _ptr(mean), _ptr(var), _ptr(fmin), float(tradeoff), M, c, _ptr(out)))
        return out

    def cb_compute(self, mean, var, tradeoff=1.0, upper=False, sign=-1.0):
        mean = _f64(mean)
        M = mean.shape[0]
        c = 1 if mean.ndim == 1 else mean.shape[1]
        var = _f64(var).ravel()
        out = np.empty(M, dtype=np.float64)
        self._ck(self._L.b7_cb_compute(self._h, _ptr(mean), _ptr(var), float(tradeoff), int(bool(upper)), float(sign),
                                       M, c, _ptr(out)))
        return out

    def argmax(self, scores):
        s = _f64(scores).ravel()
        v, i = C.c_double(), C.c_int64()
        self._ck(self._L.b7_argmax(self._h, _ptr(s), s.shape[0], C.byref(v), C.byref(i)))
        return v.value, i.value

    # ---- measurement
    def timer_start(self, slot=0):
        self._ck(self._L.b7_timer_start(self._h, slot))

    def timer_stop(self, slot=0):
        self._ck(self._L.b7_timer_stop(self._h, slot))

    def timer_ms(self, slot=0):
        ms = C.c_float()
        self._ck(self._L.b7_timer_ms(self._h, slot, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        self._ck(self._L.b7_profile_enable(self._h, int(bool(on))))

    def profile_reset(self):
        self._ck(self._L.b7_profile_reset(self._h))

    def profile_get(self, phase):
        ms, n = C.c_double(), C.c_int64()
        self._ck(self._L.b7_profile_get(self._h, phase.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


class Group(object):
    """One host process, several GPUs (b7_group_*): a candidate grid sharded over the members, the observations on every
    member, bayesopt:eval + nominate over the union as one call.  device_ids may repeat (virtual ranks on one device)."""

    def __init__(self, device_ids, lib=None):
        self._L = load(lib)
        ids = np.ascontiguousarray(np.asarray(device_ids, dtype=np.int32).ravel())
        h = C.c_void_p()
        rc = self._L.b7_group_create(C.byref(h), ids.size, _ptr(ids))
        if rc != B7_OK:
            raise Bot7HipError(rc, (self._L.b7_last_error(None) or b"").decode())
        self._h, self.n = h, int(ids.size)
        self.members = [Context(int(ids[r]), _borrowed=self._L.b7_group_ctx(h, r), lib=lib) for r in range(self.n)]
        self._data_d = -1
        self.grid_version = 0

    def close(self):
        if getattr(self, "_h", None):
            for m in self.members:
                m.close()
            self._L.b7_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != B7_OK:
            raise Bot7HipError(rc, (self._L.b7_group_last_error(self._h) or b"").decode())

    def info(self):
        n, r = C.c_int(), C.c_int()
        self._ck(self._L.b7_group_info(self._h, C.byref(n), C.byref(r)))
        return {"n": n.value, "uses_rccl": bool(r.value)}

    def set_workspace(self, nbytes):
        self._ck(self._L.b7_group_set_workspace(self._h, int(nbytes)))

    def grid_sobol(self, size, dims, skip=1, mins=None, maxes=None):
        mn, mx = Context._minmax(mins, maxes, dims)
        self._ck(self._L.b7_group_grid_sobol(self._h, int(size), int(dims), int(skip), _ptr(mn), _ptr(mx)))
        self.grid_version += 1

    def grid_random(self, size, dims, seed=0, mins=None, maxes=None):
        mn, mx = Context._minmax(mins, maxes, dims)
        self._ck(self._L.b7_group_grid_random(self._h, int(size), int(dims), int(seed) & (2 ** 64 - 1), _ptr(mn), _ptr(mx)))
        self.grid_version += 1

    def grid_upload(self, X):
        X = _f64(X)
        self._ck(self._L.b7_group_grid_upload(self._h, _ptr(X), X.shape[0], X.shape[1]))
        self.grid_version += 1

    def grid_shape(self):
        M, d, off = C.c_int64(), C.c_int(), np.zeros(self.n + 1, dtype=np.int64)
        self._ck(self._L.b7_group_grid_shape(self._h, C.byref(M), C.byref(d), _ptr(off)))
        return M.value, d.value, off

    def grid_download(self, row0=0, rows=None):
        M, d, _ = self.grid_shape()
        rows = M - row0 if rows is None else rows
        out = np.empty((rows, d), dtype=np.float64)
        self._ck(self._L.b7_group_grid_download(self._h, int(row0), int(rows), _ptr(out)))
        return out

    def grid_remove_rows(self, idx1, want_rows=True):
        idx = np.ascontiguousarray(np.asarray(idx1, dtype=np.int64).ravel())
        _, d, _ = self.grid_shape()
        rows = np.empty((idx.size, d), dtype=np.float64) if want_rows else None
        self._ck(self._L.b7_group_grid_remove_rows(self._h, _ptr(idx), idx.size, _ptr(rows)))
        self.grid_version += 1
        return rows

    def gp_set_data(self, X_obs, Y_obs):
        X = _f64(X_obs)
        Y = _f64(Y_obs)
        if Y.ndim == 1:
            Y = Y.reshape(-1, 1)
        self._ck(self._L.b7_group_gp_set_data(self._h, _ptr(X), _ptr(Y), X.shape[0], X.shape[1], Y.shape[1]))
        self._data_d = X.shape[1]
        for m in self.members:
            m._data_d = X.shape[1]
            m.ycols = Y.shape[1]
            m.fit_token += 1

    def eval_nominate(self, hyps, score="ei", fmin=None, tradeoff=None, upper=False, sign=-1.0, want_report=False):
        S = len(hyps)
        arr, keep = Context._pack_hyps(hyps, self._data_d)
        spec, fm = Context._pack_spec(score, fmin, tradeoff, upper, sign)
        jit = np.zeros(S, dtype=np.float64) if want_report else None
        info = np.zeros(S, dtype=np.int32) if want_report else None
        v, i = C.c_double(), C.c_int64()
        self._ck(self._L.b7_group_eval_nominate(self._h, S, arr, C.byref(spec), C.byref(v), C.byref(i), _ptr(jit), _ptr(info)))
        if want_report:
            return v.value, i.value, {"jitter": jit, "info": info}
        return v.value, i.value

    def nominate_commit(self, idx1_global):
        _, d, _ = self.grid_shape()
        row = np.empty(d, dtype=np.float64)
        self._ck(self._L.b7_group_nominate_commit(self._h, int(idx1_global), _ptr(row)))
        self.grid_version += 1
        return row


def torch_rand(seed, n, resolution=32):
    """torch.manualSeed(seed); torch.rand(n) -- MT19937, host-only."""
    out = np.empty(int(n), dtype=np.float64)
    rc = load().b7_torch_rand(int(seed), int(n), int(resolution), _ptr(out))
    if rc != B7_OK:
        raise Bot7HipError(rc, "resolution must be 32 or 53")
    return out


def shard_commit_rule(idx1_global, offset, M_local):
    """The library's bookkeeping rule for a stable deletion on the union of contiguous shards -> (local_idx1, new_offset).
    Host-only."""
    loc, off = C.c_int64(), C.c_int64()
    rc = load().b7_shard_commit_rule(int(idx1_global), int(offset), int(M_local), C.byref(loc), C.byref(off))
    if rc != B7_OK:
        raise Bot7HipError(rc, "bad index / offset")
    return loc.value, off.value


COMM_ID_BYTES = 128


def comm_unique_id():
    """ncclGetUniqueId through the C ABI (rank 0 calls it and distributes the 128 bytes)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load().b7_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc != B7_OK:
        raise Bot7HipError(rc, (load().b7_last_error(None) or b"").decode())
    return buf.raw


def comm_pick_winner(pairs):
    """The library's winner rule on a list of (value, global_idx1) pairs (idx1 <= 0: empty shard).  Host-only."""
    tab = np.zeros((len(pairs), 2), dtype=np.uint64)
    for r, (v, i) in enumerate(pairs):
        if int(i) > 0:
            tab[r, 0] = np.array([v], dtype=np.float64).view(np.uint64)[0]
            tab[r, 1] = int(i)
    v, i = C.c_double(), C.c_int64()
    rc = load().b7_comm_pick_winner(_ptr(tab), len(pairs), C.byref(v), C.byref(i))
    if rc != B7_OK:
        raise Bot7HipError(rc, "every shard is empty")
    return v.value, i.value


def sobol_direction_numbers(dims):
    """Host-only view of the Sobol direction-number table the kernel uses (no GPU needed)."""
    out = np.empty((dims, 30), dtype=np.uint32)
    rc = load().b7_sobol_direction_numbers(int(dims), _ptr(out))
    if rc != B7_OK:
        raise Bot7HipError(rc, "dims must be in [1, 39]")
    return out


_default = {}


def default_context(device_id=None):
    """Process-wide context for a device (LOCAL_RANK when launched one process per GPU)."""
    if device_id is None:
        device_id = int(os.environ.get("LOCAL_RANK", "0"))
    if device_id not in _default:
        _default[device_id] = Context(device_id)
    return _default[device_id]
