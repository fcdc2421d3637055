"""HIP-backed stand-in for gp.models.gp_regressor with the ardse kernel, GaussianNoise_iso noise model and
constant mean -- the defaults bots/bayesopt.lua:39-45 selects.

The reference's class lives in the un-vendored `gp` package (models/init.lua:15); what is mirrored here is the
protocol its callers use:
    bots/abstract.lua:147-149               model:init(observed, responses)
    bots/bayesopt.lua:68,74                 model:sample_hypers(X_obs, Y_obs[, nil, nil, true]) -> flat vector
    bots/bayesopt.lua:75                    model:parse_hypers(vector) -> hyp table
    scores/expected_improvement.lua:63      model:predict(X_obs, Y_obs, X_hid, hyp, {mean=true, var=true})
    scores/expected_improvement.lua:57      model:fantasize(nFantasies, X_obs, Y_obs, X_pend, hyp)
    bots/bayesopt.lua:65                    model:class()
Everything numerical happens in libbot7hip.so (b7_gp_fit / b7_gp_predict); nothing is computed here.

Hyper-parameter vector layout (ours; the reference's parse_hypers layout is unknowable from its tree):
    [ lenscale_sq_1 .. lenscale_sq_d, amp, noise, mean ]

Hyper sampling (config.sampler = 'slice', bots/bayesopt.lua:44): bot7.samplers.slice walks the log posterior
    log p(theta | X, Y) = -NLL(theta) + log prior(theta),   theta = [log lenscale_sq, log amp, log noise, mean]
with a flat prior inside explicit bounds (config.bounds; the gp package's priors are unknown).  Every evaluation
is one device fit (K + Cholesky + log-det + quadratic form, the `GP-fit ms` unit of work).  With
``config.nBurnin = 0`` and ``config.sample = False`` (default) sample_hypers returns the point estimate, which
is what the fixed-hyper parity tests and the benchmark use."""
import numpy as np

from .abstract import abstract
from .._lib import default_context
from ..grids.abstract import DeviceGrid


class gp_regressor(abstract):
    title = "bot7.models.gp_regressor"

    def __init__(self, config=None, context=None):
        self.config = dict(config or {})
        self.kernel = self.config.get("kernel", "ardse")              # bots/bayesopt.lua:41
        self.nzModel = self.config.get("nzModel", "GaussianNoise_iso")  # :42
        self.mean = self.config.get("mean", "constant")               # :43
        if (self.kernel, self.nzModel, self.mean) != ("ardse", "GaussianNoise_iso", "constant"):
            raise NotImplementedError("only ardse + GaussianNoise_iso + constant mean are built")
        self._ctx = context
        self.hyp = None
        self.last_fit = None

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    # ---- hyper-parameters -------------------------------------------------------------------------
    def init(self, X_obs, Y_obs):
        """model:init (bots/abstract.lua:147-149).  Point initialisation: lenscale_sq = d/8, amp = var(Y),
        mean = mean(Y), noise = 1e-4*amp (0 when config.noiseless; the jitter schedule then takes over)."""
        X = np.atleast_2d(np.asarray(X_obs, dtype=np.float64))
        Y = np.asarray(Y_obs, dtype=np.float64).reshape(X.shape[0], -1)
        d = X.shape[1]
        amp = float(np.var(Y)) if Y.shape[0] > 1 else 1.0
        if not amp > 0.0:
            amp = 1.0
        noise = 0.0 if self.config.get("noiseless") else 1e-4 * amp
        self.hyp = self.parse_hypers(np.concatenate([np.full(d, d / 8.0), [amp, noise, float(np.mean(Y))]]))
        return self.hyp

    # -- theta <-> hyp
    @staticmethod
    def _to_theta(h):
        return np.concatenate([np.log(h["lenscale_sq"]), [np.log(h["amp"]), np.log(max(h["noise"], 1e-300)), h["mean"]]])

    @staticmethod
    def _from_theta(t):
        d = t.size - 3
        return {"lenscale_sq": np.exp(t[:d]), "amp": float(np.exp(t[d])), "noise": float(np.exp(t[d + 1])),
                "mean": float(t[d + 2])}

    def _bounds(self, X, Y):
        # (cached per data set: the sampler asks for the bounds at every density evaluation of a trial)
        ck = (X.__array_interface__["data"][0], X.shape, Y.__array_interface__["data"][0], Y.shape)
        hit = getattr(self, "_bounds_cache", None)
        if hit is not None and hit[0] == ck and hit[1] == float(Y[-1, 0]):
            return hit[2]
        out = self._bounds_compute(X, Y)
        self._bounds_cache = (ck, float(Y[-1, 0]), out)
        return out

    def _bounds_compute(self, X, Y):
        b = self.config.get("bounds") or {}
        d = X.shape[1]
        vy = float(np.var(Y)) or 1.0
        lo = np.concatenate([np.full(d, np.log(b.get("lenscale_sq_min", 1e-3 * d))), [np.log(b.get("amp_min", 1e-3 * vy)),
                             np.log(b.get("noise_min", 1e-8 * vy)), b.get("mean_min", float(Y.min()) - 3 * np.sqrt(vy))]])
        hi = np.concatenate([np.full(d, np.log(b.get("lenscale_sq_max", 1e3 * d))), [np.log(b.get("amp_max", 1e3 * vy)),
                             np.log(b.get("noise_max", 1e0 * vy)), b.get("mean_max", float(Y.max()) + 3 * np.sqrt(vy))]])
        return lo, hi

    def log_posterior(self, theta, X_obs, Y_obs):
        """-NLL on the device + flat prior inside the bounds (-inf outside)."""
        theta = np.asarray(theta, dtype=np.float64).ravel()
        lo, hi = self._bounds(X_obs, Y_obs)
        if (theta < lo).any() or (theta > hi).any() or not np.isfinite(theta).all():
            return -np.inf
        self.nEvals = getattr(self, "nEvals", 0) + 1
        return -float(self.nll(X_obs, Y_obs, self._from_theta(theta))[0])

    def _make_resident(self, X, Y):
        """The observations on the device (b7_gp_set_data), once per data set: a cheap look first (addresses, shapes, the last
        entries, the context's fit token), the content hash only when that differs."""
        fast = (X.__array_interface__["data"][0], X.shape, Y.__array_interface__["data"][0], float(X[-1, -1]), float(Y[-1, 0]),
                self.ctx.fit_token)
        if getattr(self, "_resident_fast", None) != fast:
            key = self._data_key(X, Y)
            if getattr(self, "_resident_key", None) != (key, self.ctx.fit_token):
                self.ctx.gp_set_data(X, Y)
                self._resident_key = (key, self.ctx.fit_token)
            self._resident_fast = fast[:-1] + (self.ctx.fit_token,)

    def _density(self, X, Y):
        """log_posterior as the sampler's density for ONE update (X, Y: the float64 arrays sample_hypers holds): the same
        statements, with what cannot change between two evaluations of an update -- the bounds, the residency of the data --
        done once.  At the reference's default sizes a density evaluation is a 20-30 us library call; the per-call
        bookkeeping of log_posterior + nll was another 12."""
        if Y.shape[1] != 1:
            return lambda t, _args: self.log_posterior(t, X, Y)
        lo, hi = self._bounds(X, Y)
        d = X.shape[1]
        self._make_resident(X, Y)
        ctx = self.ctx
        one = getattr(ctx, "gp_nll1", None)

        data_key = self._resident_key

        def f(t, _args):
            t = np.asarray(t, dtype=np.float64).ravel()
            if not ((t >= lo) & (t <= hi)).all():      # also NaN: it fails both comparisons
                return -np.inf
            # the point a slice update starts from is the point the previous update ended on, whose density was the last thing
            # evaluated (samplers/slice.lua:106 after :134-164): the same vector under the same data is not sent to the device
            # again -- a pure function of both, so the value is the one a second evaluation would return, bit for bit
            tb = t.tobytes()
            memo = getattr(self, "_density_memo", None)
            if memo is not None and memo[0] is data_key and memo[1] == tb:
                return memo[2]
            self.nEvals = getattr(self, "nEvals", 0) + 1
            ls, amp, noise, mean = np.exp(t[:d]), float(np.exp(t[d])), float(np.exp(t[d + 1])), float(t[d + 2])
            if one is not None:
                v, jit, info = one(ls, amp, noise, mean)
            else:
                nll, jit, info = ctx.gp_nll_batch(ls, amp, noise, mean, want_info=True)
                v, jit, info = float(nll[0]), float(jit[0]), int(info[0])
            self.last_fit = {"nll": v, "jitter": jit, "info": info}
            self._density_memo = (data_key, tb, -v)
            return -v
        return f

    def _samplers(self):
        from .abstract import sampler_registry
        name = self.config.get("sampler", "slice")
        if name not in sampler_registry:
            raise KeyError("no sampler %r registered in bot7_amd.models.abstract.sampler_registry (the reference's "
                           "bot7.samplers is host code outside this package)" % name)
        return sampler_registry

    def sample_hypers(self, X_obs, Y_obs, _a=None, _b=None, state=None):
        """model:sample_hypers(X, Y[, nil, nil, true]) (bots/bayesopt.lua:68,74) -> flat hyper vector.

        config.sample = True: one slice-sampling update of the chain kept in ``self.theta`` per call with
        ``state`` set (the driver's per-sample call), ``config.nBurnin`` updates on a call without it (the
        driver's first call, :68).  config.sample = False (default): the current point estimate."""
        X = np.atleast_2d(np.asarray(X_obs, dtype=np.float64))
        Y = np.asarray(Y_obs, dtype=np.float64).reshape(X.shape[0], -1)
        if self.hyp is None:
            self.init(X, Y)
        if self.config.get("sample") and int(self.config.get("chains", 1)) > 1:
            return self._sample_hypers_chains(X, Y, state)
        if self.config.get("sample"):
            Samplers = self._samplers()
            if getattr(self, "_sampler", None) is None:
                self._sampler = Samplers[self.config.get("sampler", "slice")]()
                self._sopt = self._sampler.configure(dict(self.config.get("sampler_opt") or {},
                                                          seed=self.config.get("seed", 0)))
                self._sopt.setdefault("width", 0.5)
            if self.config.get("noiseless") and self.hyp["noise"] <= 0.0:
                self.hyp["noise"] = np.exp(self._bounds(X, Y)[0][-2])
            # the chain's state is theta itself: as long as nobody replaced self.hyp, the next update starts from the very vector
            # the last one returned (not from log(exp(theta)), which differs in the last bits) -- and the density there is known
            kept = getattr(self, "_chain_state", None)
            theta = kept[0] if kept is not None and kept[1] is self.hyp else self._to_theta(self.hyp)
            n_updates = 1 if state else int(self.config.get("nBurnin", 0))
            f = self._density(X, Y)
            for _ in range(n_updates):
                theta = self._sampler.sample(f, theta.reshape(1, -1), dict(self._sopt, nSamples=1), None)[0]
            self.hyp = self._from_theta(theta)
            self._chain_state = (theta, self.hyp)
        h = self.hyp
        return np.concatenate([h["lenscale_sq"], [h["amp"], h["noise"], h["mean"]]])

    # ---- several chains in lock step: one b7_gp_nll_batch per round of density evaluations -------------------------------
    def _sample_hypers_chains(self, X, Y, state):
        """config.chains = C > 1: C independent slice-sampler chains (each the reference's sampler, statement for
        statement, with its own random stream) advance TOGETHER: whenever every live chain has asked for a density
        value, all the requests go to the device as one b7_gp_nll_batch.  The driver's per-sample calls
        (bots/bayesopt.lua:74, state = true) are served one chain after the other, and a new lock-step update of all
        chains runs whenever the C samples of the last one are used up; the call without `state` (:68) runs
        config.nBurnin such updates.  A statistical variant of the reference's single chain (C chains of length k instead
        of one of length C k), offered because at the sizes Bayesian optimisation lives in a batch of 16 likelihoods costs
        what one does."""
        C = int(self.config["chains"])
        if getattr(self, "_chain_thetas", None) is None:
            Samplers = self._samplers()
            self._sampler = Samplers[self.config.get("sampler", "slice")]()
            self._sopt = self._sampler.configure(dict(self.config.get("sampler_opt") or {}))
            self._sopt.setdefault("width", 0.5)
            if self.config.get("noiseless") and self.hyp["noise"] <= 0.0:
                self.hyp["noise"] = np.exp(self._bounds(X, Y)[0][-2])
            t0 = self._to_theta(self.hyp)
            seed = int(self.config.get("seed", 0))
            self._chain_rngs = [np.random.default_rng([seed, c]) for c in range(C)]
            # chain 0 starts at the point estimate, the others a little off it (inside the bounds)
            lo, hi = self._bounds(X, Y)
            self._chain_thetas = [np.clip(t0 + (0.1 * self._chain_rngs[c].standard_normal(t0.size) if c else 0.0), lo, hi)
                                  for c in range(C)]
            self._chain_pool = []
        if not state:
            for _ in range(int(self.config.get("nBurnin", 0))):
                self._lockstep_update(X, Y)
            self._chain_pool = []
        else:
            if not self._chain_pool:
                self._lockstep_update(X, Y)
                self._chain_pool = [t.copy() for t in self._chain_thetas]
            self.hyp = self._from_theta(self._chain_pool.pop(0))
        h = self.hyp
        return np.concatenate([h["lenscale_sq"], [h["amp"], h["noise"], h["mean"]]])

    def _lockstep_update(self, X, Y):
        """One slice-sampler update of every chain, the density requests of all chains evaluated batch by batch."""
        import threading
        C, d = len(self._chain_thetas), X.shape[1]
        lo, hi = self._bounds(X, Y)
        key = self._data_key(X, Y)
        if getattr(self, "_resident_key", None) != (key, self.ctx.fit_token):
            self.ctx.gp_set_data(X, Y)
            self._resident_key = (key, self.ctx.fit_token)
        cond = threading.Condition()
        pending, results, alive, out, errors = {}, {}, [C], [None] * C, []

        def density(c):
            def f(t, _args):
                t = np.asarray(t, dtype=np.float64).ravel()
                if (t < lo).any() or (t > hi).any() or not np.isfinite(t).all():
                    return -np.inf                      # flat prior inside the bounds: no device call
                with cond:
                    pending[c] = t
                    cond.notify_all()
                    while c not in results:
                        cond.wait()
                    return results.pop(c)
            return f

        def run(c):
            try:
                opt = dict(self._sopt, nSamples=1, rng=self._chain_rngs[c])
                out[c] = self._sampler.sample(density(c), self._chain_thetas[c].reshape(1, -1), opt, None)[0]
            except Exception as e:                      # surfaced by the caller
                errors.append(e)
            finally:
                with cond:
                    alive[0] -= 1
                    cond.notify_all()

        threads = [threading.Thread(target=run, args=(c,)) for c in range(C)]
        for th in threads:
            th.start()
        with cond:
            while alive[0] > 0:
                while alive[0] > 0 and len(pending) < alive[0]:
                    cond.wait()                         # until every live chain is waiting for a value
                if not pending:
                    break
                ids = sorted(pending)
                T = np.stack([pending[c] for c in ids])
                pending.clear()
                nll = self.ctx.gp_nll_batch(np.exp(T[:, :d]), np.exp(T[:, d]), np.exp(T[:, d + 1]), T[:, d + 2])
                self.nEvals = getattr(self, "nEvals", 0) + len(ids)
                self.nBatches = getattr(self, "nBatches", 0) + 1
                for c, v in zip(ids, nll):
                    results[c] = -float(v)
                cond.notify_all()
        for th in threads:
            th.join()
        if errors:
            raise errors[0]
        self._chain_thetas = [np.asarray(o, dtype=np.float64) for o in out]

    @staticmethod
    def parse_hypers(vec):
        v = np.asarray(vec, dtype=np.float64).ravel()
        d = v.size - 3
        return {"lenscale_sq": v[:d].copy(), "amp": float(v[d]), "noise": float(v[d + 1]), "mean": float(v[d + 2])}

    def nll(self, X_obs, Y_obs, hyp=None):
        """Negative log marginal likelihood of (X_obs, Y_obs) under hyp: K + Cholesky + log-det + quadratic form
        on the device -- the unit of work behind the `GP-fit ms` metric."""
        hyp = hyp or self.hyp
        X = np.atleast_2d(np.asarray(X_obs, dtype=np.float64))
        Y = np.asarray(Y_obs, dtype=np.float64).reshape(X.shape[0], -1)
        if Y.shape[1] == 1:
            # the likelihood alone, of data that stay on the device across the sampler's evaluations: no inverse, no alpha,
            # and for N <= 128 one workgroup of one launch (b7_gp_nll_batch)
            # (the content hash is taken once per array pair, not per evaluation: the sampler calls this ~70 times a trial with the
            # same two arrays)
            self._make_resident(X, Y)
            if hasattr(self.ctx, "gp_nll1"):
                v, jit, info = self.ctx.gp_nll1(hyp["lenscale_sq"], hyp["amp"], hyp["noise"], hyp["mean"])
                self.last_fit = {"nll": np.array([v]), "jitter": jit, "info": info}
                return self.last_fit["nll"]
            nll, jit, info = self.ctx.gp_nll_batch(hyp["lenscale_sq"], hyp["amp"], hyp["noise"], hyp["mean"], want_info=True)
            self.last_fit = {"nll": nll, "jitter": float(jit[0]), "info": int(info[0])}
            return nll
        out = self.ctx.gp_fit(X, Y, hyp["lenscale_sq"], hyp["amp"], hyp["noise"], hyp["mean"], want_nll=True)
        self.last_fit = out
        return out["nll"]

    # ---- posterior ---------------------------------------------------------------------------------
    def fit(self, X_obs, Y_obs, hyp=None):
        """Fit on the device.  When the call is "the previous fit plus one observation, same hypers" (what
        bots/abstract.lua:137-149 produces trial after trial under a point estimate), the factor is extended in
        O(N^2) by b7_gp_append instead of being rebuilt (config.incremental, default True)."""
        hyp = hyp or self.hyp
        if hyp is None:
            hyp = self.init(X_obs, Y_obs)
        X = np.atleast_2d(np.asarray(X_obs, dtype=np.float64))
        Y = np.asarray(Y_obs, dtype=np.float64).reshape(X.shape[0], -1)
        key = (tuple(np.asarray(hyp["lenscale_sq"], dtype=np.float64).ravel()), hyp["amp"], hyp["noise"], hyp["mean"])
        prev = getattr(self, "_prev", None)
        # only a clean factor is extended: after a jittered fit (or the chol(I) fallback) the reference refactors
        # K + eps*I in full every trial (utils/math.lua:159-218), and so do we
        if (self.config.get("incremental", True) and prev is not None and prev.get("clean")
                and prev["token"] == self.ctx.fit_token
                and prev["key"] == key and X.shape[0] == prev["X"].shape[0] + 1 and Y.shape[1] == prev["Y"].shape[1]
                and np.array_equal(X[:-1], prev["X"]) and np.array_equal(Y[:-1], prev["Y"])):
            from .._lib import Bot7HipError
            try:
                self.ctx.gp_append(X[-1], Y[-1])
                self._prev = {"token": self.ctx.fit_token, "key": key, "X": X.copy(), "Y": Y.copy(), "clean": True}
                self.last_fit = {"nll": None, "jitter": 0.0, "info": 0, "incremental": True}
                return self.last_fit
            except Bot7HipError as e:
                if e.code != -4:   # B7_ERR_STATE: factor full or not positive definite -> rebuild below
                    raise
        self.last_fit = self.ctx.gp_fit(X, Y, hyp["lenscale_sq"], hyp["amp"], hyp["noise"], hyp["mean"])
        self._prev = {"token": self.ctx.fit_token, "key": key, "X": X.copy(), "Y": Y.copy(),
                      "clean": self.last_fit["jitter"] == 0.0 and self.last_fit["info"] == 0}
        return self.last_fit

    def _is_resident(self, X1):
        return isinstance(X1, DeviceGrid) and X1.ctx is self.ctx and X1.shape[0] == self.ctx.grid_shape()[0] \
            and getattr(X1, "version", -1) == self.ctx.grid_version

    @staticmethod
    def _data_key(X, Y):
        """What identifies the resident observations: their CONTENT (shapes and sums collide: Y and -Y with zero sum, two
        responses swapped between rows).  N x (d + c) doubles through blake2b is nothing next to a fit."""
        import hashlib
        h = hashlib.blake2b(digest_size=16)
        h.update(np.ascontiguousarray(X).tobytes())
        h.update(np.ascontiguousarray(Y).tobytes())
        return (X.shape, Y.shape, h.digest())

    def stage(self, X_obs, Y_obs, X_hid):
        """Make (X_obs, Y_obs) the resident data and X_hid the resident grid of the context, uploading only what
        is not there already (for b7_eval_nominate, which refits the resident data under S hyper samples)."""
        X = np.atleast_2d(np.asarray(X_obs, dtype=np.float64))
        Y = np.asarray(Y_obs, dtype=np.float64).reshape(X.shape[0], -1)
        key = self._data_key(X, Y)
        if getattr(self, "_resident_key", None) != (key, self.ctx.fit_token):
            self.ctx.gp_set_data(X, Y)
            self._resident_key = (key, self.ctx.fit_token)
        if not self._is_resident(X_hid):
            self.ctx.grid_upload(np.atleast_2d(np.asarray(X_hid, dtype=np.float64)))

    def predict_device(self, X_obs, Y_obs, X_hid, hyp=None):
        """fit + predict leaving mean/var on the device (for the fused score path).  Uploads X_hid only when it
        is not the grid already resident on this context."""
        self.fit(X_obs, Y_obs, hyp)
        if not self._is_resident(X_hid):
            self.ctx.grid_upload(np.atleast_2d(np.asarray(X_hid, dtype=np.float64)))
        self.ctx.gp_predict(download=False)

    def predict(self, X_obs, Y_obs, X_hid, hyp=None, req=None):
        """model:predict(X0, Y0, X1, hyp, {mean=, var=}) -> {'mean': M x 1, 'var': M}."""
        self.fit(X_obs, Y_obs, hyp)
        if self._is_resident(X_hid):
            mean, var = self.ctx.gp_predict(download=True)
        else:
            mean, var = self.ctx.gp_predict_at(np.atleast_2d(np.asarray(X_hid, dtype=np.float64)))
        req = req or {"mean": True, "var": True}
        out = {}
        if req.get("mean"):
            out["mean"] = mean
        if req.get("var"):
            out["var"] = var
        return out

    def fantasize(self, nFantasies, X_obs, Y_obs, X_pend, hyp=None):
        """model:fantasize (scores/expected_improvement.lua:57): nPend x nFantasies joint posterior draws at the
        pending points (b7_gp_fantasize; counter-based normals, config.seed + a call counter)."""
        self.fit(X_obs, Y_obs, hyp)
        self._fcalls = getattr(self, "_fcalls", 0) + 1
        seed = int(self.config.get("seed", 0)) * 1000003 + self._fcalls
        return self.ctx.gp_fantasize(X_pend, nFantasies, seed)
