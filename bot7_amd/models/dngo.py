"""bot7.models.dngo (models/dngo.lua): neural-network basis + Bayesian linear regressor, on the GPU.

What is mirrored: ``predict(X0, Y0, X1, hyp, req)`` (:108-175) -- features of X0 and X1 by a forward pass through
the network up to the basis layer (:155-171), then the Bayesian-linear predictive mean and variance (:174) -- and
``class()`` == 'bot7.models.dngo', which makes bots.bayesopt.eval skip the hyper-marginalisation loop
(bots/bayesopt.lua:65-66).  What is NOT: building and (re)training the network with nnTools (:49-106, :126-152;
out of scope): the trained layers come in through ``config['network']`` =
{'weights': [...], 'biases': [...], 'activation': 'Tanh'|'ReLU'|'Sigmoid'|None}.

The head (gp.models.bayes_linear, absent `gp` package: PARITY UNPINNED) is standard Bayesian linear regression with prior
precision ``config['alpha']`` and noise precision ``config['beta']``.  ``predict`` / ``predict_device`` take ONE point
``hyp = {'alpha', 'beta', 'mean'}``.  The reference's default hyp='marginalize' (:109, passed to the predictor at :174) is realised
as the analogue of bayesopt:eval's GP loop by ``eval_nominate`` below: a LIST of hyp tables (drawn by the host, e.g. with
bot7.samplers.slice over ``nll``) -> S heads over the same features, the acquisition of every head added in sample order,
score:div(S), score:max(1) -- one library call, b7_blr_eval_nominate_marg."""
import numpy as np

from .abstract import abstract
from .._lib import default_context
from ..grids.abstract import DeviceGrid


class dngo(abstract):
    title = "bot7.models.dngo"

    def __init__(self, config=None, context=None):
        self.config = dict(config or {})
        self._ctx = context
        self.hyp = None
        net = self.config.get("network")
        if net is None:
            raise ValueError("dngo: config['network'] = {'weights', 'biases', 'activation'} is required "
                             "(training the network with nnTools is out of scope)")
        self.weights = [np.asarray(w, dtype=np.float64) for w in net["weights"]]
        self.biases = [np.asarray(b, dtype=np.float64) for b in net["biases"]]
        self.activation = net.get("activation", "Tanh")
        self.config["zDim"] = self.weights[-1].shape[0]      # config.zDim, models/dngo.lua:105

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    def init(self, X_obs, Y_obs):
        """model:init (bots/abstract.lua:147-149): nothing to train here; set the head's point hypers."""
        Y = np.asarray(Y_obs, dtype=np.float64)
        vy = float(np.var(Y)) or 1.0
        self.hyp = {"alpha": float(self.config.get("alpha", 1.0)),
                    "beta": float(self.config.get("beta", 1.0 / (1e-2 * vy))), "mean": float(np.mean(Y))}
        return self.hyp

    def sample_hypers(self, X_obs, Y_obs, *_):
        if self.hyp is None:
            self.init(X_obs, Y_obs)
        return np.array([self.hyp["alpha"], self.hyp["beta"], self.hyp["mean"]])

    @staticmethod
    def parse_hypers(v):
        return {"alpha": float(v[0]), "beta": float(v[1]), "mean": float(v[2])}

    def basis(self, X):
        """Z = basis(X) for host rows (models/dngo.lua:155-162 for X0)."""
        return self.ctx.blr_basis(self.weights, self.biases, self.activation, X=X)

    def fit(self, X_obs, Y_obs, hyp=None, want_nll=False):
        hyp = hyp or self.hyp or self.init(X_obs, Y_obs)
        return self.ctx.blr_fit_x(self.weights, self.biases, self.activation,
                                  np.atleast_2d(np.asarray(X_obs, dtype=np.float64)), Y_obs, hyp["alpha"], hyp["beta"],
                                  hyp["mean"], want_nll=want_nll)

    def _is_resident(self, X1):
        return isinstance(X1, DeviceGrid) and X1.ctx is self.ctx and X1.shape[0] == self.ctx.grid_shape()[0] \
            and getattr(X1, "version", -1) == self.ctx.grid_version

    def predict_device(self, X_obs, Y_obs, X_hid, hyp=None):
        self.fit(X_obs, Y_obs, hyp)
        if not self._is_resident(X_hid):
            self.ctx.grid_upload(np.atleast_2d(np.asarray(X_hid, dtype=np.float64)))
        self.ctx.blr_basis(self.weights, self.biases, self.activation)   # :164-171 for X1, on the device
        self.ctx.blr_predict(download=False)

    def eval_nominate(self, X_obs, Y_obs, X_hid, hyps, score="ei", fmin=None, tradeoff=None, upper=False, sign=-1.0, want_nll=False):
        """The marginalised nomination (models/dngo.lua:109 'marginalize'): hyps = [{'alpha', 'beta', 'mean'}, ...]; returns
        (value, 1-based index[, nll per sample]).  One hyp table = b7_blr_eval_nominate's single head."""
        if isinstance(hyps, dict):
            hyps = [hyps]
        if not self._is_resident(X_hid):
            self.ctx.grid_upload(np.atleast_2d(np.asarray(X_hid, dtype=np.float64)))
        out = self.ctx.blr_eval_nominate_marg(self.weights, self.biases, self.activation, np.atleast_2d(np.asarray(X_obs, dtype=np.float64)),
                                              Y_obs, [h["alpha"] for h in hyps], [h["beta"] for h in hyps], [h["mean"] for h in hyps],
                                              score=score, fmin=fmin, tradeoff=tradeoff, upper=upper, sign=sign, want_nll=want_nll)
        return (out[0], out[1], out[3]) if want_nll else (out[0], out[1])

    def nll(self, X_obs, Y_obs, hyp=None):
        """Negative log evidence of the head under hyp (what a host-side sampler of (alpha, beta) walks)."""
        return self.fit(X_obs, Y_obs, hyp, want_nll=True)["nll"]

    def predict(self, X_obs, Y_obs, X_hid, hyp=None, req=None):
        """dngo:predict (:108-175) -> {'mean': M x 1, 'var': M}."""
        self.fit(X_obs, Y_obs, hyp)
        if not self._is_resident(X_hid):
            self.ctx.grid_upload(np.atleast_2d(np.asarray(X_hid, dtype=np.float64)))
        self.ctx.blr_basis(self.weights, self.biases, self.activation)
        mean, var = self.ctx.blr_predict(download=True)
        req = req or {"mean": True, "var": True}
        return {k: v for k, v in (("mean", mean), ("var", var)) if req.get(k)}
