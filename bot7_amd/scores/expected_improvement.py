"""bot7.scores.expected_improvement (scores/expected_improvement.lua), computed by b7_score_ei.

EI(mu, var; fmin, xi) = max(0, (fmin - mu - xi) * Phi(z) + sigma * phi(z)), z = (fmin - mu - xi)/sigma, with the
reference's own A&S erf (utils/math.lua:261-288).  ``config.tradeoff`` (xi) defaults to 0.0 and
``config.nFantasies`` to 100 (:30-31).  EI.compute's 4th parameter is shadowed by a global in the reference
(:69-70); the intended meaning -- xi from the score's config -- is what is implemented."""
import numpy as np

from .abstract import abstract


class expected_improvement(abstract):
    title = "bot7.scores.expected_improvement"

    def __init__(self, config=None):
        config = dict(config or {})
        config["tradeoff"] = config.get("tradeoff") or 0.0      # scores/expected_improvement.lua:30
        config["nFantasies"] = config.get("nFantasies") or 100  # :31
        self.config = config

    def add_to(self, ctx, Y_obs, config=None):
        config = config or self.config
        fmin = np.asarray(Y_obs, dtype=np.float64).reshape(len(Y_obs), -1).min(axis=0)  # Y_obs:min(1), :64
        ctx.score_ei(fmin, config.get("tradeoff") or 0.0)

    def device_spec(self, Y_obs, config=None):
        """Keyword arguments of Context.eval_nominate for this score (b7_score_spec)."""
        config = config or self.config
        fmin = np.asarray(Y_obs, dtype=np.float64).reshape(len(Y_obs), -1).min(axis=0)  # Y_obs:min(1), :64
        return {"score": "ei", "fmin": fmin, "tradeoff": config.get("tradeoff") or 0.0}

    def eval(self, model, hyp, X_obs, Y_obs, X_hid, X_pend=None, config=None):
        """EI.eval (:43-67): posterior at X_hid, fmins, EI.compute.  Returns the M scores on the host."""
        config = config or self.config
        X_obs = np.atleast_2d(np.asarray(X_obs, dtype=np.float64))
        Y_obs = np.asarray(Y_obs, dtype=np.float64).reshape(X_obs.shape[0], -1)
        if X_pend is not None and np.size(X_pend) > 0:  # :51-60
            X_pend = np.atleast_2d(np.asarray(X_pend, dtype=np.float64))
            nF = int(config["nFantasies"])
            Y_pend = model.fantasize(nF, X_obs, Y_obs, X_pend, hyp)            # nPend x nFantasies
            X_obs = np.concatenate([X_obs, X_pend], axis=0)                     # X_obs:cat(X_pend, 1)
            Y_obs = np.concatenate([np.tile(Y_obs[:, :1], (1, nF)), Y_pend], axis=0)
        model.predict_device(X_obs, Y_obs, X_hid, hyp)  # :63
        ctx = model.ctx
        ctx.score_reset()
        self.add_to(ctx, Y_obs, config)
        _, _, scores = ctx.score_finish(1.0, download=True)
        return scores

    @staticmethod
    def compute(ctx, fval, fvar, fmin, tradeoff=0.0):
        """EI.compute (:69-88) on caller-provided mean/var."""
        return ctx.ei_compute(fval, fvar, fmin, tradeoff)
