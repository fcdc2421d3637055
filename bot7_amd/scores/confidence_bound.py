"""bot7.scores.confidence_bound (scores/confidence_bound.lua), computed by b7_score_cb.

Defaults (:29-36): tradeoff 1.0, nFantasies 100, bound 'lower', sign -1.0 -- i.e. the score is the negative LCB,
-(mu - sqrt(var)).  The reference's pending/fantasies block shadows its own locals (:56) and has no effect; it is
ignored here as well."""
from .abstract import abstract


def _d(v, default):
    """Lua's `v or default`: only nil/false fall through (0 is truthy in Lua)."""
    return default if v is None or v is False else v


class confidence_bound(abstract):
    title = "bot7.scores.confidence_bound"

    def __init__(self, config=None):
        config = dict(config or {})
        config["tradeoff"] = _d(config.get("tradeoff"), 1.0)      # scores/confidence_bound.lua:31
        config["nFantasies"] = _d(config.get("nFantasies"), 100)  # :32
        config["bound"] = _d(config.get("bound"), "lower")      # :33
        config["sign"] = _d(config.get("sign"), -1.0)      # :34
        self.config = config

    def add_to(self, ctx, Y_obs=None, config=None):
        config = config or self.config
        bound = str(_d(config.get("bound"), "lower")).lower()      # :72
        if bound not in ("lower", "upper"):
            raise ValueError("bound must be 'lower' or 'upper'")
        ctx.score_cb(_d(config.get("tradeoff"), 1.0), bound == "upper", _d(config.get("sign"), -1.0))

    def device_spec(self, Y_obs=None, config=None):
        """Keyword arguments of Context.eval_nominate for this score (b7_score_spec)."""
        config = config or self.config
        bound = str(_d(config.get("bound"), "lower")).lower()      # :72
        if bound not in ("lower", "upper"):
            raise ValueError("bound must be 'lower' or 'upper'")
        return {"score": "cb", "tradeoff": _d(config.get("tradeoff"), 1.0), "upper": bound == "upper",
                "sign": _d(config.get("sign"), -1.0)}

    def eval(self, model, hyp, X_obs, Y_obs, X_hid, X_pend=None, config=None):
        config = config or self.config
        model.predict_device(X_obs, Y_obs, X_hid, hyp)  # :63
        ctx = model.ctx
        ctx.score_reset()
        self.add_to(ctx, Y_obs, config)
        _, _, scores = ctx.score_finish(1.0, download=True)
        return scores

    @staticmethod
    def compute(ctx, fval, fvar, config):
        """conf_bound.compute (:70-94)."""
        bound = str(_d(config.get("bound"), "lower")).lower()
        return ctx.cb_compute(fval, fvar, _d(config.get("tradeoff"), 1.0), bound == "upper", _d(config.get("sign"), -1.0))
