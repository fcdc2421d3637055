"""bot7.bots.bayesopt (bots/bayesopt.lua): MC-marginalised acquisition over GP hyper samples, then arg-max.

eval (bots/bayesopt.lua:56-82) keeps the reference's arithmetic -- score = zeros(M); per hyper sample
score:add(acq); score:div(nSamples) -- but the accumulator lives on the GPU (b7_score_reset / _ei|_cb /
_finish) so no M-vector crosses PCIe per sample; nominate (:85-99) takes the arg-max from the same finish call
(score:max(1): first maximum, 1-based).  When only the winner is wanted (nominate) and the score has a device
spec, the whole of eval + max is ONE library call, b7_eval_nominate: same hyper samples, same arithmetic, one host
synchronisation instead of one per sample (config.bot.fused, default True)."""
import numpy as np

from .abstract import abstract
from bot7_amd import models as Models
from bot7_amd import scores as Scores


class bayesopt(abstract):
    title = "bot7.bots.bayesopt"

    def __init__(self, objective, hypers, config=None, cache=None):
        cache = cache or {}
        super().__init__(objective, hypers, config, cache)
        config = self.config
        self.model = cache.get("model") or Models.registry[config["model"]["type"]](config["model"])  # :31
        self.score = cache.get("score") or Scores.registry[config["score"]["type"]](config["score"])  # :32
        self._rng = np.random.default_rng(config["bot"]["seed"])
        self.last_scores = None

    def configure(self, config):
        config = super().configure(config)
        model = dict(config.get("model") or {})
        model.setdefault("type", "gp_regressor")        # bots/bayesopt.lua:40
        model.setdefault("kernel", "ardse")             # :41
        model.setdefault("nzModel", "GaussianNoise_iso")  # :42
        model.setdefault("mean", "constant")            # :43
        model.setdefault("sampler", "slice")            # :44
        config["model"] = model
        score = dict(config.get("score") or {})
        score.setdefault("type", "expected_improvement")  # :49
        config["score"] = score
        return config

    def eval(self, candidates=None, want_scores=True):
        """bots/bayesopt.lua:56-82.  Returns (scores or None, best_value, best_index_1based)."""
        X_obs, Y_obs = self.observed, self.responses
        X_hid = self.candidates if candidates is None else candidates
        model, ctx = self.model, self.model.ctx
        if model.class_() == "bot7.models.dngo":                  # :65-66: one score call, no marginalisation loop
            model.predict_device(X_obs, Y_obs, X_hid, None)
            ctx.score_reset()
            self.score.add_to(ctx, Y_obs)
            val, idx, scores = ctx.score_finish(1.0, download=want_scores)
            self.last_scores = scores
            return scores, val, idx
        model.sample_hypers(X_obs, Y_obs)                         # :68 (burn-in call)
        nSamples = self.config["bot"]["nSamples"]
        spec = getattr(self.score, "device_spec", None)
        if hasattr(X_hid, "commit"):
            # a candidate set sharded over GPUs: every shard scores its rows, ONE exchange names the winner in the union
            # (b7_eval_nominate with a communicator / b7_group_eval_nominate); ranks run this loop in lock step
            hyps = [model.parse_hypers(model.sample_hypers(X_obs, Y_obs, None, None, True)) for _ in range(nSamples)]
            X_hid.stage_data(X_obs, Y_obs)
            sp = spec(Y_obs)
            val, idx = X_hid.eval_nominate(hyps, sp)
            self.last_scores = None
            return None, val, idx
        if spec is not None and self.config["bot"].get("fused", True) and hasattr(model, "stage"):
            # (the driver never hands pending points to the score, :66,76)
            hyps = [model.parse_hypers(model.sample_hypers(X_obs, Y_obs, None, None, True)) for _ in range(nSamples)]
            model.stage(X_obs, Y_obs, X_hid)                      # data + grid resident (uploads only what changed)
            val, idx = ctx.eval_nominate(hyps, **spec(Y_obs))     # :73-79 + :96 in one call
            scores = ctx.score_finish(1.0, download=True)[2] if want_scores else None   # the accumulator holds score / S
            self.last_scores = scores
            return scores, val, idx
        first = True
        for _ in range(nSamples):                                 # :73-78
            hyp = model.parse_hypers(model.sample_hypers(X_obs, Y_obs, None, None, True))
            model.predict_device(X_obs, Y_obs, X_hid, hyp)        # inside score(...) -> model:predict
            if first:
                ctx.score_reset()                                 # :69 torch.zeros(M)
                first = False
            self.score.add_to(ctx, Y_obs)                         # :76 score:add(...)
        val, idx, scores = ctx.score_finish(float(nSamples), download=want_scores)  # :79 div, :96 max
        self.last_scores = scores
        return scores, val, idx

    def nominate(self, candidates=None):
        """bots/bayesopt.lua:85-99."""
        cand = self.candidates if candidates is None else candidates
        if self.nTrials <= self.config["bot"]["nInitial"]:        # :90-91 floor(rand*M)+1
            return int(np.floor(self._rng.random() * cand.shape[0])) + 1
        _, _, idx = self.eval(cand, want_scores=False)            # :95-96
        return idx
