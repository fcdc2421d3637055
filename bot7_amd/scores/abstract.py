"""bot7.scores.abstract (scores/abstract.lua:15-27)."""


class abstract(object):
    title = "bot7.scores.abstract"

    def __call__(self, model, hyp, X_obs, Y_obs, X_hid, X_pend=None, config=None):
        return self.eval(model, hyp, X_obs, Y_obs, X_hid, X_pend, config or getattr(self, "config", {}))

    def eval(self, *args):  # scores/abstract.lua:25-27
        print("Error: eval() method not implemented")
        return None

    # device-side form used by bots.bayesopt.eval: add this score of the LAST predict into the context's
    # accumulator (bots/bayesopt.lua:76 score:add(...)) without a round trip through host memory
    def add_to(self, ctx, Y_obs, config=None):
        raise NotImplementedError
