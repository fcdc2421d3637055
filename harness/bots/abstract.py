"""bot7.bots.abstract (bots/abstract.lua): experiment loop and candidate / pending / observed bookkeeping.

Kept on the host exactly as in the reference; the only change is that the candidate set also lives on the GPU
(the grid classes put it there) and row `idx` is deleted there with the same stable semantics as
utils.tensor.remove (rows after idx shift up by one), so indices returned by nominate keep their meaning."""
import copy

import numpy as np

from bot7_amd import grids as Grids
from bot7_amd.grids.abstract import DeviceGrid
from .. import tensor as T


class abstract(object):
    title = "bot7.bots.abstract"

    def __init__(self, objective, hypers, config=None, cache=None):
        cache = cache or {}                                       # bots/abstract.lua:23
        self.hypers = cache.get("hypers") or hypers               # :24
        self.objective = cache.get("objective") or objective      # :25
        self.config = self.configure(cache.get("config") or config or {})  # :26
        config = self.config
        self.candidates = cache.get("candidates")                 # :30
        if self.candidates is None:
            if config["bot"]["verbose"] > 1:
                print("> Generating candidate grid...")
            self.candidates = Grids.registry[config["grid"]["type"]](config["grid"])()  # :35
        self.responses = cache.get("responses")                   # :37
        self.observed = cache.get("observed")                     # :38
        self.pending = None
        self.nTrials = 0 if self.observed is None else self.observed.shape[0]  # :39-43
        self.best = {"x": np.empty((1, config["grid"]["dims"]))}   # :46-47
        if self.responses is not None:
            self.best["y"] = self.responses.min(axis=0)
            self.best["t"] = int(self.responses[:, 0].argmin()) + 1
        else:
            self.best["t"] = -1
        self.model = None

    def configure(self, config):
        config = copy.deepcopy(dict(config))                      # :57
        bot = dict(config.get("bot") or {})
        bot.setdefault("verbose", 3)                              # :63
        bot.setdefault("budget", 100)                             # :64
        bot.setdefault("msg_freq", 1)                             # :65
        bot.setdefault("nInitial", 2)                             # :66
        bot.setdefault("nSamples", 10)                            # :67
        bot.setdefault("save", False)                             # :68
        bot.setdefault("seed", 0)  # ours: the reference draws the initial picks from Torch's global RNG
        config["bot"] = bot
        score = dict(config.get("score") or {})
        score.setdefault("type", "expected_improvement")          # :73
        config["score"] = score
        grid = dict(config.get("grid") or {})
        grid.setdefault("type", "sobol")                          # :79
        grid.setdefault("size", int(2e4))                         # :80
        hyp_list = list(self.hypers.values()) if isinstance(self.hypers, dict) else list(self.hypers or [])
        if grid.get("dims") is None:                              # :83-89
            grid["dims"] = int(sum(getattr(h, "size", 1) for h in hyp_list))
        if grid.get("mins") is None:                              # :91-97
            grid["mins"] = np.concatenate([np.full(getattr(h, "size", 1), float(h.min)) for h in hyp_list]) \
                if hyp_list else np.zeros(grid["dims"])
        if grid.get("maxes") is None:                             # :99-105
            grid["maxes"] = np.concatenate([np.full(getattr(h, "size", 1), float(h.max)) for h in hyp_list]) \
                if hyp_list else np.ones(grid["dims"])
        config["grid"] = grid
        return config

    # ---- candidate bookkeeping -----------------------------------------------------------------------
    def _steal_candidate(self, idx1):
        """pending, candidates = steal(pending, candidates, idx)  (bots/abstract.lua:118)."""
        cand = self.candidates
        if hasattr(cand, "commit"):
            # the candidate set is sharded over GPUs (harness/dist.py: ShardedScorer per rank, GroupCandidates in one
            # process): idx1 is 1-based in the UNION; the library hands every rank the nominee's coordinates and deletes
            # the row where it lives (b7_nominate_commit / b7_group_nominate_commit)
            row = np.asarray(cand.commit(idx1), dtype=np.float64)
            self.pending = T.append(self.pending, row.reshape(1, -1))
            return
        row = np.array(cand[idx1 - 1], dtype=np.float64)
        self.pending = T.append(self.pending, row.reshape(1, -1))
        host = T.remove(np.asarray(cand), [idx1])
        if isinstance(cand, DeviceGrid) and cand.ctx is not None and cand.version == cand.ctx.grid_version:
            dev_row = cand.ctx.grid_remove(idx1)   # same stable deletion on the resident copy
            assert np.array_equal(dev_row, row)
            self.candidates = None if host is None else DeviceGrid(host, cand.ctx, cand.ctx.grid_version)
        else:
            self.candidates = host

    def run_trial(self):
        """bots/abstract.lua:112-152."""
        self.nTrials += 1
        idx = int(self.nominate())                                 # :117
        self._steal_candidate(idx)                                 # :118
        idx = self.pending.shape[0]                                # :120
        nominee = self.pending[idx - 1]                            # :121
        y = self.objective(nominee)                                # :124
        y = np.asarray(y, dtype=np.float64).reshape(1, -1) if np.ndim(y) < 2 else np.asarray(y, dtype=np.float64)
        self.responses = y if self.nTrials == 1 or self.responses is None else np.concatenate([self.responses, y], 0)
        self.observed, self.pending = T.steal(self.observed, self.pending, [idx])  # :143-144
        if self.model is not None and self.nTrials == self.config["bot"]["nInitial"]:
            self.model.init(self.observed, self.responses)         # :147-149
        return nominee, y

    def run_experiment(self):
        """bots/abstract.lua:155-169."""
        x = y = None
        for t in range(1, self.config["bot"]["budget"] + 1):
            x, y = self.run_trial()
            self.update_best(x, y)
            self.progress_report(t, x, y)
        return self.best

    def update_best(self, x, y):
        """bots/abstract.lua:171-177."""
        if self.best.get("y") is None or bool(np.all(self.best["y"] > y)):
            self.best["t"] = self.nTrials
            self.best["x"] = x
            self.best["y"] = y

    def progress_report(self, t, x, y):
        cfg = self.config["bot"]
        if cfg["verbose"] > 0 and t % cfg["msg_freq"] == 0:
            print("trial %4d  y = %-14.8g best = %-14.8g (trial %d)" %
                  (t, float(np.ravel(y)[0]), float(np.ravel(self.best["y"])[0]), self.best["t"]))

    def nominate(self):  # bots/abstract.lua:224-226
        print("Error: nominate() method not implemented")
        return None
