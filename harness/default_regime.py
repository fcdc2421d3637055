"""HARNESS.  The reference's own default experiment (examples/run_benchmark.lua:30-36 over bots/abstract.lua:63-67,79-80):
hartmann6 (d = 6), a 2e4-point Sobol candidate grid, budget 100 (so N <= 100), nInitial 2, nSamples 10 slice-sampled hyper
vectors per nomination, expected improvement -- run as the trial loop of bots/abstract.lua:112-152 with every piece of
arithmetic behind the context it is handed:

    model:sample_hypers   -> samplers/slice.lua on the host, each density evaluation ONE b7_gp_nll_batch call
    bayesopt:eval + nominate -> ONE b7_eval_nominate call (S fits, K(X*,X), posterior, score:add, score:div, arg-max)
    steal(pending, candidates, idx) -> b7_nominate_commit (the nominee's row from the exchange record, stable deletion enqueued)

The SAME function drives the device (bot7_amd.Context) and the oracle (oracle.hostctx.OracleContext) -- same seeds, same
random streams, same host code -- so the two nominee sequences are comparable trial by trial: bench.py --workload default times
both and reports where they agree.  Per trial the wall time is split into sampler / nominate / commit (the objective, a
6-dimensional closed form, is excluded)."""
import time

import numpy as np

from . import benchmarks, bots, dist


class H(object):  # a hyperparam as bots/abstract.lua:83-105 reads it: name, min, max, size
    def __init__(self, name, lo=0.0, hi=1.0):
        self.name, self.min, self.max, self.size = name, lo, hi, 1


DEFAULTS = {"objective": "hartmann6", "dims": 6, "grid_size": 20000, "budget": 100, "nInitial": 2, "nSamples": 10,
            "score": "expected_improvement", "bot_seed": 4, "model_seed": 6, "nBurnin": 0}


def run(ctx, trials=None, on_trial=None, out=None, **over):
    """Runs `trials` trials (default: the whole budget) of the default experiment on `ctx`.  Returns
    {"nominees": [global 1-based index per trial], "X": trials x d, "Y": trials x 1, "draws": per model-based trial the S hyper
    vectors, "per_trial": [{"N", "sampler_ms", "nll_calls", "nominate_ms", "commit_ms", "trial_ms"}], "best": ...}."""
    import bot7_amd
    cfg = dict(DEFAULTS, **over)
    d, budget = int(cfg["dims"]), int(cfg["budget"])
    trials = budget if trials is None else min(int(trials), budget)
    fn = benchmarks.registry[cfg["objective"]]
    config = {"bot": {"verbose": 0, "budget": budget, "nInitial": int(cfg["nInitial"]), "nSamples": int(cfg["nSamples"]),
                      "seed": int(cfg["bot_seed"])},
              "grid": {"type": "sobol", "size": int(cfg["grid_size"]), "dims": d}, "score": {"type": cfg["score"]}}
    model = bot7_amd.models.gp_regressor({"sample": True, "nBurnin": int(cfg["nBurnin"]), "seed": int(cfg["model_seed"])},
                                         context=ctx)
    # the candidate grid on the context (grids/sobol.lua:58-90 defaults: skip 1, no affine map), handed to the bot as a
    # sharded set with a world of one: nominations are global indices, the steal is b7_nominate_commit
    ctx.grid_sobol(int(cfg["grid_size"]), d, 1, download=False)
    cand = dist.ShardedScorer(ctx, int(cfg["grid_size"]), 0, 1)
    bot = bots.bayesopt(fn, [H("x%d" % (k + 1)) for k in range(d)], config, cache={"candidates": cand, "model": model})

    clock = {"sampler": 0.0, "nominate": 0.0, "commit": 0.0, "nll": 0}
    draws_now = []

    def timed(name, f, after=None):
        def g(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            clock[name] += time.perf_counter() - t0
            if after:
                after(r)
            return r
        return g

    model.sample_hypers = timed("sampler", model.sample_hypers, lambda v: draws_now.append(np.array(v, dtype=np.float64)))
    cand.eval_nominate = timed("nominate", cand.eval_nominate)
    cand.commit = timed("commit", cand.commit)
    inner_nominate = bot.nominate
    picked = []

    def nominate(*a, **k):
        i = inner_nominate(*a, **k)
        picked.append(int(i))
        return i
    bot.nominate = nominate

    out = {} if out is None else out   # filled as the loop goes: a caller that stops it from on_trial keeps what was done
    out.update({"nominees": picked, "draws": [], "per_trial": [], "config": cfg})
    for t in range(trials):
        for k in ("sampler", "nominate", "commit"):
            clock[k] = 0.0
        del draws_now[:]
        n0 = getattr(model, "nEvals", 0)
        t0 = time.perf_counter()
        x, y = bot.run_trial()
        wall = time.perf_counter() - t0
        bot.update_best(x, y)
        N = 0 if bot.observed is None else bot.observed.shape[0] - 1   # observations the nomination of this trial was made from
        rec = {"trial": t + 1, "N": int(N), "sampler_ms": clock["sampler"] * 1e3, "nll_calls": int(getattr(model, "nEvals", 0) - n0),
               "nominate_ms": clock["nominate"] * 1e3, "commit_ms": clock["commit"] * 1e3, "trial_ms": wall * 1e3}
        rec["nominee"] = picked[-1]
        out["per_trial"].append(rec)
        # the burn-in call (bots/bayesopt.lua:68) returns the chain's current point; the S per-sample draws follow it
        out["draws"].append(np.array(draws_now[1:]) if len(draws_now) > 1 else np.zeros((0, d + 3)))
        if on_trial:
            on_trial(rec)
    out["X"] = np.array(bot.observed, dtype=np.float64)
    out["Y"] = np.array(bot.responses, dtype=np.float64)
    out["best"] = {"t": int(bot.best["t"]), "y": float(np.ravel(bot.best["y"])[0])}
    out["candidates_left"] = int(cand.M_global)
    return out


def agreement(a, b):
    """How far two runs of the same experiment agree: number of leading trials with the same nominee, and the largest relative
    difference between their hyper draws over those trials."""
    n = 0
    worst = 0.0
    for i, (p, q) in enumerate(zip(a["nominees"], b["nominees"])):
        if p != q:
            break
        n = i + 1
        da, db = a["draws"][i], b["draws"][i]
        if da.shape == db.shape and da.size:
            worst = max(worst, float(np.max(np.abs(da - db) / np.maximum(np.abs(db), 1e-300))))
    return n, worst


def summarise(per_trial, at=(25, 64, 100)):
    """ms per trial split sampler / nominate / commit at the trials whose nomination used N = at[i] observations (the nearest
    model-based trial when the run stopped earlier), and the totals over the model-based trials."""
    mb = [r for r in per_trial if r["nll_calls"] > 0 or r["nominate_ms"] > 0.0]
    out = {"at_N": {}}
    for n in at:
        near = [r for r in mb if abs(r["N"] - n) <= 1 or r["N"] == n]
        if not near:
            continue
        near = sorted(near, key=lambda r: abs(r["N"] - n))[:3]
        out["at_N"][str(n)] = {k: round(float(np.mean([r[k] for r in near])), 4)
                               for k in ("sampler_ms", "nominate_ms", "commit_ms", "trial_ms", "nll_calls")}
    if mb:
        out["model_based_trials"] = len(mb)
        for k in ("sampler_ms", "nominate_ms", "commit_ms", "trial_ms"):
            out["sum_" + k] = round(float(sum(r[k] for r in mb)), 3)
        out["nll_calls"] = int(sum(r["nll_calls"] for r in mb))
        out["ms_per_nll_call_incl_host_sampler"] = round(out["sum_sampler_ms"] / max(1, out["nll_calls"]), 5)
    return out
