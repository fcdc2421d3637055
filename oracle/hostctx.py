"""TEST INFRASTRUCTURE ONLY (the checker, never the product): the subset of bot7_amd.Context that the reference's trial loop
(bots/abstract.lua:112-152 over bots/bayesopt.lua:56-99, as harness/bots restates it) calls, answered on the host by the oracle
(oracle/gp.py: parity unpinned, see there; oracle/cport.py: the C restatement of scores / arg-max / row removal).

Used by: tests/test_dist_gloo.py and tests/test_sharded_loop.py (the device's stand-in on a machine without a GPU),
bench.py --workload default's `cpu_baseline` leg and harness/default_regime.py (the same loop, the same random stream, on the
host cores: the nominee sequence it produces is also the parity check of the GPU run)."""
import numpy as np

from . import cport, gp


class OracleContext(object):
    def __init__(self, X_obs=None, Y=None, X_shard=None):
        self.c, self.gp, self.X_obs, self.Y, self.X = cport, gp, X_obs, Y, X_shard
        self.acc = None
        self.fit_token = 0
        self.grid_version = 0
        self.n_nll = 0

    def comm_info(self):
        return (0, 1)                      # no communicator: a world of one (ShardedScorer then calls eval_nominate here)

    # -- data / grid
    def gp_set_data(self, X_obs, Y):
        self.X_obs, self.Y = np.asarray(X_obs, dtype=np.float64), np.asarray(Y, dtype=np.float64)
        self.fit_token += 1

    def grid_sobol(self, size, dims, skip=1, mins=None, maxes=None, download=True):
        self.X = cport.sobol(size, dims, skip, mins, maxes)      # grids/sobol.lua:58-90
        self.grid_version += 1
        return self.X.copy() if download else None

    def grid_upload(self, X):
        self.X = np.array(X, dtype=np.float64)
        self.grid_version += 1

    def grid_shape(self):
        return self.X.shape

    def grid_download(self, row0=0, rows=None):
        return self.X[row0:(None if rows is None else row0 + rows)].copy()

    def grid_remove(self, idx1):           # utils.tensor.remove (utils/tensor.lua:158-170): stable deletion
        row = self.X[idx1 - 1].copy()
        self.X = np.delete(self.X, idx1 - 1, axis=0)
        self.grid_version += 1
        return row

    def nominate_commit(self, idx1_global, lo=0):
        return self.grid_remove(idx1_global - lo), lo

    # -- model:sample_hypers' density (bots/bayesopt.lua:68,73-75 -> samplers/slice.lua:92-168)
    def gp_nll_batch(self, lenscale_sq, amp, noise, mean, want_info=False):
        ls = np.atleast_2d(np.asarray(lenscale_sq, dtype=np.float64))
        B = ls.shape[0]
        a, nz, m = (np.broadcast_to(np.asarray(v, dtype=np.float64).ravel(), (B,)) for v in (amp, noise, mean))
        nll, jit, info = np.empty(B), np.empty(B), np.empty(B, dtype=np.int32)
        for b in range(B):
            f = self.gp.fit(self.X_obs, self.Y, ls[b], float(a[b]), float(nz[b]), float(m[b]))
            nll[b], jit[b], info[b] = float(f.nll[0]), f.jitter, f.info
        self.n_nll += B
        return (nll, jit, info) if want_info else nll

    # -- bayesopt:eval + nominate (bots/bayesopt.lua:56-99)
    def eval_nominate(self, hyps, score="ei", fmin=None, tradeoff=None, upper=False, sign=-1.0, global_row_offset=0):
        for s, h in enumerate(hyps):
            self.gp_predict_hyp(h["lenscale_sq"], h["amp"], h["noise"], h["mean"])
            if s == 0:
                self.score_reset()
            if score == "ei":
                self.score_ei(fmin, tradeoff or 0.0)
            else:
                self.score_cb(1.0 if tradeoff is None else tradeoff, upper, sign)
        v, i, _ = self.score_finish(float(len(hyps)))
        return v, global_row_offset + i

    def gp_predict_hyp(self, lenscale_sq, amp, noise, mean):
        f = self.gp.fit(self.X_obs, self.Y, lenscale_sq, amp, noise, mean)
        self.mu, self.var = self.gp.predict(f, self.X)

    def score_reset(self):
        self.acc = np.zeros(self.X.shape[0])

    def score_ei(self, fmin, tradeoff):
        self.c.accumulate(self.acc, self.c.ei(self.mu, self.var, fmin, tradeoff))

    def score_cb(self, tradeoff, upper, sign):
        self.c.accumulate(self.acc, self.c.cb(self.mu, self.var, tradeoff, upper, sign))

    def score_finish(self, divisor, download=False):
        self.c.divide(self.acc, divisor)
        i, v = self.c.argmax_first(self.acc)
        return v, i, (self.acc.copy() if download else None)
