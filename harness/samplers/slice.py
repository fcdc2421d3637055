"""bot7.samplers.slice (samplers/slice.lua): slice sampler with random-direction or Gibbs updates, step-out and
step-in, log-space by default.  Host control flow only -- every density evaluation `f(x)` it requests from the GP
model is one b7_gp_fit(..., nll_out) on the device.

Same defaults and the same sequence of decisions as the reference; the random numbers come from a numpy
Generator (``opt['rng']`` or ``opt['seed']``) because Torch's global MT19937 stream is not part of the tree."""
import numpy as np


class slice_sampler(object):
    title = "bot7.samplers.slice"

    def __call__(self, f, X0, opt=None, f_args=None):
        opt = self.configure(opt)                 # samplers/slice.lua:25-29
        return self.sample(f, X0, opt, f_args)

    @staticmethod
    def configure(opt=None):
        """samplers/slice.lua:32-48."""
        opt = dict(opt or {})
        opt.setdefault("max_step", 1e3)           # :34
        opt.setdefault("nSamples", 1)             # :35
        if opt.get("step_out") is not False:      # :38-40
            opt["step_out"] = True
        if opt.get("logspace") is not False:      # :43-45
            opt["logspace"] = True
        if opt.get("rng") is None:
            opt["rng"] = np.random.default_rng(opt.get("seed", 0))
        return opt

    @classmethod
    def sample(cls, f, X0, opt, f_args=None):
        """samplers/slice.lua:51-89.  X0: 1 x xDim start; returns nSamples x xDim (each drawn from X0)."""
        rng = opt["rng"]
        X0 = np.atleast_2d(np.asarray(X0, dtype=np.float64))
        if int(opt["nSamples"]) != 1:             # (X0 is only read: the one-sample case needs no copy)
            X0 = np.tile(X0, (int(opt["nSamples"]), 1))
        N, xDim = X0.shape
        samples = np.empty((N, xDim))
        if opt.get("gibbs"):                      # :58-75
            for n in range(N):
                x0 = X0[n:n + 1]
                x1 = np.zeros((1, xDim))
                direction = np.zeros((1, xDim))
                for dd in rng.permutation(xDim):
                    direction[0, dd] = 1.0
                    x1[0, dd] = cls.directed_slice(opt, f, f_args, direction, x0)[0, dd]
                    direction[0, dd] = 0.0
                samples[n] = x1
        else:                                     # :78-86
            for n in range(N):
                x0 = X0[n:n + 1]
                direction = rng.standard_normal((1, xDim))
                direction = direction / np.linalg.norm(direction)
                samples[n] = cls.directed_slice(opt, f, f_args, direction, x0)
        return samples

    @staticmethod
    def directed_slice(opt, f, f_args, direction, x0):
        """samplers/slice.lua:92-168."""
        rng = opt["rng"]
        xDim = x0.shape[1]
        stepsize = opt.get("widths")
        if stepsize is None:
            stepsize = np.full((1, xDim), opt.get("width") or 1.0)   # :95
        stepsize = np.asarray(stepsize, dtype=np.float64).reshape(1, xDim)

        def f_dx(dx=None):                        # :100-103
            dx = np.zeros((1, xDim)) if dx is None else dx
            return float(f(x0 + direction * dx, f_args))

        Y = f_dx()                                # :106-111
        if opt["logspace"]:
            Y = Y + np.log(rng.random())
        else:
            Y = Y * rng.random()
        right = rng.random((1, xDim)) * stepsize  # :114-115
        left = right - stepsize
        if opt["step_out"]:                       # :118-130
            itr = 0
            while f_dx(right) > Y and itr < opt["max_step"]:
                itr += 1
                right = right + stepsize
            itr = 0
            while f_dx(left) > Y and itr < opt["max_step"]:
                itr += 1
                left = left - stepsize
        dx = np.zeros((1, xDim))
        while True:                               # :134-164
            dx = left + (right - left) * rng.random()
            y = f_dx(dx)
            if y != y:
                print("Error: samplers.slice encountered a NaN")
                break
            if y > Y:
                break
            if (dx == 0.0).any():
                print("Error: samplers.slice shrank to zero")
                break
            pos, neg = dx > 0, dx < 0
            right = np.where(pos, dx, right)      # :153-156
            left = np.where(neg, dx, left)        # :158-161
        return x0 + direction * dx                # :167
