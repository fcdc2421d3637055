"""bot7.grids.abstract (grids/abstract.lua:14-27) and the host view of a device-resident grid."""
import numpy as np

from .._lib import default_context


class DeviceGrid(np.ndarray):
    """Host copy of a candidate grid that also remembers which context holds it on the GPU.

    bots.abstract keeps ``self.candidates`` as a plain tensor; this subclass lets ``model.predict`` recognise
    "X_hid is the grid already resident on my context" and skip the host->device upload.  ``version`` is
    bumped by every removal so a stale host view is never mistaken for the resident one."""

    def __new__(cls, array, ctx=None, version=0):
        obj = np.asarray(array, dtype=np.float64).view(cls)
        obj.ctx = ctx
        obj.version = version
        return obj

    def __array_finalize__(self, obj):
        # Only the explicit constructor tags an array as "the grid resident on ctx": anything derived from it
        # (a slice, a permutation, grid*2+0.1, a normalised copy) has different rows than the device holds and
        # must be uploaded like a plain ndarray.
        self.ctx = None
        self.version = -1


class abstract(object):
    """grid:__call__(config) -> self:generate(config or self.config) (grids/abstract.lua:20-23)."""

    def __init__(self, config=None, context=None):
        self.config = dict(config or {})
        self._ctx = context

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    def __call__(self, config=None):
        return self.generate(config if config is not None else self.config)

    def generate(self, config):  # grids/abstract.lua:25-27 prints an error and returns nil
        print("Error: generate() method not implemented")
        return None

    @staticmethod
    def _partial_affine(grid, config):
        """The one-sided branches of grids/sobol.lua:82-86 / grids/random.lua:30-34 (host side, rare)."""
        mins, maxes = config.get("mins"), config.get("maxes")
        if mins is not None and maxes is None:  # grid:add(mins + grid:min(1)[1])
            return grid + (np.asarray(mins, dtype=np.float64).reshape(1, -1) + grid.min(axis=0))
        if maxes is not None and mins is None:  # grid:cmul(maxes / grid:max(1)[1])
            return grid * (np.asarray(maxes, dtype=np.float64).reshape(1, -1) / grid.max(axis=0))
        return grid
