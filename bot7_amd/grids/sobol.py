"""bot7.grids.sobol (grids/sobol.lua): Bratley-Fox i4_sobol grid, generated on the GPU.

Same constructor contract as the reference (grids/sobol.lua:27-56): ``config.size`` and ``config.dims`` are
required and ``dims < max_dims (40)``; ``config.skip`` defaults to 1 (:70); with both ``mins`` and ``maxes``
the points are mapped by ``x*(maxes-mins)+mins`` (:79-81)."""
import numpy as np

from .abstract import abstract, DeviceGrid


class sobol(abstract):
    title = "bot7.grids.sobol"

    def __init__(self, config=None, context=None):
        super().__init__(config, context)
        C = self.config
        C.setdefault("max_dims", 40)   # grids/sobol.lua:31
        C.setdefault("log_max", 30)    # :32
        assert C.get("size") is not None                                   # :35
        assert C.get("dims") is not None and C["dims"] < C["max_dims"]     # :36

    def generate(self, config=None):
        config = self.config if config is None else config
        size, dims = int(config["size"]), int(config["dims"])
        skip = config.get("skip")
        skip = 1 if skip is None else int(skip)  # `config.skip or 1`, :70 -- in Lua 0 is truthy: skip = 0 stays 0
        mins, maxes = config.get("mins"), config.get("maxes")
        both = mins is not None and maxes is not None
        host = self.ctx.grid_sobol(size, dims, skip, mins if both else None, maxes if both else None)
        if not both and (mins is not None or maxes is not None):
            host = self._partial_affine(host, config)
            self.ctx.grid_upload(host)
        return DeviceGrid(host, self.ctx, self.ctx.grid_version)
