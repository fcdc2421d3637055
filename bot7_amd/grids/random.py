"""bot7.grids.random (grids/random.lua:23-35): uniform grid + affine map, generated on the GPU.

torch.rand's MT19937 stream is not part of the reference tree, so the uniforms come from the documented
counter-based generator of include/bot7hip.h (``config.seed``, default 0; ``config.row_offset`` lets one rank
of a sharded job draw its own slice of the same global grid)."""
from .abstract import abstract, DeviceGrid


class random(abstract):
    title = "bot7.grids.random"

    def generate(self, config=None):
        config = self.config if config is None else config
        size, dims = int(config["size"]), int(config["dims"])
        mins, maxes = config.get("mins"), config.get("maxes")
        both = mins is not None and maxes is not None
        host = self.ctx.grid_random(size, dims, int(config.get("seed", 0)), int(config.get("row_offset", 0)),
                                    mins if both else None, maxes if both else None)
        if not both and (mins is not None or maxes is not None):
            host = self._partial_affine(host, config)
            self.ctx.grid_upload(host)
        return DeviceGrid(host, self.ctx, self.ctx.grid_version)
