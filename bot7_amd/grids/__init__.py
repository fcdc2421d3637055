"""bot7.grids registry (grids/init.lua): ``Grids[config.grid.type](config.grid)()`` -> M x d candidates."""
from .abstract import abstract, DeviceGrid  # noqa: F401
from .sobol import sobol  # noqa: F401
from .random import random  # noqa: F401

registry = {"sobol": sobol, "random": random}

