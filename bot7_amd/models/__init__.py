"""bot7.models registry.  In the reference this table IS ``require('gp.models')`` plus ``abstract`` and ``dngo``
(models/init.lua:15-17); here ``gp_regressor`` is the HIP-backed drop-in for gp.models.gp_regressor."""
from .abstract import abstract  # noqa: F401
from .gp_regressor import gp_regressor  # noqa: F401
from .dngo import dngo  # noqa: F401

registry = {"gp_regressor": gp_regressor, "dngo": dngo}
