"""bot7.scores registry (scores/init.lua:15-20)."""
from .abstract import abstract  # noqa: F401
from .expected_improvement import expected_improvement  # noqa: F401
from .confidence_bound import confidence_bound  # noqa: F401

registry = {"expected_improvement": expected_improvement, "confidence_bound": confidence_bound}
