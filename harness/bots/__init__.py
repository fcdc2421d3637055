"""bot7.bots registry (bots/init.lua).  random_search is out of scope (no model/score on its path)."""
from .abstract import abstract  # noqa: F401
from .bayesopt import bayesopt  # noqa: F401

registry = {"bayesopt": bayesopt, "bo": bayesopt}
