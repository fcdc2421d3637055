"""bot7_amd -- MI355X (gfx950) implementation of bot7's GP-posterior + acquisition-scoring hot path.

The numerical work lives in ``libbot7hip.so`` (hand-written HIP kernels behind the C ABI of
``include/bot7hip.h``).  This package is the host-side mirror of the reference's plug-in protocol
(``bot7.grids`` / ``bot7.models`` / ``bot7.scores``), so that the reference's usage reads the same here:

    from bot7_amd import grids, models, scores

(the driver that calls them, bots/bayesopt.lua, is host code that stays the reference's own; the repo's ``harness/``
package holds a Python stand-in for tests and bench).

There is no CPU fallback: importing the package is cheap, but the first call that needs the device loads
the library and raises ``Bot7HipError`` if it, or a gfx950 GPU, is missing.
"""
from ._lib import Bot7HipError, Context, Group, default_context, lib_path  # noqa: F401
from . import grids, models, scores  # noqa: F401

__all__ = ["Bot7HipError", "Context", "Group", "default_context", "lib_path", "grids", "models", "scores"]
