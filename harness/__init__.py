"""Test and bench HARNESS, not product: Python stand-ins for host code that north_star leaves in the reference's own Lua
(bots/abstract.lua, bots/bayesopt.lua, samplers/slice.lua, utils/tensor.lua, benchmarks/*.lua) plus the gloo rehearsal of
the sharded nomination.  They exist because this image has no Lua runtime to run the real files; every statement cites the
Lua line it stands in for.  The product is bot7_amd/libbot7hip.so + lua/ (+ the ctypes binding and protocol mirror in
bot7_amd/); nothing in bot7_amd/ imports this package."""
from . import benchmarks, tensor, samplers, bots, dist  # noqa: F401
