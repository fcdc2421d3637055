"""bot7.samplers registry (samplers/init.lua): host-side control flow, as in the reference."""
from .slice import slice_sampler  # noqa: F401

registry = {"slice": slice_sampler}

# the model mirror looks its sampler up here (bots/bayesopt.lua:44 `sampler = 'slice'`): in the reference that is
# bot7.samplers.slice, host code that stays Lua
from bot7_amd.models.abstract import sampler_registry as _sampler_registry  # noqa: E402
_sampler_registry.update(registry)
