"""bot7.samplers registry (samplers/init.lua): host-side control flow, as in the reference."""
from .slice import slice_sampler  # noqa: F401

registry = {"slice": slice_sampler}
