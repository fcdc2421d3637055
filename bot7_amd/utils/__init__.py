"""Host-side bookkeeping helpers mirroring utils/tensor.lua (the compute lives in libbot7hip.so)."""
from . import tensor  # noqa: F401
