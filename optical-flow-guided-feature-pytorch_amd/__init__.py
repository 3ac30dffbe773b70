"""offk -- MI355X-native OFF (Optical-Flow-guided Feature) sub-network forward.

Hand-written HIP/CDNA4 kernels behind the C-ABI library ``liboffk.so``
(include/offk.h) plus the host-side mirror of the reference's ``BNInception_OFF``
interface for that one path.  Import as ``offk_amd`` (see offk_amd.py at the repo root).
"""
from . import spec, synth  # noqa: F401  (pure-python, no GPU needed)

__all__ = ["spec", "synth"]
