"""asif_amd -- MI355X-native batched CBF-QP safety filter.

The product is asif_amd/libasif_hip.so (hand-written HIP for gfx950 behind the C ABI of
include/asif_hip.h); this package only holds the ctypes binding and workload generators that tests
and bench.py use.  Nothing here computes a filter on the CPU.
"""
from . import capi, workloads  # noqa: F401

__all__ = ["capi", "workloads"]
