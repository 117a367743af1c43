"""uvaia_amd -- MI355X-native engine for uvaia's nearest-neighbour hot path.

The product is the C-ABI shared library built from uvaia_amd/csrc (declared in include/uvaia_gpu.h) plus the C host
code that mirrors uvaia's fastaseq/min_heap API.  This Python package is only the thin ctypes binding used by the
tests and bench.py; it never computes anything itself and fails loudly when the HIP library is missing.
"""
from .capi import Engine, GpuError, build_library, library_path, load_library  # noqa: F401
