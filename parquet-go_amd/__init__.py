"""MI355X-native Parquet page-value decoder (package directory `parquet-go_amd`).

Layout:
  csrc/      hand-written gfx950 HIP kernels + the C ABI (include/pqdecode.h) -> libpqdecode.so
  capi.py    ctypes binding of that ABI (tests, benchmark)
  synth.py   synthetic page writers (numpy) for tests and the benchmark

Importable as `parquet_go_amd` through the alias module at the repository root
(the directory name is not a Python identifier).
"""
from . import capi, synth  # noqa: F401

__all__ = ["capi", "synth"]
