"""TEST INFRASTRUCTURE ONLY (golden-vector generation in the build container).

Minimal stand-in for the `numba` package so that the reference's hot-path
modules (plain Python once `@cuda.jit` is an identity decorator) can be
imported from /root/reference/src and executed on the CPU to produce golden
vectors.  Real numba is not importable in this image.

This is NOT numba and does no compilation: it only provides the three names
the reference touches (`cuda.jit`, `cuda.grid`, `cuda.to_device`).  It never
ships as product code and nothing outside oracle/gen_golden.py imports it.
"""
from . import cuda  # noqa: F401
