"""Mirror of the reference's `ray_tracing` package surface (ray_tracing/__init__.py:1):
the single public name is the kernel object `render`."""
from .kernels import render  # noqa: F401
