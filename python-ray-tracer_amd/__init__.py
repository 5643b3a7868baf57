"""python-ray-tracer_amd — MI355X-native drop-in for the render hot path of
peter-seres/python-ray-tracer.

    from python_ray_tracer_amd import cuda                     # stands where `from numba import cuda` stood
    from python_ray_tracer_amd.ray_tracing import render       # reference: ray_tracing/__init__.py:1
    from python_ray_tracer_amd.scene import Scene, Camera      # reference: scene/__init__.py:1-3
    from python_ray_tracer_amd.viewer import convert_array_to_image

Everything that computes pixels happens in libmi355rt.so (hand-written HIP for gfx950) behind the
C ABI of include/mi355rt.h; this package is the ctypes host above it.  There is no CPU fallback:
importing works anywhere, rendering raises unless the library and an AMD GPU are present.
"""
from . import _lib            # noqa: F401  (ctypes loader; raises on use if the .so is missing)
from . import cuda            # noqa: F401
from .renderer import Renderer, RenderError   # noqa: F401

__all__ = ["cuda", "Renderer", "RenderError"]
