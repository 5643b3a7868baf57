"""TEST INFRASTRUCTURE ONLY — see package docstring.

Covers the three decorator spellings used by the reference
(src/ray_tracing/kernels.py:6 bare `@cuda.jit`; common.py:5
`@cuda.jit(device=True)`; trace.py:44 `@cuda.jit(func_or_sig=None, device=True)`),
`cuda.grid(2)` (kernels.py:10) and `cuda.to_device` (main.py:19-32).

A "kernel launch" is a plain Python loop over the thread coordinates the
generator asks for; `grid(2)` returns the coordinate of the current iteration.
The arithmetic that results is NumPy-scalar / Python-float IEEE arithmetic,
which is what NUMBA_ENABLE_CUDASIM=1 would execute as well.
"""
import numpy as np

_current = [0, 0]


def grid(ndim):
    assert ndim == 2
    return _current[0], _current[1]


class _Launcher:
    def __init__(self, fn, griddim, blockdim, coords=None):
        self.fn, self.griddim, self.blockdim, self.coords = fn, griddim, blockdim, coords

    def __call__(self, *args):
        if self.coords is not None:
            it = self.coords
        else:
            gx, gy = self.griddim
            bx, by = self.blockdim
            it = ((x, y) for x in range(gx * bx) for y in range(gy * by))
        for x, y in it:
            _current[0], _current[1] = x, y
            self.fn(*args)


class _Kernel:
    """Object returned for a bare `@cuda.jit` (a __global__ kernel)."""

    def __init__(self, fn):
        self.py_func = fn

    def __getitem__(self, cfg):
        griddim, blockdim = cfg
        return _Launcher(self.py_func, griddim, blockdim)

    def over(self, coords):
        """Generator-only extension: run the kernel body for an explicit list of
        thread coordinates (used to stay inside the region where the reference is
        defined, SURVEY.md §8-Q2/Q3, and to split a frame across processes)."""
        return _Launcher(self.py_func, None, None, coords)


def jit(func_or_sig=None, device=False, **_kw):
    def wrap(fn):
        return fn if device else _Kernel(fn)

    if callable(func_or_sig):
        return wrap(func_or_sig)
    return wrap


class _DeviceArray(np.ndarray):
    def copy_to_host(self):
        return np.array(self)


def to_device(a):
    return np.array(a).view(_DeviceArray)
