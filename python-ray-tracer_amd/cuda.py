"""Stands where `from numba import cuda` stands in the reference's driver (main.py:2): the three
names it uses — `to_device` (main.py:19-32), `DeviceNDArray.copy_to_host` (main.py:51) and
`synchronize` (absent from the reference, which is why its timing is wrong).

A DeviceNDArray keeps the host values it was created from (scene arrays and camera matrices are
consumed by the C ABI as host pointers) and, once a launch has written to it, a device buffer that
`copy_to_host()` reads back after synchronising — the launch itself is asynchronous, as in numba.
"""
import itertools

import numpy as np

from .renderer import Renderer

_default = None
_serials = itertools.count(1)


def current_renderer(device=0):
    """The process-wide context (numba's implicit CUDA context)."""
    global _default
    if _default is None:
        _default = Renderer(device)
    return _default


def close():
    global _default
    if _default is not None:
        _default.close()
        _default = None


class DeviceNDArray:
    def __init__(self, host):
        self._host = host            # values at creation time (inputs) / last copied-back values
        self._dptr = None            # device buffer, allocated when a launch writes to this array
        self._renderer = None
        self._dirty = False          # a launch has written the device buffer: it, not _host, holds the contents
        self.version = 0             # bumped whenever the contents change (launch caches key on it)
        self.serial = next(_serials) # process-unique identity (id() values are reused after garbage collection)

    shape = property(lambda self: self._host.shape)
    dtype = property(lambda self: self._host.dtype)
    ndim = property(lambda self: self._host.ndim)
    nbytes = property(lambda self: self._host.nbytes)

    def _device_buffer(self, renderer):
        # a buffer of a context that has been closed since (cuda.close(), or another Renderer is current now) is gone
        # with that context: never hand its address to a new one
        if self._dptr is not None and (self._renderer is not renderer or getattr(self._renderer, "closed", False)):
            if self._dirty and not getattr(self._renderer, "closed", False):
                self._host = self.copy_to_host()
            # (dirty under a CLOSED context: what the launch wrote went with that context; the array restarts from the
            # host values it last held — copy_to_host() before the next launch would have raised)
            self._free()
        if self._dptr is None:
            self._renderer = renderer
            self._dptr = renderer.malloc(self._host.nbytes)
        return self._dptr

    def _free(self):
        if self._dptr is not None and not getattr(self._renderer, "closed", False):
            self._renderer.free(self._dptr)
        self._dptr, self._renderer, self._dirty = None, None, False

    def copy_to_host(self):
        """A new host array with the current contents (numba's semantics).  Once a launch has written the device
        buffer, that buffer is the truth: one device-to-host copy straight into the array that is returned."""
        if self._dirty:
            if getattr(self._renderer, "closed", False):
                raise RuntimeError("DeviceNDArray: its context was closed before the contents were copied back")
            out = np.empty(self._host.shape, self._host.dtype)
            self._renderer.sync()
            self._renderer.d2h(out, self._dptr)
            return out
        return self._host.copy()

    def __array__(self, dtype=None, copy=None):
        a = self.copy_to_host()
        return a if dtype is None else a.astype(dtype)

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


def to_device(a):
    host = np.array(a, order="C")
    # the pixel grid keeps its closed form, if it has one, so the kernel can generate rays itself
    d = DeviceNDArray(host)
    d.raygen = getattr(a, "raygen", None)
    return d


def device_array(shape, dtype=np.float64):
    return DeviceNDArray(np.zeros(shape, dtype=dtype))


def synchronize():
    if _default is not None:
        _default.sync()
