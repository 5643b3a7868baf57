"""Renderer — thin object wrapper over the C ABI (include/mi355rt.h).  One Renderer = one rt_ctx
= one GPU.  All pixel work happens in libmi355rt.so; this class only marshals arrays."""
import ctypes as C
import itertools
import weakref

import numpy as np

from . import _lib as L


class RenderError(RuntimeError):
    """A C-ABI call returned a non-zero rt_status."""

    def __init__(self, status, message):
        super().__init__(f"{L.STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


def refl_powers(refl, depth):
    """refl ** (i+1) evaluated on the host exactly as the reference does (trace.py:131)."""
    out = np.zeros(L.RT_MAX_DEPTH, dtype=np.float64)
    refl = float(refl)
    for i in range(min(int(depth), L.RT_MAX_DEPTH)):
        out[i] = refl ** (i + 1)
    return out


def _f32(a, rows, name):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[0] != rows:
        raise ValueError(f"{name} must have shape ({rows}, N), got {a.shape}")
    return a


_serials = itertools.count(1)


class _PinnedBlock:
    """One page-locked allocation of rt_host_alloc.  It lives as long as any numpy view of it does (the view's buffer
    object holds it) and is freed when the last one goes — not when the Renderer that made it closes, whose close() would
    otherwise leave those views pointing at freed memory.  release() frees it at once (no view may be used afterwards)."""

    def __init__(self, lib, ptr):
        self.lib, self.ptr = lib, ptr

    def release(self):
        if self.ptr:
            ptr, self.ptr = self.ptr, 0
            self.lib.rt_host_free(None, C.c_void_p(ptr))     # ctx = NULL: no context needed to free page-locked memory

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Renderer:
    def __init__(self, device=0, lib=None):
        self.serial = next(_serials)     # process-unique (id() values are reused after garbage collection)
        self._lib = lib if lib is not None else L.load()
        self._ctx = C.c_void_p()
        st = self._lib.rt_create(C.byref(self._ctx), int(device))
        if st != L.RT_OK:
            raise RenderError(st, self._lib.rt_last_error(None).decode())
        self.device = int(device)
        self.w = self.h = None
        self._pinned = weakref.WeakValueDictionary()     # host_array(): data address -> _PinnedBlock (owned by the arrays)
        self.closed = False
        self.generation = {"scene": 0, "camera": 0, "grid": 0}   # bumped by every set_*: caches above this class key on it

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, st):
        if st != L.RT_OK:
            raise RenderError(st, self._lib.rt_last_error(self._ctx).decode())

    def close(self):
        # (page-locked arrays handed out by host_array() stay valid: each is freed with its last numpy view)
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()
        self.closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- inputs -----------------------------------------------------------------------------
    def set_scene(self, spheres, lights, planes, flags=0):
        """float32 (7,S), (3,L), (9,P) as Scene.generate_scene() returns them (scene/scene.py:96-97)."""
        s, l, p = _f32(spheres, 7, "spheres"), _f32(lights, 3, "lights"), _f32(planes, 9, "planes")
        fp = C.POINTER(C.c_float)
        self._check(self._lib.rt_set_scene(self._ctx, s.ctypes.data_as(fp), s.shape[1], l.ctypes.data_as(fp), l.shape[1],
                                           p.ctypes.data_as(fp), p.shape[1], int(flags)))
        self.counts = (s.shape[1], l.shape[1], p.shape[1])
        self.generation["scene"] += 1

    def set_camera(self, origin, rotation):
        """camera_origin (3,), camera_rotation (3,3); forced to float64 (an all-int position list
        would otherwise become int64)."""
        o = np.ascontiguousarray(origin, dtype=np.float64).reshape(3)
        r = np.ascontiguousarray(rotation, dtype=np.float64).reshape(9)
        dp = C.POINTER(C.c_double)
        self._check(self._lib.rt_set_camera(self._ctx, o.ctypes.data_as(dp), r.ctypes.data_as(dp)))
        self.generation["camera"] += 1

    def set_raygen(self, w, h, px, y0, dy, z0, dz):
        self._check(self._lib.rt_set_raygen(self._ctx, int(w), int(h), float(px), float(y0), float(dy), float(z0), float(dz)))
        self.w, self.h = int(w), int(h)
        self.generation["grid"] += 1

    def set_pixel_loc(self, pixel_loc):
        a = np.ascontiguousarray(pixel_loc, dtype=np.float64)
        if a.ndim != 3 or a.shape[0] != 3:
            raise ValueError(f"pixel_loc must have shape (3, w, h), got {a.shape}")
        self._check(self._lib.rt_set_pixel_loc(self._ctx, a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[1], a.shape[2]))
        self.w, self.h = a.shape[1], a.shape[2]
        self.generation["grid"] += 1

    def set_grid(self, pixel_loc):
        """Use the closed form when the array carries one (scene.camera.PixelGrid), else upload it."""
        rg = getattr(pixel_loc, "raygen", None)
        if rg is not None:
            self.set_raygen(pixel_loc.shape[1], pixel_loc.shape[2], *rg)
        else:
            self.set_pixel_loc(pixel_loc)

    @staticmethod
    def params(amb, lamb, refl, depth, aa=False, flags=0, refl_pow=None, spp=0, seed=1):
        """aa: False/0 none, True/1 the reference's 9-tap mode, 2 (with spp, seed) stochastic supersampling."""
        p = L.rt_params()
        p.amb, p.lamb = float(amb), float(lamb)
        rp = refl_powers(refl, depth) if refl_pow is None else np.asarray(refl_pow, dtype=np.float64)
        for i in range(min(len(rp), L.RT_MAX_DEPTH)):
            p.refl_pow[i] = float(rp[i])
        p.depth, p.aa_mode, p.flags = int(depth), int(aa), int(flags)
        p.spp, p.seed = int(spp), int(seed) & 0xFFFFFFFF
        return p

    # -- launches ---------------------------------------------------------------------------
    def render(self, amb, lamb, refl, depth, aa=False, *, x0=0, x1=None, u8=True, f32=False, flags=0, refl_pow=None,
               spp=0, seed=1):
        """Synchronous render of columns [x0,x1) into new host arrays of shape (3, x1-x0, h)."""
        x1 = (self.w or 0) if x1 is None else int(x1)
        p = self.params(amb, lamb, refl, depth, aa, flags, refl_pow, spp, seed)
        n = max(x1 - int(x0), 0)     # a bad range still goes to the library, which reports it
        hwc = bool(int(flags) & L.RT_FLAG_U8_HWC)
        out8 = np.empty(((self.h or 0), n, 3) if hwc else (3, n, self.h or 0), np.uint8) if u8 else None
        out32 = np.empty((3, n, self.h or 0), np.float32) if f32 else None
        self._check(self._lib.rt_render(self._ctx, C.byref(p), int(x0), x1,
                                        out8.ctypes.data if u8 else None, out32.ctypes.data if f32 else None))
        return out8, out32

    def render_into(self, amb, lamb, refl, depth, aa, out8, out32=None, *, x0=0, x1=None, flags=0, refl_pow=None, spp=0, seed=1):
        """Synchronous render into caller-provided host arrays (C-contiguous uint8 / float32 of shape (3, x1-x0, h);
        page-locked ones from host_array() make the copy back run at the link's rate)."""
        x1 = (self.w or 0) if x1 is None else int(x1)
        p = self.params(amb, lamb, refl, depth, aa, flags, refl_pow, spp, seed)
        for a, dt in ((out8, np.uint8), (out32, np.float32)):
            if a is not None and (a.dtype != dt or not a.flags["C_CONTIGUOUS"] or a.size != 3 * (x1 - int(x0)) * (self.h or 0)):   # (3,ws,h), or (h,ws,3) with RT_FLAG_U8_HWC
                raise ValueError("output arrays must be C-contiguous uint8 / float32 with 3*(x1-x0)*h elements")
        self._check(self._lib.rt_render(self._ctx, C.byref(p), int(x0), x1,
                                        out8.ctypes.data if out8 is not None else None,
                                        out32.ctypes.data if out32 is not None else None))

    def render_begin(self, slot, amb, lamb, refl, depth, aa, out8, out32=None, *, x0=0, x1=None, flags=0, refl_pow=None, spp=0, seed=1):
        """render_into() without the wait: the frame is queued on `slot` (0 .. RENDER_SLOTS-1) and arrives in the arrays by
        render_end(slot).  Frames on different slots overlap (copy of one, rendering of the next); use page-locked
        arrays (host_array()) — a copy into pageable memory is staged by the runtime on the calling thread."""
        x1 = (self.w or 0) if x1 is None else int(x1)
        p = self.params(amb, lamb, refl, depth, aa, flags, refl_pow, spp, seed)
        for a, dt in ((out8, np.uint8), (out32, np.float32)):
            if a is not None and (a.dtype != dt or not a.flags["C_CONTIGUOUS"] or a.size != 3 * (x1 - int(x0)) * (self.h or 0)):
                raise ValueError("output arrays must be C-contiguous uint8 / float32 with 3*(x1-x0)*h elements")
        self._check(self._lib.rt_render_begin(self._ctx, C.byref(p), int(x0), x1,
                                              out8.ctypes.data if out8 is not None else None,
                                              out32.ctypes.data if out32 is not None else None, int(slot)))

    def render_end(self, slot):
        self._check(self._lib.rt_render_end(self._ctx, int(slot)))

    def host_array(self, shape, dtype):
        """A page-locked (pinned) numpy array owned by the library; give it back with release_host_array()."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = C.c_void_p()
        self._check(self._lib.rt_host_alloc(self._ctx, n, C.byref(ptr)))
        buf = (C.c_uint8 * n).from_address(ptr.value)
        buf._owner = block = _PinnedBlock(self._lib, ptr.value)     # every view of `buf` keeps the allocation alive
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned[arr.ctypes.data] = block
        return arr

    def release_host_array(self, arr):
        """Free the allocation behind `arr` now; neither `arr` nor any other view of it may be used afterwards."""
        block = self._pinned.pop(arr.ctypes.data, None)
        if block is not None:
            block.release()

    def host_arrays(self, want_f32=False, pinned=True):
        """(uint8, float32 or None) output arrays for a full frame, page-locked or plain."""
        shape = (3, self.w, self.h)
        mk = (lambda dt: self.host_array(shape, dt)) if pinned else (lambda dt: np.empty(shape, dt))
        return mk(np.uint8), (mk(np.float32) if want_f32 else None)

    def release_host_arrays(self, arrs):
        for a in arrs:
            if a is not None:
                self.release_host_array(a)

    def render_device(self, params, x0, x1, d_u8=None, d_f32=None, plane_stride=None, stream=None):
        """Asynchronous render into caller-owned device memory (raw addresses, e.g. tensor.data_ptr())."""
        if plane_stride is None:
            plane_stride = (int(x1) - int(x0)) * self.h
        self._check(self._lib.rt_render_device(self._ctx, C.byref(params), int(x0), int(x1),
                                               C.c_void_p(d_u8) if d_u8 else None, C.c_void_p(d_f32) if d_f32 else None,
                                               int(plane_stride), C.c_void_p(stream) if stream else None))

    def render_sequence(self, params, x0, x1, n, d_u8=None, d_f32=None, plane_stride=None, frame_stride=None, cameras=None,
                        streams=None, frames_per_launch=0):
        """n frames into device memory with ONE call (rt_render_sequence): frame i at d_u8 + i*frame_stride bytes /
        d_f32 + i*frame_stride floats.  cameras=None: n frames of the current camera, `frames_per_launch` of them per launch
        (one launch's frames share a grid), launch g on streams[g % len(streams)].  cameras = float64 (n, 12) array of
        (origin, rotation) rows: one launch per frame, frame i on streams[i % len(streams)]."""
        if plane_stride is None:
            plane_stride = (int(x1) - int(x0)) * self.h
        if frame_stride is None:
            frame_stride = 3 * int(plane_stride)
        cam = None
        if cameras is not None:
            cam = np.ascontiguousarray(cameras, dtype=np.float64).reshape(-1, 12)
            if cam.shape[0] != int(n):
                raise ValueError(f"cameras must hold {n} rows of 12 doubles, got {cam.shape}")
            self.generation["camera"] += 1
        sv = tuple(int(s_) for s_ in streams) if streams else ()
        if getattr(self, "_seq_streams_key", None) != sv:       # the ctypes array of handles is built once per set of streams
            self._seq_streams_key = sv
            self._seq_streams = (C.c_void_p * len(sv))(*sv) if sv else None
        self._check(self._lib.rt_render_sequence(self._ctx, C.byref(params), int(x0), int(x1), int(n),
                                                 C.c_void_p(d_u8) if d_u8 else None, C.c_void_p(d_f32) if d_f32 else None,
                                                 int(plane_stride), int(frame_stride),
                                                 cam.ctypes.data_as(C.POINTER(C.c_double)) if cam is not None else None,
                                                 self._seq_streams, len(sv), int(frames_per_launch)))

    def sync(self, stream=None):
        """Wait for the context's stream, or for `stream` (a handle from stream_create / a hipStream_t address)."""
        if stream:
            self._check(self._lib.rt_stream_sync(self._ctx, C.c_void_p(stream)))
        else:
            self._check(self._lib.rt_sync(self._ctx))

    def stream_create(self):
        """An extra stream of this context's device: queue the frames of a sequence alternately on two or more
        streams (each into its own output buffers) and one frame's tail overlaps the next frame's head."""
        s = C.c_void_p()
        self._check(self._lib.rt_stream_create(self._ctx, C.byref(s)))
        return s.value

    def stream_destroy(self, stream):
        if stream and self._ctx.value:
            self._check(self._lib.rt_stream_destroy(self._ctx, C.c_void_p(stream)))

    def stream_forget(self, stream):
        """Before the owner of a foreign stream (e.g. a torch.cuda.Stream) destroys it: wait for it and drop the
        context's reference to its handle."""
        if stream and self._ctx.value:
            self._check(self._lib.rt_stream_forget(self._ctx, C.c_void_p(stream)))

    def timer_begin(self, stream=None):
        self._check(self._lib.rt_timer_begin(self._ctx, C.c_void_p(stream) if stream else None))

    def timer_end(self, stream=None):
        ms = C.c_float()
        self._check(self._lib.rt_timer_end(self._ctx, C.c_void_p(stream) if stream else None, C.byref(ms)))
        return ms.value

    def set_tile_stats(self, dptr):
        """Device buffer (uint32 per 8x8 tile) that later launches fill with per-tile wave cycles; None = off."""
        self._check(self._lib.rt_set_tile_stats(self._ctx, C.c_void_p(dptr) if dptr else None))

    def stats(self):
        """rt_stats as a dict: launch counters, and the ray counters of RT_FLAG_COUNT_RAYS launches."""
        st = L.rt_stats()
        self._check(self._lib.rt_get_stats(self._ctx, C.byref(st)))
        return {n: (int(getattr(st, n)) if not n.startswith("bounce_") else [int(v) for v in getattr(st, n)]) for n, _ in st._fields_}

    def reset_stats(self):
        self._check(self._lib.rt_reset_stats(self._ctx))

    def kernel_info(self):
        info = L.rt_kernel_info()
        self._check(self._lib.rt_get_kernel_info(self._ctx, C.byref(info)))
        return {n: getattr(info, n) for n, _ in info._fields_ if n != "reserved"}

    # -- raw device memory (used by the cuda facade) ------------------------------------------
    def malloc(self, nbytes):
        p = C.c_void_p()
        self._check(self._lib.rt_malloc(self._ctx, int(nbytes), C.byref(p)))
        return p.value

    def free(self, dptr):
        if dptr and self._ctx.value:
            self._check(self._lib.rt_free(self._ctx, C.c_void_p(dptr)))

    def h2d(self, dptr, host):
        self._check(self._lib.rt_memcpy_h2d(self._ctx, C.c_void_p(dptr), host.ctypes.data, host.nbytes))

    def d2h(self, host, dptr):
        self._check(self._lib.rt_memcpy_d2h(self._ctx, host.ctypes.data, C.c_void_p(dptr), host.nbytes))


def device_count():
    n = C.c_int()
    L.load().rt_device_count(C.byref(n))
    return n.value
