"""TEST INFRASTRUCTURE ONLY — ctypes binding of oracle/librt_oracle.so (the CPU restatement).

Importers: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package never imports this module (tests/test_boundary.py checks that).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "librt_oracle.so")

FLAG_TYPED_BIAS = 1

_lib = None


def build(force=False):
    src = os.path.join(HERE, "rt_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "librt_oracle.so"], stdout=subprocess.DEVNULL)
    return SO


class _RayGen(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("pixel_loc", C.c_void_p),
                ("px", C.c_double), ("y0", C.c_double), ("dy", C.c_double), ("z0", C.c_double), ("dz", C.c_double)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        L = C.CDLL(SO)
        dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
        L.orc_normalize.argtypes = [dp, dp]
        L.orc_intersect_ray_sphere.argtypes = [dp, dp, fp, C.c_float]
        L.orc_intersect_ray_sphere.restype = C.c_double
        L.orc_intersect_ray_plane.argtypes = [dp, dp, fp, fp]
        L.orc_intersect_ray_plane.restype = C.c_double
        L.orc_get_intersection.argtypes = [dp, dp, fp, C.c_int, fp, C.c_int, dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_get_reflection.argtypes = [dp, dp, dp]
        L.orc_plane_normal_f32.argtypes = [fp, fp]
        L.orc_clip_color.argtypes = [C.c_double]
        L.orc_clip_color.restype = C.c_int
        L.orc_sample.argtypes = [fp, C.c_int, fp, C.c_int, fp, C.c_int, dp, dp, C.c_double, C.c_double, dp, C.c_int, C.c_int, dp]
        L.orc_render.argtypes = [C.POINTER(_RayGen), dp, dp, fp, C.c_int, fp, C.c_int, fp, C.c_int,
                                 C.c_double, C.c_double, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_longlong), C.c_int, C.c_uint32]
        L.orc_render.restype = C.c_int
        L.orc_render_pixels.argtypes = [C.POINTER(_RayGen), dp, dp, fp, C.c_int, fp, C.c_int, fp, C.c_int,
                                        C.c_double, C.c_double, dp, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32]
        L.orc_render_pixels.restype = C.c_int
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def normalize(v):
    v = _d(v); out = np.empty(3)
    lib().orc_normalize(_dp(v), _dp(out))
    return out


def intersect_ray_sphere(o, d, c, r):
    o, d, c = _d(o), _d(d), _f(c)
    return lib().orc_intersect_ray_sphere(_dp(o), _dp(d), _fp(c), C.c_float(float(np.float32(r))))


def intersect_ray_plane(o, d, po, pn):
    o, d, po, pn = _d(o), _d(d), _f(po), _f(pn)
    return lib().orc_intersect_ray_plane(_dp(o), _dp(d), _fp(po), _fp(pn))


def get_intersection(o, d, spheres, planes):
    o, d, spheres, planes = _d(o), _d(d), _f(spheres), _f(planes)
    t = C.c_double(); i = C.c_int(); ty = C.c_int()
    lib().orc_get_intersection(_dp(o), _dp(d), _fp(spheres), spheres.shape[1], _fp(planes), planes.shape[1],
                               C.byref(t), C.byref(i), C.byref(ty))
    return t.value, i.value, ty.value


def get_reflection(d, n):
    d, n = _d(d), _d(n); out = np.empty(3)
    lib().orc_get_reflection(_dp(d), _dp(n), _dp(out))
    return out


def plane_normal_f32(n):
    n = _f(n); out = np.empty(3, np.float32)
    lib().orc_plane_normal_f32(_fp(n), _fp(out))
    return out


def clip_color(c):
    return lib().orc_clip_color(float(c))


def _aa_code(aa, spp):
    """0 none, 1 the reference's 9-tap mode, 0x100|spp stochastic supersampling (aa == 2)."""
    if int(aa) == 2:
        assert 1 <= int(spp) <= 64
        return 0x100 | int(spp)
    return int(bool(aa))


def jitter(x, y, s, seed):
    """(u, v) in [-1/2, 1/2)^2 of sample s of pixel (x, y): the counter hash of rt_oracle.c:jitter_hash."""
    M = 0xFFFFFFFF
    h = (seed ^ 0x9E3779B9) & M
    h = ((h ^ x) * 0x85EBCA6B) & M; h ^= h >> 13
    h = ((h ^ y) * 0xC2B2AE35) & M; h ^= h >> 16
    h = ((h ^ s) * 0x27D4EB2F) & M; h ^= h >> 15
    h = (h * 0x165667B1) & M; h ^= h >> 13
    return (h & 0xFFFF) * 2.0 ** -16 + (2.0 ** -17 - 0.5), (h >> 16) * 2.0 ** -16 + (2.0 ** -17 - 0.5)


def refl_powers(refl, depth):
    """refl ** (i+1) as the reference evaluates it (trace.py:131)."""
    return np.array([float(refl) ** (i + 1) for i in range(max(int(depth), 1))], dtype=np.float64)


def sample(spheres, lights, planes, o, d, amb, lamb, refl, depth, flags=0):
    spheres, lights, planes, o, d = _f(spheres), _f(lights), _f(planes), _d(o), _d(d)
    rp = refl_powers(refl, depth); out = np.empty(3)
    lib().orc_sample(_fp(spheres), spheres.shape[1], _fp(lights), lights.shape[1], _fp(planes), planes.shape[1],
                     _dp(o), _dp(d), float(amb), float(lamb), _dp(rp), int(depth), int(flags), _dp(out))
    return out


def render(w, h, cam_origin, cam_rot, spheres, lights, planes, amb, lamb, refl, depth, aa=False, *,
           pixel_loc=None, raygen=None, x0=0, x1=None, flags=0, want=("u8", "f64"), nthreads=0, refl_pow=None,
           spp=0, seed=1):
    """Run the restated `render` (kernels.py:6-73) for columns [x0,x1).

    raygen = (px, y0, dy, z0, dz) closed form, or pixel_loc = explicit float64 (3,w,h) array.
    Returns dict with any of 'u8' (3,w,h) uint8 [R,B,G], 'f64' (3,w,h), 'f32' (3,w,h) and 'counters'.
    Columns outside [x0,x1) are left zero.
    """
    L = lib()
    x1 = w if x1 is None else x1
    rg = _RayGen()
    rg.w, rg.h = int(w), int(h)
    keep = None
    if pixel_loc is not None:
        keep = _d(pixel_loc)
        assert keep.shape == (3, w, h)
        rg.pixel_loc = keep.ctypes.data
    else:
        rg.pixel_loc = None
        rg.px, rg.y0, rg.dy, rg.z0, rg.dz = [float(v) for v in raygen]
    spheres, lights, planes = _f(spheres), _f(lights), _f(planes)
    o, R = _d(cam_origin), _d(cam_rot).reshape(9)
    rp = _d(refl_pow) if refl_pow is not None else refl_powers(refl, depth)
    out = {}
    u8 = np.zeros((3, w, h), np.uint8) if "u8" in want else None
    f64 = np.zeros((3, w, h), np.float64) if "f64" in want else None
    f32 = np.zeros((3, w, h), np.float32) if "f32" in want else None
    cnt = (C.c_longlong * 3)()
    rc = L.orc_render(C.byref(rg), _dp(o), _dp(R), _fp(spheres), spheres.shape[1], _fp(lights), lights.shape[1],
                      _fp(planes), planes.shape[1], float(amb), float(lamb), _dp(rp), int(depth), _aa_code(aa, spp), int(flags),
                      int(x0), int(x1),
                      u8.ctypes.data if u8 is not None else None,
                      f64.ctypes.data if f64 is not None else None,
                      f32.ctypes.data if f32 is not None else None, cnt, int(nthreads), int(seed) & 0xFFFFFFFF)
    if rc != 0:
        raise ValueError("orc_render: bad arguments")
    if u8 is not None: out["u8"] = u8
    if f64 is not None: out["f64"] = f64
    if f32 is not None: out["f32"] = f32
    out["counters"] = dict(closest=cnt[0], shadow=cnt[1], hits=cnt[2])
    return out


def render_pixels(w, h, coords, cam_origin, cam_rot, spheres, lights, planes, amb, lamb, refl, depth, aa=False, *,
                  pixel_loc=None, raygen=None, flags=0, nthreads=0, refl_pow=None, spp=0, seed=1):
    """The same per-pixel path for an explicit (n,2) list of (x,y) pixels.
    Returns (u8 (n,3) in stored order [R,B,G], f64 (n,3) = pre-clip (R,G,B))."""
    L = lib()
    rg = _RayGen()
    rg.w, rg.h = int(w), int(h)
    keep = None
    if pixel_loc is not None:
        keep = _d(pixel_loc)
        assert keep.shape == (3, w, h)
        rg.pixel_loc = keep.ctypes.data
    else:
        rg.pixel_loc = None
        rg.px, rg.y0, rg.dy, rg.z0, rg.dz = [float(v) for v in raygen]
    spheres, lights, planes = _f(spheres), _f(lights), _f(planes)
    o, R = _d(cam_origin), _d(cam_rot).reshape(9)
    rp = _d(refl_pow) if refl_pow is not None else refl_powers(refl, depth)
    co = np.ascontiguousarray(coords, dtype=np.int32).reshape(-1, 2)
    n = co.shape[0]
    u8 = np.zeros((n, 3), np.uint8)
    f64 = np.zeros((n, 3), np.float64)
    rc = L.orc_render_pixels(C.byref(rg), _dp(o), _dp(R), _fp(spheres), spheres.shape[1], _fp(lights), lights.shape[1],
                             _fp(planes), planes.shape[1], float(amb), float(lamb), _dp(rp), int(depth), _aa_code(aa, spp),
                             int(flags), co.ctypes.data, n, u8.ctypes.data, f64.ctypes.data, int(nthreads), int(seed) & 0xFFFFFFFF)
    if rc != 0:
        raise ValueError("orc_render_pixels: bad arguments")
    return u8, f64


def max_threads():
    return lib().orc_max_threads()
