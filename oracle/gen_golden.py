#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — golden-vector generator (runs ONLY in the build container).

Imports the reference's own hot-path modules from /root/reference/src (read-only,
never copied) under the identity-decorator stand-in in oracle/refshim/ and runs them
on the CPU to produce the fixtures under tests/golden/.  The fixtures are data only
(inputs + expected outputs); no reference source text is written anywhere.

What is captured per frame case:
  * `u8`     : the three bytes `render` stores per pixel, in the order it stores them
               (channel 0=R, 1=B, 2=G — reference common.py:63 swaps G/B);
  * `rgb64`  : the float64 (R, G, B) tuple `render` hands to `clip_color_vector`
               (kernels.py:69), i.e. the pre-clip colour — captured by wrapping that
               name inside the reference's `kernels` module namespace at run time.
Shader scalars are passed as np.float64 so that scalar*float32 products are evaluated
in float64 as numba's typing would.

Usage:  python oracle/gen_golden.py [--only NAME ...] [--jobs 8]
"""
import argparse
import hashlib
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF_SRC = "/root/reference/src"
OUT = os.path.join(REPO, "tests", "golden")

sys.dont_write_bytecode = True


def _import_reference():
    if not os.path.isdir(REF_SRC):
        raise SystemExit("gen_golden: /root/reference is not present; this tool only runs in the build container")
    for p in (REF_SRC, os.path.join(HERE, "refshim")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import ray_tracing.kernels as kernels  # noqa
    import ray_tracing.trace as trace  # noqa
    import ray_tracing.common as common  # noqa
    import ray_tracing.intersections as inter  # noqa
    import scene as scene_mod  # noqa
    return kernels, trace, common, inter, scene_mod


# ----------------------------------------------------------------------------- scenes
RED = [255, 70, 70]
GREEN = [70, 255, 70]
BLUE = [70, 70, 255]
YELLOW = [255, 255, 70]
GREY = [125, 125, 125]
MAGENTA = [139, 0, 139]
PALETTE = [RED, BLUE, YELLOW, MAGENTA, GREEN, GREY]


def sph(lst):
    a = np.zeros((7, len(lst)), np.float32)
    for i, (o, r, c) in enumerate(lst):
        a[0:3, i] = o
        a[3, i] = r
        a[4:7, i] = c
    return a


def lig(lst):
    a = np.zeros((3, len(lst)), np.float32)
    for i, o in enumerate(lst):
        a[:, i] = o
    return a


def pla(lst):
    a = np.zeros((9, len(lst)), np.float32)
    for i, (o, n, c) in enumerate(lst):
        n = np.array(n, dtype=np.float64)
        a[0:3, i] = o
        a[3:6, i] = n / np.linalg.norm(n)
        a[6:9, i] = c
    return a


DEFAULT_LIGHTS = [[2.5, -2.0, 3.0], [2.5, 2.0, 3.0], [5.0, 0.1, 6.0]]
DEFAULT_SPHERES = [([2.2, 0.3, 1.0], 1.0, RED), ([0.6, 0.7, 0.4], 0.4, BLUE), ([0.6, -0.8, 0.5], 0.5, YELLOW),
                   ([-1.2, 0.2, 0.5], 0.5, MAGENTA), ([-1.7, -0.5, 0.3], 0.3, GREEN), ([-2.0, 1.31, 1.3], 1.3, RED)]
EXTRA_SPHERES = [([0.9, -2.0, 0.6], 0.6, GREEN), ([3.5, -1.6, 0.8], 0.8, BLUE)]
DEFAULT_PLANE = ([5, 0, 0], [0, 0, 1], GREY)


def grid_scene(n_side, seed):
    """n_side x n_side jittered grid of spheres resting on the z=0 plane."""
    rng = np.random.default_rng(seed)
    out = []
    span_x, span_y = 9.0, 8.0
    for i in range(n_side):
        for j in range(n_side):
            r = float(rng.integers(2, 9)) / 16.0 * (8.0 / n_side)
            x = -0.5 + span_x * (i + 0.5) / n_side + float(rng.uniform(-0.25, 0.25)) * (8.0 / n_side)
            y = -span_y / 2 + span_y * (j + 0.5) / n_side + float(rng.uniform(-0.25, 0.25)) * (8.0 / n_side)
            out.append(([x, y, r], r, PALETTE[(i * n_side + j) % len(PALETTE)]))
    return out


def camera_arrays(scene_mod, w, h, position, euler, fov=45.0):
    cam = scene_mod.Camera(resolution=(w, h), position=position, euler=euler, fov=fov)
    origin = np.array(cam.position, dtype=np.float64)
    return origin, np.array(cam.rotation, dtype=np.float64), cam.generate_pixel_locations()


# ----------------------------------------------------------------------------- running the reference
_W = {}


def _worker_init():
    _W["mods"] = _import_reference()


def _run_coords(job):
    """Run the reference `render` body on a list of (x,y) thread coordinates."""
    (coords, camspec, spheres, lights, planes, amb, lamb, refl, depth, aa) = job
    kernels = _W["mods"][0]
    if _W.get("camspec") != camspec:
        w_, h_, pos_, eul_, fov_ = camspec
        _W["cam"] = camera_arrays(_W["mods"][4], w_, h_, list(pos_), list(eul_), fov_)
        _W["camspec"] = camspec
    cam_o, cam_R, pixel_loc = _W["cam"]
    captured = []
    orig = kernels.clip_color_vector

    def spy(c3):
        captured.append((float(c3[0]), float(c3[1]), float(c3[2])))
        return orig(c3)

    kernels.clip_color_vector = spy
    try:
        w, h = pixel_loc.shape[1], pixel_loc.shape[2]
        result = np.zeros((3, w, h), np.uint8)
        kernels.render.over([tuple(c) for c in coords])(
            pixel_loc, result, cam_o, cam_R, spheres, lights, planes,
            np.float64(amb), np.float64(lamb), np.float64(refl), int(depth), bool(aa))
    finally:
        kernels.clip_color_vector = orig
    rgb64 = np.array(captured, dtype=np.float64).reshape(-1, 3)
    u8 = np.stack([result[:, x, y] for x, y in coords]).astype(np.uint8) if len(coords) else np.zeros((0, 3), np.uint8)
    return rgb64, u8


def _run_stochastic(job):
    """Build-defined stochastic supersampling: the reference's sample() (trace.py:115-133) on the jittered
    directions, averaged — every arithmetic step below is the reference's own function or a plain Python float op."""
    (coords, camspec, spheres, lights, planes, amb, lamb, refl, depth, spp, seed) = job
    kernels, trace, common = _W["mods"][0], _W["mods"][1], _W["mods"][2]
    try:                                   # `python oracle/gen_golden.py` puts oracle/ itself first on sys.path
        from oracle.oracle import jitter
    except ImportError:
        from oracle import jitter
    if _W.get("camspec") != camspec:
        w_, h_, pos_, eul_, fov_ = camspec
        _W["cam"] = camera_arrays(_W["mods"][4], w_, h_, list(pos_), list(eul_), fov_)
        _W["camspec"] = camspec
    cam_o, cam_R, pixel_loc = _W["cam"]
    w, h = pixel_loc.shape[1], pixel_loc.shape[2]
    ar = int(w / h)
    dy, dz = (-ar - ar) / float(w - 1), (-1 - 1) / float(h - 1)
    o = (cam_o[0], cam_o[1], cam_o[2])
    rows = (cam_R[0, :], cam_R[1, :], cam_R[2, :])
    rgb64, u8 = [], []
    for x, y in coords:
        x, y = int(x), int(y)
        P = pixel_loc[0:3, x, y]
        acc = None
        for s_ in range(spp):
            u, v = jitter(x, y, s_, seed)
            Ps = (P[0], P[1] + u * dy, P[2] + v * dz)
            d = common.normalize(common.matmul(rows, Ps))
            c = trace.sample(o, d, spheres, lights, planes, np.float64(amb), np.float64(lamb), np.float64(refl), int(depth))
            acc = c if acc is None else (acc[0] + c[0], acc[1] + c[1], acc[2] + c[2])
        R, G, B = acc[0] / spp, acc[1] / spp, acc[2] / spp
        rgb64.append((float(R), float(G), float(B)))
        u8.append(common.clip_color_vector((R, G, B)))
    return np.array(rgb64, dtype=np.float64).reshape(-1, 3), np.array(u8, dtype=np.uint8).reshape(-1, 3)


def stochastic_case(pool, jobs, scene_mod, name, w, h, spheres, lights, planes, position, euler, amb, lamb, refl, depth, spp, seed,
                    coords=None, fov=45.0):
    t0 = time.time()
    cam_o, cam_R, _ = camera_arrays(scene_mod, 4, 4, position, euler, fov)
    coords = np.asarray(all_coords(w, h) if coords is None else coords, dtype=np.int32).reshape(-1, 2)
    camspec = (w, h, tuple(float(v) for v in position), tuple(float(v) for v in euler), float(fov))
    chunks = np.array_split(coords, max(1, min(len(coords), jobs * 8)))
    res = pool.map(_run_stochastic, [(c, camspec, spheres, lights, planes, amb, lamb, refl, depth, spp, seed) for c in chunks])
    rgb64 = np.concatenate([r[0] for r in res]); u8 = np.concatenate([r[1] for r in res])
    meta = dict(w=w, h=h, spheres=spheres, lights=lights, planes=planes, cam_origin=cam_o, cam_rot=cam_R,
                position=np.array(position, dtype=np.float64), euler=np.array(euler, dtype=np.float64), fov=fov,
                amb=amb, lamb=lamb, refl=refl, depth=depth, aa=2, spp=spp, seed=seed,
                refl_pow=np.array([np.float64(refl) ** (i + 1) for i in range(max(depth, 1))], dtype=np.float64))
    save_case(name, meta, coords, rgb64, u8)
    print(f"  {name}: {time.time()-t0:.1f}s", flush=True)


def run_case(pool, jobs, coords, camspec, spheres, lights, planes, amb, lamb, refl, depth, aa):
    coords = np.asarray(coords, dtype=np.int32).reshape(-1, 2)
    nchunk = max(1, min(len(coords), jobs * 8))
    chunks = np.array_split(coords, nchunk)
    args = [(c, camspec, spheres, lights, planes, amb, lamb, refl, depth, aa) for c in chunks]
    res = pool.map(_run_coords, args)
    rgb64 = np.concatenate([r[0] for r in res]) if res else np.zeros((0, 3))
    u8 = np.concatenate([r[1] for r in res]) if res else np.zeros((0, 3), np.uint8)
    return coords, rgb64, u8


def all_coords(w, h, x1=None, y1=None):
    x1 = w if x1 is None else x1
    y1 = h if y1 is None else y1
    return [(x, y) for x in range(x1) for y in range(y1)]


def save_case(name, meta, coords, rgb64, u8, full=None, extra=None):
    d = dict(meta)
    d["coords"] = coords.astype(np.int32)
    d["rgb64"] = rgb64
    d["u8"] = u8
    if full is not None:
        d.update(full)
    if extra:
        d.update(extra)
    path = os.path.join(OUT, f"frame_{name}.npz")
    np.savez_compressed(path, **d)
    print(f"  wrote {path}  ({os.path.getsize(path)/1024:.0f} KiB, {len(coords)} px)", flush=True)


def frame_case(pool, jobs, scene_mod, name, w, h, spheres, lights, planes, position, euler, amb, lamb, refl, depth, aa,
               coords=None, fov=45.0, keep_full_u8=False, f64_stride=None):
    t0 = time.time()
    cam_o, cam_R, pixel_loc = camera_arrays(scene_mod, w, h, position, euler, fov)
    if coords is None:
        coords = all_coords(w, h, w - 1, h - 1) if aa else all_coords(w, h)
    camspec = (w, h, tuple(float(v) for v in position), tuple(float(v) for v in euler), float(fov))
    coords, rgb64, u8 = run_case(pool, jobs, coords, camspec, spheres, lights, planes,
                                 amb, lamb, refl, depth, aa)
    meta = dict(w=w, h=h, spheres=spheres, lights=lights, planes=planes, cam_origin=cam_o, cam_rot=cam_R,
                position=np.array(position, dtype=np.float64), euler=np.array(euler, dtype=np.float64), fov=fov,
                amb=amb, lamb=lamb, refl=refl, depth=depth, aa=int(aa),
                refl_pow=np.array([np.float64(refl) ** (i + 1) for i in range(max(depth, 1))], dtype=np.float64))
    extra = {}
    if keep_full_u8:
        frame = np.zeros((3, w, h), np.uint8)
        frame[:, coords[:, 0], coords[:, 1]] = u8.T
        extra["frame_u8"] = frame
        extra["sha256_u8"] = hashlib.sha256(frame.tobytes()).hexdigest()
        f64 = np.zeros((3, w, h), np.float64)
        f64[:, coords[:, 0], coords[:, 1]] = rgb64.T
        extra["sha256_rgb64"] = hashlib.sha256(f64.tobytes()).hexdigest()
        extra["sha256_rgb32"] = hashlib.sha256(f64.astype(np.float32).tobytes()).hexdigest()
    if f64_stride:
        keep = (coords[:, 0] % f64_stride == f64_stride // 2) & (coords[:, 1] % f64_stride == f64_stride // 2)
        coords, rgb64, u8 = coords[keep], rgb64[keep], u8[keep]
    save_case(name, meta, coords, rgb64, u8, extra=extra)
    print(f"  {name}: {time.time()-t0:.1f}s", flush=True)


# ----------------------------------------------------------------------------- known-answer tests
def gen_kats(mods):
    kernels, trace, common, inter, scene_mod = mods
    rng = np.random.default_rng(20261004)
    N = 3000
    out = {}
    # intersect_ray_sphere (intersections.py:6-38): f64 ray, f32 sphere
    o = rng.uniform(-4, 4, (N, 3))
    d = rng.normal(size=(N, 3))
    d[: N // 2] /= np.linalg.norm(d[: N // 2], axis=1, keepdims=True)  # half pre-normalised, half raw
    c = rng.uniform(-3, 3, (N, 3)).astype(np.float32)
    r = rng.uniform(0.1, 2.5, N).astype(np.float32)
    # force some origin-inside-sphere and tangent-ish cases
    o[:200] = c[:200].astype(np.float64) + rng.uniform(-0.05, 0.05, (200, 3))
    t = np.array([inter.intersect_ray_sphere(tuple(o[i]), tuple(d[i]), c[i], r[i]) for i in range(N)], dtype=np.float64)
    out.update(sph_o=o, sph_d=d, sph_c=c, sph_r=r, sph_t=t)
    # intersect_ray_plane (intersections.py:41-68)
    po = rng.uniform(-3, 3, (N, 3)).astype(np.float32)
    pn = rng.normal(size=(N, 3))
    pn = (pn / np.linalg.norm(pn, axis=1, keepdims=True)).astype(np.float32)
    d2 = rng.normal(size=(N, 3))
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    # near-parallel rays around the EPS=0.001 threshold
    for i in range(300):
        n64 = pn[i].astype(np.float64)
        perp = np.cross(n64, rng.normal(size=3))
        perp /= np.linalg.norm(perp)
        v = perp + n64 * rng.uniform(-0.002, 0.002)
        d2[i] = v / np.linalg.norm(v)
    tp = np.array([inter.intersect_ray_plane(tuple(o[i]), tuple(d2[i]), po[i], pn[i]) for i in range(N)], dtype=np.float64)
    out.update(pl_o=o, pl_d=d2, pl_po=po, pl_pn=pn, pl_t=tp)
    # normalize / reflection (common.py:28-32, 113-120)
    v = rng.normal(size=(N, 3)) * rng.uniform(0.01, 50, (N, 1))
    out["nrm_in"] = v
    out["nrm_out"] = np.array([common.normalize(tuple(v[i])) for i in range(N)], dtype=np.float64)
    nn = out["nrm_out"]
    out["refl_d"] = d2
    out["refl_n"] = nn
    out["refl_out"] = np.array([common.get_reflection(tuple(d2[i]), tuple(nn[i])) for i in range(N)], dtype=np.float64)
    # float32 plane normal re-normalisation (common.py:104-110)
    planes = np.zeros((9, N), np.float32)
    planes[3:6] = (rng.normal(size=(3, N)) * rng.uniform(0.2, 3, (1, N))).astype(np.float32)
    planes[3:6, :50] = pn[:50].T
    pnn = [common.get_plane_normal(i, planes) for i in range(N)]
    assert all(isinstance(x, np.float32) for x in pnn[0])
    out["pnorm_in"] = planes[3:6].T.copy()
    out["pnorm_out"] = np.array(pnn, dtype=np.float32)
    # clip_color (common.py:52-57): half-to-even, clamps
    cv = np.concatenate([np.arange(-3, 260, 0.5), rng.uniform(-10, 400, 2000), [254.5, 255.5, 0.49999999999999994, 1e9, -1e9]])
    out["clip_in"] = cv
    out["clip_out"] = np.array([common.clip_color(np.float64(x)) for x in cv], dtype=np.int64)
    # get_intersection over a random scene (trace.py:7-41), including far hits around the 999.0 limit
    S, P = 12, 3
    spheres = np.zeros((7, S), np.float32)
    spheres[0:3] = rng.uniform(-6, 6, (3, S))
    spheres[3] = rng.uniform(0.3, 1.5, S)
    spheres[0:3, 0] = [1500.0, 0, 0]
    spheres[3, 0] = 520.0  # a sphere whose near side straddles t≈980..1000 from the origin region
    planes2 = np.zeros((9, P), np.float32)
    planes2[0:3] = rng.uniform(-2, 2, (3, P))
    n3 = rng.normal(size=(3, P))
    planes2[3:6] = n3 / np.linalg.norm(n3, axis=0, keepdims=True)
    oo = rng.uniform(-2, 2, (N, 3))
    dd = rng.normal(size=(N, 3))
    dd[:400, 0] = np.abs(dd[:400, 0]) * 20 + 5  # aim a batch at the far sphere
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    gi = [trace.get_intersection(tuple(oo[i]), tuple(dd[i]), spheres, planes2) for i in range(N)]
    out.update(gi_spheres=spheres, gi_planes=planes2, gi_o=oo, gi_d=dd,
               gi_t=np.array([g[0] for g in gi], dtype=np.float64),
               gi_idx=np.array([g[1] for g in gi], dtype=np.int64),
               gi_type=np.array([g[2] for g in gi], dtype=np.int64))
    path = os.path.join(OUT, "kat_functions.npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {path} ({os.path.getsize(path)/1024:.0f} KiB)", flush=True)


def gen_host_helpers(mods):
    """Captured outputs of the host helpers either side of the path (scene.py:69-115, rotation.py:34-43,
    camera.py:18-26, viewer/image.py:7-19) — pins the build's own Scene/Camera/viewer counterparts."""
    kernels, trace, common, inter, scene_mod = mods
    out = {}
    s = scene_mod.Scene.default_scene()
    sp, li, pl = s.generate_scene()
    out.update(default_spheres=sp, default_lights=li, default_planes=pl)
    # to_array with a non-unit plane normal
    p = scene_mod.Plane([1, 2, 3], [0.3, -0.2, 1.0], [10, 20, 30]).to_array()
    out["plane_tilted"] = p
    eulers = np.array([[0, -30, 0], [0, 0, 0], [10, -20, 15], [45, 45, 45], [-5.5, 12.25, 170], [90, 0, -90]], dtype=np.float64)
    out["eulers"] = eulers
    out["rotations"] = np.array([scene_mod.euler_rotation(*e) for e in eulers])
    out["rotation_rad"] = scene_mod.euler_rotation(0.1, -0.2, 0.3, is_radians=True)
    sizes = [(128, 128), (40, 24), (24, 40), (16, 16), (2, 2), (33, 17), (1000, 1000), (1920, 1080), (3840, 2160), (7680, 4320)]
    out["pl_sizes"] = np.array(sizes)
    shas, corners = [], []
    for (w, h) in sizes:
        cam = scene_mod.Camera((w, h), [-2, 0, 2.0], [0, -30, 0])
        pl_ = cam.generate_pixel_locations()
        assert pl_.dtype == np.float64 and pl_.shape == (3, w, h)
        shas.append(hashlib.sha256(pl_.tobytes()).hexdigest())
        corners.append([pl_[0, 0, 0], pl_[1, 0, 0], pl_[1, w - 1, 0], pl_[2, 0, 0], pl_[2, 0, h - 1], pl_[1, 1, 0], pl_[2, 0, 1]])
        if w * h <= 40 * 40:
            out[f"pixel_loc_{w}x{h}"] = pl_
    out["pl_sha256"] = np.array(shas)
    out["pl_corners"] = np.array(corners)
    cam = scene_mod.Camera((16, 16), [-2, 0, 2.0], [0, -30, 0], fov=60.0)
    out["pixel_loc_16x16_fov60"] = cam.generate_pixel_locations()
    out["cam_pos_float"] = scene_mod.Camera((4, 4), [-2, 0, 2.0], [0, -30, 0]).position
    # viewer (square frames only: the reference's non-square handling crops)
    try:
        import viewer
        rng = np.random.default_rng(7)
        x = rng.integers(0, 256, (3, 8, 8)).astype(np.uint8)
        im = viewer.convert_array_to_image(x)
        out["viewer_in_8"] = x
        out["viewer_out_8"] = np.asarray(im)
    except Exception as e:  # Pillow API drift is not part of the hot path
        print("  viewer fixture skipped:", repr(e))
    path = os.path.join(OUT, "host_helpers.npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {path} ({os.path.getsize(path)/1024:.0f} KiB)", flush=True)


# ----------------------------------------------------------------------------- closest-hit tie rule
def _norm3(v):
    n = np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
    return np.stack([v[0] / n, v[1] / n, v[2] / n])


def _numerators(o, R, a, spheres):
    """Per sphere, the numerator n = -b/2 -/+ sqrt(disc/4) and the reference's distance t = n / a (the quadratic
    of intersections.py:19-36 divided by 4, which is exact), for rays (o[:, j], R) — plain float64 numpy, no FMA."""
    N, T = [], []
    for k in range(spheres.shape[1]):
        c = spheres[0:3, k].astype(np.float64)
        r2 = np.float64(np.float32(spheres[3, k]) * np.float32(spheres[3, k]))
        L0, L1, L2 = o[0] - c[0], o[1] - c[1], o[2] - c[2]
        s = (L0 * R[0] + L1 * R[1]) + L2 * R[2]
        cc = ((L0 * L0 + L1 * L1) + L2 * L2) - r2
        D = s * s - a * cc
        ok = D >= 0
        q = np.sqrt(np.where(ok, D, 0))
        n = -s - q
        n = np.where(n > 0, n, -s + q)
        hit = ok & (n > 0)
        N.append(np.where(hit, n, np.inf))
        T.append(np.where(hit, n / a, np.inf))
    return np.array(N), np.array(T)


def find_tie_cases(ncases=6, K=16, side=8):
    """Scenes in which two spheres have DIFFERENT numerators that round to the SAME distance t, so that the
    reference's rule (smallest t, then lowest index: trace.py:26) and "smallest numerator" disagree.

    Construction: K concentric spheres of radius 4 around the world origin whose centres differ by a fraction of an
    ulp-sized step in y (sphere 0 has the largest offset), the camera a few 2^-52 from the centre: every ray leaves
    through the far side with numerator n = q - s within a few ulp of 4.  For a ray with a = R.R = 1 - 2^-52 the
    numerators 4 - 1ulp and 4 - 2ulp both give t = 4.0; with R.y > 0 the lower index has the larger numerator.
    The pixel grid is explicit (rt_set_pixel_loc): `side`^2 primary-ray targets picked from a 1024x1024 camera grid
    for having a <= 1 - 2^-52, and the camera offset is scanned for the position with the most disagreeing pixels."""
    w = h = 1024
    px = float(1 / np.tan(np.radians(45.0) / 2))
    X, Y = np.meshgrid(np.arange(w), np.arange(h), indexing="ij")
    P = np.stack([np.full((w, h), px), X * ((-1 - 1) / float(w - 1)) + 1, Y * ((-1 - 1) / float(h - 1)) + 1])
    d = _norm3(P)                        # camera rotation = identity: matmul(I, P) = P exactly
    R = _norm3(d)
    a = (R[0] * R[0] + R[1] * R[1]) + R[2] * R[2]
    sel = np.argwhere((a <= 1 - 2.0 ** -52) & (R[1] > 0.05))
    rng = np.random.default_rng(2026)
    spheres = np.zeros((7, K), np.float32)
    spheres[3] = 4.0
    spheres[1] = (np.arange(K)[::-1] - K // 2) * 2e-16
    spheres[4] = 5 + 10 * np.arange(K)
    spheres[5] = 250 - 10 * np.arange(K)
    spheres[6] = 37 + 3 * np.arange(K)
    cases = []
    for attempt in range(200):
        pick = sel[rng.choice(len(sel), side * side, replace=False)]
        Pg = P[:, pick[:, 0], pick[:, 1]]                                   # (3, side*side)
        Rg, ag = R[:, pick[:, 0], pick[:, 1]], a[pick[:, 0], pick[:, 1]]
        J = np.arange(-40, 41)
        best = None
        for i in range(side * side):                                         # camera offsets along ray i
            oj = (J * 2.0 ** -52)[None, :] * Rg[:, i][:, None]
            bad = np.zeros(len(J), int)
            for m in range(side * side):
                N, T = _numerators(oj, Rg[:, m], ag[m], spheres)
                bad += (np.argmin(T, axis=0) != np.argmin(N, axis=0)) & np.isfinite(T.min(axis=0))
            j = int(np.argmax(bad))
            if best is None or bad[j] > best[0]:
                best = (int(bad[j]), oj[:, j].copy())
        if best[0] >= 1:
            cases.append(dict(pixel_loc=Pg.reshape(3, side, side).copy(), cam_origin=best[1], spheres=spheres.copy(),
                              disagreeing=best[0]))
        if len(cases) == ncases:
            break
    return cases


def gen_tie_cases(mods):
    """tests/golden/tie_break.npz: the reference's own render() on the scenes of find_tie_cases()."""
    kernels = mods[0]
    cases = find_tie_cases()
    assert cases, "no tie case found"
    out = dict(n=len(cases))
    for ci, c in enumerate(cases):
        side = c["pixel_loc"].shape[1]
        captured = []
        orig = kernels.clip_color_vector

        def spy(c3):
            captured.append((float(c3[0]), float(c3[1]), float(c3[2])))
            return orig(c3)
        kernels.clip_color_vector = spy
        try:
            result = np.zeros((3, side, side), np.uint8)
            coords = [(x, y) for x in range(side) for y in range(side)]
            kernels.render.over(coords)(c["pixel_loc"], result, c["cam_origin"], np.eye(3), c["spheres"],
                                         np.zeros((3, 0), np.float32), np.zeros((9, 0), np.float32),
                                         np.float64(1.0), np.float64(0.0), np.float64(0.0), 0, False)
        finally:
            kernels.clip_color_vector = orig
        out[f"pixel_loc_{ci}"] = c["pixel_loc"]
        out[f"cam_origin_{ci}"] = c["cam_origin"]
        out[f"spheres_{ci}"] = c["spheres"]
        out[f"u8_{ci}"] = result
        out[f"rgb64_{ci}"] = np.array(captured, dtype=np.float64).reshape(side, side, 3).transpose(2, 0, 1)
        out[f"disagreeing_{ci}"] = c["disagreeing"]
    path = os.path.join(OUT, "tie_break.npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {path} ({os.path.getsize(path)/1024:.0f} KiB, {len(cases)} cases, "
          f"{[c['disagreeing'] for c in cases]} disagreeing pixels)", flush=True)


# ----------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--jobs", type=int, default=8)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    mods = _import_reference()
    scene_mod = mods[4]

    def want(n):
        return a.only is None or n in a.only

    if want("kat"):
        gen_kats(mods)
    if want("host"):
        gen_host_helpers(mods)
    if want("tie"):
        gen_tie_cases(mods)

    L3 = lig(DEFAULT_LIGHTS)
    P1 = pla([DEFAULT_PLANE])
    S6 = sph(DEFAULT_SPHERES)
    S8 = sph(DEFAULT_SPHERES + EXTRA_SPHERES)
    CAM = ([-2, 0, 2.0], [0, -30, 0])

    with mp.Pool(a.jobs, initializer=_worker_init) as pool:
        fc = lambda *x, **k: frame_case(pool, a.jobs, scene_mod, *x, **k)  # noqa: E731
        if want("c1"):  # BASELINE config 1
            fc("c1_128", 128, 128, sph(DEFAULT_SPHERES[:3]), L3, P1, *CAM, 0.0, 0.6, 0.3, 1, False, keep_full_u8=True)
        if want("default_d3"):
            fc("default_128_d3", 128, 128, S6, L3, P1, *CAM, 0.0, 0.6, 0.3, 3, False, keep_full_u8=True)
        if want("aa"):  # reference 9-tap AA incl. its G/B accumulation quirk; last row/col excluded (undefined there)
            fc("aa_48_d2", 48, 48, S6, L3, P1, *CAM, 0.0, 0.6, 0.3, 2, True)
            fc("aa_40x24_d1", 40, 24, S8, L3, P1, *CAM, 0.05, 0.6, 0.3, 1, True)
        if want("nonsquare"):
            fc("nonsquare_40x24", 40, 24, S6, L3, P1, *CAM, 0.0, 0.6, 0.3, 2, False)
            fc("portrait_24x40", 24, 40, S6, L3, P1, *CAM, 0.0, 0.6, 0.3, 2, False)
            fc("odd_37x29", 37, 29, S8, L3, P1, *CAM, 0.0, 0.6, 0.3, 3, False)
        if want("edge"):
            fc("spheres_only_32", 32, 32, S6, L3, np.zeros((9, 0), np.float32), *CAM, 0.0, 0.6, 0.3, 3, False)
            P2 = pla([DEFAULT_PLANE, ([6.0, 0, 0], [-1.0, 0.2, 0.3], BLUE)])
            fc("planes_only_32", 32, 32, np.zeros((7, 0), np.float32), L3, P2, *CAM, 0.0, 0.6, 0.3, 3, False)
            fc("tilted_planes_48", 48, 48, S6, L3, P2, [-3, 1, 1.5], [10, -20, 15], 0.15, 0.5, 0.45, 4, False)
            Sin = sph([([-2, 0, 2.0], 4.0, GREEN)] + DEFAULT_SPHERES[:3])
            fc("inside_sphere_32", 32, 32, Sin, L3, P1, *CAM, 0.1, 0.6, 0.3, 2, False)
            fc("nolights_24", 24, 24, S6, np.zeros((3, 0), np.float32), P1, *CAM, 0.2, 0.6, 0.3, 2, False)
            fc("onelight_depth0_32", 32, 32, S6, lig([[0.0, 0.0, 8.0]]), P1, *CAM, 0.0, 0.9, 0.3, 0, False)
            fc("fivelights_32", 32, 32, S8, lig(DEFAULT_LIGHTS + [[-3, -3, 0.5], [0.6, 0.7, 0.4]]), P1, *CAM, 0.0, 0.3, 0.5, 2, False)
            fc("horizon_64", 64, 64, S6, L3, P1, [0, 0, 8.0], [0, 0, 0], 0.0, 0.6, 0.3, 1, False)
            fc("fov70_48", 48, 48, S8, L3, P1, [-2.5, 0.4, 1.2], [5, -12, -8], 0.02, 0.6, 0.6, 6, False, fov=70.0)
        if want("c4"):  # 64 spheres, depth 5, 3840x2160 sampled every 32nd pixel
            S64 = sph(grid_scene(8, 355))
            cs = [(x, y) for x in range(16, 3840, 32) for y in range(16, 2160, 32)]
            fc("c4_s64_d5_sub32", 3840, 2160, S64, L3, P1, *CAM, 0.0, 0.6, 0.3, 5, False, coords=cs)
        if want("c5"):  # 256 spheres, depth 8, 7680x4320 sampled every 96th pixel (1 spp; the 4-spp mode is build-defined)
            S256 = sph(grid_scene(16, 356))
            cs = [(x, y) for x in range(48, 7680, 96) for y in range(48, 4320, 96)]
            fc("c5_s256_d8_sub96", 7680, 4320, S256, L3, P1, *CAM, 0.0, 0.6, 0.3, 8, False, coords=cs)
        if want("stochastic"):  # build-defined stochastic supersampling; golden = the reference's sample() per jittered ray
            sc = lambda *x, **k: stochastic_case(pool, a.jobs, scene_mod, *x, **k)  # noqa: E731
            sc("stoch_48_spp4", 48, 48, S6, L3, P1, *CAM, 0.0, 0.6, 0.3, 2, 4, 1)
            sc("stoch_40x24_spp3_seed7", 40, 24, S8, L3, P1, *CAM, 0.05, 0.6, 0.3, 1, 3, 7)
            S256 = sph(grid_scene(16, 356))   # BASELINE config 5 with its 4 spp, on the same lattice as c5_s256_d8_sub96
            cs = [(x, y) for x in range(48, 7680, 96) for y in range(48, 4320, 96)]
            sc("c5_s256_d8_spp4_sub96", 7680, 4320, S256, L3, P1, *CAM, 0.0, 0.6, 0.3, 8, 4, 1, coords=cs)
        if want("c2"):  # BASELINE config 2, the headline frame: full reference frame
            fc("c2_1080p", 1920, 1080, S8, L3, P1, *CAM, 0.0, 0.6, 0.3, 3, False, keep_full_u8=True, f64_stride=8)


if __name__ == "__main__":
    main()
