"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes -> libmi355rt.so),
against (1) the committed golden vectors produced by the reference's own Python and (2) the CPU
oracle on the same inputs.  Bars: uint8 frame bit-exact; float32 pre-clip colour within 1e-5 abs of
the reference's float64 value on the displayable range [0,255] (north_star tolerance) — and, since
the kernel evaluates the same IEEE float64 operations, bit-equal to the oracle's float32 rounding."""
import os

import numpy as np
import pytest

from conftest import frame_cases, load_frame, raygen_closed_form, clamp255

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _setup(r, g, explicit_grid=False):
    w, h = int(g["w"]), int(g["h"])
    r.set_scene(g["spheres"], g["lights"], g["planes"])
    r.set_camera(g["cam_origin"], g["cam_rot"])
    rg = raygen_closed_form(w, h, float(g["fov"]))
    if explicit_grid:
        px, y0, dy, z0, dz = rg
        grid = np.empty((3, w, h))
        grid[0] = px
        grid[1] = (np.arange(w) * dy + y0)[:, None]
        grid[2] = (np.arange(h) * dz + z0)[None, :]
        r.set_pixel_loc(grid)
    else:
        r.set_raygen(w, h, *rg)
    return w, h, rg


def _render(r, g, **kw):
    return r.render(float(g["amb"]), float(g["lamb"]), float(g["refl"]), int(g["depth"]), int(g["aa"]),
                    u8=True, f32=True, refl_pow=g["refl_pow"], spp=int(g["spp"]) if "spp" in g else 0,
                    seed=int(g["seed"]) if "seed" in g else 1, **kw)


SMALL = [c for c in frame_cases() if not c.startswith(("c2_", "c4_", "c5_"))]


@pytest.mark.parametrize("case", SMALL)
def test_frame_vs_golden(renderer, case):
    g = load_frame(case)
    _setup(renderer, g)
    u8, f32 = _render(renderer, g)
    co = g["coords"]
    got8 = u8[:, co[:, 0], co[:, 1]].T
    got32 = f32[:, co[:, 0], co[:, 1]].T.astype(np.float64)
    assert np.array_equal(got8, g["u8"]), f"{(got8 != g['u8']).any(axis=1).sum()} of {len(co)} pixels differ (uint8)"
    err = np.abs(clamp255(got32) - clamp255(g["rgb64"])).max()
    assert err <= TOL, f"max abs err {err:.3e} > {TOL}"
    # same float64 arithmetic => the float32 rounding of the reference's float64 value, exactly
    assert np.array_equal(f32[:, co[:, 0], co[:, 1]].T, g["rgb64"].astype(np.float32))


@pytest.mark.parametrize("case", ["c4_s64_d5_sub32", "c5_s256_d8_sub96", "c5_s256_d8_spp4_sub96"])
def test_large_configs_sampled(renderer, case):
    """BASELINE configs 4 and 5 (1 spp) at full resolution; golden = the reference on a pixel lattice."""
    g = load_frame(case)
    _setup(renderer, g)
    u8, f32 = _render(renderer, g)
    co = g["coords"]
    assert np.array_equal(u8[:, co[:, 0], co[:, 1]].T, g["u8"])
    assert np.array_equal(f32[:, co[:, 0], co[:, 1]].T, g["rgb64"].astype(np.float32))


def test_c2_headline_frame_full(renderer, oracle):
    """BASELINE config 2 (1920x1080, 8 spheres + plane, depth 3): every one of the 2 073 600 pixels
    against the full uint8 frame the reference produced, plus the float lattice."""
    import hashlib
    g = load_frame("c2_1080p")
    w, h, rg = _setup(renderer, g)
    u8, f32 = _render(renderer, g)
    assert np.array_equal(u8, g["frame_u8"]), f"{(u8 != g['frame_u8']).any(axis=0).sum()} pixels differ"
    assert hashlib.sha256(u8.tobytes()).hexdigest() == str(g["sha256_u8"])
    assert hashlib.sha256(f32.tobytes()).hexdigest() == str(g["sha256_rgb32"])
    co = g["coords"]
    err = np.abs(clamp255(f32[:, co[:, 0], co[:, 1]].T.astype(np.float64)) - clamp255(g["rgb64"])).max()
    assert err <= TOL


def test_closest_hit_tie_rule(renderer):
    """Known-answer test for the reference's tie rule (trace.py:26): in these scenes ~60 of 64 pixels see two spheres
    whose numerators differ while their distances t = n/a round to the same double; the reference keeps the lower
    index.  Golden = the reference's own render() (tests/golden/tie_break.npz)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tie_break.npz"))
    for ci in range(int(g["n"])):
        pl = g[f"pixel_loc_{ci}"]
        renderer.set_scene(g[f"spheres_{ci}"], np.zeros((3, 0), np.float32), np.zeros((9, 0), np.float32))
        renderer.set_camera(g[f"cam_origin_{ci}"], np.eye(3))
        renderer.set_pixel_loc(pl)
        u8, f32 = renderer.render(1.0, 0.0, 0.0, 0, 0, u8=True, f32=True)
        assert np.array_equal(u8, g[f"u8_{ci}"]), f"case {ci}: {(u8 != g[f'u8_{ci}']).any(axis=0).sum()} px differ"
        assert np.array_equal(f32, g[f"rgb64_{ci}"].astype(np.float32))


def test_explicit_pixel_loc_equals_raygen(renderer):
    g = load_frame("odd_37x29")
    _setup(renderer, g, explicit_grid=False)
    a8, a32 = _render(renderer, g)
    _setup(renderer, g, explicit_grid=True)
    b8, b32 = _render(renderer, g)
    assert np.array_equal(a8, b8) and np.array_equal(a32, b32)


def test_explicit_pixel_loc_aa(renderer):
    g = load_frame("aa_40x24_d1")
    _setup(renderer, g, explicit_grid=True)
    u8, f32 = _render(renderer, g)
    co = g["coords"]
    assert np.array_equal(u8[:, co[:, 0], co[:, 1]].T, g["u8"])


@pytest.mark.parametrize("case", ["aa_48_d2", "aa_40x24_d1"])
def test_aa_lattice_and_per_pixel_paths_agree_with_golden(renderer, case):
    """RT_AA_REFERENCE traces every half-pixel lattice sample once and sums nine per pixel (default on the closed-form
    grid); RT_FLAG_AA_PER_PIXEL traces the reference's nine taps per pixel.  Both must be the reference's bytes."""
    from python_ray_tracer_amd import _lib as L
    g = load_frame(case)
    _setup(renderer, g)
    co = g["coords"]
    for flags in (0, L.RT_FLAG_AA_PER_PIXEL):
        u8, f32 = _render(renderer, g, flags=flags)
        assert np.array_equal(u8[:, co[:, 0], co[:, 1]].T, g["u8"]), flags
        assert np.array_equal(f32[:, co[:, 0], co[:, 1]].T, g["rgb64"].astype(np.float32)), flags
    a8, a32 = _render(renderer, g)
    b8, b32 = _render(renderer, g, flags=L.RT_FLAG_AA_PER_PIXEL)
    assert np.array_equal(a8, b8) and np.array_equal(a32, b32)          # incl. the frame border (one tap)


@pytest.mark.parametrize("w,h", [(3, 3), (2, 5), (1, 1), (9, 4), (67, 35), (130, 77)])
def test_aa_lattice_odd_sizes_vs_oracle(renderer, oracle, w, h):
    """Frame sizes around the tile size and below it (no interior pixels at all for w or h < 3), slabs that start
    and end on odd columns, and frames of a sequence in flight on several streams (one lattice buffer per stream)."""
    from python_ray_tracer_amd.scene import Camera
    g = load_frame("default_128_d3")
    cam = Camera((w, h), [-2, 0.1, 2.0], [3, -28, 4], fov=50.0)
    rg = cam.raygen()
    renderer.set_scene(g["spheres"], g["lights"], g["planes"]); renderer.set_camera(cam.position, cam.rotation); renderer.set_raygen(w, h, *rg)
    ref = oracle.render(w, h, cam.position, cam.rotation, g["spheres"], g["lights"], g["planes"], 0.05, 0.6, 0.4, 2, True, raygen=rg, want=("u8", "f32"))
    u8, f32 = renderer.render(0.05, 0.6, 0.4, 2, 1, u8=True, f32=True)
    assert np.array_equal(u8, ref["u8"]) and np.array_equal(f32, ref["f32"])
    if w >= 9:
        for a, b in ((1, w - 2), (3, 4), (w // 2, w)):
            p8, p32 = renderer.render(0.05, 0.6, 0.4, 2, 1, u8=True, f32=True, x0=a, x1=b)
            assert np.array_equal(p8, ref["u8"][:, a:b]) and np.array_equal(p32, ref["f32"][:, a:b])
    if w == 130:
        p = renderer.params(0.05, 0.6, 0.4, 2, 1)
        streams = [renderer.stream_create() for _ in range(3)]
        bufs = [renderer.malloc(3 * w * h) for _ in range(9)]
        try:
            for i, b in enumerate(bufs):
                renderer.render_device(p, 0, w, b, None, w * h, stream=streams[i % 3])
            for s_ in streams:
                renderer.sync(s_)
            for b in bufs:
                got = np.empty((3, w, h), np.uint8); renderer.d2h(got, b)
                assert np.array_equal(got, ref["u8"])
        finally:
            for s_ in streams:
                renderer.stream_destroy(s_)
            for b in bufs:
                renderer.free(b)


@pytest.mark.parametrize("nranks", [2, 3, 4, 8])
def test_column_slabs_assemble_bit_identical(renderer, nranks):
    """Row-tiling used for multi-GPU: N column slabs rendered separately == the single-launch frame."""
    from python_ray_tracer_amd.distributed import slab_bounds
    g = load_frame("default_128_d3")
    w, h, _ = _setup(renderer, g)
    full8, full32 = _render(renderer, g)
    parts = [_render(renderer, g, x0=a, x1=b) for a, b in (slab_bounds(w, nranks, r) for r in range(nranks))]
    assert np.array_equal(np.concatenate([p[0] for p in parts], axis=1), full8)
    assert np.array_equal(np.concatenate([p[1] for p in parts], axis=1), full32)


def test_c2_eight_slabs_in_place_full_frame(renderer):
    """BASELINE config 3's geometry on one GPU: the 1920x1080 frame as eight 240-column slabs, each rendered IN
    PLACE into one device frame (base = column x0, plane_stride = w*h), must be the reference frame; so must the
    weighted (unequal) slab boundaries of distributed.weighted_slab_bounds."""
    import hashlib
    from python_ray_tracer_amd.distributed import slab_bounds, weighted_slab_bounds
    g = load_frame("c2_1080p")
    w, h, _ = _setup(renderer, g)
    p = renderer.params(float(g["amb"]), float(g["lamb"]), float(g["refl"]), int(g["depth"]), 0, refl_pow=g["refl_pow"])
    d8, d32 = renderer.malloc(3 * w * h), renderer.malloc(12 * w * h)
    try:
        cost = np.linspace(1.0, 3.0, (w + 7) // 8)            # any positive weights: the frame must not depend on them
        for bounds in ([slab_bounds(w, 8, r) for r in range(8)], weighted_slab_bounds(cost, w, 8), weighted_slab_bounds(cost, w, 3)):
            assert bounds[0][0] == 0 and bounds[-1][1] == w
            renderer.h2d(d8, np.zeros((3, w, h), np.uint8)); renderer.h2d(d32, np.zeros((3, w, h), np.float32))
            for (a, b) in bounds:
                renderer.render_device(p, a, b, d8 + a * h, d32 + 4 * a * h, w * h)
            renderer.sync()
            u8, f32 = np.empty((3, w, h), np.uint8), np.empty((3, w, h), np.float32)
            renderer.d2h(u8, d8); renderer.d2h(f32, d32)
            assert np.array_equal(u8, g["frame_u8"]), f"{(u8 != g['frame_u8']).any(axis=0).sum()} pixels differ"
            assert hashlib.sha256(f32.tobytes()).hexdigest() == str(g["sha256_rgb32"])
    finally:
        renderer.free(d8); renderer.free(d32)


@pytest.mark.parametrize("lanes_mins", [None, "30"])
def test_random_scenes_every_cull_variant(monkeypatch, oracle, lanes_mins):
    """Flat scenes (two- and four-wave workgroups), clustered scenes under the wave-uniform cull and under the lane-owned
    traversal (default from 161 spheres; MI355RT_LANES_MINS=30 sends every clustered scene of this test through it): config 4's
    golden lattice, and random scenes of 12..190 spheres against the oracle, with and without the 9-tap mode."""
    import python_ray_tracer_amd as pkg
    if lanes_mins:
        monkeypatch.setenv("MI355RT_LANES_MINS", lanes_mins)
    r = pkg.Renderer(0)
    try:
        g = load_frame("c4_s64_d5_sub32")
        _setup(r, g)
        u8, f32 = _render(r, g)
        co = g["coords"]
        assert np.array_equal(u8[:, co[:, 0], co[:, 1]].T, g["u8"])
        assert np.array_equal(f32[:, co[:, 0], co[:, 1]].T, g["rgb64"].astype(np.float32))
        rng = np.random.default_rng(77)
        from python_ray_tracer_amd.scene import Camera
        for trial, S in enumerate((12, 40, 97, 150, 190)):
            w, h = 72, 56
            sp = np.zeros((7, S), np.float32)
            sp[0:3] = rng.uniform(-4, 4, (3, S)); sp[2] = np.abs(sp[2]) * 0.4 + 0.1
            sp[3] = rng.uniform(0.1, 0.5, S); sp[4:7] = rng.uniform(0, 255, (3, S))
            li = np.array([[3.0, -2.0, 0.5], [1.0, 4.0, -3.0], [6.0, 5.0, 7.0]], np.float32)[:, :2 + trial % 2]
            pl = np.array([[0, 0, 0, 0, 0, 1, 120, 130, 140]], np.float32).T
            cam = Camera((w, h), [-6.0, 0.5 * trial, 2.5], [0, -20, 3 * trial], fov=50.0)
            r.set_scene(sp, li, pl); r.set_camera(cam.position, cam.rotation); r.set_raygen(w, h, *cam.raygen())
            depth, aa = 1 + trial, bool(trial % 2)
            u8, f32 = r.render(0.1, 0.6, 0.4, depth, aa, u8=True, f32=True)
            ref = oracle.render(w, h, cam.position, cam.rotation, sp, li, pl, 0.1, 0.6, 0.4, depth, aa, raygen=cam.raygen(), want=("u8", "f32"))
            assert np.array_equal(u8, ref["u8"]), f"S={S}: {(u8 != ref['u8']).any(axis=0).sum()} px differ"
            assert np.array_equal(f32, ref["f32"]), f"S={S}"
    finally:
        r.close()


def test_xcd_affine_dispatch_groups(monkeypatch):
    """MI355RT_ORDER_GROUP (read at rt_create): the dispatch order is built from groups of 2^k consecutive blocks dealt
    to the XCDs in turn (rt::order_kernel).  Any group size must yield a permutation of the blocks — every pixel of the
    frame rendered exactly once — also when the last group is short and when fewer than 8 groups exist."""
    import python_ray_tracer_amd as pkg
    g = load_frame("c2_1080p")
    small = load_frame("odd_37x29")
    for k in ("1", "4", "6", "0", None):
        if k is None:
            monkeypatch.delenv("MI355RT_ORDER_GROUP", raising=False)
            monkeypatch.setenv("MI355RT_SEQ_ORDER", "1")       # (tile order also for kernels that default to longest-first)
        else:
            monkeypatch.setenv("MI355RT_ORDER_GROUP", k)
        r = pkg.Renderer(0)
        try:
            w, h, _ = _setup(r, g)
            d8 = r.malloc(3 * w * h)
            p = r.params(float(g["amb"]), float(g["lamb"]), float(g["refl"]), int(g["depth"]), 0, refl_pow=g["refl_pow"])
            for rep in range(4):                               # plain, measured, measured, settled order
                r.render_device(p, 0, w, d8, None, w * h)
                r.sync()
                got = np.empty((3, w, h), np.uint8); r.d2h(got, d8)
                assert np.array_equal(got, g["frame_u8"]), (k, rep)
            r.free(d8)
            assert r.stats()["launches_settled"] >= 1
            # the same geometry as a multi-frame launch: all but the last frame run in the XCDs' tile order (the second
            # permutation order_kernel builds), the last one longest-first — every frame must be the golden frame
            n = 5
            dn = r.malloc(n * 3 * w * h)
            for fpl in (5, 2):
                r.h2d(dn, np.zeros(n * 3 * w * h, np.uint8))
                r.render_sequence(p, 0, w, n, dn, None, w * h, 3 * w * h, None, None, fpl)
                r.sync()
                got = np.empty((n, 3, w, h), np.uint8); r.d2h(got, dn)
                for i in range(n):
                    assert np.array_equal(got[i], g["frame_u8"]), (k, fpl, i)
            r.free(dn)
            ws, hs, _ = _setup(r, small)
            for rep in range(4):
                u8, _ = r.render(float(small["amb"]), float(small["lamb"]), float(small["refl"]), int(small["depth"]), int(small["aa"]),
                                 refl_pow=small["refl_pow"])
                co = small["coords"]
                assert np.array_equal(u8[:, co[:, 0], co[:, 1]].T, small["u8"]), (k, rep)
            ps = r.params(float(small["amb"]), float(small["lamb"]), float(small["refl"]), int(small["depth"]), int(small["aa"]), refl_pow=small["refl_pow"])
            ds = r.malloc(3 * 3 * ws * hs)                     # fewer than eight groups: every block in the order's tail
            for rep in range(3):
                r.render_sequence(ps, 0, ws, 3, ds, None, ws * hs, 3 * ws * hs, None, None, 3)
                r.sync()
                got = np.empty((3, 3, ws, hs), np.uint8); r.d2h(got, ds)
                for i in range(3):
                    assert np.array_equal(got[i][:, co[:, 0], co[:, 1]].T, small["u8"]), (k, rep, i)
            r.free(ds)
        finally:
            r.close()


def test_host_frame_sequence_over_slots(renderer, oracle):
    """rt_render_begin / rt_render_end: frames queued on the slots without waiting arrive as the same bytes as the
    synchronous call's — the headline frame (goldens) into page-locked arrays on every slot, two frames deep, and a
    camera that moves between the begins (each frame its own oracle frame), float32 included."""
    import hashlib
    import python_ray_tracer_amd as pkg
    from python_ray_tracer_amd import _lib
    from python_ray_tracer_amd.scene import Camera
    g = load_frame("c2_1080p")
    w, h, _ = _setup(renderer, g)
    args = (float(g["amb"]), float(g["lamb"]), float(g["refl"]), int(g["depth"]), 0)
    nslots = _lib.RT_RENDER_SLOTS
    bufs = [renderer.host_arrays(s == 0, pinned=True) for s in range(2 * nslots)]
    try:
        for i, b in enumerate(bufs):                          # the second round queues behind the first on each slot
            renderer.render_begin(i % nslots, *args, *b, refl_pow=g["refl_pow"])
        for s in range(nslots):
            renderer.render_end(s)
        for b in bufs:
            assert np.array_equal(b[0], g["frame_u8"])
        assert hashlib.sha256(bufs[0][1].tobytes()).hexdigest() == str(g["sha256_rgb32"])
    finally:
        for b in bufs:
            renderer.release_host_arrays(b)
    with pytest.raises(pkg.RenderError):
        renderer.render_begin(nslots, *args, np.empty((3, w, h), np.uint8))
    with pytest.raises(pkg.RenderError):
        renderer.render_end(-1)
    # moving camera, pageable and page-locked destinations mixed, an unaligned column range
    g = load_frame("default_128_d3")
    w = h = 64
    cams = [Camera((w, h), [-2.0 + 0.4 * i, 0.3 * i - 0.5, 2.0 + 0.1 * i], [0, -30 + 3 * i, 5 * i], fov=45.0) for i in range(6)]
    renderer.set_scene(g["spheres"], g["lights"], g["planes"])
    outs = [(renderer.host_array((3, 51, h), np.uint8) if i % 2 else np.empty((3, 51, h), np.uint8),
             np.empty((3, 51, h), np.float32)) for i in range(len(cams))]
    try:
        for i, c in enumerate(cams):
            renderer.set_camera(c.position, c.rotation)
            renderer.set_raygen(w, h, *c.raygen())
            renderer.render_begin(i % nslots, float(g["amb"]), float(g["lamb"]), float(g["refl"]), 3, 0, *outs[i], x0=5, x1=56)
        for s in range(nslots):
            renderer.render_end(s)
        for i, c in enumerate(cams):
            ref = oracle.render(w, h, c.position, c.rotation, g["spheres"], g["lights"], g["planes"], float(g["amb"]), float(g["lamb"]),
                                float(g["refl"]), 3, False, raygen=c.raygen(), want=("u8", "f32"))
            assert np.array_equal(outs[i][0], ref["u8"][:, 5:56]), f"frame {i}"
            assert np.array_equal(outs[i][1], ref["f32"][:, 5:56]), f"frame {i}"
    finally:
        for i, o in enumerate(outs):
            if i % 2:
                renderer.release_host_array(o[0])


def test_host_path_chunked_and_pinned(renderer):
    """rt_render of a large frame runs as a pipeline of column chunks (render | copy to the host): same bytes as the
    goldens, into pageable and into page-locked arrays, uint8 alone and with the float32 buffer."""
    import hashlib
    g = load_frame("c2_1080p")
    w, h, _ = _setup(renderer, g)
    for pinned in (False, True):
        for want32 in (False, True):
            bufs = renderer.host_arrays(want32, pinned=pinned)
            try:
                for rep in range(3):                         # measuring, measuring, settled dispatch order
                    bufs[0][...] = 0
                    renderer.render_into(float(g["amb"]), float(g["lamb"]), float(g["refl"]), int(g["depth"]), 0, *bufs, refl_pow=g["refl_pow"])
                    assert np.array_equal(bufs[0], g["frame_u8"])
                    if want32:
                        assert hashlib.sha256(bufs[1].tobytes()).hexdigest() == str(g["sha256_rgb32"])
            finally:
                renderer.release_host_arrays(bufs)
    # an unaligned slab of it, and the 9-tap mode across chunk edges
    part8 = np.empty((3, 1203, h), np.uint8)
    renderer.render_into(float(g["amb"]), float(g["lamb"]), float(g["refl"]), int(g["depth"]), 0, part8, x0=301, x1=1504, refl_pow=g["refl_pow"])
    assert np.array_equal(part8, g["frame_u8"][:, 301:1504])
    aa_chunked = np.empty((3, w, h), np.uint8)
    renderer.render_into(0.0, 0.6, 0.3, 1, 1, aa_chunked)
    d8 = renderer.malloc(3 * w * h)
    try:
        renderer.render_device(renderer.params(0.0, 0.6, 0.3, 1, 1), 0, w, d8, None, w * h)
        renderer.sync()
        one = np.empty((3, w, h), np.uint8); renderer.d2h(one, d8)
    finally:
        renderer.free(d8)
    assert np.array_equal(aa_chunked, one)


def test_unaligned_slab(renderer):
    g = load_frame("odd_37x29")
    w, h, _ = _setup(renderer, g)
    full8, _ = _render(renderer, g)
    part8, _ = _render(renderer, g, x0=5, x1=18)
    assert np.array_equal(part8, full8[:, 5:18])


def test_u8_rgb_flag(renderer):
    from python_ray_tracer_amd import _lib as L
    g = load_frame("c1_128")
    _setup(renderer, g)
    a8, _ = _render(renderer, g)
    b8, _ = _render(renderer, g, flags=L.RT_FLAG_U8_RGB)
    assert np.array_equal(a8[[0, 2, 1]], b8)


def test_typed_bias_flag_matches_oracle(renderer, oracle):
    """The float64 BIAS*N variant (numba typing) is unpinned by the reference here; it must still
    agree with the oracle's implementation of the same variant."""
    from python_ray_tracer_amd import _lib as L
    g = load_frame("tilted_planes_48")
    w, h = int(g["w"]), int(g["h"])
    renderer.set_scene(g["spheres"], g["lights"], g["planes"], flags=L.RT_FLAG_TYPED_BIAS)
    renderer.set_camera(g["cam_origin"], g["cam_rot"])
    rg = raygen_closed_form(w, h, float(g["fov"]))
    renderer.set_raygen(w, h, *rg)
    u8, f32 = _render(renderer, g)
    ref = oracle.render(w, h, g["cam_origin"], g["cam_rot"], g["spheres"], g["lights"], g["planes"], float(g["amb"]),
                        float(g["lamb"]), float(g["refl"]), int(g["depth"]), False, raygen=rg, refl_pow=g["refl_pow"],
                        flags=oracle.FLAG_TYPED_BIAS, want=("u8", "f32"))
    assert np.array_equal(u8, ref["u8"]) and np.array_equal(f32, ref["f32"])


def test_random_scenes_vs_oracle(renderer, oracle):
    """Seeded random scenes (sizes the oracle finishes in seconds): bit-exact uint8 and float32."""
    rng = np.random.default_rng(42)
    for trial in range(8):
        S, P, Ln = int(rng.integers(0, 40)), int(rng.integers(0, 4)), int(rng.integers(0, 6))
        if trial >= 6:
            S = (130, 301)[trial - 6]          # larger clustered scenes, ragged last cluster
        sp = np.zeros((7, S), np.float32)
        sp[0:3] = rng.uniform(-4, 6, (3, S)); sp[3] = rng.uniform(0.1, 1.2, S); sp[4:7] = rng.integers(0, 256, (3, S))
        pl = np.zeros((9, P), np.float32)
        pl[0:3] = rng.uniform(-3, 3, (3, P))
        n = rng.normal(size=(3, P)); pl[3:6] = n / np.linalg.norm(n, axis=0, keepdims=True); pl[6:9] = rng.integers(0, 256, (3, P))
        if trial in (3, 4) and P:                # planes with exactly axis-aligned normals of either sign (kernel shortcut)
            pl[3:6] = 0
            for j in range(P):
                pl[3 + int(rng.integers(0, 3)), j] = float(rng.choice([-1.0, 1.0]))
        li = rng.uniform(-6, 8, (3, Ln)).astype(np.float32)
        w, h = int(rng.integers(20, 90)), int(rng.integers(20, 90))
        from python_ray_tracer_amd.scene import Camera
        cam = Camera((w, h), list(rng.uniform(-3, 0, 3) + [0, 0, 2.5]), list(rng.uniform(-30, 30, 3)), fov=float(rng.uniform(30, 80)))
        rg = cam.raygen()
        amb, lamb, refl, depth, aa = float(rng.uniform(0, 0.3)), float(rng.uniform(0.2, 0.9)), float(rng.uniform(0, 0.8)), int(rng.integers(0, 7)), bool(trial % 2)
        renderer.set_scene(sp, li, pl); renderer.set_camera(cam.position, cam.rotation); renderer.set_raygen(w, h, *rg)
        u8, f32 = renderer.render(amb, lamb, refl, depth, aa, u8=True, f32=True)
        ref = oracle.render(w, h, cam.position, cam.rotation, sp, li, pl, amb, lamb, refl, depth, aa, raygen=rg, want=("u8", "f32"))
        assert np.array_equal(u8, ref["u8"]), f"trial {trial}: {(u8 != ref['u8']).any(axis=0).sum()} px differ"
        assert np.array_equal(f32, ref["f32"]), f"trial {trial}"


def test_facade_matches_reference_call_shape(renderer):
    """The driver's call sequence (main.py:15-51) through the numba-shaped facade."""
    from python_ray_tracer_amd import cuda
    from python_ray_tracer_amd.ray_tracing import render
    from python_ray_tracer_amd.scene import Scene, Camera
    g = load_frame("default_128_d3")
    w, h = 128, 128
    spheres_host, light_host, planes_host = Scene.default_scene().generate_scene()
    spheres, lights, planes = cuda.to_device(spheres_host), cuda.to_device(light_host), cuda.to_device(planes_host)
    camera = Camera(resolution=(w, h), position=[-2, 0, 2.0], euler=[0, -30, 0])
    camera_origin, camera_rotation = cuda.to_device(camera.position), cuda.to_device(camera.rotation)
    pixel_loc = cuda.to_device(camera.generate_pixel_locations())
    result = cuda.to_device(np.zeros((3, w, h), dtype=np.uint8))
    threadsperblock = (16, 16)
    blockspergrid = (int(np.ceil(w / 16)), int(np.ceil(h / 16)))
    render[blockspergrid, threadsperblock](pixel_loc, result, camera_origin, camera_rotation,
                                           spheres, lights, planes, 0.0, 0.6, 0.3, 3, False)
    out = result.copy_to_host()
    assert np.array_equal(out, g["frame_u8"])
    # plain ndarrays + an explicit (untagged) pixel grid + a partial launch grid
    res2 = np.zeros((3, w, h), np.uint8)
    render[(4, 8), (16, 16)](np.array(camera.generate_pixel_locations()), res2, camera.position, camera.rotation,
                             spheres_host, light_host, planes_host, 0.0, 0.6, 0.3, 3, False)
    assert np.array_equal(res2[:, :64], g["frame_u8"][:, :64]) and not res2[:, 64:].any()


def test_error_behaviour(renderer):
    import python_ray_tracer_amd as pkg
    r = pkg.Renderer(0)
    with pytest.raises(pkg.RenderError) as e:
        r.render(0.0, 0.6, 0.3, 1)
    assert e.value.status == -4  # RT_ERR_STATE
    g = load_frame("c1_128")
    _setup(r, g)
    with pytest.raises(pkg.RenderError):
        r.render(0.0, 0.6, 0.3, 99)          # depth > RT_MAX_DEPTH
    with pytest.raises(pkg.RenderError):
        r.render(0.0, 0.6, 0.3, 1, x0=10, x1=5)
    with pytest.raises(pkg.RenderError):
        r.render(0.0, 0.6, 0.3, 1, 2, spp=0)  # stochastic AA without a sample count
    r.set_pixel_loc(np.zeros((3, 8, 8)))
    with pytest.raises(pkg.RenderError) as e:
        r.render(0.0, 0.6, 0.3, 1, 2, spp=4)  # stochastic AA needs the closed-form grid
    assert e.value.status == -4
    with pytest.raises(pkg.RenderError):
        pkg.Renderer(10_000)                  # no such device
    r.close()


def test_feedback_dispatch_order_never_changes_pixels(renderer):
    """Longest-first dispatch: from the second launch of a geometry on, workgroups run in the order built from
    the previous launch's per-tile cycles.  Every launch must still produce the reference frame."""
    from python_ray_tracer_amd import _lib as L
    g = load_frame("default_128_d3")
    _setup(renderer, g)
    for _ in range(5):
        u8, f32 = _render(renderer, g)
        assert np.array_equal(u8, g["frame_u8"])
    plain8, plain32 = _render(renderer, g, flags=L.RT_FLAG_NO_FEEDBACK)
    assert np.array_equal(plain8, g["frame_u8"]) and np.array_equal(plain32, f32)
    # geometry changes in between (slabs), then back
    a8, _ = _render(renderer, g, x0=8, x1=72)
    b8, _ = _render(renderer, g, x0=8, x1=72)
    assert np.array_equal(a8, g["frame_u8"][:, 8:72]) and np.array_equal(b8, a8)
    u8, _ = _render(renderer, g)
    assert np.array_equal(u8, g["frame_u8"])
    # Settled order (third launch on: no measuring, no rebuild), then another golden's scene
    # (another scene and depth: stale order on the first frame after it, measuring again) and back.
    g2 = load_frame("c1_128")
    for _ in range(4):
        u8, _ = _render(renderer, g)
        assert np.array_equal(u8, g["frame_u8"])
    _setup(renderer, g2)
    for _ in range(4):
        u8, _ = _render(renderer, g2)
        assert np.array_equal(u8, g2["frame_u8"])
    _setup(renderer, g)
    for _ in range(4):
        u8, f32b = _render(renderer, g)
        assert np.array_equal(u8, g["frame_u8"]) and np.array_equal(f32b, f32)


def test_frames_pipelined_over_streams(renderer):
    """Frames of one context launched alternately on three streams, with scheduler feedback on: while the order is
    being measured only the owning stream uses it, once settled every stream dispatches in it, and a setter call in
    the middle of in-flight settled launches starts measuring again.  Every frame must be the reference frame."""
    g = load_frame("c2_1080p")
    w, h, _ = _setup(renderer, g)
    p = renderer.params(float(g["amb"]), float(g["lamb"]), float(g["refl"]), int(g["depth"]), 0, refl_pow=g["refl_pow"])
    streams = [renderer.stream_create() for _ in range(3)]
    bufs = [renderer.malloc(3 * w * h) for _ in range(3)]
    zero = np.zeros((3, w, h), np.uint8)

    def burst(n):
        for i in range(n):
            renderer.render_device(p, 0, w, bufs[i % 3], None, w * h, stream=streams[i % 3])

    def remeasure():
        # an rt_set_* call that changes something starts measuring again (the same camera again does not, by
        # design: the reference's driver passes it with every launch)
        renderer.set_camera(np.asarray(g["cam_origin"]) + 1.0, g["cam_rot"])
        renderer.set_camera(g["cam_origin"], g["cam_rot"])

    def check():
        for s in streams:
            renderer.sync(s)
        for b in bufs:
            got = np.empty((3, w, h), np.uint8)
            renderer.d2h(got, b)
            assert np.array_equal(got, g["frame_u8"])
            renderer.h2d(b, zero)
    try:
        for b in bufs:
            renderer.h2d(b, zero)
        burst(12)
        remeasure()
        burst(9)
        check()
        remeasure()
        burst(3)                                                  # measuring phase only: the other streams render in plain order
        check()
        burst(2)
        renderer.stream_destroy(streams.pop(0))                   # the owner of the feedback buffers goes away
        streams.append(renderer.stream_create())
        burst(7)
        check()
    finally:
        for s in streams:
            renderer.stream_destroy(s)
        for b in bufs:
            renderer.free(b)


def test_feedback_takeover_by_another_stream(renderer):
    """the stream that owns the scheduler-feedback buffers has settled frames queued (they read the
    dispatch order) when, after an rt_set_* call, the first launch goes to ANOTHER stream, which takes the buffers
    over and rebuilds the order.  It must wait for the owner's queued frames; every frame must be the reference."""
    g = load_frame("c2_1080p")
    w, h, _ = _setup(renderer, g)
    p = renderer.params(float(g["amb"]), float(g["lamb"]), float(g["refl"]), int(g["depth"]), 0, refl_pow=g["refl_pow"])
    s0, s1 = renderer.stream_create(), renderer.stream_create()
    nb = 24
    bufs = [renderer.malloc(3 * w * h) for _ in range(nb)]
    zero = np.zeros((3, w, h), np.uint8)
    try:
        for rep in range(3):
            for b in bufs:
                renderer.h2d(b, zero)
            for i in range(4):                                    # s0 measures twice and settles
                renderer.render_device(p, 0, w, bufs[i], None, w * h, stream=s0)
            renderer.sync(s0)                                     # the order kernel is done: a takeover is allowed
            for i in range(4, 16):                                # settled frames queued on the owner, not waited for
                renderer.render_device(p, 0, w, bufs[i], None, w * h, stream=s0)
            renderer.set_camera(np.asarray(g["cam_origin"]) + 1.0, g["cam_rot"])
            renderer.set_camera(g["cam_origin"], g["cam_rot"])    # epoch bumped: the next launch measures
            for i in range(16, nb):                               # ... on the other stream: takes over, rebuilds the order
                renderer.render_device(p, 0, w, bufs[i], None, w * h, stream=s1)
            renderer.sync(s0); renderer.sync(s1)
            for i, b in enumerate(bufs):
                got = np.empty((3, w, h), np.uint8)
                renderer.d2h(got, b)
                assert np.array_equal(got, g["frame_u8"]), f"rep {rep} frame {i}: {(got != g['frame_u8']).any(axis=0).sum()} px differ"
            s0, s1 = s1, s0
    finally:
        renderer.stream_destroy(s0); renderer.stream_destroy(s1)
        for b in bufs:
            renderer.free(b)


def test_repeated_facade_launches_settle(renderer):
    """the reference's call shape passes the camera with every launch (main.py:41-47); the same
    camera again must not restart the dispatch-order measurement.  rt_get_stats counts settled launches."""
    from python_ray_tracer_amd import cuda
    from python_ray_tracer_amd.ray_tracing import render
    from python_ray_tracer_amd.scene import Scene, Camera
    g = load_frame("default_128_d3")
    w = h = 128
    sp, li, pl = (cuda.to_device(a) for a in Scene.default_scene().generate_scene())
    camera = Camera(resolution=(w, h), position=[-2, 0, 2.0], euler=[0, -30, 0])
    co, cr = cuda.to_device(camera.position), cuda.to_device(camera.rotation)
    grid = cuda.to_device(camera.generate_pixel_locations())
    result = cuda.to_device(np.zeros((3, w, h), dtype=np.uint8))
    r = cuda.current_renderer()
    before = r.stats()
    for _ in range(6):
        render[(8, 8), (16, 16)](grid, result, co, cr, sp, li, pl, 0.0, 0.6, 0.3, 3, False)
        r.sync()          # (a new order is switched to by the first launch that finds its build complete: give each one the chance)
    assert np.array_equal(result.copy_to_host(), g["frame_u8"])
    after = r.stats()
    assert after["launches"] - before["launches"] == 6
    assert after["launches_settled"] - before["launches_settled"] >= 3      # launches 3.. dispatch in the settled order
    # plain ndarrays for the camera every time (always re-sent): still settled, the library compares the values
    for _ in range(4):
        render[(8, 8), (16, 16)](grid, result, camera.position, camera.rotation, sp, li, pl, 0.0, 0.6, 0.3, 3, False)
    assert r.stats()["launches_settled"] - after["launches_settled"] == 4


@pytest.mark.parametrize("case", ["default_128_d3", "tilted_planes_48", "stoch_48_spp4"])
def test_ray_counters_match_oracle(renderer, oracle, case):
    """rt_get_stats: the counting instantiation reports the queries it traces; together with the shadow queries it
    skips (answer unused, trace.py:101) they are exactly the reference algorithm's counts (oracle counters)."""
    from python_ray_tracer_amd import _lib as L
    g = load_frame(case)
    w, h, rg = _setup(renderer, g)
    plain8, plain32 = _render(renderer, g)
    renderer.reset_stats()
    u8, f32 = _render(renderer, g, flags=L.RT_FLAG_COUNT_RAYS)
    assert np.array_equal(u8, plain8) and np.array_equal(f32, plain32)
    st = renderer.stats()
    ref = oracle.render(w, h, g["cam_origin"], g["cam_rot"], g["spheres"], g["lights"], g["planes"], float(g["amb"]),
                        float(g["lamb"]), float(g["refl"]), int(g["depth"]), int(g["aa"]), raygen=rg, refl_pow=g["refl_pow"],
                        want=(), spp=int(g["spp"]) if "spp" in g else 0, seed=int(g["seed"]) if "seed" in g else 1)["counters"]
    assert st["closest_queries"] == ref["closest"] and st["hits"] == ref["hits"]
    assert st["shadow_traced"] + st["shadow_skipped"] == ref["shadow"]
    assert 0 < st["shadow_traced"] <= ref["shadow"]
    _render(renderer, g, flags=L.RT_FLAG_COUNT_RAYS)              # counters accumulate until reset
    assert renderer.stats()["closest_queries"] == 2 * ref["closest"]
    renderer.reset_stats()
    assert renderer.stats()["closest_queries"] == 0


def test_moving_camera_over_streams(renderer, oracle):
    """A different camera for every frame, frames queued on three streams without waiting: each camera position
    needs its own cull tables (built by a small kernel on the launching stream, cached for the last few positions,
    overwritten only behind events from the streams that read them).  Every frame must be its own oracle frame."""
    from python_ray_tracer_amd.scene import Camera
    g = load_frame("default_128_d3")
    w = h = 64
    cams = [Camera((w, h), [-2.0 + 0.4 * i, 0.3 * i - 0.5, 2.0 + 0.1 * i], [0, -30 + 3 * i, 5 * i], fov=45.0) for i in range(5)]
    p = renderer.params(float(g["amb"]), float(g["lamb"]), float(g["refl"]), 3, 0)
    refs = [oracle.render(w, h, c.position, c.rotation, g["spheres"], g["lights"], g["planes"], float(g["amb"]), float(g["lamb"]),
                          float(g["refl"]), 3, False, raygen=c.raygen(), want=("u8",))["u8"] for c in cams]
    renderer.set_scene(g["spheres"], g["lights"], g["planes"])
    streams = [renderer.stream_create() for _ in range(3)]
    nframes = 20
    bufs = [renderer.malloc(3 * w * h) for _ in range(nframes)]
    try:
        for i in range(nframes):
            c = cams[i % 5]
            renderer.set_camera(c.position, c.rotation)
            renderer.set_raygen(w, h, *c.raygen())
            renderer.render_device(p, 0, w, bufs[i], None, w * h, stream=streams[i % 3])
        for s_ in streams:
            renderer.sync(s_)
        for i in range(nframes):
            got = np.empty((3, w, h), np.uint8)
            renderer.d2h(got, bufs[i])
            assert np.array_equal(got, refs[i % 5]), f"frame {i}"
    finally:
        for s_ in streams:
            renderer.stream_destroy(s_)
        for b in bufs:
            renderer.free(b)


def test_tile_stats(renderer):
    g = load_frame("default_128_d3")
    w, h, _ = _setup(renderer, g)
    ntiles = ((w + 7) // 8) * ((h + 7) // 8)
    d = renderer.malloc(4 * ntiles)
    try:
        renderer.h2d(d, np.zeros(ntiles, np.uint32))
        renderer.set_tile_stats(d)
        u8, _ = _render(renderer, g)
        cyc = np.empty(ntiles, np.uint32)
        renderer.d2h(cyc, d)
        assert np.array_equal(u8, g["frame_u8"]) and (cyc > 0).all() and cyc.max() < 50_000_000
    finally:
        renderer.set_tile_stats(None)
        renderer.free(d)


def test_u8_hwc_image_layout(renderer):
    """RT_FLAG_U8_HWC: the device writes the interleaved (h,w,3) image the reference's viewer builds on the host."""
    from python_ray_tracer_amd import _lib as L
    from python_ray_tracer_amd.viewer import frame_to_hwc
    g = load_frame("odd_37x29")
    _setup(renderer, g)
    planar, _ = renderer.render(0.0, 0.6, 0.3, int(g["depth"]), 0, u8=True, f32=False, refl_pow=g["refl_pow"])
    img, _ = renderer.render(0.0, 0.6, 0.3, int(g["depth"]), 0, u8=True, f32=False, refl_pow=g["refl_pow"], flags=L.RT_FLAG_U8_HWC)
    assert img.shape == (29, 37, 3) and np.array_equal(img, frame_to_hwc(planar))
    rgb, _ = renderer.render(0.0, 0.6, 0.3, int(g["depth"]), 0, u8=True, f32=False, refl_pow=g["refl_pow"],
                             flags=L.RT_FLAG_U8_HWC | L.RT_FLAG_U8_RGB)
    assert np.array_equal(rgb, frame_to_hwc(planar, undo_swap=True))
    part, _ = renderer.render(0.0, 0.6, 0.3, int(g["depth"]), 0, u8=True, f32=False, refl_pow=g["refl_pow"], flags=L.RT_FLAG_U8_HWC, x0=8, x1=24)
    assert np.array_equal(part, img[:, 8:24])
    import python_ray_tracer_amd as pkg
    with pytest.raises(pkg.RenderError):
        renderer.render(0.0, 0.6, 0.3, 1, 0, u8=True, f32=True, flags=L.RT_FLAG_U8_HWC)


def _limits_scene_check(r, oracle, S, Ln, P):
    rng = np.random.default_rng(S + Ln)
    sp = np.zeros((7, S), np.float32)
    sp[0:3] = rng.uniform(-6, 8, (3, S)); sp[3] = rng.uniform(0.05, 0.4, S); sp[4:7] = rng.integers(0, 256, (3, S))
    li = rng.uniform(-6, 8, (3, Ln)).astype(np.float32)
    if Ln:
        li[2] = np.abs(li[2]) + 2.0
    pl = np.zeros((9, P), np.float32)
    if P:
        pl[0:3] = rng.uniform(-3, 3, (3, P)); pl[2] -= 6.0
        n = rng.normal(size=(3, P)); n[2] = np.abs(n[2]) + 1.0
        pl[3:6] = n / np.linalg.norm(n, axis=0, keepdims=True); pl[6:9] = rng.integers(0, 256, (3, P))
    from python_ray_tracer_amd.scene import Camera
    w, h = 48, 40
    cam = Camera((w, h), [-3.0, 0.5, 2.5], [3, -25, 4])
    rg = cam.raygen()
    r.set_scene(sp, li, pl); r.set_camera(cam.position, cam.rotation); r.set_raygen(w, h, *rg)
    for aa in (0, 1):
        u8, f32 = r.render(0.05, 0.5, 0.4, 3, aa, u8=True, f32=True)
        ref = oracle.render(w, h, cam.position, cam.rotation, sp, li, pl, 0.05, 0.5, 0.4, 3, aa, raygen=rg, want=("u8", "f32"))
        assert np.array_equal(u8, ref["u8"]), f"S={S} aa={aa}: {(u8 != ref['u8']).any(axis=0).sum()} px differ"
        assert np.array_equal(f32, ref["f32"])
    return li, pl


@pytest.mark.parametrize("S,Ln,P", [(1024, 3, 1),     # lane-owned traversal without an anchored table (no room in LDS): boxes and origin-form spheres only
                                    (700, 64, 64), (97, 0, 0), (150, 12, 2), (60, 10, 1), (256, 2, 1),
                                    (161, 3, 1), (200, 1, 0)])     # around the lane-owned traversal's threshold; 25 clusters = 3 groups + 1
def test_scene_size_limits_vs_oracle(renderer, oracle, S, Ln, P):
    """RT_MAX_SPHERES / RT_MAX_LIGHTS / RT_MAX_PLANES: LDS images beyond 64 KiB, no room for the anchored cull
    table (origin-form culling with clusters only), many lights and planes — bit-exact against the oracle."""
    li, pl = _limits_scene_check(renderer, oracle, S, Ln, P)
    import python_ray_tracer_amd as pkg
    with pytest.raises(pkg.RenderError):
        renderer.set_scene(np.zeros((7, 1025), np.float32), li, pl)


def test_wave_uniform_cull_at_large_scenes(monkeypatch, oracle):
    """MI355RT_LANES_MINS=100000 keeps the wave-uniform cluster cull for scenes the library would hand to the lane-owned
    traversal: no anchored table (too many lights for its LDS budget: origin-form certificates only), many lights, 256 spheres."""
    import python_ray_tracer_amd as pkg
    monkeypatch.setenv("MI355RT_LANES_MINS", "100000")
    r = pkg.Renderer(0)
    try:
        for S, Ln, P in ((150, 12, 2), (60, 10, 1), (256, 2, 1)):
            _limits_scene_check(r, oracle, S, Ln, P)
    finally:
        r.close()


@pytest.mark.parametrize("records", ["1", "0"])
def test_four_wave_kernels_without_float64_records(monkeypatch, oracle, records):
    """MI355RT_F32_RECORDS (default 1): four-wave wave-uniform kernels stage no float64 sphere records where that lets a CU hold
    one workgroup more (rt_device.h MODE 1: 64 and 100 spheres take the parked variant, 256 under the wave-uniform cull the
    register variant; aa=1 on the closed-form grid runs the same kernels over the half-pixel lattice).  Either setting must
    reproduce the oracle bit for bit."""
    import python_ray_tracer_amd as pkg
    monkeypatch.setenv("MI355RT_F32_RECORDS", records)
    monkeypatch.setenv("MI355RT_LANES_MINS", "100000")
    r = pkg.Renderer(0)
    try:
        for S in (64, 100, 140, 256):
            _limits_scene_check(r, oracle, S, 3, 1)
    finally:
        r.close()


@pytest.mark.parametrize("scale", [1e-30, 1e-12, 1e-4, 50.0, 1e6, 1e30])   # oracle pictures: plane only / 35 / 606 colours / scene straddling the far limit / black / black
def test_extreme_scene_magnitudes(renderer, oracle, scale):
    """The whole scene (spheres, plane, lights, camera) scaled far away from the magnitudes the reference's constants assume
    (BIAS 2e-4, far limit 999, plane threshold 1e-3): tiny scenes put numerators and discriminants at 1e-60 and below (float32
    r*r underflows to zero), huge ones overflow the float32 cull (which then certifies nothing) and push every t beyond the far
    limit.  Exercises the range assumptions of div_inrange / sqrt_inrange / normalize3's guard: GPU and oracle must still agree
    bit for bit, whatever the picture is."""
    from python_ray_tracer_amd.scene import Camera
    rng = np.random.default_rng(11)
    S, w, h = 24, 48, 40
    sp = np.zeros((7, S), np.float32)
    sp[0:3] = (rng.uniform(-4, 4, (3, S)) * scale).astype(np.float32); sp[2] = np.abs(sp[2])
    sp[3] = (rng.uniform(0.2, 0.9, S) * scale).astype(np.float32); sp[4:7] = rng.integers(0, 256, (3, S))
    li = (np.array([[3.0, -2.0, 5.0], [1.0, 4.0, 6.0]]).T * scale).astype(np.float32)
    pl = np.array([[0, 0, 0, 0, 0, 1, 120, 130, 140]], np.float32).T
    cam = Camera((w, h), [-9.0 * scale, 0.5 * scale, 2.5 * scale], [0, -12, 3], fov=50.0)
    rg = cam.raygen()
    renderer.set_scene(sp, li, pl); renderer.set_camera(cam.position, cam.rotation); renderer.set_raygen(w, h, *rg)
    for aa in (0, 1):
        u8, f32 = renderer.render(0.1, 0.6, 0.4, 3, aa, u8=True, f32=True)
        ref = oracle.render(w, h, cam.position, cam.rotation, sp, li, pl, 0.1, 0.6, 0.4, 3, aa, raygen=rg, want=("u8", "f32"))
        assert np.array_equal(u8, ref["u8"]), f"scale={scale} aa={aa}: {(u8 != ref['u8']).any(axis=0).sum()} px differ"
        assert np.array_equal(f32, ref["f32"], equal_nan=True), f"scale={scale} aa={aa}"


def test_max_depth(renderer, oracle):
    g = load_frame("fov70_48")
    w, h, rg = _setup(renderer, g)
    u8, f32 = renderer.render(0.02, 0.6, 0.6, 16, 0, u8=True, f32=True)
    ref = oracle.render(w, h, g["cam_origin"], g["cam_rot"], g["spheres"], g["lights"], g["planes"], 0.02, 0.6, 0.6, 16, 0,
                        raygen=rg, want=("u8", "f32"))
    assert np.array_equal(u8, ref["u8"]) and np.array_equal(f32, ref["f32"])


@pytest.mark.parametrize("case", ["aa_48_d2", "stoch_48_spp4"])
def test_column_slabs_with_antialiasing(renderer, case):
    """Slab tiling must not change AA results: the 9-tap mode reads neighbours across the slab edge, the
    stochastic mode hashes absolute pixel coordinates."""
    from python_ray_tracer_amd.distributed import slab_bounds
    g = load_frame(case)
    w, h, _ = _setup(renderer, g)
    full8, full32 = _render(renderer, g)
    for n in (2, 3):
        parts = [_render(renderer, g, x0=a, x1=b) for a, b in (slab_bounds(w, n, r) for r in range(n))]
        assert np.array_equal(np.concatenate([p[0] for p in parts], axis=1), full8)
        assert np.array_equal(np.concatenate([p[1] for p in parts], axis=1), full32)
    _setup(renderer, g, explicit_grid=(case == "aa_48_d2"))
    if case == "aa_48_d2":          # explicit pixel_loc + slabs + 9 taps
        part8, _ = _render(renderer, g, x0=16, x1=32)
        assert np.array_equal(part8, full8[:, 16:32])


def test_c_abi_example_without_python(tmp_path, renderer):
    """examples/render_c_abi.c links the library from plain C (gcc), renders and writes an image; the same inputs
    through the Python host must give the same bytes."""
    import math, subprocess
    from conftest import REPO
    exe, out = str(tmp_path / "render_c_abi"), str(tmp_path / "img.ppm")
    libdir = os.path.join(REPO, "python-ray-tracer_amd")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "examples", "render_c_abi.c"),
                           "-L", libdir, "-lmi355rt", "-lm", f"-Wl,-rpath,{libdir}", "-o", exe])
    log = subprocess.check_output([exe, out], text=True)
    assert "wrote" in log
    assert "sequence: 5 frames identical" in log, log          # rt_render_sequence from plain C: batches of 3 + 2 frames
    raw = open(out, "rb").read()
    assert raw.startswith(b"P6\n512 512\n255\n")
    img = np.frombuffer(raw[len(b"P6\n512 512\n255\n"):], np.uint8).reshape(512, 512, 3)
    from python_ray_tracer_amd.scene import Scene
    from python_ray_tracer_amd import _lib as L
    sp, li, pl = Scene.default_scene().generate_scene()
    th = -30.0 * math.pi / 180.0
    c, s = math.cos(th), math.sin(th)
    renderer.set_scene(sp, li, pl)
    renderer.set_camera([-2.0, 0.0, 2.0], [c, 0, -s, 0, 1, 0, s, 0, c])
    renderer.set_raygen(512, 512, 1.0 / math.tan((45.0 * math.pi / 180.0) / 2.0), 1.0, -2.0 / 511.0, 1.0, -2.0 / 511.0)
    ref, _ = renderer.render(0.0, 0.6, 0.3, 2, 1, u8=True, f32=False, flags=L.RT_FLAG_U8_HWC | L.RT_FLAG_U8_RGB,
                             refl_pow=[math.pow(0.3, i + 1) for i in range(2)])
    assert np.array_equal(img, ref)


def test_renderer_example_writes_png(tmp_path):
    """examples/render_png.py: Renderer API, device-side (h,w,3) image into page-locked memory, PNG on disk."""
    import subprocess, sys
    from conftest import REPO
    out = str(tmp_path / "r.png")
    log = subprocess.check_output([sys.executable, os.path.join(REPO, "examples", "render_png.py"), "--size", "160x96", "--depth", "2",
                                   "--aa", "--frames", "3", "--out", out], text=True)
    assert "wrote" in log
    from PIL import Image
    im = np.asarray(Image.open(out))
    assert im.shape == (96, 160, 3) and im.any()


def test_many_launch_geometries_share_the_feedback_slots(renderer):
    """The context keeps the dispatch-order feedback of 8 launch geometries; 11 different column ranges, each launched
    three times in rotation (so every slot is evicted and re-measured), must all produce their part of the frame."""
    g = load_frame("default_128_d3")
    w, h, _ = _setup(renderer, g)
    full8, _ = _render(renderer, g)
    ranges = [(8 * i, 8 * i + 40) for i in range(11)]
    before = renderer.stats()
    for rep in range(3):
        for a, b in ranges:
            part8, _ = _render(renderer, g, x0=a, x1=b)
            assert np.array_equal(part8, full8[:, a:b]), (rep, a, b)
    st = renderer.stats()
    assert st["launches"] - before["launches"] == 33 and st["launches_measuring"] - before["launches_measuring"] >= 22


def test_stream_forget(renderer):
    """A caller-owned stream (here one made by the library, standing in for a torch stream) can be forgotten and used again."""
    g = load_frame("c1_128")
    w, h, _ = _setup(renderer, g)
    p = renderer.params(float(g["amb"]), float(g["lamb"]), float(g["refl"]), int(g["depth"]), 0, refl_pow=g["refl_pow"])
    s_ = renderer.stream_create()
    d8 = renderer.malloc(3 * w * h)
    try:
        for rep in range(3):
            for _ in range(4):
                renderer.render_device(p, 0, w, d8, None, w * h, stream=s_)
            renderer.stream_forget(s_)                     # waits for the stream, drops the context's references
            got = np.empty((3, w, h), np.uint8); renderer.d2h(got, d8)
            assert np.array_equal(got, g["frame_u8"])
    finally:
        renderer.stream_destroy(s_); renderer.free(d8)


@pytest.mark.parametrize("name", ["c4_3840x2160_s64_d5", "c5_7680x4320_s256_d8", "c5_7680x4320_s256_d8_spp4"])
def test_workload_ray_counts_are_the_device_counters(renderer, name):
    """The ray counts bench.py quotes for configs 4 and 5 (workloads.CONFIGS) are what the counting instantiation counts
    for those frames at full size — the same counters test_ray_counters_match_oracle pins to the oracle on small frames."""
    from python_ray_tracer_amd import workloads, _lib as L
    wl = workloads.build(name)
    cam, w, h = wl["camera"], wl["w"], wl["h"]
    renderer.set_scene(wl["spheres"], wl["lights"], wl["planes"]); renderer.set_camera(cam.position, cam.rotation)
    renderer.set_raygen(w, h, *cam.raygen())
    d8 = renderer.malloc(3 * w * h)
    try:
        renderer.reset_stats()
        p = renderer.params(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], wl["aa"], spp=wl["spp"], seed=wl["seed"], flags=L.RT_FLAG_COUNT_RAYS)
        renderer.render_device(p, 0, w, d8, None, w * h)
        st = renderer.stats()
    finally:
        renderer.free(d8)
    assert st["closest_queries"] == wl["rays"]["closest"] and st["shadow_traced"] + st["shadow_skipped"] == wl["rays"]["shadow"]


# ---- round 3: rt_render_sequence, scene changes with frames in flight, the XCD-affine order --------------------------

def test_render_sequence_every_frame_is_the_golden(renderer):
    """rt_render_sequence (main.py:41-47 launches frame after frame): n frames of the headline scene with ONE call —
    launches of several frames each, a partly filled last launch, one / three streams, uint8 and float32 — every frame of
    every batch must be the reference's frame.  The first calls measure the tile costs frame by frame (single launches),
    the later ones run whole batches per launch (launches < frames in rt_stats)."""
    import hashlib
    g = load_frame("c2_1080p")
    w, h, _ = _setup(renderer, g)
    p = renderer.params(float(g["amb"]), float(g["lamb"]), float(g["refl"]), int(g["depth"]), 0, refl_pow=g["refl_pow"])
    n = 11
    d8, d32 = renderer.malloc(n * 3 * w * h), renderer.malloc(n * 12 * w * h)
    streams = [renderer.stream_create() for _ in range(3)]
    try:
        for fpl, strs, want32 in ((4, streams, True), (0, None, False), (11, streams[:1], True), (1, streams, False)):
            renderer.reset_stats()
            zero = np.zeros(n * 3 * w * h, np.uint8)
            renderer.h2d(d8, zero)
            renderer.render_sequence(p, 0, w, n, d8, d32 if want32 else None, w * h, 3 * w * h, None, strs, fpl)
            for s_ in streams:
                renderer.sync(s_)
            renderer.sync()
            got = np.empty((n, 3, w, h), np.uint8)
            renderer.d2h(got, d8)
            for i in range(n):
                assert np.array_equal(got[i], g["frame_u8"]), (fpl, i, int((got[i] != g["frame_u8"]).any(axis=0).sum()))
            if want32:
                f32 = np.empty((n, 3, w, h), np.float32)
                renderer.d2h(f32, d32)
                for i in (0, n // 2, n - 1):
                    assert hashlib.sha256(f32[i].tobytes()).hexdigest() == str(g["sha256_rgb32"]), (fpl, i)
            st = renderer.stats()
            assert st["frames"] == n
        assert st["launches"] == n                                   # fpl = 1: a launch per frame
        renderer.reset_stats()
        renderer.render_sequence(p, 0, w, n, d8, None, w * h, 3 * w * h, None, streams, 4)
        for s_ in streams:
            renderer.sync(s_)
        st = renderer.stats()
        assert st["frames"] == n and st["launches"] == 3 and st["launches_settled"] == 3, st     # 4 + 4 + 3 frames
    finally:
        for s_ in streams:
            renderer.stream_destroy(s_)
        renderer.free(d8); renderer.free(d32)


@pytest.mark.parametrize("aa", [0, 1])
def test_render_sequence_slab_and_aa(renderer, oracle, aa):
    """A column slab rendered in place inside full frames (plane_stride = w*h, frame_stride = 3*w*h), with and without
    the 9-tap mode (which runs lattice + resolve per frame), against the oracle."""
    g = load_frame("default_128_d3")
    w, h, rg = _setup(renderer, g)
    p = renderer.params(float(g["amb"]), float(g["lamb"]), float(g["refl"]), 2, aa)
    ref = oracle.render(w, h, g["cam_origin"], g["cam_rot"], g["spheres"], g["lights"], g["planes"], float(g["amb"]), float(g["lamb"]),
                        float(g["refl"]), 2, bool(aa), raygen=rg, want=("u8", "f32"))
    n, x0, x1 = 5, 40, 99
    d8, d32 = renderer.malloc(n * 3 * w * h), renderer.malloc(n * 12 * w * h)
    try:
        for rep in range(3):                                         # measuring, measuring, settled
            renderer.h2d(d8, np.zeros(n * 3 * w * h, np.uint8))
            renderer.render_sequence(p, x0, x1, n, d8 + x0 * h, d32 + 4 * x0 * h, w * h, 3 * w * h, None, None, 3)
            renderer.sync()
            got8, got32 = np.empty((n, 3, w, h), np.uint8), np.empty((n, 3, w, h), np.float32)
            renderer.d2h(got8, d8); renderer.d2h(got32, d32)
            for i in range(n):
                assert np.array_equal(got8[i][:, x0:x1], ref["u8"][:, x0:x1]), (rep, i)
                assert np.array_equal(got32[i][:, x0:x1], ref["f32"][:, x0:x1]), (rep, i)
                assert not got8[i][:, :x0].any() and not got8[i][:, x1:].any()
    finally:
        renderer.free(d8); renderer.free(d32)
    import python_ray_tracer_amd as pkg
    with pytest.raises(pkg.RenderError):
        renderer.render_sequence(p, 0, w, 2, 1, None, w * h, 2 * w * h)            # frame_stride < 3 planes
    with pytest.raises(pkg.RenderError):
        renderer.render_sequence(p, 0, w, -1, 1, None, w * h, 3 * w * h)


@pytest.mark.parametrize("remeasure", ["24", "2", "0"])
def test_moving_camera_sequence(monkeypatch, oracle, remeasure):
    """rt_render_sequence with a camera per frame (bench.py's `dynamic` block; README.md:23, scene/camera.py:8-16): position
    and rotation change with EVERY frame, frames round-robin on three streams.  The dispatch order measured under an earlier
    camera is kept for MI355RT_REMEASURE launches and then measured again; cull tables follow the camera position.  Every
    frame must be its own oracle frame, uint8 and float32."""
    import python_ray_tracer_amd as pkg
    from python_ray_tracer_amd import workloads
    monkeypatch.setenv("MI355RT_REMEASURE", remeasure)
    wl = workloads.build("c2_1920x1080_s8_d3")
    w, h = 160, 96
    from python_ray_tracer_amd.scene import Camera
    rg = Camera((w, h), [0, 0, 0], [0, 0, 0], fov=45.0).raygen()
    n = 14
    cams = workloads.camera_path(n, period=9)
    r = pkg.Renderer(0)
    try:
        r.set_scene(wl["spheres"], wl["lights"], wl["planes"])
        r.set_raygen(w, h, *rg)
        p = r.params(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], 0)
        streams = [r.stream_create() for _ in range(3)]
        d8, d32 = r.malloc(n * 3 * w * h), r.malloc(n * 12 * w * h)
        for rep in range(2):
            r.render_sequence(p, 0, w, n, d8, d32, w * h, 3 * w * h, cams, streams)
            for s_ in streams:
                r.sync(s_)
            got8, got32 = np.empty((n, 3, w, h), np.uint8), np.empty((n, 3, w, h), np.float32)
            r.d2h(got8, d8); r.d2h(got32, d32)
            for i in range(n):
                ref = oracle.render(w, h, cams[i, 0:3], cams[i, 3:12].reshape(3, 3), wl["spheres"], wl["lights"], wl["planes"],
                                    wl["amb"], wl["lamb"], wl["refl"], wl["depth"], False, raygen=rg, want=("u8", "f32"))
                assert np.array_equal(got8[i], ref["u8"]), (rep, i)
                assert np.array_equal(got32[i], ref["f32"]), (rep, i)
        st = r.stats()
        assert st["frames"] == 2 * n and st["table_builds"] >= n
        if remeasure == "24":
            assert st["launches_measuring"] <= 3, st             # the first frames only: the order is kept while the camera moves
        for s_ in streams:
            r.stream_destroy(s_)
        r.free(d8); r.free(d32)
    finally:
        r.close()


def test_scene_changes_with_frames_in_flight(renderer, oracle):
    """rt_set_scene between launches that are still in flight on other streams (an animation that moves its spheres): every
    launch keeps the scene buffer it was queued with (a ring inside the context), so each frame must be the oracle's frame
    of ITS scene — more scene changes than the ring has buffers, frames large enough to still be running when the next
    scene arrives."""
    g = load_frame("default_128_d3")
    w, h = 512, 384
    from python_ray_tracer_amd.scene import Camera
    cam = Camera((w, h), [-2, 0, 2.0], [0, -30, 0], fov=45.0)
    rg = cam.raygen()
    renderer.set_camera(cam.position, cam.rotation)
    renderer.set_raygen(w, h, *rg)
    p = renderer.params(0.05, 0.6, 0.3, 3, 0)
    nscenes = 7
    scenes = []
    for i in range(nscenes):
        sp = np.array(g["spheres"], np.float32).copy()
        sp[0] += 0.15 * i; sp[1, ::2] -= 0.2 * i; sp[3] *= (1.0 + 0.05 * i)
        sp = sp[:, : sp.shape[1] - (i % 3)]                       # the sphere count changes too
        scenes.append(sp)
    streams = [renderer.stream_create() for _ in range(3)]
    bufs = [renderer.malloc(3 * w * h) for _ in range(nscenes)]
    try:
        for i in range(nscenes):                                     # no waiting between scene change and launch
            renderer.set_scene(scenes[i], g["lights"], g["planes"])
            renderer.render_device(p, 0, w, bufs[i], None, w * h, stream=streams[i % 3])
        for s_ in streams:
            renderer.sync(s_)
        for i in range(nscenes):
            got = np.empty((3, w, h), np.uint8)
            renderer.d2h(got, bufs[i])
            ref = oracle.render(w, h, cam.position, cam.rotation, scenes[i], g["lights"], g["planes"], 0.05, 0.6, 0.3, 3, False,
                                raygen=rg, want=("u8",))["u8"]
            assert np.array_equal(got, ref), f"scene {i}: {(got != ref).any(axis=0).sum()} pixels differ"
    finally:
        for s_ in streams:
            renderer.stream_destroy(s_)
        for b in bufs:
            renderer.free(b)


def test_pinned_arrays_outlive_their_renderer():
    """Page-locked arrays from Renderer.host_array() belong to the arrays: closing the Renderer must not free memory a
    numpy view still points at (round-2 advice), and a DeviceNDArray never hands a closed context's pointer to a new one."""
    import python_ray_tracer_amd as pkg
    from python_ray_tracer_amd import cuda
    g = load_frame("default_128_d3")
    r = pkg.Renderer(0)
    w, h, _ = _setup(r, g)
    a8, _ = r.host_arrays(False, pinned=True)
    r.render_into(float(g["amb"]), float(g["lamb"]), float(g["refl"]), int(g["depth"]), 0, a8, refl_pow=g["refl_pow"])
    view = a8[1]
    r.close()
    assert np.array_equal(a8, g["frame_u8"]) and np.array_equal(view, g["frame_u8"][1])      # still readable
    a8[:] = 7                                                                                   # ... and writable
    del a8, view
    # facade: result handle used across cuda.close()
    cuda.close()
    from python_ray_tracer_amd.ray_tracing import render
    from python_ray_tracer_amd.scene import Camera
    cam = Camera((w, h), [-2, 0, 2.0], [0, -30, 0], fov=float(g["fov"]))
    res = cuda.to_device(np.zeros((3, w, h), np.uint8))
    args = (cuda.to_device(cam.generate_pixel_locations()), res, cuda.to_device(np.array(g["cam_origin"])), cuda.to_device(np.array(g["cam_rot"])),
            cuda.to_device(g["spheres"]), cuda.to_device(g["lights"]), cuda.to_device(g["planes"]),
            float(g["amb"]), float(g["lamb"]), float(g["refl"]), int(g["depth"]), False)
    render[(w // 8, h // 8), (8, 8)](*args)
    first = res.copy_to_host()
    cuda.close()                                                     # the context (and res's device buffer) is gone
    render[(w // 8, h // 8), (8, 8)](*args)                          # a new context: res gets a new buffer
    assert np.array_equal(res.copy_to_host(), first)
    cuda.close()


def test_tile_stats_through_the_chunked_host_path(renderer):
    """rt_render splits a 1080p frame into four column chunks (separate launches): every chunk must record its tiles at
    the FRAME's tile indices (round-2 advice: the chunks used to overwrite each other from index 0)."""
    g = load_frame("c2_1080p")
    w, h, _ = _setup(renderer, g)
    tiles_y = (h + 7) // 8
    ntiles = ((w + 7) // 8) * tiles_y
    d = renderer.malloc(4 * ntiles)
    try:
        renderer.h2d(d, np.zeros(ntiles, np.uint32))
        renderer.set_tile_stats(d)
        u8, _ = renderer.render(float(g["amb"]), float(g["lamb"]), float(g["refl"]), int(g["depth"]), 0, refl_pow=g["refl_pow"])
        cyc = np.empty(ntiles, np.uint32)
        renderer.d2h(cyc, d)
        assert np.array_equal(u8, g["frame_u8"])
        assert (cyc > 0).all(), f"{(cyc == 0).sum()} of {ntiles} tiles recorded nothing"
    finally:
        renderer.set_tile_stats(None)
        renderer.free(d)
