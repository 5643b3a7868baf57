"""The build's own Scene / Camera / euler_rotation / viewer against arrays captured from the
reference's helpers (tests/golden/host_helpers.npz): byte-for-byte."""
import hashlib
import os

import numpy as np

from conftest import GOLDEN, load_frame

from python_ray_tracer_amd.scene import Scene, Camera, Plane, Sphere, Light, euler_rotation, PixelGrid
from python_ray_tracer_amd import workloads

H = np.load(os.path.join(GOLDEN, "host_helpers.npz"))


def test_default_scene_arrays():
    sp, li, pl = Scene.default_scene().generate_scene()
    for got, key in ((sp, "default_spheres"), (li, "default_lights"), (pl, "default_planes")):
        assert got.dtype == np.float32 and got.tobytes() == H[key].tobytes() and got.shape == H[key].shape


def test_to_array_layouts():
    assert np.array_equal(Plane([1, 2, 3], [0.3, -0.2, 1.0], [10, 20, 30]).to_array(), H["plane_tilted"])
    assert Sphere([1, 2, 3], 0.5, [1, 2, 3]).to_array().shape == (7,) and Light([1, 2, 3]).to_array().shape == (3,)
    s = Scene([], [], [])
    assert s.get_spheres().shape == (7, 0) and s.get_lights().shape == (3, 0) and s.get_planes().shape == (9, 0)


def test_euler_rotation():
    for e, r in zip(H["eulers"], H["rotations"]):
        assert euler_rotation(*e).tobytes() == r.tobytes()
    assert euler_rotation(0.1, -0.2, 0.3, is_radians=True).tobytes() == H["rotation_rad"].tobytes()


def test_pixel_grid_bit_identical_and_closed_form():
    for (w, h), sha in zip(H["pl_sizes"], H["pl_sha256"]):
        if w * h > 4_000_000:
            continue                      # 4K/8K are covered by the closed-form corner values below
        grid = Camera((int(w), int(h)), [-2, 0, 2.0], [0, -30, 0]).generate_pixel_locations()
        assert isinstance(grid, PixelGrid) and grid.dtype == np.float64
        assert hashlib.sha256(np.asarray(grid).tobytes()).hexdigest() == str(sha)
        px, y0, dy, z0, dz = grid.raygen
        assert grid[1, w - 1, 0] == (w - 1) * dy + y0 and grid[2, 0, h - 1] == (h - 1) * dz + z0 and grid[0, 0, 0] == px
        assert grid[:, 1:].raygen is None       # a slice no longer carries the closed form
    for (w, h), c in zip(H["pl_sizes"], H["pl_corners"]):
        px, y0, dy, z0, dz = Camera((int(w), int(h)), [-2, 0, 2.0], [0, -30, 0]).raygen()
        got = [px, y0, (w - 1) * dy + y0, z0, (h - 1) * dz + z0, 1 * dy + y0, 1 * dz + z0]
        assert np.array_equal(np.array(got), c)
    assert np.array_equal(Camera((16, 16), [-2, 0, 2.0], [0, -30, 0], fov=60.0).generate_pixel_locations(), H["pixel_loc_16x16_fov60"])
    assert Camera((4, 4), [-2, 0, 2.0], [0, -30, 0]).position.dtype == H["cam_pos_float"].dtype


def test_viewer_square_frame():
    from python_ray_tracer_amd.viewer import convert_array_to_image, frame_to_hwc
    assert np.array_equal(np.asarray(convert_array_to_image(H["viewer_in_8"])), H["viewer_out_8"])
    x = np.arange(3 * 6 * 4, dtype=np.uint8).reshape(3, 6, 4)
    y = frame_to_hwc(x)
    assert y.shape == (4, 6, 3) and y[2, 5, 1] == x[1, 5, 2]
    assert np.array_equal(frame_to_hwc(x, undo_swap=True)[..., 1], x[2].T)


def test_workloads_match_golden_scenes():
    for name, case in (("c1_128x128_s3_d1", "c1_128"), ("c2_1920x1080_s8_d3", "c2_1080p"),
                       ("c4_3840x2160_s64_d5", "c4_s64_d5_sub32"), ("c5_7680x4320_s256_d8", "c5_s256_d8_sub96")):
        wl, g = workloads.build(name), load_frame(case)
        assert (wl["w"], wl["h"], wl["depth"]) == (int(g["w"]), int(g["h"]), int(g["depth"]))
        for k in ("spheres", "lights", "planes"):
            assert wl[k].tobytes() == g[k].tobytes(), (name, k)
        assert np.array_equal(wl["camera"].position, g["cam_origin"]) and np.array_equal(wl["camera"].rotation, g["cam_rot"])
        assert (wl["amb"], wl["lamb"], wl["refl"]) == (float(g["amb"]), float(g["lamb"]), float(g["refl"]))
    wl, g = workloads.build("c5_7680x4320_s256_d8_spp4"), load_frame("c5_s256_d8_spp4_sub96")
    assert (wl["aa"], wl["spp"], wl["seed"]) == (int(g["aa"]), int(g["spp"]), int(g["seed"]))
    assert wl["spheres"].tobytes() == g["spheres"].tobytes()


def test_true_aspect_option():
    ref = Camera((1920, 1080), [-2, 0, 2.0], [0, -30, 0]).raygen()
    tru = Camera((1920, 1080), [-2, 0, 2.0], [0, -30, 0], true_aspect=True).raygen()
    assert ref[1] == 1.0 and abs(tru[1] - 16 / 9) < 1e-15 and tru[2] == (-2 * tru[1]) / 1919.0 and ref[3:] == tru[3:]
    g = Camera((32, 18), [-2, 0, 2.0], [0, -30, 0], true_aspect=True).generate_pixel_locations()
    assert g[1, 0, 0] == 32 / 18 and g[1, -1, 0] == -g[1, 0, 0] and g.raygen is not None


def test_division_by_launch_constants():
    """rt_device.h div_magic / div_by (tile index / tiles per column, block index / blocks per frame): q = mulhi(n, M) >> sh with
    M = floor(2^(31+l) / d) + 1, l = ceil(log2 d), sh = l - 1, must equal n // d for every n < 2^31 — restated here and checked
    exhaustively for small operands, at the edges, and on random ones (the GPU suite then runs the device code on odd frame sizes)."""
    import random

    def magic(d):
        if d <= 1:
            return 0, 0
        l = 0
        while (1 << l) < d:
            l += 1
        return (1 << (31 + l)) // d + 1, l - 1

    def div_by(n, d, M, sh):
        return n if d == 1 else ((n * M) >> 32) >> sh

    for d in range(1, 1200):
        M, sh = magic(d)
        assert M < 2 ** 32
        for n in list(range(0, 2500)) + [2 ** 31 - 1, 2 ** 31 - 2, 2 ** 30, d * 1000 - 1, d * 1000]:
            assert div_by(n, d, M, sh) == n // d, (n, d)
    rng = random.Random(7)
    for _ in range(300000):
        d = max(1, rng.randint(1, 2 ** rng.randint(1, 31) - 1))
        n = rng.randint(0, 2 ** 31 - 1)
        M, sh = magic(d)
        assert M < 2 ** 32 and div_by(n, d, M, sh) == n // d, (n, d)
