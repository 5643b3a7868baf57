"""The synthetic workloads BASELINE.json names, as scene arrays in the reference's
own formats.  Camera, shader scalars, lights and ground plane are the reference driver's literals
(main.py:11,24; scene/scene.py:102-113).  tests/test_host_helpers.py checks these arrays against the
committed golden fixtures, so bench.py, the tests and the goldens all describe the same scenes."""
import numpy as np

from .scene import Scene, Sphere, Camera
from .scene.colors import RED, GREEN, BLUE, YELLOW, GREY, MAGENTA

PALETTE = [RED, BLUE, YELLOW, MAGENTA, GREEN, GREY]
CAMERA = dict(position=[-2, 0, 2.0], euler=[0, -30, 0], fov=45.0)          # main.py:24
SHADER = dict(amb=0.0, lamb=0.6, refl=0.3)                                   # main.py:11
EXTRA_SPHERES = [([0.9, -2.0, 0.6], 0.6, GREEN), ([3.5, -1.6, 0.8], 0.8, BLUE)]


def grid_spheres(n_side, seed):
    """n_side x n_side jittered grid of spheres resting on the ground plane (configs 4 and 5)."""
    rng = np.random.default_rng(seed)
    out, span_x, span_y = [], 9.0, 8.0
    for i in range(n_side):
        for j in range(n_side):
            r = float(rng.integers(2, 9)) / 16.0 * (8.0 / n_side)
            x = -0.5 + span_x * (i + 0.5) / n_side + float(rng.uniform(-0.25, 0.25)) * (8.0 / n_side)
            y = -span_y / 2 + span_y * (j + 0.5) / n_side + float(rng.uniform(-0.25, 0.25)) * (8.0 / n_side)
            out.append(Sphere([x, y, r], r, PALETTE[(i * n_side + j) % len(PALETTE)]))
    return out


def _scene(spheres):
    base = Scene.default_scene()
    return Scene(base.lights, spheres, base.planes)


# name -> (w, h, depth, aa, scene factory, rays per frame {closest, shadow} or None)
# Ray counts are the oracle's counters for the same frame (tests/test_oracle_golden.py re-derives C1/C2).
CONFIGS = {
    "c1_128x128_s3_d1": (128, 128, 1, False, lambda: _scene(Scene.default_scene().spheres[:3]), None),
    "c2_1920x1080_s8_d3": (1920, 1080, 3, False,
                           lambda: _scene(Scene.default_scene().spheres + [Sphere(*s) for s in EXTRA_SPHERES]),
                           dict(closest=6309069, shadow=14017035)),
    # (the oracle's counters for these frames; the device's counting instantiation reports the same numbers —
    # tests/test_gpu_parity.py::test_ray_counters_match_oracle; config 4 is re-derived in tests/test_oracle_golden.py, config 5 at
    # 1 spp was re-derived with the oracle once, 100 s on 8 cores, and both config-5 rows are checked against the device counters)
    "c4_3840x2160_s64_d5": (3840, 2160, 5, False, lambda: _scene(grid_spheres(8, 355)), dict(closest=22674256, shadow=44160822)),
    "c5_7680x4320_s256_d8": (7680, 4320, 8, False, lambda: _scene(grid_spheres(16, 356)), dict(closest=106764199, shadow=224018220)),
    # BASELINE config 5 proper: 4 stochastic samples per pixel (aa mode 2 = RT_AA_STOCHASTIC, seed 1)
    "c5_7680x4320_s256_d8_spp4": (7680, 4320, 8, 2, lambda: _scene(grid_spheres(16, 356)),
                                  dict(closest=427049977, shadow=896048151)),
}
# not BASELINE configurations: scene sizes between them, for choosing the kernel variants' thresholds (tools/ab_bench.py)
CONFIGS.update({
    "x_1920x1080_s16_d3": (1920, 1080, 3, False, lambda: _scene(grid_spheres(4, 364)), None),
    "x_1920x1080_s25_d3": (1920, 1080, 3, False, lambda: _scene(grid_spheres(5, 360)), None),
    "x_1920x1080_s36_d3": (1920, 1080, 3, False, lambda: _scene(grid_spheres(6, 361)), None),
    "x_1920x1080_s49_d3": (1920, 1080, 3, False, lambda: _scene(grid_spheres(7, 362)), None),
    "x_3840x2160_s49_d5": (3840, 2160, 5, False, lambda: _scene(grid_spheres(7, 362)), None),
    "x_3840x2160_s36_d5": (3840, 2160, 5, False, lambda: _scene(grid_spheres(6, 361)), None),
    "x_7680x4320_s196_d8": (7680, 4320, 8, False, lambda: _scene(grid_spheres(14, 359)), None),
    "x_3840x2160_s256_d5": (3840, 2160, 5, False, lambda: _scene(grid_spheres(16, 356)), None),
    "x_3840x2160_s169_d5": (3840, 2160, 5, False, lambda: _scene(grid_spheres(13, 363)), None),
    "x_3840x2160_s100_d5": (3840, 2160, 5, False, lambda: _scene(grid_spheres(10, 357)), None),
    "x_3840x2160_s144_d5": (3840, 2160, 5, False, lambda: _scene(grid_spheres(12, 358)), None),
    "x_1920x1080_s196_d3": (1920, 1080, 3, False, lambda: _scene(grid_spheres(14, 359)), None),
})
STOCHASTIC = {"c5_7680x4320_s256_d8_spp4": dict(spp=4, seed=1)}
HEADLINE = "c2_1920x1080_s8_d3"


def camera_path(n, period=240):
    """float64 (n, 12): origin[3] + rotation[9] per frame of a camera that moves with EVERY frame around the reference
    driver's pose (main.py:24) — position on a small loop, pitch and yaw swinging a few degrees — for bench.py's `dynamic`
    block and the moving-camera parity tests (README.md:23 "real-time display"; scene/camera.py:8-16)."""
    from .scene.rotation import euler_rotation
    out = np.empty((n, 12), np.float64)
    for i in range(n):
        t = 2.0 * np.pi * i / period
        out[i, 0:3] = [CAMERA["position"][0] + 0.30 * np.sin(t), CAMERA["position"][1] + 0.25 * np.sin(2 * t),
                       CAMERA["position"][2] + 0.10 * (1.0 - np.cos(t))]
        out[i, 3:12] = euler_rotation(1.5 * np.sin(3 * t), CAMERA["euler"][1] + 2.0 * np.sin(t), 3.0 * np.sin(2 * t)).reshape(9)
    return out


def build(name):
    """-> dict(w, h, depth, aa, spheres, lights, planes, camera, rays)"""
    w, h, depth, aa, make, rays = CONFIGS[name]
    spheres, lights, planes = make().generate_scene()
    cam = Camera(resolution=(w, h), **CAMERA)
    st = STOCHASTIC.get(name, dict(spp=0, seed=1))
    return dict(name=name, w=w, h=h, depth=depth, aa=aa, spheres=spheres, lights=lights, planes=planes,
                camera=cam, rays=rays, **st, **SHADER)
