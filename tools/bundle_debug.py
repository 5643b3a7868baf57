#!/usr/bin/env python3
"""Bundle pre-cull vs plain per-ray cull (RT_FLAG_NO_BUNDLES): frames must be identical.  Localises a mismatch by depth
and by lights on/off.   python3 tools/bundle_debug.py [--workload c4_3840x2160_s64_d5] [--scale 4]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import python_ray_tracer_amd as pkg
from python_ray_tracer_amd import workloads, _lib as L
from python_ray_tracer_amd.scene import Camera
ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="c4_3840x2160_s64_d5"); ap.add_argument("--scale", type=int, default=4)
a = ap.parse_args()
wl = workloads.build(a.workload)
w, h = wl["w"] // a.scale, wl["h"] // a.scale
cam = Camera(resolution=(w, h), **workloads.CAMERA)
r = pkg.Renderer(0)
r.set_camera(cam.position, cam.rotation); r.set_raygen(w, h, *cam.raygen())
for lights in ("all", "none"):
    r.set_scene(wl["spheres"], wl["lights"] if lights == "all" else np.zeros((3, 0), np.float32), wl["planes"])
    for depth in range(0, wl["depth"] + 1):
        for aa in (0, 1):
            a8, a32 = r.render(0.1, wl["lamb"], wl["refl"], depth, aa, u8=True, f32=True)
            b8, b32 = r.render(0.1, wl["lamb"], wl["refl"], depth, aa, u8=True, f32=True, flags=L.RT_FLAG_NO_BUNDLES)
            bad = (a32 != b32).any(axis=0)
            msg = ""
            if bad.any():
                xs, ys = np.nonzero(bad)
                msg = f"  first at (x={xs[0]}, y={ys[0]}) tile ({xs[0]//8},{ys[0]//8}) lane {(xs[0]%8)*8+ys[0]%8}: {a32[:, xs[0], ys[0]]} vs {b32[:, xs[0], ys[0]]}"
            print(f"{a.workload}/{a.scale} lights={lights} depth={depth} aa={aa}: {int(bad.sum())} px differ{msg}", flush=True)
