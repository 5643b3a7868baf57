#!/usr/bin/env python3
"""Launches the headline frame in five configurations (full, no lights, depth 0, depth 0 + no lights, no spheres)
so that `rocprofv3 --pmc SQ_INSTS_VALU -- python3 tools/instr_breakdown.py` attributes the kernel's VALU
instructions to bounces / light loops / sphere work by differences (dispatch order = the order printed)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import python_ray_tracer_amd as pkg
from python_ray_tracer_amd import workloads

wl = workloads.build(workloads.HEADLINE); cam, w, h = wl["camera"], wl["w"], wl["h"]
r = pkg.Renderer(0)
r.set_camera(cam.position, cam.rotation); r.set_raygen(w, h, *cam.raygen())
d8, d32 = r.malloc(3 * w * h), r.malloc(12 * w * h)
nol = np.zeros((3, 0), np.float32); nos = np.zeros((7, 0), np.float32)
for name, sp, li, depth in (("full", wl["spheres"], wl["lights"], 3), ("no_lights", wl["spheres"], nol, 3),
                            ("depth0", wl["spheres"], wl["lights"], 0), ("depth0_no_lights", wl["spheres"], nol, 0),
                            ("no_spheres", nos, wl["lights"], 3)):
    r.set_scene(sp, li, wl["planes"])
    p = r.params(wl["amb"], wl["lamb"], wl["refl"], depth, 0, flags=4)
    r.render_device(p, 0, w, d8, d32, w * h); r.sync()
    print(name)
