#!/usr/bin/env python3
"""Measurement build only (hipcc -DRT_LANE_STATS -> build_variants/stats.so, MI355RT_SO=that): how the lane-owned
traversal's loops are filled.  Per wave-query: clusters opened — the per-wave MAXIMUM over the lanes (= iterations the wave
runs now) against ceil(total / 64) (= rounds if (ray, cluster) pairs were spread over all lanes); the same for the spheres
that reach the float64 test inside every cluster iteration."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import python_ray_tracer_amd as pkg
from python_ray_tracer_amd import workloads

name = sys.argv[1] if len(sys.argv) > 1 else "c5_7680x4320_s256_d8"
wl = workloads.build(name); cam, w, h = wl["camera"], wl["w"], wl["h"]
r = pkg.Renderer(0)
r.set_scene(wl["spheres"], wl["lights"], wl["planes"]); r.set_camera(cam.position, cam.rotation); r.set_raygen(w, h, *cam.raygen())
ntiles = ((w + 7) // 8) * ((h + 7) // 8)
nb = 4 * (ntiles + 16)
d = r.malloc(nb); r.h2d(d, np.zeros(ntiles + 16, np.uint32))
d8 = r.malloc(3 * w * h)
r.set_tile_stats(d)
p = r.params(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], 0, flags=4)
r.render_device(p, 0, w, d8, None, w * h); r.sync()
out = np.empty(ntiles + 16, np.uint32); r.d2h(out, d)
c = out[ntiles:].astype(np.int64)
for label, b in (("closest: clusters", 0), ("closest: f64 spheres per cluster iteration", 4), ("shadow: clusters", 8), ("shadow: f64 spheres per cluster iteration", 12)):
    calls, mx, rounds, tot = c[b:b + 4]
    print(f"{label:46s} calls {calls:10d}  sum(max over lanes) {mx:10d} ({mx / max(calls, 1):.2f}/call)  sum(ceil(total/64)) {rounds:10d} ({rounds / max(calls, 1):.2f}/call)  total {tot}  ({tot / max(calls, 1):.1f}/call)")
