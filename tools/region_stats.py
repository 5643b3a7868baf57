#!/usr/bin/env python3
"""Measurement build only (hipcc -DRT_REGION_STATS -> build_variants/regions.so, MI355RT_SO=that): where the render
kernel's wave cycles go, by code region of trace_bounce, separately for bounces 0-1 and bounces 2 and later.
    MI355RT_SO=build_variants/regions.so python tools/region_stats.py [workload]"""
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
d = r.malloc(4 * (ntiles + 64)); r.h2d(d, np.zeros(ntiles + 64, np.uint32))
d8 = r.malloc(3 * w * h)
p = r.params(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], wl["aa"], flags=4, spp=wl["spp"], seed=wl["seed"])
r.render_device(p, 0, w, d8, None, w * h); r.sync()            # warm (tables)
r.set_tile_stats(d)
r.render_device(p, 0, w, d8, None, w * h); r.sync()
out = np.empty(ntiles + 64, np.uint32); r.d2h(out, d)
c = out[ntiles + 16: ntiles + 40].astype(np.float64) * 64
names = ["re-normalise (closest)", "closest: cluster bounds", "closest: cluster loop (f32 + f64 sphere tests)", "closest: planes, select", "hit point, normal",
         "light direction, Lambert", "re-normalise (shadow)", "shadow: cluster bounds", "shadow: cluster loop", "shadow: planes", "reflection", "rest (ray gen, loop, store)"]
tot = c.sum()
print(f"{name}: {tot / 1e9:.2f} G wave-cycles in marked regions (tiles' cycles: {out[:ntiles].astype(np.float64).sum() / 1e9:.2f} G)")
for cls, off in (("bounces 0-1", 0), ("bounces 2+", 12)):
    print(f"-- {cls}: {c[off:off + 12].sum() / tot * 100:.1f} % of the wave cycles")
    for i, n in enumerate(names):
        print(f"   {n:48s} {c[off + i] / tot * 100:6.2f} %")
