#!/usr/bin/env python3
"""Kernel time of one workload as a function of the reflection depth (and optionally without shadow-casting lights):
where a deep configuration (BASELINE configs 4 and 5) spends its time.  One stream, HIP events, settled dispatch order.
    python3 tools/depth_sweep.py --workload c5_7680x4320_s256_d8 [--launches 5]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import python_ray_tracer_amd as pkg
from python_ray_tracer_amd import workloads, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c5_7680x4320_s256_d8")
ap.add_argument("--launches", type=int, default=5)
a = ap.parse_args()
wl = workloads.build(a.workload); cam, w, h = wl["camera"], wl["w"], wl["h"]
r = pkg.Renderer(0)
r.set_camera(cam.position, cam.rotation); r.set_raygen(w, h, *cam.raygen())
d8, d32 = r.malloc(3 * w * h), r.malloc(12 * w * h)
for lights in ("all", "none"):
    li = wl["lights"] if lights == "all" else np.zeros((3, 0), np.float32)
    r.set_scene(wl["spheres"], li, wl["planes"])
    prev = 0.0
    for depth in range(0, wl["depth"] + 1):
        p = r.params(0.1 if lights == "none" else wl["amb"], wl["lamb"], wl["refl"], depth, wl["aa"], spp=wl["spp"], seed=wl["seed"])
        for _ in range(3):
            r.render_device(p, 0, w, d8, d32, w * h)
        r.sync()
        r.timer_begin()
        for _ in range(a.launches):
            r.render_device(p, 0, w, d8, d32, w * h)
        ms = r.timer_end() / a.launches
        r.reset_stats()
        pc = r.params(0.1 if lights == "none" else wl["amb"], wl["lamb"], wl["refl"], depth, wl["aa"], spp=wl["spp"], seed=wl["seed"], flags=_lib.RT_FLAG_COUNT_RAYS)
        r.render_device(pc, 0, w, d8, d32, w * h); r.sync()
        st = r.stats()
        print(f"{a.workload} lights={lights} depth={depth}: {ms:.4f} ms (+{ms - prev:.4f})  closest {st['closest_queries']/w/h:.3f}/px "
              f"shadow traced {st['shadow_traced']/w/h:.3f}/px skipped {st['shadow_skipped']/w/h:.3f}/px hits {st['hits']/w/h:.3f}/px", flush=True)
        prev = ms

st = r.stats()
print("lane utilisation per bounce (last configuration):", " ".join(f"b{b}:{st['bounce_lanes'][b] / max(64 * st['bounce_waves'][b], 1):.3f}({st['bounce_waves'][b]})" for b in range(wl["depth"] + 1)))
