#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point rt_render (launch + D2H into pageable numpy memory)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import python_ray_tracer_amd as pkg
from python_ray_tracer_amd import workloads
wl = workloads.build(workloads.HEADLINE); cam, w, h = wl["camera"], wl["w"], wl["h"]
r = pkg.Renderer(0)
r.set_scene(wl["spheres"], wl["lights"], wl["planes"]); r.set_camera(cam.position, cam.rotation); r.set_raygen(w, h, *cam.raygen())
for u8, f32 in ((True, False), (True, True)):
    for _ in range(3):
        r.render(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], 0, u8=u8, f32=f32)
    t0 = time.perf_counter(); n = 30
    for _ in range(n):
        r.render(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], 0, u8=u8, f32=f32)
    dt = (time.perf_counter() - t0) / n
    mb = (3 * w * h * (1 if u8 else 0) + 12 * w * h * (1 if f32 else 0)) / 1e6
    print(f"rt_render u8={u8} f32={f32}: {dt*1e3:.3f} ms/frame ({mb:.1f} MB to host) -> {20326104/dt/1e6:.0f} Mrays/s")
