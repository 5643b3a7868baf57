#!/usr/bin/env python3
"""Per-tile wave cycles of one frame (rt_set_tile_stats) -> gpurun_out/tile_cycles_<workload>.npy, plus a
replay of the hardware's greedy dispatch (W wave slots) in launch order vs longest-first order."""
import heapq, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import python_ray_tracer_amd as pkg
from python_ray_tracer_amd import workloads

name = sys.argv[1] if len(sys.argv) > 1 else workloads.HEADLINE
wl = workloads.build(name); cam, w, h = wl["camera"], wl["w"], wl["h"]
r = pkg.Renderer(0)
r.set_scene(wl["spheres"], wl["lights"], wl["planes"]); r.set_camera(cam.position, cam.rotation); r.set_raygen(w, h, *cam.raygen())
tx, ty = (w + 7) // 8, (h + 7) // 8
d8, d32, dc = r.malloc(3 * w * h), r.malloc(12 * w * h), r.malloc(4 * tx * ty)
p = r.params(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], wl["aa"], spp=wl["spp"], seed=wl["seed"])
r.set_tile_stats(dc)
for _ in range(3):
    r.render_device(p, 0, w, d8, d32, w * h)
r.sync()
c = np.empty(tx * ty, np.uint32); r.d2h(c, dc)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
np.save(os.path.join(REPO, "gpurun_out", f"tile_cycles_{name}.npy"), c.reshape(tx, ty))
c = c.astype(np.float64)
print(f"{name}: tiles {c.size}, cycles mean {c.mean():.0f} median {np.median(c):.0f} p90 {np.percentile(c, 90):.0f} max {c.max():.0f}; sum/mean-slot {c.sum():.3e}")


def replay(costs, slots):
    heap = [0.0] * slots
    for x in costs:
        t = heapq.heappop(heap)
        heapq.heappush(heap, t + x)
    return max(heap)


for slots in (256 * 4 * 7, 256 * 4 * 5):
    base = replay(c, slots); lpt = replay(np.sort(c)[::-1], slots); ideal = c.sum() / slots
    print(f"  slots {slots}: launch order {base:.0f}  longest-first {lpt:.0f}  perfectly balanced {ideal:.0f}  (makespan in wave-cycles; ratio {base/lpt:.3f})")
