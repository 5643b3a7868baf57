#!/usr/bin/env python3
"""One GPU standing in for each of N ranks in turn: frame period of every rank's column slab (frames round-robin on 3
streams, as bench.py runs them) with equal-width slabs and after python_ray_tracer_amd.distributed.SlabBalancer has
moved the boundaries by measured times — the same procedure bench.py runs with real ranks (bench.py:balance_slabs).
Writes gpurun_out/slab_balance.json (copied to profiles/).   python3 tools/slab_balance.py [--rounds 4]"""
import argparse, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import python_ray_tracer_amd as pkg
from python_ray_tracer_amd import workloads
from python_ray_tracer_amd.distributed import SlabBalancer

ap = argparse.ArgumentParser(); ap.add_argument("--rounds", type=int, default=4); ap.add_argument("--frames", type=int, default=600)
ap.add_argument("--workload", default=workloads.HEADLINE)
a = ap.parse_args()
wl = workloads.build(a.workload); cam, w, h = wl["camera"], wl["w"], wl["h"]
r = pkg.Renderer(0)
r.set_scene(wl["spheres"], wl["lights"], wl["planes"]); r.set_camera(cam.position, cam.rotation); r.set_raygen(w, h, *cam.raygen())
NS = 3
bufs = [(torch.empty(3 * w * h, dtype=torch.uint8, device="cuda"), torch.empty(3 * w * h, dtype=torch.float32, device="cuda")) for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
p = r.params(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], wl["aa"], spp=wl["spp"], seed=wl["seed"])


def slab_ms(x0, x1, n):
    def burst(k):
        for i in range(k):
            r.render_device(p, x0, x1, bufs[i % NS][0].data_ptr(), bufs[i % NS][1].data_ptr(), (x1 - x0) * h, stream=streams[i % NS].cuda_stream)
    burst(2); torch.cuda.synchronize(); burst(60); torch.cuda.synchronize()
    t = time.perf_counter(); burst(n); torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for _ in range(300):                                  # clocks
    r.render_device(p, 0, w, bufs[0][0].data_ptr(), bufs[0][1].data_ptr(), w * h, stream=streams[0].cuda_stream)
torch.cuda.synchronize()
out = {"workload": a.workload, "streams": NS, "frames_per_measurement": a.frames, "worlds": {}}
for world in (2, 4, 8):
    bal = SlabBalancer(w, world)
    hist = []
    for it in range(a.rounds + 1):
        times = [slab_ms(x0, x1, a.frames) for x0, x1 in bal.bounds]
        hist.append({"bounds": [list(b) for b in bal.bounds], "slab_ms": [round(t, 5) for t in times],
                     "max_over_mean": round(max(times) / (sum(times) / world), 4), "max_ms": round(max(times), 5)})
        print(f"world {world} round {it}: max/mean {hist[-1]['max_over_mean']:.4f}  max {max(times):.4f} ms  widths {[b - a_ for a_, b in bal.bounds]}", flush=True)
        if it < a.rounds:
            bal.update(times)
    best = min(hist, key=lambda e: e["max_over_mean"])
    out["worlds"][str(world)] = {"equal_width": hist[0], "balanced": best, "history": hist}
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "slab_balance.json"), "w"), indent=1)
