#!/usr/bin/env python3
"""Frame period of one GPU rendering only its column slab (world 1/2/4/8): a launch per frame on 1..6 streams (round 2's
mode), and whole batches of 8 / 16 frames per launch through rt_render_sequence (round 3) on 1 and 3 streams: the tables in
DESIGN.md §6.  Export GPU_MAX_HW_QUEUES=8 to keep the streams on distinct hardware queues."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import python_ray_tracer_amd as pkg
from python_ray_tracer_amd import workloads
from python_ray_tracer_amd.distributed import slab_bounds
wl = workloads.build(workloads.HEADLINE); cam, w, h = wl["camera"], wl["w"], wl["h"]
r = pkg.Renderer(0)
r.set_scene(wl["spheres"], wl["lights"], wl["planes"]); r.set_camera(cam.position, cam.rotation); r.set_raygen(w, h, *cam.raygen())
bufs = [(torch.empty(3 * w * h, dtype=torch.uint8, device="cuda"), torch.empty(3 * w * h, dtype=torch.float32, device="cuda")) for _ in range(6)]
streams = [torch.cuda.Stream() for _ in range(6)]
p = r.params(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], wl["aa"])
def run(x0, x1, ns, n=2000):
    ws = x1 - x0
    for i in range(100):
        r.render_device(p, x0, x1, bufs[i % ns][0].data_ptr(), bufs[i % ns][1].data_ptr(), ws * h, stream=streams[i % ns].cuda_stream)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n):
        r.render_device(p, x0, x1, bufs[i % ns][0].data_ptr(), bufs[i % ns][1].data_ptr(), ws * h, stream=streams[i % ns].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
F = 16
seq8 = [torch.empty(F * 3 * w * h, dtype=torch.uint8, device="cuda") for _ in range(3)]
seq32 = [torch.empty(F * 3 * w * h, dtype=torch.float32, device="cuda") for _ in range(3)]
def run_seq(x0, x1, ns, fpl, n=1920):
    """batches of fpl frames, one rt_render_sequence call (= one kernel launch) each, round-robin on ns streams"""
    ws = x1 - x0
    def go(frames):
        for b in range(frames // fpl):
            r.render_sequence(p, x0, x1, fpl, seq8[b % ns].data_ptr(), seq32[b % ns].data_ptr(), ws * h, 3 * ws * h, None, (streams[b % ns].cuda_stream,), fpl)
    go(6 * fpl)
    torch.cuda.synchronize(); t = time.perf_counter()
    go(n)
    t_sub = time.perf_counter() - t
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, t_sub / n * 1e3
for world in (1, 2, 4, 8):
    for rank in sorted({0, world // 2, world - 1}):
        x0, x1 = slab_bounds(w, world, rank)
        print(f"world {world} rank {rank} [{x0},{x1}): " + "  ".join(f"{ns}s {run(x0, x1, ns):.4f}" for ns in (1, 2, 3, 4, 6)), flush=True)
        print(f"world {world} rank {rank} [{x0},{x1}) batched (ms per frame / host ms per frame): " +
              "  ".join(f"{ns}s x{fpl}: {a_:.4f} / {b_:.5f}" for ns, fpl in ((1, 8), (3, 8), (1, 16), (3, 16)) for a_, b_ in (run_seq(x0, x1, ns, fpl),)), flush=True)
