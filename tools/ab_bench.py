#!/usr/bin/env python3
"""Interleaved A/B timing of several builds of libmi355rt.so in ONE process on ONE device
(cdna_hip_programming.md §5.4 rule 24): tools/ab_bench.py a.so b.so [--workload NAME] [--rounds 15].
Prints per build the median / min kernel time (hipEvent pair around `--launches` back-to-back launches)
and checks that every build produces the same frame bytes."""
import argparse
import hashlib
import json
import os
import statistics
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # --streams: keep the streams on distinct hardware queues (see bench.py)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--workload", default=None)
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--aa", action="store_true")
    ap.add_argument("--streams", type=int, default=1, help=">1: frames round-robin on that many streams, wall-clock per frame")
    a = ap.parse_args()
    import numpy as np
    import python_ray_tracer_amd as pkg
    from python_ray_tracer_amd import workloads, _lib
    wl = workloads.build(a.workload or workloads.HEADLINE)
    cam, w, h = wl["camera"], wl["w"], wl["h"]
    rs = []
    flags = {}
    for path in a.libs:                      # "build.so" or "build.so:FLAGS" (rt_params.flags, e.g. :4 = no scheduler feedback)
        so, _, fl = path.partition(":")
        flags[path] = int(fl or 0)
        r = pkg.Renderer(0, lib=_lib.bind(os.path.abspath(so)))
        r.set_scene(wl["spheres"], wl["lights"], wl["planes"]); r.set_camera(cam.position, cam.rotation); r.set_raygen(w, h, *cam.raygen())
        d8, d32 = r.malloc(3 * w * h), r.malloc(12 * w * h)
        if a.streams > 1:
            r._ab_streams = [r.stream_create() for _ in range(a.streams)]
            r._ab_bufs = [(d8, d32)] + [(r.malloc(3 * w * h), r.malloc(12 * w * h)) for _ in range(a.streams - 1)]
        rs.append((path, r, d8, d32))
    ps = {path: pkg.Renderer.params(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], 1 if a.aa else wl["aa"], flags=flags[path],
                                    spp=wl["spp"], seed=wl["seed"]) for path in a.libs}
    times = {path: [] for path, *_ in rs}
    for rnd in range(a.rounds + 2):
        for path, r, d8, d32 in rs:
            if a.streams > 1:
                import time
                for s_ in r._ab_streams:
                    r.sync(s_)
                t0 = time.perf_counter()
                for i in range(a.launches):
                    b8, b32 = r._ab_bufs[i % a.streams]
                    r.render_device(ps[path], 0, w, b8, b32, w * h, stream=r._ab_streams[i % a.streams])
                for s_ in r._ab_streams:
                    r.sync(s_)
                ms = (time.perf_counter() - t0) * 1e3 / a.launches
            else:
                r.timer_begin()
                for _ in range(a.launches):
                    r.render_device(ps[path], 0, w, d8, d32, w * h)
                ms = r.timer_end() / a.launches
            if rnd >= 2:
                times[path].append(ms)
    out = {}
    for path, r, d8, d32 in rs:
        host = np.empty((3, w, h), np.uint8); r.d2h(host, d8)
        h32 = np.empty((3, w, h), np.float32); r.d2h(h32, d32)
        t = times[path]
        out[path] = dict(median_ms=round(statistics.median(t), 5), min_ms=round(min(t), 5), max_ms=round(max(t), 5),
                         sha_u8=hashlib.sha256(host.tobytes()).hexdigest()[:16], sha_f32=hashlib.sha256(h32.tobytes()).hexdigest()[:16],
                         vgprs=r.kernel_info()["vgprs"])
    base = out[a.libs[0]]["median_ms"]
    for path in a.libs:
        out[path]["vs_first"] = round(out[path]["median_ms"] / base, 4)
        print(path, json.dumps(out[path]))
    shas = {(v["sha_u8"], v["sha_f32"]) for v in out.values()}
    print("frames identical across builds:", len(shas) == 1)


if __name__ == "__main__":
    main()
