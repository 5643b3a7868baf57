#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point rt_render (launch + D2H + sync per frame) for 1/2/4/8 column
chunks (MI355RT_CHUNKS), pageable and page-locked destinations; plus the bare device-to-host copy of a frame."""
import os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import python_ray_tracer_amd as pkg
    from python_ray_tracer_amd import workloads
    wl = workloads.build(workloads.HEADLINE); cam, w, h = wl["camera"], wl["w"], wl["h"]
    r = pkg.Renderer(0)
    r.set_scene(wl["spheres"], wl["lights"], wl["planes"]); r.set_camera(cam.position, cam.rotation); r.set_raygen(w, h, *cam.raygen())
    res = []
    for want32 in (False, True):
        for pinned in (False, True):
            bufs = r.host_arrays(want32, pinned=pinned)
            for _ in range(5):
                r.render_into(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], 0, *bufs)
            t0 = time.perf_counter(); n = 60
            for _ in range(n):
                r.render_into(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], 0, *bufs)
            res.append(f"{'u8+f32' if want32 else 'u8'}/{'pinned' if pinned else 'pageable'} {(time.perf_counter() - t0) / n * 1e3:.3f}")
            r.release_host_arrays(bufs)
    # frame sequences over 1..4 slots (rt_render_begin / rt_render_end), page-locked destinations
    for want32 in (False, True):
        for slots in (1, 2, 3, 4):
            bufs = [r.host_arrays(want32, pinned=True) for _ in range(slots)]
            def seq(n):
                for i in range(n):
                    if i >= slots:
                        r.render_end(i % slots)
                    r.render_begin(i % slots, wl["amb"], wl["lamb"], wl["refl"], wl["depth"], 0, *bufs[i % slots])
                for sl in range(slots):
                    r.render_end(sl)
            seq(8)
            t0 = time.perf_counter(); n = 120
            seq(n)
            res.append(f"{'u8+f32' if want32 else 'u8'}/sequence{slots} {(time.perf_counter() - t0) / n * 1e3:.3f}")
            for b in bufs:
                r.release_host_arrays(b)
    # the bare copy of the uint8 frame
    d = r.malloc(3 * w * h)
    for pinned in (False, True):
        a = r.host_array((3, w, h), np.uint8) if pinned else np.empty((3, w, h), np.uint8)
        for _ in range(3): r.d2h(a, d)
        t0 = time.perf_counter()
        for _ in range(30): r.d2h(a, d)
        res.append(f"bare-d2h-u8/{'pinned' if pinned else 'pageable'} {(time.perf_counter() - t0) / 30 * 1e3:.3f}")
    print(f"chunks={os.environ.get('MI355RT_CHUNKS')}: " + "  ".join(res), flush=True)
else:
    for mode in ("-1", "0", "1"):                      # -1: by destination memory type (the default)
        for ch in ("1", "4"):
            print("mode", mode, end=" ", flush=True)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, MI355RT_CHUNKS=ch, MI355RT_CHUNK_MODE=mode))
