#!/usr/bin/env python3
"""Render a frame with the `Renderer` API and save it as a PNG.

    python examples/render_png.py [--size 1000x1000] [--depth 4] [--aa] [--frames 50] [--out output/render.png]

The device writes the interleaved (h, w, 3) image directly (RT_FLAG_U8_HWC | RT_FLAG_U8_RGB) into page-locked host
memory; the frame time is measured with HIP events over `--frames` launches.  For the numba-shaped call the
reference's driver makes, see INTEGRATION.md §1 and tests/test_gpu_parity.py::test_facade_matches_reference_call_shape.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import python_ray_tracer_amd as pkg
from python_ray_tracer_amd import _lib as L
from python_ray_tracer_amd.scene import Scene, Camera
from python_ray_tracer_amd.viewer import convert_array_to_image


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1000x1000")
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--aa", action="store_true", help="the reference's 9-tap anti-aliasing")
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "output", "render.png"))
    a = ap.parse_args()
    w, h = (int(v) for v in a.size.lower().split("x"))
    cam = Camera(resolution=(w, h), position=[-2, 0, 2.0], euler=[0, -30, 0])
    with pkg.Renderer(0) as r:
        r.set_scene(*Scene.default_scene().generate_scene())
        r.set_camera(cam.position, cam.rotation)
        r.set_raygen(w, h, *cam.raygen())
        image = r.host_array((h, w, 3), np.uint8)
        flags = L.RT_FLAG_U8_HWC | L.RT_FLAG_U8_RGB
        r.render_into(0.0, 0.6, 0.3, a.depth, a.aa, image, flags=flags)
        dev = r.malloc(3 * w * h)
        p = r.params(0.0, 0.6, 0.3, a.depth, a.aa, flags=flags)
        for _ in range(3):
            r.render_device(p, 0, w, dev, None, w)
        r.timer_begin()
        for _ in range(a.frames):
            r.render_device(p, 0, w, dev, None, w)
        ms = r.timer_end() / a.frames
        r.free(dev)
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        convert_array_to_image(np.array(image)).save(a.out)
        print(f"{w}x{h} depth {a.depth} aa={a.aa}: {ms:.4f} ms per frame on the device; wrote {a.out}")


if __name__ == "__main__":
    main()
