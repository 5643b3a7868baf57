#!/usr/bin/env python3
"""The reference driver's call sequence (peter-seres/python-ray-tracer src/main.py:8-55) against this package:
only the imports differ.  Renders the default scene and writes a PNG.

    python examples/main.py [--size 1000] [--depth 2] [--no-aa] [--out output/render.png]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from python_ray_tracer_amd import cuda                                  # reference: from numba import cuda
from python_ray_tracer_amd.ray_tracing import render                    # reference: from ray_tracing import render
from python_ray_tracer_amd.scene import Scene, Camera                   # reference: from scene import Scene, Camera
from python_ray_tracer_amd.viewer import convert_array_to_image         # reference: from viewer import convert_array_to_image


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--no-aa", action="store_true")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "output", "render.png"))
    a = ap.parse_args()

    # 1) Render and shader settings (main.py:10-12)
    w, h = a.size, a.size
    amb, lamb, refl, refl_depth = 0.0, 0.6, 0.3, a.depth
    aliasing = not a.no_aa

    # 2) Scene -> device (main.py:15-21)
    spheres_host, light_host, planes_host = Scene.default_scene().generate_scene()
    spheres, lights, planes = cuda.to_device(spheres_host), cuda.to_device(light_host), cuda.to_device(planes_host)

    # 3) Camera and rays (main.py:24-29)
    camera = Camera(resolution=(w, h), position=[-2, 0, 2.0], euler=[0, -30, 0])
    camera_origin = cuda.to_device(camera.position)
    camera_rotation = cuda.to_device(camera.rotation)
    pixel_loc = cuda.to_device(camera.generate_pixel_locations())

    # 4) Result buffer, 5) launch grid (main.py:32-38)
    result = cuda.to_device(np.zeros((3, w, h), dtype=np.uint8))
    threadsperblock = (16, 16)
    blockspergrid = (int(np.ceil(w / threadsperblock[0])), int(np.ceil(h / threadsperblock[1])))

    # 6) Launch (main.py:41-49) — with the synchronisation the reference's timing lacks
    render[blockspergrid, threadsperblock](pixel_loc, result, camera_origin, camera_rotation,
                                           spheres, lights, planes, amb, lamb, refl, refl_depth, aliasing)
    cuda.synchronize()
    st = time.time()
    render[blockspergrid, threadsperblock](pixel_loc, result, camera_origin, camera_rotation,
                                           spheres, lights, planes, amb, lamb, refl, refl_depth + 2, aliasing)
    cuda.synchronize()
    print(f"time: {1000 * (time.time() - st):,.3f} ms")

    # 7) Read back and save (main.py:51-53)
    image = convert_array_to_image(result.copy_to_host())
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    image.save(a.out)
    print("wrote", os.path.abspath(a.out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
