#!/usr/bin/env python3
"""tools/ab_bench.py on 1080p depth-3 scenes of 9/16/25/36 spheres (--workload mid_s9 ... mid_s36): the sweep behind
the workgroup-size threshold in mi355rt.hip."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from python_ray_tracer_amd import workloads as wl
for n in (3, 4, 5, 6):
    wl.CONFIGS[f"mid_s{n*n}"] = (1920, 1080, 3, False, (lambda n=n: wl._scene(wl.grid_spheres(n, 100 + n))), dict(closest=1, shadow=1))
import ab_bench
ab_bench.main()
