import glob
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def frame_cases():
    return sorted(os.path.basename(p)[len("frame_"):-len(".npz")] for p in glob.glob(os.path.join(GOLDEN, "frame_*.npz")))


def load_frame(name):
    return np.load(os.path.join(GOLDEN, f"frame_{name}.npz"))


def raygen_closed_form(w, h, fov):
    """scene/camera.py:18-26 as np.mgrid evaluates it: value = index*step + start."""
    ar = int(w / h)
    px = float(1 / np.tan(np.radians(fov) / 2))
    return (px, float(ar), (-ar - ar) / float(w - 1), 1.0, (-1 - 1) / float(h - 1))


def clamp255(a):
    return np.clip(a, 0.0, 255.0)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def renderer():
    """One rt_ctx on cuda:0 for the whole GPU session; fails (does not skip) without the HIP library."""
    import python_ray_tracer_amd as pkg
    r = pkg.Renderer(0)
    yield r
    r.close()
