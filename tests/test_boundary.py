"""The drop-in boundary without a GPU: the C-ABI library loads, exports every symbol that
include/mi355rt.h declares (and the ctypes table binds exactly that set), the product never touches
the oracle or a CPU fallback, and it fails loudly when no HIP device is present."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from conftest import REPO

HEADER = os.path.join(REPO, "include", "mi355rt.h")
PKG = os.path.join(REPO, "python-ray-tracer_amd")


def declared_functions():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from python_ray_tracer_amd import _lib
    names = declared_functions()
    assert len(names) >= 15
    lib = ctypes.CDLL(_lib.SO_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mi355rt.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names, "ctypes table and header disagree"
    assert _lib.load().rt_abi_version() == _lib.RT_ABI_VERSION


def test_header_constants_match_python():
    from python_ray_tracer_amd import _lib
    src = open(HEADER).read()
    for name in ("RT_ABI_VERSION", "RT_MAX_DEPTH", "RT_MAX_SPHERES", "RT_MAX_PLANES", "RT_MAX_LIGHTS",
                 "RT_AA_NONE", "RT_AA_REFERENCE", "RT_AA_STOCHASTIC", "RT_MAX_SPP", "RT_RENDER_SLOTS", "RT_FLAG_TYPED_BIAS", "RT_FLAG_U8_RGB"):
        m = re.search(rf"#define\s+{name}\s+(-?\d+)", src)
        assert m and int(m.group(1)) == getattr(_lib, name), name
    for name, val in re.findall(r"(RT_(?:OK|ERR_[A-Z_]+))\s*=\s*(-?\d+)", src):
        assert getattr(_lib, name) == int(val)
    assert ctypes.sizeof(_lib.rt_params) == 8 * 2 + 8 * 16 + 4 * 6


def test_header_compiles_as_c():
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", HEADER])


def test_product_never_uses_the_oracle_or_a_cpu_path():
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\.*oracle", text, flags=re.M), f"{f} imports the oracle"
                assert "librt_oracle" not in text and "rt_oracle" not in text and "orc_" not in text, f
    out = subprocess.check_output(["ldd", os.path.join(PKG, "libmi355rt.so")], text=True)
    assert "oracle" not in out and "libamdhip64" in out


def test_no_device_is_a_loud_error():
    """On the CPU container there is no HIP device: creating a renderer must raise, not fall back."""
    import python_ray_tracer_amd as pkg
    from python_ray_tracer_amd.renderer import device_count
    if device_count() > 0:
        pytest.skip("a GPU is present; the error path is exercised in test_gpu_parity.py")
    with pytest.raises(pkg.RenderError) as e:
        pkg.Renderer(0)
    assert e.value.status == -3
    from python_ray_tracer_amd.ray_tracing import render
    import numpy as np
    with pytest.raises(pkg.RenderError):
        render[(1, 1), (8, 8)](np.zeros((3, 8, 8)), np.zeros((3, 8, 8), np.uint8), np.zeros(3), np.eye(3),
                               np.zeros((7, 0), np.float32), np.zeros((3, 0), np.float32), np.zeros((9, 0), np.float32),
                               0.0, 0.6, 0.3, 1, False)


def test_missing_library_is_a_loud_error(tmp_path):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import python_ray_tracer_amd._lib as L\n"
            "L.SO_PATH = %r\n"
            "try:\n    L.load()\nexcept ImportError as e:\n    print('LOUD', e)\n") % (REPO, str(tmp_path / "nope.so"))
    out = subprocess.check_output([sys.executable, "-c", code], text=True)
    assert "LOUD" in out and "no CPU fallback" in out


def test_facade_argument_validation():
    from python_ray_tracer_amd.ray_tracing import render
    with pytest.raises(TypeError):
        render(1, 2, 3)
    with pytest.raises(ValueError):
        render[(1, 1)]
    with pytest.raises(ValueError):
        render[(1, 1), (0, 8)]


class _FakeRenderer:
    """Records what the facade sends (no GPU): stands in for python_ray_tracer_amd.Renderer."""

    def __init__(self):
        from python_ray_tracer_amd.renderer import Renderer
        self.serial = -1
        self.generation = {"scene": 0, "camera": 0, "grid": 0}
        self.scenes, self.cameras, self.grids = [], [], []
        self.w = self.h = None
        self.params = Renderer.params

    def set_scene(self, s, l, p, flags=0):
        self.scenes.append(float(s[0, 0])); self.generation["scene"] += 1

    def set_camera(self, o, r):
        self.cameras.append(float(o[0])); self.generation["camera"] += 1

    def set_raygen(self, w, h, *a):
        self.w, self.h = w, h; self.grids.append(("closed", w, h)); self.generation["grid"] += 1

    def set_pixel_loc(self, a):
        self.w, self.h = a.shape[1], a.shape[2]; self.grids.append(("explicit",) + a.shape[1:]); self.generation["grid"] += 1

    def render(self, *a, x0=0, x1=None, **k):
        import numpy as np
        return np.zeros((3, x1 - x0, self.h), np.uint8), None


def test_facade_never_aliases_fresh_handles(monkeypatch):
    """a scene built per frame through fresh cuda.to_device handles (whose id() CPython reuses) must be
    re-sent every frame; the same handles again must not be; a changed camera handle must be."""
    import numpy as np
    from python_ray_tracer_amd import cuda
    from python_ray_tracer_amd.ray_tracing.kernels import RenderKernel
    fake = _FakeRenderer()
    monkeypatch.setattr(cuda, "_default", fake)
    render = RenderKernel()
    w = h = 8
    grid = cuda.to_device(np.zeros((3, w, h)))
    cam_o, cam_R = cuda.to_device(np.zeros(3)), cuda.to_device(np.eye(3))

    def build_scene(i):
        sp = np.zeros((7, 1), np.float32); sp[0, 0] = i
        return cuda.to_device(sp), cuda.to_device(np.zeros((3, 1), np.float32)), cuda.to_device(np.zeros((9, 1), np.float32))

    for i in range(6):                                   # handles die at the end of every iteration
        render[(1, 1), (8, 8)](grid, np.zeros((3, w, h), np.uint8), cam_o, cam_R, *build_scene(i), 0.0, 0.6, 0.3, 1, False)
    assert fake.scenes == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]
    assert len(fake.cameras) == 1 and len(fake.grids) == 1
    keep = build_scene(9)
    for _ in range(3):
        render[(1, 1), (8, 8)](grid, np.zeros((3, w, h), np.uint8), cam_o, cam_R, *keep, 0.0, 0.6, 0.3, 1, False)
    assert fake.scenes[6:] == [9.0]                      # sent once, then cached
    cam2 = cuda.to_device(np.ones(3))
    render[(1, 1), (8, 8)](grid, np.zeros((3, w, h), np.uint8), cam2, cam_R, *keep, 0.0, 0.6, 0.3, 1, False)
    assert fake.cameras == [0.0, 1.0]
    fake.set_scene(np.full((7, 1), 7.0, np.float32), None, None)   # someone else talks to the same context
    render[(1, 1), (8, 8)](grid, np.zeros((3, w, h), np.uint8), cam2, cam_R, *keep, 0.0, 0.6, 0.3, 1, False)
    assert fake.scenes[-1] == 9.0 and len(fake.scenes) == 9          # the facade re-sends its scene
