"""CPU replays of the exact-arithmetic shortcuts the HIP kernel uses (rt_device.h), checked against the
straightforward evaluation on tens of millions of inputs.  The shortcuts are algorithms, not hardware
features, so they can be validated here without a GPU; the GPU parity tests then confirm the device code."""
import os
import subprocess

from conftest import REPO

ALGO = os.path.join(REPO, "tests", "algo")


def _build(name, tmp_path):
    exe = str(tmp_path / name)
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ALGO, name + ".c"), "-lm"])
    return exe


def test_renormalize_unit_is_bit_identical_to_sqrt_and_divide(tmp_path):
    out = subprocess.check_output([_build("renorm_check", tmp_path), "3000000"], text=True)
    assert "mismatches=0" in out and "fast_path=3000000" in out, out


def test_shared_reciprocal_normalize_is_bit_identical(tmp_path):
    out = subprocess.check_output([_build("normalize_check", tmp_path), "5000000"], text=True)
    assert "mismatches=0" in out and "sqrt_mismatches=0" in out, out


def test_in_range_division_and_sqrt_are_bit_identical(tmp_path):
    """rt_device.h div_inrange / sqrt_inrange (the scene queries' t = num/den, t = n/a, sqrt(D)): the backend's sequences
    without their range handling, over operand ranges wider than float32 scenes produce."""
    out = subprocess.check_output([_build("divsqrt_check", tmp_path), "20000000"], text=True)
    assert "div_mismatches=0" in out and "sqrt_mismatches=0" in out, out


def test_oracle_under_asan_ubsan(tmp_path):
    """the CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (the GPU side cannot run
    sanitizers on this pool).  Two scenes (a golden's, and an empty one) through every oracle entry point; the
    sanitized build must finish without a report and produce the bytes of the regular build."""
    import struct
    import numpy as np
    from conftest import load_frame, raygen_closed_form
    src = os.path.join(ALGO, "oracle_sanitize.c")
    plain, san = str(tmp_path / "plain"), str(tmp_path / "san")
    base = ["gcc", "-O1", "-g", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fexcess-precision=standard", "-fopenmp"]
    subprocess.check_call(base + ["-o", plain, src, "-lm"])
    subprocess.check_call(base + ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-o", san, src, "-lm"])
    g = load_frame("tilted_planes_48")
    w, h = 31, 22                                                     # not multiples of anything
    scenes = [(g["spheres"], g["lights"], g["planes"]),
              (np.zeros((7, 0), np.float32), np.zeros((3, 0), np.float32), np.zeros((9, 0), np.float32))]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    for i, (sp, li, pl) in enumerate(scenes):
        dump = str(tmp_path / f"scene{i}.bin")
        depth = 3
        with open(dump, "wb") as f:
            f.write(struct.pack("8i", w, h, sp.shape[1], li.shape[1], pl.shape[1], depth, 3, 7))
            f.write(np.asarray(g["cam_origin"], np.float64).tobytes()); f.write(np.asarray(g["cam_rot"], np.float64).tobytes())
            f.write(np.asarray(raygen_closed_form(w, h, 45.0), np.float64).tobytes())
            f.write(np.asarray([0.1, 0.5, 0.45] + [0.45 ** (k + 1) for k in range(depth)], np.float64).tobytes())
            for a in (sp, li, pl):
                f.write(np.ascontiguousarray(a, np.float32).tobytes())
        outs = []
        for exe in (plain, san):
            out = str(tmp_path / (os.path.basename(exe) + f"{i}.out"))
            res = subprocess.run([exe, dump, out], env=env, capture_output=True, text=True)
            assert res.returncode == 0 and "ok" in res.stdout, res.stdout + res.stderr
            assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr, res.stderr
            outs.append(open(out, "rb").read())
        assert outs[0] == outs[1] and len(outs[0]) > 3 * w * h * 5
        if i == 0:
            assert any(outs[0][:3 * w * h])                           # the golden's scene renders something
