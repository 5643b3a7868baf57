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
