"""Camera rotation matrix, reference scene/rotation.py:8-43: R = Rz(yaw) @ Ry(pitch) @ Rx(roll),
angles in degrees unless is_radians.  Note Ry's sign convention [[c,0,-s],[0,1,0],[s,0,c]]
(rotation.py:18-20).  The three factors are multiplied in the reference's association so the
float64 result is bit-identical (pinned by tests/golden/host_helpers.npz)."""
import numpy as np


def _axis_rotation(axis, angle):
    c, s = np.cos(angle), np.sin(angle)
    if axis == "x":
        rows = [[1, 0, 0], [0, c, -s], [0, s, c]]
    elif axis == "y":
        rows = [[c, 0, -s], [0, 1, 0], [s, 0, c]]
    else:
        rows = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    return np.array(rows)


def euler_rotation(roll, pitch, yaw, is_radians=False):
    if not is_radians:
        roll, pitch, yaw = np.deg2rad(roll), np.deg2rad(pitch), np.deg2rad(yaw)
    return _axis_rotation("z", yaw) @ _axis_rotation("y", pitch) @ _axis_rotation("x", roll)
