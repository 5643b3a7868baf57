"""Camera and the pixel grid, reference scene/camera.py:7-26.

The reference materialises pixel_loc as a float64 (3,w,h) array with np.mgrid (24 B/pixel that its
kernel then reads back, kernels.py:19).  np.mgrid with a complex step evaluates `index*step + start`
with step = (stop-start)/float(n-1): a closed form.  Camera.generate_pixel_locations() returns that
same array (bit-identical, pinned by tests/golden/host_helpers.npz) as a PixelGrid that also
carries the closed form, so the render facade can hand the five scalars to the kernel and skip the
array entirely."""
import numpy as np

from .rotation import euler_rotation


class PixelGrid(np.ndarray):
    """float64 (3,w,h) pixel-location array + `.raygen = (px, y0, dy, z0, dz)` with
    grid[:, x, y] == (px, x*dy + y0, y*dz + z0)."""
    raygen = None

    def __array_finalize__(self, obj):
        # slices / arithmetic results no longer match the closed form
        self.raygen = None


class Camera:
    def __init__(self, resolution, position, euler, fov=45.0, true_aspect=False):
        self.resolution = resolution
        self._position = position
        self.rotation = euler_rotation(euler[0], euler[1], euler[2])
        self.field_of_view = fov
        self.true_aspect = true_aspect      # False: the reference's AR = int(w/h); True: AR = w/h

    @property
    def position(self):
        return np.array(self._position)

    def raygen(self):
        """(px, y0, dy, z0, dz).  AR = int(width/height) truncates as the reference does
        (camera.py:22): 1 for 16:9, 0 for portrait frames.  true_aspect=True uses w/h instead
        (an undistorted 16:9 image; not what the reference renders)."""
        width, height = self.resolution
        ar = width / height if self.true_aspect else int(width / height)
        px = float(1 / np.tan(np.radians(self.field_of_view) / 2))
        dy = (-ar - ar) / float(width - 1) if width > 1 else 1.0
        dz = (-1 - 1) / float(height - 1) if height > 1 else 1.0
        return px, float(ar), dy, 1.0, dz

    def generate_pixel_locations(self):
        width, height = self.resolution
        px, y0, dy, z0, dz = self.raygen()
        grid = np.empty((3, width, height), dtype=np.float64)
        grid[0] = px
        grid[1] = (np.arange(width, dtype=np.float64) * dy + y0)[:, None]
        grid[2] = (np.arange(height, dtype=np.float64) * dz + z0)[None, :]
        out = grid.view(PixelGrid)
        out.raygen = (px, y0, dy, z0, dz)
        return out
