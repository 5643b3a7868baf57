"""Host helpers on the input side of the path — same public names as the reference's `scene`
package (scene/__init__.py:1-3).  Their output arrays are the hot path's input format and are
pinned byte-for-byte by tests/golden/host_helpers.npz."""
from .scene import Scene, Light, Plane, Sphere  # noqa: F401
from .rotation import euler_rotation  # noqa: F401
from .camera import Camera, PixelGrid  # noqa: F401
