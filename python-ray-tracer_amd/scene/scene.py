"""Scene description -> the float32 arrays the kernel consumes, reference scene/scene.py:9-115:
spheres (7,S) rows cx,cy,cz,r,R,G,B; lights (3,L); planes (9,P) rows ox,oy,oz,nx,ny,nz,R,G,B with the
normal normalised in float64 before the float32 store (scene.py:50)."""
from dataclasses import dataclass
from typing import ClassVar, List

import numpy as np

from .colors import RED, BLUE, MAGENTA, YELLOW, GREEN, GREY


@dataclass
class Sphere:
    origin: object
    radius: float
    color: object
    data_length: ClassVar[int] = 7

    def to_array(self):
        return np.concatenate([np.asarray(self.origin, dtype=np.float64), [self.radius],
                               np.asarray(self.color, dtype=np.float64)]).astype(np.float32)


@dataclass
class Light:
    origin: object
    data_length: ClassVar[int] = 3

    def to_array(self):
        return np.asarray(self.origin, dtype=np.float64).astype(np.float32)


@dataclass
class Plane:
    origin: object
    normal: object
    color: object
    data_length: ClassVar[int] = 9

    def to_array(self):
        n = np.array(self.normal)
        return np.concatenate([np.asarray(self.origin, dtype=np.float64), n / np.linalg.norm(n),
                               np.asarray(self.color, dtype=np.float64)]).astype(np.float32)


def _columns(items, rows):
    out = np.zeros((rows, len(items)), dtype=np.float32)
    for i, it in enumerate(items):
        out[:, i] = it.to_array()
    return out


class Scene:
    def __init__(self, lights: List[Light], spheres: List[Sphere], planes: List[Plane]):
        self.lights, self.spheres, self.planes = lights, spheres, planes

    def get_spheres(self):
        return _columns(self.spheres, Sphere.data_length)

    def get_planes(self):
        return _columns(self.planes, Plane.data_length)

    def get_lights(self):
        return _columns(self.lights, Light.data_length)

    def generate_scene(self):
        return self.get_spheres(), self.get_lights(), self.get_planes()

    @staticmethod
    def default_scene():
        """The reference's built-in scene (scene.py:99-115): 3 lights, 6 spheres, 1 ground plane."""
        lights = [Light(p) for p in ([2.5, -2.0, 3.0], [2.5, 2.0, 3.0], [5.0, 0.1, 6.0])]
        spheres = [Sphere(o, r, c) for o, r, c in (([2.2, 0.3, 1.0], 1.0, RED), ([0.6, 0.7, 0.4], 0.4, BLUE),
                                                    ([0.6, -0.8, 0.5], 0.5, YELLOW), ([-1.2, 0.2, 0.5], 0.5, MAGENTA),
                                                    ([-1.7, -0.5, 0.3], 0.3, GREEN), ([-2.0, 1.31, 1.3], 1.3, RED))]
        return Scene(lights, spheres, [Plane([5, 0, 0], [0, 0, 1], GREY)])
