"""raytracer_amd -- MI355X (gfx950) render backend for the `Camera::render(&World) -> Canvas` path of
SokichiFujita/raytracer, behind the C ABI of include/rt355.h.

host.py     mirror of the reference's host API (Material, Shape, World, Camera, Canvas ...)
matrix.py   transform builders (src/matrix.rs) and view_transformation (src/tuple.rs:50-60)
wavefront.py / group.py   the steps before the path: OBJ loading, group flattening
scenes.py   frozen benchmark / parity scenes
dist.py     tile split + canvas gather across ranks (torch.distributed)
csrc/       HIP kernels + C ABI implementation (librt355.so)
"""
from . import matrix  # noqa: F401
from .host import (Camera, Canvas, Color, Device, Material, MultiDevice, Pattern, PointLight, RefractiveIndex, Scene,  # noqa: F401
                   Shape, World)

EPSILON = 1.0e-3  # src/epsilon.rs:1
