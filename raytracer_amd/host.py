"""Host-side mirror of the reference's public API for the render path.

Same names, argument meaning and defaults as the Rust structs, so a scene written against the
reference reads the same here:

    Material   src/material.rs:6-106      Pattern    src/pattern.rs:6-91
    Shape.*    src/sphere.rs, plane.rs, cube.rs, cylinder.rs, cone.rs, triangle.rs,
               smooth_triangle.rs, group.rs:171-200 (constructors and defaults)
    PointLight src/pointlight.rs:7-22     World      src/world.rs:17-34
    Camera     src/camera.rs:11-88        Canvas     src/canvas.rs:8-25, src/ppm.rs:30-59
    RefractiveIndex src/refractive.rs:5-9

`Camera.render(world)` is the drop-in boundary: it flattens the World into rt355_shape_desc records
and calls the HIP backend through the C ABI (include/rt355.h).  There is no CPU path in this module.
"""
import ctypes as C
import math

import numpy as np

from . import _abi
from . import matrix as mx

F = np.float32
INFINITY = float("inf")


class RefractiveIndex:
    VACCUME = 1.0
    AIR = 1.00029
    WATER = 1.333
    GLASS = 1.5
    DIAMOND = 2.417


class Pattern:
    """`enum Pattern`; the transformation argument is accepted and ignored, as the reference stores it
    and never reads it (src/pattern.rs:16,94)."""

    def __init__(self, kind, c1=(0, 0, 0), c2=(0, 0, 0), transformation=None):
        self.kind = kind
        self.c1 = tuple(float(x) for x in c1)
        self.c2 = tuple(float(x) for x in c2)
        self.transformation = mx.identity() if transformation is None else transformation

    @staticmethod
    def Stripe(c1, c2, transformation=None):
        return Pattern(_abi.PATTERN_STRIPE, c1, c2, transformation)

    @staticmethod
    def Gradient(c1, c2, transformation=None):
        return Pattern(_abi.PATTERN_GRADIENT, c1, c2, transformation)

    @staticmethod
    def Ring(c1, c2, transformation=None):
        return Pattern(_abi.PATTERN_RING, c1, c2, transformation)

    @staticmethod
    def Checker(c1, c2, transformation=None):
        return Pattern(_abi.PATTERN_CHECKER, c1, c2, transformation)

    @staticmethod
    def Test():
        return Pattern(_abi.PATTERN_TEST)


class Material:
    def __init__(self, color=None, ambient=None, diffuse=None, specular=None, shininess=None,
                 reflective=None, refractive=None, transparency=None, pattern=None):
        # defaults: src/material.rs:31-72
        self.color = (1.0, 1.0, 1.0) if color is None else tuple(float(x) for x in color)
        self.ambient = 0.1 if ambient is None else ambient
        self.diffuse = 0.9 if diffuse is None else diffuse
        self.specular = 0.9 if specular is None else specular
        self.shininess = 200.0 if shininess is None else shininess
        self.reflective = 0.0 if reflective is None else reflective
        self.refractive = RefractiveIndex.VACCUME if refractive is None else refractive
        self.transparency = 0.0 if transparency is None else transparency
        self.pattern = pattern

    new = classmethod(lambda cls, *a, **k: cls(*a, **k))

    @classmethod
    def new_default(cls):
        return cls()

    @classmethod
    def new_default_world(cls):  # src/material.rs:80-92
        return cls(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2)

    @classmethod
    def new_glass(cls):  # src/material.rs:94-106
        return cls(refractive=1.5, transparency=1.0)

    def copy(self):
        m = Material.__new__(Material)
        m.__dict__.update(self.__dict__)
        return m

    def to_desc(self):
        d = _abi.MaterialDesc()
        d.color[:] = self.color
        d.ambient, d.diffuse, d.specular, d.shininess = self.ambient, self.diffuse, self.specular, self.shininess
        d.reflective, d.refractive, d.transparency = self.reflective, self.refractive, self.transparency
        if self.pattern is None:
            d.pattern_kind = _abi.PATTERN_NONE
        else:
            d.pattern_kind = self.pattern.kind
            d.c1[:] = self.pattern.c1
            d.c2[:] = self.pattern.c2
        return d


class Shape:
    """One entry of `World.shapes`.  Use the constructors below (`Shape.Sphere(...)` etc.)."""

    def __init__(self, kind, transformation=None, material=None, min=-INFINITY, max=INFINITY, closed=False,
                 p1=(0, 0, 0), p2=(0, 0, 0), p3=(0, 0, 0), n1=(0, 0, 0), n2=(0, 0, 0), n3=(0, 0, 0)):
        self.kind = kind
        self.transformation = mx.identity() if transformation is None else np.asarray(transformation, dtype=F)
        self.material = Material() if material is None else material
        self.min, self.max, self.closed = min, max, closed
        self.p1, self.p2, self.p3 = (tuple(float(v) for v in p[:3]) for p in (p1, p2, p3))
        self.n1, self.n2, self.n3 = (tuple(float(v) for v in n[:3]) for n in (n1, n2, n3))

    # constructors -------------------------------------------------------------------------------
    @staticmethod
    def Sphere(transformation=None, material=None):
        return Shape(_abi.SPHERE, transformation, material)

    @staticmethod
    def Plane(transformation=None, material=None):
        return Shape(_abi.PLANE, transformation, material)

    @staticmethod
    def Cube(transformation=None, material=None):
        return Shape(_abi.CUBE, transformation, material)

    @staticmethod
    def Cylinder(transformation=None, material=None, min=None, max=None, closed=None):
        return Shape(_abi.CYLINDER, transformation, material, -INFINITY if min is None else min,
                     INFINITY if max is None else max, bool(closed))

    @staticmethod
    def Cone(transformation=None, material=None, min=None, max=None, closed=None):
        return Shape(_abi.CONE, transformation, material, -INFINITY if min is None else min,
                     INFINITY if max is None else max, bool(closed))

    @staticmethod
    def Group(transformation=None, material=None):
        return Shape(_abi.GROUP, transformation, material)

    @staticmethod
    def Triangle(transformation, material, p1, p2, p3):
        return Shape(_abi.TRIANGLE, transformation, material, p1=p1, p2=p2, p3=p3)

    @staticmethod
    def SmoothTriangle(transformation, material, p1, p2, p3, n1=(0, 0, 0), n2=(0, 0, 0), n3=(0, 0, 0)):
        return Shape(_abi.SMOOTH_TRIANGLE, transformation, material, p1=p1, p2=p2, p3=p3, n1=n1, n2=n2, n3=n3)

    # helpers with the reference's names (src/shape.rs:40-67)
    new_sphere = staticmethod(lambda: Shape.Sphere())
    new_plane = staticmethod(lambda: Shape.Plane())

    @staticmethod
    def new_world_defaults():
        return [Shape.Sphere(None, Material.new_default_world()), Shape.Sphere(mx.scaling(0.5, 0.5, 0.5))]

    def set_material(self, material):
        self.material = material
        return self

    def set_transform(self, transformation):
        self.transformation = np.asarray(transformation, dtype=F)

    def copy(self):
        s = Shape.__new__(Shape)
        s.__dict__.update(self.__dict__)
        s.material = self.material.copy()
        s.transformation = self.transformation.copy()
        return s

    def to_desc(self):
        d = _abi.ShapeDesc()
        self.fill_desc(d)
        return d

    def fill_desc(self, d):
        d.kind = self.kind
        d.m[:] = mx.to_abi(self.transformation).tolist()
        d.mat = self.material.to_desc()
        d.min, d.max, d.closed = self.min, self.max, 1 if self.closed else 0
        d.p1[:], d.p2[:], d.p3[:] = self.p1, self.p2, self.p3
        d.n1[:], d.n2[:], d.n3[:] = self.n1, self.n2, self.n3


class PointLight:
    def __init__(self, position=(-10.0, 10.0, -10.0), intensity=(1.0, 1.0, 1.0)):
        self.position = tuple(float(x) for x in position[:3])
        self.intensity = tuple(float(x) for x in intensity[:3])

    new = classmethod(lambda cls, position, intensity: cls(position, intensity))
    from_tuple = classmethod(lambda cls, position, intensity: cls(position, intensity))

    @classmethod
    def new_default(cls):  # src/pointlight.rs:20-22
        return cls()

    def to_desc(self):
        d = _abi.LightDesc()
        d.position[:] = self.position
        d.intensity[:] = self.intensity
        return d


def shapes_to_descs(shapes):
    """Pack a list of Shape (or a prebuilt ctypes array) into a contiguous rt355_shape_desc array."""
    if isinstance(shapes, C.Array):
        return shapes
    arr = (_abi.ShapeDesc * max(len(shapes), 1))()
    for i, s in enumerate(shapes):
        s.fill_desc(arr[i])
    return arr


class World:
    """`struct World { pointlight, shapes }` (src/world.rs:17-34)."""

    def __init__(self, pointlight=None, shapes=None):
        self.pointlight = PointLight.new_default() if pointlight is None else pointlight
        self.shapes = [] if shapes is None else shapes

    new = classmethod(lambda cls, pointlight, shapes: cls(pointlight, shapes))

    @classmethod
    def new_default(cls):
        return cls(None, Shape.new_world_defaults())

    def n_shapes(self):
        return len(self.shapes)

    def descs(self):
        return shapes_to_descs(self.shapes)


class Color:
    """to_u8 of src/color.rs:97-111 for host-side checks (round half away from zero, clamp, NaN -> 0)."""

    @staticmethod
    def to_u8(rgb):
        a = np.asarray(rgb, dtype=np.float32) * np.float32(255.0)
        r = np.where(a >= 0, np.floor(a + np.float32(0.5)), np.ceil(a - np.float32(0.5)))
        r = np.where(np.isnan(r), 0, r)
        return np.clip(r, 0, 255).astype(np.uint8)


class Canvas:
    """`struct Canvas { width, height, pixels }` (src/canvas.rs:8-12).  `pixels` is a float32 array of
    shape (height, width, 3): row-major from the top-left, unclamped linear RGB."""

    def __init__(self, width, height, pixels):
        self.width, self.height = int(width), int(height)
        self.pixels = np.asarray(pixels, dtype=np.float32).reshape(self.height, self.width, 3)

    def pixel_at(self, x, y):
        return self.pixels[y, x]

    def to_u8(self):
        return Color.to_u8(self.pixels)

    def to_ppm(self):
        """P3 text exactly as `Ppm::new` lays it out (src/ppm.rs:30-59): 4 pixels per line, every
        number followed by a space, len/4 + 1 lines."""
        u8 = self.to_u8().reshape(-1, 3)
        n = u8.shape[0]
        lines = []
        for k in range(n // 4 + 1):
            chunk = u8[k * 4:(k + 1) * 4]
            lines.append("".join(f"{p[0]} {p[1]} {p[2]} " for p in chunk))
        return f"P3\n{self.width} {self.height}\n255\n" + "\n".join(lines) + "\n"

    def to_ppm_p6(self):
        return f"P6\n{self.width} {self.height}\n255\n".encode() + self.to_u8().tobytes()


class Device:
    """One rt355 context (one GPU, one stream).  Contexts are cached per (device, real_bits)."""

    _cache = {}

    def __init__(self, device_id=0, real_bits=64, flags=0):
        self.lib = _abi.lib()
        self.handle = C.c_void_p()
        _abi.check(self.lib.rt355_ctx_create_ex(C.byref(self.handle), device_id, real_bits, flags))
        self.device_id, self.real_bits = device_id, real_bits

    @classmethod
    def get(cls, device_id=0, real_bits=64):
        key = (device_id, real_bits)
        if key not in cls._cache:
            cls._cache[key] = cls(device_id, real_bits)
        return cls._cache[key]

    def set_pool(self, rays_per_pixel):
        _abi.check(self.lib.rt355_ctx_set_pool(self.handle, float(rays_per_pixel)))

    def set_exact_cones(self, mode):
        _abi.check(self.lib.rt355_ctx_set_exact_cones(self.handle, int(mode)))

    def set_counting(self, mode):
        """0 / False: off; 1 / True: the frame's kernels instrumented; 2: per-ray count (one ray per lane throughout)."""
        _abi.check(self.lib.rt355_ctx_set_counting(self.handle, int(mode)))

    def set_timing(self, on):
        _abi.check(self.lib.rt355_ctx_set_timing(self.handle, int(bool(on))))

    def upload(self, world):
        return Scene(self, world)

    def to_u8(self, rgb):
        a = np.ascontiguousarray(rgb, dtype=np.float32)
        out = np.empty(a.shape, dtype=np.uint8)
        _abi.check(self.lib.rt355_to_u8(self.handle, a.ctypes.data, a.size, out.ctypes.data))
        return out

    def close(self):
        if self.handle:
            self.lib.rt355_ctx_destroy(self.handle)
            self.handle = C.c_void_p()
        Device._cache = {k: v for k, v in Device._cache.items() if v is not self}


class MultiDevice:
    """Several GPUs of one node driven by this one process (rt355_multi_*): one context and one replica of
    the scene per device; `render` is Camera::render with the frame's tiles spread over the devices."""

    def __init__(self, device_ids, real_bits=64):
        self.lib = _abi.lib()
        self.handle = C.c_void_p()
        ids = (C.c_int * len(device_ids))(*device_ids)
        _abi.check(self.lib.rt355_multi_create(C.byref(self.handle), ids, len(device_ids), real_bits))
        self.n_devices, self.real_bits = len(device_ids), real_bits

    def gather_kind(self):
        return self.lib.rt355_multi_gather_kind(self.handle).decode()

    def upload(self, world):
        light = world.pointlight.to_desc()
        _abi.check(self.lib.rt355_multi_upload(self.handle, world.descs(), world.n_shapes(), C.byref(light)))

    def render(self, camera, max_depth=5, stats=None):
        w, h = int(camera.horizontal_size), int(camera.vertical_size)
        out = np.empty((h, w, 3), dtype=np.float32)
        cd = camera.to_desc()
        _abi.check(self.lib.rt355_multi_render(self.handle, C.byref(cd), max_depth, out.ctypes.data,
                                               C.byref(stats) if stats is not None else None))
        return Canvas(w, h, out)

    def render_path(self, cameras, max_depth=5, stats=None):
        """Camera::render for every camera of a path (one image size): each device renders its tiles of up to
        RT355_MAX_BATCH frames in one launch chain."""
        cameras = list(cameras)
        out = []
        for i in range(0, len(cameras), _abi.MAX_BATCH):
            part = cameras[i:i + _abi.MAX_BATCH]
            w, h = int(part[0].horizontal_size), int(part[0].vertical_size)
            buf = np.empty((len(part), h, w, 3), dtype=np.float32)
            cds = (_abi.CameraDesc * len(part))(*[c.to_desc() for c in part])
            _abi.check(self.lib.rt355_multi_render_batch(self.handle, cds, len(part), max_depth, buf.ctypes.data,
                                                         C.byref(stats) if stats is not None else None))
            out += [Canvas(w, h, buf[f]) for f in range(len(part))]
        return out

    def close(self):
        if self.handle:
            self.lib.rt355_multi_destroy(self.handle)
            self.handle = C.c_void_p()


class Scene:
    """Device-resident World: shape tables + BVH, built once by rt355_scene_upload."""

    def __init__(self, device, world):
        self.device = device
        self.lib = device.lib
        self.handle = C.c_void_p()
        descs = world.descs()
        light = world.pointlight.to_desc()
        _abi.check(self.lib.rt355_scene_upload(device.handle, descs, world.n_shapes(), C.byref(light), C.byref(self.handle)))
        self.n_shapes = world.n_shapes()

    def info(self):
        out = (C.c_uint64 * 5)()
        _abi.check(self.lib.rt355_scene_info(self.handle, out))
        return dict(nodes=out[0], bounded=out[1], unbounded=out[2], depth=out[3], device_bytes=out[4])

    def render(self, camera, max_depth=5, stats=None):
        w, h = int(camera.horizontal_size), int(camera.vertical_size)
        out = np.empty((h, w, 3), dtype=np.float32)
        cd = camera.to_desc()
        _abi.check(self.lib.rt355_render(self.device.handle, self.handle, C.byref(cd), max_depth, out.ctypes.data,
                                         C.byref(stats) if stats is not None else None))
        return Canvas(w, h, out)

    def render_path(self, cameras, max_depth=5, stats=None):
        """`Camera::render` for every camera of a path (all of one image size) in launch chains of up to
        RT355_MAX_BATCH frames: the rays of a chain's frames are traced together.  Returns one Canvas per camera."""
        cameras = list(cameras)
        out = []
        for i in range(0, len(cameras), _abi.MAX_BATCH):
            part = cameras[i:i + _abi.MAX_BATCH]
            w, h = int(part[0].horizontal_size), int(part[0].vertical_size)
            buf = np.empty((len(part), h, w, 3), dtype=np.float32)
            cds = (_abi.CameraDesc * len(part))(*[c.to_desc() for c in part])
            _abi.check(self.lib.rt355_render_batch(self.device.handle, self.handle, cds, len(part), max_depth, buf.ctypes.data,
                                                   C.byref(stats) if stats is not None else None))
            out += [Canvas(w, h, buf[f]) for f in range(len(part))]
        return out

    def render_u8(self, camera, max_depth=5, stats=None):
        """Camera::render + Color::to_u8 on the device: (height, width, 3) uint8, the body of a binary PPM."""
        w, h = int(camera.horizontal_size), int(camera.vertical_size)
        out = np.empty((h, w, 3), dtype=np.uint8)
        cd = camera.to_desc()
        _abi.check(self.lib.rt355_render_u8(self.device.handle, self.handle, C.byref(cd), max_depth, out.ctypes.data,
                                            C.byref(stats) if stats is not None else None))
        return out

    def color_at(self, rays, max_depth=5, stats=None):
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        out = np.empty((r.shape[0], 3), dtype=np.float32)
        _abi.check(self.lib.rt355_color_at(self.device.handle, self.handle, r.ctypes.data, r.shape[0], max_depth,
                                           out.ctypes.data, C.byref(stats) if stats is not None else None))
        return out

    def trace(self, rays):
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        t = np.empty(r.shape[0], dtype=np.float64)
        idx = np.empty(r.shape[0], dtype=np.int32)
        _abi.check(self.lib.rt355_trace_sample(self.device.handle, self.handle, r.ctypes.data, r.shape[0], t.ctypes.data, idx.ctypes.data))
        return t, idx

    def is_shadowed(self, points):
        p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        out = np.empty(p.shape[0], dtype=np.uint8)
        _abi.check(self.lib.rt355_is_shadowed(self.device.handle, self.handle, p.ctypes.data, p.shape[0], out.ctypes.data))
        return out.astype(bool)

    def close(self):
        if self.handle:
            self.lib.rt355_scene_destroy(self.handle)
            self.handle = C.c_void_p()


class Camera:
    """`struct Camera` (src/camera.rs:11-51).  The derived fields are computed in f32 as the reference
    does; the device library re-derives them itself from (hsize, vsize, fov, transform)."""

    def __init__(self, horizontal_size, vertical_size, field_of_view, transform=None):
        self.horizontal_size = F(horizontal_size)
        self.vertical_size = F(vertical_size)
        self.field_of_view = F(field_of_view)
        self.transform = mx.identity() if transform is None else np.asarray(transform, dtype=F)
        half_view = F(math.tan(float(self.field_of_view / F(2.0))))
        aspect_ratio = self.horizontal_size / self.vertical_size
        self.half_width = half_view if aspect_ratio >= 1.0 else F(half_view * aspect_ratio)
        self.half_height = F(half_view / aspect_ratio) if aspect_ratio >= 1.0 else half_view
        self.pixel_size = F((self.half_width * F(2.0)) / self.horizontal_size)

    new = classmethod(lambda cls, h, v, fov, transform: cls(h, v, fov, transform))
    new_default = classmethod(lambda cls, h, v, fov: cls(h, v, fov, None))

    def to_desc(self):
        d = _abi.CameraDesc()
        d.hsize, d.vsize, d.fov = float(self.horizontal_size), float(self.vertical_size), float(self.field_of_view)
        d.transform[:] = mx.to_abi(self.transform).tolist()
        return d

    def render(self, world, max_depth=5, device=None, stats=None):
        """`Camera::render(&self, world: &World) -> Canvas` (src/camera.rs:74-88) on the GPU.
        `max_depth` is the literal 5 of src/camera.rs:79."""
        dev = device or Device.get()
        scene = dev.upload(world)
        try:
            return scene.render(self, max_depth, stats)
        finally:
            scene.close()
