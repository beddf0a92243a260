"""ctypes wrapper of the CPU parity oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, by __graft_entry__.smoke() and by the cpu_baseline leg of bench.py -- never by
the product package.  Build with `make -C oracle` (or __graft_entry__.build()).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import _abi
from raytracer_amd.host import shapes_to_descs

HERE = os.path.dirname(os.path.abspath(__file__))
# RT355_ORACLE_LIB: another build of the same oracle (the sanitizer build of `make -C oracle asan`)
LIB_PATH = os.environ.get("RT355_ORACLE_LIB") or os.path.join(HERE, "liboracle.so")
STRICT_UV = 1


def build():
    subprocess.check_call(["make", "-C", HERE, "--no-print-directory"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        for name in ("rto_render", "rto_color_at", "rto_hit", "rto_world_intersect", "rto_shape_intersect",
                     "rto_normal_at", "rto_comps", "rto_lighting", "rto_is_shadowed", "rto_camera_rays"):
            getattr(_lib, name).restype = C.c_int
    return _lib


def _f32(a, cols):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1, cols))


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _world_args(world):
    descs = world.descs() if hasattr(world, "descs") else shapes_to_descs(world)
    n = world.n_shapes() if hasattr(world, "n_shapes") else len(world)
    light = world.pointlight.to_desc() if hasattr(world, "pointlight") else None
    return descs, C.c_size_t(n), light


def render(world, camera, depth=5, real_bits=32, flags=0, window=None, threads=1):
    """Camera::render on the CPU.  Returns (pixels[h,w,3] float32, counters dict, seconds)."""
    descs, n, light = _world_args(world)
    cd = camera.to_desc()
    w, h = int(camera.horizontal_size), int(camera.vertical_size)
    x0, x1, y0, y1 = window if window else (0, w, 0, h)
    out = np.zeros((h, w, 3), dtype=np.float32)
    cnt = np.zeros(2 * _abi.RT355_MAX_GEN + 1, dtype=np.uint64)
    sec = C.c_double(0)
    rc = lib().rto_render(descs, n, C.byref(light), C.byref(cd), depth, real_bits, flags, x0, x1, y0, y1, threads,
                          _p(out), _p(cnt), C.byref(sec))
    if rc != 0:
        raise RuntimeError("oracle: singular transformation")
    G = _abi.RT355_MAX_GEN
    counters = dict(radiance=cnt[:G].copy(), hits=cnt[G:2 * G].copy(), panics=int(cnt[2 * G]))
    return out, counters, sec.value


def color_at(world, rays, depth=5, real_bits=32, flags=0, threads=1):
    descs, n, light = _world_args(world)
    r = _f32(rays, 6)
    out = np.zeros((r.shape[0], 3), dtype=np.float64)
    rc = lib().rto_color_at(descs, n, C.byref(light), _p(r), C.c_size_t(r.shape[0]), depth, real_bits, flags, threads, _p(out))
    assert rc == 0
    return out


def hit(world, rays, real_bits=32, with_beyond=False, threads=1):
    """Nearest hit per ray: (t float64 [NaN = miss], shape index int32 [-1 = miss], gap to the nearest
    intersection of another shape[, flag: truncated-cone hit outside its extent])."""
    descs, n, _ = _world_args(world)
    r = _f32(rays, 6)
    t = np.zeros(r.shape[0], dtype=np.float64)
    idx = np.zeros(r.shape[0], dtype=np.int32)
    gap = np.zeros(r.shape[0], dtype=np.float64)
    beyond = np.zeros(r.shape[0], dtype=np.uint8)
    rc = lib().rto_hit(descs, n, _p(r), C.c_size_t(r.shape[0]), real_bits, threads, _p(t), _p(idx), _p(gap), _p(beyond))
    assert rc == 0
    if with_beyond:
        return t, idx, gap, beyond.astype(bool)
    return t, idx, gap


def world_intersect(world, ray, real_bits=32, cap=256):
    descs, n, _ = _world_args(world)
    r = _f32(ray, 6)
    t = np.zeros(cap, dtype=np.float64)
    idx = np.zeros(cap, dtype=np.int32)
    k = lib().rto_world_intersect(descs, n, _p(r), real_bits, _p(t), _p(idx), cap)
    assert k >= 0
    return t[:min(k, cap)], idx[:min(k, cap)]


def shape_intersect(shape, origin, direction, real_bits=32, cap=8):
    """Shape::intersect for one shape, emission order: (t[], u[], v[])."""
    d = shape.to_desc()
    r = _f32(list(origin[:3]) + list(direction[:3]), 6)
    t = np.zeros(cap)
    u = np.zeros(cap)
    v = np.zeros(cap)
    k = lib().rto_shape_intersect(C.byref(d), _p(r), real_bits, _p(t), _p(u), _p(v), cap)
    assert k >= 0
    return t[:k], u[:k], v[:k]


def normal_at(shape, point, uv=None, real_bits=32):
    d = shape.to_desc()
    p = _f32(point[:3], 3)
    out = np.zeros(3)
    rc = lib().rto_normal_at(C.byref(d), _p(p), 1 if uv else 0, C.c_double(uv[0] if uv else 0.0),
                             C.c_double(uv[1] if uv else 0.0), real_bits, _p(out))
    assert rc == 0
    return out


def comps(world, ray, xs, which, use_list=True, remaining=5, real_bits=32, flags=0):
    """prepare_computation + schlick + shade_hit + reflected/refracted_color for an explicit
    intersection list xs = [(t, shape_index[, u, v]), ...], as the reference's unit tests build them."""
    descs, n, light = _world_args(world)
    r = _f32(ray, 6)
    t = np.array([x[0] for x in xs], dtype=np.float64)
    idx = np.array([x[1] for x in xs], dtype=np.int32)
    u = np.array([x[2] if len(x) > 2 else np.nan for x in xs], dtype=np.float64)
    v = np.array([x[3] if len(x) > 3 else np.nan for x in xs], dtype=np.float64)
    out = np.zeros(32)
    rc = lib().rto_comps(descs, n, C.byref(light), _p(r), _p(t), _p(idx), _p(u), _p(v), len(xs), which,
                         1 if use_list else 0, remaining, real_bits, flags, _p(out))
    assert rc == 0
    return dict(point=out[0:3], eyev=out[3:6], normalv=out[6:9], over_point=out[9:12], under_point=out[12:15],
                reflectv=out[15:18], inside=bool(out[18]), n1=out[19], n2=out[20], schlick=out[21],
                shade_hit=out[22:25], reflected_color=out[25:28], refracted_color=out[28:31],
                would_panic=bool(out[31]))


def lighting(material, light, point, eyev, normalv, shadowed, real_bits=32):
    m = material.to_desc()
    l = light.to_desc()
    out = np.zeros(3)
    pa, ea, na = _f32(point[:3], 3), _f32(eyev[:3], 3), _f32(normalv[:3], 3)  # keep alive across the call
    rc = lib().rto_lighting(C.byref(m), C.byref(l), _p(pa), _p(ea), _p(na), 1 if shadowed else 0, real_bits, _p(out))
    assert rc == 0
    return out


def is_shadowed(world, points, real_bits=32, threads=1):
    descs, n, light = _world_args(world)
    p = _f32(points, 3)
    out = np.zeros(p.shape[0], dtype=np.uint8)
    rc = lib().rto_is_shadowed(descs, n, C.byref(light), _p(p), C.c_size_t(p.shape[0]), real_bits, threads, _p(out))
    assert rc == 0
    return out.astype(bool)


def camera_rays(camera, pxy, real_bits=32):
    cd = camera.to_desc()
    p = _f32(pxy, 2)
    out = np.zeros((p.shape[0], 6))
    ps = C.c_double(0)
    rc = lib().rto_camera_rays(C.byref(cd), _p(p), C.c_size_t(p.shape[0]), real_bits, _p(out), C.byref(ps))
    assert rc == 0
    return out, ps.value


def to_u8(rgb):
    a = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.zeros(a.shape, dtype=np.uint8)
    lib().rto_to_u8(_p(a), C.c_size_t(a.size), _p(out))
    return out
