"""Host code under sanitizers (no GPU): `make -C oracle asan` builds the CPU oracle with UndefinedBehaviorSanitizer as a
library the known-answer suite can load, and a driver that runs the oracle and the library's host-side scene builder
(raytracer_amd/csrc/scene_build.cpp) in one AddressSanitizer + UndefinedBehaviorSanitizer process.  A finding aborts
the process; the results must equal the ordinary builds'."""
import ctypes as C
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raytracer_amd import _abi, scenes  # noqa: E402

SAN = os.path.join(ROOT, "oracle", "_san")


@pytest.fixture(scope="module")
def san_build():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "--no-print-directory", "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return SAN


def test_known_answer_suite_under_ubsan(san_build):
    """Every known-answer test of the reference that the oracle is pinned by, with the oracle built with
    -fsanitize=undefined -fno-sanitize-recover=all: signed overflow, bad shifts, out-of-range float -> int casts,
    misaligned or null accesses abort the run."""
    env = dict(os.environ, RT355_ORACLE_LIB=os.path.join(san_build, "liboracle_ubsan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_kat.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


SCENES = [
    ("glass", lambda: scenes.glass(32, 24), (-5, -0.5, -6), (5, 4, 9), 64, 0),
    ("north_star_low", lambda: scenes.north_star(32, 18, mesh="teapot-low.obj"), (-32, -1, -20), (22, 20, 45), 32, 0),
    ("stress_small", lambda: scenes.stress(32, 18, n_top=2, n_mid=4, n_leaf=12, extent=40.0), (-22, 0.2, -22), (22, 7, 22), 64, 1),
    ("teapot_field", lambda: scenes.teapot_field(32, 18, "teapot-low.obj", 2, through_groups=True), (-60, 0.2, -60), (60, 30, 60), 64, 0),
]


@pytest.mark.parametrize("name,build,lo,hi,bits,exact_cones", SCENES, ids=[s[0] for s in SCENES])
def test_oracle_and_scene_builder_under_asan(san_build, oracle, tmp_path, name, build, lo, hi, bits, exact_cones):
    """Nearest hits, colours, shadow flags and a small frame from the sanitizer process equal the ordinary oracle's
    bit for bit, and its scene tables (both tree levels for instanced meshes) pass the structural check with the
    same shape as rt355_scene_check reports from the ordinary library."""
    world, cam = build()
    rng = np.random.default_rng(7)
    n_rays, n_pts, depth = 400, 300, 5
    o = rng.uniform(lo, hi, size=(n_rays, 3))
    d = rng.uniform(lo, hi, size=(n_rays, 3)) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    pts = rng.uniform(lo, hi, size=(n_pts, 3)).astype(np.float32)
    scene_file, out_file = tmp_path / "scene.bin", tmp_path / "out.bin"
    light, cd = world.pointlight.to_desc(), cam.to_desc()
    with open(scene_file, "wb") as f:
        f.write(struct.pack("6i", world.n_shapes(), n_rays, n_pts, depth, bits, exact_cones))
        f.write(bytes(world.descs())[: world.n_shapes() * C.sizeof(_abi.ShapeDesc)])
        f.write(bytes(light))
        f.write(bytes(cd))
        f.write(rays.tobytes())
        f.write(pts.tobytes())
    r = subprocess.run([os.path.join(san_build, "asan_driver"), str(scene_file), str(out_file)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
    w, h = int(cam.horizontal_size), int(cam.vertical_size)
    buf = open(out_file, "rb").read()
    off = 0

    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(buf, dtype=dtype, count=count, offset=off)
        off += a.nbytes
        return a

    t, idx, rgb = take(np.float64, n_rays), take(np.int32, n_rays), take(np.float64, n_rays * 3).reshape(-1, 3)
    shadowed, canvas, info = take(np.uint8, n_pts), take(np.float32, w * h * 3).reshape(h, w, 3), take(np.uint64, 5)
    assert off == len(buf)
    t0, idx0, _ = oracle.hit(world, rays, bits)
    assert np.array_equal(idx, idx0) and np.array_equal(t, t0, equal_nan=True) and (idx >= 0).sum() > 50
    assert np.array_equal(rgb, oracle.color_at(world, rays, depth, bits), equal_nan=True)
    assert np.array_equal(shadowed.astype(bool), oracle.is_shadowed(world, pts, bits))
    assert np.array_equal(canvas, oracle.render(world, cam, depth, bits)[0], equal_nan=True)
    ref = (C.c_uint64 * 5)()
    assert _abi.lib().rt355_scene_check(world.descs(), world.n_shapes(), bits, exact_cones, ref) == 0
    assert list(info) == list(ref)
