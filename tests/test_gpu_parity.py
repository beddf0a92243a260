"""GPU parity tests: the HIP backend, called through the C ABI (include/rt355.h), against the CPU oracle.

Bars (from the reference's own tolerances, src/epsilon.rs:1 and Color::to_u8 src/color.rs:97-111):
  * nearest hit: |t_dev - t_oracle| <= EPSILON = 1e-3 and the same shape index, on sampled rays; rays whose
    runner-up intersection lies within 1e-6 of the hit are ties, excluded and counted;
  * images: per-channel |u8_dev - u8_oracle| <= 1; the fraction of pixels outside that is reported and
    bounded (silhouette and shadow-terminator pixels flip with one-ulp differences);
  * ray counts (radiance and shadow rays per generation) equal the oracle's counters;
  * a frame rendered as tiles over 1, 2, 3 ranks is bit-identical to the single-call frame.
"""
import math

import numpy as np
import pytest

from raytracer_amd import Camera, Device, Material, Pattern, PointLight, Shape, World, _abi, scenes
from raytracer_amd import matrix as mx
from raytracer_amd._abi import Rt355Error, Stats

pytestmark = pytest.mark.gpu
EPS = 1e-3
S2 = math.sqrt(2.0)


@pytest.fixture(scope="module", params=[64, 32])
def dev(request):
    return Device.get(0, request.param)


def mismatch(a, b):
    d = np.abs(a.astype(int) - b.astype(int)).max(axis=2)
    return float((d > 1).mean())


def rand_rays(n, seed, center=(0, 0, 0), radius=6.0, spread=1.5):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3))
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * radius + np.asarray(center)
    target = np.asarray(center) + rng.uniform(-spread, spread, size=(n, 3))
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def check_trace(oracle, dev, world, rays, min_hits=10, exact_cones=True):
    scene = dev.upload(world)
    try:
        t, idx = scene.trace(rays)
    finally:
        scene.close()
    to, io, gap, beyond = oracle.hit(world, rays, dev.real_bits, with_beyond=True)
    # a runner-up intersection this close to the hit is a tie: excluded and counted
    tie = gap < (1e-6 if dev.real_bits == 64 else EPS)
    if not exact_cones:
        # cones in the BVH: hits of the parallel-ray branch outside the cone's extent are a declared
        # deviation (include/rt355.h, rt355_ctx_set_exact_cones); excluded and counted
        print(f"truncated-cone hits beyond the extent: {int(beyond.sum())} of {len(beyond)} rays")
        assert beyond.mean() < 0.002
        tie = tie | beyond
    hit_o, hit_d = io >= 0, idx >= 0
    assert hit_o.sum() >= min_hits
    ok = ~tie
    # In double every non-tie ray must agree.  In float the device (FMA-contracted) and the oracle round
    # differently; thresholds of the reference flip for a few rays (|d.y| < EPSILON, the cone's
    # discriminant rounded to 2 decimals at src/shape.rs:300 moves t by up to ~0.05), so the float bar is
    # relative to t and allows 0.5 % outliers, which are counted.
    allowed = 0.0 if dev.real_bits == 64 else 0.005
    same = (hit_o == hit_d) & (idx == io)
    both = ok & hit_o & hit_d & (idx == io)
    err = np.abs(t[both] - to[both])
    bound = EPS if dev.real_bits == 64 else EPS * np.maximum(1.0, np.abs(to[both]))
    n_bad = int((ok & ~same).sum()) + int((err > bound).sum())
    print(f"rays outside the bar: {n_bad} of {int(ok.sum())}")
    assert n_bad <= allowed * ok.sum(), (n_bad, err.max(initial=0.0))
    print(f"ties excluded: {int(tie.sum())} of {len(tie)}; max |dt| = {err.max(initial=0.0):.3e}")
    return int(tie.sum())


# ---------------------------------------------------------------- reference KATs through the C ABI
def test_default_world_hits(dev):  # world.rs:146-156 + intersection.rs:149-154
    scene = dev.upload(World.new_default())
    t, idx = scene.trace([[0, 0, -5, 0, 0, 1], [0, 0, 0, 0, 0, 1], [0, 0, 0.75, 0, 0, 1], [0, 0, 5, 0, 0, 1], [0, 5, -5, 0, 0, 1]])
    scene.close()
    assert list(t[:3]) == [4.0, 0.5, 0.25] and list(idx[:3]) == [0, 1, 0]
    assert np.isnan(t[3]) and idx[3] == -1 and idx[4] == -1


def test_is_shadowed(dev):  # material.rs:268-298
    scene = dev.upload(World.new_default())
    got = scene.is_shadowed([(0, 10, 0), (10, -10, 10), (-20, 20, -20), (-2, 2, -2)])
    scene.close()
    assert list(got) == [False, True, False, False]


def test_rendering_world_with_camera(dev):  # camera.rs:155-169
    cam = Camera(11, 11, math.pi / 2, mx.view_transformation(mx.point(0, 0, -5), mx.point(0, 0, 0), mx.vector(0, 1, 0)))
    canvas = cam.render(World.new_default(), 5, dev)
    assert np.allclose(canvas.pixel_at(5, 5), [0.38066, 0.47583, 0.2855], atol=EPS)


def test_color_at_kats(dev):
    # world.rs:159-167 shade_hit outside; :170-179 inside with moved light
    scene = dev.upload(World.new_default())
    c = scene.color_at([[0, 0, -5, 0, 0, 1]])
    scene.close()
    assert np.allclose(c[0], [0.38066, 0.47583, 0.2855], atol=EPS)
    scene = dev.upload(World(PointLight((0, 0.25, 0), (1, 1, 1)), Shape.new_world_defaults()))
    c = scene.color_at([[0, 0, 0, 0, 0, 1]])
    scene.close()
    assert np.allclose(c[0], [0.90498] * 3, atol=EPS)
    # world.rs:290-367: transparent floor over a red ball
    ray = [0, 0, -3, 0, -S2 / 2, S2 / 2]
    ball = Shape.Sphere(mx.translation(0, -3.5, -0.5), Material(color=(1, 0, 0), ambient=0.5))
    floor = Shape.Plane(mx.translation(0, -1, 0), Material(refractive=1.5, transparency=0.5))
    scene = dev.upload(World(None, Shape.new_world_defaults() + [floor, ball]))
    c = scene.color_at([ray])
    scene.close()
    assert np.allclose(c[0], [0.93642, 0.68642, 0.68642], atol=EPS)
    # world.rs:370-416: reflective + transparent floor (Schlick mix)
    floor = Shape.Plane(mx.translation(0, -1, 0), Material(reflective=0.5, refractive=1.5, transparency=0.5))
    scene = dev.upload(World(None, [floor, ball]))
    c = scene.color_at([ray])
    scene.close()
    assert np.allclose(c[0], [0.93391, 0.69643, 0.69243], atol=100 * EPS)
    # intersection.rs:309-339: reflected colour off a plane, depth 1
    plane = Shape.Plane(mx.translation(0, -1, 0), Material(reflective=0.5))
    scene = dev.upload(World(None, Shape.new_world_defaults() + [plane]))
    full = scene.color_at([ray], max_depth=1)
    direct = scene.color_at([ray], max_depth=0)
    scene.close()
    assert np.allclose(full[0] - direct[0], [0.19032, 0.2379, 0.14274], atol=EPS)


def test_to_u8(dev, oracle):  # color.rs:97-111
    vals = np.array([1.0, 0.6, 0.4, 0.1, 0.2, 0.8, -0.5, 1.5, 0.5 / 255, 1.5 / 255, 2.5 / 255, np.nan], dtype=np.float32)
    assert list(dev.to_u8(vals)) == [255, 153, 102, 26, 51, 204, 0, 255, 1, 2, 3, 0]
    rng = np.random.default_rng(1)
    big = rng.uniform(-0.2, 1.2, size=100000).astype(np.float32)
    assert np.array_equal(dev.to_u8(big), oracle.to_u8(big))


# ---------------------------------------------------------------- nearest hit, per primitive kind
PRIMS = {
    "sphere": lambda: Shape.Sphere(mx.translation(0.3, -0.2, 0.1) @ mx.rotation_z(0.4) @ mx.scaling(1.5, 0.7, 1.1)),
    "cube": lambda: Shape.Cube(mx.rotation_y(0.6) @ mx.rotation_x(0.3) @ mx.scaling(1.2, 0.8, 1.0)),
    "cylinder_closed": lambda: Shape.Cylinder(mx.rotation_x(0.5) @ mx.scaling(0.9, 1.3, 0.9), None, -1.0, 1.0, True),
    "cylinder_open": lambda: Shape.Cylinder(mx.rotation_z(0.2), None, -0.5, 1.5, False),
    "cylinder_infinite": lambda: Shape.Cylinder(mx.rotation_z(1.0) @ mx.scaling(0.5, 1, 0.5)),
    "cone_closed": lambda: Shape.Cone(mx.translation(0, 0.5, 0) @ mx.rotation_x(-0.4), None, -1.0, 0.0, True),
    "cone_double": lambda: Shape.Cone(mx.scaling(1.0, 1.0, 1.0), None, -1.0, 1.0, True),
    "cone_infinite": lambda: Shape.Cone(mx.rotation_x(0.7)),
    "plane": lambda: Shape.Plane(mx.translation(0, -0.5, 0) @ mx.rotation_z(0.3)),
    "triangle": lambda: Shape.Triangle(None, None, (0, 1.5, 0), (-1.5, -1, 0.3), (1.5, -1, -0.3)),
    "triangle_xf": lambda: Shape.Triangle(mx.rotation_y(0.5) @ mx.scaling(2, 1, 1), None, (0, 1, 0), (-1, -1, 0), (1, -1, 0)),
    "smooth_triangle": lambda: Shape.SmoothTriangle(None, None, (0, 1.5, 0), (-1.5, -1, 0), (1.5, -1, 0), (0, 1, -1), (-1, 0, -1), (1, 0, -1)),
}


@pytest.mark.parametrize("name", sorted(PRIMS))
def test_trace_single_primitive(oracle, dev, name):
    world = World(None, [Shape.Group(), PRIMS[name](), Shape.Group()])  # groups shift the index, never hit
    check_trace(oracle, dev, world, rand_rays(20000, 7), min_hits=1000)


def test_trace_mixed_scene(oracle, dev):
    world, _ = scenes.stress(64, 36, n_top=2, n_mid=6, n_leaf=25, extent=40.0)
    rays = rand_rays(30000, 11, center=(0, 1.5, 0), radius=45.0, spread=18.0)
    check_trace(oracle, dev, world, rays, min_hits=3000, exact_cones=False)
    dev.set_exact_cones(1)
    try:
        check_trace(oracle, dev, world, rays, min_hits=3000, exact_cones=True)
    finally:
        dev.set_exact_cones(0)


def test_trace_teapot(oracle, dev):
    world, _ = scenes.teapot_low(64, 36)
    rays = rand_rays(20000, 13, center=(0, 0, 7), radius=60.0, spread=12.0)
    check_trace(oracle, dev, world, rays, min_hits=3000)


def test_trace_ties_lowest_index_wins(dev):
    # coincident shapes: equal t -> lowest index in World.shapes (world.rs:42 + intersection.rs:151)
    tri = lambda: Shape.Triangle(None, None, (0, 1, 0), (-1, -1, 0), (1, -1, 0))
    world = World(None, [Shape.Group(), tri(), tri(), Shape.Sphere(mx.translation(0, 0, 1)), Shape.Sphere(mx.translation(0, 0, 1)), Shape.Plane(mx.rotation_x(math.pi / 2)), tri()])
    scene = dev.upload(world)
    t, idx = scene.trace([[0, 0, -5, 0, 0, 1], [0, 0, 5, 0, 0, -1], [5, 5, -5, 0, 0, 1]])
    scene.close()
    assert t[0] == 5.0 and idx[0] == 1  # triangles 1, 2, 6 and plane 5 all at t = 5
    assert t[1] == 3.0 and idx[1] == 3  # twin spheres
    assert abs(t[2] - 5.0) < 1e-5 and idx[2] == 5  # only the plane


# ---------------------------------------------------------------- images
# (name, builder, allowed mismatching-pixel fraction in double, in float).
# ch11 (src/main.rs:389-535) is ill-conditioned by construction: its floor is a sphere squashed to 0.01
# (thinner than 10 * EPSILON) and its walls are rotated planes whose untransformed normal (src/shape.rs:144)
# leaves over_point ON the wall, so the shadow test there is decided by the last bit.  In float the
# reference's own image is rounding noise in those regions: its float bar (None) is the masked comparison of
# test_gpu_configs.py::test_ch11_f32_outside_the_ill_conditioned_pixels, not a whole-image fraction.
IMAGES = [
    ("ch9", lambda: scenes.ch9(200, 150), 0.002, 0.002),
    ("ch11", lambda: scenes.ch11(160, 120), 0.004, None),
    ("glass", lambda: scenes.glass(160, 120), 0.004, 0.01),
    ("teapot_low", lambda: scenes.teapot_low(160, 90), 0.002, 0.004),
    ("north_star_low", lambda: scenes.north_star(128, 72, mesh="teapot-low.obj"), 0.004, 0.01),
    ("stress_small", lambda: scenes.stress(128, 72, n_top=2, n_mid=6, n_leaf=25, extent=40.0), 0.004, 0.01),
]


@pytest.mark.parametrize("name,build,tol64,tol32", IMAGES, ids=[i[0] for i in IMAGES])
def test_image_parity(oracle, dev, name, build, tol64, tol32):
    tol = tol64 if dev.real_bits == 64 else tol32
    if tol is None:
        pytest.skip("compared under a mask of the ill-conditioned pixels in test_gpu_configs.py")
    world, cam = build()
    scene = dev.upload(world)
    st = Stats()
    try:
        got = scene.render(cam, 5, st).pixels
    finally:
        scene.close()
    want, cnt, _ = oracle.render(world, cam, 5, dev.real_bits, threads=8)
    frac = mismatch(oracle.to_u8(got), oracle.to_u8(want))
    print(f"{name} real={dev.real_bits}: mismatching pixels {frac:.5%}, rays {int(st.total_radiance)} + {int(st.total_shadow)} shadow")
    assert frac <= tol
    # ray counts: identical unless a pixel's ray tree changed (a mismatching pixel can move its whole tree
    # of at most 63 rays; a grazing deep ray of small weight can flip without moving the pixel by 1/255)
    tot_o = int(cnt["radiance"].sum())
    # generation 0 of the device includes tile padding pixels
    w, h = int(cam.horizontal_size), int(cam.vertical_size)
    dev_radiance = int(st.total_radiance) - int(st.radiance_rays[0]) + w * h
    n_bad = int(round(frac * w * h))
    slack = 8 + 64 * n_bad + (0.0005 if dev.real_bits == 64 else 0.01) * tot_o
    print(f"{name} real={dev.real_bits}: radiance rays {dev_radiance} vs {tot_o}, shadow rays {int(st.total_shadow)} vs {int(cnt['hits'].sum())}")
    assert abs(dev_radiance - tot_o) <= slack
    assert abs(int(st.total_shadow) - int(cnt["hits"].sum())) <= slack


def test_image_f64_device_vs_f32_oracle_reported(oracle):
    """The reference computes in f32; the graded device configuration is f64.  Report how many pixels
    that changes (branch thresholds such as |d.y| < EPSILON flip for a few rays)."""
    world, cam = scenes.ch9(200, 150)
    got = cam.render(world, 5, Device.get(0, 64)).pixels
    want, _, _ = oracle.render(world, cam, 5, 32, threads=8)
    frac = mismatch(oracle.to_u8(got), oracle.to_u8(want))
    print(f"ch9 200x150: f64 device vs f32 oracle mismatching pixels {frac:.5%}")
    assert frac <= 0.005


@pytest.mark.parametrize("name", ["ch9", "glass", "north_star_low"])
def test_counts_exact_on_small_scenes(oracle, dev, name):
    """Radiance rays and hits (= shadow rays) per generation equal the oracle's counters exactly in f64 on
    the small parity scenes (96x64: whole tiles, no padding); in f32 the two sides round differently
    (contracted FMAs on the device) and a handful of grazing rays may flip."""
    build = {"ch9": scenes.ch9, "glass": scenes.glass, "north_star_low": lambda w, h: scenes.north_star(w, h, mesh="teapot-low.obj")}[name]
    world, cam = build(96, 64)
    scene = dev.upload(world)
    st = Stats()
    scene.render(cam, 5, st)
    scene.close()
    _, cnt, _ = oracle.render(world, cam, 5, dev.real_bits, threads=8)
    got_r, got_h = [int(x) for x in st.radiance_rays], [int(x) for x in st.hits]
    want_r, want_h = [int(x) for x in cnt["radiance"]], [int(x) for x in cnt["hits"]]
    print(f"{name} real={dev.real_bits}: rays {got_r} vs {want_r}; hits {got_h} vs {want_h}")
    assert got_r[0] == 96 * 64 == want_r[0]
    if dev.real_bits == 64:
        assert got_r == want_r and got_h == want_h
    else:
        assert sum(abs(a - b) for a, b in zip(got_r, want_r)) <= 0.003 * sum(want_r)
        assert sum(abs(a - b) for a, b in zip(got_h, want_h)) <= 0.003 * sum(want_h)


# ---------------------------------------------------------------- tiles / multi-rank layout
@pytest.mark.parametrize("size", [(200, 150), (96, 64), (33, 31)])
def test_tiles_bit_identical_to_single_call(dev, size):
    import torch
    from raytracer_amd import dist as rd
    world, cam = scenes.glass(*size)
    scene = dev.upload(world)
    try:
        full = scene.render(cam, 5).pixels
        for n_ranks in (1, 2, 3, 8):
            tpr = rd.tiles_per_rank(size[0], size[1], n_ranks)
            gathered = torch.zeros((n_ranks, tpr * 1024 * 3), dtype=torch.float32, device="cuda:0")
            for r in range(n_ranks):
                rd.render_tiles(scene, cam, 5, r, n_ranks, gathered[r])
            out = torch.empty((size[1], size[0], 3), dtype=torch.float32, device="cuda:0")
            rd.untile(dev, cam, n_ranks, gathered, out)
            assert np.array_equal(out.cpu().numpy(), full), n_ranks
            assert np.array_equal(rd.untile_host(gathered.cpu().numpy(), size[0], size[1], n_ranks), full)
    finally:
        scene.close()


def test_render_is_deterministic(dev):
    world, cam = scenes.glass(160, 120)
    scene = dev.upload(world)
    a = scene.render(cam, 5).pixels.copy()
    b = scene.render(cam, 5).pixels
    scene.close()
    assert np.array_equal(a, b)


@pytest.mark.parametrize("env", [{"RT355_QUAD": "3"}, {"RT355_QUAD": "1"}, {"RT355_PACKET_GENS": "0", "RT355_QUAD": "3"},
                                 {"RT355_PACKET_GENS": "0"}, {"RT355_PACKET_GENS": "6"},
                                 {"RT355_COMBINE_LEVELS": "0"}, {"RT355_COMBINE_LEVELS": "2"}, {"RT355_PACKET_GENS": "1"}, {"RT355_MERGE_SPAWN": "1"},
                                 {"RT355_MERGE_SPAWN": "1", "RT355_PACKET_GENS": "0"}, {"RT355_MERGE_SPAWN": "1", "RT355_PACKET_GENS": "6"}],
                         ids=["quad", "quad-trace", "quad-all", "per-lane-all", "packet-all", "combine-per-generation", "combine-two-levels",
                              "packet-primary-only", "merged-spawn", "merged-spawn-per-lane", "merged-spawn-packets"])
@pytest.mark.parametrize("real_bits", [64, 32])
def test_every_launch_plan_renders_the_same_image(env, real_bits):
    """The frame can be run through other kernels than the default ones (chosen at context creation from the
    environment): the traversal with four lanes to a ray, wave packets or single lanes for every generation,
    one combine launch per generation, the reflected rays written by the shadow launch (the plan of chains that
    overlap) or by k_spawn (the plan of a frame rendered alone).  A ray's colour is a
    function of the ray alone, so every plan must give the default plan's image (bit for bit in double), and the
    same hits and shadow flags in the probes."""
    import os
    base_dev = Device.get(0, real_bits)
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        alt_dev = Device(0, real_bits)  # a context of its own: the knobs are read when it is created
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        for build in (lambda: scenes.glass(200, 150), lambda: scenes.north_star(160, 90, mesh="teapot-low.obj"),
                      lambda: scenes.stress(128, 72, n_top=2, n_mid=6, n_leaf=25, extent=40.0),
                      lambda: scenes.teapot_field(160, 90, "teapot-low.obj", 3)):  # instanced meshes: two-level walks in every plan
            world, cam = build()
            # truncated cones are tested by every ray here: with cones inside the BVH the reference's hits beyond a
            # cone's extent are found or not depending on the order of the traversal, which differs between plans
            base_dev.set_exact_cones(1), alt_dev.set_exact_cones(1)
            try:
                a_scene, b_scene = base_dev.upload(world), alt_dev.upload(world)
            finally:
                base_dev.set_exact_cones(0), alt_dev.set_exact_cones(0)
            sa, sb = Stats(), Stats()
            try:
                a = a_scene.render(cam, 5, sa).pixels
                b = b_scene.render(cam, 5, sb).pixels
                rays = rand_rays(5000, 3, center=(0, 1, 0), radius=8.0, spread=4.0)
                ta, ia = a_scene.trace(rays)
                tb, ib = b_scene.trace(rays)
                pts = np.random.default_rng(4).uniform(-5, 5, (5000, 3)).astype(np.float32)
                fa, fb = a_scene.is_shadowed(pts), b_scene.is_shadowed(pts)
            finally:
                a_scene.close()
                b_scene.close()
            if real_bits == 64:
                assert np.array_equal(a, b, equal_nan=True)
                assert list(sa.radiance_rays) == list(sb.radiance_rays) and list(sa.hits) == list(sb.hits)
                assert np.array_equal(ia, ib) and np.array_equal(ta, tb, equal_nan=True) and np.array_equal(fa, fb)
            else:
                # the same expressions are compiled into different kernels, where the compiler contracts
                # multiply-adds differently: in float that shows in the last bits, and flips a grazing ray now and then
                d = np.abs(np.nan_to_num(a) - np.nan_to_num(b)).max(axis=2)
                assert (d > 1e-4).mean() <= 0.002 and (ia != ib).mean() <= 0.002 and (fa != fb).mean() <= 0.002
                same = ia == ib
                assert np.allclose(ta[same], tb[same], rtol=1e-4, atol=1e-5, equal_nan=True)
    finally:
        alt_dev.close()


# ---------------------------------------------------------------- edge cases and errors
def test_empty_and_group_only_worlds(dev):
    cam = Camera(40, 30, math.pi / 3)
    for shapes in ([], [Shape.Group(), Shape.Group()]):
        canvas = cam.render(World(None, shapes), 5, dev)
        assert canvas.pixels.shape == (30, 40, 3) and not canvas.pixels.any()


def test_one_pixel_and_one_shape(dev, oracle):
    world = World(None, [Shape.Sphere()])
    cam = Camera(1, 1, math.pi / 3, mx.view_transformation(mx.point(0, 0, -5), mx.point(0, 0, 0), mx.vector(0, 1, 0)))
    got = cam.render(world, 5, dev).pixels
    want, _, _ = oracle.render(world, cam, 5, dev.real_bits)
    assert np.allclose(got, want, atol=1e-5)


def test_errors(dev):
    singular = Shape.Sphere(mx.scaling(1, 0, 1))
    with pytest.raises(Rt355Error, match="singular"):
        dev.upload(World(None, [singular]))
    bad = Shape.Sphere()
    bad.kind = 99
    with pytest.raises(Rt355Error):
        dev.upload(World(None, [bad]))
    scene = dev.upload(World.new_default())
    with pytest.raises(Rt355Error, match="max_depth"):
        scene.render(Camera(8, 8, 1.0), 9)
    scene.close()


def test_pool_overflow_is_reported(dev):
    # two facing mirrors + glass: every hit spawns rays; a pool of 1 ray per pixel cannot hold them
    world, cam = scenes.glass(160, 120)
    scene = dev.upload(world)
    dev.set_pool(1.0)
    try:
        with pytest.raises(Rt355Error, match="pool"):
            scene.render(cam, 5)
    finally:
        dev.set_pool(8.0)
    assert scene.render(cam, 5).pixels.any()
    scene.close()


def test_counting_mode_reports_work(dev):
    world, cam = scenes.teapot_low(96, 64)
    scene = dev.upload(world)
    dev.set_counting(True)
    dev.set_timing(True)
    st = Stats()
    try:
        a = scene.render(cam, 5, st).pixels.copy()
    finally:
        dev.set_counting(False)
        dev.set_timing(False)
    b = scene.render(cam, 5).pixels
    scene.close()
    assert np.array_equal(a, b)  # instrumented kernels compute the same image
    kinds = list(st.launch_kind[: st.n_launch])
    assert kinds == [_abi.K_TRACE, _abi.K_SCAN, _abi.K_COMBINE]  # one generation: traced and shaded in one launch
    assert st.launch_box_tests[0] > 0 and st.launch_prim_tests[0] > 0 and st.launch_flops[0] > 0
    assert all(st.launch_ms[i] >= 0 for i in range(st.n_launch))


def test_per_ray_count_does_not_exceed_the_packet_count(dev):
    """Counting mode 2 traces every generation one ray per lane: the image is the same, the rays and hits are the same,
    and a ray is charged only for what its own path through the BVH tests -- never more than what the wave packets
    of the default plan charge their lanes."""
    world, cam = scenes.north_star(160, 90, mesh="teapot-low.obj")
    scene = dev.upload(world)
    a, b = Stats(), Stats()
    try:
        dev.set_counting(1)
        img_a = scene.render(cam, 5, a).pixels.copy()
        dev.set_counting(2)
        img_b = scene.render(cam, 5, b).pixels.copy()
    finally:
        dev.set_counting(0)
    scene.close()
    assert np.array_equal(img_a, img_b)
    assert list(a.radiance_rays) == list(b.radiance_rays) and list(a.hits) == list(b.hits)
    box_a, box_b = sum(a.launch_box_tests[: a.n_launch]), sum(b.launch_box_tests[: b.n_launch])
    prim_a, prim_b = sum(a.launch_prim_tests[: a.n_launch]), sum(b.launch_prim_tests[: b.n_launch])
    assert 0 < box_b <= box_a and 0 < prim_b <= prim_a
    with pytest.raises(Exception):
        dev.set_counting(3)


def test_deepest_recursion_with_instrumentation(dev, oracle):
    """max_depth 7 is the limit of the interface: every generation, with and without per-launch
    instrumentation (which also moves the shadow launches from the side stream to the main one)."""
    world, cam = scenes.glass(64, 48)
    scene = dev.upload(world)
    plain = scene.render(cam, 7).pixels.copy()
    dev.set_counting(True)
    dev.set_timing(True)
    st = Stats()
    try:
        inst = scene.render(cam, 7, st).pixels.copy()
    finally:
        dev.set_counting(False)
        dev.set_timing(False)
    scene.close()
    assert np.array_equal(plain, inst)
    assert st.n_launch <= _abi.RT355_MAX_LAUNCH
    kinds = list(st.launch_kind[: st.n_launch])
    assert kinds.count(_abi.K_TRACE) == 8 and kinds.count(_abi.K_SPAWN) == 7
    # the deepest generation shades in its trace; the first combine launch folds generations 2 to 6 (four levels below it)
    assert kinds.count(_abi.K_SHADE) == 7 and kinds.count(_abi.K_COMBINE) == 3
    want, cnt, _ = oracle.render(world, cam, 7, dev.real_bits, threads=8)
    frac = mismatch(oracle.to_u8(plain), oracle.to_u8(want))
    assert frac <= (0.004 if dev.real_bits == 64 else 0.02)
    tot_o = int(cnt["radiance"].sum())
    dev_radiance = int(st.total_radiance) - int(st.radiance_rays[0]) + 64 * 48  # generation 0 includes tile padding
    assert abs(dev_radiance - tot_o) <= max(8, 0.03 * tot_o)
    assert abs(int(st.total_shadow) - int(cnt["hits"].sum())) <= max(8, 0.03 * tot_o)


def test_render_device_equals_render(dev):
    """rt355_render_device leaves in HBM exactly the canvas rt355_render copies to the host."""
    import torch
    from raytracer_amd import dist as rd
    world, cam = scenes.glass(70, 45)
    scene = dev.upload(world)
    want = scene.render(cam, 5).pixels
    out = torch.empty(45 * 70 * 3, dtype=torch.float32, device="cuda")
    rd.render_canvas(scene, cam, 5, out)
    scene.close()
    assert np.array_equal(out.cpu().numpy().reshape(45, 70, 3), want)


def test_headline_frame_at_full_size(oracle):
    """The bench workload at its full size (teapot-high + floor + glass sphere, 1920x1080, depth 5, f64),
    through properties that do not need a full CPU render: the rank split is bit-identical to the frame
    rendered in one piece, ray counts obey the recursion's bounds, and a sample of pixels spread over the
    frame equals World::color_at of the CPU oracle on the same camera rays."""
    import torch
    from raytracer_amd import dist as rd
    w, h = 1920, 1080
    world, cam = scenes.north_star(w, h)
    dev = Device.get(0, 64)
    scene = dev.upload(world)
    st = Stats()
    try:
        full = torch.empty(h * w * 3, dtype=torch.float32, device="cuda")
        rd.render_canvas(scene, cam, 5, full, st)
        full = full.cpu().numpy().reshape(h, w, 3)
        for n_ranks in (2, 8):
            tpr = rd.tiles_per_rank(w, h, n_ranks)
            gathered = torch.zeros((n_ranks, tpr * 1024 * 3), dtype=torch.float32, device="cuda")
            for r in range(n_ranks):
                rd.render_tiles(scene, cam, 5, r, n_ranks, gathered[r])
            out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
            rd.untile(dev, cam, n_ranks, gathered, out)
            assert np.array_equal(out.cpu().numpy(), full), n_ranks
    finally:
        scene.close()
    rays = [int(x) for x in st.radiance_rays]
    hits = [int(x) for x in st.hits]
    assert rays[0] == rd.tiles_per_rank(w, h, 1) * 1024 and hits[0] <= w * h  # generation 0 includes tile padding
    for g in range(5):
        assert hits[g] <= rays[g] and rays[g + 1] <= 2 * hits[g]
    assert rays[6] == 0 and int(st.total_shadow) == sum(hits)
    # sampled pixels against the oracle (brute force over 6 322 shapes, recursive, f64)
    rng = np.random.default_rng(355)
    pxy = np.stack([rng.integers(0, w, 400), rng.integers(0, h, 400)], axis=1)
    cam_rays, _ = oracle.camera_rays(cam, pxy, 64)
    want = oracle.color_at(world, cam_rays, 5, 64)
    got = full[pxy[:, 1], pxy[:, 0]]
    bad = (np.abs(oracle.to_u8(got).astype(int) - oracle.to_u8(want).astype(int)).max(axis=1) > 1).mean()
    print(f"headline frame: {bad:.3%} of 400 sampled pixels differ by more than 1/255")
    assert bad <= 0.01
    # How much of this frame lies where the reference itself has no answer: a hit on a smooth triangle whose ray has
    # another shape's intersection as xs[0] (the floor at a small negative t behind a reflected ray) makes the
    # reference panic (src/intersection.rs:67-71); device and oracle take the hit's own u, v there.  Counted by the
    # oracle's strict_uv switch over windows spread across the frame.
    import oracle as o
    size, pixels, panics, hits = 8, 0, 0, 0
    for wy in range(60, h - size, 120):       # a regular grid of 8x8 windows: 9 rows x 12 columns
        for wx in range(80, w - size, 160):
            _, cnt, _ = oracle.render(world, cam, 5, 64, o.STRICT_UV, window=(wx, wx + size, wy, wy + size), threads=16)
            pixels += size * size
            panics += cnt["panics"]
            hits += int(cnt["hits"].sum())
    print(f"headline frame: {panics} of {pixels} sampled pixels ({panics / pixels:.1%}, about {panics / pixels * w * h:.0f} of the frame's {w * h}) "
          f"carry a hit where the reference would panic (smooth triangle hit by a ray whose xs[0] is another shape); "
          f"{hits} hits in the sampled pixels' ray trees")
    assert 0 < panics < pixels


@pytest.mark.parametrize("second_index", [1.0, 1.3])
def test_refractive_indices_with_overlapping_shapes(dev, oracle, second_index):
    """n1 / n2 (src/intersection.rs:86-127) where shapes overlap: a glass sphere intersecting a transparent
    sphere, both over a floor, rays starting inside and outside.  With second_index 1.0 every shape but the
    glass one has refractive index 1, which lets the device skip most of the container search; with 1.3 it
    cannot.  Both must give the oracle's colours."""
    glass = Material.new_glass()
    glass.reflective = 0.3
    other = Material(color=(0.2, 0.8, 0.3), transparency=0.6, refractive=second_index, reflective=0.1 if second_index != 1.0 else 0.0)
    floor = Material(color=(0.9, 0.9, 0.9), reflective=0.2)
    floor.pattern = Pattern.Checker((1, 1, 1), (0.1, 0.1, 0.1))
    world = World(PointLight((-4, 6, -5), (1, 1, 1)), [
        Shape.Sphere(mx.translation(0, 0, 0), glass),
        Shape.Sphere(mx.translation(0.9, 0.2, 0.3) @ mx.scaling(1.2, 1.2, 1.2), other),
        Shape.Cube(mx.translation(-1.2, 0, 0.5) @ mx.scaling(0.6, 0.6, 0.6), Material(color=(0.8, 0.3, 0.2))),
        Shape.Plane(mx.translation(0, -1.5, 0), floor),
    ])
    rng = np.random.default_rng(17)
    n = 6000
    org = rng.uniform(-2.5, 2.5, (n, 3))
    org[: n // 2] *= 0.3  # half of the rays start inside one or both spheres
    tgt = rng.uniform(-1.0, 1.0, (n, 3))
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d], axis=1).astype(np.float32)
    scene = dev.upload(world)
    dev.set_pool(64.0)  # every hit may spawn two rays here: 1 + 2 + ... + 32 per primary ray at most
    try:
        got = scene.color_at(rays, 5)
    finally:
        dev.set_pool(8.0)
        scene.close()
    want = oracle.color_at(world, rays, 5, dev.real_bits)
    tol = 1e-4 if dev.real_bits == 64 else 2e-2
    bad = (np.abs(got - want).max(axis=1) > tol).mean()
    print(f"overlapping shapes, index {second_index}, real={dev.real_bits}: {bad:.3%} of {n} rays differ by more than {tol}")
    assert bad <= (0.003 if dev.real_bits == 64 else 0.03)


@pytest.mark.parametrize("case", ["apart", "floor-through-the-glass", "cube-inside-the-glass", "cube-touching-the-box"])
@pytest.mark.parametrize("glass_kind", ["sphere", "cube", "cylinder"])
def test_refractive_indices_of_a_glass_shape_alone_in_its_box(dev, oracle, case, glass_kind):
    """A ray leaving a transparent shape looks for other shapes entered on the stretch inside it
    (src/intersection.rs:86-127).  When no other shape reaches into the glass shape's bounding box the device
    decides at upload that there is nothing to find and skips the search; a floor cutting through the glass, an
    opaque cube inside it or one that reaches into the corner of its box must keep the search.  Rays start inside
    and outside the glass; every case must give the oracle's colours."""
    glass = Material.new_glass()
    glass.reflective = 0.4
    make = {"sphere": lambda t: Shape.Sphere(t, glass), "cube": lambda t: Shape.Cube(t, glass),
            "cylinder": lambda t: Shape.Cylinder(t, glass, -1.0, 1.0, True)}[glass_kind]
    floor = Material(color=(0.9, 0.9, 0.9), reflective=0.2)
    floor.pattern = Pattern.Checker((1, 1, 1), (0.1, 0.1, 0.1))
    red = Material(color=(0.8, 0.3, 0.2))
    shapes = [make(mx.rotation_y(0.4) @ mx.scaling(1.0, 1.2, 1.0)),
              Shape.Plane(mx.translation(0, -0.4 if case == "floor-through-the-glass" else -2.0, 0), floor)]
    if case == "cube-inside-the-glass":
        shapes.append(Shape.Cube(mx.translation(0.1, 0.2, 0.0) @ mx.scaling(0.3, 0.3, 0.3), red))
    elif case == "cube-touching-the-box":  # inside the glass shape's box, outside a sphere or cylinder
        shapes.append(Shape.Cube(mx.translation(1.35, 1.3, 1.35) @ mx.scaling(0.25, 0.25, 0.25), red))
    else:
        shapes.append(Shape.Cube(mx.translation(3.0, 0.0, 1.0) @ mx.scaling(0.5, 0.5, 0.5), red))
    world = World(PointLight((-4, 6, -5), (1, 1, 1)), shapes)
    rng = np.random.default_rng(23)
    n = 4000
    org = rng.uniform(-2.5, 2.5, (n, 3))
    org[: n // 2] *= 0.3
    tgt = rng.uniform(-1.0, 1.0, (n, 3))
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d], axis=1).astype(np.float32)
    scene = dev.upload(world)
    dev.set_pool(64.0)
    dev.set_counting(True)
    st = Stats()
    try:
        got = scene.color_at(rays, 5, st)
    finally:
        dev.set_counting(False)
        dev.set_pool(8.0)
        scene.close()
    want = oracle.color_at(world, rays, 5, dev.real_bits)
    tol = 1e-4 if dev.real_bits == 64 else 2e-2
    bad = (np.abs(got - want).max(axis=1) > tol).mean()
    # the search is the only traversal of the spawn launches: their box tests tell whether it ran
    searched = sum(int(st.launch_box_tests[i]) for i in range(st.n_launch) if st.launch_kind[i] == _abi.K_SPAWN)
    print(f"glass {glass_kind}, {case}, real={dev.real_bits}: {bad:.3%} of {n} rays differ by more than {tol}; box tests of the n1/n2 search: {searched}")
    assert bad <= (0.003 if dev.real_bits == 64 else 0.03)
    assert (searched == 0) == (case == "apart")


def test_reciprocal_square_root_is_good_to_an_ulp():
    """normalize() and the shadow ray multiply by 1 / |v| from the hardware's reciprocal square root and two Newton
    steps (kernels.h, inv_sqrt<double>) where nalgebra divides by a square root: tests/cpp/inv_sqrt_check.hip runs the
    kernels' own function over a million arguments across 160 binades against the long-double value and fails beyond
    1.5 units of the last place (the square root and the division it replaces are good to one between them)."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "inv_sqrt_check")
    assert os.path.exists(exe), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout.strip())
    assert r.returncode == 0, r.stdout + r.stderr
    assert float(r.stdout.split("largest error")[1].split()[0]) <= 1.5


def test_cpp_host_mirror_renders_like_the_python_one(tmp_path):
    """include/rt355.hpp: the reference's camera test (src/camera.rs:155-169) and its default scene
    (src/main.rs:202-273) written against the C++ mirror; the canvas must equal the one the Python mirror
    gets for the same scene (same library underneath; the two hosts build their matrices independently)."""
    import os
    import subprocess
    from raytracer_amd.host import Color
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "cpp", "host_mirror_test")
    assert os.path.exists(exe), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    out = tmp_path / "ch9.f32"
    r = subprocess.run([exe, "--gpu", str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out, dtype=np.float32).reshape(150, 200, 3)
    world, cam = scenes.ch9(200, 150)
    want = cam.render(world, 5, Device.get(0, 64)).pixels
    d = np.abs(Color.to_u8(got).astype(int) - Color.to_u8(want).astype(int))
    assert (d.max(axis=2) > 1).mean() <= 0.001
    assert np.abs(got - want).max() < 0.05


def test_canvas_to_u8_on_the_device(dev):
    """rt355_to_u8_device on a canvas left in HBM equals Color::to_u8 (src/color.rs:97-111) on the host,
    and gives the body of Canvas.to_ppm_p6."""
    import torch
    from raytracer_amd import dist as rd
    from raytracer_amd.host import Canvas, Color
    world, cam = scenes.ch9(70, 45)
    scene = dev.upload(world)
    canvas = torch.empty((45, 70, 3), dtype=torch.float32, device="cuda")
    rd.render_canvas(scene, cam, 5, canvas.view(-1))
    scene.close()
    canvas[0, 0] = torch.tensor([float("nan"), -1.0, 7.0])  # NaN -> 0, clamps
    got = rd.canvas_to_u8(dev, canvas).cpu().numpy()
    host = canvas.cpu().numpy()
    assert np.array_equal(got, Color.to_u8(host))
    assert got[0, 0].tolist() == [0, 0, 255]
    assert Canvas(70, 45, host).to_ppm_p6().endswith(got.tobytes())


def test_frame_loop_on_one_rank(dev):
    """FrameLoop (what bench.py steps): with one rank the float canvas is written in place; with as_u8 the tiles
    go through rt355_to_u8_device and rt355_untile_u8.  Both must give Camera::render's canvas."""
    from raytracer_amd import dist as rd
    from raytracer_amd.host import Color
    world, cam = scenes.glass(100, 70)
    scene = dev.upload(world)
    try:
        want = scene.render(cam, 5).pixels
        loop = rd.FrameLoop(scene, dev, 100, 70, 0, 1)
        for _ in range(3):
            got = loop.step(cam).cpu().numpy()
            assert np.array_equal(got, want)
        loop8 = rd.FrameLoop(scene, dev, 100, 70, 0, 1, as_u8=True)
        for _ in range(3):
            got8 = loop8.step(cam).cpu().numpy()
            assert got8.dtype == np.uint8 and np.array_equal(got8, Color.to_u8(want))
        # chains in flight, several frames per chain: different cameras back to back, the canvases come back in order
        cams = [scenes.glass(100, 70)[1]] + [Camera(100, 70, math.pi / 3, mx.view_transformation(mx.point(0.3 * k, 1.5, -5.0), mx.point(0, 1, 0), mx.vector(0, 1, 0))) for k in range(1, 7)]
        wants = [scene.render(c, 5).pixels for c in cams]
        for as_u8, in_flight, batch in ((False, 3, 1), (True, 3, 1), (False, 1, 4), (True, 2, 3), (False, 2, 2)):
            loop3 = rd.FrameLoop(scene, dev, 100, 70, 0, 1, as_u8=as_u8, in_flight=in_flight, batch=batch)
            st = Stats()
            outs = []
            for c in cams:
                o = loop3.step(c, st)
                if o is not None:
                    outs.append(o.cpu().numpy().copy())
            outs += [o.cpu().numpy() for o in loop3.flush(st)]
            loop3.close()
            assert len(outs) == len(cams) and int(st.total_radiance) > 7000
            for o, wnt in zip(outs, wants):
                assert np.array_equal(o, Color.to_u8(wnt) if as_u8 else wnt)
    finally:
        scene.close()


def _two_rank_loop_worker(rank, world_size, port, as_u8, q):
    import os
    import torch
    import torch.distributed as dist
    from raytracer_amd import dist as rd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)  # both ranks on GPU 0: gloo stages CUDA tensors
    try:
        torch.cuda.set_device(0)
        world, cam0 = scenes.glass(150, 100)
        cams = [cam0] + [scenes.orbit(cam0, 0.05 * k) for k in range(1, 7)]
        d = Device(0, 64)
        scene = d.upload(world)
        loop = rd.FrameLoop(scene, d, 150, 100, rank, world_size, as_u8=as_u8, in_flight=2, batch=3)
        outs = []
        for c in cams:
            o = loop.step(c)
            if o is not None:
                outs.append(o.cpu().numpy().copy())
        outs += [o.cpu().numpy() for o in loop.flush()]
        rays = loop.rays_traced
        loop.close()
        if rank == 0:
            wants = [scene.render(c, 5).pixels for c in cams]
            q.put((outs, wants, rays))
        else:
            assert outs == []
            q.put(rays)
        scene.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("as_u8", [False, True])
def test_frame_loop_two_ranks_on_one_gpu(as_u8):
    """The loop bench.py runs for N > 1, with the library rendering: two processes (both on GPU 0, gloo in place of
    RCCL) render their tiles of a camera path, three frames per chain and two chains in flight, gather them and
    un-tile on rank 0.  Every canvas must equal Camera::render of its camera, and the rays the two ranks count
    must add up to the rays of the whole frames."""
    import socket
    import torch.multiprocessing as mp
    from raytracer_amd.host import Color
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_loop_worker, args=(r, 2, port, as_u8, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    import time
    got, deadline = [], time.time() + 240
    while len(got) < 2 and time.time() < deadline:
        try:
            got.append(q.get(timeout=2))
        except queue.Empty:
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), [p.exitcode for p in procs]  # a rank died: do not wait for it
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.kill()
    assert len(got) == 2 and all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    (outs, wants, rays0), rays1 = (got[0], got[1]) if isinstance(got[0], tuple) else (got[1], got[0])
    assert len(outs) == len(wants) == 7
    for o, w in zip(outs, wants):
        assert np.array_equal(o, Color.to_u8(w) if as_u8 else w)
    world, cam0 = scenes.glass(150, 100)
    d = Device.get(0, 64)
    scene = d.upload(world)
    try:
        total = 0
        for c in [cam0] + [scenes.orbit(cam0, 0.05 * k) for k in range(1, 7)]:
            st = Stats()
            scene.render(c, 5, st)
            total += int(st.total_radiance + st.total_shadow) - (int(st.radiance_rays[0]) - 150 * 100)
    finally:
        scene.close()
    assert rays0 + rays1 == total


@pytest.mark.parametrize("real_bits", [64, 32])
def test_batch_of_frames_in_one_launch_chain(real_bits):
    """rt355_render_tiles_batch_begin / rt355_untile_batch: several cameras of one size rendered by ONE launch
    chain, as rank r of 3.  Every frame's tiles must be bit-identical to rt355_render_tiles of that camera
    alone, the statistics must be the sums, and the gathered rank buffers must un-tile to the frames'
    canvases.  Cameras of different sizes are refused."""
    import ctypes as C
    import torch
    from raytracer_amd import _abi
    from raytracer_amd import dist as rd
    w, h, n_ranks = 150, 100, 3
    world, cam0 = scenes.glass(w, h)
    cams = [cam0] + [Camera(w, h, math.pi / (3 + 0.2 * k), mx.view_transformation(mx.point(0.4 * k, 1.5 + 0.1 * k, -5.0), mx.point(0, 1, 0), mx.vector(0, 1, 0))) for k in range(1, 5)]
    dev = Device.get(0, real_bits)
    scene = dev.upload(world)
    lib = dev.lib
    try:
        n = rd.tiles_per_rank(w, h, n_ranks) * 1024 * 3
        cds = (_abi.CameraDesc * len(cams))(*[c.to_desc() for c in cams])
        gathered = torch.empty((n_ranks, len(cams) * n), dtype=torch.float32, device="cuda")
        for rank in range(n_ranks):
            singles, rays = [], 0
            for c in cams:
                t = torch.empty(n, dtype=torch.float32, device="cuda")
                st = Stats()
                rd.render_tiles(scene, c, 5, rank, n_ranks, t, st)
                singles.append(t)
                rays += int(st.total_radiance) + int(st.total_shadow)
            st = Stats()
            _abi.check(lib.rt355_render_tiles_batch_begin(dev.handle, scene.handle, cds, len(cams), 5, rank, n_ranks, C.c_void_p(gathered[rank].data_ptr())))
            _abi.check(lib.rt355_frame_end(dev.handle, C.byref(st)))
            assert int(st.total_radiance) + int(st.total_shadow) == rays
            assert torch.equal(gathered[rank], torch.cat(singles))
        canvases = torch.empty((len(cams), h, w, 3), dtype=torch.float32, device="cuda")
        _abi.check(lib.rt355_untile_batch(dev.handle, C.byref(cds[0]), n_ranks, len(cams), C.c_void_p(gathered.data_ptr()), C.c_void_p(canvases.data_ptr())))
        direct = torch.empty_like(canvases)
        _abi.check(lib.rt355_render_device_batch_begin(dev.handle, scene.handle, cds, len(cams), 5, C.c_void_p(direct.data_ptr())))
        _abi.check(lib.rt355_frame_end(dev.handle, None))
        for f, c in enumerate(cams):
            want = scene.render(c, 5).pixels
            assert np.array_equal(canvases[f].cpu().numpy(), want), f
            assert np.array_equal(direct[f].cpu().numpy(), want), f
        paths = scene.render_path(cams, 5)  # rt355_render_batch: the same chain, canvases on the host
        assert len(paths) == len(cams) and all(np.array_equal(paths[f].pixels, scene.render(c, 5).pixels) for f, c in enumerate(cams))
        # one size per chain
        other = Camera(w + 32, h, math.pi / 3, cam0.transform)
        bad = (_abi.CameraDesc * 2)(cam0.to_desc(), other.to_desc())
        rc = lib.rt355_render_device_batch_begin(dev.handle, scene.handle, bad, 2, 5, C.c_void_p(direct.data_ptr()))
        assert rc == -1 and b"share" in lib.rt355_last_error()  # RT355_EINVAL
        assert lib.rt355_render_device_batch_begin(dev.handle, scene.handle, cds, _abi.MAX_BATCH + 1, 5, C.c_void_p(direct.data_ptr())) == -1
    finally:
        scene.close()


@pytest.mark.parametrize("real_bits", [64, 32])
def test_multi_device_render_from_one_process(real_bits):
    """rt355_multi_*: the frame's tiles spread over several contexts driven by one process, one host thread per
    context, gathered to the first and un-tiled there.  A one-GPU box lists its device several times, which
    walks everything but the RCCL calls (the gather then uses device-to-device copies); the canvas must be
    bit-identical to rt355_render's for every device count, and the ray counts must add up."""
    from raytracer_amd import MultiDevice
    world, cam = scenes.glass(200, 150)
    dev = Device.get(0, real_bits)
    scene = dev.upload(world)
    st1 = Stats()
    want = scene.render(cam, 5, st1).pixels
    scene.close()
    for ids in ([0], [0, 0], [0, 0, 0]):
        multi = MultiDevice(ids, real_bits)
        try:
            assert multi.gather_kind() == ("none" if len(ids) == 1 else "peer copies")
            multi.upload(world)
            st = Stats()
            for _ in range(2):
                got = multi.render(cam, 5, st).pixels
                assert np.array_equal(got, want), ids
            # generation 0 counts tile padding, which depends on the split; the rest must add up exactly
            assert list(st.radiance_rays)[1:] == list(st1.radiance_rays)[1:] and list(st.hits) == list(st1.hits)
            # a camera path: every device renders its tiles of all the frames in one launch chain (rt355_multi_render_batch)
            cam2 = Camera(200, 150, math.pi / 3, mx.view_transformation(mx.point(0.6, 1.5, -5.0), mx.point(0, 1, 0), mx.vector(0, 1, 0)))
            want2 = dev.upload(world)
            try:
                w2 = want2.render(cam2, 5).pixels
            finally:
                want2.close()
            path = multi.render_path([cam, cam2, cam], 5)
            assert [np.array_equal(c.pixels, w_) for c, w_ in zip(path, (want, w2, want))] == [True] * 3, ids
            assert np.array_equal(multi.render(cam2, 5).pixels, w2)  # a shorter chain after a longer one
            multi.upload(World.new_default())  # a second scene replaces the first on every device
            small = Camera(40, 30, math.pi / 3, mx.view_transformation(mx.point(0, 0, -5), mx.point(0, 0, 0), mx.vector(0, 1, 0)))
            assert np.array_equal(multi.render(small, 5).pixels, small.render(World.new_default(), 5, dev).pixels)
        finally:
            multi.close()


def test_render_u8_and_pinned_canvas(dev):
    """rt355_render_u8 = Color::to_u8 of rt355_render; rt355_render into memory from rt355_host_alloc."""
    import ctypes as C
    from raytracer_amd.host import Color
    world, cam = scenes.glass(70, 45)
    scene = dev.upload(world)
    try:
        want = scene.render(cam, 5).pixels
        assert np.array_equal(scene.render_u8(cam, 5), Color.to_u8(want))
        n = 45 * 70 * 3
        p = dev.lib.rt355_host_alloc(n * 4)
        assert p
        try:
            cd = cam.to_desc()
            _abi.check(dev.lib.rt355_render(dev.handle, scene.handle, C.byref(cd), 5, C.c_void_p(p), None))
            got = np.ctypeslib.as_array((C.c_float * n).from_address(p)).reshape(45, 70, 3).copy()
        finally:
            dev.lib.rt355_host_free(C.c_void_p(p))
        assert np.array_equal(got, want)
    finally:
        scene.close()


def test_glass_scene_at_8k(oracle):
    """The glass scene at 7680x4320, the largest frame size of the configurations (33 M primary rays, a 31 GB
    ray pool in f64; the stress scene of that configuration is rendered at this size in test_gpu_configs.py):
    the 8-rank tile split reassembles to the frame rendered in one piece, bit for bit, and a few pixels
    agree with the oracle."""
    import torch
    from raytracer_amd import dist as rd
    w, h = 7680, 4320
    world, cam = scenes.glass(w, h)
    dev = Device.get(0, 64)
    scene = dev.upload(world)
    st = Stats()
    try:
        full = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        rd.render_canvas(scene, cam, 5, full.view(-1), st)
        n_ranks = 8
        tpr = rd.tiles_per_rank(w, h, n_ranks)
        gathered = torch.zeros((n_ranks, tpr * 1024 * 3), dtype=torch.float32, device="cuda")
        for r in range(n_ranks):
            rd.render_tiles(scene, cam, 5, r, n_ranks, gathered[r])
        out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        rd.untile(dev, cam, n_ranks, gathered, out)
        assert torch.equal(out, full)
    finally:
        scene.close()
    assert int(st.radiance_rays[0]) == rd.tiles_per_rank(w, h, 1) * 1024 and int(st.hits[0]) <= w * h
    rng = np.random.default_rng(8)
    pxy = np.stack([rng.integers(0, w, 100), rng.integers(0, h, 100)], axis=1)
    cam_rays, _ = oracle.camera_rays(cam, pxy, 64)
    want = oracle.color_at(world, cam_rays, 5, 64)
    got = full[torch.from_numpy(pxy[:, 1]).cuda(), torch.from_numpy(pxy[:, 0]).cuda()].cpu().numpy()
    bad = (np.abs(oracle.to_u8(got).astype(int) - oracle.to_u8(want).astype(int)).max(axis=1) > 1).mean()
    assert bad <= 0.03
