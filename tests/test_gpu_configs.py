"""GPU parity at the sizes the configurations are defined at, and the probes the smaller tests leave out.

  * C3 (teapot-high.obj, 6 320 smooth triangles, 3840x2160) and C5 (100 000 instances + floor, 7680x4320) at
    full size, through properties that do not need a full CPU render: tile splits reassemble bit for bit, ray
    counts obey the recursion's bounds, sampled pixels equal the oracle's World::color_at on the same camera
    rays, sampled rays equal the oracle's nearest hit;
  * every Pattern variant (src/pattern.rs:93-121) on the device;
  * World::is_shadowed (src/world.rs:45-57) on random points of every scene, against the oracle;
  * the graded precision (f64) against the reference's precision (f32 oracle) on every configuration: the
    mismatch fraction is printed and bounded;
  * nearest hit on >= 10^5 sampled rays per scene.

Bars: bit-exact where the arithmetic is the same on both sides (tile splits), |dt| <= EPSILON (src/epsilon.rs:1)
and the same shape index for hits, |du8| <= 1 per channel for pixels with the fraction outside printed.
"""
import math
import os

import numpy as np
import pytest

from raytracer_amd import Camera, Device, Material, Pattern, PointLight, Shape, World, scenes
from raytracer_amd import matrix as mx
from raytracer_amd._abi import Stats

pytestmark = pytest.mark.gpu
EPS = 1e-3
THREADS = max(1, min(16, os.cpu_count() or 1))  # oracle threads: the GPU box gives 16 cores to a one-GPU job


def u8_mismatch(oracle, a, b):
    d = np.abs(oracle.to_u8(a).astype(int) - oracle.to_u8(b).astype(int)).max(axis=-1)
    return float((d > 1).mean())


def camera_sample_rays(oracle, cam, n, seed):
    w, h = int(cam.horizontal_size), int(cam.vertical_size)
    rng = np.random.default_rng(seed)
    pxy = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], axis=1)
    rays, _ = oracle.camera_rays(cam, pxy, 64)
    return pxy, rays.astype(np.float32)


def scattered_rays(n, seed, lo, hi):
    """Rays between random point pairs of the box [lo, hi]: every direction, origins inside the scene."""
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    o = rng.uniform(lo, hi, size=(n, 3))
    t = rng.uniform(lo, hi, size=(n, 3))
    d = t - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def compare_hits(oracle, world, scene, rays, real_bits, allow_beyond=False):
    """rt355_trace_sample against World::intersect + hit of the oracle; returns (bad, compared, ties)."""
    t, idx = scene.trace(rays)
    to, io, gap, beyond = oracle.hit(world, rays, real_bits, with_beyond=True, threads=THREADS)
    tie = gap < (1e-6 if real_bits == 64 else EPS)
    if allow_beyond:
        tie = tie | beyond
    ok = ~tie
    same = ((io >= 0) == (idx >= 0)) & (idx == io)
    both = ok & (io >= 0) & same
    err = np.abs(t[both] - to[both])
    bound = EPS if real_bits == 64 else EPS * np.maximum(1.0, np.abs(to[both]))
    bad = int((ok & ~same).sum()) + int((err > bound).sum())
    return bad, int(ok.sum()), int(tie.sum()), int((io >= 0).sum())


def check_split(dev, scene, cam, depth, full, ranks):
    import torch
    from raytracer_amd import dist as rd
    w, h = int(cam.horizontal_size), int(cam.vertical_size)
    for n_ranks in ranks:
        tpr = rd.tiles_per_rank(w, h, n_ranks)
        gathered = torch.zeros((n_ranks, tpr * 1024 * 3), dtype=torch.float32, device="cuda")
        for r in range(n_ranks):
            rd.render_tiles(scene, cam, depth, r, n_ranks, gathered[r])
        out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        rd.untile(dev, cam, n_ranks, gathered, out)
        assert torch.equal(out, full), n_ranks
        del gathered, out


def check_count_bounds(st, w, h, depth):
    from raytracer_amd import dist as rd
    rays = [int(x) for x in st.radiance_rays]
    hits = [int(x) for x in st.hits]
    assert rays[0] == rd.tiles_per_rank(w, h, 1) * 1024 and hits[0] <= w * h  # generation 0 includes tile padding
    for g in range(depth):
        assert hits[g] <= rays[g] and rays[g + 1] <= 2 * hits[g]
    assert rays[depth + 1] == 0 and int(st.total_shadow) == sum(hits)


# ------------------------------------------------------------------------------------------ C3 at full size
def test_teapot_high_at_full_size(oracle):
    """Config 3: teapot-high.obj as 6 320 smooth triangles, Z-up camera, 3840x2160, f64."""
    import torch
    from raytracer_amd import dist as rd
    w, h = 3840, 2160
    world, cam = scenes.teapot_high(w, h)
    assert world.n_shapes() == 6320
    dev = Device.get(0, 64)
    scene = dev.upload(world)
    st = Stats()
    try:
        full = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        rd.render_canvas(scene, cam, 5, full.view(-1), st)
        check_split(dev, scene, cam, 5, full, (2, 8))
        check_count_bounds(st, w, h, 0)  # no reflective or transparent material: one generation
        pxy, rays = camera_sample_rays(oracle, cam, 50000, 3)
        more = scattered_rays(50000, 4, (-20, -15, -2), (20, 15, 18))
        bad, n, ties, hits = compare_hits(oracle, world, scene, np.concatenate([rays, more]), 64)
        print(f"teapot-high: {bad} of {n} sampled rays outside the bar, {ties} ties excluded, {hits} hits")
        assert bad == 0 and hits > 20000
    finally:
        scene.close()
    sel = slice(0, 600)
    want = oracle.color_at(world, rays[sel], 5, 64, threads=THREADS)
    got = full[torch.from_numpy(pxy[sel, 1]).cuda(), torch.from_numpy(pxy[sel, 0]).cuda()].cpu().numpy()
    frac = u8_mismatch(oracle, got, want)
    print(f"teapot-high 3840x2160: {frac:.3%} of 600 sampled pixels differ by more than 1/255")
    assert frac <= 0.005


# ------------------------------------------------------------------------------------------ C5 at full size
def test_stress_scene_at_full_size(oracle):
    """Config 5: 100 000 instances (cubes, closed cylinders, closed cones, spheres) from a 3-level group tree
    + floor = 101 012 entries of World.shapes (the groups stay as inert entries), 7680x4320, f64.  The cones
    live in the BVH here (more than 32 of them), so hits of the reference's parallel-ray branch beyond a
    cone's extent are the declared deviation: counted and excluded."""
    import torch
    from raytracer_amd import dist as rd
    w, h = 7680, 4320
    world, cam = scenes.stress(w, h)
    assert world.n_shapes() == 101012
    dev = Device.get(0, 64)
    scene = dev.upload(world)
    info = scene.info()
    print(f"stress scene: {info}")
    assert info["bounded"] == 100000 and info["unbounded"] == 1
    st = Stats()
    try:
        full = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        rd.render_canvas(scene, cam, 5, full.view(-1), st)
        check_split(dev, scene, cam, 5, full, (8,))
        check_count_bounds(st, w, h, 5)
        pxy, rays = camera_sample_rays(oracle, cam, 12000, 5)
        more = scattered_rays(8000, 6, (-200, 0.5, -200), (200, 42, 200))
        bad, n, ties, hits = compare_hits(oracle, world, scene, np.concatenate([rays, more]), 64, allow_beyond=True)
        print(f"stress scene: {bad} of {n} sampled rays outside the bar, {ties} ties / beyond-extent cone hits excluded, {hits} hits")
        assert bad == 0 and hits > 10000 and ties < 0.01 * (n + ties)
    finally:
        scene.close()
    sel = slice(0, 400)
    want = oracle.color_at(world, rays[sel], 5, 64, threads=THREADS)
    got = full[torch.from_numpy(pxy[sel, 1]).cuda(), torch.from_numpy(pxy[sel, 0]).cuda()].cpu().numpy()
    frac = u8_mismatch(oracle, got, want)
    print(f"stress 7680x4320: {frac:.3%} of 400 sampled pixels differ by more than 1/255")
    assert frac <= 0.02


# ------------------------------------------------------------------------------------------ patterns
def pattern_world():
    """One shape per Pattern variant (src/pattern.rs:6-12), a reflective patterned floor, a transparent
    gradient sphere: pattern_at runs under lighting for primary, reflected and refracted hits."""
    P = Pattern
    floor = Shape.Plane(None, Material(specular=0.2, reflective=0.25, pattern=P.Gradient((0.9, 0.2, 0.1), (0.1, 0.3, 0.9))))
    wall = Shape.Cube(mx.translation(0, 0, 6.0) @ mx.scaling(12.0, 12.0, 0.1), Material(pattern=P.Test()))
    shapes = [floor, wall]
    pats = [P.Stripe((1, 1, 1), (0.1, 0.5, 0.1)), P.Gradient((1, 1, 0), (0, 0, 1)), P.Ring((1, 0, 0), (1, 1, 1)),
            P.Checker((0.9, 0.9, 0.9), (0.2, 0.2, 0.2)), P.Test()]
    for k, p in enumerate(pats):
        shapes.append(Shape.Sphere(mx.translation(-4.0 + 2.0 * k, 1.0, 0.5 * (k % 2)), Material(pattern=p, specular=0.4)))
    shapes.append(Shape.Sphere(mx.translation(0.3, 0.8, -2.2) @ mx.scaling(0.8, 0.8, 0.8),
                               Material(transparency=0.7, refractive=1.3, reflective=0.2, pattern=P.Gradient((0.2, 0.9, 0.4), (0.9, 0.1, 0.8)))))
    shapes.append(Shape.Cylinder(mx.translation(3.0, 0, -2.0) @ mx.scaling(0.5, 1.0, 0.5), Material(pattern=P.Test(), ambient=0.3), 0.0, 1.5, True))
    world = World(PointLight((-6.0, 9.0, -8.0), (1.0, 0.95, 0.9)), shapes)
    cam = Camera(200, 120, math.pi / 3, mx.view_transformation(mx.point(0.5, 3.0, -9.0), mx.point(0, 1, 0), mx.vector(0, 1, 0)))
    return world, cam


@pytest.mark.parametrize("real_bits", [64, 32])
def test_every_pattern_variant(oracle, real_bits):
    """Stripe, Gradient, Ring, Checker and Test patterns through Material::lighting on the device.  The
    reference has no pattern tests of its own, so the oracle's pattern_at is pinned only by reading
    src/pattern.rs:93-121 (parity unpinned for this function); device and oracle must agree."""
    world, cam = pattern_world()
    dev = Device.get(0, real_bits)
    scene = dev.upload(world)
    try:
        got = scene.render(cam, 5).pixels
        rays = scattered_rays(20000, 9, (-6, 0.05, -6), (6, 4, 5))
        c_dev = scene.color_at(rays, 5)
    finally:
        scene.close()
    want, _, _ = oracle.render(world, cam, 5, real_bits, threads=THREADS)
    frac = u8_mismatch(oracle, got, want)
    c_or = oracle.color_at(world, rays, 5, real_bits, threads=THREADS)
    tol = 1e-4 if real_bits == 64 else 2e-2
    bad = float((np.abs(c_dev - c_or).max(axis=1) > tol).mean())
    print(f"patterns real={real_bits}: image mismatch {frac:.4%}; {bad:.3%} of 20000 rays differ by more than {tol}")
    assert frac <= (0.003 if real_bits == 64 else 0.01)
    assert bad <= (0.003 if real_bits == 64 else 0.03)
    # every variant really was on screen: the Test pattern paints colour = position, the gradient mixes
    assert got[..., 0].max() > 1.0 or got[..., 1].max() > 1.0


# ------------------------------------------------------------------------------------------ is_shadowed
SHADOW_SCENES = [
    ("ch9", lambda: scenes.ch9(64, 48), (-4, -0.5, -4), (4, 4, 4)),
    ("glass", lambda: scenes.glass(64, 48), (-4, -0.5, -4), (4, 4, 9)),
    ("teapot_low", lambda: scenes.teapot_low(64, 36), (-20, -15, -2), (20, 15, 18)),
    ("north_star", lambda: scenes.north_star(64, 36), (-32, -1, -20), (22, 18, 22)),
    ("stress_small", lambda: scenes.stress(64, 36, n_top=2, n_mid=6, n_leaf=25, extent=40.0), (-22, -0.5, -22), (22, 6, 22)),
]


@pytest.mark.parametrize("real_bits", [64, 32])
@pytest.mark.parametrize("name,build,lo,hi", SHADOW_SCENES, ids=[s[0] for s in SHADOW_SCENES])
def test_is_shadowed_on_random_points(oracle, name, build, lo, hi, real_bits):
    """World::is_shadowed (src/world.rs:45-57) for 10^5 random points of the scene's volume: the device's
    any-hit traversal against the oracle's full intersect + hit.  A point whose shadow ray grazes a surface
    flips with the last bit, so a few disagreements are allowed and counted."""
    world, _ = build()
    rng = np.random.default_rng(21)
    pts = rng.uniform(lo, hi, size=(100000, 3)).astype(np.float32)
    dev = Device.get(0, real_bits)
    if name == "stress_small":
        dev.set_exact_cones(1)
    try:
        scene = dev.upload(world)
        got = scene.is_shadowed(pts)
        scene.close()
    finally:
        dev.set_exact_cones(0)
    want = oracle.is_shadowed(world, pts, real_bits, threads=THREADS)
    diff = int((got != want).sum())
    print(f"is_shadowed {name} real={real_bits}: {diff} of {len(pts)} points differ; {int(want.sum())} shadowed")
    assert 0.02 * len(pts) < want.sum() < 0.98 * len(pts)  # both answers occur
    assert diff <= (20 if real_bits == 64 else 300)


# ------------------------------------------------------------------------------------------ f64 device vs f32 oracle
PRECISION_SCENES = [
    ("C1 ch9", lambda: scenes.ch9(200, 150), 0.005),
    ("C2 teapot_low", lambda: scenes.teapot_low(192, 108), 0.01),
    ("C3 teapot_high", lambda: scenes.teapot_high(192, 108), 0.01),
    ("C4 glass", lambda: scenes.glass(160, 120), 0.02),
    ("C5 stress (1200 instances)", lambda: scenes.stress(160, 90, n_top=2, n_mid=6, n_leaf=100, extent=60.0), 0.02),
    ("NS north_star", lambda: scenes.north_star(160, 90), 0.02),
]


@pytest.mark.parametrize("name,build,bar", PRECISION_SCENES, ids=[s[0].split()[0] for s in PRECISION_SCENES])
def test_f64_device_against_the_f32_reference_precision(oracle, name, build, bar):
    """The reference computes in f32 (src/epsilon.rs:1, src/camera.rs:12-18); the graded device
    configuration computes in f64.  For every configuration: the fraction of pixels whose u8 value differs by
    more than 1 between the f64 device and the f32 oracle, next to the f64-oracle-vs-f32-oracle fraction
    (what precision alone changes) -- printed, and bounded."""
    world, cam = build()
    dev = Device.get(0, 64)
    scene = dev.upload(world)
    try:
        got = scene.render(cam, 5).pixels
    finally:
        scene.close()
    want32, _, _ = oracle.render(world, cam, 5, 32, threads=THREADS)
    want64, _, _ = oracle.render(world, cam, 5, 64, threads=THREADS)
    f_dev = u8_mismatch(oracle, got, want32)
    f_prec = u8_mismatch(oracle, want64, want32)
    f_same = u8_mismatch(oracle, got, want64)
    print(f"{name}: f64 device vs f32 oracle {f_dev:.4%}; f64 oracle vs f32 oracle {f_prec:.4%}; f64 device vs f64 oracle {f_same:.4%}")
    assert f_dev <= bar
    assert f_dev <= f_prec + f_same + 1e-3  # the device adds nothing beyond what precision and its own bar explain


def test_ch11_f32_outside_the_ill_conditioned_pixels(oracle):
    """ch11 (src/main.rs:389-535) in the reference's precision.  Its walls are rotated planes whose normal
    stays (0,1,0) (src/shape.rs:144), so over_point never leaves the wall and the shadow test there is decided
    by the last bit; its floor is a sphere squashed thinner than 10 EPSILON.  Those pixels are found with the
    oracle alone: wherever its f32 render differs from its f64 render, or from any of six f32 renders whose
    camera is moved by ~1e-5 (a few tens of ulps of the camera position, a thousandth of a pixel).  About 70 %
    of this frame is such noise in f32.  Those pixels and their neighbours are masked; everywhere else the
    f32 device must match the f32 oracle like on any other scene."""
    w, h = 160, 120
    world, cam = scenes.ch11(w, h)
    dev = Device.get(0, 32)
    scene = dev.upload(world)
    try:
        got = scene.render(cam, 5).pixels
    finally:
        scene.close()

    def oracle_u8(bits, jitter=None):
        c = cam
        if jitter is not None:
            frm = np.array([0.0, 1.5, -5.0]) + jitter
            c = Camera(w, h, math.pi / 3, mx.view_transformation(mx.point(*frm), mx.point(0, 1, 0), mx.vector(0, 1, 0)))
        img, _, _ = oracle.render(world, c, 5, bits, threads=THREADS)
        return oracle.to_u8(img).astype(int)

    base = oracle_u8(32)
    rng = np.random.default_rng(5)
    ill = np.abs(base - oracle_u8(64)).max(axis=2) > 1
    f_prec = float(ill.mean())
    for _ in range(6):
        ill |= np.abs(base - oracle_u8(32, rng.normal(size=3) * 1e-5)).max(axis=2) > 1
    grow = ill.copy()  # one pixel of margin: the flips move with sub-pixel offsets
    grow[1:, :] |= ill[:-1, :]
    grow[:-1, :] |= ill[1:, :]
    grow[:, 1:] |= ill[:, :-1]
    grow[:, :-1] |= ill[:, 1:]
    d_dev = np.abs(oracle.to_u8(got).astype(int) - base).max(axis=2)
    outside = ~grow
    frac = float((d_dev[outside] > 1).mean())
    print(f"ch11 f32: f32 vs f64 oracle differ on {f_prec:.2%} of the pixels, {ill.mean():.2%} unstable under 1e-5 camera jitter, "
          f"{grow.mean():.2%} masked; outside the mask {frac:.3%} of {int(outside.sum())} pixels differ; unmasked overall {float((d_dev > 1).mean()):.2%}")
    assert outside.mean() > 0.08
    assert frac <= 0.01


# ------------------------------------------------------------------------------------------ >= 10^5 rays per scene
RAY_SCENES = [
    ("ch9", lambda: scenes.ch9(64, 48), (-5, -0.5, -6), (5, 4, 5), False),
    ("ch11", lambda: scenes.ch11(64, 48), (-6, -0.5, -6), (6, 4, 8), False),
    ("glass", lambda: scenes.glass(64, 48), (-5, -0.5, -6), (5, 4, 9), False),
    ("teapot_low", lambda: scenes.teapot_low(64, 36), (-22, -16, -3), (22, 16, 20), False),
    ("north_star", lambda: scenes.north_star(64, 36), (-32, -1, -20), (22, 20, 45), False),
    ("stress_small", lambda: scenes.stress(64, 36, n_top=2, n_mid=6, n_leaf=25, extent=40.0), (-22, 0.2, -22), (22, 7, 22), True),
]


@pytest.mark.parametrize("name,build,lo,hi,cones", RAY_SCENES, ids=[s[0] for s in RAY_SCENES])
def test_nearest_hit_on_1e5_rays(oracle, name, build, lo, hi, cones):
    """rt355_trace_sample against World::intersect + hit (src/world.rs:35-44, src/intersection.rs:149-154) on
    10^5 rays per scene in f64: half through the scene's camera, half between random points of its volume."""
    world, cam = build()
    big = Camera(1000, 600, float(cam.field_of_view), cam.transform)
    _, rays = camera_sample_rays(oracle, big, 50000, 31)
    rays = np.concatenate([rays, scattered_rays(50000, 32, lo, hi)])
    dev = Device.get(0, 64)
    if cones:
        dev.set_exact_cones(1)
    try:
        scene = dev.upload(world)
        bad, n, ties, hits = compare_hits(oracle, world, scene, rays, 64)
        scene.close()
    finally:
        dev.set_exact_cones(0)
    print(f"{name}: {bad} of {n} rays outside the bar, {ties} ties excluded, {hits} hits")
    assert bad == 0 and hits > 30000 and n > 90000


# ------------------------------------------------------------------------------------------ smaller regressions
def test_axis_parallel_ray_through_glass_inside_a_refractive_cube(oracle):
    """n1 / n2 for rays with exactly-zero direction components: the device looks for the enclosing shapes
    along the REVERSED ray, and the cube's slab test (src/shape.rs:455-466) branches on d >= 0, which a
    negated zero passes.  A glass sphere inside a refractive cube, rays along the axes."""
    g = Material.new_glass()
    water = Material(color=(0.2, 0.3, 0.9), transparency=0.7, refractive=1.333, diffuse=0.3)
    world = World(PointLight((-3, 8, -6), (1, 1, 1)), [
        Shape.Cube(mx.scaling(2.0, 2.0, 2.0), water),
        Shape.Sphere(None, g),
        Shape.Plane(mx.translation(0, -3, 0), Material(pattern=Pattern.Checker((1, 1, 1), (0.1, 0.1, 0.1)))),
    ])
    rays = []
    for o, d in [((0, 0, -5), (0, 0, 1)), ((0.2, 0.1, -5), (0, 0, 1)), ((-5, 0.3, 0.1), (1, 0, 0)), ((0.1, 5, 0.2), (0, -1, 0)),
                 ((0.3, -0.2, 5), (0, 0, -1)), ((0, 0, 0), (0, 0, 1)), ((0, 0, 1.5), (0, 0, 1)), ((0.5, 0.5, -5), (0, 0, 1)),
                 ((0, 0.4, -5), (0.0, -0.0, 1.0)), ((5, 0.1, 0.3), (-1, 0, -0.0))]:
        rays.append(list(o) + list(d))
    rays = np.asarray(rays, dtype=np.float32)
    for real_bits in (64, 32):
        dev = Device.get(0, real_bits)
        scene = dev.upload(world)
        dev.set_pool(64.0)
        try:
            got = scene.color_at(rays, 5)
        finally:
            dev.set_pool(8.0)
            scene.close()
        want = oracle.color_at(world, rays, 5, real_bits)
        off = np.abs(got - want).max(axis=1) > (1e-5 if real_bits == 64 else 5e-3)
        # A ray of this axis-aligned arrangement may lie EXACTLY on a discontinuity of the image (ray 7, through
        # x = y = 0.5, does from the fourth bounce on): which side it falls on is decided by the last bit of a
        # normalised vector, and the device forms 1 / |v| differently from the oracle (both good to an ulp).  Such a ray
        # must then carry the colour of one of the two sides: the oracle's own colours a step of 1e-4 to either side
        # of the ray must differ from each other, and the device must match one of them.
        for k in np.nonzero(off)[0]:
            o, d = rays[k, :3].astype(np.float64), rays[k, 3:].astype(np.float64)
            side = np.eye(3)[np.argsort(np.abs(d))[:2]]  # the two axes across the ray
            near = np.asarray([list(o + s * 1e-4 * a) + list(d) for a in side for s in (-1.0, 1.0)], dtype=np.float32)
            sides = oracle.color_at(world, near, 5, real_bits)
            assert np.abs(sides - sides[0]).max() > 1e-2, (real_bits, k, "differs from the oracle away from any discontinuity", got[k], want[k])
            assert np.abs(sides - got[k]).max(axis=1).min() < 1e-3, (real_bits, k, got[k], sides)
        # in f32 the oracle itself changes its answer for the last of these rays (it differs from its own f64
        # result): a threshold of this axis-aligned arrangement flips with single-precision rounding
        assert off.sum() <= 1, (real_bits, got, want)


@pytest.mark.parametrize("mesh,n_copies,n_shapes", [("teapot_low", 112, 240), ("teapot_high", 3120, 6320)])
@pytest.mark.parametrize("real_bits", [64, 32])
def test_the_teapot_as_the_references_own_parser_builds_it(oracle, real_bits, mesh, n_copies, n_shapes):
    """Wavefront::parse fans every polygon over the GLOBAL vertex list (src/wavefront.rs:101-116, :143-155), so the
    reference's teapot-low is 112 copies of triangle (V1, V2, V3), 112 copies of (V1, V3, V4) and 16 real triangles
    (teapot-high: 3 120 + 3 120 + 80).
    That is the World the reference renders from the file, and a hard case for the nearest-hit tie rule (stable sort
    src/world.rs:42 + first minimum src/intersection.rs:151: among equal t the lowest shape index wins): coincident
    triangles by the hundred, boxes that cannot be told apart, leaves forced at the depth limit.  Rays aimed at the
    copies must report the FIRST copy; every other ray the oracle's shape or, where two different triangles lie
    within rounding of each other, one of the two."""
    scenes.REF_PARSER_QUIRK = True
    try:
        world, cam = scenes.CONFIGS[mesh](160, 120)
    finally:
        scenes.REF_PARSER_QUIRK = False
    key = [tuple(np.asarray([s.p1, s.p2, s.p3], dtype=np.float32).ravel().tolist()) for s in world.shapes]
    first_of = {}
    for i, k in enumerate(key):
        first_of.setdefault(k, i)
    copies = {k: sum(1 for q in key if q == k) for k in set(key)}
    big = sorted(copies.values())[-2:]
    assert len(key) == n_shapes and big == [n_copies, n_copies], (len(key), big)
    rng = np.random.default_rng(5)
    rays = [camera_sample_rays(oracle, cam, 6000, 3)[1], scattered_rays(6000, 4, (-18, -12, -2), (20, 12, 18))]
    # rays aimed into the two copied triangles, from all around
    for k, n in copies.items():
        if n < 100:
            continue
        v = np.asarray(k, dtype=np.float64).reshape(3, 3)
        w = rng.dirichlet((1.0, 1.0, 1.0), size=3000)
        target = w @ v
        org = target + rng.normal(size=(3000, 3)) * 8.0
        d = target - org
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays.append(np.concatenate([org, d], axis=1).astype(np.float32))
    rays = np.concatenate(rays, axis=0)
    dev = Device.get(0, real_bits)
    scene = dev.upload(world)
    try:
        t, idx = scene.trace(rays)
        got = scene.render(cam, 5).pixels.copy()
    finally:
        scene.close()
    to, io, gap = oracle.hit(world, rays, real_bits, threads=THREADS)
    on_copies = 0
    for r in np.nonzero(io >= 0)[0]:
        if copies[key[io[r]]] >= 100:  # the oracle's own answer is the first copy, and so must the device's be
            on_copies += 1
            assert io[r] == first_of[key[io[r]]]
    assert on_copies > 1000, on_copies
    differ = np.nonzero(idx != io)[0]
    # a different shape only where the oracle itself sees two different triangles within rounding of each other
    for r in differ:
        assert idx[r] >= 0 and io[r] >= 0 and key[idx[r]] != key[io[r]] and gap[r] < (1e-6 if real_bits == 64 else EPS), (r, idx[r], io[r], gap[r])
    assert len(differ) <= (0.002 if real_bits == 64 else 0.02) * len(rays), len(differ)
    both = (idx == io) & (io >= 0)
    err = np.abs(t[both] - to[both])
    assert (err <= (EPS if real_bits == 64 else EPS * np.maximum(1.0, np.abs(to[both])))).all()
    want, _, _ = oracle.render(world, cam, 5, real_bits, threads=THREADS)
    frac = u8_mismatch(oracle, got, want)
    print(f"reference-parser {mesh}, real={real_bits}: {on_copies} rays on the copied triangles, {len(differ)} of {len(rays)} rays on another "
          f"shape than the oracle (near-ties of different triangles), image mismatch {frac:.3%}")
    assert frac <= (0.004 if real_bits == 64 else 0.02)


@pytest.mark.parametrize("real_bits", [64, 32])
def test_headline_frame_is_linear_in_the_light(real_bits):
    """A property that holds at any size and needs no oracle: every term of Material::lighting
    (src/material.rs:108-162) carries the light's intensity once, per channel, and shade_hit (src/world.rs:104-127)
    only adds and scales colours, so the frame is linear in the intensity.  Scaled by powers of two -- every product
    and sum then scales exactly -- the headline frame at its full 1920x1080 must come out scaled bit for bit, through
    all six generations, the Schlick mix and the conversion to the float canvas."""
    world, cam = scenes.north_star(1920, 1080)
    dev = Device.get(0, real_bits)
    scene = dev.upload(world)
    try:
        base = scene.render(cam, 5).pixels.copy()
    finally:
        scene.close()
    factor = np.asarray([0.5, 0.25, 2.0], dtype=np.float32)
    li = world.pointlight.intensity
    world.pointlight.intensity = tuple(float(a) * float(f) for a, f in zip(li, factor))
    scene = dev.upload(world)
    try:
        scaled = scene.render(cam, 5).pixels.copy()
    finally:
        scene.close()
    assert base.max() > 0.5 and np.isfinite(base).all()
    assert np.array_equal(scaled, base * factor)


def test_generations_do_not_depend_on_the_depth_limit():
    """Another size-independent property, at the headline's full size: a ray of generation g exists and hits
    whatever `remaining` it started from, as long as g is within the limit (src/world.rs:69, :79 only ask whether
    remaining is zero).  Rendered with depth limits 2, 3 and 5, the frame must trace the same number of rays and
    count the same hits in every generation the limits share, and a limit of 0 is the primary rays alone."""
    world, cam = scenes.north_star(1920, 1080)
    dev = Device.get(0, 64)
    scene = dev.upload(world)
    stats = {}
    try:
        for depth in (0, 2, 3, 5):
            st = Stats()
            scene.render(cam, depth, st)
            stats[depth] = ([int(x) for x in st.radiance_rays], [int(x) for x in st.hits])
    finally:
        scene.close()
    full_rays, full_hits = stats[5]
    for depth in (0, 2, 3):
        rays, hits = stats[depth]
        assert rays[: depth + 1] == full_rays[: depth + 1] and hits[: depth + 1] == full_hits[: depth + 1], (depth, rays, full_rays)
        assert not any(rays[depth + 1:]) and not any(hits[depth + 1:])
    assert full_rays[0] >= 1920 * 1080 and all(full_rays[g + 1] <= 2 * full_hits[g] for g in range(5))


def test_negative_coefficients_follow_the_reference(oracle):
    """src/world.rs:69,79 test `== 0.0` (a negative reflective / transparency still spawns its ray, weighted
    negatively) while src/world.rs:120 tests `> 0.0` (Schlick only when both are positive)."""
    odd = Material(color=(0.6, 0.7, 0.8), reflective=-0.3, transparency=0.5, refractive=1.4)
    world = World(None, [Shape.Sphere(None, odd), Shape.Plane(mx.translation(0, -1.5, 0), Material(reflective=0.4, color=(0.9, 0.5, 0.2))),
                         Shape.Sphere(mx.translation(2.5, 0.3, 1.0), Material(color=(0.1, 0.8, 0.3)))])
    rays = scattered_rays(4000, 41, (-4, -1, -5), (4, 3, 4))
    for real_bits in (64, 32):
        dev = Device.get(0, real_bits)
        scene = dev.upload(world)
        dev.set_pool(64.0)
        try:
            got = scene.color_at(rays, 5)
        finally:
            dev.set_pool(8.0)
            scene.close()
        want = oracle.color_at(world, rays, 5, real_bits, threads=THREADS)
        tol = 1e-4 if real_bits == 64 else 2e-2
        bad = float((np.abs(got - want).max(axis=1) > tol).mean())
        print(f"negative coefficients real={real_bits}: {bad:.3%} of {len(rays)} rays differ by more than {tol}")
        assert bad <= (0.003 if real_bits == 64 else 0.03)


def test_bad_camera_sizes_are_rejected():
    from raytracer_amd._abi import Rt355Error
    dev = Device.get(0, 64)
    scene = dev.upload(World.new_default())
    try:
        for hs, vs in [(float("nan"), 8), (8, float("inf")), (0, 8), (-5, 8), (1e9, 8), (8, 3e10)]:
            with pytest.raises(Rt355Error, match="hsize"):
                scene.render(_RawCamera(hs, vs), 5)
    finally:
        scene.close()


class _RawCamera:
    """A camera descriptor with sizes that Camera() itself would not produce."""

    def __init__(self, hs, vs):
        self.horizontal_size, self.vertical_size = 4, 4  # size of the (unused) output buffer
        self._hs, self._vs = hs, vs

    def to_desc(self):
        from raytracer_amd import _abi
        d = _abi.CameraDesc()
        d.hsize, d.vsize, d.fov = self._hs, self._vs, 1.0
        d.transform[:] = mx.to_abi(mx.identity()).tolist()
        return d


@pytest.mark.parametrize("real_bits", [64, 32])
def test_cube_slab_test_sees_the_sign_of_a_zero_direction(oracle, real_bits):
    """src/shape.rs:455-466 branches on `direction >= 0`, which -0.0 passes: the quotients then come out with swapped
    signs and a ray lying INSIDE that slab misses the cube.  The sign of a zero component after inverse * direction is
    that of the four-term row sum (src/ray.rs:35), whose w term m[k][3] * 0.0 is -0.0 for a cube moved to a positive
    position.  Device and oracle must agree ray by ray, for both signs of the translation and of the zero."""
    worlds = []
    for tx in (3.0, -3.0, 0.0):
        cube = Shape.Cube(mx.translation(tx, 0.0, 0.0)) if tx else Shape.Cube()
        worlds.append((tx, World(None, [cube])))
    dirs = [(-0.0, -1.0, -1.0), (0.0, -1.0, -1.0), (-0.0, 1.0, -1.0), (-0.0, -1.0, 1.0), (0.0, 1.0, 1.0)]
    dev = Device.get(0, real_bits)
    n_miss = n_hit = 0
    for tx, world in worlds:
        rr = np.array([[tx + 0.5, -5.0 * d[1], -5.0 * d[2], d[0], d[1] / math.sqrt(2), d[2] / math.sqrt(2)] for d in dirs], dtype=np.float32)
        scene = dev.upload(world)
        t_dev, i_dev = scene.trace(rr)
        scene.close()
        t_or, i_or, _ = oracle.hit(world, rr, real_bits)
        for k in range(len(dirs)):
            assert (i_dev[k] >= 0) == (i_or[k] >= 0), (tx, dirs[k], t_dev[k], t_or[k])
            if i_or[k] >= 0:
                assert abs(t_dev[k] - t_or[k]) <= 1e-3
                n_hit += 1
            else:
                n_miss += 1
    assert n_miss > 0 and n_hit > 0  # both outcomes occur: the sign of the zero decides


# ------------------------------------------------------------------------------------------ instanced meshes
def _instancing(on):
    """RT355_INSTANCE_MIN is read at every upload: a run of at least that many triangles sharing one non-identity
    transformation becomes one instance over an object-space tree."""
    if on:
        os.environ.pop("RT355_INSTANCE_MIN", None)
    else:
        os.environ["RT355_INSTANCE_MIN"] = str(1 << 30)


@pytest.mark.parametrize("where", ["between-the-teapots", "inside-a-teapots-box"])
@pytest.mark.parametrize("real_bits", [64, 32])
def test_glass_ball_among_instanced_meshes(oracle, real_bits, where):
    """A glass ball in a field of instanced teapots: n1 / n2 (src/intersection.rs:86-127) in the kernels that walk two
    levels of trees.  Between the teapots nothing reaches into the ball's box and the search inside it is decided at
    upload; moved into the (world-space) box of one instance -- though not into the teapot itself -- the search must run
    and descend into that mesh.  Colours of rays from all around against the oracle, which knows neither trees nor
    instances."""
    world, cam = scenes.teapot_field(160, 90, "teapot-low.obj", 2)
    g = Material.new_glass()
    g.reflective = 0.5
    # instance boxes of the 2 x 2 field sit around (+-26, ., +-26); the teapots are about 35 wide before scaling
    centre = (0.0, 9.0, 0.0) if where == "between-the-teapots" else (26.0 - 9.0, 14.0, -26.0 + 12.0)
    world.shapes.append(Shape.Sphere(mx.translation(*centre) @ mx.scaling(4.0, 4.0, 4.0), g))
    rng = np.random.default_rng(29)
    n = 4000
    tgt = np.asarray(centre) + rng.normal(size=(n, 3)) * 2.5
    org = np.asarray(centre) + rng.normal(size=(n, 3)) * 14.0
    org[: n // 4] = np.asarray(centre) + rng.normal(size=(n // 4, 3)) * 1.5  # a quarter of the rays start inside the ball
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d], axis=1).astype(np.float32)
    dev = Device.get(0, real_bits)
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
    want = oracle.color_at(world, rays, 5, real_bits, threads=THREADS)
    tol = 1e-4 if real_bits == 64 else 2e-2
    bad = (np.abs(got - want).max(axis=1) > tol).mean()
    from raytracer_amd import _abi
    searched = sum(int(st.launch_box_tests[i]) for i in range(st.n_launch) if st.launch_kind[i] == _abi.K_SPAWN)
    print(f"glass ball {where}, real={real_bits}: {bad:.3%} of {n} rays differ by more than {tol}; box tests of the n1/n2 search: {searched}")
    assert bad <= (0.004 if real_bits == 64 else 0.03)
    assert (searched == 0) == (where == "between-the-teapots")


@pytest.mark.parametrize("real_bits", [64, 32])
def test_group_transformed_meshes_as_instances(oracle, real_bits):
    """16 copies of teapot-low.obj under 16 groups with transformations of their own, flattened the reference's way
    (src/group.rs:130-168): the upload turns every copy into one instance with an object-space tree below a top-level
    tree.  Nearest hits on 10^5 rays against World::intersect + hit, the image against the oracle, and both bit for
    bit against the same scene uploaded with instancing switched off (every triangle a world-space leaf)."""
    import torch
    w, h = 160, 90
    world, cam = scenes.teapot_field(w, h, "teapot-low.obj", 4, through_groups=True)
    direct, _ = scenes.teapot_field(w, h, "teapot-low.obj", 4, through_groups=False)
    assert world.n_shapes() == direct.n_shapes() + 1  # the root group is one more inert entry
    dev = Device.get(0, real_bits)
    big = Camera(1000, 600, float(cam.field_of_view), cam.transform)
    _, rays = camera_sample_rays(oracle, big, 50000, 41)
    rays = np.concatenate([rays, scattered_rays(50000, 42, (-110, 0.2, -110), (110, 30, 110))])
    try:
        _instancing(True)
        scene = dev.upload(world)
        info = scene.info()
        assert info["bounded"] == 16 and info["unbounded"] == 1, info  # 16 instances + the floor
        bad, n, ties, hits = compare_hits(oracle, world, scene, rays, real_bits)
        print(f"teapot field f{real_bits}: {bad} of {n} rays outside the bar, {ties} ties excluded, {hits} hits")
        assert bad == 0 and hits > 30000 and n > 90000
        t_inst, i_inst = scene.trace(rays)
        st = Stats()
        img_inst = scene.render(cam, 5, st).pixels.copy()
        scene.close()
        _instancing(False)
        scene = dev.upload(world)
        assert scene.info()["bounded"] == 16 * 240
        t_flat, i_flat = scene.trace(rays)
        img_flat = scene.render(cam, 5).pixels.copy()
        scene.close()
    finally:
        _instancing(True)
    diff = np.flatnonzero(~((t_inst == t_flat) | (np.isnan(t_inst) & np.isnan(t_flat))))
    print(f"t differs on {len(diff)} rays", [(int(k), int(i_inst[k]), float(t_inst[k]), float(t_flat[k])) for k in diff[:8]])
    assert np.array_equal(i_inst, i_flat) and len(diff) == 0
    assert np.array_equal(img_inst, img_flat)
    assert int(st.radiance_rays[1]) > 0  # the floor reflects: secondary rays cross the instances too
    want, _, _ = oracle.render(world, cam, 5, real_bits, threads=THREADS)
    frac = u8_mismatch(oracle, img_inst, want)
    print(f"teapot field {w}x{h} f{real_bits}: {frac:.3%} of the pixels differ by more than 1/255")
    assert frac <= (0.002 if real_bits == 64 else 0.01)


def test_sixteen_teapot_high_instances_at_full_size(oracle):
    """16 copies of teapot-high.obj (101 120 triangles) under 16 group transformations at 1920x1080, f64: sampled
    nearest hits and pixels against the oracle, and the cost per ray beside one teapot's and beside the same
    scene with every triangle as a world-space leaf."""
    import torch
    from raytracer_amd import dist as rd
    w, h = 1920, 1080
    world, cam = scenes.teapot_field(w, h, "teapot-high.obj", 4)
    one, cam_one = scenes.teapot_field(w, h, "teapot-high.obj", 1)
    dev = Device.get(0, 64)

    def timed(scene, camera):
        out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        st = Stats()
        best = 1e30
        for _ in range(4):
            rd.render_canvas(scene, camera, 5, out.view(-1), st)
            best = min(best, float(st.frame_ms))
        return out, best, int(st.total_radiance) + int(st.total_shadow)

    try:
        _instancing(True)
        scene = dev.upload(world)
        info = scene.info()
        assert info["bounded"] == 16, info
        full, ms_inst, rays_inst = timed(scene, cam)
        pxy, rays = camera_sample_rays(oracle, cam, 60000, 43)
        more = scattered_rays(40000, 44, (-110, 0.2, -110), (110, 30, 110))
        bad, n, ties, hits = compare_hits(oracle, world, scene, np.concatenate([rays, more]), 64)
        print(f"16 x teapot-high: {bad} of {n} sampled rays outside the bar, {ties} ties excluded, {hits} hits")
        assert bad == 0 and hits > 30000
        scene.close()
        scene = dev.upload(one)
        _, ms_one, rays_one = timed(scene, cam_one)
        scene.close()
        _instancing(False)
        scene = dev.upload(world)
        flat, ms_flat, rays_flat = timed(scene, cam)
        scene.close()
    finally:
        _instancing(True)
    assert torch.equal(full, flat) and rays_flat == rays_inst
    per_inst, per_one, per_flat = ms_inst / rays_inst, ms_one / rays_one, ms_flat / rays_flat
    print(f"16 x teapot-high 1920x1080 f64: instanced {ms_inst:.3f} ms / {rays_inst} rays, one teapot {ms_one:.3f} ms / {rays_one} rays, "
          f"world-space leaves {ms_flat:.3f} ms; per ray: instanced {per_inst / per_one:.2f}x one teapot, world-space {per_flat / per_one:.2f}x")
    assert per_inst <= 1.5 * per_one
    sel = slice(0, 500)
    want = oracle.color_at(world, rays[sel], 5, 64, threads=THREADS)
    got = full[torch.from_numpy(pxy[sel, 1]).cuda(), torch.from_numpy(pxy[sel, 0]).cuda()].cpu().numpy()
    frac = u8_mismatch(oracle, got, want)
    print(f"16 x teapot-high 1920x1080: {frac:.3%} of 500 sampled pixels differ by more than 1/255")
    assert frac <= 0.005
