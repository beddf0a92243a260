"""Frozen scene definitions for the benchmark / parity configurations.

The reference defines only the first (`ch9`, src/main.rs:202-273); it has no CLI, no scene files and
no meshes-with-floor scene that it can render itself, so the others are authored here once and feed
both the device library and the CPU oracle through the same rt355_shape_desc records.  Random
placement uses SplitMix64 with seed 0x355.

Each builder returns (World, Camera) for a given resolution.
"""
import math
import os

import numpy as np

from . import _abi
from . import matrix as mx
from .group import Node, to_transformed_shapes
from .host import Camera, Material, Pattern, PointLight, Shape, World
from .wavefront import Wavefront

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
PI = math.pi


def _camera(w, h, fov, frm, to, up):
    return Camera(w, h, fov, mx.view_transformation(mx.point(*frm), mx.point(*to), mx.vector(*up)))


def orbit(camera, angle):
    """The camera turned by `angle` (radians) about the world's Y axis, everything else kept: one step of a
    camera path around a scene."""
    return Camera(camera.horizontal_size, camera.vertical_size, camera.field_of_view, camera.transform @ mx.rotation_y(-angle))


def ch9(width=200, height=150):
    """Config 1: src/main.rs:202-273 verbatim (floor plane, two spheres, one light)."""
    floor = Shape.Plane(None, Material(color=(1.0, 0.9, 0.9)))
    middle = Shape.Sphere(mx.translation(-0.5, 1.0, 0.5), Material(color=(1.0, 0.2, 0.2)))
    right = Shape.Sphere(mx.translation(1.5, 0.5, -0.5) @ mx.scaling(0.5, 0.5, 0.5), Material(color=(0.3, 0.4, 1.0)))
    world = World(PointLight((-10.0, 10.0, -10.0), (1.0, 1.0, 1.0)), [floor, middle, right])
    return world, _camera(width, height, PI / 3.0, (0.0, 1.5, -5.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0))


def ch11(width=600, height=450):
    """src/main.rs:389-535: checker 'floor' sphere, two reflective walls, reflective / transparent
    spheres.  Used as a small reflection + refraction parity scene."""
    floor = Shape.Sphere(mx.scaling(10.0, 0.01, 10.0),
                         Material(color=(1.0, 0.9, 0.9), specular=0.0, shininess=0.0, reflective=0.3,
                                  pattern=Pattern.Checker((1, 1, 1), (0, 0, 0), mx.scaling(0.5, 0.5, 0.5))))
    left_wall = Shape.Plane(mx.translation(0, 0, 8) @ mx.rotation_y(-PI / 4) @ mx.rotation_x(PI / 2),
                            Material(color=(0.8, 0.8, 0.8), reflective=0.5))
    right_wall = Shape.Plane(mx.translation(0, 0, 8) @ mx.rotation_y(PI / 4) @ mx.rotation_x(PI / 2),
                             Material(color=(0.8, 0.8, 0.8), reflective=0.5))
    middle = Shape.Sphere(mx.translation(-0.5, 1.0, 0.5), Material(color=(1.0, 0.2, 0.2), reflective=0.3))
    right_back = Shape.Sphere(mx.translation(1.5, 0.5, 1.5) @ mx.scaling(0.8, 0.8, 0.8),
                              Material(color=(0.3, 0.4, 1.0), transparency=0.8,
                                       pattern=Pattern.Stripe((0.0, 0.3, 1.0), (1, 1, 1), mx.scaling(0.7, 0.7, 0.7))))
    right_fwd = Shape.Sphere(mx.translation(1.5, 0.5, -0.5) @ mx.scaling(0.3, 0.3, 0.3),
                             Material(color=(0.1, 0.9, 0.2), reflective=0.4))
    world = World(PointLight((-10.0, 10.0, -10.0), (1, 1, 1)),
                  [floor, left_wall, right_wall, middle, right_fwd, right_back])
    return world, _camera(width, height, PI / 3.0, (0.0, 1.5, -5.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0))


_mesh_cache = {}
# True: the meshes are triangulated the way the reference's parser does it -- polygons fanned over the GLOBAL vertex
# list instead of the face's own (src/wavefront.rs:101-116, :143-155); the scenes are then the ones the reference would
# build from the same files (bench.py --ref-parser-quirk).
REF_PARSER_QUIRK = False


def _teapot(name, y_up):
    """Smooth-triangle list of a teapot OBJ.  The files are Z-up; `y_up` bakes (x,y,z) -> (x,z,-y) into
    the vertices AND the vertex normals, because the reference never transforms triangle normals
    (src/shape.rs:198-203), so a shape transformation cannot be used to re-orient a mesh."""
    key = (name, y_up, REF_PARSER_QUIRK)
    if key not in _mesh_cache:
        w = Wavefront.parse(os.path.join(ASSETS, name), ref_parser_quirk=REF_PARSER_QUIRK)
        shapes = w.shapes
        if y_up:
            rot = lambda p: (p[0], p[2], -p[1])
            for s in shapes:
                s.p1, s.p2, s.p3 = rot(s.p1), rot(s.p2), rot(s.p3)
                s.n1, s.n2, s.n3 = rot(s.n1), rot(s.n2), rot(s.n3)
        _mesh_cache[key] = shapes
    return [s.copy() for s in _mesh_cache[key]]


def teapot_low(width=1920, height=1080):
    """Config 2: teapot-low.obj (240 smooth triangles), default material, one light."""
    world = World(PointLight((-30.0, -40.0, 50.0), (1, 1, 1)), _teapot("teapot-low.obj", False))
    return world, _camera(width, height, PI / 3.0, (0.0, -45.0, 25.0), (0.0, 0.0, 7.0), (0.0, 0.0, 1.0))


def teapot_high(width=3840, height=2160):
    """Config 3: teapot-high.obj (6 320 smooth triangles), same camera and light."""
    world = World(PointLight((-30.0, -40.0, 50.0), (1, 1, 1)), _teapot("teapot-high.obj", False))
    return world, _camera(width, height, PI / 3.0, (0.0, -45.0, 25.0), (0.0, 0.0, 7.0), (0.0, 0.0, 1.0))


def north_star(width=1920, height=1080, mesh="teapot-high.obj"):
    """The headline workload: teapot-high (Y-up) on a reflective checker floor next to one glass
    sphere, shadows, reflection / refraction depth 5."""
    floor = Shape.Plane(None, Material(color=(1.0, 0.9, 0.9), specular=0.3, reflective=0.3,
                                       pattern=Pattern.Checker((1, 1, 1), (0.15, 0.15, 0.2))))
    glass = Material.new_glass()
    glass.reflective = 0.9
    glass.ambient, glass.diffuse = 0.05, 0.1
    sphere = Shape.Sphere(mx.translation(-23.0, 6.5, 10.0) @ mx.scaling(6.0, 6.0, 6.0), glass)
    shapes = _teapot(mesh, True) + [floor, sphere]
    world = World(PointLight((-30.0, 50.0, 40.0), (1, 1, 1)), shapes)
    return world, _camera(width, height, PI / 3.0, (0.0, 25.0, 45.0), (0.0, 7.0, 0.0), (0.0, 1.0, 0.0))


def teapot_field(width=1920, height=1080, mesh="teapot-high.obj", n_side=4, through_groups=False):
    """n_side x n_side copies of one teapot mesh, each under a group with a transformation of its own (placed on a grid,
    turned about Y, scaled), on a reflective floor: the reference's way of instancing a mesh -- Wavefront::parse
    puts the triangles under a group (src/wavefront.rs:41-44) and Scene::to_transformed_shapes folds the group's
    transformation into every triangle (src/group.rs:130-168, used at src/main.rs:869-881).  The World holds the
    flattened list: runs of triangles that share one transformation.  `through_groups` builds the tree with Node
    objects and flattens it with group.to_transformed_shapes (small meshes); otherwise the same list is made directly."""
    from . import group as grp
    tris = _teapot(mesh, True)
    floor = Shape.Plane(None, Material(color=(0.9, 0.9, 1.0), specular=0.3, reflective=0.25,
                                       pattern=Pattern.Checker((1, 1, 1), (0.2, 0.2, 0.25))))
    rnd = _splitmix64(3 * n_side * n_side, seed=0x517)
    shapes = [floor]
    root = grp.Node(Shape.Group())
    pitch = 52.0
    for k in range(n_side * n_side):
        gx, gz = k % n_side, k // n_side
        t = (mx.translation((gx - (n_side - 1) / 2.0) * pitch, 0.0, (gz - (n_side - 1) / 2.0) * pitch) @ mx.rotation_y(2.0 * PI * float(rnd[3 * k]))
             @ mx.scaling(*([0.7 + 0.6 * float(rnd[3 * k + 1])] * 3))).astype(np.float32)
        if through_groups:
            g = root.add(Shape.Group(t))
            for tri in tris:
                g.add(tri.copy())
        else:
            shapes.append(Shape.Group(t))
            for tri in tris:
                c = tri.copy()
                c.set_transform(t)
                shapes.append(c)
    if through_groups:
        shapes += grp.to_transformed_shapes(root)
    world = World(PointLight((-60.0, 120.0, 90.0), (1, 1, 1)), shapes)
    d = pitch * n_side
    return world, _camera(width, height, PI / 3.0, (0.15 * d, 0.55 * d, 1.0 * d), (0.0, 5.0, 0.0), (0.0, 1.0, 0.0))


def glass(width=1920, height=1080):
    """Config 4: Schlick + recursion depth 5 with checker / ring / stripe patterns.  Patterns and walls
    follow src/main.rs:389-535; the nested glass-with-air-bubble case follows the n1/n2 test
    (src/intersection.rs:370-422)."""
    floor = Shape.Plane(None, Material(color=(1.0, 0.9, 0.9), specular=0.0, shininess=0.0, reflective=0.3,
                                       pattern=Pattern.Checker((1, 1, 1), (0, 0, 0))))
    # a slab, not a rotated Plane: the reference returns a plane's object-space normal untransformed
    # (src/shape.rs:144), so a rotated plane self-shadows at random (its over_point never leaves it)
    back_wall = Shape.Cube(mx.translation(0, 0, 10.1) @ mx.scaling(30.0, 30.0, 0.1), Material(color=(0.8, 0.8, 0.8), reflective=0.5))
    g = Material.new_glass()
    g.reflective = 0.9
    g.ambient, g.diffuse = 0.02, 0.1
    outer = Shape.Sphere(mx.translation(-0.5, 1.0, 0.5), g)
    air = Material(refractive=1.0, transparency=1.0, ambient=0.0, diffuse=0.0, specular=0.0)
    inner = Shape.Sphere(mx.translation(-0.5, 1.0, 0.5) @ mx.scaling(0.5, 0.5, 0.5), air)
    ring = Shape.Sphere(mx.translation(1.7, 0.5, 2.5) @ mx.scaling(1.3, 1.3, 1.3),
                        Material(color=(0.3, 0.4, 1.0), pattern=Pattern.Ring((1, 0, 0), (1, 1, 1))))
    stripe = Shape.Sphere(mx.translation(1.5, 0.5, -0.5) @ mx.scaling(0.5, 0.5, 0.5),
                          Material(color=(0.3, 0.4, 1.0), transparency=0.8,
                                   pattern=Pattern.Stripe((0.0, 0.3, 1.0), (1, 1, 1))))
    green = Shape.Sphere(mx.translation(-2.2, 0.4, -0.8) @ mx.scaling(0.4, 0.4, 0.4),
                         Material(color=(0.1, 0.9, 0.2), reflective=0.4))
    cube = Shape.Cube(mx.translation(2.8, 0.5, 1.0) @ mx.rotation_y(0.5) @ mx.scaling(0.5, 0.5, 0.5),
                      Material(color=(0.9, 0.8, 0.1), reflective=0.2))
    world = World(PointLight((-10.0, 10.0, -10.0), (1, 1, 1)), [floor, back_wall, outer, inner, ring, stripe, green, cube])
    return world, _camera(width, height, PI / 3.0, (0.0, 1.5, -5.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0))


# --- stress scene ---------------------------------------------------------------------------------
def _splitmix64(n, seed=0x355):
    """n SplitMix64 outputs as uniform doubles in [0,1)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def shape_dtype():
    """numpy mirror of rt355_shape_desc, for building very large scenes without Python objects."""
    mat = np.dtype([("color", "<f4", 3), ("ambient", "<f4"), ("diffuse", "<f4"), ("specular", "<f4"),
                    ("shininess", "<f4"), ("reflective", "<f4"), ("refractive", "<f4"), ("transparency", "<f4"),
                    ("pattern_kind", "<i4"), ("c1", "<f4", 3), ("c2", "<f4", 3)])
    dt = np.dtype([("kind", "<i4"), ("m", "<f4", 16), ("mat", mat), ("min", "<f4"), ("max", "<f4"), ("closed", "<i4"),
                   ("p1", "<f4", 3), ("p2", "<f4", 3), ("p3", "<f4", 3), ("n1", "<f4", 3), ("n2", "<f4", 3), ("n3", "<f4", 3)])
    import ctypes
    assert dt.itemsize == ctypes.sizeof(_abi.ShapeDesc), (dt.itemsize, ctypes.sizeof(_abi.ShapeDesc))
    return dt


class BulkWorld(World):
    """A World whose shapes live in one numpy record array (rt355_shape_desc layout)."""

    def __init__(self, pointlight, records):
        super().__init__(pointlight, [])
        self.records = np.ascontiguousarray(records)

    def n_shapes(self):
        return int(self.records.shape[0])

    def descs(self):
        import ctypes
        return (_abi.ShapeDesc * self.n_shapes()).from_buffer(self.records)


def stress(width=7680, height=4320, n_top=10, n_mid=100, n_leaf=100, extent=400.0):
    """Config 5: n_top*n_mid*n_leaf instances (25 % each cube / closed cylinder 0..1 / closed cone -1..0 /
    sphere) in a 3-level group tree, flattened with the reference's one-level composition
    (src/group.rs:130-168): a leaf gets mid_group.transformation * leaf.transformation and the top
    groups' transformations do not accumulate.  Group entries stay in the list as inert shapes.
    Positions are uniform in an extent x 40 x extent slab (400 by default), scale 0.3-0.8, random Y
    rotation, 10 % reflective."""
    n_groups_mid = n_top * n_mid
    n = n_groups_mid * n_leaf
    u = _splitmix64(n * 8 + n_groups_mid * 2)
    ul = u[: n * 8].reshape(n, 8)
    ug = u[n * 8:].reshape(n_groups_mid, 2)
    # mid groups on a grid covering x,z in [-extent/2, extent/2)
    half = extent / 2.0
    gx_n = int(math.ceil(math.sqrt(n_groups_mid)))
    gz_n = int(math.ceil(n_groups_mid / gx_n))
    cell_x, cell_z = extent / gx_n, extent / gz_n
    gid = np.arange(n_groups_mid)
    g_tx = -half + (gid % gx_n) * cell_x + 0.1 * ug[:, 0]
    g_tz = -half + (gid // gx_n) * cell_z + 0.1 * ug[:, 1]

    total = 1 + n_top + n_groups_mid + n + 1  # root, tops, mids, leaves, floor
    rec = np.zeros(total, dtype=shape_dtype())
    ident = np.eye(4, dtype=np.float32).T.reshape(16)
    rec["m"][:] = ident
    rec["mat"]["color"][:] = 1.0
    rec["mat"]["ambient"], rec["mat"]["diffuse"], rec["mat"]["specular"] = 0.1, 0.9, 0.9
    rec["mat"]["shininess"], rec["mat"]["refractive"] = 200.0, 1.0
    rec["min"], rec["max"] = -np.inf, np.inf

    # depth-first order: root, then per top group: top, then per mid: mid, leaves
    pos = 0
    rec["kind"][pos] = _abi.GROUP
    pos += 1
    leaf_slots = np.empty(n, dtype=np.int64)
    mid_slots = np.empty(n_groups_mid, dtype=np.int64)
    for t in range(n_top):
        rec["kind"][pos] = _abi.GROUP
        # a top-level transformation that must NOT reach the leaves (one-level composition)
        rec["m"][pos] = mx.to_abi(mx.translation(0.0, 1000.0 * (t + 1), 0.0))
        pos += 1
        for m in range(n_mid):
            g = t * n_mid + m
            mid_slots[g] = pos
            pos += 1
            leaf_slots[g * n_leaf:(g + 1) * n_leaf] = np.arange(pos, pos + n_leaf)
            pos += n_leaf
    floor_slot = pos
    rec["kind"][mid_slots] = _abi.GROUP
    # the root's child transformation rule applies to tops (root is identity) and mids get top*mid, but a
    # mid's OWN transformation (re-read by the recursion) is what composes into the leaves.
    mid_m = np.tile(np.eye(4, dtype=np.float64), (n_groups_mid, 1, 1))
    mid_m[:, 0, 3], mid_m[:, 2, 3] = g_tx, g_tz
    top_of_mid = np.repeat(np.arange(n_top), n_mid)
    mid_in_list = mid_m.copy()
    mid_in_list[:, 1, 3] += 1000.0 * (top_of_mid + 1)  # top * mid, as stored in the flattened list
    rec["m"][mid_slots] = mid_in_list.astype(np.float32).transpose(0, 2, 1).reshape(-1, 16)

    kind_cycle = np.array([_abi.CUBE, _abi.CYLINDER, _abi.CONE, _abi.SPHERE])
    kinds = kind_cycle[np.arange(n) % 4]
    s = 0.3 + 0.5 * ul[:, 0]
    ang = 2 * PI * ul[:, 1]
    lx, ly, lz = cell_x * ul[:, 2], 1.0 + min(38.0, extent / 10.0 - 2.0) * ul[:, 3], cell_z * ul[:, 4]
    c, sn = np.cos(ang), np.sin(ang)
    leaf = np.tile(np.eye(4, dtype=np.float64), (n, 1, 1))
    # translation * rotation_y * scaling
    leaf[:, 0, 0], leaf[:, 0, 2] = c * s, sn * s
    leaf[:, 1, 1] = s
    leaf[:, 2, 0], leaf[:, 2, 2] = -sn * s, c * s
    leaf[:, 0, 3], leaf[:, 1, 3], leaf[:, 2, 3] = lx, ly, lz
    leaf32 = leaf.astype(np.float32)
    mid32 = mid_m.astype(np.float32)
    comp = np.einsum("nij,njk->nik", np.repeat(mid32, n_leaf, axis=0), leaf32).astype(np.float32)
    rec["m"][leaf_slots] = comp.transpose(0, 2, 1).reshape(-1, 16)
    rec["kind"][leaf_slots] = kinds
    is_cyl, is_cone = kinds == _abi.CYLINDER, kinds == _abi.CONE
    rec["min"][leaf_slots[is_cyl]], rec["max"][leaf_slots[is_cyl]] = 0.0, 1.0
    rec["min"][leaf_slots[is_cone]], rec["max"][leaf_slots[is_cone]] = -1.0, 0.0
    rec["closed"][leaf_slots[is_cyl | is_cone]] = 1
    col = np.stack([0.25 + 0.75 * ul[:, 5], 0.25 + 0.75 * ul[:, 6], 0.25 + 0.75 * ul[:, 7]], axis=1)
    rec["mat"]["color"][leaf_slots] = col.astype(np.float32)
    refl = (np.arange(n) % 10) == 0
    rec["mat"]["reflective"][leaf_slots[refl]] = 0.4

    rec["kind"][floor_slot] = _abi.PLANE
    rec["mat"]["color"][floor_slot] = (0.9, 0.9, 0.9)
    rec["mat"]["specular"][floor_slot] = 0.1
    k = extent / 400.0
    world = BulkWorld(PointLight((-150.0 * k, 400.0 * k, -300.0 * k), (1, 1, 1)), rec)
    return world, _camera(width, height, PI / 3.0, (0.0, 140.0 * k, -420.0 * k), (0.0, 10.0 * k, 0.0), (0.0, 1.0, 0.0))


def stress_tree(n_top=2, n_mid=3, n_leaf=4):
    """The same tree shape as `stress`, small, built from Node objects: used by the tests to check the
    vectorised flattening above against group.to_transformed_shapes."""
    root = Node(Shape.Group())
    k = 0
    for t in range(n_top):
        top = root.add(Shape.Group(mx.translation(0.0, 1000.0 * (t + 1), 0.0)))
        for m in range(n_mid):
            mid = top.add(Shape.Group(mx.translation(float(k), 0.0, float(2 * k))))
            k += 1
            for l in range(n_leaf):
                mid.add(Shape.Sphere(mx.translation(float(l), 1.0, 0.0)))
    return to_transformed_shapes(root)


CONFIGS = {
    "ch9": ch9,
    "ch11": ch11,
    "teapot_low": teapot_low,
    "teapot_high": teapot_high,
    "glass": glass,
    "stress": stress,
    "north_star": north_star,
    "teapot_field": teapot_field,
}
