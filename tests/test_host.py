"""CPU-only checks of everything around the kernels: the C-ABI library loads and exports what the header
declares, the ctypes mirrors have the C layout, and the host-side mirror of the reference API (matrices,
OBJ loader, group flattening, canvas / PPM, tile split, multi-rank gather over gloo) behaves like the
reference's own unit tests say (cited per test)."""
import ctypes as C
import math
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raytracer_amd import _abi, group, scenes  # noqa: E402
from raytracer_amd import dist as rd  # noqa: E402
from raytracer_amd import matrix as mx  # noqa: E402
from raytracer_amd.host import Camera, Canvas, Color, Material, Pattern, Shape, World  # noqa: E402
from raytracer_amd.wavefront import Wavefront  # noqa: E402

HEADER = os.path.join(ROOT, "include", "rt355.h")
OBJ = os.path.join(ROOT, "tests", "golden", "obj")
EPS = 1e-3  # src/epsilon.rs


# ------------------------------------------------------------------------------------------------
# C ABI
# ------------------------------------------------------------------------------------------------
def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return sorted(set(re.findall(r"\b(rt355_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 20
    assert sorted(n for n, _, _ in _abi.SYMBOLS) == names
    lib = _abi.lib()  # raises when the HIP library has not been built
    for n in names:
        assert hasattr(lib, n), n
    assert lib.rt355_version() >= 1


def test_ctypes_structs_match_the_c_layout():
    prog = r"""
#include <stdio.h>
#include <stddef.h>
#include "rt355.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu ", sizeof(rt355_material), sizeof(rt355_shape_desc), sizeof(rt355_light_desc),
         sizeof(rt355_camera_desc), sizeof(rt355_stats));
  printf("%zu %zu %zu %zu %zu\n", offsetof(rt355_shape_desc, mat), offsetof(rt355_shape_desc, p1),
         offsetof(rt355_shape_desc, n3), offsetof(rt355_stats, frame_ms), offsetof(rt355_stats, launch_flops));
  return 0;
}
"""
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "l.c"), os.path.join(td, "l")
        open(src, "w").write(prog)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        got = [int(x) for x in subprocess.check_output([exe]).split()]
    want = [C.sizeof(_abi.MaterialDesc), C.sizeof(_abi.ShapeDesc), C.sizeof(_abi.LightDesc), C.sizeof(_abi.CameraDesc),
            C.sizeof(_abi.Stats), _abi.ShapeDesc.mat.offset, _abi.ShapeDesc.p1.offset, _abi.ShapeDesc.n3.offset,
            _abi.Stats.frame_ms.offset, _abi.Stats.launch_flops.offset]
    assert got == want


def _c_decls():
    """(functions {name: n_args}, structs {name: [field names]}) of include/rt355.h."""
    import re
    src = open(os.path.join(ROOT, "include", "rt355.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    funcs = {}
    for m in re.finditer(r"\b(rt355_\w+)\s*\(([^;{]*?)\)\s*;", src):
        args = m.group(2).strip()
        funcs[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    structs = {}
    for m in re.finditer(r"typedef struct (rt355_\w+) \{(.*?)\} \1;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                name = re.sub(r"\[.*?\]", "", part).strip().split()[-1].lstrip("*")
                fields.append(name)
        structs[m.group(1)] = fields
    return funcs, structs


def test_rust_sys_crate_matches_the_header():
    """rust/rt355-sys/src/lib.rs (what the reference's crate would link against; it cannot be compiled here, there
    is no Rust toolchain) declares exactly the functions of include/rt355.h with the same number of arguments,
    and its #[repr(C)] structs list the header's fields in the header's order."""
    import re
    rs = open(os.path.join(ROOT, "rust", "rt355-sys", "src", "lib.rs")).read()
    rs = re.sub(r"//[^\n]*", "", rs)
    rs_funcs = {}
    for m in re.finditer(r"pub fn (rt355_\w+)\s*\((.*?)\)\s*(?:->[^;]*)?;", rs, flags=re.S):
        args = m.group(2).strip()
        rs_funcs[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    rs_structs = {}
    for m in re.finditer(r"pub struct (rt355_\w+)\s*\{(.*?)\}", rs, flags=re.S):
        rs_structs[m.group(1)] = re.findall(r"pub (\w+)\s*:", m.group(2))
    c_funcs, c_structs = _c_decls()
    assert set(rs_funcs) == set(c_funcs), (sorted(set(c_funcs) - set(rs_funcs)), sorted(set(rs_funcs) - set(c_funcs)))
    assert rs_funcs == c_funcs
    assert set(rs_structs) == set(c_structs)
    for name, fields in c_structs.items():
        assert rs_structs[name] == fields, name
    # the ctypes view is held to the same header
    assert {n for n, _, _ in _abi.SYMBOLS} == set(c_funcs)
    for n, _, args in _abi.SYMBOLS:
        assert len(args) == c_funcs[n], n
    # constants
    for name, value in (("RT355_MAX_GEN", 8), ("RT355_MAX_LAUNCH", 48), ("RT355_ECAPACITY", -3)):
        assert re.search(r"pub const %s: \w+ = %d;" % (name, value), rs), name
    # every file the integration notes point at exists
    for rel in ("rust/rt355-sys/Cargo.toml", "rust/rt355-sys/build.rs", "rust/raytracer-gpu/src/gpu.rs",
                "rust/raytracer-gpu/src/accessors.rs", "rust/raytracer-gpu/tests/gpu_camera.rs"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel


def test_header_is_plain_c():
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "h.c")
        open(src, "w").write('#include "rt355.h"\nint main(void){return RT355_OK;}\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                               "-c", src, "-o", os.path.join(td, "h.o")])


def test_bulk_dtype_matches_shape_desc():
    dt = scenes.shape_dtype()
    assert dt.itemsize == C.sizeof(_abi.ShapeDesc)
    for name in ("kind", "m", "min", "max", "closed", "p1", "n3"):
        assert dt.fields[name][1] == getattr(_abi.ShapeDesc, name).offset, name


def test_flop_table_is_the_documented_one():
    out = (C.c_double * 16)()
    _abi.lib().rt355_flop_table(out)
    t = list(out)
    # primary ray, slab test per box, ray -> object space, then per kind in rt355_kind order
    assert t[:3] == [40.0, 24.0, 33.0]
    assert t[3:11] == [28.0, 4.0, 20.0, 50.0, 62.0, 0.0, 52.0, 52.0]


def test_context_creation_without_a_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _abi.lib()
    h = C.c_void_p()
    rc = lib.rt355_ctx_create(C.byref(h), 0, 64)
    assert rc != 0 and not h.value
    assert lib.rt355_last_error()
    with pytest.raises(_abi.Rt355Error):
        _abi.check(rc)
    assert lib.rt355_ctx_create(C.byref(h), 0, 48) == -1  # RT355_EINVAL: real_bits is 32 or 64


# ------------------------------------------------------------------------------------------------
# Matrices (src/matrix.rs:15-89, src/tuple.rs:38-60)
# ------------------------------------------------------------------------------------------------
def test_reflect():  # src/tuple.rs:73-88
    r = mx.reflect(mx.vector(1, -1, 0), mx.vector(0, 1, 0))
    np.testing.assert_allclose(r, mx.vector(1, 1, 0), atol=1e-6)
    s = math.sqrt(2) / 2
    r = mx.reflect(mx.vector(0, -1, 0), mx.vector(s, s, 0))
    np.testing.assert_allclose(r, mx.vector(1, 0, 0), atol=1e-6)


def test_view_transformation():  # src/tuple.rs:91-137
    up = mx.vector(0, 1, 0)
    np.testing.assert_allclose(mx.view_transformation(mx.point(0, 0, 0), mx.point(0, 0, -1), up), mx.identity(), atol=1e-6)
    np.testing.assert_allclose(mx.view_transformation(mx.point(0, 0, 0), mx.point(0, 0, 1), up), mx.scaling(-1, 1, -1), atol=1e-6)
    np.testing.assert_allclose(mx.view_transformation(mx.point(0, 0, 8), mx.point(0, 0, 0), up), mx.translation(0, 0, -8), atol=1e-6)
    got = mx.view_transformation(mx.point(1, 3, 2), mx.point(4, -2, 8), mx.vector(1, 1, 0))
    want = np.array([[-0.50709, 0.50709, 0.67612, -2.36643], [0.76772, 0.60609, 0.12122, -2.82843],
                     [-0.35857, 0.59761, -0.71714, 0.0], [0, 0, 0, 1]])
    np.testing.assert_allclose(got, want, atol=1e-3)


def test_transform_builders():
    p = mx.point(0, 1, 0)
    np.testing.assert_allclose(mx.rotation_x(math.pi / 2) @ p, mx.point(0, 0, 1), atol=1e-6)
    np.testing.assert_allclose(mx.rotation_y(math.pi / 2) @ mx.point(0, 0, 1), mx.point(1, 0, 0), atol=1e-6)
    np.testing.assert_allclose(mx.rotation_z(math.pi / 2) @ p, mx.point(-1, 0, 0), atol=1e-6)
    np.testing.assert_allclose(mx.translation(5, -3, 2) @ mx.point(-3, 4, 5), mx.point(2, 1, 7), atol=1e-6)
    np.testing.assert_allclose(mx.translation(5, -3, 2) @ mx.vector(-3, 4, 5), mx.vector(-3, 4, 5), atol=1e-6)
    np.testing.assert_allclose(mx.scaling(2, 3, 4) @ mx.point(-4, 6, 8), mx.point(-8, 18, 32), atol=1e-6)
    m = mx.translation(1, 2, 3)
    abi = mx.to_abi(m)  # column-major, as nalgebra stores
    assert list(abi[12:15]) == [1.0, 2.0, 3.0] and abi[15] == 1.0


def test_camera_derived_fields():  # src/camera.rs:116-135
    c = Camera(200, 125, math.pi / 2)
    assert abs(float(c.pixel_size) - 0.01) < 1e-6
    c = Camera(125, 200, math.pi / 2)
    assert abs(float(c.pixel_size) - 0.01) < 1e-6
    d = c.to_desc()
    assert (d.hsize, d.vsize) == (125.0, 200.0)


# ------------------------------------------------------------------------------------------------
# OBJ loader (src/wavefront.rs:39-184; tests :192-305; fixtures are the reference's own small files)
# ------------------------------------------------------------------------------------------------
def _p(shape, k):
    return tuple(np.asarray(getattr(shape, k), dtype=np.float32)[:3].tolist())


def test_parse_vertex():  # src/wavefront.rs:192-201
    w = Wavefront.parse(os.path.join(OBJ, "vertex.obj"))
    assert w.vertices[1:5] == [(-1.0, 1.0, 0.0), (-1.0, 0.5, 0.0), (1.0, 0.0, 0.0), (1.0, 1.0, 0.0)]


def test_parse_face():  # src/wavefront.rs:203-228
    w = Wavefront.parse(os.path.join(OBJ, "face.obj"))
    s1, s2 = w.shapes[:2]
    assert s1.kind == _abi.SMOOTH_TRIANGLE
    assert (_p(s1, "p1"), _p(s1, "p2"), _p(s1, "p3")) == (w.vertices[1], w.vertices[2], w.vertices[3])
    assert (_p(s2, "p1"), _p(s2, "p2"), _p(s2, "p3")) == (w.vertices[1], w.vertices[3], w.vertices[4])


@pytest.mark.parametrize("quirk", [False, True])
def test_parse_polygon(quirk):  # src/wavefront.rs:230-262 (the face is `1 2 3 4 5`, where both fans coincide)
    w = Wavefront.parse(os.path.join(OBJ, "polygones.obj"), ref_parser_quirk=quirk)
    assert len(w.shapes) == 3
    for s, (a, b, c) in zip(w.shapes, [(1, 2, 3), (1, 3, 4), (1, 4, 5)]):
        assert (_p(s, "p1"), _p(s, "p2"), _p(s, "p3")) == (w.vertices[a], w.vertices[b], w.vertices[c])


def test_parse_vertex_normal():  # src/wavefront.rs:264-273
    w = Wavefront.parse(os.path.join(OBJ, "vertex_normal.obj"))
    f = lambda *v: tuple(float(np.float32(x)) for x in v)  # noqa: E731
    assert w.normals[1:4] == [f(0, 0, 1), f(0.707, 0, -0.707), f(1, 2, 3)]


def test_parse_face_with_normal():  # src/wavefront.rs:275-305
    w = Wavefront.parse(os.path.join(OBJ, "face_with_normal.obj"))
    assert len(w.shapes) == 2
    for s in w.shapes:
        assert (_p(s, "p1"), _p(s, "p2"), _p(s, "p3")) == (w.vertices[1], w.vertices[2], w.vertices[3])
        assert (_p(s, "n1"), _p(s, "n2"), _p(s, "n3")) == (w.normals[3], w.normals[1], w.normals[2])


def test_fan_over_the_face_not_the_file():
    """A quad that is not `1 2 3 4`: the proper fan uses the face's own vertices, the reference's fan
    (src/wavefront.rs:101-116) uses file-global indices 1, k, k+1."""
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "q.obj")
        open(path, "w").write("v 0 0 0\nv 9 9 9\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 3 4 5\n")
        good = Wavefront.parse(path)
        quirk = Wavefront.parse(path, ref_parser_quirk=True)
    assert [(_p(s, "p1"), _p(s, "p2"), _p(s, "p3")) for s in good.shapes] == [
        ((0, 0, 0), (1, 0, 0), (1, 1, 0)), ((0, 0, 0), (1, 1, 0), (0, 1, 0))]
    assert [(_p(s, "p1"), _p(s, "p2"), _p(s, "p3")) for s in quirk.shapes] == [
        ((0, 0, 0), (9, 9, 9), (1, 0, 0)), ((0, 0, 0), (1, 0, 0), (1, 1, 0))]


def test_face_index_forms():
    """`v//vn` and `v/vt/vn` carry a normal index; `v/vt` does not (the reference panics on it,
    src/wavefront.rs:93) and is read as a vertex without a normal."""
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "f.obj")
        open(path, "w").write("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\nf 1/7/1 2/8/1 3/9/1\nf 1/7 2/8 3/9\n")
        w = Wavefront.parse(path)
    assert len(w.shapes) == 3
    assert [_p(w.shapes[k], "n1") for k in range(3)] == [(0, 0, 1), (0, 0, 1), (0, 0, 0)]
    assert all(_p(s, "p2") == (1, 0, 0) for s in w.shapes)


def test_teapots_have_the_expected_triangle_counts():
    assets = os.path.join(ROOT, "raytracer_amd", "assets")
    assert len(Wavefront.parse(os.path.join(assets, "teapot-low.obj")).shapes) == 240
    assert len(Wavefront.parse(os.path.join(assets, "teapot-high.obj")).shapes) == 6320


# ------------------------------------------------------------------------------------------------
# Group flattening (src/group.rs:117-168)
# ------------------------------------------------------------------------------------------------
def _tree():
    g1 = group.Node(Shape.Group(mx.rotation_y(math.pi / 2)))
    g2 = g1.add(Shape.Group(mx.scaling(2, 2, 2)))
    s = g2.add(Shape.Sphere(mx.translation(5, 0, 0)))
    return g1, g2, s


def test_flatten_composes_only_the_immediate_parent():
    g1, g2, s = _tree()
    flat = group.to_transformed_shapes(g1)
    assert [x.kind for x in flat] == [_abi.GROUP, _abi.GROUP, _abi.SPHERE]
    np.testing.assert_allclose(flat[1].transformation, mx.rotation_y(math.pi / 2) @ mx.scaling(2, 2, 2), atol=1e-6)
    # the sphere sees g2's ORIGINAL transform only: outer groups do not accumulate (src/group.rs:146)
    np.testing.assert_allclose(flat[2].transformation, mx.scaling(2, 2, 2) @ mx.translation(5, 0, 0), atol=1e-6)
    # inputs are not modified
    np.testing.assert_allclose(s.shape.transformation, mx.translation(5, 0, 0))


def test_flatten_with_accumulation():
    g1, g2, s = _tree()
    flat = group.to_transformed_shapes(g1, accumulate=True)
    want = mx.rotation_y(math.pi / 2) @ mx.scaling(2, 2, 2) @ mx.translation(5, 0, 0)
    np.testing.assert_allclose(flat[2].transformation, want, atol=1e-5)
    # world -> object of src/group.rs:299-316 through the accumulated transform
    p = np.linalg.inv(flat[2].transformation.astype(np.float64)) @ mx.point(-2, 0, -10)
    np.testing.assert_allclose(p, mx.point(0, 0, -1), atol=1e-3)


def test_converting_point_from_world_to_object_space():  # src/group.rs:299-316
    g1, g2, s = _tree()
    p = group.world_to_object(s, mx.point(-2, 0, -10))
    np.testing.assert_allclose(p, mx.point(0, 0, -1), atol=EPS)
    assert p.dtype == np.float32


def _tree_123():
    g1 = group.Node(Shape.Group(mx.rotation_y(math.pi / 2)))
    g2 = g1.add(Shape.Group(mx.scaling(1, 2, 3)))
    s = g2.add(Shape.Sphere(mx.translation(5, 0, 0)))
    return s


def test_converting_normal_from_object_to_world_space():  # src/group.rs:318-346 (the reference passes a POINT: w = 1)
    r = math.sqrt(3) / 3
    n = group.normal_to_world(_tree_123(), mx.point(r, r, r))
    np.testing.assert_allclose(n, mx.vector(0.2857, 0.4286, -0.8571), atol=10 * EPS)


def test_finding_normal_on_child():  # src/group.rs:348-370
    n = group.normal(_tree_123(), mx.point(1.7321, 1.1547, -5.5774))
    np.testing.assert_allclose(n, mx.vector(0.2857, 0.4286, -0.8571), atol=10 * EPS)


def test_children_of_a_non_group_pass_through():  # src/group.rs:154-166
    root = group.Node(Shape.Sphere(mx.scaling(3, 3, 3)))
    root.add(Shape.Cube(mx.translation(1, 0, 0)))
    flat = group.to_transformed_shapes(root)
    assert [x.kind for x in flat] == [_abi.SPHERE, _abi.CUBE]
    np.testing.assert_allclose(flat[1].transformation, mx.translation(1, 0, 0))


def test_bulk_stress_scene_has_the_layout_of_the_tree_flattening():
    """scenes.stress writes its 3-level tree directly as a numpy table.  The same tree shape through Node +
    to_transformed_shapes must give the same sequence of group / leaf entries, and both must show the
    one-level composition: mid groups carry top * mid in the list, leaves only mid * leaf."""
    kw = dict(n_top=2, n_mid=3, n_leaf=4)
    bulk, _ = scenes.stress(64, 48, extent=40.0, **kw)
    flat = scenes.stress_tree(**kw)
    a = np.frombuffer(bytes(bulk.descs()), dtype=scenes.shape_dtype())
    assert len(a) == bulk.n_shapes() == len(flat) + 1  # + floor
    assert a["kind"][-1] == _abi.PLANE
    is_group_bulk = a["kind"][:-1] == _abi.GROUP
    is_group_tree = np.array([s.kind == _abi.GROUP for s in flat])
    assert (is_group_bulk == is_group_tree).all()
    ty_bulk = a["m"][:-1, 13]  # column-major: element (1,3), the y translation
    ty_tree = np.array([s.transformation[1, 3] for s in flat])
    for ty in (ty_bulk, ty_tree):
        assert (ty[~is_group_tree] < 100.0).all()  # a top group's +1000*k never reaches a leaf
        assert (ty[is_group_tree][1:] >= 1000.0).all()  # tops, and mids as top * mid
    kinds = a["kind"][:-1][~is_group_bulk]
    assert set(kinds.tolist()) == {_abi.CUBE, _abi.CYLINDER, _abi.CONE, _abi.SPHERE}
    cyl = a[:-1][a["kind"][:-1] == _abi.CYLINDER]
    assert (cyl["min"] == 0).all() and (cyl["max"] == 1).all() and (cyl["closed"] == 1).all()


# ------------------------------------------------------------------------------------------------
# Colour, canvas, PPM (src/color.rs:97-111, src/canvas.rs, src/ppm.rs:30-59)
# ------------------------------------------------------------------------------------------------
def test_to_u8_rounding():
    got = Color.to_u8([[0.0, 0.5, 1.0], [1.5, -0.2, float("nan")], [0.002, 0.0019, 0.998]])
    assert got.tolist() == [[0, 128, 255], [255, 0, 0], [1, 0, 254]]


def test_oracle_to_u8_agrees_with_host(oracle):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-0.5, 1.5, 4000), np.arange(0, 256) / 255.0, (np.arange(0, 256) + 0.5) / 255.0]).astype(np.float32)
    assert (oracle.to_u8(x) == Color.to_u8(x)).all()


def test_ppm_layout():  # src/ppm.rs:81-103 and the layout of :30-59
    c1, c2 = (1.0, 0.6, 0.4), (0.1, 0.2, 0.8)
    cv = Canvas(2, 3, [c1, c2, c1, c2, c1, c2])
    text = cv.to_ppm()
    lines = text.split("\n")
    assert lines[:3] == ["P3", "2 3", "255"]
    assert lines[3] == "255 153 102 26 51 204 255 153 102 26 51 204 "
    assert lines[4] == "255 153 102 26 51 204 "
    assert lines[5] == "" and len(lines) == 6  # len/4 + 1 body lines, each newline-terminated
    p6 = cv.to_ppm_p6()
    assert p6.startswith(b"P6\n2 3\n255\n") and len(p6) == 11 + 18
    assert tuple(cv.pixel_at(1, 2)) == tuple(np.float32(c2))


def test_materials_and_patterns_fill_the_abi_record():
    m = Material.new_glass()
    d = m.to_desc()
    assert (d.transparency, d.refractive) == (1.0, 1.5)
    m = Material.new_default()
    m.pattern = Pattern.Checker((1, 1, 1), (0, 0, 0))
    d = m.to_desc()
    assert d.pattern_kind == _abi.PATTERN_CHECKER and list(d.c1) == [1.0, 1.0, 1.0]
    d = Material.new_default().to_desc()  # src/material.rs:31-72
    assert [round(x, 6) for x in (d.ambient, d.diffuse, d.specular, d.shininess, d.reflective, d.transparency, d.refractive)] == [
        0.1, 0.9, 0.9, 200.0, 0.0, 0.0, 1.0]
    w = World.new_default()
    assert w.n_shapes() == 2 and C.sizeof(w.descs()) == 2 * C.sizeof(_abi.ShapeDesc)


# ------------------------------------------------------------------------------------------------
# Tile split (one process per GPU; tile t -> rank t % N)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("size", [(64, 64), (70, 45), (1, 1), (33, 32), (1920, 1080)])
@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_tiles_cover_every_pixel_exactly_once(size, n):
    w, h = size
    seen = np.zeros(w * h, dtype=np.int32)
    for r in range(n):
        idx = rd.tile_pixels(w, h, r, n)
        assert idx.shape == (rd.tiles_per_rank(w, h, n), 32, 32)
        np.add.at(seen, idx[idx >= 0], 1)
    assert (seen == 1).all()


def test_tiles_per_rank_agrees_with_the_library():
    lib = _abi.lib()
    for (w, h) in [(64, 64), (70, 45), (1920, 1080), (7680, 4320)]:
        cd = Camera(w, h, 1.0).to_desc()
        for n in (1, 2, 3, 8):
            assert lib.rt355_tiles_per_rank(C.byref(cd), n) == rd.tiles_per_rank(w, h, n)


def test_untile_host_round_trip():
    w, h, n = 70, 45, 3
    rng = np.random.default_rng(0)
    canvas = rng.random((h, w, 3), dtype=np.float32)
    bufs = []
    for r in range(n):
        idx = rd.tile_pixels(w, h, r, n)
        b = np.zeros(idx.shape + (3,), dtype=np.float32)
        b[idx >= 0] = canvas.reshape(-1, 3)[idx[idx >= 0]]
        bufs.append(b.reshape(-1))
    got = rd.untile_host(np.stack(bufs), w, h, n)
    assert (got == canvas).all()


def _gather_worker(rank, world_size, port, w, h, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        # every rank "renders" its own tiles: pixel value = its row-major index (+ channel / 4)
        idx = rd.tile_pixels(w, h, rank, world_size)
        local = np.zeros(idx.shape + (3,), dtype=np.float32)
        for c in range(3):
            local[..., c] = np.where(idx >= 0, idx + c / 4.0, -7.0)
        out = rd.gather_canvas(torch.from_numpy(local.reshape(-1)), w, h, rank, world_size)
        if rank == 0:
            q.put(out.numpy())
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size", [2, 3])
def test_gather_canvas_over_gloo(world_size):
    """The N>1 path on CPU: one process per rank, tile buffers gathered to rank 0 and un-tiled."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    w, h = 100, 70
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world_size, port, w, h, q)) for r in range(world_size)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = np.arange(w * h, dtype=np.float32).reshape(h, w, 1) + np.array([0, 0.25, 0.5], dtype=np.float32)
    assert got.shape == (h, w, 3) and (got == want).all()


def _frame_loop_worker(rank, world_size, port, w, h, as_u8, in_flight, batch, n_frames, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        idx = rd.tile_pixels(w, h, rank, world_size)

        def fake_render(frame_no, out_tensor, stats):
            # frame f: pixel value = (row-major index mod 251 + f) / 255 in every channel, padding slots poisoned
            v = np.where(idx >= 0, ((idx % 251) + frame_no) / 255.0, -7.0).astype(np.float32)
            out_tensor.copy_(torch.from_numpy(np.repeat(v[..., None], 3, axis=-1).reshape(-1)))

        loop = rd.FrameLoop(None, None, w, h, rank, world_size, tensor_device="cpu", render=fake_render, as_u8=as_u8, in_flight=in_flight, batch=batch)
        frames = []
        for f in range(n_frames):  # different frames back to back: every slot's tile buffer gets reused
            out = loop.step(f)
            if rank == 0 and out is not None:
                frames.append(out.numpy().copy())
            elif rank != 0:
                assert out is None
            if in_flight > 1 or batch > 1:  # the first canvas comes back when the first chain's slot is taken again
                assert (out is None) == (f < (in_flight + 1) * batch - 1 or rank != 0)
        rest = loop.flush()
        if rank == 0:
            frames += [c.numpy() for c in rest]
            q.put(frames)
        else:
            assert rest == []
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("as_u8,in_flight,batch", [(False, 1, 1), (True, 1, 1), (False, 3, 1), (True, 2, 1), (False, 1, 2), (False, 2, 3), (True, 3, 2)])
def test_frame_loop_over_gloo(as_u8, in_flight, batch):
    """The loop bench.py times for N > 1 (FrameLoop.step: render tiles, gather, un-tile), on CPU tensors over
    gloo with a stand-in renderer: consecutive DIFFERENT frames must each arrive intact on rank 0, as float
    tiles and as tiles taken through Color::to_u8 before the gather -- one frame per chain or `batch` frames per
    chain (the last chain then holds fewer), one chain or several in flight."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    w, h, world_size = 100, 70, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_frames = 5 if batch == 1 else 11
    procs = [ctx.Process(target=_frame_loop_worker, args=(r, world_size, port, w, h, as_u8, in_flight, batch, n_frames, q)) for r in range(world_size)]
    for p in procs:
        p.start()
    frames = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    base = (np.arange(w * h) % 251).reshape(h, w, 1)
    assert len(frames) == n_frames  # every frame arrives, in order, whatever the pipeline depth
    for f, got in enumerate(frames):
        if as_u8:
            want = np.repeat(np.clip(base + f, 0, 255), 3, axis=2).astype(np.uint8)
            assert got.dtype == np.uint8
        else:
            want = np.repeat(((base + f) / 255.0).astype(np.float32), 3, axis=2)
        assert got.shape == (h, w, 3) and (got == want).all(), f


def test_tile_render_of_the_oracle_assembles_to_the_full_frame(oracle):
    """Tile ownership stated with the oracle: rendering each rank's pixels separately (here: windows of
    the CPU oracle) and un-tiling gives the frame rendered in one piece."""
    world, cam = scenes.ch9(48, 40)
    full, _, _ = oracle.render(world, cam, 5, 64)
    full = np.asarray(full, dtype=np.float32).reshape(40, 48, 3)
    n = 2
    bufs = []
    for r in range(n):
        idx = rd.tile_pixels(48, 40, r, n)
        b = np.zeros(idx.shape + (3,), dtype=np.float32)
        b[idx >= 0] = full.reshape(-1, 3)[idx[idx >= 0]]
        bufs.append(b.reshape(-1))
    assert (rd.untile_host(np.stack(bufs), 48, 40, n) == full).all()


# ------------------------------------------------------------------------------------------------
# Scene tables + BVH built on the host (no GPU): rt355_scene_check
# ------------------------------------------------------------------------------------------------
def _check(world, real_bits=64, exact_cones=0):
    out = (C.c_uint64 * 5)()
    rc = _abi.lib().rt355_scene_check(world.descs(), world.n_shapes(), real_bits, exact_cones, out)
    return rc, dict(nodes=out[0], bounded=out[1], unbounded=out[2], stack=out[3], materials=out[4])


@pytest.mark.parametrize("real_bits", [32, 64])
def test_bvh_structure_teapot(real_bits):
    world, _ = scenes.north_star(64, 36)
    rc, info = _check(world, real_bits)
    assert rc == 0, _abi.lib().rt355_last_error()
    assert info["bounded"] == 6321 and info["unbounded"] == 1  # teapot + sphere in the BVH, the floor plane beside it
    assert 6321 / 4 / 3 <= info["nodes"] <= 6321  # 4-wide nodes over leaves of up to 4 triangles
    assert info["stack"] <= 62
    assert info["materials"] == 3  # de-duplicated: teapot, floor, glass


def test_bvh_structure_mixed_and_degenerate_inputs():
    world, _ = scenes.stress(64, 36, n_top=2, n_mid=10, n_leaf=50, extent=60.0)
    rc, info = _check(world)
    assert rc == 0, _abi.lib().rt355_last_error()
    assert info["bounded"] == 1000 and info["unbounded"] == 1  # groups are inert placeholders, the floor is unbounded
    rc, info = _check(world, exact_cones=1)  # truncated cones move to the every-ray list
    assert rc == 0 and info["unbounded"] == 1 + 250 and info["bounded"] == 750
    # one shape, no shape, groups only, many coincident shapes
    for shapes, bounded in ([Shape.Sphere()], 1), ([], 0), ([Shape.Group(), Shape.Group()], 0), ([Shape.Cube() for _ in range(40)], 40):
        rc, info = _check(World(None, shapes))
        assert rc == 0 and info["bounded"] == bounded, (_abi.lib().rt355_last_error(), info)


def test_scene_check_rejects_what_the_reference_panics_on():
    lib = _abi.lib()
    out = (C.c_uint64 * 5)()
    singular = World(None, [Shape.Sphere(mx.scaling(1, 0, 1))])  # try_inverse().unwrap() in src/ray.rs:34
    assert lib.rt355_scene_check(singular.descs(), 1, 64, 0, out) == -1
    assert b"singular" in lib.rt355_last_error() or b"invert" in lib.rt355_last_error()
    nan = World(None, [Shape.Sphere(mx.translation(float("nan"), 0, 0))])
    assert lib.rt355_scene_check(nan.descs(), 1, 64, 0, out) == -1
    assert lib.rt355_scene_check(singular.descs(), 1, 48, 0, out) == -1


@pytest.mark.parametrize("real_bits", [32, 64])
def test_group_transformed_meshes_become_instances(real_bits, monkeypatch):
    """Runs of triangles that share one group transformation (src/group.rs:130-168) are built as instances: one
    entry each in the top-level tree, an object-space tree per mesh below.  rt355_scene_check walks both levels:
    every triangle reachable exactly once, every box encloses what lies below it."""
    world, _ = scenes.teapot_field(64, 36, "teapot-low.obj", 3, through_groups=True)
    assert world.n_shapes() == 1 + 1 + 9 * (1 + 240)  # floor, root group, per copy its group and its triangles
    rc, info = _check(world, real_bits)
    assert rc == 0, _abi.lib().rt355_last_error()
    assert info["bounded"] == 9 and info["unbounded"] == 1  # nine instances; the floor plane beside the tree
    assert info["stack"] <= 62
    nodes_instanced = info["nodes"]
    # meshes below the size that pays for a tree of their own stay world-space leaves; so does everything
    # with the feature switched off
    monkeypatch.setenv("RT355_INSTANCE_MIN", "100000")
    rc, info = _check(world, real_bits)
    assert rc == 0 and info["bounded"] == 9 * 240 and info["unbounded"] == 1
    assert info["nodes"] != nodes_instanced
    monkeypatch.delenv("RT355_INSTANCE_MIN")
    # an identity group leaves its triangles as plain world-space triangles: nothing to instance
    plain, _ = scenes.teapot_low(64, 36)
    rc, info = _check(plain, real_bits)
    assert rc == 0 and info["bounded"] == plain.n_shapes()


# ------------------------------------------------------------------------------------------------
# C++ host mirror (include/rt355.hpp): its own test program, built by __graft_entry__.build()
# ------------------------------------------------------------------------------------------------
CPP_TEST = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")


def test_cpp_host_mirror_cpu_part():
    """Matrix / view-transformation / camera / colour / PPM known answers of the reference, the ABI records
    the C++ types fill, the OBJ loader on the reference's fixtures and the group flattening, with no GPU."""
    assert os.path.exists(CPP_TEST), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    r = subprocess.run([CPP_TEST, "--cpu", OBJ], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpu: 0 failure(s)" in r.stdout


def test_orbit_turns_the_camera_about_the_world_y_axis(oracle):
    """scenes.orbit (the camera path bench.py steps through): angle 0 keeps the camera, and the orbiting camera's
    rays are the first camera's rays turned about Y (checked with the oracle's ray_for_pixel restatement through a
    render of a scene that is symmetric about the Y axis: the image does not change)."""
    from raytracer_amd import scenes
    from raytracer_amd.host import Material, PointLight, Shape, World
    world, cam = scenes.ch9(40, 30)
    same = scenes.orbit(cam, 0.0)
    assert np.array_equal(same.transform, cam.transform)
    # a sphere on the Y axis, lit from a point on the Y axis: symmetric under any turn about Y
    sym = World(PointLight((0.0, 10.0, 0.0), (1, 1, 1)), [Shape.Sphere(mx.translation(0, 1, 0), Material(color=(0.8, 0.3, 0.2)))])
    a, _, _ = oracle.render(sym, cam, 5, 64)
    b, _, _ = oracle.render(sym, scenes.orbit(cam, 0.7), 5, 64)
    assert np.abs(a - b).max() < 1e-5 and a.max() > 0.1
    # and a scene that is not symmetric does change
    c, _, _ = oracle.render(world, cam, 5, 64)
    d, _, _ = oracle.render(world, scenes.orbit(cam, 0.7), 5, 64)
    assert np.abs(c - d).max() > 0.05
