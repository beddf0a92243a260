// Tests of the C++ host mirror (include/rt355.hpp), restating the reference's own unit tests for the host
// side of the render path.  `--cpu` needs no GPU; `--gpu <out.f32>` renders through the C ABI and also
// writes the ch9 canvas (raw float32, 200x150x3) for the caller to compare with the Python path.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "rt355.hpp"

using namespace raytracer;

static int failures = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);          \
      ++failures;                                                          \
    }                                                                      \
  } while (0)

static bool near(float a, float b, float eps = 1e-5f) { return std::fabs(a - b) <= eps; }
static bool near(const Vector4& a, const Vector4& b, float eps = 1e-5f) {
  return near(a.x, b.x, eps) && near(a.y, b.y, eps) && near(a.z, b.z, eps) && near(a.w, b.w, eps);
}
static bool near(const Matrix4& a, const Matrix4& b, float eps = 1e-5f) {
  for (int k = 0; k < 16; ++k)
    if (!near(a.m[k], b.m[k], eps)) return false;
  return true;
}
static const float PI = 3.14159265358979323846f;

static void cpu_tests() {
  // src/tuple.rs:73-88
  CHECK(near(Vector4::vector(1, -1, 0).reflect(Vector4::vector(0, 1, 0)), Vector4::vector(1, 1, 0)));
  const float s = std::sqrt(2.f) / 2.f;
  CHECK(near(Vector4::vector(0, -1, 0).reflect(Vector4::vector(s, s, 0)), Vector4::vector(1, 0, 0)));
  // src/tuple.rs:91-137
  const Vector4 up = Vector4::vector(0, 1, 0);
  CHECK(near(Vector4::point(0, 0, 0).view_transformation(Vector4::point(0, 0, -1), up), Matrix4::identity()));
  CHECK(near(Vector4::point(0, 0, 0).view_transformation(Vector4::point(0, 0, 1), up), Matrix4::scaling(-1, 1, -1)));
  CHECK(near(Vector4::point(0, 0, 8).view_transformation(Vector4::point(0, 0, 0), up), Matrix4::translation(0, 0, -8)));
  {
    const Matrix4 v = Vector4::point(1, 3, 2).view_transformation(Vector4::point(4, -2, 8), Vector4::vector(1, 1, 0));
    const float want[4][4] = {{-0.50709f, 0.50709f, 0.67612f, -2.36643f}, {0.76772f, 0.60609f, 0.12122f, -2.82843f},
                              {-0.35857f, 0.59761f, -0.71714f, 0.f}, {0, 0, 0, 1}};
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) CHECK(near(v.at(r, c), want[r][c], EPSILON));
  }
  // transform builders (src/matrix.rs:15-89): column-major storage, row-major meaning
  CHECK(near(Matrix4::rotation_x(PI / 2) * Vector4::point(0, 1, 0), Vector4::point(0, 0, 1)));
  CHECK(near(Matrix4::rotation_y(PI / 2) * Vector4::point(0, 0, 1), Vector4::point(1, 0, 0)));
  CHECK(near(Matrix4::rotation_z(PI / 2) * Vector4::point(0, 1, 0), Vector4::point(-1, 0, 0)));
  CHECK(near(Matrix4::translation(5, -3, 2) * Vector4::point(-3, 4, 5), Vector4::point(2, 1, 7)));
  CHECK(near(Matrix4::translation(5, -3, 2) * Vector4::vector(-3, 4, 5), Vector4::vector(-3, 4, 5)));
  CHECK(near(Matrix4::scaling(2, 3, 4) * Vector4::point(-4, 6, 8), Vector4::point(-8, 18, 32)));
  CHECK(Matrix4::translation(1, 2, 3).m[12] == 1.f && Matrix4::translation(1, 2, 3).m[14] == 3.f);
  CHECK(near((Matrix4::translation(1, 0, 0) * Matrix4::scaling(2, 2, 2)) * Vector4::point(1, 1, 1), Vector4::point(3, 2, 2)));
  // src/camera.rs:116-135
  CHECK(near(Camera::new_default(200, 125, PI / 2).pixel_size(), 0.01f, 1e-6f));
  CHECK(near(Camera::new_default(125, 200, PI / 2).pixel_size(), 0.01f, 1e-6f));
  // src/color.rs:97-111
  {
    const ColorU8 a = Color::rgb(1.0f, 0.6f, 0.4f).to_u8(), b = Color::rgb(1.5f, -0.2f, NAN).to_u8(), c = Color::rgb(0.5f / 255, 1.5f / 255, 2.5f / 255).to_u8();
    CHECK(a.r == 255 && a.g == 153 && a.b == 102);
    CHECK(b.r == 255 && b.g == 0 && b.b == 0);
    CHECK(c.r == 1 && c.g == 2 && c.b == 3);
  }
  // src/ppm.rs:81-103 and the layout of :30-59
  {
    Canvas cv;
    cv.width = 2, cv.height = 3;
    const Color c1 = Color::rgb(1.0f, 0.6f, 0.4f), c2 = Color::rgb(0.1f, 0.2f, 0.8f);
    cv.pixels = {c1, c2, c1, c2, c1, c2};
    CHECK(cv.to_ppm() == "P3\n2 3\n255\n255 153 102 26 51 204 255 153 102 26 51 204 \n255 153 102 26 51 204 \n");
    CHECK(cv.pixel_at(1, 2).b == c2.b);
  }
  // materials and shapes fill the ABI records (src/material.rs:31-106, src/shape.rs:52-67)
  {
    const rt355_material g = Material::new_glass().desc();
    CHECK(g.transparency == 1.0f && g.refractive == 1.5f && g.ambient == 0.1f && g.shininess == 200.f);
    Material m(Color::rgb(1, 0, 0), {}, {}, {}, {}, 0.5f, {}, {}, Pattern::checker(Color::rgb(1, 1, 1), Color::rgb(0, 0, 0)));
    const rt355_material d = m.desc();
    CHECK(d.color[0] == 1.f && d.color[1] == 0.f && d.reflective == 0.5f && d.diffuse == 0.9f && d.pattern_kind == RT355_PATTERN_CHECKER && d.c1[2] == 1.f);
    const World w = World::new_default();
    CHECK(w.shapes.size() == 2 && w.pointlight.position.x == -10.f && w.pointlight.intensity.g == 1.f);
    const rt355_shape_desc s0 = w.shapes[0].desc(), s1 = w.shapes[1].desc();
    CHECK(s0.kind == RT355_SPHERE && near(s0.mat.color[0], 0.8f) && near(s0.mat.diffuse, 0.7f) && near(s0.mat.specular, 0.2f));
    CHECK(s1.m[0] == 0.5f && s1.m[5] == 0.5f && s1.m[10] == 0.5f && s1.m[15] == 1.f);
    const Shape cyl = Shape::cylinder(Matrix4::translation(1, 2, 3), {}, 0.f, 1.f, true);
    const rt355_shape_desc cd = cyl.desc();
    CHECK(cd.kind == RT355_CYLINDER && cd.min == 0.f && cd.max == 1.f && cd.closed == 1 && cd.m[13] == 2.f);
    const rt355_shape_desc dd = Shape::cone().desc();
    CHECK(dd.kind == RT355_CONE && std::isinf(dd.min) && dd.min < 0 && std::isinf(dd.max) && dd.closed == 0);
    const Shape t = Shape::smooth_triangle({}, {}, Vector4::point(0, 1, 0), Vector4::point(-1, 0, 0), Vector4::point(1, 0, 0),
                                           Vector4::vector(0, 1, 0), Vector4::vector(-1, 0, 0), Vector4::vector(1, 0, 0));
    const rt355_shape_desc td = t.desc();
    CHECK(td.kind == RT355_SMOOTH_TRIANGLE && td.p2[0] == -1.f && td.n3[0] == 1.f);
    // the same records through the host-only builder of the library
    uint64_t info[5];
    std::vector<rt355_shape_desc> descs = {s0, s1, cd, td, Shape::plane().desc(), Shape::group().desc()};
    CHECK(rt355_scene_check(descs.data(), descs.size(), 64, 0, info) == RT355_OK);
    CHECK(info[1] == 4 && info[2] == 1);  // plane is unbounded, the group is a placeholder
    // a singular transformation: the reference panics in try_inverse().unwrap() (src/ray.rs:34)
    rt355_shape_desc bad = Shape::sphere(Matrix4::scaling(1, 0, 1)).desc();
    CHECK(rt355_scene_check(&bad, 1, 64, 0, info) == RT355_EINVAL);
  }
}

static bool same3(const Vector4& a, const Vector4& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// src/wavefront.rs:192-305 (fixtures: the reference's own small OBJ files) and src/group.rs:117-168
static void loader_tests(const std::string& obj_dir) {
  {
    const Wavefront w = Wavefront::parse(obj_dir + "/vertex.obj");
    CHECK(w.vertices.size() == 5 && same3(w.vertices[1], Vector4::point(-1, 1, 0)) && same3(w.vertices[2], Vector4::point(-1, 0.5f, 0)) &&
          same3(w.vertices[3], Vector4::point(1, 0, 0)) && same3(w.vertices[4], Vector4::point(1, 1, 0)));
  }
  {
    const Wavefront w = Wavefront::parse(obj_dir + "/face.obj");
    CHECK(w.shapes.size() == 2 && w.shapes[0].kind == RT355_SMOOTH_TRIANGLE);
    CHECK(same3(w.shapes[0].p1, w.vertices[1]) && same3(w.shapes[0].p2, w.vertices[2]) && same3(w.shapes[0].p3, w.vertices[3]));
    CHECK(same3(w.shapes[1].p1, w.vertices[1]) && same3(w.shapes[1].p2, w.vertices[3]) && same3(w.shapes[1].p3, w.vertices[4]));
  }
  for (int quirk = 0; quirk < 2; ++quirk) {  // the face is `1 2 3 4 5`, where both fans coincide
    const Wavefront w = Wavefront::parse(obj_dir + "/polygones.obj", quirk != 0);
    CHECK(w.shapes.size() == 3);
    const int idx[3][3] = {{1, 2, 3}, {1, 3, 4}, {1, 4, 5}};
    for (size_t k = 0; k < 3 && k < w.shapes.size(); ++k)
      CHECK(same3(w.shapes[k].p1, w.vertices[idx[k][0]]) && same3(w.shapes[k].p2, w.vertices[idx[k][1]]) && same3(w.shapes[k].p3, w.vertices[idx[k][2]]));
  }
  {
    const Wavefront w = Wavefront::parse(obj_dir + "/vertex_normal.obj");
    CHECK(w.normals.size() == 4 && same3(w.normals[1], Vector4::vector(0, 0, 1)) && same3(w.normals[2], Vector4::vector(0.707f, 0, -0.707f)) &&
          same3(w.normals[3], Vector4::vector(1, 2, 3)));
  }
  {
    const Wavefront w = Wavefront::parse(obj_dir + "/face_with_normal.obj");
    CHECK(w.shapes.size() == 2);
    for (const Shape& s : w.shapes)
      CHECK(same3(s.p1, w.vertices[1]) && same3(s.p2, w.vertices[2]) && same3(s.p3, w.vertices[3]) && same3(s.n1, w.normals[3]) &&
            same3(s.n2, w.normals[1]) && same3(s.n3, w.normals[2]));
  }
  {
    bool threw = false;
    try {
      Wavefront::parse(obj_dir + "/no_such_file.obj");
    } catch (const Error&) {
      threw = true;
    }
    CHECK(threw);
  }
  // groups: one-level composition, depth-first order, inputs untouched
  {
    Node g1(Shape::group(Matrix4::rotation_y(PI / 2)));
    Node& g2 = g1.add(Shape::group(Matrix4::scaling(2, 2, 2)));
    g2.add(Shape::sphere(Matrix4::translation(5, 0, 0)));
    const std::vector<Shape> flat = to_transformed_shapes(g1);
    CHECK(flat.size() == 3 && flat[0].kind == RT355_GROUP && flat[1].kind == RT355_GROUP && flat[2].kind == RT355_SPHERE);
    CHECK(near(flat[1].transformation, Matrix4::rotation_y(PI / 2) * Matrix4::scaling(2, 2, 2)));
    CHECK(near(flat[2].transformation, Matrix4::scaling(2, 2, 2) * Matrix4::translation(5, 0, 0)));
    const std::vector<Shape> acc = to_transformed_shapes(g1, true);
    CHECK(near(acc[2].transformation, Matrix4::rotation_y(PI / 2) * Matrix4::scaling(2, 2, 2) * Matrix4::translation(5, 0, 0), 1e-4f));
    Node root(Shape::sphere(Matrix4::scaling(3, 3, 3)));
    root.add(Shape::cube(Matrix4::translation(1, 0, 0)));
    const std::vector<Shape> pass = to_transformed_shapes(root);  // src/group.rs:154-166
    CHECK(pass.size() == 2 && pass[1].kind == RT355_CUBE && near(pass[1].transformation, Matrix4::translation(1, 0, 0)));
  }
}

static World ch9_world() {  // src/main.rs:202-273
  const Shape floor = Shape::plane({}, Material(Color::rgb(1.0f, 0.9f, 0.9f), {}, {}, {}, {}, {}, {}, {}, {}));
  const Shape middle = Shape::sphere(Matrix4::translation(-0.5f, 1.0f, 0.5f), Material(Color::rgb(1.0f, 0.2f, 0.2f), {}, {}, {}, {}, {}, {}, {}, {}));
  const Shape right = Shape::sphere(Matrix4::translation(1.5f, 0.5f, -0.5f) * Matrix4::scaling(0.5f, 0.5f, 0.5f),
                                    Material(Color::rgb(0.3f, 0.4f, 1.0f), {}, {}, {}, {}, {}, {}, {}, {}));
  return World(PointLight(Vector4::point(-10, 10, -10), Color::rgb(1, 1, 1)), {floor, middle, right});
}

static void gpu_tests(const char* out_path) {
  // src/camera.rs:155-169 rendering_world_with_camera
  {
    const Camera c(11, 11, PI / 2, Vector4::point(0, 0, -5).view_transformation(Vector4::point(0, 0, 0), Vector4::vector(0, 1, 0)));
    const Canvas image = c.render(World::new_default());
    CHECK(image.width == 11 && image.height == 11 && image.pixels.size() == 121);
    const Color p = image.pixel_at(5, 5);
    CHECK(near(p.r, 0.38066f, EPSILON) && near(p.g, 0.47583f, EPSILON) && near(p.b, 0.2855f, EPSILON));
  }
  // error behaviour: what panics in the reference throws here
  {
    bool threw = false;
    try {
      Camera::new_default(4, 4, 1.f).render(World({}, {Shape::sphere(Matrix4::scaling(0, 1, 1))}));
    } catch (const Error& e) {
      threw = e.code == RT355_EINVAL;
    }
    CHECK(threw);
  }
  // the default `cargo run` scene, one upload, two renders through an explicit device
  {
    Device dev(0, 64);
    const World w = ch9_world();
    const Camera c(200, 150, PI / 3, Vector4::point(0, 1.5f, -5).view_transformation(Vector4::point(0, 1, 0), Vector4::vector(0, 1, 0)));
    Scene scene(dev, w);
    const Canvas a = scene.render(c), b = scene.render(c);
    CHECK(a.pixels.size() == 200u * 150u);
    bool same = true, lit = false;
    for (size_t i = 0; i < a.pixels.size(); ++i) {
      same = same && a.pixels[i].r == b.pixels[i].r && a.pixels[i].g == b.pixels[i].g && a.pixels[i].b == b.pixels[i].b;
      lit = lit || a.pixels[i].r > 0.5f;
    }
    CHECK(same && lit);
    CHECK(a.to_ppm().rfind("P3\n200 150\n255\n", 0) == 0);
    // a camera path through one launch chain: every frame equals its own render
    {
      const Camera c2(200, 150, PI / 3, Vector4::point(0.5f, 1.5f, -5).view_transformation(Vector4::point(0, 1, 0), Vector4::vector(0, 1, 0)));
      const std::vector<Canvas> path = scene.render_path({c, c2, c});
      const Canvas b2 = scene.render(c2);
      CHECK(path.size() == 3);
      bool eq = path.size() == 3;
      for (size_t i = 0; eq && i < a.pixels.size(); ++i)
        eq = path[0].pixels[i].r == a.pixels[i].r && path[1].pixels[i].g == b2.pixels[i].g && path[2].pixels[i].b == a.pixels[i].b;
      CHECK(eq);
    }
    if (out_path) {
      FILE* f = std::fopen(out_path, "wb");
      CHECK(f != nullptr);
      if (f) {
        for (const Color& p : a.pixels) {
          const float v[3] = {p.r, p.g, p.b};
          std::fwrite(v, sizeof(float), 3, f);
        }
        std::fclose(f);
      }
    }
  }
}

int main(int argc, char** argv) {
  const bool gpu = argc > 1 && std::strcmp(argv[1], "--gpu") == 0;
  try {
    if (gpu)
      gpu_tests(argc > 2 ? argv[2] : nullptr);
    else {
      cpu_tests();
      if (argc > 2 && std::strcmp(argv[1], "--cpu") == 0) loader_tests(argv[2]);
    }
  } catch (const std::exception& e) {
    std::printf("FAIL exception: %s\n", e.what());
    ++failures;
  }
  std::printf("%s: %d failure(s)\n", gpu ? "gpu" : "cpu", failures);
  return failures ? 1 : 0;
}
