// rt355.hpp -- C++ host side above the C ABI of rt355.h: the reference's own host interface for the render
// path, name for name, so that code written against SokichiFujita/raytracer reads the same here.
//
//   reference (Rust)                                   here
//   Matrix4::translation(..) * Matrix4::scaling(..)    Matrix4::translation(..) * Matrix4::scaling(..)     src/matrix.rs:4-89
//   Vector4::point / vector / reflect / view_transformation                                                src/tuple.rs:5-66
//   Color::rgb, to_u8                                                                                      src/color.rs:72-111
//   Material::new(Option..) / new_default / new_default_world / new_glass                                  src/material.rs:19-106
//   Stripe::new / Gradient::new / Ring::new / Checker::new / Test::new  -> Pattern::stripe(..) ...         src/pattern.rs:44-91
//   Shape::Sphere(Sphere::new(t, m, ..)) ...           Shape::sphere(t, m) ... (std::optional = Option)    src/sphere.rs:22 etc.
//   PointLight::new / new_default, World::new / new_default                                                src/pointlight.rs:14-22, src/world.rs:23-34
//   Camera::new(h, v, fov, transform).render(&world) -> Canvas                                             src/camera.rs:23-88
//   Canvas { width, height, pixels }, to_ppm                                                               src/canvas.rs:8-25, src/ppm.rs:30-59
//
// Where the reference panics (singular transformation, NaN) these calls throw raytracer::Error carrying
// rt355_last_error().  Rendering needs librt355 and a GPU; everything else in this header is plain host
// arithmetic in f32, like the reference's.
#ifndef RT355_HPP
#define RT355_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <fstream>
#include <limits>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt355.h"

namespace raytracer {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc) {
  if (rc != RT355_OK) {
    const char* m = rt355_last_error();
    throw Error(rc, std::string("rt355 error ") + std::to_string(rc) + ": " + (m ? m : ""));
  }
}

constexpr float EPSILON = 1.0e-3f;  // src/epsilon.rs:1

struct Matrix4;

// ---- tuple.rs -----------------------------------------------------------------------------------
struct Vector4 {
  float x = 0, y = 0, z = 0, w = 0;
  static Vector4 point(float x, float y, float z) { return {x, y, z, 1.f}; }   // src/tuple.rs:18-20
  static Vector4 vector(float x, float y, float z) { return {x, y, z, 0.f}; }  // src/tuple.rs:22-24
  static bool is_point(const Vector4& v) { return v.w == 1.f; }
  static bool is_vector(const Vector4& v) { return v.w == 0.f; }
  float dot(const Vector4& o) const { return x * o.x + y * o.y + z * o.z + w * o.w; }
  float magnitude() const { return std::sqrt(dot(*this)); }
  Vector4 normalize() const {
    const float m = magnitude();
    return {x / m, y / m, z / m, w / m};
  }
  Vector4 operator+(const Vector4& o) const { return {x + o.x, y + o.y, z + o.z, w + o.w}; }
  Vector4 operator-(const Vector4& o) const { return {x - o.x, y - o.y, z - o.z, w - o.w}; }
  Vector4 operator*(float k) const { return {x * k, y * k, z * k, w * k}; }
  Vector4 operator-() const { return {-x, -y, -z, -w}; }
  Vector4 reflect(const Vector4& normal) const { return *this - normal * 2.f * dot(normal); }  // src/tuple.rs:38-40
  Vector4 cross4(const Vector4& o) const {                                                     // src/tuple.rs:42-48
    return vector(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x);
  }
  inline Matrix4 view_transformation(const Vector4& to, const Vector4& up) const;  // src/tuple.rs:50-60
};

// ---- matrix.rs ----------------------------------------------------------------------------------
// Stored column-major, as nalgebra stores Matrix4<f32> and as rt355_shape_desc.m expects.
struct Matrix4 {
  float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float& at(int r, int c) { return m[c * 4 + r]; }
  float at(int r, int c) const { return m[c * 4 + r]; }
  static Matrix4 identity() { return Matrix4(); }
  static Matrix4 rotation_x(float a) {
    Matrix4 r;
    const float c = std::cos(a), s = std::sin(a);
    r.at(1, 1) = c, r.at(1, 2) = -s, r.at(2, 1) = s, r.at(2, 2) = c;
    return r;
  }
  static Matrix4 rotation_y(float a) {
    Matrix4 r;
    const float c = std::cos(a), s = std::sin(a);
    r.at(0, 0) = c, r.at(0, 2) = s, r.at(2, 0) = -s, r.at(2, 2) = c;
    return r;
  }
  static Matrix4 rotation_z(float a) {
    Matrix4 r;
    const float c = std::cos(a), s = std::sin(a);
    r.at(0, 0) = c, r.at(0, 1) = -s, r.at(1, 0) = s, r.at(1, 1) = c;
    return r;
  }
  static Matrix4 scaling(float x, float y, float z) {
    Matrix4 r;
    r.at(0, 0) = x, r.at(1, 1) = y, r.at(2, 2) = z;
    return r;
  }
  static Matrix4 translation(float x, float y, float z) {
    Matrix4 r;
    r.at(0, 3) = x, r.at(1, 3) = y, r.at(2, 3) = z;
    return r;
  }
  static Matrix4 translate(float x, float y, float z) { return translation(x, y, z); }  // src/matrix.rs:85-89
  Matrix4 operator*(const Matrix4& o) const {
    Matrix4 r;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        float s = 0;
        for (int k = 0; k < 4; ++k) s += at(i, k) * o.at(k, j);
        r.at(i, j) = s;
      }
    return r;
  }
  Vector4 operator*(const Vector4& v) const {
    const float in[4] = {v.x, v.y, v.z, v.w};
    float out[4];
    for (int i = 0; i < 4; ++i) {
      float s = 0;
      for (int k = 0; k < 4; ++k) s += at(i, k) * in[k];
      out[i] = s;
    }
    return {out[0], out[1], out[2], out[3]};
  }
};

inline Matrix4 Vector4::view_transformation(const Vector4& to, const Vector4& up) const {
  const Vector4 forward = (to - *this).normalize();
  const Vector4 up_normal = up.normalize();
  const Vector4 left = forward.cross4(up_normal);
  const Vector4 true_up = left.cross4(forward);
  Matrix4 o;
  o.at(0, 0) = left.x, o.at(0, 1) = left.y, o.at(0, 2) = left.z;
  o.at(1, 0) = true_up.x, o.at(1, 1) = true_up.y, o.at(1, 2) = true_up.z;
  o.at(2, 0) = -forward.x, o.at(2, 1) = -forward.y, o.at(2, 2) = -forward.z;
  return o * Matrix4::translation(-x, -y, -z);
}

// ---- color.rs -----------------------------------------------------------------------------------
struct ColorU8 {
  uint8_t r, g, b;
};
struct Color {
  float r = 0, g = 0, b = 0;
  static Color rgb(float r, float g, float b) { return {r, g, b}; }
  Color operator+(const Color& o) const { return {r + o.r, g + o.g, b + o.b}; }
  Color operator-(const Color& o) const { return {r - o.r, g - o.g, b - o.b}; }
  Color operator*(float k) const { return {r * k, g * k, b * k}; }
  Color hadamard_product(const Color& o) const { return {r * o.r, g * o.g, b * o.b}; }
  // src/color.rs:97-111: round half away from zero, clamp to 0..255, NaN -> 0
  static uint8_t channel_to_u8(float c) {
    const float v = std::round(c * 255.f);
    if (!(v > 0.f)) return 0;
    return v >= 255.f ? 255 : (uint8_t)v;
  }
  ColorU8 to_u8() const { return {channel_to_u8(r), channel_to_u8(g), channel_to_u8(b)}; }
};

// ---- refractive.rs ------------------------------------------------------------------------------
namespace RefractiveIndex {
constexpr float VACCUME = 1.0f, AIR = 1.00029f, WATER = 1.333f, GLASS = 1.5f, DIAMOND = 2.417f;  // src/refractive.rs:5-9
}

// ---- pattern.rs ---------------------------------------------------------------------------------
struct Pattern {
  int32_t kind = RT355_PATTERN_NONE;
  Color c1, c2;
  Matrix4 transformation;  // stored and never read by the render path (src/pattern.rs:94), as in the reference
  static Pattern stripe(Color a, Color b, Matrix4 t = Matrix4()) { return {RT355_PATTERN_STRIPE, a, b, t}; }
  static Pattern gradient(Color a, Color b, Matrix4 t = Matrix4()) { return {RT355_PATTERN_GRADIENT, a, b, t}; }
  static Pattern ring(Color a, Color b, Matrix4 t = Matrix4()) { return {RT355_PATTERN_RING, a, b, t}; }
  static Pattern checker(Color a, Color b, Matrix4 t = Matrix4()) { return {RT355_PATTERN_CHECKER, a, b, t}; }
  static Pattern test() { return {RT355_PATTERN_TEST, Color(), Color(), Matrix4()}; }
};

// ---- material.rs --------------------------------------------------------------------------------
struct Material {
  Color color = Color::rgb(1, 1, 1);
  float ambient = 0.1f, diffuse = 0.9f, specular = 0.9f, shininess = 200.f;
  float reflective = 0.f, refractive = RefractiveIndex::VACCUME, transparency = 0.f;
  std::optional<Pattern> pattern;
  Material() = default;
  // Material::new: None keeps the default of src/material.rs:31-72
  Material(std::optional<Color> color_, std::optional<float> ambient_, std::optional<float> diffuse_,
           std::optional<float> specular_, std::optional<float> shininess_, std::optional<float> reflective_,
           std::optional<float> refractive_, std::optional<float> transparency_, std::optional<Pattern> pattern_) {
    if (color_) color = *color_;
    if (ambient_) ambient = *ambient_;
    if (diffuse_) diffuse = *diffuse_;
    if (specular_) specular = *specular_;
    if (shininess_) shininess = *shininess_;
    if (reflective_) reflective = *reflective_;
    if (refractive_) refractive = *refractive_;
    if (transparency_) transparency = *transparency_;
    pattern = pattern_;
  }
  static Material new_default() { return Material(); }
  static Material new_default_world() {  // src/material.rs:80-92
    Material m;
    m.color = Color::rgb(0.8f, 1.0f, 0.6f), m.diffuse = 0.7f, m.specular = 0.2f;
    return m;
  }
  static Material new_glass() {  // src/material.rs:94-106
    Material m;
    m.transparency = 1.0f, m.refractive = RefractiveIndex::GLASS;
    return m;
  }
  rt355_material desc() const {
    rt355_material d{};
    d.color[0] = color.r, d.color[1] = color.g, d.color[2] = color.b;
    d.ambient = ambient, d.diffuse = diffuse, d.specular = specular, d.shininess = shininess;
    d.reflective = reflective, d.refractive = refractive, d.transparency = transparency;
    d.pattern_kind = pattern ? pattern->kind : RT355_PATTERN_NONE;
    if (pattern) {
      d.c1[0] = pattern->c1.r, d.c1[1] = pattern->c1.g, d.c1[2] = pattern->c1.b;
      d.c2[0] = pattern->c2.r, d.c2[1] = pattern->c2.g, d.c2[2] = pattern->c2.b;
    }
    return d;
  }
};

// ---- shape.rs and the shape structs --------------------------------------------------------------
struct Shape {
  int32_t kind = RT355_SPHERE;
  Matrix4 transformation;
  Material material;
  float min = -std::numeric_limits<float>::infinity(), max = std::numeric_limits<float>::infinity();
  bool closed = false;
  Vector4 p1, p2, p3, n1, n2, n3;

  using OptM = std::optional<Matrix4>;
  using OptMat = std::optional<Material>;
  static Shape make(int32_t kind, const OptM& t, const OptMat& m) {
    Shape s;
    s.kind = kind;
    if (t) s.transformation = *t;
    if (m) s.material = *m;
    return s;
  }
  static Shape sphere(OptM t = {}, OptMat m = {}) { return make(RT355_SPHERE, t, m); }
  static Shape plane(OptM t = {}, OptMat m = {}) { return make(RT355_PLANE, t, m); }
  static Shape cube(OptM t = {}, OptMat m = {}) { return make(RT355_CUBE, t, m); }
  static Shape group(OptM t = {}, OptMat m = {}) { return make(RT355_GROUP, t, m); }
  static Shape cylinder(OptM t = {}, OptMat m = {}, std::optional<float> min = {}, std::optional<float> max = {},
                        std::optional<bool> closed = {}) {
    Shape s = make(RT355_CYLINDER, t, m);
    if (min) s.min = *min;
    if (max) s.max = *max;
    if (closed) s.closed = *closed;
    return s;
  }
  static Shape cone(OptM t = {}, OptMat m = {}, std::optional<float> min = {}, std::optional<float> max = {},
                    std::optional<bool> closed = {}) {
    Shape s = cylinder(t, m, min, max, closed);
    s.kind = RT355_CONE;
    return s;
  }
  static Shape triangle(OptM t, OptMat m, Vector4 p1, Vector4 p2, Vector4 p3) {  // src/triangle.rs:28-44
    Shape s = make(RT355_TRIANGLE, t, m);
    s.p1 = p1, s.p2 = p2, s.p3 = p3;
    return s;
  }
  static Shape smooth_triangle(OptM t, OptMat m, Vector4 p1, Vector4 p2, Vector4 p3, Vector4 n1 = Vector4::vector(0, 0, 0),
                               Vector4 n2 = Vector4::vector(0, 0, 0), Vector4 n3 = Vector4::vector(0, 0, 0)) {  // src/smooth_triangle.rs:30-88
    Shape s = make(RT355_SMOOTH_TRIANGLE, t, m);
    s.p1 = p1, s.p2 = p2, s.p3 = p3, s.n1 = n1, s.n2 = n2, s.n3 = n3;
    return s;
  }
  static std::vector<Shape> new_world_defaults() {  // src/shape.rs:52-67
    return {sphere({}, Material::new_default_world()), sphere(Matrix4::scaling(0.5f, 0.5f, 0.5f), {})};
  }
  rt355_shape_desc desc() const {
    rt355_shape_desc d{};
    d.kind = kind;
    for (int k = 0; k < 16; ++k) d.m[k] = transformation.m[k];
    d.mat = material.desc();
    d.min = min, d.max = max, d.closed = closed ? 1 : 0;
    const Vector4* src[6] = {&p1, &p2, &p3, &n1, &n2, &n3};
    float* dst[6] = {d.p1, d.p2, d.p3, d.n1, d.n2, d.n3};
    for (int k = 0; k < 6; ++k) dst[k][0] = src[k]->x, dst[k][1] = src[k]->y, dst[k][2] = src[k]->z;
    return d;
  }
};

// ---- group.rs: the scene tree and its flattening ---------------------------------------------------
// `Scene::to_transformed_shapes` (src/group.rs:117-168) runs on the host before the render path: a World
// only ever holds a flat list in which Group entries are inert.  The reference composes only the IMMEDIATE
// parent group's transformation into a child (src/group.rs:146: the recursion re-reads the child's original
// transformation, so outer groups do not accumulate), pushes the root and every child in depth-first
// order, and lets children of a non-group node pass through unchanged.  accumulate = true composes the
// whole parent chain instead (the evident intent), for new scenes.
struct Node {
  Shape shape;
  std::vector<Node> children;
  explicit Node(Shape s) : shape(std::move(s)) {}
  Node& add(Shape s) {
    children.emplace_back(std::move(s));
    return children.back();
  }
};
inline void fold_transformation(const Node& node, const Matrix4* carried, bool accumulate, std::vector<Shape>& out) {
  const bool is_group = node.shape.kind == RT355_GROUP;
  for (const Node& ch : node.children) {
    Shape s = ch.shape;
    if (is_group) s.transformation = (accumulate && carried ? *carried : node.shape.transformation) * ch.shape.transformation;
    out.push_back(s);
    const Matrix4 next = is_group ? s.transformation : ch.shape.transformation;
    fold_transformation(ch, accumulate ? &next : nullptr, accumulate, out);
  }
}
inline std::vector<Shape> to_transformed_shapes(const Node& root, bool accumulate = false) {
  std::vector<Shape> out{root.shape};
  fold_transformation(root, accumulate ? &root.shape.transformation : nullptr, accumulate, out);
  return out;
}

// ---- wavefront.rs: OBJ -> smooth triangles ------------------------------------------------------------
// `Wavefront::parse` (src/wavefront.rs:39-184): `v`, `vn`, `g`, `f` lines; indices are 1-based (slot 0 of
// vertices / normals is a dummy, :48-49); a face whose first vertex contains '/' is read as v/vt/vn; every
// triangle becomes a SmoothTriangle.  Polygons are fanned over the FACE's own vertices; the reference fans
// over the file-global vertices 1, k, k+1 (:101-116, :143-155), which coincides only for faces that are
// literally `1 2 3 4 ...` -- ref_parser_quirk = true reproduces that.
struct Wavefront {
  std::vector<Vector4> vertices{Vector4::point(0, 0, 0)};
  std::vector<Vector4> normals{Vector4::vector(0, 0, 0)};
  std::vector<std::string> groups, comments;
  std::vector<Shape> shapes;

  static Wavefront parse(const std::string& path, bool ref_parser_quirk = false, std::optional<Material> material = {},
                         std::optional<Matrix4> transformation = {}) {
    std::ifstream in(path);
    if (!in) throw Error(RT355_EINVAL, "cannot open " + path);
    Wavefront w;
    std::string line;
    while (std::getline(in, line)) {
      std::istringstream ls(line);
      std::vector<std::string> words;
      for (std::string t; ls >> t;) words.push_back(t);
      if (words.empty() || words[0].size() > 2) {
        w.comments.push_back(line);
      } else if (words[0] == "v" && words.size() >= 4) {
        w.vertices.push_back(Vector4::point(std::stof(words[1]), std::stof(words[2]), std::stof(words[3])));
      } else if (words[0] == "vn" && words.size() >= 4) {
        w.normals.push_back(Vector4::vector(std::stof(words[1]), std::stof(words[2]), std::stof(words[3])));
      } else if (words[0] == "g") {
        w.groups.push_back(line);
      } else if (words[0] == "f") {
        w.face(std::vector<std::string>(words.begin() + 1, words.end()), ref_parser_quirk, material, transformation);
      } else {
        w.comments.push_back(line);
      }
    }
    return w;
  }

 private:
  void face(const std::vector<std::string>& words, bool quirk, const std::optional<Material>& material, const std::optional<Matrix4>& transformation) {
    if (words.size() < 3) return;
    const bool with_normals = words[0].find('/') != std::string::npos;
    const Vector4 zero = Vector4::vector(0, 0, 0);
    std::vector<Vector4> vs, ns;
    for (const std::string& word : words) {
      if (with_normals) {
        std::vector<std::string> f;
        std::string part;
        std::istringstream ws(word);
        while (std::getline(ws, part, '/')) f.push_back(part);
        // `v/vt` (no normal index) is read as a vertex without a normal; the reference indexes f[2]
        // unconditionally (src/wavefront.rs:93) and panics on such a file
        vs.push_back(vertices.at((size_t)std::stoi(f.at(0))));
        ns.push_back(f.size() >= 3 && !f[2].empty() ? normals.at((size_t)std::stoi(f[2])) : zero);
      } else {
        vs.push_back(vertices.at((size_t)std::stoi(word)));
        ns.push_back(zero);
      }
    }
    const size_t l = vs.size();
    auto emit = [&](Vector4 a, Vector4 b, Vector4 c, Vector4 na, Vector4 nb, Vector4 nc) {
      shapes.push_back(Shape::smooth_triangle(transformation, material, a, b, c, na, nb, nc));
    };
    if (l == 3) {
      emit(vs[0], vs[1], vs[2], ns[0], ns[1], ns[2]);
    } else if (quirk) {
      for (size_t index = 2; index < l; ++index)
        emit(vertices.at(1), vertices.at(index), vertices.at(index + 1), with_normals ? normals.at(1) : zero,
             with_normals ? normals.at(index) : zero, with_normals ? normals.at(index + 1) : zero);
    } else {
      for (size_t i = 1; i + 1 < l; ++i) emit(vs[0], vs[i], vs[i + 1], ns[0], ns[i], ns[i + 1]);
    }
  }
};

// ---- pointlight.rs, world.rs ----------------------------------------------------------------------
struct PointLight {
  Vector4 position = Vector4::point(-10.f, 10.f, -10.f);  // src/pointlight.rs:20-22
  Color intensity = Color::rgb(1, 1, 1);
  PointLight() = default;
  PointLight(Vector4 p, Color i) : position(p), intensity(i) {}
  static PointLight new_default() { return PointLight(); }
};

struct World {
  PointLight pointlight;
  std::vector<Shape> shapes;
  World(std::optional<PointLight> light, std::vector<Shape> shapes_) : pointlight(light ? *light : PointLight()), shapes(std::move(shapes_)) {}
  static World new_default() { return World({}, Shape::new_world_defaults()); }  // src/world.rs:32-34
};

// ---- canvas.rs, ppm.rs ----------------------------------------------------------------------------
struct Canvas {
  size_t width = 0, height = 0;
  std::vector<Color> pixels;  // row-major from the top-left, unclamped linear RGB
  const Color& pixel_at(size_t x, size_t y) const { return pixels.at(y * width + x); }
  // `Canvas::to_ppm` -> `Ppm::new` (src/ppm.rs:30-59): header, then 4 pixels per line, every number followed
  // by a space, len/4 + 1 lines
  std::string to_ppm() const {
    std::string s = "P3\n" + std::to_string(width) + " " + std::to_string(height) + "\n255\n";
    for (size_t n = 0; n < pixels.size() / 4 + 1; ++n) {
      for (size_t k = n * 4; k < n * 4 + 4 && k < pixels.size(); ++k) {
        const ColorU8 c = pixels[k].to_u8();
        s += std::to_string(c.r) + " " + std::to_string(c.g) + " " + std::to_string(c.b) + " ";
      }
      s += "\n";
    }
    return s;
  }
};

// ---- the device: one context per GPU, scenes uploaded once ----------------------------------------
class Device {
 public:
  explicit Device(int device_id = 0, int real_bits = 64) { check(rt355_ctx_create(&ctx_, device_id, real_bits)); }
  ~Device() { rt355_ctx_destroy(ctx_); }
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;
  rt355_ctx* handle() const { return ctx_; }

 private:
  rt355_ctx* ctx_ = nullptr;
};

class Camera;

class Scene {
 public:
  Scene(Device& dev, const World& world) : dev_(dev) {
    std::vector<rt355_shape_desc> descs;
    descs.reserve(world.shapes.size());
    for (const Shape& s : world.shapes) descs.push_back(s.desc());
    rt355_light_desc light{};
    light.position[0] = world.pointlight.position.x, light.position[1] = world.pointlight.position.y, light.position[2] = world.pointlight.position.z;
    light.intensity[0] = world.pointlight.intensity.r, light.intensity[1] = world.pointlight.intensity.g, light.intensity[2] = world.pointlight.intensity.b;
    check(rt355_scene_upload(dev.handle(), descs.data(), descs.size(), &light, &scene_));
  }
  ~Scene() { rt355_scene_destroy(scene_); }
  Scene(const Scene&) = delete;
  Scene& operator=(const Scene&) = delete;
  inline Canvas render(const Camera& camera, int max_depth = 5, rt355_stats* stats = nullptr) const;
  // Camera::render for every camera of a path (all of one image size), in launch chains of up to RT355_MAX_BATCH
  // frames whose rays are traced together (rt355_render_batch).
  inline std::vector<Canvas> render_path(const std::vector<Camera>& cameras, int max_depth = 5) const;

 private:
  Device& dev_;
  rt355_scene* scene_ = nullptr;
};

// ---- camera.rs ------------------------------------------------------------------------------------
class Camera {
 public:
  Camera(float horizontal_size, float vertical_size, float field_of_view, Matrix4 transform = Matrix4())  // src/camera.rs:23-51
      : horizontal_size_(horizontal_size), vertical_size_(vertical_size), field_of_view_(field_of_view), transform_(transform) {
    const float half_view = std::tan(field_of_view / 2.0f);
    const float aspect_ratio = horizontal_size / vertical_size;
    half_width_ = aspect_ratio >= 1.0f ? half_view : half_view * aspect_ratio;
    half_height_ = aspect_ratio >= 1.0f ? half_view / aspect_ratio : half_view;
    pixel_size_ = (half_width_ * 2.0f) / horizontal_size;
  }
  static Camera new_default(float h, float v, float fov) { return Camera(h, v, fov, Matrix4::identity()); }
  float horizontal_size() const { return horizontal_size_; }
  float vertical_size() const { return vertical_size_; }
  float pixel_size() const { return pixel_size_; }
  float half_width() const { return half_width_; }
  float half_height() const { return half_height_; }
  rt355_camera_desc desc() const {
    rt355_camera_desc d{};
    d.hsize = horizontal_size_, d.vsize = vertical_size_, d.fov = field_of_view_;
    for (int k = 0; k < 16; ++k) d.transform[k] = transform_.m[k];
    return d;
  }
  // `Camera::render(&self, world: &World) -> Canvas` (src/camera.rs:74-88); the recursion depth is the
  // literal 5 of src/camera.rs:79.  Uploads the world, renders, releases it.
  Canvas render(const World& world, Device& device) const { return Scene(device, world).render(*this, 5); }
  Canvas render(const World& world) const {
    static Device device(0, 64);
    return render(world, device);
  }

 private:
  float horizontal_size_, vertical_size_, field_of_view_;
  Matrix4 transform_;
  float half_width_, half_height_, pixel_size_;
};

inline Canvas Scene::render(const Camera& camera, int max_depth, rt355_stats* stats) const {
  const rt355_camera_desc cd = camera.desc();
  Canvas c;
  c.width = (size_t)camera.horizontal_size(), c.height = (size_t)camera.vertical_size();
  std::vector<float> rgb(c.width * c.height * 3);
  check(rt355_render(dev_.handle(), scene_, &cd, max_depth, rgb.data(), stats));
  c.pixels.resize(c.width * c.height);
  for (size_t i = 0; i < c.pixels.size(); ++i) c.pixels[i] = Color::rgb(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
  return c;
}

inline std::vector<Canvas> Scene::render_path(const std::vector<Camera>& cameras, int max_depth) const {
  std::vector<Canvas> out;
  for (size_t first = 0; first < cameras.size(); first += RT355_MAX_BATCH) {
    const size_t n = std::min<size_t>(RT355_MAX_BATCH, cameras.size() - first);
    std::vector<rt355_camera_desc> cds;
    for (size_t f = 0; f < n; ++f) cds.push_back(cameras[first + f].desc());
    const size_t w = (size_t)cameras[first].horizontal_size(), h = (size_t)cameras[first].vertical_size();
    std::vector<float> rgb(n * w * h * 3);
    check(rt355_render_batch(dev_.handle(), scene_, cds.data(), (int)n, max_depth, rgb.data(), nullptr));
    for (size_t f = 0; f < n; ++f) {
      Canvas c;
      c.width = w, c.height = h;
      c.pixels.resize(w * h);
      const float* px = rgb.data() + f * w * h * 3;
      for (size_t i = 0; i < w * h; ++i) c.pixels[i] = Color::rgb(px[3 * i], px[3 * i + 1], px[3 * i + 2]);
      out.push_back(std::move(c));
    }
  }
  return out;
}

}  // namespace raytracer

#endif  // RT355_HPP
