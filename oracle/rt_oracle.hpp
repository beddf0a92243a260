// rt_oracle.hpp -- CPU restatement of the reference's render path.  TEST INFRASTRUCTURE ONLY.
//
// This header is the parity oracle for the HIP backend.  Nothing in the product (raytracer_amd/,
// include/) may include, link or call it; only tests/, the smoke check and the cpu_baseline leg of
// bench.py do, and only as the checker.
//
// It follows the reference file by file and keeps its algorithmic shape on purpose (brute force
// over every shape, a flattened + stably sorted intersection list, recursive shading), so that it
// is an independent check on the device's BVH / wavefront design:
//   camera.rs:23-88  world.rs:35-127  intersection.rs:60-154  computation.rs:20-33
//   shape.rs:111-514 material.rs:108-162 pattern.rs:93-121 ray.rs:22-37 tuple.rs:18-60
//   color.rs:97-111  epsilon.rs:1  triangle.rs:65-85 smooth_triangle.rs:67-87
// The reference cannot be built here (no cargo/rustc), so the oracle is pinned by the reference's
// own known-answer unit tests, restated as data in tests/test_oracle_kat.py.
//
// Deviations from the reference, all in places where the reference panics:
//   * NaN `t` values are dropped before the sort (world.rs:42 would panic in partial_cmp().unwrap()).
//   * A smooth triangle takes the u,v of the hit itself; `strict_uv` restores the reference's
//     "u,v of xs[0]" (intersection.rs:67-71, shape.rs:199-203) and reports where it would panic.
//   * The inverse of each transformation is computed once per shape instead of per ray
//     (ray.rs:34-35); same values, fewer repeats.
//
// Templated on the scalar type: float is the reference's precision (epsilon.rs:1 is f32), double is
// the configuration the device is graded in.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/rt355.h"

namespace rto {

template <class R>
struct V4 {
  R x, y, z, w;
};
template <class R>
inline V4<R> point(R x, R y, R z) { return {x, y, z, R(1)}; }  // tuple.rs:18-20
template <class R>
inline V4<R> vec(R x, R y, R z) { return {x, y, z, R(0)}; }  // tuple.rs:22-24
template <class R>
inline V4<R> operator+(V4<R> a, V4<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
template <class R>
inline V4<R> operator-(V4<R> a, V4<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
template <class R>
inline V4<R> operator*(V4<R> a, R k) { return {a.x * k, a.y * k, a.z * k, a.w * k}; }
template <class R>
inline V4<R> operator*(R k, V4<R> a) { return {a.x * k, a.y * k, a.z * k, a.w * k}; }
// nalgebra Vector4::dot: all four lanes.
template <class R>
inline R dot(V4<R> a, V4<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
template <class R>
inline R magnitude(V4<R> a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w); }  // tuple.rs:34-36
// nalgebra normalize: no zero guard, a zero vector gives NaN.
template <class R>
inline V4<R> normalize(V4<R> a) {
  R n = std::sqrt(dot(a, a));
  return {a.x / n, a.y / n, a.z / n, a.w / n};
}
template <class R>
inline V4<R> cross4(V4<R> a, V4<R> b) {  // tuple.rs:42-48
  return vec<R>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <class R>
inline V4<R> reflect(V4<R> v, V4<R> n) {  // tuple.rs:38-40
  return v - n * R(2) * dot(v, n);
}

// Column-major 4x4 like nalgebra's Matrix4 storage.
template <class R>
struct M4 {
  R m[16];
  R at(int r, int c) const { return m[c * 4 + r]; }
};
template <class R>
inline V4<R> operator*(const M4<R>& a, V4<R> v) {
  return {a.at(0, 0) * v.x + a.at(0, 1) * v.y + a.at(0, 2) * v.z + a.at(0, 3) * v.w,
          a.at(1, 0) * v.x + a.at(1, 1) * v.y + a.at(1, 2) * v.z + a.at(1, 3) * v.w,
          a.at(2, 0) * v.x + a.at(2, 1) * v.y + a.at(2, 2) * v.z + a.at(2, 3) * v.w,
          a.at(3, 0) * v.x + a.at(3, 1) * v.y + a.at(3, 2) * v.z + a.at(3, 3) * v.w};
}
template <class R>
inline M4<R> transpose(const M4<R>& a) {
  M4<R> t;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) t.m[c * 4 + r] = a.m[r * 4 + c];
  return t;
}
// General 4x4 inverse by cofactors (what Matrix4::try_inverse does for this size); returns false
// when the determinant is zero.  nalgebra itself is not vendored in the reference tree; the
// reference tests that pin this boundary are camera.rs:138-153, sphere.rs:219-255, ray.rs:96-110.
template <class R>
inline bool inverse(const M4<R>& a, M4<R>& out) {
  const R* m = a.m;
  R inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  R det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  if (det == R(0)) return false;
  R id = R(1) / det;
  for (int i = 0; i < 16; ++i) out.m[i] = inv[i] * id;
  return true;
}

template <class R>
struct Eps {
  static constexpr R value = R(1.0e-3);  // epsilon.rs:1
};

// approx::relative_eq!(x, 0., epsilon = EPSILON) as used at shape.rs:252,295,296,331,355,484,498
// (approx 0.4: equal -> true; either infinite -> false; |x| <= epsilon -> true; otherwise a relative
// test against max_relative = machine epsilon, which cannot pass against 0).  NaN -> false.
template <class R>
inline bool approx_zero(R x) {
  if (x == R(0)) return true;
  if (std::isinf(x)) return false;
  return std::fabs(x) <= Eps<R>::value;  // false for NaN
}

template <class R>
struct Color {
  R r, g, b;
};
template <class R>
inline Color<R> operator+(Color<R> a, Color<R> b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
template <class R>
inline Color<R> operator-(Color<R> a, Color<R> b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
template <class R>
inline Color<R> operator*(Color<R> a, R k) { return {a.r * k, a.g * k, a.b * k}; }
template <class R>
inline Color<R> hadamard(Color<R> a, Color<R> b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }

// color.rs:97-111.  f32::round is half-away-from-zero; `c as u8` of NaN is 0 in Rust.
template <class R>
inline uint8_t to_u8(R c) {
  R v = std::round(c * R(255));
  if (v < R(0)) return 0;
  if (v > R(255)) return 255;
  if (std::isnan(v)) return 0;
  return (uint8_t)v;
}

template <class R>
struct Material {
  Color<R> color;
  R ambient, diffuse, specular, shininess, reflective, refractive, transparency;
  int pattern_kind;
  Color<R> c1, c2;
};

template <class R>
struct Light {
  V4<R> position;
  Color<R> intensity;
};

template <class R>
struct Ray {
  V4<R> origin, direction;
  V4<R> position(R t) const { return origin + direction * t; }  // ray.rs:22-24
};

template <class R>
struct Shape {
  int kind;
  M4<R> transformation;
  M4<R> inv;  // transformation.try_inverse().unwrap(), cached
  Material<R> material;
  R min, max;
  bool closed;
  V4<R> p1, p2, p3, n1, n2, n3, e1, e2, normal;
};

template <class R>
struct Intersection {
  R t;
  int shape;  // index into World.shapes: stands in for the ULID id (shape.rs:22-36)
  bool has_uv;
  R u, v;
};

template <class R>
struct Computations {  // computation.rs:4-17
  R t;
  int shape;
  V4<R> point, under_point, over_point, eyev, normalv, reflectv;
  bool inside;
  R n1, n2;
  bool would_panic;  // strict_uv only: xs[0] carried no u,v
};

struct Counters {
  uint64_t radiance[RT355_MAX_GEN] = {0};
  uint64_t hits[RT355_MAX_GEN] = {0};
  uint64_t panics = 0;
  void add(const Counters& o) {
    for (int i = 0; i < RT355_MAX_GEN; ++i) radiance[i] += o.radiance[i], hits[i] += o.hits[i];
    panics += o.panics;
  }
};

template <class R>
inline Material<R> make_material(const rt355_material& d) {
  Material<R> m;
  m.color = {R(d.color[0]), R(d.color[1]), R(d.color[2])};
  m.ambient = R(d.ambient), m.diffuse = R(d.diffuse), m.specular = R(d.specular);
  m.shininess = R(d.shininess), m.reflective = R(d.reflective), m.refractive = R(d.refractive);
  m.transparency = R(d.transparency);
  m.pattern_kind = d.pattern_kind;
  m.c1 = {R(d.c1[0]), R(d.c1[1]), R(d.c1[2])};
  m.c2 = {R(d.c2[0]), R(d.c2[1]), R(d.c2[2])};
  return m;
}

template <class R>
inline bool make_shape(const rt355_shape_desc& d, Shape<R>& s) {
  s.kind = d.kind;
  for (int i = 0; i < 16; ++i) s.transformation.m[i] = R(d.m[i]);
  if (!inverse(s.transformation, s.inv)) return false;
  s.material = make_material<R>(d.mat);
  s.min = R(d.min), s.max = R(d.max), s.closed = d.closed != 0;
  s.p1 = point<R>(d.p1[0], d.p1[1], d.p1[2]);
  s.p2 = point<R>(d.p2[0], d.p2[1], d.p2[2]);
  s.p3 = point<R>(d.p3[0], d.p3[1], d.p3[2]);
  s.n1 = vec<R>(d.n1[0], d.n1[1], d.n1[2]);
  s.n2 = vec<R>(d.n2[0], d.n2[1], d.n2[2]);
  s.n3 = vec<R>(d.n3[0], d.n3[1], d.n3[2]);
  // triangle.rs:65-66,83-85: the stored edges are p2-p1 and p3-p1, but the flat normal is built
  // from the LOCAL e2 = p3-p2 and e1 = p2-p1.
  V4<R> le1 = s.p2 - s.p1, le2 = s.p3 - s.p2;
  s.e1 = s.p2 - s.p1;
  s.e2 = s.p3 - s.p1;
  s.normal = normalize(cross4(le2, le1));
  return true;
}

template <class R>
struct World {
  Light<R> light;
  std::vector<Shape<R>> shapes;
  bool strict_uv = false;

  // ---- shape.rs:455-514 helpers -------------------------------------------------------------
  static void check_cube_axis(R origin, R direction, R& t0, R& t1) {
    if (direction >= R(0)) {
      t0 = (R(-1) - origin) / direction;
      t1 = (R(1) - origin) / direction;
    } else {
      t1 = (R(-1) - origin) / direction;
      t0 = (R(1) - origin) / direction;
    }
  }
  static bool check_cylinder_cap(const Ray<R>& ray, R t) {
    R x = ray.origin.x + t * ray.direction.x;
    R z = ray.origin.z + t * ray.direction.z;
    return x * x + z * z <= R(1) + Eps<R>::value;
  }
  static bool check_cone_cap(const Ray<R>& ray, R t, R r) {
    R x = ray.origin.x + t * ray.direction.x;
    R z = ray.origin.z + t * ray.direction.z;
    return x * x + z * z <= r * r + Eps<R>::value;
  }
  void intersect_caps(int si, const Ray<R>& ray, std::vector<Intersection<R>>& out) const {
    const Shape<R>& s = shapes[si];
    if (!s.closed || approx_zero(ray.direction.y)) return;
    R t_lower = (s.min - ray.origin.y) / ray.direction.y;
    R t_upper = (s.max - ray.origin.y) / ray.direction.y;
    if (s.kind == RT355_CYLINDER) {
      if (check_cylinder_cap(ray, t_lower)) out.push_back({t_lower, si, false, 0, 0});
      if (check_cylinder_cap(ray, t_upper)) out.push_back({t_upper, si, false, 0, 0});
    } else {
      if (check_cone_cap(ray, t_lower, s.min)) out.push_back({t_lower, si, false, 0, 0});
      if (check_cone_cap(ray, t_upper, s.max)) out.push_back({t_upper, si, false, 0, 0});
    }
  }

  // ---- shape.rs:208-378 -----------------------------------------------------------------------
  void local_intersect(int si, const Ray<R>& ray, std::vector<Intersection<R>>& out) const {
    const Shape<R>& s = shapes[si];
    const R EPS = Eps<R>::value;
    switch (s.kind) {
      case RT355_PLANE: {
        if (std::fabs(ray.direction.y) < EPS) return;
        out.push_back({-ray.origin.y / ray.direction.y, si, false, 0, 0});
        return;
      }
      case RT355_SPHERE: {
        V4<R> sphere_to_ray = ray.origin - point<R>(0, 0, 0);
        R a = dot(ray.direction, ray.direction);
        R b = R(2) * dot(ray.direction, sphere_to_ray);
        R c = dot(sphere_to_ray, sphere_to_ray) - R(1);
        R disc = b * b - R(4) * a * c;
        if (disc < R(0)) return;
        R d = std::sqrt(disc);
        out.push_back({(-b - d) / (R(2) * a), si, false, 0, 0});
        out.push_back({(-b + d) / (R(2) * a), si, false, 0, 0});
        return;
      }
      case RT355_CUBE: {
        R xmin, xmax, ymin, ymax, zmin, zmax;
        check_cube_axis(ray.origin.x, ray.direction.x, xmin, xmax);
        check_cube_axis(ray.origin.y, ray.direction.y, ymin, ymax);
        check_cube_axis(ray.origin.z, ray.direction.z, zmin, zmax);
        // f32::max / f32::min ignore a NaN operand, like fmax / fmin.
        R tmin = std::fmax(std::fmax(xmin, ymin), zmin);
        R tmax = std::fmin(std::fmin(xmax, ymax), zmax);
        if (tmin > tmax) return;
        out.push_back({tmin, si, false, 0, 0});
        out.push_back({tmax, si, false, 0, 0});
        return;
      }
      case RT355_CYLINDER: {
        R a = ray.direction.x * ray.direction.x + ray.direction.z * ray.direction.z;
        if (approx_zero(a)) {
          intersect_caps(si, ray, out);
          return;
        }
        R b = R(2) * ray.origin.x * ray.direction.x + R(2) * ray.origin.z * ray.direction.z;
        R c = ray.origin.x * ray.origin.x + ray.origin.z * ray.origin.z - R(1);
        R d = b * b - R(4) * a * c;
        if (d < R(0)) return;
        R t0 = (-b - std::sqrt(d)) / (R(2) * a);
        R t1 = (-b + std::sqrt(d)) / (R(2) * a);
        if (t0 > t1) std::swap(t0, t1);
        R y0 = ray.origin.y + t0 * ray.direction.y;
        if (s.min < y0 && y0 < s.max) out.push_back({t0, si, false, 0, 0});
        R y1 = ray.origin.y + t1 * ray.direction.y;
        if (s.min < y1 && y1 < s.max) out.push_back({t1, si, false, 0, 0});
        intersect_caps(si, ray, out);
        return;
      }
      case RT355_CONE: {
        R a = ray.direction.x * ray.direction.x - ray.direction.y * ray.direction.y + ray.direction.z * ray.direction.z;
        R b = R(2) * (ray.origin.x * ray.direction.x - ray.origin.y * ray.direction.y + ray.origin.z * ray.direction.z);
        R c = ray.origin.x * ray.origin.x - ray.origin.y * ray.origin.y + ray.origin.z * ray.origin.z;
        if (approx_zero(a) && !approx_zero(b)) {
          out.push_back({c / (R(-2) * b), si, false, 0, 0});
        } else {
          // shape.rs:300: the discriminant is rounded to two decimals before the sign test.
          R d = std::round((b * b - R(4) * a * c) * R(100)) / R(100);
          if (d < R(0)) return;
          R t0 = (-b - std::sqrt(d)) / (R(2) * a);
          R t1 = (-b + std::sqrt(d)) / (R(2) * a);
          if (t0 > t1) std::swap(t0, t1);
          R y0 = ray.origin.y + t0 * ray.direction.y;
          if (s.min < y0 && y0 < s.max) out.push_back({t0, si, false, 0, 0});
          R y1 = ray.origin.y + t1 * ray.direction.y;
          if (s.min < y1 && y1 < s.max) out.push_back({t1, si, false, 0, 0});
        }
        intersect_caps(si, ray, out);
        return;
      }
      case RT355_TRIANGLE:
      case RT355_SMOOTH_TRIANGLE: {
        V4<R> dir_cross_e2 = cross4(ray.direction, s.e2);
        R det = dot(s.e1, dir_cross_e2);
        if (approx_zero(std::fabs(det))) return;
        R f = R(1) / det;
        V4<R> p1_to_origin = ray.origin - s.p1;
        R u = f * dot(p1_to_origin, dir_cross_e2);
        if (u < R(0) || u > R(1)) return;
        V4<R> origin_cross_e1 = cross4(p1_to_origin, s.e1);
        R v = f * dot(ray.direction, origin_cross_e1);
        if (v < R(0) || (u + v) > R(1)) return;
        R t = f * dot(s.e2, origin_cross_e1);
        if (s.kind == RT355_SMOOTH_TRIANGLE)
          out.push_back({t, si, true, u, v});
        else
          out.push_back({t, si, false, 0, 0});
        return;
      }
      default:  // Group: shape.rs:376
        return;
    }
  }

  // shape.rs:380-415 with ray.rs:32-37
  void shape_intersect(int si, const Ray<R>& ray, std::vector<Intersection<R>>& out) const {
    const Shape<R>& s = shapes[si];
    Ray<R> inv_ray{s.inv * ray.origin, s.inv * ray.direction};
    local_intersect(si, inv_ray, out);
  }

  // world.rs:35-44
  std::vector<Intersection<R>> intersect(const Ray<R>& ray) const {
    std::vector<Intersection<R>> xs;
    for (int i = 0; i < (int)shapes.size(); ++i) shape_intersect(i, ray, xs);
    xs.erase(std::remove_if(xs.begin(), xs.end(), [](const Intersection<R>& i) { return std::isnan(i.t); }), xs.end());
    std::stable_sort(xs.begin(), xs.end(), [](const Intersection<R>& a, const Intersection<R>& b) { return a.t < b.t; });
    return xs;
  }

  // intersection.rs:149-154: filter(t >= 0).min(); Iterator::min keeps the FIRST of equal minima.
  static int hit(const std::vector<Intersection<R>>& xs) {
    int best = -1;
    for (int i = 0; i < (int)xs.size(); ++i) {
      if (!(xs[i].t >= R(0))) continue;
      if (best < 0 || xs[i].t < xs[best].t) best = i;
    }
    return best;
  }

  // shape.rs:138-206
  V4<R> local_normal(int si, V4<R> lp, const Intersection<R>* hit, bool* panic) const {
    const Shape<R>& s = shapes[si];
    const R EPS = Eps<R>::value;
    switch (s.kind) {
      case RT355_PLANE:
        return vec<R>(0, 1, 0);
      case RT355_SPHERE: {
        V4<R> n = lp - vec<R>(0, 0, 0);
        V4<R> nw = transpose(s.inv) * n;
        return normalize(vec<R>(nw.x, nw.y, nw.z));
      }
      case RT355_CUBE: {
        R maxc = std::fmax(std::fmax(std::fabs(lp.x), std::fabs(lp.y)), std::fabs(lp.z));
        if (maxc == std::fabs(lp.x)) return vec<R>(lp.x, 0, 0);
        if (maxc == std::fabs(lp.y)) return vec<R>(0, lp.y, 0);
        return vec<R>(0, 0, lp.z);
      }
      case RT355_CYLINDER: {
        R dist = lp.x * lp.x + lp.z * lp.z;
        if (dist < R(1) && lp.y >= (s.max - EPS)) return vec<R>(0, 1, 0);
        if (dist < R(1) && lp.y <= (s.min + EPS)) return vec<R>(0, -1, 0);
        return vec<R>(lp.x, 0, lp.z);
      }
      case RT355_CONE: {
        R dist = lp.x * lp.x + lp.z * lp.z;
        if (dist < s.max * s.max && lp.y >= s.max - EPS) return vec<R>(0, 1, 0);
        if (dist < s.min * s.min && lp.y <= s.min + EPS) return vec<R>(0, -1, 0);
        return vec<R>(lp.x, lp.y > R(0) ? -std::sqrt(dist) : std::sqrt(dist), lp.z);
      }
      case RT355_TRIANGLE:
        return s.normal;
      case RT355_SMOOTH_TRIANGLE: {
        if (!hit || !hit->has_uv) {
          if (panic) *panic = true;  // hit.unwrap().u.unwrap() panics
          return vec<R>(std::numeric_limits<R>::quiet_NaN(), std::numeric_limits<R>::quiet_NaN(), std::numeric_limits<R>::quiet_NaN());
        }
        return normalize(s.n2 * hit->u + s.n3 * hit->v + s.n1 * (R(1) - hit->u - hit->v));
      }
      default:
        return vec<R>(0, 0, 0);
    }
  }
  // shape.rs:111-136
  V4<R> normal(int si, V4<R> p, const Intersection<R>* hit, bool* panic) const {
    if (shapes[si].kind == RT355_GROUP) return vec<R>(0, 0, 0);
    return local_normal(si, shapes[si].inv * p, hit, panic);
  }

  // intersection.rs:60-142.  `self_` is the intersection being shaded; `all` the sorted list or
  // nullptr (the reference's Option::None).
  Computations<R> prepare_computation(const Intersection<R>& self_, const Ray<R>& ray,
                                      const std::vector<Intersection<R>>* all) const {
    Computations<R> c;
    c.would_panic = false;
    const R EPS = Eps<R>::value;
    V4<R> p = ray.position(self_.t);
    V4<R> eyev = R(-1) * ray.direction;
    const Intersection<R>* nh = nullptr;
    if (strict_uv) {
      if (all && !all->empty()) nh = &(*all)[0];
    } else {
      nh = &self_;
    }
    V4<R> normalv_temp = normal(self_.shape, p, nh, &c.would_panic);
    R angle = dot(normalv_temp, eyev);
    bool inside = angle < R(0);
    V4<R> normalv = inside ? R(-1) * normalv_temp : R(1) * normalv_temp;
    V4<R> offset = EPS * normalv;
    c.over_point = p + offset;
    c.under_point = p - offset;
    c.reflectv = reflect(ray.direction, normalv_temp);

    std::vector<int> container;
    R n1 = R(1), n2 = R(1);
    std::vector<Intersection<R>> one{self_};
    const std::vector<Intersection<R>>& xs = all ? *all : one;
    for (const Intersection<R>& x : xs) {
      bool is_self = (x.t == self_.t) && (x.shape == self_.shape);  // intersection.rs:17-21
      if (is_self) n1 = container.empty() ? R(1) : shapes[container.back()].material.refractive;
      auto it = std::find(container.begin(), container.end(), x.shape);
      if (it != container.end())
        container.erase(std::remove(container.begin(), container.end(), x.shape), container.end());
      else
        container.push_back(x.shape);
      if (is_self) n2 = container.empty() ? R(1) : shapes[container.back()].material.refractive;
    }
    c.t = self_.t, c.shape = self_.shape, c.point = p, c.eyev = eyev, c.normalv = normalv;
    c.inside = inside, c.n1 = n1, c.n2 = n2;
    return c;
  }

  // computation.rs:20-33
  static R schlick(const Computations<R>& c) {
    R cs = dot(c.eyev, c.normalv);
    if (c.n1 > c.n2) {
      R n = c.n1 / c.n2;
      R sin2_t = n * n * (R(1) - cs * cs);
      if (sin2_t > R(1)) return R(1);
      cs = std::sqrt(R(1) - sin2_t);
    }
    R r0 = (c.n1 - c.n2) / (c.n1 + c.n2);
    r0 = r0 * r0;
    R k = R(1) - cs;
    return r0 + ((R(1) - r0) * (k * k * k * k * k));
  }

  // pattern.rs:93-121; `%` is Rust's truncated remainder = fmod.
  static Color<R> pattern_at(const Material<R>& m, V4<R> p) {
    switch (m.pattern_kind) {
      case RT355_PATTERN_STRIPE:
        return std::fmod(std::floor(p.x), R(2)) == R(0) ? m.c1 : m.c2;
      case RT355_PATTERN_GRADIENT: {
        R f = p.x - std::floor(p.x);
        Color<R> d = m.c2 - m.c1;
        return {m.c1.r + f * d.r, m.c1.g + f * d.g, m.c1.b + f * d.b};
      }
      case RT355_PATTERN_RING: {
        R d = std::floor(std::fmod(std::sqrt(p.x * p.x + p.z * p.z), R(2)));
        return d == R(0) ? m.c1 : m.c2;
      }
      case RT355_PATTERN_CHECKER: {
        R d = std::fmod(std::floor(p.x) + std::floor(p.y) + std::floor(p.z), R(2));
        return d == R(0) ? m.c1 : m.c2;
      }
      case RT355_PATTERN_TEST:
        return {p.x, p.y, p.z};
      default:
        return m.color;
    }
  }

  // material.rs:108-162
  static Color<R> lighting(const Material<R>& m, const Light<R>& light, V4<R> p, V4<R> eyev,
                           V4<R> normalv, bool is_shadow) {
    Color<R> c = m.pattern_kind != RT355_PATTERN_NONE ? pattern_at(m, p) : m.color;
    Color<R> eff = hadamard(c, light.intensity);
    Color<R> ambient = eff * m.ambient;
    if (is_shadow) return ambient;
    V4<R> lightv = normalize(light.position - p);
    R ldn = dot(lightv, normalv);
    Color<R> diffuse{0, 0, 0}, specular{0, 0, 0};
    if (ldn >= R(0)) {
      R k = m.diffuse * ldn;
      diffuse = eff * k;
      V4<R> reflectv = reflect(R(-1) * lightv, normalv);
      R rde = dot(reflectv, eyev);
      if (rde > R(0)) {
        R factor = std::pow(rde, m.shininess);
        R fs = factor * m.specular;
        specular = light.intensity * fs;
      }
    }
    return ambient + diffuse + specular;
  }

  // world.rs:45-57
  bool is_shadowed(V4<R> p) const {
    V4<R> lv = light.position - p;
    R distance = magnitude(lv);
    V4<R> direction = normalize(lv);
    Ray<R> ray{p, direction};
    auto xs = intersect(ray);
    int h = hit(xs);
    return h >= 0 && xs[h].t < distance + Eps<R>::value;
  }

  // world.rs:58-66
  Color<R> color_at(const Ray<R>& ray, int remaining, int depth, Counters* cnt) const {
    if (cnt) cnt->radiance[depth - remaining]++;
    auto xs = intersect(ray);
    int h = hit(xs);
    if (h < 0) return {0, 0, 0};
    if (cnt) cnt->hits[depth - remaining]++;
    Computations<R> c = prepare_computation(xs[h], ray, &xs);
    if (c.would_panic && cnt) cnt->panics++;
    return shade_hit(c, remaining, depth, cnt);
  }
  // world.rs:68-76
  Color<R> reflected_color(const Computations<R>& c, int remaining, int depth, Counters* cnt) const {
    const Material<R>& m = shapes[c.shape].material;
    if (remaining <= 0 || m.reflective == R(0)) return {0, 0, 0};
    Ray<R> r{c.over_point, c.reflectv};
    return color_at(r, remaining - 1, depth, cnt) * m.reflective;
  }
  // world.rs:78-102
  Color<R> refracted_color(const Computations<R>& c, int remaining, int depth, Counters* cnt) const {
    const Material<R>& m = shapes[c.shape].material;
    if (remaining <= 0 || m.transparency == R(0)) return {0, 0, 0};
    R n_ratio = c.n1 / c.n2;
    R cos_ti = dot(c.eyev, c.normalv);
    R sin2_t = n_ratio * n_ratio * (R(1) - cos_ti * cos_ti);
    if (sin2_t > R(1)) return {0, 0, 0};
    R cos_t = std::sqrt(R(1) - sin2_t);
    V4<R> direction = (n_ratio * cos_ti - cos_t) * c.normalv - n_ratio * c.eyev;
    Ray<R> r{c.under_point, direction};
    return color_at(r, remaining - 1, depth, cnt) * m.transparency;
  }
  // world.rs:104-127
  Color<R> shade_hit(const Computations<R>& c, int remaining, int depth, Counters* cnt) const {
    bool shadowed = is_shadowed(c.over_point);
    const Material<R>& m = shapes[c.shape].material;
    Color<R> surface = lighting(m, light, c.over_point, c.eyev, c.normalv, shadowed);
    Color<R> refl = reflected_color(c, remaining, depth, cnt);
    Color<R> refr = refracted_color(c, remaining, depth, cnt);
    if (m.reflective > R(0) && m.transparency > R(0)) {
      R reflectance = schlick(c);
      return surface + refl * reflectance + refr * (R(1) - reflectance);
    }
    return surface + refr + refl;
  }
};

template <class R>
struct Camera {  // camera.rs:11-51
  R hsize, vsize, fov, half_width, half_height, pixel_size;
  M4<R> transform, inv;
  bool ok;
  explicit Camera(const rt355_camera_desc& d) {
    hsize = R(d.hsize), vsize = R(d.vsize), fov = R(d.fov);
    for (int i = 0; i < 16; ++i) transform.m[i] = R(d.transform[i]);
    R half_view = std::tan(fov / R(2));
    R aspect = hsize / vsize;
    half_width = aspect >= R(1) ? half_view : half_view * aspect;
    half_height = aspect >= R(1) ? half_view / aspect : half_view;
    pixel_size = (half_width * R(2)) / hsize;
    ok = inverse(transform, inv);
  }
  Ray<R> ray_for_pixel(R px, R py) const {  // camera.rs:62-72
    R x_offset = (px + R(0.5)) * pixel_size;
    R y_offset = (py + R(0.5)) * pixel_size;
    R world_x = half_width - x_offset;
    R world_y = half_height - y_offset;
    V4<R> pixel = inv * point<R>(world_x, world_y, R(-1));
    V4<R> origin = inv * point<R>(0, 0, 0);
    return {origin, normalize(pixel - origin)};
  }
};

}  // namespace rto
