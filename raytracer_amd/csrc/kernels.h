// kernels.h -- HIP kernels of the wavefront render pipeline (gfx950, wave64).
//
// The reference renders a pixel by recursion: Camera::render -> World::color_at -> shade_hit ->
// is_shadowed / reflected_color / refracted_color -> color_at ... (src/camera.rs:74-88,
// src/world.rs:58-127).  Here the recursion is unrolled into GENERATIONS of rays (generation g holds the
// rays whose `remaining` is max_depth - g) and each generation runs data-parallel kernels:
//
//   k_trace      World::intersect + Intersections::hit     one thread per ray, BVH traversal
//   k_scan       (no counterpart)                          child-ray slots of the generation: prefix sums
//   k_spawn      n1 / n2 + reflect / refract / schlick     writes the rays of generation g+1
//   k_shadow     is_shadowed + Material::lighting          off the critical path: second stream
//   k_combine    shade_hit's final sum                     run from the deepest generation back to 0
//
// Every hit keeps its own record (the two child ids, the mix weight, the shadow flag), so k_combine
// evaluates exactly the reference's expression  surface + reflected*R + refracted*(1-R)  (or
// surface + refracted + reflected) per hit, in the reference's order, with no atomics on colours: the
// image does not depend on queue order and is bit-identical across tile splits.
//
// Arithmetic follows the reference expression by expression (file:line cited at each function),
// templated on the scalar type R (double = graded configuration, float = the reference's precision).
#pragma once
#include <hip/hip_runtime.h>

#include <limits>

#include "rt355_device.h"

namespace rt355 {

// ------------------------------------------------------------------------------------------------
// small vector helpers
// ------------------------------------------------------------------------------------------------
template <class R>
struct V3 {
  R x, y, z;
};
#define RT_FN __device__ __forceinline__
template <class R> RT_FN V3<R> mk(R x, R y, R z) { return V3<R>{x, y, z}; }
template <class R> RT_FN V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> RT_FN V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> RT_FN V3<R> operator*(V3<R> a, R k) { return {a.x * k, a.y * k, a.z * k}; }
template <class R> RT_FN V3<R> operator*(R k, V3<R> a) { return {a.x * k, a.y * k, a.z * k}; }
template <class R> RT_FN R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class R> RT_FN V3<R> cross(V3<R> a, V3<R> b) {  // src/tuple.rs:42-48
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// 1 / sqrt(x).  In double: the hardware's estimate (v_rsq_f64, some 26 bits) and two Newton steps, good to about an
// ulp like the square root followed by the division it stands for, at a dozen instructions instead of some fifty.
// Zero gives NaN (infinity times zero in the first step), which is what the callers' zero vectors are meant to become.
template <class R> RT_FN R inv_sqrt(R x) { return R(1) / sqrt(x); }
template <> RT_FN double inv_sqrt<double>(double x) {
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double e = __builtin_fma(-(x * y), y, 1.0);  // 1 - x y^2
    y = __builtin_fma(0.5 * y, e, y);
  }
  return y;
}
// nalgebra's normalize (no zero guard: a zero vector gives NaN).  The three quotients by the common norm are formed
// with one reciprocal (within an ulp of the divisions).
template <class R> RT_FN V3<R> normalize(V3<R> a) {
  const R inv = inv_sqrt(dot(a, a));
  return {a.x * inv, a.y * inv, a.z * inv};
}
template <class R> RT_FN V3<R> reflect(V3<R> v, V3<R> n) {  // src/tuple.rs:38-40
  return v - n * R(2) * dot(v, n);
}
// A record read in full, 16 bytes per request, all requests issued before the first value is used.  The empty asm pins
// the values here: without it the compiler moves each field's load down to the block that uses it.
template <class T>
RT_FN T load_whole(const T* p) {
  static_assert(sizeof(T) % 8 == 0, "records are read in 8-byte words");
  constexpr int N = sizeof(T) / 8;
  union U { T t; unsigned long long w[N]; RT_FN U() {} } u;
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
#pragma unroll
  for (int k = 0; k < N; ++k) u.w[k] = q[k];
#pragma unroll
  for (int k = 0; k < N; ++k) asm volatile("" : "+v"(u.w[k]));
  return u.t;
}

template <class R> RT_FN V3<R> ld3(const R* p) { return {p[0], p[1], p[2]}; }
template <class R> RT_FN void st3(R* p, V3<R> v) { p[0] = v.x, p[1] = v.y, p[2] = v.z; }

template <class R> RT_FN R eps() { return R(1.0e-3); }  // src/epsilon.rs:1
template <class R> RT_FN R inf() { return R(INFINITY); }

// approx::relative_eq!(x, 0., epsilon = EPSILON) (call sites src/shape.rs:252,295,296,331,355,484,498):
// true for |x| <= EPSILON, false for infinities and NaN.
template <class R> RT_FN bool approx_zero(R x) { return fabs(x) <= eps<R>(); }

// rows 0..2 of an affine inverse applied to a point / a vector (src/ray.rs:32-37)
template <class R> RT_FN V3<R> xf_point(const R* r, V3<R> p) {
  return {r[0] * p.x + r[1] * p.y + r[2] * p.z + r[3], r[4] * p.x + r[5] * p.y + r[6] * p.z + r[7],
          r[8] * p.x + r[9] * p.y + r[10] * p.z + r[11]};
}
template <class R> RT_FN V3<R> xf_vec(const R* r, V3<R> v) {
  return {r[0] * v.x + r[1] * v.y + r[2] * v.z, r[4] * v.x + r[5] * v.y + r[6] * v.z,
          r[8] * v.x + r[9] * v.y + r[10] * v.z};
}

#ifdef RT355_EXP_STAMPS
struct DbgAcc {
  unsigned long long cyc_node = 0, cyc_tri = 0, cyc_prim = 0;
  uint32_t trips = 0, n_node = 0, n_leaf = 0;
};
#endif
// Every traversal function and kernel is compiled per MODE: whether it counts its work, and whether the scene holds
// instanced meshes (prims of kind RT355_INSTANCE with a tree of their own below them).  Scenes without instances --
// the common case -- run kernels that carry no trace of them.
constexpr int M_COUNT = 1, M_INST = 2;

struct Work {  // per-thread work counters (counting mode only)
  uint32_t box = 0, xf = 0, prim[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef RT355_EXP_STAMPS
  DbgAcc dbg;  // of the traversal in flight
#endif
};
#ifdef RT355_EXP_STAMPS
// Diagnostic build only: one record per (launch, block, wave), summed over the wave's trips: 100 MHz wall
// clock of its first traversal start and last traversal end, shader cycles spent in node steps, triangle
// leaves and other leaves, loop trips of the wave.
struct DbgRec {
  unsigned long long t0, t1;
  unsigned long long cyc_node, cyc_tri, cyc_prim, cyc_total;
  uint32_t trips, lanes, n_node, n_leaf;
};
constexpr uint32_t DBG_PER_LAUNCH = 1u << 15;
__device__ DbgRec g_dbg[RT355_MAX_LAUNCH * DBG_PER_LAUNCH];
RT_FN void dbg_record(int launch, unsigned long long t0, unsigned long long c0, const DbgAcc& a, bool live) {
  const unsigned long long t1 = wall_clock64(), c1 = __builtin_amdgcn_s_memtime();
  const uint32_t lanes = __popcll(__ballot(live));
  if ((threadIdx.x & 63u) == 0 && launch < RT355_MAX_LAUNCH) {
    const uint32_t slot = (blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) & (DBG_PER_LAUNCH - 1);
    DbgRec& r = g_dbg[(size_t)launch * DBG_PER_LAUNCH + slot];
    if (r.trips == 0) r.t0 = t0;
    r.t1 = t1;
    r.cyc_node += a.cyc_node, r.cyc_tri += a.cyc_tri, r.cyc_prim += a.cyc_prim, r.cyc_total += c1 - c0;
    r.trips += a.trips, r.lanes += lanes, r.n_node += a.n_node, r.n_leaf += a.n_leaf;
  }
}
#endif
// Counters of one thread -> FrameCtl.  Reduced over the wave first (every lane of the wave calls this at
// the end of its kernel), so the per-launch totals cost 11 atomics per wave, not per lane.
template <bool COUNTING> RT_FN void flush_work(FrameCtl* ctl, int launch, const Work& w) {
  if constexpr (COUNTING) {
    if (launch >= RT355_MAX_LAUNCH) return;
    auto wsum = [](uint32_t v) {
      unsigned long long t = v;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
      return t;
    };
    uint32_t mx = w.box;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, off));
    const unsigned long long box = wsum(w.box), xf = wsum(w.xf);
    unsigned long long prim[8];
    for (int k = 0; k < 8; ++k) prim[k] = wsum(w.prim[k]);
    if ((threadIdx.x & 63u) != 0) return;
    if (box) atomicAdd(&ctl->work[launch][0], box);
    if (mx) atomicMax(&ctl->work[launch][10], (unsigned long long)mx);  // longest traversal of one lane
    if (xf) atomicAdd(&ctl->work[launch][1], xf);
    for (int k = 0; k < 8; ++k)
      if (prim[k]) atomicAdd(&ctl->work[launch][2 + k], prim[k]);
  }
}

// A record every lane of the wave reads at the same address: through the scalar cache into scalar registers, one
// request per wave and a far shorter latency than 64 identical vector loads.
template <class T> using cptr = const __attribute__((address_space(4))) T*;
template <class T>
RT_FN T load_uniform(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "whole dwords");
  T out;
  cptr<uint32_t> q = (cptr<uint32_t>)(const uint32_t*)p;
  uint32_t* o = (uint32_t*)&out;
#pragma unroll
  for (size_t i = 0; i < sizeof(T) / 4; ++i) o[i] = q[i];
  return out;
}

// ------------------------------------------------------------------------------------------------
// Shape::intersect = local_intersect(ray.inv_transform(transformation))  (src/shape.rs:208-415)
// emit(t, u, v, emission_index) is called once per intersection, in the reference's emission order.
// ------------------------------------------------------------------------------------------------
// UNIFORM: p is the same in every lane of the wave (the every-ray list, the leaves of a packet traversal).
// NONNEG: the caller has no use for intersections with t < 0 (nearest hit and any hit ask for t >= 0,
// src/intersection.rs:151); where the sign of t is known before t is, such an intersection is not computed.
template <class R, int MODE, bool UNIFORM = false, bool NONNEG = false, class Emit>
RT_FN void prim_hits(const DevScene<R>& sc, int p, int flags, V3<R> wo, V3<R> wd, Work& wk, Emit&& emit) {
  const int kind = flags & PF_KIND_MASK;
  if constexpr (MODE & M_COUNT) wk.prim[kind]++;
  V3<R> o = wo, d = wd;
  R ext_min = R(0), ext_max = R(0);
  // the w term of inverse * direction, m[k][3] * 0.0 (src/ray.rs:35): a signed zero, -0.0 where the inverse's
  // translation entry is negative.  It decides nothing but the sign of a zero direction component, which the cube's
  // slab test tells apart (below).  The identity's entries are +0.0.
  V3<R> w_term = {R(0), R(0), R(0)};
  const bool instanced = (MODE & M_INST) != 0 && (flags & PF_INSTANCED) != 0;  // scenes without instances never look at inst_of
  if (!(flags & PF_IDENTITY) || instanced || kind == RT355_CYLINDER || kind == RT355_CONE) {
    // a triangle of an instanced mesh is the identity inside its mesh's tree; seen from the world (this function is
    // called with world-space rays: the hit shape's own intersections for n1 / n2) it carries its instance's transformation
    int xp = p;
    if constexpr ((MODE & M_INST) != 0) xp = instanced ? sc.inst_of[p] : p;
    DevXform<R> X;
    if constexpr (UNIFORM) X = load_uniform(&sc.xform[xp]); else X = sc.xform[xp];
    ext_min = X.ymin, ext_max = X.ymax;
    if (!(flags & PF_IDENTITY) || instanced) {
      o = xf_point(X.r, wo);
      d = xf_vec(X.r, wd);
      w_term = {X.r[3] * R(0), X.r[7] * R(0), X.r[11] * R(0)};
      if constexpr (MODE & M_COUNT) wk.xf++;
    }
  }
  const R EPS = eps<R>();
  if (kind >= RT355_TRIANGLE) {
    R t, u, v;
    if (tri_hit(sc.tri[p], o, d, t, u, v)) emit(t, u, v, 0);
    return;
  }
  switch (kind) {
    case RT355_PLANE: {  // src/shape.rs:210-217
      if (fabs(d.y) < EPS) return;
      // t = -o.y / d.y is negative exactly when o.y and d.y have the same sign and o.y is not zero (a zero o.y gives
      // t = +-0.0, which counts as t >= 0): a ray leaving the floor upwards is told so without the division, which
      // takes some thirty instructions in double
      if constexpr (NONNEG)
        if (o.y != R(0) && ((o.y < R(0)) == (d.y < R(0)))) return;
      emit(-o.y / d.y, R(0), R(0), 0);
      return;
    }
    case RT355_SPHERE: {  // src/shape.rs:218-233
      R a = dot(d, d);
      R b = R(2) * dot(d, o);
      R c = dot(o, o) - R(1);
      R disc = b * b - R(4) * a * c;
      if (disc < R(0)) return;
      R sq = sqrt(disc);
      emit((-b - sq) / (R(2) * a), R(0), R(0), 0);
      emit((-b + sq) / (R(2) * a), R(0), R(0), 1);
      return;
    }
    case RT355_CUBE: {  // src/shape.rs:234-248, 455-466
      // The branch `direction >= 0` is also taken by -0.0, whose quotients then come out with the signs
      // swapped: a ray with a zero direction component misses or hits a slab it lies inside depending on the
      // SIGN of that zero.  In the reference that sign is whatever the 4-term row sums of inverse * direction
      // leave; the rows here have three terms (none for the identity), so the fourth, a signed zero, is added now:
      // -0.0 survives only when all four terms are -0.0, as in the reference.
      d = d + w_term;
      R t0x, t1x, t0y, t1y, t0z, t1z;
      if (d.x >= R(0)) { t0x = (R(-1) - o.x) / d.x; t1x = (R(1) - o.x) / d.x; } else { t1x = (R(-1) - o.x) / d.x; t0x = (R(1) - o.x) / d.x; }
      if (d.y >= R(0)) { t0y = (R(-1) - o.y) / d.y; t1y = (R(1) - o.y) / d.y; } else { t1y = (R(-1) - o.y) / d.y; t0y = (R(1) - o.y) / d.y; }
      if (d.z >= R(0)) { t0z = (R(-1) - o.z) / d.z; t1z = (R(1) - o.z) / d.z; } else { t1z = (R(-1) - o.z) / d.z; t0z = (R(1) - o.z) / d.z; }
      R tmin = fmax(fmax(t0x, t0y), t0z);
      R tmax = fmin(fmin(t1x, t1y), t1z);
      if (tmin > tmax) return;
      emit(tmin, R(0), R(0), 0);
      emit(tmax, R(0), R(0), 1);
      return;
    }
    case RT355_CYLINDER:
    case RT355_CONE: {  // src/shape.rs:249-327, caps 468-514
      const R ymin = ext_min, ymax = ext_max;
      const bool cone = kind == RT355_CONE;
      int em = 0;
      if (!cone) {
        R a = d.x * d.x + d.z * d.z;
        if (!approx_zero(a)) {
          R b = R(2) * o.x * d.x + R(2) * o.z * d.z;
          R c = o.x * o.x + o.z * o.z - R(1);
          R disc = b * b - R(4) * a * c;
          if (disc < R(0)) return;  // returns before the caps are tried (src/shape.rs:260-262)
          R t0 = (-b - sqrt(disc)) / (R(2) * a);
          R t1 = (-b + sqrt(disc)) / (R(2) * a);
          if (t0 > t1) { R s = t0; t0 = t1; t1 = s; }
          R y0 = o.y + t0 * d.y;
          if (ymin < y0 && y0 < ymax) emit(t0, R(0), R(0), em++);
          R y1 = o.y + t1 * d.y;
          if (ymin < y1 && y1 < ymax) emit(t1, R(0), R(0), em++);
        }
      } else {
        R a = d.x * d.x - d.y * d.y + d.z * d.z;
        R b = R(2) * (o.x * d.x - o.y * d.y + o.z * d.z);
        R c = o.x * o.x - o.y * o.y + o.z * o.z;
        if (approx_zero(a) && !approx_zero(b)) {
          emit(c / (R(-2) * b), R(0), R(0), em++);
        } else {
          // discriminant rounded to two decimals before the sign test (src/shape.rs:300)
          R disc = round((b * b - R(4) * a * c) * R(100)) / R(100);
          if (disc < R(0)) return;
          R t0 = (-b - sqrt(disc)) / (R(2) * a);
          R t1 = (-b + sqrt(disc)) / (R(2) * a);
          if (t0 > t1) { R s = t0; t0 = t1; t1 = s; }
          R y0 = o.y + t0 * d.y;
          if (ymin < y0 && y0 < ymax) emit(t0, R(0), R(0), em++);
          R y1 = o.y + t1 * d.y;
          if (ymin < y1 && y1 < ymax) emit(t1, R(0), R(0), em++);
        }
      }
      // intersect_caps (src/shape.rs:480-514)
      if (!(flags & PF_CLOSED) || approx_zero(d.y)) return;
      R t_lower = (ymin - o.y) / d.y;
      R t_upper = (ymax - o.y) / d.y;
      R rl = cone ? ymin * ymin : R(1), ru = cone ? ymax * ymax : R(1);
      {
        R x = o.x + t_lower * d.x, z = o.z + t_lower * d.z;
        if (x * x + z * z <= rl + EPS) emit(t_lower, R(0), R(0), em++);
      }
      {
        R x = o.x + t_upper * d.x, z = o.z + t_upper * d.z;
        if (x * x + z * z <= ru + EPS) emit(t_upper, R(0), R(0), em++);
      }
      return;
    }
    default:
      return;
  }
}

// ------------------------------------------------------------------------------------------------
// Moeller-Trumbore on one triangle record (src/shape.rs:328-375); the ray is already in object space.
// ------------------------------------------------------------------------------------------------
template <class R>
RT_FN bool tri_hit(const DevTri& T, V3<R> o, V3<R> d, R& t, R& u, R& v) {
#ifdef RT355_EXP_NO_TRI
  return false;  // timing experiment only: what the frame costs without any triangle arithmetic
#endif
  // Straight-line code with two wave-level exits: every lane evaluates the same expressions and carries a
  // predicate; the wave leaves early when no lane of it can still hit.  (Per-lane early returns cost an exec-mask
  // save / restore and a set of register copies at every exit, more than the arithmetic they skip.)
  const V3<R> p1 = {R(T.p1[0]), R(T.p1[1]), R(T.p1[2])};
  const V3<R> e1 = {R(T.p2[0]) - p1.x, R(T.p2[1]) - p1.y, R(T.p2[2]) - p1.z};  // src/triangle.rs:83
  const V3<R> e2 = {R(T.p3[0]) - p1.x, R(T.p3[1]) - p1.y, R(T.p3[2]) - p1.z};  // src/triangle.rs:84
  const V3<R> dir_cross_e2 = cross(d, e2);
  const R det = dot(e1, dir_cross_e2);
  const V3<R> p1_to_origin = o - p1;
  // The reference divides first (f = 1/det) and then tests u = f*un, v = f*vn.  Most tests miss, and a
  // miss can be told without the division: u < 0 exactly when un and det differ in sign, and u > 1
  // (or u + v > 1) for certain when the numerator exceeds |det| by more than the rounding of the
  // products.  Whatever is not rejected here goes through the reference's own expressions below.
  const R adet = fabs(det);
  const R slack = adet * (R(1) + R(16) * std::numeric_limits<R>::epsilon());
  const R tiny = adet * (R(16) * std::numeric_limits<R>::min());
  const R un = dot(p1_to_origin, dir_cross_e2);
  const R aun = fabs(un);
  // (bitwise operators on purpose: no short-circuit evaluation, hence no branches around single compares)
  const bool det_neg = det < R(0);
  bool maybe = !approx_zero(adet);
  maybe = maybe & !((aun > tiny) & (((un < R(0)) != det_neg) | (aun > slack)));
  if (__builtin_amdgcn_ballot_w64(maybe) == 0ull) return false;
  const V3<R> origin_cross_e1 = cross(p1_to_origin, e1);
  const R vn = dot(d, origin_cross_e1);
  const R avn = fabs(vn);
  maybe = maybe & !((avn > tiny) & (((vn < R(0)) != det_neg) | ((aun > tiny) & (aun + avn > slack))));
  if (__builtin_amdgcn_ballot_w64(maybe) == 0ull) return false;
  const R f = R(1) / det;
  u = f * un;
  v = f * vn;
  t = f * dot(e2, origin_cross_e1);
  return maybe & !((u < R(0)) | (u > R(1))) & !((v < R(0)) | ((u + v) > R(1)));
}

// ------------------------------------------------------------------------------------------------
// Box culling in single precision, conservative.  The boxes only decide which primitives get tested; the
// tests themselves stay in R.  So the slab arithmetic may run in float, as long as it never reports a
// miss for a box the exact ray touches.  With P a box plane, exact entry/exit parameters are
// (P - o) / d.  Here they are evaluated as fma(P, id, c) with id = rcp(float(d)) (the hardware reciprocal, good to
// one unit in the last place) and c = -(float(o) * id) -/+ E.  Relative to (|P| + |o|) * |id| the errors are: d to
// float 2^-24, the reciprocal 2^-23, o to float, the product o * id, the fma and the sum c, 2^-24 each: 7 * 2^-24 in
// all.  E is 2^-20 of that quantity with |P| replaced by the largest box coordinate of the scene, subtracted on the
// near side and added on the far side.  1/d is clamped to +-1e28, so axis-parallel rays stay finite.  The sign of
// id picks the near plane, so there is no min / max per axis.  (A reciprocal and a product in double, rounded
// afterwards, would allow a tighter E; they cost three double-precision divisions per ray and per traversal.)
// ------------------------------------------------------------------------------------------------
struct SlabRay {
  float ix, iy, iz;  // 1 / d
  float nx, ny, nz;  // near-side constants
  float fx, fy, fz;  // far-side constants
};
template <class R>
RT_FN SlabRay make_slab_ray(V3<R> o, V3<R> d, float abs_max) {
  SlabRay s;
  auto axis = [&](R oc, R dc, float& i, float& n, float& f) {
    const float df = (float)dc, of = (float)oc;
    i = __builtin_amdgcn_rcpf(df);
    if (!(fabsf(i) < 1e28f)) i = copysignf(1e28f, df);
    if (fabsf(i) < 1e-30f) i = copysignf(1e-30f, df);  // never zero: an infinite plane times i stays infinite
    const float oi = of * i;
    const float e = (abs_max + fabsf(of) * 1.0000002f) * fabsf(i) * 9.5367432e-7f;  // 2^-20
    n = -oi - e;
    f = -oi + e;
  };
  axis(o.x, d.x, s.ix, s.nx, s.fx);
  axis(o.y, d.y, s.iy, s.ny, s.fy);
  axis(o.z, d.z, s.iz, s.nz, s.fz);
  return s;
}
// [tmin, tmax] of the visitor, widened to the enclosing floats
template <class R>
RT_FN void slab_interval(R tmin, R tmax, float& lo, float& hi) {
  const float a = (float)tmin, b = (float)tmax;
  lo = a - fabsf(a) * 1.1920929e-7f;
  hi = fminf(b + fabsf(b) * 1.1920929e-7f, 3.4028235e38f);  // finite: an empty child slot (x planes at +inf) then never passes
}
// true when the ray may touch the box within [tlo, thi]; tn = (conservative) entry parameter
RT_FN bool slab_test(const SlabRay& s, float lox, float loy, float loz, float hix, float hiy, float hiz, float tlo, float thi, float& tn) {
  const bool px = s.ix >= 0.f, py = s.iy >= 0.f, pz = s.iz >= 0.f;
  const float nx = fmaf(px ? lox : hix, s.ix, s.nx), fx = fmaf(px ? hix : lox, s.ix, s.fx);
  const float ny = fmaf(py ? loy : hiy, s.iy, s.ny), fy = fmaf(py ? hiy : loy, s.iy, s.fy);
  const float nz = fmaf(pz ? loz : hiz, s.iz, s.nz), fz = fmaf(pz ? hiz : loz, s.iz, s.fz);
  tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, tlo));
  const float tf = fminf(fminf(fx, fy), fminf(fz, thi));
  return tn <= tf;
}

// ------------------------------------------------------------------------------------------------
// BVH traversal.  Replaces the brute-force loop of World::intersect (src/world.rs:35-44).
// Visitor: tmin(), tmax() give the live ray interval; prim(p) tests one prim of any kind, tri(p, T) one
// identity-transform triangle whose record is already loaded; both return true to stop.
// The stack lives in LDS, one column per lane (lds[k * BLOCK]), so a push / pop is a conflict-free
// ds_write / ds_read instead of a scratch round trip; entries beyond LDS_STACK spill to scratch.
// ------------------------------------------------------------------------------------------------
// walk(): one tree, from `root`, for the ray (o, d) the visitor holds; the stack entries [0, sp0) belong to an outer
// walk and are left alone.  Returns true when the visitor asked to stop.
// A leaf prim of kind RT355_INSTANCE (MODE & M_INST only) is an instanced mesh: the ray is taken to the mesh's object
// space -- once, not once per triangle -- and the mesh's own tree is walked with it, by this same function one level
// down.  The parameter t is the same on both sides (the direction is transformed, not normalised: src/ray.rs:32-37),
// so the visitor's interval carries over; only its ray is exchanged for the inner walk and put back afterwards.
template <class R, int MODE, class Vis>
RT_FN bool walk(const DevScene<R>& sc, int root, V3<R> o, V3<R> d, Vis& vis, Work& wk, int* lds, int* spill, const int sp0) {
  const SlabRay sr = make_slab_ray(o, d, sc.box_abs_max);
  // every visitor's lower bound is fixed for the length of a traversal; the upper one moves with the hits
  float tlo;
  {
    const float a = (float)vis.tmin();
    tlo = a - fabsf(a) * 1.1920929e-7f;
  }
  int sp = sp0;
  int cur = root;
  for (;;) {
#ifdef RT355_EXP_STAMPS
    DbgAcc& dacc = wk.dbg;
    // wave-level view: a trip runs the node body when some lane is at a node, the leaf bodies when some lane is at a leaf
    const unsigned long long dc0 = __builtin_amdgcn_s_memtime();
    const bool d_node = __ballot(cur >= 0) != 0ull;
    const bool d_tri = __ballot(cur < 0 && (((uint32_t)~cur) & LEAF_TRI_BIT)) != 0ull, d_prim = __ballot(cur < 0 && !(((uint32_t)~cur) & LEAF_TRI_BIT)) != 0ull;
    dacc.trips++, dacc.n_node += d_node, dacc.n_leaf += d_tri || d_prim;
    struct Closer { DbgAcc& a; unsigned long long c0; bool n, t, p; RT_FN ~Closer() { const unsigned long long dt = __builtin_amdgcn_s_memtime() - c0; const int k = (int)n + (int)t + (int)p; if (k) { if (n) a.cyc_node += dt / k; if (t) a.cyc_tri += dt / k; if (p) a.cyc_prim += dt / k; } } } closer{dacc, dc0, d_node, d_tri, d_prim};
#endif
    if (cur >= 0) {
      const BvhNode4 N = sc.nodes[cur];
      float thi;
      {
        const float b = (float)vis.tmax();
        thi = fminf(b + fabsf(b) * 1.1920929e-7f, 3.4028235e38f);  // finite: an empty child slot (x planes at +inf) then never passes
      }
      // entry parameter of every child the ray may touch, +inf for the others
      float k0, k1, k2, k3;
      if (!slab_test(sr, N.box[0][0], N.box[0][2], N.box[0][4], N.box[0][1], N.box[0][3], N.box[0][5], tlo, thi, k0)) k0 = INFINITY;
      if (!slab_test(sr, N.box[1][0], N.box[1][2], N.box[1][4], N.box[1][1], N.box[1][3], N.box[1][5], tlo, thi, k1)) k1 = INFINITY;
      if (!slab_test(sr, N.box[2][0], N.box[2][2], N.box[2][4], N.box[2][1], N.box[2][3], N.box[2][5], tlo, thi, k2)) k2 = INFINITY;
      if (!slab_test(sr, N.box[3][0], N.box[3][2], N.box[3][4], N.box[3][1], N.box[3][3], N.box[3][5], tlo, thi, k3)) k3 = INFINITY;
      int c0 = N.c[0], c1 = N.c[1], c2 = N.c[2], c3 = N.c[3];
      if constexpr (MODE & M_COUNT) wk.box += (c0 != -1) + (c1 != -1) + (c2 != -1) + (c3 != -1);
      // nearest first: 5-exchange sorting network on (entry, child)
      auto order = [](float& ka, int& ca, float& kb, int& cb) {
        const bool sw = kb < ka;
        const float kt = sw ? kb : ka, ku = sw ? ka : kb;
        const int ct = sw ? cb : ca, cu = sw ? ca : cb;
        ka = kt, kb = ku, ca = ct, cb = cu;
      };
      order(k0, c0, k1, c1), order(k2, c2, k3, c3), order(k0, c0, k2, c2), order(k1, c1, k3, c3), order(k1, c1, k2, c2);
      auto push = [&](int child) {
        if (sp < LDS_STACK) lds[sp * BLOCK] = child; else spill[sp - LDS_STACK] = child;
        ++sp;
      };
      if (k3 < INFINITY) push(c3);
      if (k2 < INFINITY) push(c2);
      if (k1 < INFINITY) push(c1);
      if (k0 < INFINITY) { cur = c0; continue; }
    } else
    {
      const uint32_t leaf = (uint32_t)~cur;
      const int first = (int)(leaf & LEAF_FIRST_MASK), count = (int)(leaf >> 26);
      if (leaf & LEAF_TRI_BIT) {
        // all records of the leaf are requested before the first is used: one memory latency per leaf
        DevTri T[LEAF_MAX];
#pragma unroll
        for (int k = 0; k < LEAF_MAX; ++k) T[k] = sc.tri[first + (k < count ? k : 0)];
#pragma unroll
        for (int k = 0; k < LEAF_MAX; ++k)
          if (k < count) {
            if constexpr (MODE & M_COUNT) wk.prim[T[k].flags & PF_KIND_MASK]++;
            if (vis.tri(first + k, T[k])) return true;
          }
        for (int k = LEAF_MAX; k < count; ++k) {  // only leaves forced at the depth limit are larger
          const DevTri Tk = sc.tri[first + k];
          if constexpr (MODE & M_COUNT) wk.prim[Tk.flags & PF_KIND_MASK]++;
          if (vis.tri(first + k, Tk)) return true;
        }
      } else {
        for (int k = 0; k < count; ++k) {
          const int p = first + k;
          if constexpr ((MODE & M_INST) != 0) {
            if ((sc.prim_info[p].flags & PF_KIND_MASK) == RT355_INSTANCE) {
              const DevXform<R> X = sc.xform[p];
              const V3<R> so = vis.o, sd = vis.d;
              vis.o = xf_point(X.r, so), vis.d = xf_vec(X.r, sd);
              if constexpr (MODE & M_COUNT) wk.xf++;
              const bool stop = walk<R, MODE & ~M_INST>(sc, (int)X.root, vis.o, vis.d, vis, wk, lds, spill, sp);
              vis.o = so, vis.d = sd;
              if (stop) return true;
              continue;
            }
          }
          if (vis.prim(p)) return true;
        }
      }
    }
    if (sp == sp0) return false;
    --sp;
    cur = sp < LDS_STACK ? lds[sp * BLOCK] : spill[sp - LDS_STACK];
  }
}

template <class R, int MODE, class Vis>
RT_FN void traverse(const DevScene<R>& sc, V3<R> o, V3<R> d, Vis& vis, Work& wk, int* lds) {
  for (int p = sc.n_bounded; p < sc.n_top; ++p)
    if (vis.prim_u(p)) return;
  if (sc.n_bounded == 0) return;
  int spill[STACK_DEPTH - LDS_STACK];
  (void)walk<R, MODE>(sc, 0, o, d, vis, wk, lds, spill, 0);
}

// ------------------------------------------------------------------------------------------------
// Packet traversal: the 64 rays of a wave walk the BVH TOGETHER.  The current node, the stack and every
// branch are wave-uniform, so node and triangle records come in through scalar loads (one request per
// wave instead of 64 divergent 128-byte fetches) and the vector ALU does nothing but slab and triangle
// arithmetic.  A child is entered when ANY live lane's ray overlaps its box.  Every lane tests every triangle of a visited leaf against its own interval, so
// each lane still finds exactly its own nearest / any hit.  Good for coherent rays (primary, shadow,
// mirror reflections); incoherent generations use the per-lane traverse above.
// All 64 lanes must call this together; `live` is false for lanes without a ray.
// ------------------------------------------------------------------------------------------------
template <class R>
RT_FN DevTri load_tri_uniform(const DevScene<R>& sc, int p) {
  DevTri T;
  cptr<float> q = (cptr<float>)(const float*)&sc.tri[p];
#pragma unroll
  for (int k = 0; k < 3; ++k) T.p1[k] = q[k], T.p2[k] = q[3 + k], T.p3[k] = q[6 + k];
  cptr<int32_t> qi = (cptr<int32_t>)(const int32_t*)&sc.tri[p].shape;
  T.shape = qi[0], T.flags = qi[1], T.pad = 0;
  return T;
}

constexpr int PARK_INTS = 6 * BLOCK * 2;  // six values of R (at most 8 bytes) per lane
constexpr int PACKET_LEAF_GROUP = 2;  // triangle records of a leaf the packet traversal holds at a time

// The slab constants of a packet's ray.  Per axis the two planes of a box give a = fma(lo, i, ca) and
// b = fma(hi, i, cb) with ca = -(o * i) - e * sign(i), cb = -(o * i) + e * sign(i): whichever of the two is the
// near plane for this lane's direction carries -e and the other +e, so min(a, b) is the conservative entry and
// max(a, b) the conservative exit of the axis without a select per plane.  (lo, hi) sit side by side in the node
// and (ca, cb) side by side here: one packed instruction evaluates both.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct SlabRayP {
  float ix, iy, iz;
  f32x2 cx, cy, cz;  // (ca, cb) per axis
};
RT_FN SlabRayP make_slab_ray_p(const SlabRay& s) {
  SlabRayP r;
  r.ix = s.ix, r.iy = s.iy, r.iz = s.iz;
  r.cx = s.ix >= 0.f ? f32x2{s.nx, s.fx} : f32x2{s.fx, s.nx};
  r.cy = s.iy >= 0.f ? f32x2{s.ny, s.fy} : f32x2{s.fy, s.ny};
  r.cz = s.iz >= 0.f ? f32x2{s.nz, s.fz} : f32x2{s.fz, s.nz};
  return r;
}
// One box against the lane's ray: true when the ray may touch it within [tlo, thi]; tn = conservative entry, >= tlo.
// An empty child slot has both its x planes at +inf: entry and exit of that axis are then +inf or -inf together and
// the test fails for every finite [tlo, thi].
RT_FN bool slab_test_p(const SlabRayP& s, f32x2 bx, f32x2 by, f32x2 bz, float tlo, float thi, float& tn) {
  const f32x2 tx = __builtin_elementwise_fma(bx, f32x2{s.ix, s.ix}, s.cx);
  const f32x2 ty = __builtin_elementwise_fma(by, f32x2{s.iy, s.iy}, s.cy);
  const f32x2 tz = __builtin_elementwise_fma(bz, f32x2{s.iz, s.iz}, s.cz);
  tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), tlo));
  const float tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), thi));
  return tn <= tf;
}

// walk_packet(): one tree, from `root`, for the wave's rays (o, d); the stack slots [0, sp0) belong to an outer walk,
// which passes its stack register BY VALUE: this walk pushes and pops above sp0 in its own copy and the caller's is
// as it was (taken by reference, the register costs every packet kernel 24 more vector registers and a wave per SIMD).
// Lanes whose visitor is done leave `live` (returned); the walk ends when the tree is exhausted or no lane is left.  An instanced
// mesh in a leaf (MODE & M_INST) takes every ray of the wave to the mesh's object space with one transformation read
// through the scalar cache, and the wave walks the mesh's tree, one level down, the same way.
template <class R, int MODE, class Vis>
RT_FN bool walk_packet(const DevScene<R>& sc, int root, V3<R> o, V3<R> d, bool live, Vis& vis, Work& wk, int stackv, const int sp0) {
  const SlabRayP sr = make_slab_ray_p(make_slab_ray(o, d, sc.box_abs_max));
  // every visitor's lower bound is fixed for the length of a traversal
  float tlo;
  {
    const float a = (float)vis.tmin();
    tlo = a - fabsf(a) * 1.1920929e-7f;
  }
  const int lane_id = (int)(threadIdx.x & 63u);
  int sp = sp0;
  int cur = root;
  for (;;) {
    const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(live);
    if (live_mask == 0ull) return live;
#ifdef RT355_EXP_STAMPS
    DbgAcc& dacc = wk.dbg;
    const unsigned long long dc0 = __builtin_amdgcn_s_memtime();
    const bool d_node = cur >= 0, d_tri = cur < 0 && (((uint32_t)~cur) & LEAF_TRI_BIT), d_prim = cur < 0 && !d_tri;
    dacc.trips++, dacc.n_node += d_node, dacc.n_leaf += !d_node;
    struct Closer { DbgAcc& a; unsigned long long c0; bool n, t, p; RT_FN ~Closer() { const unsigned long long dt = __builtin_amdgcn_s_memtime() - c0; if (n) a.cyc_node += dt; if (t) a.cyc_tri += dt; if (p) a.cyc_prim += dt; } } closer{dacc, dc0, d_node, d_tri, d_prim};
#endif
    if (cur >= 0) {
      // the whole node through scalar loads issued together: one wait per node
      cptr<f32x2> q = (cptr<f32x2>)(const f32x2*)&sc.nodes[cur].box[0][0];
      cptr<int32_t> qc = (cptr<int32_t>)(const int32_t*)&sc.nodes[cur].c[0];
      f32x2 B[BVH_WIDTH][3];
      int ch[BVH_WIDTH];
#pragma unroll
      for (int k = 0; k < BVH_WIDTH; ++k) B[k][0] = q[3 * k], B[k][1] = q[3 * k + 1], B[k][2] = q[3 * k + 2], ch[k] = qc[k];
      float thi;
      {
        const float b = (float)vis.tmax();
        thi = fminf(b + fabsf(b) * 1.1920929e-7f, 3.4028235e38f);
      }
      // A child is entered when any live lane may touch it.
      unsigned long long m[BVH_WIDTH];
      float tn[BVH_WIDTH];
      int n_valid = 0;
#pragma unroll
      for (int k = 0; k < BVH_WIDTH; ++k) {
        // the compare's own lane mask, narrowed to the live lanes on the scalar unit
        m[k] = __builtin_amdgcn_ballot_w64(slab_test_p(sr, B[k][0], B[k][1], B[k][2], tlo, thi, tn[k])) & live_mask;
        if constexpr (MODE & M_COUNT) n_valid += ch[k] != -1;
      }
#ifndef RT355_EXP_NO_SINGLE_CHILD
      // Deep in the tree a packet mostly touches one child of a node, or none: those steps need no order and no stack
      // traffic, and leave before the keys are even formed.
      {
        const int touched = (m[0] != 0ull) + (m[1] != 0ull) + (m[2] != 0ull) + (m[3] != 0ull);
        if (touched <= 1) {
          if constexpr (MODE & M_COUNT) wk.box += live ? n_valid : 0;
          if (touched == 1) {
            cur = m[0] ? ch[0] : m[1] ? ch[1] : m[2] ? ch[2] : ch[3];
            continue;
          }
          if (sp == sp0) return live;
          cur = __builtin_amdgcn_readlane(stackv, --sp);
          continue;
        }
      }
#endif
      // Order: by the entry parameter of the first lane that touches the child (a scalar, so the sort runs on the
      // scalar unit).
      uint32_t key[BVH_WIDTH];
#pragma unroll
      for (int k = 0; k < BVH_WIDTH; ++k) {
        // entry parameter of the first lane that touches the child; with no such lane the scalar bit search yields -1,
        // the lane read is then any lane's and the key is discarded.  Negative entries (visitors whose interval starts
        // below zero) order as zero.
        const int lane = __builtin_ffsll((long long)m[k]) - 1;
        const int bits = __builtin_amdgcn_readlane((int)__float_as_uint(tn[k]), lane);
        key[k] = m[k] ? (uint32_t)(bits < 0 ? 0 : bits) : 0xFFFFFFFFu;
      }
      if constexpr (MODE & M_COUNT) wk.box += live ? n_valid : 0;
      auto order = [](uint32_t& ka, int& ca, uint32_t& kb, int& cb) {
        if (kb < ka) {
          const uint32_t kt = ka; ka = kb, kb = kt;
          const int ct = ca; ca = cb, cb = ct;
        }
      };
      order(key[0], ch[0], key[1], ch[1]), order(key[2], ch[2], key[3], ch[3]), order(key[0], ch[0], key[2], ch[2]);
      order(key[1], ch[1], key[3], ch[3]), order(key[1], ch[1], key[2], ch[2]);
      // the touched children come first, nearest in ch[0]: that one is entered, the others go on the stack, farthest lowest
      const uint32_t INV = 0xFFFFFFFFu;
      const int nv = key[3] != INV ? 4 : key[2] != INV ? 3 : key[1] != INV ? 2 : key[0] != INV ? 1 : 0;  // touched children
#pragma unroll
      for (int j = 1; j < BVH_WIDTH; ++j) {
        const int slot = nv > j ? sp + nv - 1 - j : -1;  // scalar; -1 is no lane's number
        stackv = lane_id == slot ? ch[j] : stackv;
      }
      sp += (nv > 1 ? nv : 1) - 1;
      if (nv > 0) {
        cur = ch[0];
        continue;
      }
    } else {
      const uint32_t leaf = (uint32_t)~cur;
      const int first = (int)(leaf & LEAF_FIRST_MASK), count = (int)(leaf >> 26);
      if (leaf & LEAF_TRI_BIT) {
        // two records at a time: four would hold 48 scalar registers at once, and the packet kernels are short of
        // scalar registers (what does not fit is parked in vector lanes and fetched back every trip); measured 3 %
        // of the frame against one more dependent scalar load per leaf of three or four triangles
#pragma unroll
        for (int h = 0; h < LEAF_MAX; h += PACKET_LEAF_GROUP) {
          if (h >= count) break;
          DevTri T[PACKET_LEAF_GROUP];
#pragma unroll
          for (int k = 0; k < PACKET_LEAF_GROUP; ++k) T[k] = load_tri_uniform(sc, first + (h + k < count ? h + k : 0));
#pragma unroll
          for (int k = 0; k < PACKET_LEAF_GROUP; ++k)
            if (h + k < count) {  // wave-uniform; lanes without a ray run along, the visitor ignores them
              if constexpr (MODE & M_COUNT) wk.prim[T[k].flags & PF_KIND_MASK] += live ? 1 : 0;
              live = live & !vis.tri_masked(first + h + k, T[k], live);
            }
        }
        for (int k = LEAF_MAX; k < count; ++k) {
          const DevTri Tk = load_tri_uniform(sc, first + k);
          if constexpr (MODE & M_COUNT) wk.prim[Tk.flags & PF_KIND_MASK] += live ? 1 : 0;
          live = live & !vis.tri_masked(first + k, Tk, live);
        }
      } else {
        for (int k = 0; k < count; ++k) {
          const int p = first + k;
          if constexpr ((MODE & M_INST) != 0) {
            if ((load_uniform(&sc.prim_info[p]).flags & PF_KIND_MASK) == RT355_INSTANCE) {
              const DevXform<R> X = load_uniform(&sc.xform[p]);
              const V3<R> so = vis.o, sd = vis.d;
              vis.o = xf_point(X.r, so), vis.d = xf_vec(X.r, sd);
              if constexpr (MODE & M_COUNT) wk.xf += live ? 1 : 0;
              live = walk_packet<R, MODE & ~M_INST>(sc, (int)X.root, vis.o, vis.d, live, vis, wk, stackv, sp);
              vis.o = so, vis.d = sd;
              continue;
            }
          }
          if (live && vis.prim_u(p)) live = false;
        }
      }
    }
    if (sp == sp0) return live;
    cur = __builtin_amdgcn_readlane(stackv, --sp);
  }
}

// The stack of a packet traversal is ONE vector register: slot k lives in lane k (STACK_DEPTH == 64 lanes), pushed with
// a compare and a select against the lane number and popped with v_readlane from a scalar stack pointer -- no LDS
// round trip between a leaf and the next node.
template <class R, int MODE, class Vis>
RT_FN void traverse_packet(const DevScene<R>& sc, V3<R> o, V3<R> d, bool live, Vis& vis, Work& wk, int* /*unused*/) {
  static_assert(STACK_DEPTH == 64, "the packet stack is one lane per slot");
  for (int p = sc.n_bounded; p < sc.n_top; ++p)
    if (live && vis.prim_u(p)) live = false;
  if (sc.n_bounded == 0) return;
  int stackv = 0;
  (void)walk_packet<R, MODE>(sc, 0, o, d, live, vis, wk, stackv, 0);
}

// ------------------------------------------------------------------------------------------------
// Quad traversal: FOUR adjacent lanes walk the tree for ONE ray.  At an inner node each lane fetches and
// tests one of the four child boxes (the quad reads the node as one 128-byte line); in a leaf each lane
// tests one of the (up to four) primitives.  The dependent instruction chain of a ray is a quarter of what
// one lane alone would issue, which is what bounds the deeper generations: they hold about as many rays as
// the chip has lanes, so their launches last as long as the slowest single traversal.  The stack is one LDS
// column per quad, read by all four lanes and written by the lane that owns the child.  Lanes of a quad
// exchange values through DPP quad permutes (no LDS, no scalar unit); every exchange sits where the four
// lanes are converged.
// Visitors are the same as for the other traversals; each lane keeps the result of the primitives it
// tested itself and the visitor's quad_* member merges the four afterwards.
// ------------------------------------------------------------------------------------------------
constexpr int QUADS = BLOCK / 4;  // rays a block traverses at a time, and the stride of a quad's stack column

template <int J> RT_FN int quad_bcast(int v) { return __builtin_amdgcn_mov_dpp(v, J * 0x55, 0xF, 0xF, true); }
RT_FN int quad_swap1(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }  // lanes 1,0,3,2
RT_FN int quad_swap2(int v) { return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true); }  // lanes 2,3,0,1
RT_FN float quad_swap1(float v) { return __int_as_float(quad_swap1(__float_as_int(v))); }
RT_FN float quad_swap2(float v) { return __int_as_float(quad_swap2(__float_as_int(v))); }
RT_FN double quad_swap1(double v) { return __hiloint2double(quad_swap1(__double2hiint(v)), quad_swap1(__double2loint(v))); }
RT_FN double quad_swap2(double v) { return __hiloint2double(quad_swap2(__double2hiint(v)), quad_swap2(__double2loint(v))); }
RT_FN float quad_min(float v) {
  v = fminf(v, quad_swap1(v));
  return fminf(v, quad_swap2(v));
}
RT_FN bool quad_any(bool b) {
  int v = b ? 1 : 0;
  v |= quad_swap1(v);
  v |= quad_swap2(v);
  return v != 0;
}

template <class R, int MODE, class Vis>
RT_FN void traverse_quad(const DevScene<R>& sc, V3<R> o, V3<R> d, Vis& vis, Work& wk, int* stack) {
  const int ql = threadIdx.x & 3;
  bool stop = false;
  for (int p = sc.n_bounded + ql; p < sc.n_top; p += 4) stop |= vis.prim(p);
  if (quad_any(stop) || sc.n_nodes == 0) return;
  const SlabRay sr = make_slab_ray(o, d, sc.box_abs_max);
  float tlo, thi;
  slab_interval(vis.tmin(), vis.tmax(), tlo, thi);
  thi = quad_min(thi);
  int sp = 0;
  int cur = 0;
  for (;;) {
    if (cur >= 0) {
      const float* b = sc.nodes[cur].box[ql];
      const float lox = b[0], hix = b[1], loy = b[2], hiy = b[3], loz = b[4], hiz = b[5];
      const int child = sc.nodes[cur].c[ql];
      if constexpr (MODE & M_COUNT) wk.box += child != -1;
      float tn;
      const bool h = slab_test(sr, lox, loy, loz, hix, hiy, hiz, tlo, thi, tn);
      // Nearest first.  Keys are made distinct by the lane number in their two lowest bits, misses sort
      // last; a lane's rank is the number of smaller keys in its quad.
      const uint32_t MISS = 0xFFFFFFFCu;
      const uint32_t key = (h ? (__float_as_uint(fmaxf(tn, 0.f)) & ~3u) : MISS) | (uint32_t)ql;
      const uint32_t k0 = (uint32_t)quad_bcast<0>((int)key), k1 = (uint32_t)quad_bcast<1>((int)key);
      const uint32_t k2 = (uint32_t)quad_bcast<2>((int)key), k3 = (uint32_t)quad_bcast<3>((int)key);
      const int rank = (k0 < key) + (k1 < key) + (k2 < key) + (k3 < key);
      const int nh = (k0 < MISS) + (k1 < MISS) + (k2 < MISS) + (k3 < MISS);  // the same in all four lanes
      if (h && rank > 0) stack[(sp + nh - 1 - rank) * QUADS] = child;  // farthest lowest: the next pop is the nearest
      int next = (h && rank == 0) ? child : 0;
      next |= quad_swap1(next);
      next |= quad_swap2(next);
      if (nh > 0) {
        sp += nh - 1;
        cur = next;
        continue;
      }
    } else {
      const uint32_t leaf = (uint32_t)~cur;
      const int first = (int)(leaf & LEAF_FIRST_MASK), count = (int)(leaf >> 26);
      if (leaf & LEAF_TRI_BIT) {
        for (int k = ql; k < count; k += 4) {
          const DevTri T = sc.tri[first + k];
          if constexpr (MODE & M_COUNT) wk.prim[T.flags & PF_KIND_MASK]++;
          stop |= vis.tri(first + k, T);
        }
      } else {
        for (int k = ql; k < count; k += 4) stop |= vis.prim(first + k);
      }
      if (quad_any(stop)) return;
      slab_interval(vis.tmin(), vis.tmax(), tlo, thi);
      thi = quad_min(thi);
    }
    if (sp == 0) return;
    --sp;
    cur = stack[sp * QUADS];
  }
}

// Nearest hit with t >= 0; among equal t the lowest shape index wins (stable sort src/world.rs:42 +
// first minimum src/intersection.rs:151).  NaN t never qualifies.
template <class R, int MODE>
struct ClosestVis {
  const DevScene<R>& sc;
  V3<R> o, d;
  Work& wk;
  R best_t, best_u, best_v;
  int best_p, best_s, best_flags;
  RT_FN ClosestVis(const DevScene<R>& s, V3<R> o_, V3<R> d_, Work& w)
      : sc(s), o(o_), d(d_), wk(w), best_t(inf<R>()), best_u(0), best_v(0), best_p(-1), best_s(0x7fffffff), best_flags(0) {}
  RT_FN R tmin() const { return R(0); }
  RT_FN R tmax() const { return best_t; }
  RT_FN void accept(R t, R u, R v, int p, int shape, int flags) {
    if (t >= R(0) && (t < best_t || (t == best_t && shape < best_s))) best_t = t, best_u = u, best_v = v, best_p = p, best_s = shape, best_flags = flags;
  }
  // the same as selects: for callers that run every lane through the same straight-line code
  RT_FN void accept_if(bool hit, R t, R u, R v, int p, int shape, int flags) {
    const bool better = hit & (t >= R(0)) & ((t < best_t) | ((t == best_t) & (shape < best_s)));
    best_t = better ? t : best_t, best_u = better ? u : best_u, best_v = better ? v : best_v;
    best_p = better ? p : best_p, best_s = better ? shape : best_s, best_flags = better ? flags : best_flags;
  }
  RT_FN bool prim(int p) {
    const PrimInfo info = sc.prim_info[p];
    prim_hits<R, MODE, false, true>(sc, p, info.flags, o, d, wk, [&](R t, R u, R v, int) { accept(t, u, v, p, info.shape, info.flags); });
    return false;
  }
  // p is the same in every lane of the wave
  RT_FN bool prim_u(int p) {
    const PrimInfo info = load_uniform(&sc.prim_info[p]);
    prim_hits<R, MODE, true, true>(sc, p, info.flags, o, d, wk, [&](R t, R u, R v, int) { accept(t, u, v, p, info.shape, info.flags); });
    return false;
  }
  RT_FN bool tri(int p, const DevTri& T) {
    R t = R(0), u = R(0), v = R(0);
    const bool hit = tri_hit(T, o, d, t, u, v);
    accept_if(hit, t, u, v, p, T.shape, T.flags);
    return false;
  }
  // for the packet traversal: every lane of the wave calls, `on` marks the lanes that have a ray
  RT_FN bool tri_masked(int p, const DevTri& T, bool on) {
    R t = R(0), u = R(0), v = R(0);
    const bool hit = tri_hit(T, o, d, t, u, v);
    accept_if(hit & on, t, u, v, p, T.shape, T.flags);
    return false;
  }
  // After a quad traversal: true in the ONE lane of the quad whose own result is the quad's nearest hit by
  // the same order as accept() (t, then shape index; the lane number settles what is left, which is only
  // the case of four misses).  That lane writes the ray's record.
  RT_FN bool quad_winner() const {
    const int ql = threadIdx.x & 3;
    R t = best_t;
    int s = best_s, l = ql;
    {
      const R tp = quad_swap1(t);
      const int sp = quad_swap1(s), lp = quad_swap1(l);
      if (tp < t || (tp == t && (sp < s || (sp == s && lp < l)))) t = tp, s = sp, l = lp;
    }
    {
      const R tp = quad_swap2(t);
      const int sp = quad_swap2(s), lp = quad_swap2(l);
      if (tp < t || (tp == t && (sp < s || (sp == s && lp < l)))) t = tp, s = sp, l = lp;
    }
    return l == ql;
  }
};

// is_shadowed: some intersection with 0 <= t < limit (src/world.rs:45-57; limit = distance + EPSILON).
template <class R, int MODE>
struct AnyVis {
  const DevScene<R>& sc;
  V3<R> o, d;
  Work& wk;
  R limit;
  bool found;
  RT_FN AnyVis(const DevScene<R>& s, V3<R> o_, V3<R> d_, R lim, Work& w) : sc(s), o(o_), d(d_), wk(w), limit(lim), found(false) {}
  RT_FN R tmin() const { return R(0); }
  RT_FN R tmax() const { return limit; }
  RT_FN bool prim(int p) {
    const PrimInfo info = sc.prim_info[p];
    prim_hits<R, MODE, false, true>(sc, p, info.flags, o, d, wk, [&](R t, R, R, int) {
      if (t >= R(0) && t < limit) found = true;
    });
    return found;
  }
  RT_FN bool prim_u(int p) {
    const PrimInfo info = load_uniform(&sc.prim_info[p]);
    prim_hits<R, MODE, true, true>(sc, p, info.flags, o, d, wk, [&](R t, R, R, int) {
      if (t >= R(0) && t < limit) found = true;
    });
    return found;
  }
  RT_FN bool tri(int, const DevTri& T) {
    R t = R(0), u, v;
    const bool hit = tri_hit(T, o, d, t, u, v);
    found = found | (hit & (t >= R(0)) & (t < limit));
    return found;
  }
  RT_FN bool tri_masked(int, const DevTri& T, bool on) {
    R t = R(0), u, v;
    const bool hit = tri_hit(T, o, d, t, u, v);
    found = found | (on & hit & (t >= R(0)) & (t < limit));
    return found;
  }
};

// n1/n2 of prepare_computation (src/intersection.rs:86-127) without materialising the sorted list.
// The reference walks the sorted intersections and toggles shapes in a container: a shape is present
// after an odd number of its intersections, and the container's last element is the present shape whose
// latest intersection comes last in the sorted order.  So the answer is found walking BACKWARDS from the
// hit: take the intersection with the largest (t, shape index) key below a bound (initially the hit's
// key); if its shape has an odd number of intersections before the hit it is the container's last
// element, otherwise lower the bound to that key and look again.  Each step is a nearest-hit query along
// the reversed ray, culled like any other, instead of a walk over the whole line.
// The visitor works on the reversed ray (o, -d): its parameter is tr = -t (negating d negates every t
// exactly), so "largest t below the bound" is "smallest tr above -bound".
template <class R, int MODE>
struct BackwardVis {
  const DevScene<R>& sc;
  V3<R> o, d;  // d is the REVERSED direction
  Work& wk;
  int p_hit;
  R bound_t;   // forward t of the bound, exclusive together with bound_s
  int bound_s;
  R floor_t;   // forward t below which nothing is of interest, exclusive together with floor_s
  int floor_s;
  bool have;
  R ct;        // forward t of the candidate
  int cs, cp, cflags;
  RT_FN BackwardVis(const DevScene<R>& s, V3<R> o_, V3<R> d_, int ph, R bt, int bs, R ft, int fs, Work& w)
      : sc(s), o(o_), d(d_), wk(w), p_hit(ph), bound_t(bt), bound_s(bs), floor_t(ft), floor_s(fs), have(false), ct(-inf<R>()), cs(-1), cp(-1), cflags(0) {}
  RT_FN R tmin() const { return -bound_t; }
  RT_FN R tmax() const { return have ? -ct : -floor_t; }
  RT_FN void accept(R tr, int p, int shape, int flags) {
    const R t = -tr;
    const bool below = t < bound_t || (t == bound_t && shape < bound_s);
    const bool above = t > floor_t || (t == floor_t && shape > floor_s);
    const bool better = !have || t > ct || (t == ct && shape > cs);
    if (below && above && better) have = true, ct = t, cs = shape, cp = p, cflags = flags;  // NaN fails `below`
  }
  RT_FN bool prim(int p) {
    if (p == p_hit) return false;
    const PrimInfo info = sc.prim_info[p];
    prim_hits<R, MODE>(sc, p, info.flags, o, d, wk, [&](R tr, R, R, int) { accept(tr, p, info.shape, info.flags); });
    return false;
  }
  RT_FN bool prim_u(int p) {
    if (p == p_hit) return false;
    const PrimInfo info = load_uniform(&sc.prim_info[p]);
    prim_hits<R, MODE, true>(sc, p, info.flags, o, d, wk, [&](R tr, R, R, int) { accept(tr, p, info.shape, info.flags); });
    return false;
  }
  RT_FN bool tri(int p, const DevTri& T) {
    if (p == p_hit) return false;
    R tr, u, v;
    if (tri_hit(T, o, d, tr, u, v)) accept(tr, p, T.shape, T.flags);
    return false;
  }
  RT_FN bool tri_masked(int p, const DevTri& T, bool on) {
    R tr = R(0), u, v;
    const bool hit = tri_hit(T, o, d, tr, u, v);
    if (on & hit & (p != p_hit)) accept(tr, p, T.shape, T.flags);
    return false;
  }
};

// ------------------------------------------------------------------------------------------------
// Block-level compaction.  Each wave counts with __ballot / __popcll, the four wave totals meet in LDS,
// and ONE lane per block touches the global counter: a single word takes fewer than 100 atomics per
// microsecond on this chip, so one atomic per wave (32 400 for a 1080p frame) would by itself cost more
// than the traversal.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t WAVE_HAS_TRANSPARENT = 0x80000000u;  // in a wave's entry of blk_hits, above the count (at most 64)
RT_FN unsigned long long lanes_below() { return (1ull << (threadIdx.x & 63u)) - 1ull; }
RT_FN uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// Id range of generation `gen`: generations are stacked in the pool in order.  A generation that did
// not fit is cut at the pool's capacity (the frame is then reported as RT355_ECAPACITY).
RT_FN void gen_range(const FrameCtl* ctl, int gen, long long capacity, uint32_t& start, uint32_t& count) {
  unsigned long long s = 0;
  for (int g = 0; g < gen; ++g) s += ctl->gen_count[g].v;
  unsigned long long c = ctl->gen_count[gen].v;
  if (s > (unsigned long long)capacity) s = capacity;
  if (s + c > (unsigned long long)capacity) c = capacity - s;
  start = (uint32_t)s, count = (uint32_t)c;
}

// a / b for 0 <= a < 2^23, 0 < b <= a's range, with inv_b = 1.f / b: the float product is within one of the
// quotient, one correction step settles it.  (A 64-bit integer division costs some 100 instructions per ray.)
RT_FN void div_small(uint32_t a, uint32_t b, float inv_b, uint32_t& q, uint32_t& r) {
  q = (uint32_t)((float)a * inv_b);
  int rem = (int)a - (int)(q * b);
  if (rem < 0) --q, rem += (int)b;
  else if (rem >= (int)b) ++q, rem -= (int)b;
  r = (uint32_t)rem;
}

// The hit of a ray as the trace kernels leave it: prim id (-1 = miss) and the prim's flags (kind, material,
// reflective / transparent), so that the kernels behind need no look-up to learn what was hit.
struct HitRec {
  int32_t prim, flags;
};
template <class R> RT_FN HitRec load_hit(const RayPool<R>& pool, uint32_t r) {
  const int2 h = *(const int2*)(pool.hit_prim + 2ull * r);
  return HitRec{h.x, h.y};
}
template <class R> RT_FN void store_hit(const RayPool<R>& pool, uint32_t r, int prim, int flags) { *(int2*)(pool.hit_prim + 2ull * r) = make_int2(prim, flags); }

// Origin of ray r.  The rays of generation 0 of a frame all start at the camera: their origin is not kept per ray
// but read from the frame's view (scalar loads: the 64 rays of a wave belong to one tile, hence to one frame).
// pool.views is null when generation 0 was loaded from caller-supplied rays.
template <class R>
RT_FN V3<R> view_origin(const DevView<R>& v) { return {v.inv[3], v.inv[7], v.inv[11]}; }  // inverse * point(0, 0, 0), src/camera.rs:68
template <class R>
RT_FN V3<R> ray_origin(const RayPool<R>& pool, uint32_t r) {
  if (pool.views != nullptr && r < pool.n_view_rays) {  // wave-uniform: a launch works on one generation
    uint32_t frame = 0, rest;
    if (pool.n_views > 1) div_small(__builtin_amdgcn_readfirstlane(r), pool.rays_per_view, pool.inv_rays_per_view, frame, rest);
    const DevView<R>* v = pool.views + __builtin_amdgcn_readfirstlane(frame);
    cptr<R> q = (cptr<R>)(const R*)v->inv;
    return {q[3], q[7], q[11]};
  }
  return ld3(pool.org + 3ull * r);
}

// ------------------------------------------------------------------------------------------------
// Primary rays: Camera::ray_for_pixel (src/camera.rs:62-72) with the inverse hoisted out of the pixel
// loop.  Ray i of generation 0 is pixel (x,y) of this rank's tile list: 8x8 pixel blocks per wave.
// ------------------------------------------------------------------------------------------------
// tile_slot: the tile's place in this rank's output (frame-major over the batch); frame: which view.
// Tile counts stay below 2^22 (canvas edges are at most 65536 pixels), batches below 2^6 frames; the host refuses
// chains whose slot count would pass 2^23.
template <class R>
RT_FN bool pixel_of(const DevCamera<R>& cam, uint32_t i, int& x, int& y, int& tile_slot, int& in_tile, int& frame) {
  tile_slot = (int)(i / TILE_PIX);
  const int j = (int)(i % TILE_PIX);
  const int blk = j >> 6, lane = j & 63;
  const int xt = (blk & 3) * 8 + (lane & 7), yt = (blk >> 2) * 8 + (lane >> 3);
  in_tile = yt * TILE + xt;
  uint32_t tile_local = (uint32_t)tile_slot, f = 0;
  if (cam.n_frames > 1) div_small((uint32_t)tile_slot, (uint32_t)cam.tiles_per_frame, cam.inv_tiles_per_frame, f, tile_local);
  frame = (int)f;
  const uint32_t tile = tile_local * (uint32_t)cam.n_ranks + (uint32_t)cam.rank;
  if (tile >= (uint32_t)cam.n_tiles) return false;
  uint32_t ty, tx;
  div_small(tile, (uint32_t)cam.tiles_x, cam.inv_tiles_x, ty, tx);
  x = (int)tx * TILE + xt;
  y = (int)ty * TILE + yt;
  return x < cam.width && y < cam.height;
}

template <class R>
RT_FN void primary_ray(const DevCamera<R>& cam, uint32_t i, V3<R>& org, V3<R>& dir) {
  int x, y, ts, it, frame;
  org = {R(0), R(0), R(0)}, dir = {R(0), R(0), R(0)};  // zero direction = inactive ray (tile padding)
  if (pixel_of(cam, i, x, y, ts, it, frame)) {
    // a tile is 1024 consecutive rays, so the 64 rays of a wave share their frame: the view comes in through
    // scalar loads
    const DevView<R>& v = cam.n_frames > 1 ? cam.views[__builtin_amdgcn_readfirstlane(frame)] : cam.view;
    const R x_offset = (R(x) + R(0.5)) * v.pixel_size;
    const R y_offset = (R(y) + R(0.5)) * v.pixel_size;
    const R world_x = v.half_width - x_offset;
    const R world_y = v.half_height - y_offset;
    const V3<R> pixel = xf_point(v.inv, mk<R>(world_x, world_y, R(-1)));
    org = view_origin(v);
    dir = normalize(pixel - org);
  }
}

// ------------------------------------------------------------------------------------------------
// k_trace: World::intersect + hit() for every ray of generation `gen`.  With `primary` set (generation 0
// of a frame) the rays come from the camera instead of the pool.
// ------------------------------------------------------------------------------------------------
template <class R> struct HitRay;
struct SpawnAt;
template <class R, int MODE, bool PACKET, bool SPAWN>
RT_FN void shade_surface(const DevScene<R>& sc, const RayPool<R>& pool, uint32_t r, bool valid, int p, int flags, const HitRay<R>& h, int* lds, Work& wk, const SpawnAt* at);

// One chunk of BLOCK rays of generation `gen`: World::intersect + hit() for the ray of every lane, records to
// the pool.  With `primary` set (generation 0 of a frame) the rays come from the camera instead of the pool.
// LAST: this is the deepest generation.  Its rays spawn nothing and nothing waits for their hit records
// except their own shading, so the chunk goes on to shade each hit (shadow ray + lighting) and leaves the
// ray's final colour in pool.color: one launch, and one slowest-wave tail, instead of three.
// wh / wc: this wave's hits and the upper bound of the child rays they spawn.
template <class R, int MODE, bool PACKET, bool LAST>
RT_FN void trace_chunk(const DevScene<R>& sc, const RayPool<R>& pool, FrameCtl* ctl, const DevCamera<R>& cam, int primary, uint32_t start, uint32_t count,
                       uint32_t base, int remaining, int launch, int* lds, Work& wk, uint32_t& wh, uint32_t& wc) {
  const uint32_t i = base + threadIdx.x;
  const bool valid = i < count;
  const uint32_t r = start + (valid ? i : 0u);
  V3<R> o, d;
  if (primary) {  // generation 0 of a frame: the rays are made here; the direction is kept for the later kernels, the
    primary_ray(cam, r, o, d);  // origin is the view's (ray_origin)
    if (valid) st3(pool.dir + 3ull * r, d);
  } else {
    o = ray_origin(pool, r), d = ld3(pool.dir + 3ull * r);
  }
  ClosestVis<R, MODE> vis(sc, o, d, wk);
  const bool live = valid && !(d.x == R(0) && d.y == R(0) && d.z == R(0));
  long long t_start = 0;
  if constexpr (MODE & M_COUNT) t_start = clock64();
#ifdef RT355_EXP_STAMPS
  const unsigned long long dbg_t0 = wall_clock64(), dbg_c0 = __builtin_amdgcn_s_memtime();
  wk.dbg = DbgAcc{};
#endif
  if constexpr (PACKET) {
    traverse_packet<R, MODE>(sc, o, d, live, vis, wk, lds);
  } else {
    if (live) traverse<R, MODE>(sc, o, d, vis, wk, lds);
  }
#ifdef RT355_EXP_STAMPS
  dbg_record(launch, dbg_t0, dbg_c0, wk.dbg, live);
#endif
  if constexpr (MODE & M_COUNT) {
    // longest traversal of one wave, in shader clocks (all lanes reconverge here)
    const unsigned long long dt = (unsigned long long)(clock64() - t_start);
    if ((threadIdx.x & 63u) == 0 && launch < RT355_MAX_LAUNCH) atomicMax(&ctl->work[launch][11], dt);
  }
  uint32_t n_child = 0;
  const bool hit = valid && vis.best_p >= 0;
  if (valid) {
    store_hit(pool, r, vis.best_p, vis.best_flags);
    if (hit) {
      if constexpr (!LAST) {  // the deepest generation shades its hits right here, from registers
        pool.hit_t[r] = vis.best_t;
        if ((vis.best_flags & PF_KIND_MASK) == RT355_SMOOTH_TRIANGLE)  // the only kind whose normal reads u, v (src/shape.rs:199-203)
          pool.hit_uv[2ull * r] = vis.best_u, pool.hit_uv[2ull * r + 1] = vis.best_v;
      }
      if (remaining > 0) n_child = ((vis.best_flags & PF_REFLECTIVE) ? 1u : 0u) + ((vis.best_flags & PF_TRANSPARENT) ? 1u : 0u);
    }
  }
  if constexpr (LAST) {
    if (__ballot(hit) != 0ull) {
      const HitRay<R> h{o, d, vis.best_t, vis.best_u, vis.best_v};
      shade_surface<R, MODE, PACKET, false>(sc, pool, r, hit, hit ? vis.best_p : 0, vis.best_flags, h, lds, wk, nullptr);
    }
    if (valid && !hit) st3(pool.color + 3ull * r, V3<R>{R(0), R(0), R(0)});  // src/world.rs:61-63: a miss is black
  }
  // children of the wave's hits: a reflected ray per reflective hit, a refracted one per transparent hit
  wh = __popcll(__builtin_amdgcn_ballot_w64(hit)), wc = __popcll(__builtin_amdgcn_ballot_w64((n_child & 1u) != 0)) + 2u * __popcll(__builtin_amdgcn_ballot_w64((n_child & 2u) != 0));
  // bit 31 of the hit count: some hit of the wave is on a transparent shape and may spawn (k_spawn<..., ONLY_T> looks at
  // this word before it looks at any hit record)
  if (__builtin_amdgcn_ballot_w64(hit && remaining > 0 && (vis.best_flags & PF_TRANSPARENT)) != 0ull) wh |= WAVE_HAS_TRANSPARENT;
}

// k_trace: trace_chunk over generation `gen`, one chunk per block and trip.  Every wave leaves the hit count and the
// child bound of its own 64 rays for k_scan (entry i / 64 of blk_hits / blk_children): no wave waits for another.
template <class R, int MODE, bool PACKET, bool LAST>
__global__ void __launch_bounds__(BLOCK, PACKET ? (LAST ? RT355_WPS_PACKET_LAST : RT355_WPS_PACKET) : RT355_WPS_LANE) k_trace(DevScene<R> sc, RayPool<R> pool, FrameCtl* ctl, int gen, int remaining, int launch, DevCamera<R> cam, int primary) {
  __shared__ int lds_stack[PACKET ? PARK_INTS : LDS_STACK * BLOCK];  // packets: shade_surface parks two vectors per lane here
  int* lds = lds_stack + (PACKET ? 0 : threadIdx.x);
  uint32_t start, count;
  gen_range(ctl, gen, pool.capacity, start, count);
  Work wk;
  // one ray per lane per trip; the trip count is the same for the 64 lanes of a wave, so they stay converged for the
  // ballots of the packet traversal
  for (uint32_t base = blockIdx.x * BLOCK; base < count; base += gridDim.x * BLOCK) {
    uint32_t wh, wc;
    trace_chunk<R, MODE, PACKET, LAST>(sc, pool, ctl, cam, primary, start, count, base, remaining, launch, lds, wk, wh, wc);
    if ((threadIdx.x & 63u) == 0 && base + threadIdx.x < count) {
      const uint32_t e = (base + threadIdx.x) / GROUP;
      pool.blk_hits[e] = wh;
      pool.blk_children[e] = wc;
    }
  }
  flush_work<(MODE & M_COUNT) != 0>(ctl, launch, wk);
}

// k_trace with the quad traversal: a wave takes 64 consecutive rays per trip, 16 at a time, four lanes to a ray.
// Same outputs as k_trace (hit records, per-wave hit and child counts for k_scan).
template <class R, int MODE>
__global__ void __launch_bounds__(BLOCK, 4) k_trace_q(DevScene<R> sc, RayPool<R> pool, FrameCtl* ctl, int gen, int remaining, int launch, DevCamera<R> cam, int primary) {
  __shared__ int lds_stack[QUADS * STACK_DEPTH];
  int* stack = lds_stack + (threadIdx.x >> 2);
  uint32_t start, count;
  gen_range(ctl, gen, pool.capacity, start, count);
  Work wk;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  for (uint32_t base = blockIdx.x * BLOCK; base < count; base += gridDim.x * BLOCK) {
    uint32_t wh = 0, wc = 0, wt = 0;  // this wave's hits and child-ray bound over the four passes; the flag of trace_chunk
    for (int sub = 0; sub < 4; ++sub) {
      const uint32_t i = base + wave * 64u + (uint32_t)sub * 16u + (lane >> 2);
      const bool valid = i < count;
      const uint32_t r = start + (valid ? i : 0u);
      const bool writer = (threadIdx.x & 3u) == 0;
      V3<R> o, d;
      if (primary) {
        primary_ray(cam, r, o, d);
        if (valid && writer) st3(pool.dir + 3ull * r, d);
      } else {
        o = ray_origin(pool, r), d = ld3(pool.dir + 3ull * r);
      }
      ClosestVis<R, MODE> vis(sc, o, d, wk);
      const bool live = valid && !(d.x == R(0) && d.y == R(0) && d.z == R(0));  // the same in the four lanes of a quad
      if (live) traverse_quad<R, MODE>(sc, o, d, vis, wk, stack);
      const bool mine = vis.quad_winner() && valid;  // one lane per ray
      const bool hit = mine && vis.best_p >= 0;
      if (mine) {
        store_hit(pool, r, vis.best_p, vis.best_flags);
        if (hit) {
          pool.hit_t[r] = vis.best_t;
          if ((vis.best_flags & PF_KIND_MASK) == RT355_SMOOTH_TRIANGLE)
            pool.hit_uv[2ull * r] = vis.best_u, pool.hit_uv[2ull * r + 1] = vis.best_v;
        } else if (remaining == 0) {
          // deepest generation: no combine launch of its own will visit this ray, its parent reads the colour
          // as it is (src/world.rs:61-63: a miss is black)
          st3(pool.color + 3ull * r, V3<R>{R(0), R(0), R(0)});
        }
      }
      const bool spawns = hit && remaining > 0;
      wh += __popcll(__ballot(hit));
      wc += __popcll(__ballot(spawns && (vis.best_flags & PF_REFLECTIVE))) + __popcll(__ballot(spawns && (vis.best_flags & PF_TRANSPARENT)));
      if (__ballot(spawns && (vis.best_flags & PF_TRANSPARENT)) != 0ull) wt = WAVE_HAS_TRANSPARENT;
    }
    if (lane == 0 && base + wave * 64u < count) {
      const uint32_t e = (base + wave * 64u) / GROUP;
      pool.blk_hits[e] = wh | wt;
      pool.blk_children[e] = wc;
    }
  }
  flush_work<(MODE & M_COUNT) != 0>(ctl, launch, wk);
}

// Frame control block back to zero, generation 0 sized: a launch instead of a host-to-device copy, which
// from pageable memory would stall the host for longer than a small frame takes to render.
static __global__ void __launch_bounds__(256) k_reset_ctl(FrameCtl* ctl, uint32_t n_primary) {
  uint32_t* w = (uint32_t*)ctl;
  for (uint32_t i = threadIdx.x; i < sizeof(FrameCtl) / 4; i += blockDim.x) w[i] = 0;
  __syncthreads();
  if (threadIdx.x == 0) ctl->gen_count[0].v = n_primary;
}

// ------------------------------------------------------------------------------------------------
// k_scan: exclusive prefix of the per-block child bounds of generation `gen` (no atomics on the data), plus the
// generation's hit total and the size of generation gen+1.  Child-ray ids become a pure function of the ray
// order, so a frame always lays its rays out the same way.
// Two levels in one launch: a block scans chunks of SCAN_CHUNK entries -- four consecutive entries per thread, one
// 16-byte load each, so a wave reads 1 KiB in a row -- and leaves the prefixes inside each chunk in blk_base and the
// chunk's totals in blk_top; the block that finishes last (one ticket per block, a few dozen in all) scans the chunk
// totals.  A consumer adds blk_top's prefix of its chunk to blk_base (child_base()).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t SCAN_CHUNK = 4096;  // entries per block: 1024 threads x 4

RT_FN uint32_t child_base(const uint32_t* blk_base, const uint32_t* blk_top, uint32_t k) { return blk_base[k] + blk_top[3u * (k / SCAN_CHUNK) + 2u]; }

// inclusive scan over the 1024 threads of the block; *total = sum over the block.  s: 16 words of LDS.
RT_FN uint32_t block_scan_1024(uint32_t v, uint32_t* s, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if ((int)lane >= off) inc += t;
  }
  if (lane == 63u) s[wave] = inc;
  __syncthreads();
  uint32_t before = 0, all = 0;
  for (uint32_t w = 0; w < 16u; ++w) {
    const uint32_t x = s[w];
    before += w < wave ? x : 0u;
    all += x;
  }
  __syncthreads();
  *total = all;
  return before + inc;
}

static __global__ void __launch_bounds__(1024) k_scan(const uint32_t* blk_hits, const uint32_t* blk_children, uint32_t* blk_base, uint32_t* blk_top, FrameCtl* ctl, int gen, long long capacity) {
  __shared__ uint32_t s_scan[16];
  __shared__ uint32_t s_last;
  uint32_t start, count;
  gen_range(ctl, gen, capacity, start, count);
  const uint32_t n = (count + GROUP - 1) / GROUP;
  const uint32_t nchunk = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  // the grid is fixed (the generation's size is known on the device only): a block takes the chunks
  // blockIdx.x, blockIdx.x + gridDim.x, ... -- one each for all but the largest generations
  for (uint32_t chunk = blockIdx.x; chunk < nchunk; chunk += gridDim.x) {  // block-uniform
    const uint32_t k0 = chunk * SCAN_CHUNK + threadIdx.x * 4u;
    uint32_t c[4] = {0, 0, 0, 0}, h[4] = {0, 0, 0, 0};
    if (k0 + 3u < n) {  // the arrays are 16-byte aligned and k0 is a multiple of 4
      const uint4 cv = *(const uint4*)(blk_children + k0), hv = *(const uint4*)(blk_hits + k0);
      c[0] = cv.x, c[1] = cv.y, c[2] = cv.z, c[3] = cv.w, h[0] = hv.x, h[1] = hv.y, h[2] = hv.z, h[3] = hv.w;
    } else {
      for (uint32_t j = 0; j < 4u; ++j)
        if (k0 + j < n) c[j] = blk_children[k0 + j], h[j] = blk_hits[k0 + j];
    }
    for (uint32_t j = 0; j < 4u; ++j) h[j] &= ~WAVE_HAS_TRANSPARENT;  // the flag above the count is k_spawn's
    const uint32_t csum = c[0] + c[1] + c[2] + c[3], hsum = h[0] + h[1] + h[2] + h[3];
    uint32_t ctot, htot;
    const uint32_t cinc = block_scan_1024(csum, s_scan, &ctot);
    (void)block_scan_1024(hsum, s_scan, &htot);
    uint32_t run = cinc - csum;
    if (k0 + 3u < n) {
      *(uint4*)(blk_base + k0) = make_uint4(run, run + c[0], run + c[0] + c[1], run + c[0] + c[1] + c[2]);
    } else {
      for (uint32_t j = 0; j < 4u; ++j)
        if (k0 + j < n) blk_base[k0 + j] = run, run += c[j];
    }
    if (threadIdx.x == 0) {  // chunk totals to memory every block can read
      __hip_atomic_store(&blk_top[3u * chunk], ctot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&blk_top[3u * chunk + 1u], htot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // one ticket per block; the block that draws the last one scans the chunk totals
  if (threadIdx.x == 0) {
    __threadfence();
    const uint32_t ticket = __hip_atomic_fetch_add(&ctl->scan_ticket[gen].v, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == gridDim.x - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  const uint32_t per = (nchunk + 1023u) / 1024u;
  const uint32_t lo = threadIdx.x * per, hi = lo + per < nchunk ? lo + per : nchunk;
  uint32_t mc = 0, mh = 0;
  for (uint32_t k = lo; k < hi; ++k) {
    mc += __hip_atomic_load(&blk_top[3u * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mh += __hip_atomic_load(&blk_top[3u * k + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  uint32_t all_c, all_h;
  const uint32_t pc = block_scan_1024(mc, s_scan, &all_c);
  (void)block_scan_1024(mh, s_scan, &all_h);
  uint32_t prefix = pc - mc;
  for (uint32_t k = lo; k < hi; ++k) {
    blk_top[3u * k + 2u] = prefix;
    prefix += __hip_atomic_load(&blk_top[3u * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) {
    ctl->hit_count[gen].v = all_h;
    unsigned long long total = all_c;
    const unsigned long long room = (unsigned long long)capacity - ((unsigned long long)start + count);
    if (total > room) total = room, ctl->overflow.v = 1;
    ctl->gen_count[gen + 1].v = (uint32_t)total;
  }
}

// ------------------------------------------------------------------------------------------------
// Shading helpers
// ------------------------------------------------------------------------------------------------
// Shape::normal / local_normal (src/shape.rs:111-206).  Only the sphere's normal is taken back to world
// space and normalised; every other shape returns its object-space normal as is, unnormalised for cube /
// cylinder side / cone side.
template <class R>
RT_FN V3<R> normal_at(const DevScene<R>& sc, int p, int flags, V3<R> point, R u, R v) {
  const int kind = flags & PF_KIND_MASK;
  const R EPS = eps<R>();
  if (kind == RT355_SMOOTH_TRIANGLE) {
    const DevTriNormals<R>& N = sc.trin[p];
    return normalize(ld3(N.n2) * u + ld3(N.n3) * v + ld3(N.n1) * (R(1) - u - v));
  }
  if (kind == RT355_TRIANGLE) return ld3(sc.trin[p].nf);
  if (kind == RT355_PLANE) return mk<R>(R(0), R(1), R(0));
  const DevXform<R>& X = sc.xform[p];
  const V3<R> lp = xf_point(X.r, point);
  switch (kind) {
    case RT355_SPHERE: {
      const R* r = X.r;  // transpose(inverse) * local_point, xyz only
      V3<R> n = {r[0] * lp.x + r[4] * lp.y + r[8] * lp.z, r[1] * lp.x + r[5] * lp.y + r[9] * lp.z,
                 r[2] * lp.x + r[6] * lp.y + r[10] * lp.z};
      return normalize(n);
    }
    case RT355_CUBE: {
      const R ax = fabs(lp.x), ay = fabs(lp.y), az = fabs(lp.z);
      const R maxc = fmax(fmax(ax, ay), az);
      if (maxc == ax) return mk<R>(lp.x, R(0), R(0));
      if (maxc == ay) return mk<R>(R(0), lp.y, R(0));
      return mk<R>(R(0), R(0), lp.z);
    }
    case RT355_CYLINDER: {
      const R dist = lp.x * lp.x + lp.z * lp.z;
      if (dist < R(1) && lp.y >= (X.ymax - EPS)) return mk<R>(R(0), R(1), R(0));
      if (dist < R(1) && lp.y <= (X.ymin + EPS)) return mk<R>(R(0), R(-1), R(0));
      return mk<R>(lp.x, R(0), lp.z);
    }
    case RT355_CONE: {
      const R dist = lp.x * lp.x + lp.z * lp.z;
      if (dist < X.ymax * X.ymax && lp.y >= X.ymax - EPS) return mk<R>(R(0), R(1), R(0));
      if (dist < X.ymin * X.ymin && lp.y <= X.ymin + EPS) return mk<R>(R(0), R(-1), R(0));
      return mk<R>(lp.x, lp.y > R(0) ? -sqrt(dist) : sqrt(dist), lp.z);
    }
    default:
      return mk<R>(R(0), R(0), R(0));
  }
}

// x % 2.0 (Rust's float remainder = fmod, src/pattern.rs:100-117) without the library's general remainder loop, which
// takes over a hundred instructions in double.  x / 2 is exact in binary floating point, so is its truncation, so is
// twice that, and the difference is exact because it is smaller than x and on x's grid: x - 2 * trunc(x / 2) IS
// fmod(x, 2) for every finite x, up to the sign of a zero result (the callers compare with zero or take the floor of a
// non-negative value); an infinite or NaN x gives NaN either way.
template <class R>
RT_FN R fmod2(R x) { return x - R(2) * trunc(x * R(0.5)); }

// Pattern::pattern_at (src/pattern.rs:93-121): evaluated on the world-space point, no transformation.
template <class R>
RT_FN V3<R> pattern_at(const DevMaterial<R>& m, V3<R> p) {
  const V3<R> c1 = ld3(m.c1), c2 = ld3(m.c2);
  switch (m.pattern_kind) {
    case RT355_PATTERN_STRIPE:
      return fmod2(floor(p.x)) == R(0) ? c1 : c2;
    case RT355_PATTERN_GRADIENT: {
      const R f = p.x - floor(p.x);
      const V3<R> dd = c2 - c1;
      return {c1.x + f * dd.x, c1.y + f * dd.y, c1.z + f * dd.z};
    }
    case RT355_PATTERN_RING:
      return floor(fmod2(sqrt(p.x * p.x + p.z * p.z))) == R(0) ? c1 : c2;
    case RT355_PATTERN_CHECKER:
      return fmod2(floor(p.x) + floor(p.y) + floor(p.z)) == R(0) ? c1 : c2;
    case RT355_PATTERN_TEST:
      return p;
    default:
      return ld3(m.color);
  }
}

// powf(x, y) of the specular term (src/material.rs:157), x > 0.  A shininess that is a whole number -- the default
// 200.0 (src/material.rs:31-72) and every value the reference's scenes use -- is raised by repeated squaring: a
// dozen multiplications where the general pow takes over two hundred instructions in double.  Every lane of the
// wave takes the same path, so the general case costs nothing when no lane needs it.
template <class R>
RT_FN R pow_shininess(R x, R y) {
  const bool whole = y >= R(0) && y <= R(65535) && y == floor(y);
  R result = R(1);
  if (whole) {
    uint32_t n = (uint32_t)y;
    R b = x;
    while (n) {
      if (n & 1u) result *= b;
      b *= b;
      n >>= 1;
    }
  } else {
    result = pow(x, y);
  }
  return result;
}

// Material::lighting (src/material.rs:108-162)
// lightv_known: normalize(light - point) when the caller holds it already (the direction of the shadow ray it has just
// traced from the same point is that very expression, src/world.rs:46-49), else null.
template <class R>
RT_FN V3<R> lighting(const DevScene<R>& sc, const DevMaterial<R>& m, V3<R> point, V3<R> eyev, V3<R> normalv, bool shadowed, const V3<R>* lightv_known = nullptr) {
  const V3<R> c = m.pattern_kind != RT355_PATTERN_NONE ? pattern_at(m, point) : ld3(m.color);
  const V3<R> li = ld3(sc.light_int);
  const V3<R> eff = {c.x * li.x, c.y * li.y, c.z * li.z};
  const V3<R> ambient = eff * m.ambient;
  if (shadowed) return ambient;
  const V3<R> lightv = lightv_known ? *lightv_known : normalize(ld3(sc.light_pos) - point);
  const R ldn = dot(lightv, normalv);
  V3<R> diffuse = {R(0), R(0), R(0)}, specular = {R(0), R(0), R(0)};
  if (ldn >= R(0)) {
    const R k = m.diffuse * ldn;
    diffuse = eff * k;
    const V3<R> rv = reflect(R(-1) * lightv, normalv);
    const R rde = dot(rv, eyev);
    if (rde > R(0)) {
      const R factor = pow_shininess(rde, m.shininess);
      specular = li * (factor * m.specular);
    }
  }
  return ambient + diffuse + specular;
}

// ------------------------------------------------------------------------------------------------
// The part of prepare_computation (src/intersection.rs:60-85) that needs no intersection list.  It is
// evaluated from the ray and hit records wherever it is needed (k_shadow, k_spawn, k_combine) instead of
// being carried in registers or in memory: a few dozen flops against a dozen live doubles.
// ------------------------------------------------------------------------------------------------
template <class R>
struct Prepared {
  V3<R> d, eyev, normalv, normalv_temp, over_point, under_point;
};
// a ray and its hit, as prepare() needs them
template <class R>
struct HitRay {
  V3<R> o, d;
  R t, u, v;
};
template <class R>
RT_FN HitRay<R> load_hit_ray(const RayPool<R>& pool, uint32_t r, int flags) {
  HitRay<R> h;
  h.o = ray_origin(pool, r);
  h.d = ld3(pool.dir + 3ull * r);
  h.t = pool.hit_t[r];
  h.u = R(0), h.v = R(0);
  if ((flags & PF_KIND_MASK) == RT355_SMOOTH_TRIANGLE) h.u = pool.hit_uv[2ull * r], h.v = pool.hit_uv[2ull * r + 1];
  return h;
}
template <class R>
RT_FN Prepared<R> prepare(const DevScene<R>& sc, const HitRay<R>& h, int p, int flags) {
  Prepared<R> c;
  const V3<R> o = h.o;
  c.d = h.d;
  const R t = h.t, u = h.u, v = h.v;
  const V3<R> point = o + c.d * t;  // src/ray.rs:22-24
  c.eyev = R(-1) * c.d;
  c.normalv_temp = normal_at(sc, p, flags, point, u, v);
  const bool inside = dot(c.normalv_temp, c.eyev) < R(0);
  c.normalv = inside ? R(-1) * c.normalv_temp : R(1) * c.normalv_temp;
  const V3<R> offset = eps<R>() * c.normalv;
  c.over_point = point + offset;
  c.under_point = point - offset;
  return c;
}

// The shadow ray of a point (src/world.rs:46-49): direction to the light, normalised, and the parameter limit
// distance + EPSILON (src/world.rs:52).
template <class R>
RT_FN void light_ray(const DevScene<R>& sc, V3<R> point, V3<R>& ldir, R& limit) {
  // the same expression as normalize(): lighting() may take ldir for its lightv
  const V3<R> lv = ld3(sc.light_pos) - point;
  const R dd = dot(lv, lv);
  const R inv = inv_sqrt(dd);
  ldir = {lv.x * inv, lv.y * inv, lv.z * inv};
  limit = dd * inv + eps<R>();  // the distance (src/world.rs:47), within an ulp or two, + EPSILON
}

// Where the child rays of a wave's hits go: generation gen + 1 starts at child_start, this wave's children at
// child_start + base (k_scan's prefix), reflected before refracted inside the wave.
struct SpawnAt {
  FrameCtl* ctl;
  uint32_t child_start, base;
};

// The reflected ray of a hit (src/world.rs:68-76, src/intersection.rs:85) into its slot of the next generation, and
// the hit's child record as far as it is known here.  Called by every lane of the wave; slot_r / slot_t mark the
// lanes whose hit is on a reflective / transparent shape.  A transparent hit's refracted ray, its n1 / n2 and its
// Schlick weight are k_spawn's part (ONLY_TRANSPARENT): it fills the other half of the child record.
template <class R>
RT_FN void spawn_reflection(const RayPool<R>& pool, uint32_t r, bool slot_r, bool slot_t, const Prepared<R>& c, const SpawnAt& at) {
  const unsigned long long m_r = __builtin_amdgcn_ballot_w64(slot_r);
  const uint32_t off_r = __popcll(m_r & lanes_below());
  int32_t child_r = -1;
  if (slot_r) {
    const unsigned long long id = (unsigned long long)at.child_start + at.base + off_r;
    if (id < (unsigned long long)pool.capacity) {
      child_r = (int32_t)id;
      st3(pool.org + 3ull * id, c.over_point);
      st3(pool.dir + 3ull * id, reflect(c.d, c.normalv_temp));  // un-flipped normal (src/intersection.rs:85)
    } else {
      at.ctl->overflow.v = 1;
    }
  }
  if (slot_r || slot_t) {
    pool.child[2ull * r] = child_r;
    if (!slot_t) pool.child[2ull * r + 1] = -1;  // the Schlick weight is read only for prims flagged PF_SCHLICK, which are transparent
  }
}

// ------------------------------------------------------------------------------------------------
// shade_surface: is_shadowed (src/world.rs:45-57) of the hit of ray r, then Material::lighting
// (src/material.rs:108-162) with that flag; the surface colour goes to pool.color[r].  Called by all 64
// lanes (the packet traversal needs them); `valid` marks the lanes whose ray hit prim p.
// SPAWN: the reflected rays of the hits are written on the way (spawn_reflection), from the same prepared hit.
// ------------------------------------------------------------------------------------------------
template <class R, int MODE, bool PACKET, bool SPAWN>
RT_FN void shade_surface(const DevScene<R>& sc, const RayPool<R>& pool, uint32_t r, bool valid, int p, int flags, const HitRay<R>& h, int* lds, Work& wk, const SpawnAt* at) {
  Prepared<R> pc;
  pc.over_point = {R(0), R(0), R(0)};
  if (valid) pc = prepare(sc, h, p, flags);
  if constexpr (SPAWN) spawn_reflection(pool, r, valid && (flags & PF_REFLECTIVE), valid && (flags & PF_TRANSPARENT), pc, *at);
  V3<R> ldir;
  R limit;
  light_ray(sc, pc.over_point, ldir, limit);
  AnyVis<R, MODE> av(sc, pc.over_point, ldir, limit, wk);
  V3<R> eyev = pc.eyev, normalv = pc.normalv;
#ifndef RT355_EXP_NO_SHADOW_TRAV
  if constexpr (PACKET) {
    // The packet kernels run at the register budget of four waves per SIMD and the traversal fills it: what only
    // the lighting needs waits in LDS meanwhile (one column per lane, so no bank conflicts and no barrier) instead
    // of being spilled to scratch memory by the compiler.
    R* park = reinterpret_cast<R*>(lds) + (threadIdx.x & (BLOCK - 1));
#ifndef RT355_EXP_NO_PARK
    park[0 * BLOCK] = eyev.x, park[1 * BLOCK] = eyev.y, park[2 * BLOCK] = eyev.z;
    park[3 * BLOCK] = normalv.x, park[4 * BLOCK] = normalv.y, park[5 * BLOCK] = normalv.z;
#endif
    traverse_packet<R, MODE>(sc, pc.over_point, ldir, valid, av, wk, nullptr);
#ifndef RT355_EXP_NO_PARK
    eyev = {park[0 * BLOCK], park[1 * BLOCK], park[2 * BLOCK]};
    normalv = {park[3 * BLOCK], park[4 * BLOCK], park[5 * BLOCK]};
#endif
  } else {
    if (valid) traverse<R, MODE>(sc, pc.over_point, ldir, av, wk, lds);
  }
#endif
  if (valid) {
#ifdef RT355_EXP_NO_LIGHT
    st3(pool.color + 3ull * r, av.found ? pc.over_point : ldir);
#else
    // the whole record at once: read field by field as lighting() comes to them, every field is a memory round trip
    // of its own behind the branch that needs it (pattern kind, then its colours, then ambient, diffuse, shininess ...)
    const DevMaterial<R> m = load_whole(&sc.materials[flags >> PF_MAT_SHIFT]);
    st3(pool.color + 3ull * r, lighting(sc, m, pc.over_point, eyev, normalv, av.found, &ldir));
#endif
  }
}

// ------------------------------------------------------------------------------------------------
// container_indices: n1 / n2 of prepare_computation (src/intersection.rs:86-127) for a hit on a transparent
// shape.  The reference walks the whole sorted intersection list of the ray; what decides n1 / n2 is only
// (a) how many of the hit shape's own intersections come before the hit and (b) the last shape entered and
// not left before the hit, which is found by walking BACKWARDS from the hit: the nearest intersection
// behind it whose shape has an odd number of intersections before the hit.
// Called by every lane of the wave (the packet traversal needs them all); `need` marks the lanes that
// want an answer.
// ------------------------------------------------------------------------------------------------
template <class R, int MODE, bool PACKET>
RT_FN void container_indices(const DevScene<R>& sc, bool need, int p, PrimInfo info, V3<R> o, V3<R> d, R t_hit, int* lds, Work& wk, R& n1, R& n2) {
  const R t = need ? t_hit : R(1);
  // the hit shape's own intersections, emission order
  R h0 = R(0), h1 = R(0), h2 = R(0), h3 = R(0);
  int nh = 0;
  if (need)
    prim_hits<R, MODE>(sc, p, info.flags, o, d, wk, [&](R tt, R, R, int em) {
      if (em == 0) h0 = tt; else if (em == 1) h1 = tt; else if (em == 2) h2 = tt; else if (em == 3) h3 = tt;
      nh = em + 1;
    });
  const R ht[4] = {h0, h1, h2, h3};
  // the entry the reference treats as "the hit" last: same t, latest emission
  int m_last = 0;
  R best = inf<R>();
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < nh) {
      const R dd = fabs(ht[k] - t);
      if (dd <= best) best = dd, m_last = k;
    }
  R tm = t;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (nh && k == m_last) tm = ht[k];
  int c_before = 0;
  R hmt = R(0);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < nh && (ht[k] < tm || (ht[k] == tm && k < m_last))) {
      if (c_before == 0 || ht[k] >= hmt) hmt = ht[k];
      ++c_before;
    }
  // the container's last element among the OTHER shapes: walk backwards from the hit.
  // When every other shape has refractive index 1 the walk only has to tell whether some shape was
  // entered AFTER the hit shape's own last event (then n1 is that shape's 1 and not the hit shape's
  // index): nothing to look for when the ray is entering the hit shape, and nothing behind that event
  // otherwise.
  const bool h_before = (c_before & 1) != 0;
  const bool others_unit = (info.flags & PF_OTHERS_UNIT) != 0;
  const R floor_t = (others_unit && h_before) ? hmt : -inf<R>();
  const int floor_s = (others_unit && h_before) ? info.shape : -1;
  // Reversed ray: 0 - d, so that a zero component stays +0.0 (negation would make it -0.0, and the cube's
  // slab test, which branches on d >= 0 (src/shape.rs:455-466), tells the two apart).
  const V3<R> rd = {R(0) - d.x, R(0) - d.y, R(0) - d.z};
  // nor when the hit shape's box holds no intersection of any other prim (PF_ISOLATED, decided at upload): the stretch
  // between two of the shape's own intersections lies inside that box
  const bool isolated = (info.flags & PF_ISOLATED) != 0;
  bool searching = need && !(others_unit && (!h_before || isolated)), found = false;
  int other_flags = 0, other_s = -1;
  R other_t = R(0);
  R bound_t = t;
  int bound_s = info.shape;
  for (int step = 0; step < 4096; ++step) {
    if (PACKET ? (__ballot(searching) == 0ull) : !searching) break;
    BackwardVis<R, MODE> bv(sc, o, rd, p, bound_t, bound_s, floor_t, floor_s, wk);
    if constexpr (PACKET) {
      traverse_packet<R, MODE>(sc, o, rd, searching, bv, wk, lds);
    } else {
      if (searching) traverse<R, MODE>(sc, o, rd, bv, wk, lds);
    }
    if (searching) {
      if (!bv.have) {
        searching = false;
      } else {
        // is that shape present, i.e. does it have an odd number of intersections before the hit?
        int cnt = 0;
        prim_hits<R, MODE>(sc, bv.cp, bv.cflags, o, d, wk, [&](R tt, R, R, int) {
          if (tt < t || (tt == t && bv.cs < info.shape)) ++cnt;
        });
        if (cnt & 1) {
          found = true, other_flags = bv.cflags, other_s = bv.cs, other_t = bv.ct;
          searching = false;
        } else {
          bound_t = bv.ct, bound_s = bv.cs;
        }
      }
    }
  }
  const R own = sc.materials[info.flags >> PF_MAT_SHIFT].refractive;
  const R other = found ? sc.materials[other_flags >> PF_MAT_SHIFT].refractive : R(1);
  const bool own_last = h_before && (!found || hmt > other_t || (hmt == other_t && info.shape > other_s));
  n1 = own_last ? own : other;
  n2 = ((c_before + 1) & 1) ? own : other;
}

// ------------------------------------------------------------------------------------------------
// k_shadow: is_shadowed (src/world.rs:45-57) and the surface colour (Material::lighting) of every hit of
// generation `gen`, left in pool.color.  Nothing on the way to the next generation depends on them (only
// k_combine reads them), so the host runs these launches on a second stream, beside the trace / spawn
// chain of the later generations.
// One thread per ray; lanes whose ray missed idle.
// ------------------------------------------------------------------------------------------------
// SPAWN: the launch also writes the reflected rays of the generation's hits (generation gen + 1): hit record, ray and
// prepared hit are read and made once for both.  Chains that overlap with others run this way; k_spawn is then left
// with the transparent hits alone.
template <class R, int MODE, bool PACKET, bool SPAWN>
__global__ void __launch_bounds__(BLOCK, PACKET ? RT355_WPS_SHADOW_PACKET : RT355_WPS_LANE) k_shadow(DevScene<R> sc, RayPool<R> pool, FrameCtl* ctl, int gen, int launch) {
  uint32_t start, gcount;
  gen_range(ctl, gen, pool.capacity, start, gcount);
  __shared__ int lds_stack[PACKET ? PARK_INTS : LDS_STACK * BLOCK];  // packets: shade_surface parks two vectors per lane here
  int* lds = lds_stack + (PACKET ? 0 : threadIdx.x);
  Work wk;
  // the hit record of the next trip is requested before this trip's work starts: its latency (a round trip to HBM)
  // passes behind the traversal instead of in front of the next one
  const uint32_t stride = gridDim.x * BLOCK;
  uint32_t rbase = blockIdx.x * BLOCK;
  HitRec next = HitRec{-1, 0};
  if (rbase < gcount) next = load_hit(pool, start + (rbase + threadIdx.x < gcount ? rbase + threadIdx.x : 0u));
  for (; rbase < gcount; rbase += stride) {
    const uint32_t i = rbase + threadIdx.x;
    const uint32_t r = start + (i < gcount ? i : 0u);
    const HitRec hit = next;
    if (rbase + stride < gcount) next = load_hit(pool, start + (i + stride < gcount ? i + stride : 0u));
    const bool valid = i < gcount && hit.prim >= 0;
    if (__ballot(valid) != 0ull) {  // wave-uniform
      HitRay<R> h{};
      if (valid) h = load_hit_ray(pool, r, hit.flags);
      SpawnAt at{ctl, start + gcount, 0u};
      if constexpr (SPAWN) at.base = child_base(pool.blk_base, pool.blk_top, __builtin_amdgcn_readfirstlane(i) / GROUP);
      shade_surface<R, MODE, PACKET, SPAWN>(sc, pool, r, valid, valid ? hit.prim : 0, hit.flags, h, lds, wk, &at);
    }
  }
  flush_work<(MODE & M_COUNT) != 0>(ctl, launch, wk);
}

// k_shadow with the quad traversal.  Per trip a wave takes 64 rays: (1) one lane per ray prepares the shadow
// ray of its hit and leaves it in LDS, the lanes with a hit are listed compactly; (2) the wave walks that list
// 16 shadow rays at a time, four lanes to a ray; (3) one lane per ray again: lighting with the flag found.
// Everything a wave exchanges through LDS is its own (no block barrier); LDS operations of one wave execute
// in order, the fences only keep the compiler from moving them.
template <class R, int MODE>
__global__ void __launch_bounds__(BLOCK, 4) k_shadow_q(DevScene<R> sc, RayPool<R> pool, FrameCtl* ctl, int gen, int launch) {
  uint32_t start, gcount;
  gen_range(ctl, gen, pool.capacity, start, gcount);
  __shared__ int lds_stack[QUADS * STACK_DEPTH];
  __shared__ R s_ray[BLOCK * 7];  // over_point, direction to the light, distance + EPSILON
  __shared__ uint8_t s_list[BLOCK], s_found[BLOCK];
  int* stack = lds_stack + (threadIdx.x >> 2);
  const uint32_t lane = threadIdx.x & 63u, wbase = threadIdx.x & ~63u;
  Work wk;
  for (uint32_t rbase = blockIdx.x * BLOCK; rbase < gcount; rbase += gridDim.x * BLOCK) {
    const uint32_t i = rbase + threadIdx.x;
    const uint32_t r = start + (i < gcount ? i : 0u);
    const HitRec hit = load_hit(pool, r);
    const bool valid = i < gcount && hit.prim >= 0;
    const unsigned long long m = __ballot(valid);
    if (m == 0ull) continue;  // wave-uniform
    const int p = valid ? hit.prim : 0;
    Prepared<R> pc;
    pc.over_point = {R(0), R(0), R(0)};
    if (valid) {
      pc = prepare(sc, load_hit_ray(pool, r, hit.flags), p, hit.flags);
      V3<R> ldir;
      R limit;
      light_ray(sc, pc.over_point, ldir, limit);
      R* q = s_ray + (size_t)threadIdx.x * 7;
      st3(q, pc.over_point);
      st3(q + 3, ldir);
      q[6] = limit;
      s_list[wbase + __popcll(m & lanes_below())] = (uint8_t)lane;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t n_valid = __popcll(m);
    for (uint32_t sb = 0; sb * 16u < n_valid; ++sb) {
      const uint32_t slot = sb * 16u + (lane >> 2);
      if (slot < n_valid) {  // the same in the four lanes of a quad
        const uint32_t src = s_list[wbase + slot];
        const R* q = s_ray + (size_t)(wbase + src) * 7;
        const V3<R> so = ld3(q), sd = ld3(q + 3);
        AnyVis<R, MODE> av(sc, so, sd, q[6], wk);
        traverse_quad<R, MODE>(sc, so, sd, av, wk, stack);
        const bool found = quad_any(av.found);
        if ((lane & 3u) == 0) s_found[wbase + src] = found ? 1 : 0;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (valid) {
      const DevMaterial<R>& mat = sc.materials[hit.flags >> PF_MAT_SHIFT];
      st3(pool.color + 3ull * r, lighting(sc, mat, pc.over_point, pc.eyev, pc.normalv, s_found[threadIdx.x] != 0));
    }
  }
  flush_work<(MODE & M_COUNT) != 0>(ctl, launch, wk);
}

// ------------------------------------------------------------------------------------------------
// k_spawn: the reflected / refracted rays of every hit of generation `gen` (src/world.rs:68-102) and the
// Schlick reflectance (src/computation.rs:20-33), with n1 / n2 looked up first for hits on transparent
// shapes.  Launched while rays may still be spawned (remaining > 0); the deepest generation has no
// children and k_combine is told so.
// ------------------------------------------------------------------------------------------------
// The 64 rays of one wave: the child rays of their hits are placed from `base` (an offset into generation gen + 1,
// this wave's own prefix) on, reflected before refracted.  Called by every lane of the wave.
// ONLY_T: the reflected rays were written by k_shadow<..., SPAWN>; what is left are the hits on transparent shapes --
// n1 / n2, the refracted ray, the Schlick weight -- and waves without such a hit leave at once.
template <class R, int MODE, bool PACKET, bool ONLY_T>
RT_FN void spawn_chunk(const DevScene<R>& sc, const RayPool<R>& pool, FrameCtl* ctl, int gen, uint32_t start, uint32_t gcount, uint32_t i, HitRec hit,
                       int any_transparent, int* lds, Work& wk) {
  const uint32_t child_start = start + gcount;
  const uint32_t r = start + (i < gcount ? i : 0u);
  const bool valid = i < gcount && hit.prim >= 0;
  // hits on shapes that neither reflect nor transmit have nothing to do here
  const bool slot_r = valid && (hit.flags & PF_REFLECTIVE);
  const bool slot_t = valid && (hit.flags & PF_TRANSPARENT);
  if (__builtin_amdgcn_ballot_w64(ONLY_T ? slot_t : (slot_r | slot_t)) == 0ull) return;  // wave-uniform
  const uint32_t base = child_base(pool.blk_base, pool.blk_top, __builtin_amdgcn_readfirstlane(i) / GROUP);
  const int p = valid ? hit.prim : 0;
  PrimInfo info;
  info.flags = hit.flags, info.shape = 0;
  HitRay<R> h{};
  if (ONLY_T ? slot_t : (slot_r || slot_t)) h = load_hit_ray(pool, r, hit.flags);
  R n1 = R(1), n2 = R(1);
  if (any_transparent && __ballot(slot_t) != 0ull) {  // wave-uniform
    if (slot_t) info.shape = sc.prim_info[p].shape;
    container_indices<R, MODE, PACKET>(sc, slot_t, p, info, h.o, h.d, h.t, lds, wk, n1, n2);
  }
  bool want_refr = false;
  V3<R> refl_o = {R(0), R(0), R(0)}, refl_d = refl_o, refr_o = refl_o, refr_d = refl_o;
  R rs = R(-1);
  if (ONLY_T ? slot_t : (slot_r || slot_t)) {
    const Prepared<R> c = prepare(sc, h, p, hit.flags);
    refl_o = c.over_point;
    refl_d = reflect(c.d, c.normalv_temp);  // un-flipped normal (src/intersection.rs:85)
    if (slot_t) {
      const R n_ratio = n1 / n2;
      const R cos_ti = dot(c.eyev, c.normalv);
      const R sin2_t = n_ratio * n_ratio * (R(1) - cos_ti * cos_ti);
      if (!(sin2_t > R(1))) {
        const R cos_t = sqrt(R(1) - sin2_t);
        refr_o = c.under_point;
        refr_d = (n_ratio * cos_ti - cos_t) * c.normalv - n_ratio * c.eyev;
        want_refr = true;
      }
      if (info.flags & PF_SCHLICK) {  // reflective > 0 && transparency > 0 (src/world.rs:120): schlick()
        R cs = cos_ti;
        bool tir = false;
        if (n1 > n2) {
          const R n = n1 / n2;
          const R s2 = n * n * (R(1) - cs * cs);
          if (s2 > R(1)) tir = true;
          else cs = sqrt(R(1) - s2);
        }
        if (tir) {
          rs = R(1);
        } else {
          R r0 = (n1 - n2) / (n1 + n2);
          r0 = r0 * r0;
          const R k = R(1) - cs;
          rs = r0 + ((R(1) - r0) * (k * k * k * k * k));
        }
      }
    }
  }

  // child ids: the wave's base + rank inside the wave; the bound counts a refracted ray for every
  // transparent hit, so a hit that turns out to be total internal reflection leaves its slot empty (zero
  // direction = no ray).
  const unsigned long long m_r = __ballot(slot_r), m_t = __ballot(slot_t);
  const uint32_t n_r = __popcll(m_r);
  const uint32_t off_r = __popcll(m_r & lanes_below()), off_t = n_r + __popcll(m_t & lanes_below());
  const uint32_t n_holes = __popcll(__ballot(slot_t && !want_refr));
  if ((threadIdx.x & 63u) == 0 && n_holes) atomicAdd(&ctl->holes[gen + 1].v, n_holes);

  int32_t child_r = -1, child_t = -1;
  if (!ONLY_T && slot_r) {
    const unsigned long long id = (unsigned long long)child_start + base + off_r;
    if (id < (unsigned long long)pool.capacity) {
      child_r = (int32_t)id;
      st3(pool.org + 3ull * id, refl_o);
      st3(pool.dir + 3ull * id, refl_d);
    } else {
      ctl->overflow.v = 1;
    }
  }
  if (slot_t) {
    const unsigned long long id = (unsigned long long)child_start + base + off_t;
    if (id < (unsigned long long)pool.capacity) {
      if (want_refr) child_t = (int32_t)id;
      st3(pool.org + 3ull * id, refr_o);
      st3(pool.dir + 3ull * id, refr_d);  // zero when there is no refracted ray
    } else {
      ctl->overflow.v = 1;
    }
  }
  if constexpr (ONLY_T) {
    if (slot_t) pool.child[2ull * r + 1] = child_t;
    if (slot_t && (info.flags & PF_SCHLICK)) pool.mixr[r] = rs;
  } else if (slot_r || slot_t) {  // k_combine looks at these only for hits on reflective / transparent shapes
    pool.child[2ull * r] = child_r;
    pool.child[2ull * r + 1] = child_t;
    if (info.flags & PF_SCHLICK) pool.mixr[r] = rs;  // -1 (plain sum) for every other prim: k_combine does not read it then
  }
}

template <class R, int MODE, bool PACKET, bool ONLY_T>
__global__ void __launch_bounds__(BLOCK, PACKET ? RT355_WPS_SPAWN_PACKET : RT355_WPS_SPAWN) k_spawn(DevScene<R> sc, RayPool<R> pool, FrameCtl* ctl, int gen, int any_transparent, int launch) {
  __shared__ int lds_stack[PACKET ? 1 : LDS_STACK * BLOCK];
  int* lds = lds_stack + (PACKET ? 0 : threadIdx.x);
  Work wk;
  uint32_t start, gcount;
  gen_range(ctl, gen, pool.capacity, start, gcount);
  // Whether a wave of rays has anything to spawn is known from the words its trace left for k_scan -- the child bound,
  // or for ONLY_T the flag above the hit count -- four bytes per 64 rays, read through the scalar cache two trips
  // ahead; the hit records (eight bytes per ray) are requested one trip ahead, and only for waves that spawn.
  // (Reading the words of 64 trips with one vector load and visiting only the trips that spawn is no faster where few
  // waves spawn and 10 % slower where most do.)
  static_assert(BLOCK == GROUP, "one entry of the bookkeeping per block and trip");
  const uint32_t stride = gridDim.x * BLOCK;
  uint32_t rbase = blockIdx.x * BLOCK;
  auto spawns = [&](uint32_t base) -> bool {  // wave-uniform
    if (base >= gcount) return false;
    cptr<uint32_t> q = (cptr<uint32_t>)(const uint32_t*)(ONLY_T ? pool.blk_hits : pool.blk_children);
    const uint32_t w = q[base / GROUP];
    return ONLY_T ? (w & WAVE_HAS_TRANSPARENT) != 0u : w != 0u;
  };
  bool w_cur = spawns(rbase), w_next = spawns(rbase + stride);
  HitRec next = HitRec{-1, 0};
  if (w_cur) next = load_hit(pool, start + (rbase + threadIdx.x < gcount ? rbase + threadIdx.x : 0u));
  for (; rbase < gcount; rbase += stride) {
    const uint32_t i = rbase + threadIdx.x;
    const HitRec hit = next;
    const bool w = w_cur;
    w_cur = w_next;
    if (w_cur) next = load_hit(pool, start + (i + stride < gcount ? i + stride : 0u));
    w_next = spawns(rbase + 2u * stride);
    if (w) spawn_chunk<R, MODE, PACKET, ONLY_T>(sc, pool, ctl, gen, start, gcount, i, hit, any_transparent, lds, wk);
  }
  flush_work<(MODE & M_COUNT) != 0>(ctl, launch, wk);
}

// ------------------------------------------------------------------------------------------------
// k_combine: shade_hit's sum (src/world.rs:117-126) for every hit of generation `gen`, on top of the surface
// colour k_shadow left in pool.color; runs from the deepest generation back to 0, so the children's colours
// are final when their parent adds them.  Generation 0 may write the canvas directly.  `has_children` is 0
// for the deepest generation, whose rays spawned nothing (remaining == 0: src/world.rs:69, :79).
// ------------------------------------------------------------------------------------------------
enum OutMode { OUT_NONE = -1, OUT_LINEAR = 0, OUT_ROWMAJOR = 1, OUT_TILES = 2 };

template <class R>
RT_FN void write_pixel(const DevCamera<R>& cam, uint32_t i, int mode, float* out, V3<R> c) {
  size_t o;
  if (mode == OUT_LINEAR) {
    o = (size_t)i * 3;
  } else {
    int x, y, ts, it, frame;
    const bool ok = pixel_of(cam, i, x, y, ts, it, frame);
    if (mode == OUT_ROWMAJOR) {
      if (!ok) return;
      o = (((size_t)frame * cam.height + y) * cam.width + x) * 3;  // one canvas per frame of the batch
    } else {
      o = ((size_t)ts * TILE_PIX + it) * 3;  // padding pixels are written as black
    }
  }
  out[o] = (float)c.x, out[o + 1] = (float)c.y, out[o + 2] = (float)c.z;
}

// Colour of ray r (World::color_at of the recursion below it), LEVELS generations of children deep: the
// children's colours are evaluated on the way (LEVELS > 0) or taken as final from pool.color (LEVELS == 0).
// The expression per hit is the one of src/world.rs:117-126, children first.
template <class R, int LEVELS>
RT_FN V3<R> ray_colour(const DevScene<R>& sc, const RayPool<R>& pool, uint32_t r, bool has_children) {
  const HitRec info = load_hit(pool, r);
  if (info.prim < 0) return V3<R>{R(0), R(0), R(0)};  // world.rs:61-63: a miss is black
  const DevMaterial<R>& m = sc.materials[info.flags >> PF_MAT_SHIFT];
  const V3<R> surface = ld3(pool.color + 3ull * r);  // left there by k_shadow
  int32_t cr = -1, ct = -1;
  R rs = R(-1);
  if (has_children && (info.flags & (PF_REFLECTIVE | PF_TRANSPARENT))) {
    cr = pool.child[2ull * r], ct = pool.child[2ull * r + 1];
    if (info.flags & PF_SCHLICK) rs = pool.mixr[r];  // written for these prims only
  }
  V3<R> refl = {R(0), R(0), R(0)}, refr = {R(0), R(0), R(0)};
  if constexpr (LEVELS > 0) {
    // every generation walked here lies above the deepest one, so its hits may have children
    if (cr >= 0) refl = ray_colour<R, LEVELS - 1>(sc, pool, (uint32_t)cr, true) * m.reflective;
    if (ct >= 0) refr = ray_colour<R, LEVELS - 1>(sc, pool, (uint32_t)ct, true) * m.transparency;
  } else {
    if (cr >= 0) refl = ld3(pool.color + 3ull * cr) * m.reflective;
    if (ct >= 0) refr = ld3(pool.color + 3ull * ct) * m.transparency;
  }
  if (rs >= R(0) || rs != rs) return surface + refl * rs + refr * (R(1) - rs);
  return surface + refr + refl;
}

// LEVELS: how many generations below `gen` are folded in by walking each ray's own subtree (0: the children's
// colours are final already).  With LEVELS = n the generations gen+1 .. gen+n need no launch of their own,
// and generation gen+n+1 must be final (the deepest one, shaded inside its trace launch).  The default folds four:
// a frame of max_depth 5 is combined by one launch from generation 0.
template <class R, int LEVELS>
__global__ void __launch_bounds__(256, RT355_WPS_COMBINE) k_combine(DevScene<R> sc, RayPool<R> pool, FrameCtl* ctl, int gen, int has_children, DevCamera<R> cam, int mode, float* out) {
  uint32_t start, gcount;
  gen_range(ctl, gen, pool.capacity, start, gcount);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < gcount; i += gridDim.x * blockDim.x) {
    const uint32_t r = start + i;
    const V3<R> c = ray_colour<R, LEVELS>(sc, pool, r, has_children != 0);
    if (gen > 0 || mode == OUT_NONE) st3(pool.color + 3ull * r, c);
    if (gen == 0 && mode != OUT_NONE) write_pixel(cam, i, mode, out, c);
  }
}



// rank buffers (tile-major) -> row-major canvases, after the gather on rank 0.  Every rank's buffer holds
// n_frames frames of tiles_per_rank tiles each; frame f lands in out[f * width * height ...].
template <class T>
__global__ void __launch_bounds__(256) k_untile(const T* gathered, T* out, int width, int height, int tiles_x, int n_tiles, int n_ranks, long long tiles_per_rank,
                                                int n_frames) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long per_frame = (long long)width * height;
  if (i >= per_frame * n_frames) return;
  const long long f = i / per_frame, k = i - f * per_frame;
  const int x = (int)(k % width), y = (int)(k / width);
  const long long tile = (long long)(y / TILE) * tiles_x + (x / TILE);
  const long long rank = tile % n_ranks, tl = tile / n_ranks;
  const long long src = (((rank * n_frames + f) * tiles_per_rank + tl) * TILE_PIX + (long long)(y % TILE) * TILE + (x % TILE)) * 3;
  out[i * 3] = gathered[src], out[i * 3 + 1] = gathered[src + 1], out[i * 3 + 2] = gathered[src + 2];
}

// World::is_shadowed for arbitrary points (parity probe)
template <class R, int MODE>
__global__ void __launch_bounds__(BLOCK) k_shadow_points(DevScene<R> sc, const float* pts, uint32_t n, uint8_t* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  __shared__ int lds_stack[LDS_STACK * BLOCK];
  Work wk;
  const V3<R> p = {R(pts[3 * i]), R(pts[3 * i + 1]), R(pts[3 * i + 2])};
  V3<R> ldir;
  R limit;
  light_ray(sc, p, ldir, limit);
  AnyVis<R, MODE> av(sc, p, ldir, limit, wk);
  traverse<R, MODE>(sc, p, ldir, av, wk, lds_stack + threadIdx.x);
  out[i] = av.found ? 1 : 0;
}

// the same with the quad traversal: four lanes to a point
template <class R>
__global__ void __launch_bounds__(BLOCK) k_shadow_points_q(DevScene<R> sc, const float* pts, uint32_t n, uint8_t* out) {
  __shared__ int lds_stack[QUADS * STACK_DEPTH];
  const uint32_t i = blockIdx.x * QUADS + (threadIdx.x >> 2);
  if (i >= n) return;  // whole quads leave together
  Work wk;
  const V3<R> p = {R(pts[3 * i]), R(pts[3 * i + 1]), R(pts[3 * i + 2])};
  V3<R> ldir;
  R limit;
  light_ray(sc, p, ldir, limit);
  AnyVis<R, 0> av(sc, p, ldir, limit, wk);
  traverse_quad<R, 0>(sc, p, ldir, av, wk, lds_stack + (threadIdx.x >> 2));
  const bool found = quad_any(av.found);
  if ((threadIdx.x & 3u) == 0) out[i] = found ? 1 : 0;
}

// caller-supplied rays (n*6 floats) -> pool generation 0
template <class R>
__global__ void __launch_bounds__(256) k_load_rays(const float* rays, RayPool<R> pool, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st3(pool.org + 3ull * i, mk<R>(R(rays[6 * i]), R(rays[6 * i + 1]), R(rays[6 * i + 2])));
  st3(pool.dir + 3ull * i, mk<R>(R(rays[6 * i + 3]), R(rays[6 * i + 4]), R(rays[6 * i + 5])));
}

// nearest-hit probe output: t as double, prim -> shape index
template <class R>
__global__ void __launch_bounds__(256) k_export_hits(DevScene<R> sc, RayPool<R> pool, uint32_t n, double* t_out, int32_t* idx_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int p = load_hit(pool, i).prim;
  if (p < 0) {
    t_out[i] = (double)NAN;
    idx_out[i] = -1;
  } else {
    t_out[i] = (double)pool.hit_t[i];
    idx_out[i] = sc.prim_info[p].shape;
  }
}

// Color::to_u8 (src/color.rs:97-111): round half away from zero, clamp, NaN -> 0.
static __global__ void __launch_bounds__(256) k_to_u8(const float* in, size_t n, uint8_t* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = roundf(in[i] * 255.0f);
  uint8_t b;
  if (v < 0.0f) b = 0;
  else if (v > 255.0f) b = 255;
  else if (v != v) b = 0;
  else b = (uint8_t)v;
  out[i] = b;
}

}  // namespace rt355
