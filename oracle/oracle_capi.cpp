// oracle_capi.cpp -- C entry points of the CPU parity oracle (rt_oracle.hpp).  TEST INFRASTRUCTURE
// ONLY: loaded with ctypes by tests/, by __graft_entry__.smoke() and by the cpu_baseline leg of
// bench.py.  The product never links or loads this library.
//
// Every function takes the same rt355_*_desc structs the device library takes, so one scene
// description feeds both sides.  `real_bits` picks the float (reference precision, epsilon.rs:1) or
// double instantiation.  flags bit 0 = strict_uv (intersection.rs:67-71 behaviour).
#include <atomic>
#include <chrono>
#include <thread>

#include "rt_oracle.hpp"

using namespace rto;

namespace {

template <class R>
bool build_world(const rt355_shape_desc* shapes, size_t n, const rt355_light_desc* light, int flags, World<R>& w) {
  w.shapes.resize(n);
  for (size_t i = 0; i < n; ++i)
    if (!make_shape<R>(shapes[i], w.shapes[i])) return false;
  if (light) {
    w.light.position = point<R>(light->position[0], light->position[1], light->position[2]);
    w.light.intensity = {R(light->intensity[0]), R(light->intensity[1]), R(light->intensity[2])};
  } else {
    w.light.position = point<R>(-10, 10, -10);  // pointlight.rs:20-22
    w.light.intensity = {1, 1, 1};
  }
  w.strict_uv = (flags & 1) != 0;
  return true;
}

// Runs body(i) for i in [0, n) on nthreads host threads (work handed out in blocks of 64 indices).  The rays /
// points of one call are independent, exactly as the pixels of Camera::render are (camera.rs:76-82).
template <class F>
void parallel_for(size_t n, int nthreads, F&& body) {
  if (nthreads <= 1 || n < 128) {
    for (size_t i = 0; i < n; ++i) body(i);
    return;
  }
  std::atomic<size_t> next(0);
  auto worker = [&]() {
    for (;;) {
      const size_t b = next.fetch_add(64);
      if (b >= n) break;
      for (size_t i = b; i < std::min(n, b + 64); ++i) body(i);
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t) th.emplace_back(worker);
  for (auto& t : th) t.join();
}

template <class R>
int render_impl(const rt355_shape_desc* shapes, size_t n, const rt355_light_desc* light,
                const rt355_camera_desc* cam, int depth, int flags, int x0, int x1, int y0, int y1,
                int nthreads, float* rgb_out, uint64_t* counters, double* seconds) {
  World<R> w;
  if (!build_world<R>(shapes, n, light, flags, w)) return -1;
  Camera<R> camera(*cam);
  if (!camera.ok) return -1;
  int W = (int)cam->hsize, H = (int)cam->vsize;
  if (x0 < 0) x0 = 0;
  if (y0 < 0) y0 = 0;
  if (x1 > W) x1 = W;
  if (y1 > H) y1 = H;
  if (nthreads < 1) nthreads = 1;
  std::vector<Counters> cnt(nthreads);
  std::atomic<int> next_row(y0);
  auto t0 = std::chrono::steady_clock::now();
  auto worker = [&](int tid) {
    for (;;) {
      int y = next_row.fetch_add(1);
      if (y >= y1) break;
      for (int x = x0; x < x1; ++x) {
        // camera.rs:76-82
        Ray<R> ray = camera.ray_for_pixel(R(x), R(y));
        Color<R> c = w.color_at(ray, depth, depth, &cnt[tid]);
        float* o = rgb_out + ((size_t)y * W + x) * 3;
        o[0] = (float)c.r, o[1] = (float)c.g, o[2] = (float)c.b;
      }
    }
  };
  if (nthreads == 1) {
    worker(0);
  } else {
    std::vector<std::thread> th;
    for (int i = 0; i < nthreads; ++i) th.emplace_back(worker, i);
    for (auto& t : th) t.join();
  }
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  if (counters) {
    Counters total;
    for (auto& c : cnt) total.add(c);
    for (int i = 0; i < RT355_MAX_GEN; ++i) counters[i] = total.radiance[i], counters[RT355_MAX_GEN + i] = total.hits[i];
    counters[2 * RT355_MAX_GEN] = total.panics;
  }
  return 0;
}

template <class R>
int color_at_impl(const rt355_shape_desc* shapes, size_t n, const rt355_light_desc* light,
                  const float* rays, size_t nrays, int depth, int flags, int nthreads, double* rgb_out) {
  World<R> w;
  if (!build_world<R>(shapes, n, light, flags, w)) return -1;
  parallel_for(nrays, nthreads, [&](size_t i) {
    const float* r = rays + i * 6;
    Ray<R> ray{point<R>(r[0], r[1], r[2]), vec<R>(r[3], r[4], r[5])};
    Color<R> c = w.color_at(ray, depth, depth, nullptr);
    rgb_out[i * 3 + 0] = c.r, rgb_out[i * 3 + 1] = c.g, rgb_out[i * 3 + 2] = c.b;
  });
  return 0;
}

template <class R>
int hit_impl(const rt355_shape_desc* shapes, size_t n, const float* rays, size_t nrays, int nthreads, double* t_out,
             int32_t* idx_out, double* runner_up_gap, uint8_t* beyond_extent) {
  World<R> w;
  if (!build_world<R>(shapes, n, nullptr, 0, w)) return -1;
  parallel_for(nrays, nthreads, [&](size_t i) {
    const float* r = rays + i * 6;
    Ray<R> ray{point<R>(r[0], r[1], r[2]), vec<R>(r[3], r[4], r[5])};
    auto xs = w.intersect(ray);
    int h = World<R>::hit(xs);
    if (h < 0) {
      t_out[i] = std::numeric_limits<double>::quiet_NaN();
      idx_out[i] = -1;
      if (runner_up_gap) runner_up_gap[i] = std::numeric_limits<double>::infinity();
      if (beyond_extent) beyond_extent[i] = 0;
      return;
    }
    t_out[i] = xs[h].t;
    idx_out[i] = xs[h].shape;
    if (beyond_extent) {
      // a truncated cone hit outside min..max: the parallel-ray branch of shape.rs:295-298
      const Shape<R>& s = w.shapes[xs[h].shape];
      beyond_extent[i] = 0;
      if (s.kind == RT355_CONE && std::isfinite((double)s.min) && std::isfinite((double)s.max)) {
        V4<R> lp = s.inv * ray.position(xs[h].t);
        if ((double)lp.y < (double)s.min - 1e-6 || (double)lp.y > (double)s.max + 1e-6) beyond_extent[i] = 1;
      }
    }
    if (runner_up_gap) {
      // distance in t to the nearest intersection of a DIFFERENT shape at t >= 0: marks ties
      double gap = std::numeric_limits<double>::infinity();
      for (auto& x : xs)
        if (x.t >= R(0) && x.shape != xs[h].shape) gap = std::min(gap, std::fabs((double)x.t - (double)xs[h].t));
      runner_up_gap[i] = gap;
    }
  });
  return 0;
}

template <class R>
int world_intersect_impl(const rt355_shape_desc* shapes, size_t n, const float* r, double* t_out,
                         int32_t* idx_out, int cap) {
  World<R> w;
  if (!build_world<R>(shapes, n, nullptr, 0, w)) return -1;
  Ray<R> ray{point<R>(r[0], r[1], r[2]), vec<R>(r[3], r[4], r[5])};
  auto xs = w.intersect(ray);
  int k = 0;
  for (auto& x : xs) {
    if (k >= cap) break;
    t_out[k] = x.t, idx_out[k] = x.shape, ++k;
  }
  return (int)xs.size();
}

template <class R>
int shape_intersect_impl(const rt355_shape_desc* shape, const float* r, double* t_out, double* u_out,
                         double* v_out, int cap) {
  World<R> w;
  if (!build_world<R>(shape, 1, nullptr, 0, w)) return -1;
  Ray<R> ray{point<R>(r[0], r[1], r[2]), vec<R>(r[3], r[4], r[5])};
  std::vector<Intersection<R>> xs;
  w.shape_intersect(0, ray, xs);  // emission order, unsorted (shape.rs:380-415)
  int k = 0;
  for (auto& x : xs) {
    if (k >= cap) break;
    t_out[k] = x.t;
    if (u_out) u_out[k] = x.has_uv ? (double)x.u : std::numeric_limits<double>::quiet_NaN();
    if (v_out) v_out[k] = x.has_uv ? (double)x.v : std::numeric_limits<double>::quiet_NaN();
    ++k;
  }
  return (int)xs.size();
}

template <class R>
int normal_impl(const rt355_shape_desc* shape, const float* p, int has_uv, double u, double v, double* out) {
  World<R> w;
  if (!build_world<R>(shape, 1, nullptr, 0, w)) return -1;
  Intersection<R> h{R(1), 0, has_uv != 0, R(u), R(v)};
  V4<R> nrm = w.normal(0, point<R>(p[0], p[1], p[2]), has_uv ? &h : nullptr, nullptr);
  out[0] = nrm.x, out[1] = nrm.y, out[2] = nrm.z;
  return 0;
}

template <class R>
int comps_impl(const rt355_shape_desc* shapes, size_t n, const rt355_light_desc* light, const float* r,
               const double* xs_t, const int32_t* xs_idx, const double* xs_u, const double* xs_v, int n_xs,
               int which, int use_list, int remaining, int flags, double* out) {
  World<R> w;
  if (!build_world<R>(shapes, n, light, flags, w)) return -1;
  Ray<R> ray{point<R>(r[0], r[1], r[2]), vec<R>(r[3], r[4], r[5])};
  std::vector<Intersection<R>> xs;
  for (int i = 0; i < n_xs; ++i) {
    bool huv = xs_u && !std::isnan(xs_u[i]);
    xs.push_back({R(xs_t[i]), xs_idx[i], huv, huv ? R(xs_u[i]) : R(0), huv ? R(xs_v[i]) : R(0)});
  }
  Computations<R> c = w.prepare_computation(xs[which], ray, use_list ? &xs : nullptr);
  auto put = [&](int o, V4<R> v) { out[o] = v.x, out[o + 1] = v.y, out[o + 2] = v.z; };
  put(0, c.point), put(3, c.eyev), put(6, c.normalv), put(9, c.over_point), put(12, c.under_point), put(15, c.reflectv);
  out[18] = c.inside ? 1 : 0, out[19] = c.n1, out[20] = c.n2, out[21] = World<R>::schlick(c);
  Color<R> sh = w.shade_hit(c, remaining, remaining, nullptr);
  Color<R> rl = w.reflected_color(c, remaining, remaining, nullptr);
  Color<R> rr = w.refracted_color(c, remaining, remaining, nullptr);
  out[22] = sh.r, out[23] = sh.g, out[24] = sh.b;
  out[25] = rl.r, out[26] = rl.g, out[27] = rl.b;
  out[28] = rr.r, out[29] = rr.g, out[30] = rr.b;
  out[31] = c.would_panic ? 1 : 0;
  return 0;
}

template <class R>
int lighting_impl(const rt355_material* m, const rt355_light_desc* light, const float* p, const float* eyev,
                  const float* normalv, int shadowed, double* rgb) {
  Material<R> mat = make_material<R>(*m);
  Light<R> l{point<R>(light->position[0], light->position[1], light->position[2]),
             {R(light->intensity[0]), R(light->intensity[1]), R(light->intensity[2])}};
  Color<R> c = World<R>::lighting(mat, l, point<R>(p[0], p[1], p[2]), vec<R>(eyev[0], eyev[1], eyev[2]),
                                  vec<R>(normalv[0], normalv[1], normalv[2]), shadowed != 0);
  rgb[0] = c.r, rgb[1] = c.g, rgb[2] = c.b;
  return 0;
}

template <class R>
int shadowed_impl(const rt355_shape_desc* shapes, size_t n, const rt355_light_desc* light, const float* pts,
                  size_t npts, int nthreads, uint8_t* out) {
  World<R> w;
  if (!build_world<R>(shapes, n, light, 0, w)) return -1;
  parallel_for(npts, nthreads, [&](size_t i) { out[i] = w.is_shadowed(point<R>(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2])) ? 1 : 0; });
  return 0;
}

template <class R>
int camera_impl(const rt355_camera_desc* cam, const float* pxy, size_t n, double* rays_out, double* pixel_size) {
  Camera<R> c(*cam);
  if (!c.ok) return -1;
  if (pixel_size) *pixel_size = c.pixel_size;
  for (size_t i = 0; i < n; ++i) {
    Ray<R> r = c.ray_for_pixel(R(pxy[i * 2]), R(pxy[i * 2 + 1]));
    double* o = rays_out + i * 6;
    o[0] = r.origin.x, o[1] = r.origin.y, o[2] = r.origin.z;
    o[3] = r.direction.x, o[4] = r.direction.y, o[5] = r.direction.z;
  }
  return 0;
}

}  // namespace

#define DISPATCH(bits, fn, ...) ((bits) == 32 ? fn<float>(__VA_ARGS__) : fn<double>(__VA_ARGS__))

extern "C" {

// Camera::render (camera.rs:74-88) over the pixel window [x0,x1) x [y0,y1).  rgb_out is the full
// hsize*vsize*3 canvas; only the window is written.  counters: [0..8) radiance rays per generation,
// [8..16) hits (= shadow rays) per generation, [16] pixels where strict_uv would have panicked.
int rto_render(const rt355_shape_desc* shapes, size_t n, const rt355_light_desc* light,
               const rt355_camera_desc* cam, int depth, int real_bits, int flags, int x0, int x1, int y0, int y1,
               int nthreads, float* rgb_out, uint64_t* counters, double* seconds) {
  return DISPATCH(real_bits, render_impl, shapes, n, light, cam, depth, flags, x0, x1, y0, y1, nthreads, rgb_out, counters, seconds);
}
int rto_color_at(const rt355_shape_desc* shapes, size_t n, const rt355_light_desc* light, const float* rays,
                 size_t nrays, int depth, int real_bits, int flags, int nthreads, double* rgb_out) {
  return DISPATCH(real_bits, color_at_impl, shapes, n, light, rays, nrays, depth, flags, nthreads, rgb_out);
}
int rto_hit(const rt355_shape_desc* shapes, size_t n, const float* rays, size_t nrays, int real_bits, int nthreads,
            double* t_out, int32_t* idx_out, double* runner_up_gap, uint8_t* beyond_extent) {
  return DISPATCH(real_bits, hit_impl, shapes, n, rays, nrays, nthreads, t_out, idx_out, runner_up_gap, beyond_extent);
}
int rto_world_intersect(const rt355_shape_desc* shapes, size_t n, const float* ray, int real_bits,
                        double* t_out, int32_t* idx_out, int cap) {
  return DISPATCH(real_bits, world_intersect_impl, shapes, n, ray, t_out, idx_out, cap);
}
int rto_shape_intersect(const rt355_shape_desc* shape, const float* ray, int real_bits, double* t_out,
                        double* u_out, double* v_out, int cap) {
  return DISPATCH(real_bits, shape_intersect_impl, shape, ray, t_out, u_out, v_out, cap);
}
int rto_normal_at(const rt355_shape_desc* shape, const float* p, int has_uv, double u, double v, int real_bits,
                  double* out) {
  return DISPATCH(real_bits, normal_impl, shape, p, has_uv, u, v, out);
}
// out[0..3) point [3..6) eyev [6..9) normalv [9..12) over_point [12..15) under_point [15..18) reflectv
// [18] inside [19] n1 [20] n2 [21] schlick [22..25) shade_hit [25..28) reflected_color
// [28..31) refracted_color [31] would_panic
int rto_comps(const rt355_shape_desc* shapes, size_t n, const rt355_light_desc* light, const float* ray,
              const double* xs_t, const int32_t* xs_idx, const double* xs_u, const double* xs_v, int n_xs, int which,
              int use_list, int remaining, int real_bits, int flags, double* out) {
  return DISPATCH(real_bits, comps_impl, shapes, n, light, ray, xs_t, xs_idx, xs_u, xs_v, n_xs, which, use_list, remaining, flags, out);
}
int rto_lighting(const rt355_material* m, const rt355_light_desc* light, const float* p, const float* eyev,
                 const float* normalv, int shadowed, int real_bits, double* rgb) {
  return DISPATCH(real_bits, lighting_impl, m, light, p, eyev, normalv, shadowed, rgb);
}
int rto_is_shadowed(const rt355_shape_desc* shapes, size_t n, const rt355_light_desc* light, const float* pts,
                    size_t npts, int real_bits, int nthreads, uint8_t* out) {
  return DISPATCH(real_bits, shadowed_impl, shapes, n, light, pts, npts, nthreads, out);
}
int rto_camera_rays(const rt355_camera_desc* cam, const float* pxy, size_t n, int real_bits, double* rays_out,
                    double* pixel_size) {
  return DISPATCH(real_bits, camera_impl, cam, pxy, n, rays_out, pixel_size);
}
void rto_to_u8(const float* rgb, size_t n, uint8_t* out) {
  for (size_t i = 0; i < n; ++i) out[i] = to_u8<float>(rgb[i]);
}
}
