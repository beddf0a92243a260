// asan_driver.cpp -- TEST INFRASTRUCTURE.  Runs the CPU oracle (oracle_capi.cpp: the restatement of src/world.rs,
// src/intersection.rs, src/material.rs, src/camera.rs of the reference) and the library's host-side scene builder
// (raytracer_amd/csrc/scene_build.cpp) over one scene file in a process built with -fsanitize=address,undefined, and
// writes what they computed, so that a test can compare it with the ordinary builds' results.
//
// usage: asan_driver <scene.bin> <out.bin>
// scene.bin: int32 header {n_shapes, n_rays, n_points, depth, real_bits, exact_cones}, rt355_shape_desc[n_shapes],
//            rt355_light_desc, rt355_camera_desc, float rays[n_rays*6], float points[n_points*3]
// out.bin:   double t[n_rays], int32 idx[n_rays], double rgb[n_rays*3], uint8 shadowed[n_points],
//            float canvas[h*w*3], uint64 scene_info[5]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../raytracer_amd/csrc/rt355_device.h"

extern "C" {
int rto_render(const rt355_shape_desc*, size_t, const rt355_light_desc*, const rt355_camera_desc*, int, int, int, int, int, int, int, int, float*,
               uint64_t*, double*);
int rto_color_at(const rt355_shape_desc*, size_t, const rt355_light_desc*, const float*, size_t, int, int, int, int, double*);
int rto_hit(const rt355_shape_desc*, size_t, const float*, size_t, int, int, double*, int32_t*, double*, uint8_t*);
int rto_is_shadowed(const rt355_shape_desc*, size_t, const rt355_light_desc*, const float*, size_t, int, int, uint8_t*);
}

template <class T>
static bool rd(FILE* f, T* p, size_t n) { return n == 0 || fread(p, sizeof(T), n, f) == n; }
template <class T>
static bool wr(FILE* f, const T* p, size_t n) { return n == 0 || fwrite(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t hdr[6];
  if (!rd(f, hdr, 6)) return 2;
  const size_t n_shapes = (size_t)hdr[0], n_rays = (size_t)hdr[1], n_pts = (size_t)hdr[2];
  const int depth = hdr[3], bits = hdr[4], exact_cones = hdr[5];
  std::vector<rt355_shape_desc> shapes(n_shapes);
  rt355_light_desc light;
  rt355_camera_desc cam;
  std::vector<float> rays(n_rays * 6), pts(n_pts * 3);
  if (!rd(f, shapes.data(), n_shapes) || !rd(f, &light, 1) || !rd(f, &cam, 1) || !rd(f, rays.data(), rays.size()) || !rd(f, pts.data(), pts.size())) return 2;
  fclose(f);

  std::vector<double> t(n_rays), gap(n_rays), rgb(n_rays * 3);
  std::vector<int32_t> idx(n_rays);
  std::vector<uint8_t> beyond(n_rays), shadowed(n_pts);
  if (rto_hit(shapes.data(), n_shapes, rays.data(), n_rays, bits, 2, t.data(), idx.data(), gap.data(), beyond.data())) return 3;
  if (rto_color_at(shapes.data(), n_shapes, &light, rays.data(), n_rays, depth, bits, 0, 2, rgb.data())) return 3;
  if (rto_is_shadowed(shapes.data(), n_shapes, &light, pts.data(), n_pts, bits, 2, shadowed.data())) return 3;
  const int w = (int)cam.hsize, h = (int)cam.vsize;
  std::vector<float> canvas((size_t)w * h * 3);
  uint64_t counters[2 * RT355_MAX_GEN + 1] = {0};
  double seconds = 0;
  if (rto_render(shapes.data(), n_shapes, &light, &cam, depth, bits, 0, 0, w, 0, h, 2, canvas.data(), counters, &seconds)) return 3;

  // the library's host side: scene tables, both levels of trees, and their structural check
  uint64_t info[5] = {0};
  {
    rt355::HostScene hs;
    char err[256] = {0};
    const double pad_rel = bits == 32 ? 256.0 * 1.1920929e-7 : 1e-9;
    if (rt355::build_host_scene(shapes.data(), n_shapes, pad_rel, exact_cones ? INT64_MAX : 32, hs, err, sizeof(err))) { fprintf(stderr, "build: %s\n", err); return 4; }
    if (rt355::verify_host_scene(hs, err, sizeof(err))) { fprintf(stderr, "verify: %s\n", err); return 4; }
    info[0] = hs.nodes.size(), info[1] = (uint64_t)hs.n_bounded, info[2] = (uint64_t)(hs.n_top - hs.n_bounded), info[3] = (uint64_t)hs.max_depth, info[4] = hs.materials.size();
  }

  FILE* o = fopen(argv[2], "wb");
  if (!o) return 2;
  const bool ok = wr(o, t.data(), n_rays) && wr(o, idx.data(), n_rays) && wr(o, rgb.data(), rgb.size()) && wr(o, shadowed.data(), n_pts) &&
                  wr(o, canvas.data(), canvas.size()) && wr(o, info, 5);
  fclose(o);
  return ok ? 0 : 2;
}
