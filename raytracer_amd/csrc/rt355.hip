// rt355.hip -- implementation of the C ABI declared in include/rt355.h.
//
// One context = one GPU + one HIP stream + the per-frame ray pool.  A frame is a fixed sequence of
// kernel launches on that stream with no host round trip in between: generation sizes live in a
// device-side control block that the kernels read (grid-stride loops over device-side counts), and the
// block is read back once, after the frame, for the statistics.
#include <mutex>

#include "frame.h"

using namespace rt355;

thread_local std::string rt355::g_err;

// the launch chains are compiled one (type, mode) pair per translation unit: frame_inst.hip
#define RT355_FRAME_EXTERN(R, MODE)                                                                                                       \
  extern template int rt355::run_generations<R, MODE>(rt355_ctx*, const rt355_scene*, RayPool<R>&, FrameCtl*, uint32_t, int, const DevCamera<R>&, bool, \
                                                      int, float*, LaunchLog&);
RT355_FRAME_EXTERN(double, 0) RT355_FRAME_EXTERN(double, 1) RT355_FRAME_EXTERN(double, 2) RT355_FRAME_EXTERN(double, 3)
RT355_FRAME_EXTERN(float, 0) RT355_FRAME_EXTERN(float, 1) RT355_FRAME_EXTERN(float, 2) RT355_FRAME_EXTERN(float, 3)
#undef RT355_FRAME_EXTERN

namespace {

// FLOP weights per unit of work: 1 per add/sub/mul/div/sqrt/min/max/compare, FMA = 2.
const double F_PRIMARY = 40, F_BOX = 24, F_XFORM = 33;
const double F_KIND[8] = {28 /*sphere*/, 4 /*plane*/, 20 /*cube*/, 50 /*cylinder*/, 62 /*cone*/, 0, 52 /*triangle*/, 52 /*smooth*/};
// F_SHADE: prepare_computation + Phong lighting + pattern per hit, 150, plus the pow of the specular term, 40.
const double F_SHADE = 190, F_REFLECT = 30, F_REFRACT = 35, F_SCHLICK = 25;

// Fixed workspace of rt355_render_device_noalloc: the frame's ray pool and control block for callers that may
// not allocate (serving loops, harnesses that forbid it).  The call allocates nothing; it is synchronous with respect
// to the host (it waits for the frame), so it is not a call to make under stream capture.  One copy per GPU the library runs on, held until the
// process ends.  In host code the array's name is only a handle: the device address is asked for once per
// device (scratch_address) and everything on the host side goes through that pointer.
static __device__ __align__(256) unsigned char scratch[RT355_NOALLOC_BYTES];
constexpr int MAX_DEVICES = 64;
void* g_scratch_ptr[MAX_DEVICES] = {};
std::mutex g_scratch_mutex;

// Address of this device's copy of `scratch`; the device must be current.  Looked up on first use per device.
int scratch_address(int device, void** out) {
  if (device < 0 || device >= MAX_DEVICES) return set_err(RT355_EINVAL, "device id out of range");
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) return set_err(RT355_EHIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    g_scratch_ptr[device] = p;
  }
  *out = g_scratch_ptr[device];
  return RT355_OK;
}

}  // namespace

namespace {

// hsize / vsize as the integers the reference loops over: it keeps them as f32 (src/camera.rs:12-13) and renders
// `0..(size as usize)` rows and columns (src/camera.rs:76-77, :84-85).  Checked before any cast or allocation: NaN,
// infinities, sizes below 1 or beyond what a canvas can be are rejected.
constexpr float MAX_CANVAS_EDGE = 65536.f;
int camera_dims(const rt355_camera_desc* c, int& w, int& h) {
  if (!c) return set_err(RT355_EINVAL, "null camera");
  if (!std::isfinite(c->hsize) || !std::isfinite(c->vsize) || !(c->hsize >= 1.f) || !(c->vsize >= 1.f) || c->hsize > MAX_CANVAS_EDGE ||
      c->vsize > MAX_CANVAS_EDGE)
    return set_err(RT355_EINVAL, "camera: hsize and vsize must be finite and in [1, 65536]");
  w = (int)c->hsize, h = (int)c->vsize;
  return RT355_OK;
}

// Camera::new (src/camera.rs:23-51) in R, plus the inverse of the camera transformation that
// ray_for_pixel recomputes per pixel (src/camera.rs:67).
template <class R>
int dev_view(const rt355_camera_desc* c, DevView<R>& out) {
  int cw = 0, chh = 0;
  if (int rc = camera_dims(c, cw, chh)) return rc;
  if (!std::isfinite(c->fov)) return set_err(RT355_EINVAL, "camera: bad field of view");
  const float* m = c->transform;
  for (int k = 0; k < 16; ++k)
    if (!std::isfinite(m[k])) return set_err(RT355_EINVAL, "camera: non-finite transformation");
  if (m[3] != 0.f || m[7] != 0.f || m[11] != 0.f || m[15] != 1.f) return set_err(RT355_EINVAL, "camera: transformation is not affine");
  auto at = [&](int r, int cc) { return (double)m[cc * 4 + r]; };
  double a = at(0, 0), b = at(0, 1), cc = at(0, 2), d = at(1, 0), e = at(1, 1), f = at(1, 2), g = at(2, 0), h = at(2, 1), i = at(2, 2);
  double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  double det = a * A + b * B + cc * C;
  if (det == 0.0 || !std::isfinite(det)) return set_err(RT355_EINVAL, "camera: singular transformation");
  double id = 1.0 / det;
  double r[9] = {A * id, -(b * i - cc * h) * id, (b * f - cc * e) * id, B * id, (a * i - cc * g) * id, -(a * f - cc * d) * id,
                 C * id, -(a * h - b * g) * id, (a * e - b * d) * id};
  double tx = at(0, 3), ty = at(1, 3), tz = at(2, 3);
  for (int row = 0; row < 3; ++row) {
    out.inv[row * 4 + 0] = R(r[row * 3 + 0]);
    out.inv[row * 4 + 1] = R(r[row * 3 + 1]);
    out.inv[row * 4 + 2] = R(r[row * 3 + 2]);
    out.inv[row * 4 + 3] = R(-(r[row * 3 + 0] * tx + r[row * 3 + 1] * ty + r[row * 3 + 2] * tz));
  }
  const R hsize = R(c->hsize), vsize = R(c->vsize), fov = R(c->fov);
  const R half_view = std::tan(fov / R(2));
  const R aspect = hsize / vsize;
  out.half_width = aspect >= R(1) ? half_view : half_view * aspect;
  out.half_height = aspect >= R(1) ? half_view / aspect : half_view;
  out.pixel_size = (out.half_width * R(2)) / hsize;
  out.pad = R(0);
  return RT355_OK;
}

// The cameras of one launch chain: n_frames views of one image size.  The views of a batch (n_frames > 1) are
// written to `views_host`, which the caller copies to the device address it stores in out.views.
template <class R>
int dev_camera(const rt355_camera_desc* cams, int n_frames, int rank, int n_ranks, DevCamera<R>& out, DevView<R>* views_host) {
  int cw = 0, chh = 0;
  if (int rc = camera_dims(cams, cw, chh)) return rc;
  if (int rc = dev_view<R>(cams, out.view)) return rc;
  for (int f = 0; f < n_frames; ++f) {
    int w2 = 0, h2 = 0;
    if (int rc = camera_dims(cams + f, w2, h2)) return rc;
    if (w2 != cw || h2 != chh) return set_err(RT355_EINVAL, "the cameras of a batch must share hsize and vsize");
    if (int rc = dev_view<R>(cams + f, views_host[f])) return rc;
  }
  out.width = cw, out.height = chh;
  out.tiles_x = (out.width + TILE - 1) / TILE, out.tiles_y = (out.height + TILE - 1) / TILE;
  out.n_tiles = out.tiles_x * out.tiles_y;
  out.rank = rank, out.n_ranks = n_ranks;
  out.n_frames = n_frames;
  out.tiles_per_frame = (int32_t)((out.n_tiles + n_ranks - 1) / n_ranks);
  out.inv_tiles_x = 1.f / (float)out.tiles_x, out.inv_tiles_per_frame = 1.f / (float)out.tiles_per_frame;
  if ((int64_t)out.tiles_per_frame * n_frames >= (1 << 23)) return set_err(RT355_EINVAL, "batch: too many tiles for one launch chain");
  out.views = nullptr;
  return RT355_OK;
}

int64_t tiles_per_rank_of(int width, int height, int n_ranks) {
  int64_t tx = (width + TILE - 1) / TILE, ty = (height + TILE - 1) / TILE;
  return (tx * ty + n_ranks - 1) / n_ranks;
}

template <class R>
int ensure_pool(rt355_ctx* c, uint64_t n_primary, bool secondary, RayPool<R>& pool) {
  double factor = secondary ? std::max(1.0, c->pool_factor) : 1.0;
  uint64_t cap = (uint64_t)std::ceil((double)n_primary * factor) + 64;
  if (cap >= 0x7fffffffull) return set_err(RT355_EINVAL, "ray pool too large for 32-bit ray ids");
  int rc;
  if ((rc = c->org.ensure(cap * 3 * sizeof(R)))) return rc;
  if ((rc = c->dir.ensure(cap * 3 * sizeof(R)))) return rc;
  if ((rc = c->hit_t.ensure(cap * sizeof(R)))) return rc;
  if ((rc = c->hit_uv.ensure(cap * 2 * sizeof(R)))) return rc;
  if ((rc = c->hit_prim.ensure(cap * 2 * sizeof(int32_t)))) return rc;
  if ((rc = c->color.ensure(cap * 3 * sizeof(R)))) return rc;
  if ((rc = c->mixr.ensure(cap * sizeof(R)))) return rc;
  if ((rc = c->child.ensure(cap * 2 * sizeof(int32_t)))) return rc;
  const uint64_t nblk = cap / GROUP + 8;
  if ((rc = c->blk_hits.ensure(nblk * sizeof(uint32_t)))) return rc;
  if ((rc = c->blk_children.ensure(nblk * sizeof(uint32_t)))) return rc;
  if ((rc = c->blk_base.ensure(nblk * sizeof(uint32_t)))) return rc;
  if ((rc = c->blk_top.ensure((scan_blocks(cap) + 1) * 3 * sizeof(uint32_t)))) return rc;
  if ((rc = c->ctl.ensure(sizeof(FrameCtl)))) return rc;
  pool.org = (R*)c->org.p, pool.dir = (R*)c->dir.p, pool.hit_t = (R*)c->hit_t.p, pool.hit_uv = (R*)c->hit_uv.p;
  pool.hit_prim = (int32_t*)c->hit_prim.p, pool.color = (R*)c->color.p, pool.mixr = (R*)c->mixr.p;
  pool.child = (int32_t*)c->child.p;
  pool.blk_hits = (uint32_t*)c->blk_hits.p, pool.blk_children = (uint32_t*)c->blk_children.p, pool.blk_base = (uint32_t*)c->blk_base.p;
  pool.blk_top = (uint32_t*)c->blk_top.p;
  pool.views = nullptr, pool.n_view_rays = 0, pool.rays_per_view = 1, pool.n_views = 1, pool.inv_rays_per_view = 1.f;
  pool.capacity = (int64_t)cap;
  return RT355_OK;
}

// The same pool laid out in a caller-independent fixed buffer: nothing is allocated, the capacity is whatever
// fits.  Returns RT355_ECAPACITY when the frame needs more than the buffer holds.
template <class R>
int carve_pool(rt355_ctx* c, void* base, size_t bytes, uint64_t n_primary, bool secondary, RayPool<R>& pool, FrameCtl** ctl) {
  const double factor = secondary ? std::max(1.0, c->pool_factor) : 1.0;
  const uint64_t cap = (uint64_t)std::ceil((double)n_primary * factor) + 64;
  const uint64_t nblk = cap / GROUP + 8;
  size_t off = 0;
  auto take = [&](size_t n) {
    void* p = (unsigned char*)base + off;
    off += (n + 255) & ~(size_t)255;
    return p;
  };
  *ctl = (FrameCtl*)take(sizeof(FrameCtl));
  pool.org = (R*)take(cap * 3 * sizeof(R)), pool.dir = (R*)take(cap * 3 * sizeof(R));
  pool.hit_t = (R*)take(cap * sizeof(R)), pool.hit_uv = (R*)take(cap * 2 * sizeof(R));
  pool.hit_prim = (int32_t*)take(cap * 2 * sizeof(int32_t)), pool.color = (R*)take(cap * 3 * sizeof(R));
  pool.mixr = (R*)take(cap * sizeof(R)), pool.child = (int32_t*)take(cap * 2 * sizeof(int32_t));
  pool.blk_hits = (uint32_t*)take(nblk * sizeof(uint32_t)), pool.blk_children = (uint32_t*)take(nblk * sizeof(uint32_t));
  pool.blk_base = (uint32_t*)take(nblk * sizeof(uint32_t));
  pool.blk_top = (uint32_t*)take((scan_blocks(cap) + 1) * 3 * sizeof(uint32_t));
  pool.views = nullptr, pool.n_view_rays = 0, pool.rays_per_view = 1, pool.n_views = 1, pool.inv_rays_per_view = 1.f;
  pool.capacity = (int64_t)cap;
  if (off > bytes)
    return set_err(RT355_ECAPACITY, "frame too large for the fixed workspace: " + std::to_string(off) + " bytes needed, " + std::to_string(bytes) +
                                        " available (" + std::to_string(n_primary) + " primary rays x pool factor " + std::to_string(factor) + ")");
  return RT355_OK;
}



// run_generations with the kernels the context's counting mode and the scene's contents ask for
template <class R>
int run_generations_mode(rt355_ctx* c, const rt355_scene* s, RayPool<R>& pool, FrameCtl* ctl, uint32_t n_primary, int max_depth, const DevCamera<R>& cam, bool primary,
                         int out_mode, float* out_dev, LaunchLog& log) {
  const bool inst = s->n_instances > 0;
  if (c->counting) {
    return inst ? run_generations<R, M_COUNT | M_INST>(c, s, pool, ctl, n_primary, max_depth, cam, primary, out_mode, out_dev, log)
                : run_generations<R, M_COUNT>(c, s, pool, ctl, n_primary, max_depth, cam, primary, out_mode, out_dev, log);
  }
  return inst ? run_generations<R, M_INST>(c, s, pool, ctl, n_primary, max_depth, cam, primary, out_mode, out_dev, log)
              : run_generations<R, 0>(c, s, pool, ctl, n_primary, max_depth, cam, primary, out_mode, out_dev, log);
}

// After the stream has been synchronised behind run_generations.
int finish_stats(rt355_ctx* c, const LaunchLog& log, double frame_ms, rt355_stats* st, bool* overflow) {
  const FrameCtl& h = *c->ctl_host;
  *overflow = h.overflow.v != 0;
  if (!st) return RT355_OK;
  memset(st, 0, sizeof(*st));
  uint64_t hits_of[RT355_MAX_GEN + 1] = {0};
  for (int g = 0; g < RT355_MAX_GEN; ++g) {
    const uint64_t hits = h.hit_count[g].v;
    hits_of[g] = hits;
    st->radiance_rays[g] = h.gen_count[g].v - h.holes[g].v;
    st->hits[g] = hits;
    st->total_radiance += h.gen_count[g].v - h.holes[g].v;
    st->total_shadow += hits;
  }
  st->frame_ms = frame_ms;
  st->n_launch = log.n;
  for (int i = 0; i < log.n && i < RT355_MAX_LAUNCH; ++i) {
    st->launch_kind[i] = log.kind[i], st->launch_gen[i] = log.gen[i];
    if (c->timing) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
      st->launch_ms[i] = ms;
    }
    if (c->counting) {
      const unsigned long long* w = h.work[i];
      st->launch_box_tests[i] = w[0];
      uint64_t prims = 0;
      double fl = F_BOX * (double)w[0] + F_XFORM * (double)w[1];
      for (int k = 0; k < 8; ++k) prims += w[2 + k], fl += F_KIND[k] * (double)w[2 + k];
      st->launch_prim_tests[i] = prims;
      st->launch_max_box_tests[i] = w[10];
      st->launch_max_wave_clocks[i] = w[11];
      const int g = log.gen[i];
      if (log.kind[i] == RT355_K_SHADE || (log.kind[i] == RT355_K_TRACE && g == log.depth && log.shaded_in_trace))
        fl += F_SHADE * (double)hits_of[g];  // over_point + lighting of every hit
      // reflect / refract of the spawned rays: charged to the launch that writes the reflected ones
      if ((log.kind[i] == RT355_K_SPAWN && !log.merged[g]) || (log.kind[i] == RT355_K_SHADE && log.merged[g]))
        fl += 0.5 * (F_REFLECT + F_REFRACT) * (double)h.gen_count[g + 1].v;
      if (log.kind[i] == RT355_K_COMBINE) fl += 8.0 * (double)hits_of[g];  // surface + refl * R + refr * (1 - R)
      st->launch_flops[i] = fl;
    }
  }
  return RT355_OK;
}

// Waits for the frame queued on the context, then reads its statistics and its overflow flag.
int end_frame(rt355_ctx* c, rt355_stats* st) {
  if (!c->pending) return set_err(RT355_EINVAL, "no frame in flight on this context");
  c->pending = false;
  HIP_TRY(hipStreamSynchronize(c->stream));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c->pending_t0).count();
  bool overflow = false;
  int rc = finish_stats(c, c->pending_log, ms, st, &overflow);
  if (rc) return rc;
  // primary-ray generation is charged to the first trace launch; padding pixels are not counted
  if (st && c->counting && st->n_launch > 0) st->launch_flops[0] += c->pending_primary_flops;
  if (overflow) return set_err(RT355_ECAPACITY, "secondary-ray pool exhausted: raise rt355_ctx_set_pool");
  return RT355_OK;
}

// `n_frames` cameras of one size: one launch chain renders them all (generation 0 holds the frames one after
// the other), into out_dev frame after frame.
template <class R>
int render_impl(rt355_ctx* c, const rt355_scene* s, const rt355_camera_desc* cam, int max_depth, int rank, int n_ranks,
                int out_mode, float* out_dev, rt355_stats* st, bool fixed_workspace = false, bool begin_only = false, int n_frames = 1) {
  if (c->pending) return set_err(RT355_EINVAL, "a frame begun on this context has not been ended (rt355_frame_end)");
  if (n_frames < 1 || n_frames > RT355_MAX_BATCH) return set_err(RT355_EINVAL, "batch: between 1 and RT355_MAX_BATCH frames");
  DevCamera<R> dc;
  int rc = dev_camera<R>(cam, n_frames, rank, n_ranks, dc, (DevView<R>*)c->views_host);
  if (rc) return rc;
  const int64_t tpr = tiles_per_rank_of(dc.width, dc.height, n_ranks);
  const uint64_t n_primary = (uint64_t)tpr * TILE_PIX * (uint64_t)n_frames;
  RayPool<R> pool;
  FrameCtl* ctl = nullptr;
  if (fixed_workspace) {
    if ((rc = carve_pool<R>(c, c->fixed_scratch, RT355_NOALLOC_BYTES, n_primary, s->any_secondary && max_depth > 0, pool, &ctl))) return rc;
  } else {
    if ((rc = ensure_pool<R>(c, n_primary, s->any_secondary && max_depth > 0, pool))) return rc;
    ctl = (FrameCtl*)c->ctl.p;
  }
  // the views go to device memory (generation 0 reads its ray origins there, a batch all of each frame's view); the
  // staging buffer is free: the context's previous chain was waited for in end_frame
  HIP_TRY(hipMemcpyAsync(c->views_dev, c->views_host, (size_t)n_frames * sizeof(DevView<R>), hipMemcpyHostToDevice, c->stream));
  dc.views = (const DevView<R>*)c->views_dev;
  pool.views = dc.views, pool.n_view_rays = (uint32_t)n_primary, pool.rays_per_view = (uint32_t)(tpr * TILE_PIX), pool.n_views = (uint32_t)n_frames;
  pool.inv_rays_per_view = 1.f / (float)pool.rays_per_view;
  c->throughput_chain = begin_only || n_frames > 1;
  auto t0 = std::chrono::steady_clock::now();
  LaunchLog log;
  rc = run_generations_mode<R>(c, s, pool, ctl, (uint32_t)n_primary, max_depth, dc, true, out_mode, out_dev, log);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  c->pending = true, c->pending_log = log, c->pending_t0 = t0;
  c->pending_primary_flops = F_PRIMARY * (double)dc.width * dc.height / n_ranks * n_frames;
  return begin_only ? RT355_OK : end_frame(c, st);
}

template <class R>
int rays_impl(rt355_ctx* c, const rt355_scene* s, const float* rays_host, size_t n, int max_depth, bool trace_only,
              float* rgb_out, double* t_out, int32_t* idx_out, rt355_stats* st) {
  if (n == 0) return RT355_OK;
  RayPool<R> pool;
  int rc;
  if ((rc = ensure_pool<R>(c, n, !trace_only && s->any_secondary && max_depth > 0, pool))) return rc;
  if ((rc = c->io_in.ensure(n * 6 * sizeof(float)))) return rc;
  HIP_TRY(hipMemcpyAsync(c->io_in.p, rays_host, n * 6 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  const int block = BLOCK;
  const uint32_t grid0 = (uint32_t)((n + block - 1) / block);
  hipLaunchKernelGGL((k_load_rays<R>), dim3(grid0), dim3(block), 0, c->stream, (const float*)c->io_in.p, pool, (uint32_t)n);
  LaunchLog log;
  bool overflow = false;
  if (trace_only) {
    DevScene<R> sc = dev_scene<R>(s);
    hipLaunchKernelGGL(k_reset_ctl, dim3(1), dim3(256), 0, c->stream, (FrameCtl*)c->ctl.p, (uint32_t)n);
    if (s->n_instances > 0)
      hipLaunchKernelGGL((k_trace<R, M_INST, false, false>), dim3(grid0), dim3(block), 0, c->stream, sc, pool, (FrameCtl*)c->ctl.p, 0, 0, 0, DevCamera<R>{}, 0);
    else if (c->quad & 1)
      hipLaunchKernelGGL((k_trace_q<R, 0>), dim3(grid0), dim3(block), 0, c->stream, sc, pool, (FrameCtl*)c->ctl.p, 0, 0, 0, DevCamera<R>{}, 0);
    else
      hipLaunchKernelGGL((k_trace<R, 0, false, false>), dim3(grid0), dim3(block), 0, c->stream, sc, pool, (FrameCtl*)c->ctl.p, 0, 0, 0, DevCamera<R>{}, 0);
    if ((rc = c->io_out.ensure(n * sizeof(double)))) return rc;
    if ((rc = c->io_out2.ensure(n * sizeof(int32_t)))) return rc;
    hipLaunchKernelGGL((k_export_hits<R>), dim3(grid0), dim3(block), 0, c->stream, sc, pool, (uint32_t)n, (double*)c->io_out.p, (int32_t*)c->io_out2.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(t_out, c->io_out.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(idx_out, c->io_out2.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT355_OK;
  }
  c->throughput_chain = false;
  auto t0 = std::chrono::steady_clock::now();
  if ((rc = c->io_out.ensure(n * 3 * sizeof(float)))) return rc;
  DevCamera<R> dummy;
  memset(&dummy, 0, sizeof(dummy));
  rc = run_generations_mode<R>(c, s, pool, (FrameCtl*)c->ctl.p, (uint32_t)n, max_depth, dummy, false, (int)OUT_LINEAR, (float*)c->io_out.p, log);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(rgb_out, c->io_out.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if ((rc = finish_stats(c, log, ms, st, &overflow))) return rc;
  if (overflow) return set_err(RT355_ECAPACITY, "secondary-ray pool exhausted: raise rt355_ctx_set_pool");
  return RT355_OK;
}

template <class R>
int upload_impl(rt355_ctx* c, const HostScene& hs, rt355_scene* s) {
  const size_t np = hs.prims.size(), nn = hs.nodes.size(), nm = hs.materials.size();
  std::vector<BvhNode4> nodes(nn);
  s->box_abs_max = 0.f;
  for (size_t i = 0; i < nn; ++i) {
    // boxes were padded in double; when narrowing to float round outward
    auto down = [](double v) { float r = (float)v; if ((double)r > v) r = std::nextafter(r, -std::numeric_limits<float>::infinity()); return r; };
    auto up = [](double v) { float r = (float)v; if ((double)r < v) r = std::nextafter(r, std::numeric_limits<float>::infinity()); return r; };
    for (int k = 0; k < BVH_WIDTH; ++k) {
      for (int a = 0; a < 3; ++a) {
        nodes[i].box[k][2 * a] = down(hs.nodes[i].lo[k][a]), nodes[i].box[k][2 * a + 1] = up(hs.nodes[i].hi[k][a]);
        // the largest coordinate of any box (the top tree's in world space, the instanced meshes' in their own)
        for (float v : {nodes[i].box[k][2 * a], nodes[i].box[k][2 * a + 1]})
          if (std::isfinite(v)) s->box_abs_max = std::max(s->box_abs_max, std::fabs(v));
      }
      nodes[i].c[k] = hs.nodes[i].c[k];
      if (nodes[i].c[k] == -1) {  // empty slot: both x planes at +inf, which no ray interval with finite ends overlaps
        nodes[i].box[k][0] = nodes[i].box[k][1] = std::numeric_limits<float>::infinity();
        for (int a = 2; a < 6; ++a) nodes[i].box[k][a] = 0.f;
      }
    }
    for (int k = 0; k < 4; ++k) nodes[i].pad[k] = 0;
  }
  if ((int64_t)nm > MAX_MATERIALS) return set_err(RT355_ECAPACITY, "more than " + std::to_string(MAX_MATERIALS) + " distinct materials");
  std::vector<PrimInfo> info(np);
  int64_t n_secondary = 0;
  std::vector<DevXform<R>> xf(np);
  bool any_tri = false;
  for (size_t i = 0; i < np; ++i) {
    const HostPrim& p = hs.prims[i];
    info[i].shape = p.shape;
    info[i].flags = p.flags | (p.material << PF_MAT_SHIFT);
    if (hs.materials[p.material].reflective != 0.f) info[i].flags |= PF_REFLECTIVE;
    if (hs.materials[p.material].transparency != 0.f) info[i].flags |= PF_TRANSPARENT;
    if (hs.materials[p.material].reflective > 0.f && hs.materials[p.material].transparency > 0.f) info[i].flags |= PF_SCHLICK;
    if (info[i].flags & (PF_REFLECTIVE | PF_TRANSPARENT)) n_secondary++;
    for (int k = 0; k < 12; ++k) xf[i].r[k] = (R)p.inv[k];
    xf[i].ymin = (R)p.ymin, xf[i].ymax = (R)p.ymax, xf[i].root = (R)(p.blas_root >= 0 ? p.blas_root : 0), xf[i].pad1 = 0;
    if ((p.flags & PF_KIND_MASK) >= RT355_TRIANGLE) any_tri = true;
  }
  size_t n_optical = 0;  // prims whose refractive index is not 1 (the default, src/material.rs:31-72)
  for (size_t i = 0; i < np; ++i) n_optical += hs.materials[hs.prims[i].material].refractive != 1.f;
  for (size_t i = 0; i < np; ++i)
    if (n_optical == (hs.materials[hs.prims[i].material].refractive != 1.f ? 1u : 0u)) info[i].flags |= PF_OTHERS_UNIT;
  // PF_ISOLATED.  The n1 / n2 search of a ray that leaves a transparent shape looks for intersections of OTHER shapes on
  // the stretch between two of the shape's own intersections.  Both ends lie on the shape, so the stretch lies inside the
  // shape's bounding box; when no other prim can have an intersection there, the search is skipped.  Decided here, once:
  // bounded prims by their boxes (with a margin far above the rounding of any intersection), planes by the side of the
  // plane the box's corners lie on; any other unbounded prim counts as reaching everywhere.  Spheres, cubes and
  // cylinders only: a cone's parallel-ray intersection (src/shape.rs:295-298) can lie outside its box.  Few transparent
  // prims only: the test is quadratic.
  {
    std::vector<size_t> cand;
    const size_t n_top = (size_t)hs.n_top;
    for (size_t i = 0; i < n_top && cand.size() <= 64; ++i) {
      const int kind = hs.prims[i].flags & PF_KIND_MASK;
      if ((info[i].flags & PF_TRANSPARENT) && (info[i].flags & PF_OTHERS_UNIT) && hs.prims[i].bounded &&
          (kind == RT355_SPHERE || kind == RT355_CUBE || kind == RT355_CYLINDER))
        cand.push_back(i);
    }
    if (cand.size() <= 64)
      for (size_t i : cand) {
        const HostPrim& a = hs.prims[i];
        double lo[3], hi[3];
        for (int k = 0; k < 3; ++k) {
          const double m = 1e-3 + 1e-6 * std::max(std::fabs(a.lo[k]), std::fabs(a.hi[k]));
          lo[k] = a.lo[k] - m, hi[k] = a.hi[k] + m;
        }
        bool isolated = true;
        for (size_t j = 0; j < n_top && isolated; ++j) {
          if (j == i) continue;
          const HostPrim& b = hs.prims[j];
          if (b.bounded) {
            bool apart = false;
            for (int k = 0; k < 3; ++k) apart = apart || b.hi[k] < lo[k] || b.lo[k] > hi[k];
            isolated = apart;
          } else if ((b.flags & PF_KIND_MASK) == RT355_PLANE) {
            // object-space height of the box's corners above the plane y = 0 (row 1 of the inverse transformation)
            int above = 0, below = 0;
            for (int c8 = 0; c8 < 8; ++c8) {
              const double x = (c8 & 1) ? hi[0] : lo[0], y = (c8 & 2) ? hi[1] : lo[1], z = (c8 & 4) ? hi[2] : lo[2];
              const double h = b.inv[4] * x + b.inv[5] * y + b.inv[6] * z + b.inv[7];
              const double m = 1e-6 * (std::fabs(b.inv[4] * x) + std::fabs(b.inv[5] * y) + std::fabs(b.inv[6] * z) + std::fabs(b.inv[7])) + 1e-12;
              above += h > m, below += h < -m;
            }
            isolated = above == 8 || below == 8;
          } else {
            isolated = false;
          }
        }
        if (isolated) info[i].flags |= PF_ISOLATED;
      }
  }
  std::vector<DevTri> tri(any_tri ? np : 0);
  std::vector<DevTriNormals<R>> trin(any_tri ? np : 0);
  if (any_tri)
    for (size_t i = 0; i < np; ++i) {
      const HostPrim& p = hs.prims[i];
      tri[i].shape = info[i].shape, tri[i].flags = info[i].flags, tri[i].pad = 0;
      for (int k = 0; k < 3; ++k) {
        tri[i].p1[k] = (float)p.p1[k], tri[i].p2[k] = (float)p.p2[k], tri[i].p3[k] = (float)p.p3[k];
        trin[i].n1[k] = (R)p.n1[k], trin[i].n2[k] = (R)p.n2[k], trin[i].n3[k] = (R)p.n3[k], trin[i].nf[k] = (R)p.nf[k];
      }
    }
  std::vector<DevMaterial<R>> mats(nm);
  for (size_t i = 0; i < nm; ++i) {
    const rt355_material& m = hs.materials[i];
    for (int k = 0; k < 3; ++k) mats[i].color[k] = (R)m.color[k], mats[i].c1[k] = (R)m.c1[k], mats[i].c2[k] = (R)m.c2[k];
    mats[i].ambient = (R)m.ambient, mats[i].diffuse = (R)m.diffuse, mats[i].specular = (R)m.specular;
    mats[i].shininess = (R)m.shininess, mats[i].reflective = (R)m.reflective, mats[i].refractive = (R)m.refractive;
    mats[i].transparency = (R)m.transparency, mats[i].pattern_kind = m.pattern_kind, mats[i].pad = 0;
  }
  auto put = [&](DevBuf& b, const void* src, size_t bytes) -> int {
    int rc = b.ensure(std::max<size_t>(bytes, 16));
    if (rc) return rc;
    if (bytes) HIP_TRY(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    s->device_bytes += bytes;
    return RT355_OK;
  };
  int rc;
  if ((rc = put(s->nodes, nodes.data(), nn * sizeof(BvhNode4)))) return rc;
  if ((rc = put(s->prim_info, info.data(), np * sizeof(PrimInfo)))) return rc;
  if ((rc = put(s->xform, xf.data(), np * sizeof(DevXform<R>)))) return rc;
  if ((rc = put(s->tri, tri.data(), tri.size() * sizeof(DevTri)))) return rc;
  if ((rc = put(s->trin, trin.data(), trin.size() * sizeof(DevTriNormals<R>)))) return rc;
  if ((rc = put(s->materials, mats.data(), nm * sizeof(DevMaterial<R>)))) return rc;
  if (hs.n_instances > 0) {
    if (nn >= (1u << 24)) return set_err(RT355_EINVAL, "too many BVH nodes for a scene with instanced meshes");  // roots are kept as whole numbers in R
    std::vector<int32_t> inst_of(np);
    for (size_t i = 0; i < np; ++i) inst_of[i] = hs.prims[i].inst_prim;
    if ((rc = put(s->inst_of, inst_of.data(), np * sizeof(int32_t)))) return rc;
  }
  s->n_nodes = (int32_t)nn, s->n_bounded = hs.n_bounded, s->n_top = hs.n_top, s->n_prims = (int32_t)np, s->n_materials = (int32_t)nm;
  s->n_instances = hs.n_instances;
  s->max_depth = hs.max_depth, s->any_secondary = hs.any_secondary, s->any_transparent = hs.any_transparent;
  s->n_secondary_prims = n_secondary;
  return RT355_OK;
}

int check_ctx(rt355_ctx* c) {
  if (!c) return set_err(RT355_EINVAL, "null context");
  HIP_TRY(hipSetDevice(c->device));
  return RT355_OK;
}

// n_frames > 1: every rank's buffer holds the tiles of n_frames frames, frame after frame (the layout
// rt355_render_tiles_batch_begin writes); out_dev receives the frames one canvas after the other.
template <class T>
int untile_impl(rt355_ctx* c, const rt355_camera_desc* cam, int n_ranks, int n_frames, const T* gathered_dev, T* out_dev) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!cam || !gathered_dev || !out_dev || n_ranks < 1 || n_frames < 1 || n_frames > RT355_MAX_BATCH) return set_err(RT355_EINVAL, "bad argument");
  int w = 0, h = 0;
  if ((rc = camera_dims(cam, w, h))) return rc;
  const int tiles_x = (w + TILE - 1) / TILE, tiles_y = (h + TILE - 1) / TILE;
  const long long total = (long long)w * h * n_frames;
  hipLaunchKernelGGL((k_untile<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, gathered_dev, out_dev, w, h, tiles_x,
                     tiles_x * tiles_y, n_ranks, (long long)tiles_per_rank_of(w, h, n_ranks), n_frames);
  HIP_TRY(hipGetLastError());
  if (!c->async_helpers) HIP_TRY(hipStreamSynchronize(c->stream));
  return RT355_OK;
}

}  // namespace

extern "C" {

const char* rt355_last_error(void) { return g_err.c_str(); }
int rt355_version(void) { return RT355_VERSION; }

int rt355_ctx_create(rt355_ctx** out, int device_id, int real_bits) { return rt355_ctx_create_ex(out, device_id, real_bits, 0); }

int rt355_ctx_create_ex(rt355_ctx** out, int device_id, int real_bits, int flags) {
  if (!out) return set_err(RT355_EINVAL, "null out pointer");
  if (flags & ~RT355_CTX_ONE_STREAM) return set_err(RT355_EINVAL, "unknown context flag");
  *out = nullptr;
  if (real_bits != 32 && real_bits != 64) return set_err(RT355_EINVAL, "real_bits must be 32 or 64");
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device_id < 0 || device_id >= n) return set_err(RT355_EINVAL, "no such device");
  HIP_TRY(hipSetDevice(device_id));
  rt355_ctx* c = new rt355_ctx();
  c->device = device_id, c->real_bits = real_bits;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
  c->grid0_cap = (uint32_t)c->n_cu * 32 * (256 / BLOCK);
  // Main stream: the chain every generation waits on, at the highest priority the device offers.  Side
  // stream: the shadow launches, at the lowest, so that their waves yield to the chain's (measured: 2 % of
  // the frame).
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  // One-stream contexts are meant to be used several at a time on one GPU (frames in flight): all their
  // launches go to one stream of default priority, so that the frames share the device evenly and each context
  // occupies one hardware queue instead of two (measured: three such frames in flight give the best rate).
  if (flags & RT355_CTX_ONE_STREAM) prio_least = prio_greatest = 0, c->use_side = 0;
  bool ok = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest) == hipSuccess;
  // no side stream for a one-stream context: streams are spread over a handful of hardware queues, and two
  // chains that land on the same queue run one after the other
  if (c->use_side) ok = ok && hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_least) == hipSuccess;
  for (int i = 0; i <= RT355_MAX_LAUNCH && ok; ++i) ok = hipEventCreate(&c->ev[i]) == hipSuccess;
  for (int i = 0; i < RT355_MAX_GEN && ok; ++i) ok = hipEventCreateWithFlags(&c->ev_traced[i], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ev_side_done, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    rt355_ctx_destroy(c);  // releases whatever was created
    return set_err(RT355_EHIP, "stream / event creation failed");
  }
  if (hipHostMalloc((void**)&c->ctl_host, sizeof(FrameCtl), hipHostMallocDefault) != hipSuccess) {
    c->ctl_host = nullptr;
    rt355_ctx_destroy(c);
    return set_err(RT355_ENOMEM, "pinned host allocation failed");
  }
  const size_t views_bytes = (size_t)RT355_MAX_BATCH * sizeof(DevView<double>);
  if (hipHostMalloc(&c->views_host, views_bytes, hipHostMallocDefault) != hipSuccess || hipMalloc(&c->views_dev, views_bytes) != hipSuccess) {
    rt355_ctx_destroy(c);
    return set_err(RT355_ENOMEM, "allocation of the batch view table failed");
  }
  // the fixed workspace of rt355_render_device_noalloc: looked up here, where the device is current and a set-up call may
  // still fail softly.  Without it (a device id beyond the table, a failed look-up) the context works for every
  // other entry point and the no-allocation one reports the reason.
  if (scratch_address(device_id, &c->fixed_scratch) != RT355_OK) c->fixed_scratch = nullptr, c->fixed_scratch_error = g_err;
  if (const char* e = getenv("RT355_PACKET_GENS")) c->packet_gens = std::max(0, std::min((int)RT355_MAX_GEN, atoi(e)));
  if (const char* e = getenv("RT355_COMBINE_LEVELS")) c->combine_levels = std::max(0, std::min(4, atoi(e)));
  if (const char* e = getenv("RT355_SHADOW_LAG")) c->shadow_lag = std::max(0, atoi(e));
  if (const char* e = getenv("RT355_QUAD")) c->quad = atoi(e) & 3;
  if (const char* e = getenv("RT355_MERGE_SPAWN")) c->merge_spawn = atoi(e) != 0;
  if (const char* e = getenv("RT355_GRID0_CAP")) c->grid0_cap = (uint32_t)std::max(0, atoi(e));
  *out = c;
  return RT355_OK;
}

void rt355_ctx_destroy(rt355_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->side) (void)hipStreamSynchronize(c->side);
  for (DevBuf* b : {&c->org, &c->dir, &c->hit_t, &c->hit_uv, &c->hit_prim, &c->color, &c->mixr, &c->child, &c->blk_hits, &c->blk_children, &c->blk_base, &c->blk_top, &c->ctl, &c->io_in, &c->io_out, &c->io_out2}) b->release();
  for (hipEvent_t e : c->ev)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_traced)
    if (e) (void)hipEventDestroy(e);
  if (c->ev_side_done) (void)hipEventDestroy(c->ev_side_done);
  if (c->ev_order) (void)hipEventDestroy(c->ev_order);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->ctl_host) (void)hipHostFree(c->ctl_host);
  if (c->views_host) (void)hipHostFree(c->views_host);
  if (c->views_dev) (void)hipFree(c->views_dev);
  delete c;
}

int rt355_ctx_set_pool(rt355_ctx* c, double rays_per_pixel) {
  if (!c || !(rays_per_pixel >= 1.0) || rays_per_pixel > 128.0) return set_err(RT355_EINVAL, "pool factor must be in [1,128]");
  c->pool_factor = rays_per_pixel;
  return RT355_OK;
}
int rt355_ctx_set_exact_cones(rt355_ctx* c, int mode) {
  if (!c || (mode != 0 && mode != 1)) return set_err(RT355_EINVAL, "mode must be 0 or 1");
  c->exact_cones = mode;
  return RT355_OK;
}
int rt355_ctx_set_counting(rt355_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 2) return set_err(RT355_EINVAL, "counting mode must be 0, 1 or 2");
  c->counting = mode;
  return RT355_OK;
}
int rt355_ctx_set_timing(rt355_ctx* c, int on) {
  if (!c) return set_err(RT355_EINVAL, "null context");
  c->timing = on != 0;
  return RT355_OK;
}

int rt355_scene_upload(rt355_ctx* c, const rt355_shape_desc* shapes, size_t n_shapes, const rt355_light_desc* light, rt355_scene** out) {
  if (!out) return set_err(RT355_EINVAL, "null out pointer");
  *out = nullptr;
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!shapes && n_shapes) return set_err(RT355_EINVAL, "null shapes");
  if (!light) return set_err(RT355_EINVAL, "null light");
  for (int k = 0; k < 3; ++k)
    if (std::isnan(light->position[k]) || std::isnan(light->intensity[k])) return set_err(RT355_EINVAL, "NaN in light");
  HostScene hs;
  char err[256] = {0};
  // outward padding of every BVH box, relative to its magnitude: a few hundred ulps of the kernel's type
  const double pad_rel = c->real_bits == 32 ? 256.0 * 1.1920929e-7 : 1e-9;
  rc = build_host_scene(shapes, n_shapes, pad_rel, c->exact_cones ? INT64_MAX : 32, hs, err, sizeof(err));
  if (rc) return set_err(rc, err);
  rt355_scene* s = new rt355_scene();
  s->device = c->device, s->real_bits = c->real_bits;
  for (int k = 0; k < 3; ++k) s->light_pos[k] = light->position[k], s->light_int[k] = light->intensity[k];
  rc = c->real_bits == 32 ? upload_impl<float>(c, hs, s) : upload_impl<double>(c, hs, s);
  if (rc) {
    rt355_scene_destroy(s);
    return rc;
  }
  *out = s;
  return RT355_OK;
}

void rt355_scene_destroy(rt355_scene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  for (DevBuf* b : {&s->nodes, &s->prim_info, &s->xform, &s->tri, &s->trin, &s->materials, &s->inst_of}) b->release();
  delete s;
}

int rt355_scene_info(const rt355_scene* s, uint64_t out[5]) {
  if (!s || !out) return set_err(RT355_EINVAL, "null argument");
  out[0] = s->n_nodes, out[1] = s->n_bounded, out[2] = s->n_top - s->n_bounded, out[3] = s->max_depth, out[4] = s->device_bytes;
  return RT355_OK;
}

int64_t rt355_tiles_per_rank(const rt355_camera_desc* cam, int n_ranks) {
  int w = 0, h = 0;
  if (n_ranks < 1 || camera_dims(cam, w, h)) return -1;
  return tiles_per_rank_of(w, h, n_ranks);
}

static int check_render_args(rt355_ctx* c, const rt355_scene* s, const rt355_camera_desc* cam, int max_depth) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!s || !cam) return set_err(RT355_EINVAL, "null argument");
  if (s->device != c->device || s->real_bits != c->real_bits) return set_err(RT355_EINVAL, "scene was uploaded on another device or for another precision");
  if (max_depth < 0 || max_depth >= RT355_MAX_GEN) return set_err(RT355_EINVAL, "max_depth must be in [0,7]");
  int w = 0, h = 0;
  return camera_dims(cam, w, h);
}

int rt355_render_tiles_begin(rt355_ctx* c, const rt355_scene* s, const rt355_camera_desc* cam, int max_depth, int rank, int n_ranks, float* tiles_out_dev) {
  int rc = check_render_args(c, s, cam, max_depth);
  if (rc) return rc;
  if (!tiles_out_dev) return set_err(RT355_EINVAL, "null output");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return set_err(RT355_EINVAL, "bad rank / n_ranks");
  return c->real_bits == 32 ? render_impl<float>(c, s, cam, max_depth, rank, n_ranks, OUT_TILES, tiles_out_dev, nullptr, false, true)
                            : render_impl<double>(c, s, cam, max_depth, rank, n_ranks, OUT_TILES, tiles_out_dev, nullptr, false, true);
}

int rt355_render_device_begin(rt355_ctx* c, const rt355_scene* s, const rt355_camera_desc* cam, int max_depth, float* rgb_out_dev) {
  int rc = check_render_args(c, s, cam, max_depth);
  if (rc) return rc;
  if (!rgb_out_dev) return set_err(RT355_EINVAL, "null output");
  return c->real_bits == 32 ? render_impl<float>(c, s, cam, max_depth, 0, 1, OUT_ROWMAJOR, rgb_out_dev, nullptr, false, true)
                            : render_impl<double>(c, s, cam, max_depth, 0, 1, OUT_ROWMAJOR, rgb_out_dev, nullptr, false, true);
}

static int check_batch_args(rt355_ctx* c, const rt355_scene* s, const rt355_camera_desc* cams, int n_frames, int max_depth) {
  if (n_frames < 1 || n_frames > RT355_MAX_BATCH) return set_err(RT355_EINVAL, "batch: between 1 and RT355_MAX_BATCH frames");
  return check_render_args(c, s, cams, max_depth);
}

int rt355_render_tiles_batch_begin(rt355_ctx* c, const rt355_scene* s, const rt355_camera_desc* cams, int n_frames, int max_depth, int rank, int n_ranks,
                                   float* tiles_out_dev) {
  int rc = check_batch_args(c, s, cams, n_frames, max_depth);
  if (rc) return rc;
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks || !tiles_out_dev) return set_err(RT355_EINVAL, "bad rank / n_ranks / output");
  return c->real_bits == 32 ? render_impl<float>(c, s, cams, max_depth, rank, n_ranks, OUT_TILES, tiles_out_dev, nullptr, false, true, n_frames)
                            : render_impl<double>(c, s, cams, max_depth, rank, n_ranks, OUT_TILES, tiles_out_dev, nullptr, false, true, n_frames);
}

int rt355_render_device_batch_begin(rt355_ctx* c, const rt355_scene* s, const rt355_camera_desc* cams, int n_frames, int max_depth, float* rgb_out_dev) {
  int rc = check_batch_args(c, s, cams, n_frames, max_depth);
  if (rc) return rc;
  if (!rgb_out_dev) return set_err(RT355_EINVAL, "null output");
  return c->real_bits == 32 ? render_impl<float>(c, s, cams, max_depth, 0, 1, OUT_ROWMAJOR, rgb_out_dev, nullptr, false, true, n_frames)
                            : render_impl<double>(c, s, cams, max_depth, 0, 1, OUT_ROWMAJOR, rgb_out_dev, nullptr, false, true, n_frames);
}

int rt355_frame_end(rt355_ctx* c, rt355_stats* stats) {
  int rc = check_ctx(c);
  if (rc) return rc;
  return end_frame(c, stats);
}

int rt355_render(rt355_ctx* c, const rt355_scene* s, const rt355_camera_desc* cam, int max_depth, float* rgb_out, rt355_stats* stats) {
  int rc = check_render_args(c, s, cam, max_depth);
  if (rc) return rc;
  if (!rgb_out) return set_err(RT355_EINVAL, "null output");
  const size_t n = (size_t)(int)cam->hsize * (size_t)(int)cam->vsize * 3;
  if ((rc = c->io_out.ensure(n * sizeof(float)))) return rc;
  rc = c->real_bits == 32 ? render_impl<float>(c, s, cam, max_depth, 0, 1, OUT_ROWMAJOR, (float*)c->io_out.p, stats)
                          : render_impl<double>(c, s, cam, max_depth, 0, 1, OUT_ROWMAJOR, (float*)c->io_out.p, stats);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(rgb_out, c->io_out.p, n * sizeof(float), hipMemcpyDeviceToHost));
  return RT355_OK;
}

int rt355_render_batch(rt355_ctx* c, const rt355_scene* s, const rt355_camera_desc* cams, int n_frames, int max_depth, float* rgb_out, rt355_stats* stats) {
  int rc = check_batch_args(c, s, cams, n_frames, max_depth);
  if (rc) return rc;
  if (!rgb_out) return set_err(RT355_EINVAL, "null output");
  const size_t n = (size_t)(int)cams->hsize * (size_t)(int)cams->vsize * 3 * (size_t)n_frames;
  if ((rc = c->io_out.ensure(n * sizeof(float)))) return rc;
  rc = c->real_bits == 32 ? render_impl<float>(c, s, cams, max_depth, 0, 1, OUT_ROWMAJOR, (float*)c->io_out.p, stats, false, false, n_frames)
                          : render_impl<double>(c, s, cams, max_depth, 0, 1, OUT_ROWMAJOR, (float*)c->io_out.p, stats, false, false, n_frames);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(rgb_out, c->io_out.p, n * sizeof(float), hipMemcpyDeviceToHost));
  return RT355_OK;
}

int rt355_render_u8(rt355_ctx* c, const rt355_scene* s, const rt355_camera_desc* cam, int max_depth, uint8_t* rgb_out, rt355_stats* stats) {
  int rc = check_render_args(c, s, cam, max_depth);
  if (rc) return rc;
  if (!rgb_out) return set_err(RT355_EINVAL, "null output");
  const size_t n = (size_t)(int)cam->hsize * (size_t)(int)cam->vsize * 3;
  if ((rc = c->io_out.ensure(n * sizeof(float)))) return rc;
  if ((rc = c->io_out2.ensure(n))) return rc;
  rc = c->real_bits == 32 ? render_impl<float>(c, s, cam, max_depth, 0, 1, OUT_ROWMAJOR, (float*)c->io_out.p, stats)
                          : render_impl<double>(c, s, cam, max_depth, 0, 1, OUT_ROWMAJOR, (float*)c->io_out.p, stats);
  if (rc) return rc;
  hipLaunchKernelGGL(k_to_u8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float*)c->io_out.p, n, (uint8_t*)c->io_out2.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(rgb_out, c->io_out2.p, n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT355_OK;
}

void* rt355_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)set_err(RT355_ENOMEM, "pinned host allocation failed");
    return nullptr;
  }
  return p;
}
void rt355_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int rt355_set_error(int code, const char* msg) { return set_err(code, msg ? msg : ""); }

int rt355_render_device(rt355_ctx* c, const rt355_scene* s, const rt355_camera_desc* cam, int max_depth, float* rgb_out_dev, rt355_stats* stats) {
  int rc = check_render_args(c, s, cam, max_depth);
  if (rc) return rc;
  if (!rgb_out_dev) return set_err(RT355_EINVAL, "null output");
  return c->real_bits == 32 ? render_impl<float>(c, s, cam, max_depth, 0, 1, OUT_ROWMAJOR, rgb_out_dev, stats)
                            : render_impl<double>(c, s, cam, max_depth, 0, 1, OUT_ROWMAJOR, rgb_out_dev, stats);
}

int rt355_render_device_noalloc(rt355_ctx* c, const rt355_scene* s, const rt355_camera_desc* cam, int max_depth, float* rgb_out_dev, rt355_stats* stats) {
  int rc = check_render_args(c, s, cam, max_depth);
  if (rc) return rc;
  if (!rgb_out_dev) return set_err(RT355_EINVAL, "null output");
  if (!c->fixed_scratch) return set_err(RT355_EHIP, "fixed workspace unavailable on this context: " + c->fixed_scratch_error);
  // frames larger than the workspace are refused by carve_pool, before anything is launched
  return c->real_bits == 32 ? render_impl<float>(c, s, cam, max_depth, 0, 1, OUT_ROWMAJOR, rgb_out_dev, stats, true)
                            : render_impl<double>(c, s, cam, max_depth, 0, 1, OUT_ROWMAJOR, rgb_out_dev, stats, true);
}

int rt355_render_tiles(rt355_ctx* c, const rt355_scene* s, const rt355_camera_desc* cam, int max_depth, int rank, int n_ranks,
                       float* tiles_out_dev, rt355_stats* stats) {
  int rc = check_render_args(c, s, cam, max_depth);
  if (rc) return rc;
  if (!tiles_out_dev) return set_err(RT355_EINVAL, "null output");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return set_err(RT355_EINVAL, "bad rank / n_ranks");
  return c->real_bits == 32 ? render_impl<float>(c, s, cam, max_depth, rank, n_ranks, OUT_TILES, tiles_out_dev, stats)
                            : render_impl<double>(c, s, cam, max_depth, rank, n_ranks, OUT_TILES, tiles_out_dev, stats);
}

int rt355_untile(rt355_ctx* c, const rt355_camera_desc* cam, int n_ranks, const float* gathered_dev, float* rgb_out_dev) {
  return untile_impl<float>(c, cam, n_ranks, 1, gathered_dev, rgb_out_dev);
}
int rt355_untile_u8(rt355_ctx* c, const rt355_camera_desc* cam, int n_ranks, const uint8_t* gathered_dev, uint8_t* rgb_out_dev) {
  return untile_impl<uint8_t>(c, cam, n_ranks, 1, gathered_dev, rgb_out_dev);
}
int rt355_untile_batch(rt355_ctx* c, const rt355_camera_desc* cam, int n_ranks, int n_frames, const float* gathered_dev, float* rgb_out_dev) {
  return untile_impl<float>(c, cam, n_ranks, n_frames, gathered_dev, rgb_out_dev);
}
int rt355_untile_batch_u8(rt355_ctx* c, const rt355_camera_desc* cam, int n_ranks, int n_frames, const uint8_t* gathered_dev, uint8_t* rgb_out_dev) {
  return untile_impl<uint8_t>(c, cam, n_ranks, n_frames, gathered_dev, rgb_out_dev);
}

int rt355_color_at(rt355_ctx* c, const rt355_scene* s, const float* rays, size_t n, int max_depth, float* rgb_out, rt355_stats* stats) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!s || (!rays && n) || (!rgb_out && n)) return set_err(RT355_EINVAL, "null argument");
  if (s->device != c->device || s->real_bits != c->real_bits) return set_err(RT355_EINVAL, "scene was uploaded on another device or for another precision");
  if (max_depth < 0 || max_depth >= RT355_MAX_GEN) return set_err(RT355_EINVAL, "max_depth must be in [0,7]");
  return c->real_bits == 32 ? rays_impl<float>(c, s, rays, n, max_depth, false, rgb_out, nullptr, nullptr, stats)
                            : rays_impl<double>(c, s, rays, n, max_depth, false, rgb_out, nullptr, nullptr, stats);
}

int rt355_trace_sample(rt355_ctx* c, const rt355_scene* s, const float* rays, size_t n, double* t_out, int32_t* shape_index_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!s || (!rays && n) || ((!t_out || !shape_index_out) && n)) return set_err(RT355_EINVAL, "null argument");
  if (s->device != c->device || s->real_bits != c->real_bits) return set_err(RT355_EINVAL, "scene was uploaded on another device or for another precision");
  return c->real_bits == 32 ? rays_impl<float>(c, s, rays, n, 0, true, nullptr, t_out, shape_index_out, nullptr)
                            : rays_impl<double>(c, s, rays, n, 0, true, nullptr, t_out, shape_index_out, nullptr);
}

int rt355_is_shadowed(rt355_ctx* c, const rt355_scene* s, const float* points, size_t n, uint8_t* out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!s || (!points && n) || (!out && n)) return set_err(RT355_EINVAL, "null argument");
  if (s->device != c->device || s->real_bits != c->real_bits) return set_err(RT355_EINVAL, "scene was uploaded on another device or for another precision");
  if (n == 0) return RT355_OK;
  if (n >= 0x3fffffffull) return set_err(RT355_EINVAL, "too many points in one call");  // before anything is allocated or copied
  if ((rc = c->io_in.ensure(n * 3 * sizeof(float)))) return rc;
  if ((rc = c->io_out.ensure(n))) return rc;
  HIP_TRY(hipMemcpyAsync(c->io_in.p, points, n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK), grid_q = (unsigned)((n + QUADS - 1) / QUADS);
  const bool inst = s->n_instances > 0;
  auto launch_points = [&](auto real, auto mode) {
    using R = decltype(real);
    hipLaunchKernelGGL((k_shadow_points<R, decltype(mode)::value>), dim3(grid), dim3(BLOCK), 0, c->stream, dev_scene<R>(s), (const float*)c->io_in.p, (uint32_t)n, (uint8_t*)c->io_out.p);
  };
  if ((c->quad & 2) && !inst) {  // the traversal the frame's shadow launches use
    if (c->real_bits == 32)
      hipLaunchKernelGGL((k_shadow_points_q<float>), dim3(grid_q), dim3(BLOCK), 0, c->stream, dev_scene<float>(s), (const float*)c->io_in.p, (uint32_t)n, (uint8_t*)c->io_out.p);
    else
      hipLaunchKernelGGL((k_shadow_points_q<double>), dim3(grid_q), dim3(BLOCK), 0, c->stream, dev_scene<double>(s), (const float*)c->io_in.p, (uint32_t)n, (uint8_t*)c->io_out.p);
  } else if (c->real_bits == 32) {
    if (inst) launch_points(float(), std::integral_constant<int, M_INST>()); else launch_points(float(), std::integral_constant<int, 0>());
  } else {
    if (inst) launch_points(double(), std::integral_constant<int, M_INST>()); else launch_points(double(), std::integral_constant<int, 0>());
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, c->io_out.p, n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT355_OK;
}

int rt355_to_u8(rt355_ctx* c, const float* rgb, size_t n, uint8_t* out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if ((!rgb || !out) && n) return set_err(RT355_EINVAL, "null argument");
  if (n == 0) return RT355_OK;
  if ((rc = c->io_in.ensure(n * sizeof(float)))) return rc;
  if ((rc = c->io_out.ensure(n))) return rc;
  HIP_TRY(hipMemcpyAsync(c->io_in.p, rgb, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_to_u8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float*)c->io_in.p, n, (uint8_t*)c->io_out.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, c->io_out.p, n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT355_OK;
}

int rt355_to_u8_device(rt355_ctx* c, const float* rgb_dev, size_t n, uint8_t* out_dev) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if ((!rgb_dev || !out_dev) && n) return set_err(RT355_EINVAL, "null argument");
  if (n == 0) return RT355_OK;
  hipLaunchKernelGGL(k_to_u8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, rgb_dev, n, out_dev);
  HIP_TRY(hipGetLastError());
  if (!c->async_helpers) HIP_TRY(hipStreamSynchronize(c->stream));
  return RT355_OK;
}

int rt355_ctx_wait_stream(rt355_ctx* c, void* hip_stream) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->ev_order) HIP_TRY(hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(c->ev_order, (hipStream_t)hip_stream));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_order, 0));
  return RT355_OK;
}

int rt355_stream_wait_ctx(rt355_ctx* c, void* hip_stream) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (c->pending) return set_err(RT355_EINVAL, "a frame is in flight on this context: end it with rt355_frame_end first");
  if (!c->ev_order) HIP_TRY(hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
  // every entry point leaves the side stream joined to the main one, so the main stream's tail covers both
  HIP_TRY(hipEventRecord(c->ev_order, c->stream));
  HIP_TRY(hipStreamWaitEvent((hipStream_t)hip_stream, c->ev_order, 0));
  return RT355_OK;
}

int rt355_ctx_set_async(rt355_ctx* c, int on) {
  int rc = check_ctx(c);
  if (rc) return rc;
  c->async_helpers = on != 0;
  return RT355_OK;
}

int rt355_ctx_sync(rt355_ctx* c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT355_OK;
}

int rt355_scene_check(const rt355_shape_desc* shapes, size_t n_shapes, int real_bits, int exact_cones, uint64_t out[5]) {
  if ((!shapes && n_shapes) || !out) return set_err(RT355_EINVAL, "null argument");
  if (real_bits != 32 && real_bits != 64) return set_err(RT355_EINVAL, "real_bits must be 32 or 64");
  HostScene hs;
  char err[256] = {0};
  const double pad_rel = real_bits == 32 ? 256.0 * 1.1920929e-7 : 1e-9;
  int rc = build_host_scene(shapes, n_shapes, pad_rel, exact_cones ? INT64_MAX : 32, hs, err, sizeof(err));
  if (rc) return set_err(rc, err);
  if ((rc = verify_host_scene(hs, err, sizeof(err)))) return set_err(rc, err);
  out[0] = hs.nodes.size(), out[1] = (uint64_t)hs.n_bounded, out[2] = (uint64_t)(hs.n_top - hs.n_bounded);
  out[3] = (uint64_t)hs.max_depth, out[4] = hs.materials.size();
  return RT355_OK;
}

#ifdef RT355_EXP_STAMPS
// Diagnostic build only: copies out and clears the per-wave traversal records (see DbgRec).
int rt355_dbg_read(void* out, size_t bytes) {
  const size_t all = sizeof(DbgRec) * (size_t)RT355_MAX_LAUNCH * DBG_PER_LAUNCH;
  if (bytes < all) return (int)(all >> 20);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), all) != hipSuccess) return -1;
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_dbg)) != hipSuccess || hipMemset(p, 0, all) != hipSuccess) return -1;
  return 0;
}
#endif

void rt355_flop_table(double out[16]) {
  out[0] = F_PRIMARY, out[1] = F_BOX, out[2] = F_XFORM;
  for (int k = 0; k < 8; ++k) out[3 + k] = F_KIND[k];
  out[11] = F_SHADE, out[12] = F_REFLECT, out[13] = F_REFRACT, out[14] = F_SCHLICK, out[15] = 0;
}
}
