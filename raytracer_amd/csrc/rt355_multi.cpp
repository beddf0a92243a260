// rt355_multi.cpp -- one frame on several GPUs of one node, driven by ONE host process through the C ABI.
//
// The reference renders a frame with one call, Camera::render (src/camera.rs:74-88); a host that owns several
// GPUs keeps that shape: one rt355_multi holds a context and a replica of the scene per device, and one call
// renders the frame.  Pixels are independent (src/camera.rs:76-82), so the frame is cut into 32x32 tiles, tile
// t goes to device t % n, every device renders its tiles on a host thread of its own, and the only exchange is
// the gather of the tile buffers to the first device: RCCL send / receive pairs over xGMI (the library is
// looked up at run time, so a host without RCCL still loads this library), or peer copies where RCCL cannot be
// used (the same physical device listed twice, as the tests on a one-GPU box do).  The first device un-tiles.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "rt355_device.h"

extern "C" int rt355_set_error(int code, const char* msg);  // rt355.hip: records the message for rt355_last_error

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load() {
    if (handle) return true;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (handle) break;
    }
    if (!handle) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
    GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
    Send = (decltype(Send))dlsym(handle, "ncclSend");
    Recv = (decltype(Recv))dlsym(handle, "ncclRecv");
    GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
  }
};
Rccl g_rccl;

int fail(int code, const std::string& msg) { return rt355_set_error(code, msg.c_str()); }

#define MHIP(expr)                                                                        \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) return fail(RT355_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

}  // namespace

struct rt355_multi {
  int n = 0, real_bits = 64;
  std::vector<int> dev;
  std::vector<rt355_ctx*> ctx;
  std::vector<rt355_scene*> scene;
  std::vector<hipStream_t> stream;     // gather streams, one per device
  std::vector<ncclComm_t> comm;        // empty when the gather goes through peer copies
  std::vector<float*> tiles;           // per device > 0: its tile buffer (device 0 renders into `gathered`)
  size_t tiles_floats = 0;             // capacity of each tile buffer
  float* gathered = nullptr;           // on device 0: n tile buffers back to back
  float* canvas = nullptr;             // on device 0: the row-major canvas
  size_t canvas_floats = 0;
  std::string gather_kind;
};

extern "C" {

void rt355_multi_destroy(rt355_multi* m) {
  if (!m) return;
  for (int d = 0; d < (int)m->scene.size(); ++d)
    if (m->scene[d]) rt355_scene_destroy(m->scene[d]);
  for (size_t d = 0; d < m->comm.size(); ++d)
    if (m->comm[d] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comm[d]);
  for (int d = 0; d < (int)m->dev.size(); ++d) {
    (void)hipSetDevice(m->dev[d]);
    if (d < (int)m->tiles.size() && m->tiles[d]) (void)hipFree(m->tiles[d]);
    if (d < (int)m->stream.size() && m->stream[d]) (void)hipStreamDestroy(m->stream[d]);
    if (d == 0) {
      if (m->gathered) (void)hipFree(m->gathered);
      if (m->canvas) (void)hipFree(m->canvas);
    }
  }
  for (rt355_ctx* c : m->ctx)
    if (c) rt355_ctx_destroy(c);
  delete m;
}

int rt355_multi_create(rt355_multi** out, const int* device_ids, int n_devices, int real_bits) {
  if (!out) return fail(RT355_EINVAL, "null out pointer");
  *out = nullptr;
  if (!device_ids || n_devices < 1 || n_devices > 64) return fail(RT355_EINVAL, "n_devices must be in [1,64]");
  rt355_multi* m = new rt355_multi();
  m->n = n_devices, m->real_bits = real_bits;
  m->dev.assign(device_ids, device_ids + n_devices);
  m->ctx.assign(n_devices, nullptr), m->scene.assign(n_devices, nullptr), m->stream.assign(n_devices, nullptr), m->tiles.assign(n_devices, nullptr);
  for (int d = 0; d < n_devices; ++d) {
    int rc = rt355_ctx_create(&m->ctx[d], device_ids[d], real_bits);
    if (rc) {
      rt355_multi_destroy(m);
      return rc;
    }
    if (hipSetDevice(device_ids[d]) != hipSuccess || hipStreamCreateWithFlags(&m->stream[d], hipStreamNonBlocking) != hipSuccess) {
      rt355_multi_destroy(m);
      return fail(RT355_EHIP, "stream creation failed");
    }
  }
  // RCCL needs distinct devices; the same device listed twice (tests on a one-GPU box) gathers with copies
  const bool distinct = std::set<int>(m->dev.begin(), m->dev.end()).size() == (size_t)n_devices;
  m->gather_kind = n_devices == 1 ? "none" : "peer copies";
  if (n_devices > 1 && distinct && !getenv("RT355_MULTI_NO_RCCL") && g_rccl.load()) {
    m->comm.assign(n_devices, nullptr);
    ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), n_devices, m->dev.data());
    if (r == ncclSuccess) {
      m->gather_kind = "rccl";
    } else {
      m->comm.clear();  // fall back to peer copies
    }
  }
  if (m->comm.empty() && n_devices > 1) {
    (void)hipSetDevice(m->dev[0]);
    for (int d = 1; d < n_devices; ++d)
      if (m->dev[d] != m->dev[0]) (void)hipDeviceEnablePeerAccess(m->dev[d], 0);  // already enabled is fine
  }
  *out = m;
  return RT355_OK;
}

const char* rt355_multi_gather_kind(const rt355_multi* m) { return m ? m->gather_kind.c_str() : ""; }

int rt355_multi_upload(rt355_multi* m, const rt355_shape_desc* shapes, size_t n_shapes, const rt355_light_desc* light) {
  if (!m) return fail(RT355_EINVAL, "null handle");
  // the host-side work of an upload (inverses, BVH) is the same for every replica: one thread per device
  std::vector<int> rcs(m->n, 0);
  std::vector<std::string> errs(m->n);
  std::vector<rt355_scene*> fresh(m->n, nullptr);
  std::vector<std::thread> th;
  for (int d = 0; d < m->n; ++d)
    th.emplace_back([&, d]() {
      rcs[d] = rt355_scene_upload(m->ctx[d], shapes, n_shapes, light, &fresh[d]);
      if (rcs[d]) errs[d] = rt355_last_error();
    });
  for (auto& t : th) t.join();
  for (int d = 0; d < m->n; ++d)
    if (rcs[d]) {
      for (rt355_scene* s : fresh)
        if (s) rt355_scene_destroy(s);
      return fail(rcs[d], errs[d]);
    }
  for (int d = 0; d < m->n; ++d) {
    if (m->scene[d]) rt355_scene_destroy(m->scene[d]);
    m->scene[d] = fresh[d];
  }
  return RT355_OK;
}

int rt355_multi_render(rt355_multi* m, const rt355_camera_desc* cam, int max_depth, float* rgb_out, rt355_stats* stats) {
  return rt355_multi_render_batch(m, cam, 1, max_depth, rgb_out, stats);
}

int rt355_multi_render_batch(rt355_multi* m, const rt355_camera_desc* cam, int n_frames, int max_depth, float* rgb_out, rt355_stats* stats) {
  if (!m || !cam || !rgb_out) return fail(RT355_EINVAL, "null argument");
  if (n_frames < 1 || n_frames > RT355_MAX_BATCH) return fail(RT355_EINVAL, "batch: between 1 and RT355_MAX_BATCH frames");
  if (!m->scene[0]) return fail(RT355_EINVAL, "no scene uploaded");
  const int64_t tpr = rt355_tiles_per_rank(cam, m->n);
  if (tpr < 0) return fail(RT355_EINVAL, "camera: hsize and vsize must be finite and in [1, 65536]");
  // per device: the tiles of all the frames, frame after frame; on device 0: the same for every device, then the canvases
  const size_t count = (size_t)tpr * rt355::TILE_PIX * 3 * (size_t)n_frames;
  const size_t canvas_floats = (size_t)(int)cam->hsize * (size_t)(int)cam->vsize * 3 * (size_t)n_frames;
  auto t0 = std::chrono::steady_clock::now();
  if (count > m->tiles_floats || canvas_floats > m->canvas_floats) {  // first frame of a larger size
    // the recorded sizes describe the buffers only once every allocation has succeeded: a failure below leaves them
    // at zero, so the next call allocates again instead of rendering into a freed or null buffer
    m->tiles_floats = 0, m->canvas_floats = 0;
    for (int d = 0; d < m->n; ++d) {
      MHIP(hipSetDevice(m->dev[d]));
      if (d > 0) {
        if (m->tiles[d]) (void)hipFree(m->tiles[d]);
        m->tiles[d] = nullptr;
        MHIP(hipMalloc((void**)&m->tiles[d], count * sizeof(float)));
      } else {
        if (m->gathered) (void)hipFree(m->gathered);
        if (m->canvas) (void)hipFree(m->canvas);
        m->gathered = m->canvas = nullptr;
        MHIP(hipMalloc((void**)&m->gathered, count * m->n * sizeof(float)));
        MHIP(hipMalloc((void**)&m->canvas, canvas_floats * sizeof(float)));
      }
    }
    m->tiles_floats = count, m->canvas_floats = canvas_floats;
  }
  // every device renders its tiles; device 0 straight into its slot of the gather buffer
  std::vector<int> rcs(m->n, 0);
  std::vector<std::string> errs(m->n);
  std::vector<rt355_stats> st(m->n);
  {
    std::vector<std::thread> th;
    for (int d = 0; d < m->n; ++d)
      th.emplace_back([&, d]() {
        float* dst = d == 0 ? m->gathered : m->tiles[d];
        rcs[d] = rt355_render_tiles_batch_begin(m->ctx[d], m->scene[d], cam, n_frames, max_depth, d, m->n, dst);
        if (!rcs[d]) rcs[d] = rt355_frame_end(m->ctx[d], &st[d]);
        if (rcs[d]) errs[d] = rt355_last_error();
      });
    for (auto& t : th) t.join();
  }
  for (int d = 0; d < m->n; ++d)
    if (rcs[d]) return fail(rcs[d], errs[d]);
  // gather to device 0
  if (m->n > 1) {
    if (!m->comm.empty()) {
      // Nothing returns between GroupStart and GroupEnd: a group left open would poison every later collective on
      // these communicators.  A failure inside the group is recorded and reported behind GroupEnd.
      ncclResult_t r = g_rccl.GroupStart();
      hipError_t he = hipSuccess;
      if (r == ncclSuccess) {
        for (int d = 1; d < m->n && r == ncclSuccess && he == hipSuccess; ++d) {
          he = hipSetDevice(m->dev[d]);
          if (he == hipSuccess) r = g_rccl.Send(m->tiles[d], count, ncclFloat, 0, m->comm[d], m->stream[d]);
        }
        if (he == hipSuccess) he = hipSetDevice(m->dev[0]);
        for (int d = 1; d < m->n && r == ncclSuccess && he == hipSuccess; ++d)
          r = g_rccl.Recv(m->gathered + (size_t)d * count, count, ncclFloat, d, m->comm[0], m->stream[0]);
        const ncclResult_t r2 = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
      }
      if (he != hipSuccess) return fail(RT355_EHIP, std::string("RCCL gather: hipSetDevice: ") + hipGetErrorString(he));
      if (r != ncclSuccess) return fail(RT355_EHIP, std::string("RCCL gather: ") + g_rccl.GetErrorString(r));
      for (int d = 0; d < m->n; ++d) {
        MHIP(hipSetDevice(m->dev[d]));
        MHIP(hipStreamSynchronize(m->stream[d]));
      }
    } else {
      MHIP(hipSetDevice(m->dev[0]));
      for (int d = 1; d < m->n; ++d) {
        if (m->dev[d] == m->dev[0])
          MHIP(hipMemcpyAsync(m->gathered + (size_t)d * count, m->tiles[d], count * sizeof(float), hipMemcpyDeviceToDevice, m->stream[0]));
        else
          MHIP(hipMemcpyPeerAsync(m->gathered + (size_t)d * count, m->dev[0], m->tiles[d], m->dev[d], count * sizeof(float), m->stream[0]));
      }
      MHIP(hipStreamSynchronize(m->stream[0]));
    }
  }
  int rc = rt355_untile_batch(m->ctx[0], cam, m->n, n_frames, m->gathered, m->canvas);
  if (rc) return rc;
  MHIP(hipSetDevice(m->dev[0]));
  MHIP(hipMemcpy(rgb_out, m->canvas, canvas_floats * sizeof(float), hipMemcpyDeviceToHost));
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    for (int d = 0; d < m->n; ++d) {
      for (int g = 0; g < RT355_MAX_GEN; ++g) stats->radiance_rays[g] += st[d].radiance_rays[g], stats->hits[g] += st[d].hits[g];
      stats->total_radiance += st[d].total_radiance, stats->total_shadow += st[d].total_shadow;
    }
    stats->frame_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  return RT355_OK;
}
}
