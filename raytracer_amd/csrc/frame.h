// frame.h -- what the translation units of the library share: error reporting, the context and scene records behind the
// opaque handles of include/rt355.h, and run_generations(), the launch chain of one frame.  run_generations is a
// template over the arithmetic type and the kernel mode; every (type, mode) pair instantiates its own set of traversal
// kernels, so each pair is compiled in a translation unit of its own (frame_inst.hip) and rt355.hip only declares them.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.h"

namespace rt355 {

extern thread_local std::string g_err;  // text behind rt355_last_error(), per host thread

inline int set_err(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return set_err(RT355_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));               \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return RT355_OK;
    if (p) (void)hipFree(p);
    p = nullptr, bytes = 0;
    size_t want = need + need / 8;
    if (hipMalloc(&p, want) != hipSuccess) {
      p = nullptr;
      return set_err(RT355_ENOMEM, "hipMalloc of " + std::to_string(want) + " bytes failed");
    }
    bytes = want;
    return RT355_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr, bytes = 0;
  }
};

}  // namespace rt355

using namespace rt355;  // this header is internal to the library's own translation units

constexpr int64_t PACKET_SECONDARY_PRIMS = 64;  // scenes with at most this many reflective / transparent prims trace every generation in packets

struct LaunchLog {
  int n = 0;
  int depth = 0;  // deepest generation
  bool shaded_in_trace = true;  // its trace launch also shades (k_trace<..., LAST>); false: a shadow launch of its own
  int kind[RT355_MAX_LAUNCH], gen[RT355_MAX_LAUNCH];
  bool merged[RT355_MAX_GEN + 1] = {};  // the generation's shadow launch also wrote its reflected rays (k_shadow<..., SPAWN>)
};

struct rt355_ctx {
  int device = 0;
  int real_bits = 64;
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;  // shadow launches, beside the trace / spawn chain
  hipEvent_t ev_traced[RT355_MAX_GEN] = {}, ev_side_done = nullptr;
  double pool_factor = 8.0;
  int counting = 0;  // 0 off, 1 the frame's own kernels instrumented, 2 per ray (every generation one ray per lane)
  bool timing = false;
  int exact_cones = 0;
  // Generations 0..packet_gens-1 use the wave-packet traversal, the later ones one lane per ray.  -1: chosen per scene --
  // every generation when few prims can spawn secondary rays (their reflections and refractions leave a handful of
  // surfaces and stay coherent: 5 to 7 % more frames per second on the teapot and glass scenes), generation 0 only
  // otherwise (100 000 small reflective shapes scatter the rays: packets are 11 to 42 % slower there).
  int packet_gens = -1;
  bool throughput_chain = false;  // the chain being queued was begun without waiting, or carries several frames
  int combine_levels = 4;  // generations folded into the first combine launch by walking ray subtrees (0..4); 4 = the whole ray tree of a
                           // depth-5 frame in ONE launch from generation 0: every hit record is read once and no colour is written back
  int shadow_lag = 0;   // the shadow launch of generation g is released behind the spawn of generation g + lag
  int merge_spawn = -1;  // the shadow launch of a generation also writes its reflected rays, k_spawn only the refracted ones: 1 always, 0 never,
                         // -1 for chains that overlap with others (one read of every hit for both, fewer launches; a frame rendered alone
                         // keeps the shadow rays off its critical path instead)
  int use_side = 1;     // shadow launches on the side stream (0: everything on the main stream, e.g. with several frames in flight)
  int quad = 0;         // generations beyond packet_gens, four lanes to a ray instead of one: bit 0 k_trace_q, bit 1 k_shadow_q
  uint32_t grid0_cap = 0;  // blocks of a generation-0 launch at most (set to 32 per CU at creation; 0: one block per 256 rays).  A capped grid
                           // walks its rays in several trips, which spreads each wave's start-up (some 150 instructions) over more rays
  DevBuf org, dir, hit_t, hit_uv, hit_prim, color, mixr, child, blk_hits, blk_children, blk_base, blk_top, ctl, io_in, io_out, io_out2;
  hipEvent_t ev[RT355_MAX_LAUNCH + 1] = {};  // timing mode: one before every launch, one behind the last
  FrameCtl* ctl_host = nullptr;  // pinned
  void* views_dev = nullptr;   // the views of a batch of frames (RT355_MAX_BATCH x DevView<double>), read by generation 0
  void* views_host = nullptr;  // pinned staging of the same
  void* fixed_scratch = nullptr;  // this device's copy of the fixed workspace (rt355_render_device_noalloc)
  std::string fixed_scratch_error;  // why it is null, if it is
  // a frame begun with rt355_render_tiles_begin and not yet ended: its launches are queued, nothing was waited for
  bool pending = false;
  LaunchLog pending_log;
  std::chrono::steady_clock::time_point pending_t0;
  double pending_primary_flops = 0;
  int n_cu = 256;
  hipEvent_t ev_order = nullptr;  // rt355_ctx_wait_stream / rt355_stream_wait_ctx
  bool async_helpers = false;     // rt355_untile* / rt355_to_u8_device only queue their launch
};

struct rt355_scene {
  int device = 0;  // the scene's tables live on this device; any context of the device may render it
  int real_bits = 64;
  DevBuf nodes, prim_info, xform, tri, trin, materials, inst_of;
  int32_t n_nodes = 0, n_bounded = 0, n_top = 0, n_prims = 0, n_materials = 0, max_depth = 0, n_instances = 0;
  bool any_secondary = false, any_transparent = false;
  int64_t n_secondary_prims = 0;  // prims whose material is reflective or transparent
  float light_pos[3], light_int[3];
  float box_abs_max = 0.f;
  size_t device_bytes = 0;
};

namespace rt355 {

template <class R>
DevScene<R> dev_scene(const rt355_scene* s) {
  DevScene<R> d;
  d.nodes = (const BvhNode4*)s->nodes.p;
  d.prim_info = (const PrimInfo*)s->prim_info.p;
  d.xform = (const DevXform<R>*)s->xform.p;
  d.tri = (const DevTri*)s->tri.p;
  d.trin = (const DevTriNormals<R>*)s->trin.p;
  d.materials = (const DevMaterial<R>*)s->materials.p;
  d.inst_of = (const int32_t*)s->inst_of.p;
  d.n_nodes = s->n_nodes, d.n_bounded = s->n_bounded, d.n_top = s->n_top, d.n_prims = s->n_prims, d.n_materials = s->n_materials;
  d.box_abs_max = s->box_abs_max;
  for (int k = 0; k < 3; ++k) d.light_pos[k] = R(s->light_pos[k]), d.light_int[k] = R(s->light_int[k]);
  return d;
}

// blocks of a k_scan launch over a generation of at most `rays` rays
inline uint32_t scan_blocks(uint64_t rays) { return std::max<uint32_t>(1u, (uint32_t)(((rays + GROUP - 1) / GROUP + SCAN_CHUNK - 1) / SCAN_CHUNK)); }

// The frame.  Generation 0 (ids 0..n_primary) is either made from the camera by the first trace launch
// (`primary`) or already in the pool.  Leaves the combined
// generation-0 colours in pool.color[0..n_primary).
// MODE: M_COUNT (instrumented kernels) | M_INST (the scene holds instanced meshes)
template <class R, int MODE>
int run_generations(rt355_ctx* c, const rt355_scene* s, RayPool<R>& pool, FrameCtl* ctl, uint32_t n_primary, int max_depth, const DevCamera<R>& cam, bool primary,
                    int out_mode, float* out_dev, LaunchLog& log) {
  DevScene<R> sc = dev_scene<R>(s);
  // Packets do more arithmetic per ray and stall less: best for the rate of chains that overlap (begun, not waited
  // for) or carry several frames; one frame alone, whose every kernel ends on its slowest wave, is done soonest with
  // packets for generations 0 and 1 only.
  // Scenes with instanced meshes walk one lane per ray throughout: a packet enters a mesh when ANY of its rays touches
  // the mesh's (loose, world-space) box and then drags all 64 through the mesh's tree; measured on 16 instanced
  // teapots of 6 320 triangles, packets are 14 % (7680x4320) to 27 % (1920x1080) slower than one lane per ray.
  const int packet_gens = c->counting == 2 ? 0
                          : c->packet_gens >= 0 ? c->packet_gens
                          : (MODE & M_INST) ? 0
                          : s->n_secondary_prims > PACKET_SECONDARY_PRIMS ? 1
                          : c->throughput_chain ? (int)RT355_MAX_GEN : 2;
  hipLaunchKernelGGL(k_reset_ctl, dim3(1), dim3(256), 0, c->stream, ctl, n_primary);
  const int depth = s->any_secondary ? max_depth : 0;  // no reflective / transparent material: one generation
  const int block = BLOCK;
  uint32_t grid0 = (n_primary + BLOCK - 1) / BLOCK;  // one block per 256 rays
  if (c->grid0_cap) grid0 = std::min(grid0, c->grid0_cap);
  const uint32_t grid_cap = (uint32_t)c->n_cu * 16 * (256 / BLOCK);
  auto rec = [&](int kind, int gen) {
    if (c->timing && log.n < RT355_MAX_LAUNCH) HIP_TRY(hipEventRecord(c->ev[log.n], c->stream));
    if (log.n < RT355_MAX_LAUNCH) log.kind[log.n] = kind, log.gen[log.n] = gen;
    return RT355_OK;
  };
  // Shadow rays feed nothing but k_combine, so they leave the critical path: with per-launch timing or
  // counting off they run on the side stream.
  const bool merge_all = c->merge_spawn >= 0 ? c->merge_spawn != 0 : c->throughput_chain;
  const bool overlap = !c->timing && !(MODE & M_COUNT) && c->use_side && !merge_all;
  for (int g = 0; g <= depth; ++g) {
    // generation g > 0 can hold at most twice the hits of g-1; its size is only known on the device
    const uint32_t grid = g == 0 ? grid0 : std::min<uint32_t>(grid_cap, (uint32_t)((pool.capacity + BLOCK - 1) / BLOCK));
    int rc;
    if ((rc = rec(RT355_K_TRACE, g))) return rc;
    const bool packet = g < packet_gens;
    const bool quad = !packet && (c->quad & 1) && !(MODE & M_INST);  // the quad traversal does not descend into instanced meshes
    const bool last = g == depth;
    const int prim = (g == 0 && primary) ? 1 : 0, any_t = s->any_transparent ? 1 : 0;
    // with the side stream in use, the event it waits for rides on the completion signal of the generation's
    // last launch on the main stream: a separate event record would put a marker packet, and some
    // microseconds, between that launch and the next trace
    hipEvent_t done = (overlap && !last) ? c->ev_traced[g] : nullptr;
    if (quad)  // the deepest generation is shaded by a k_shadow_q launch of its own, below
      hipLaunchKernelGGL((k_trace_q<R, MODE & M_COUNT>), dim3(grid), dim3(block), 0, c->stream, sc, pool, ctl, g, depth - g, log.n, cam, prim);
    else if (packet && last)
      hipLaunchKernelGGL((k_trace<R, MODE, true, true>), dim3(grid), dim3(block), 0, c->stream, sc, pool, ctl, g, 0, log.n, cam, prim);
    else if (packet)
      hipLaunchKernelGGL((k_trace<R, MODE, true, false>), dim3(grid), dim3(block), 0, c->stream, sc, pool, ctl, g, depth - g, log.n, cam, prim);
    else if (last)
      hipLaunchKernelGGL((k_trace<R, MODE, false, true>), dim3(grid), dim3(block), 0, c->stream, sc, pool, ctl, g, 0, log.n, cam, prim);
    else
      hipLaunchKernelGGL((k_trace<R, MODE, false, false>), dim3(grid), dim3(block), 0, c->stream, sc, pool, ctl, g, depth - g, log.n, cam, prim);
    log.n++;
    if ((rc = rec(RT355_K_SCAN, g))) return rc;
    hipLaunchKernelGGL(k_scan, dim3(std::min<uint32_t>(64u, scan_blocks(std::min<uint64_t>((uint64_t)pool.capacity, (uint64_t)n_primary << g)))), dim3(1024), 0, c->stream, pool.blk_hits, pool.blk_children, pool.blk_base,
                       pool.blk_top, ctl, g, (long long)pool.capacity);
    log.n++;
    // merged plan: one launch shades the generation's hits and writes their reflected rays; k_spawn follows for the
    // transparent hits only, when the scene has any
    const bool merged = merge_all && !last && !(!packet && (c->quad & 2) && !(MODE & M_INST));
    log.merged[g] = merged;
    if (merged) {
      if ((rc = rec(RT355_K_SHADE, g))) return rc;
      if (packet)
        hipLaunchKernelGGL((k_shadow<R, MODE, true, true>), dim3(grid), dim3(block), 0, c->stream, sc, pool, ctl, g, log.n);
      else
        hipLaunchKernelGGL((k_shadow<R, MODE, false, true>), dim3(grid), dim3(block), 0, c->stream, sc, pool, ctl, g, log.n);
      log.n++;
      if (any_t) {
        if ((rc = rec(RT355_K_SPAWN, g))) return rc;
        if (packet)
          hipLaunchKernelGGL((k_spawn<R, MODE, true, true>), dim3(grid), dim3(block), 0, c->stream, sc, pool, ctl, g, any_t, log.n);
        else
          hipLaunchKernelGGL((k_spawn<R, MODE, false, true>), dim3(grid), dim3(block), 0, c->stream, sc, pool, ctl, g, any_t, log.n);
        log.n++;
      }
    } else if (g < depth) {  // the deepest generation spawns nothing
      if ((rc = rec(RT355_K_SPAWN, g))) return rc;
      const int launch_id = log.n;
      if (packet)
        hipExtLaunchKernelGGL((k_spawn<R, MODE, true, false>), dim3(grid), dim3(block), 0, c->stream, nullptr, done, 0, sc, pool, ctl, g, any_t, launch_id);
      else
        hipExtLaunchKernelGGL((k_spawn<R, MODE, false, false>), dim3(grid), dim3(block), 0, c->stream, nullptr, done, 0, sc, pool, ctl, g, any_t, launch_id);
      log.n++;
    }
    // Shadow launches.  The one of generation g - lag is released here, behind the spawn of generation g:
    // the next trace then reaches the device first and the shadow waves fill the SIMDs its tail leaves
    // idle, instead of taking the block slots the chain is waiting for.
    auto shadow = [&](int sg, hipStream_t st) {
      const uint32_t sgrid = sg == 0 ? grid0 : std::min<uint32_t>(grid_cap, (uint32_t)((pool.capacity + BLOCK - 1) / BLOCK));
      int rc2;
      if ((rc2 = rec(RT355_K_SHADE, sg))) return rc2;
      if (sg < packet_gens)
        hipLaunchKernelGGL((k_shadow<R, MODE, true, false>), dim3(sgrid), dim3(block), 0, st, sc, pool, ctl, sg, log.n);
      else if ((c->quad & 2) && !(MODE & M_INST))
        hipLaunchKernelGGL((k_shadow_q<R, MODE & M_COUNT>), dim3(sgrid), dim3(block), 0, st, sc, pool, ctl, sg, log.n);
      else
        hipLaunchKernelGGL((k_shadow<R, MODE, false, false>), dim3(sgrid), dim3(block), 0, st, sc, pool, ctl, sg, log.n);
      log.n++;
      return (int)RT355_OK;
    };
    const bool shaded_in_trace = last && !quad;  // k_trace<..., LAST> shades its own hits
    if (!overlap) {
      if (!shaded_in_trace && !merged && (rc = shadow(g, c->stream))) return rc;
      continue;
    }
    const int lag = c->shadow_lag;
    if (!last) {
      HIP_TRY(hipStreamWaitEvent(c->side, c->ev_traced[g], 0));
      if (g - lag >= 0 && (rc = shadow(g - lag, c->side))) return rc;
    } else {
      // every generation still waiting: their traces finished long ago
      for (int sg = std::max(0, g - lag); sg < g; ++sg)
        if ((rc = shadow(sg, c->side))) return rc;
      if (!shaded_in_trace && (rc = shadow(g, c->stream))) return rc;  // behind its own trace, on the same stream
    }
  }
  log.depth = depth;
  log.shaded_in_trace = !(depth >= packet_gens && (c->quad & 1) && !(MODE & M_INST));
  if (overlap && depth > 0) {
    HIP_TRY(hipEventRecord(c->ev_side_done, c->side));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_side_done, 0));
  }
  // The deepest generation's colours are final already; generation 0 still has to reach the canvas.
  // One launch folds up to four generations below its own: each ray of generation `top` walks its own subtree (at most
  // 2^levels - 1 hits below it).  With max_depth 5 that is the whole frame from generation 0: every hit record is read
  // once and no combined colour goes back to memory (0.70 ms per chain of 16 headline frames against 1.09 with the
  // walk limited to the two sparsest generations and a launch each for generations 1 and 0).
  const int levels = std::max(0, std::min(c->combine_levels, depth - 1));
  const int top = depth > 0 ? depth - 1 - levels : 0;
  for (int g = top; g >= 0; --g) {
    // no traversal here: 256-thread blocks
    const uint32_t grid = std::max<uint32_t>(1u, (g == 0 ? grid0 : std::min<uint32_t>(grid_cap, (uint32_t)((pool.capacity + BLOCK - 1) / BLOCK))) / (256 / BLOCK));
    int rc;
    if ((rc = rec(RT355_K_COMBINE, g))) return rc;
    const int has_children = g < depth ? 1 : 0;
    switch (g == top ? levels : 0) {
      case 4: hipLaunchKernelGGL((k_combine<R, 4>), dim3(grid), dim3(256), 0, c->stream, sc, pool, ctl, g, has_children, cam, out_mode, out_dev); break;
      case 3: hipLaunchKernelGGL((k_combine<R, 3>), dim3(grid), dim3(256), 0, c->stream, sc, pool, ctl, g, has_children, cam, out_mode, out_dev); break;
      case 2: hipLaunchKernelGGL((k_combine<R, 2>), dim3(grid), dim3(256), 0, c->stream, sc, pool, ctl, g, has_children, cam, out_mode, out_dev); break;
      case 1: hipLaunchKernelGGL((k_combine<R, 1>), dim3(grid), dim3(256), 0, c->stream, sc, pool, ctl, g, has_children, cam, out_mode, out_dev); break;
      default: hipLaunchKernelGGL((k_combine<R, 0>), dim3(grid), dim3(256), 0, c->stream, sc, pool, ctl, g, has_children, cam, out_mode, out_dev); break;
    }
    log.n++;
  }
  if (c->timing && log.n <= RT355_MAX_LAUNCH) HIP_TRY(hipEventRecord(c->ev[log.n], c->stream));
  // counters and the overflow flag travel to pinned host memory behind the last kernel: the caller's one
  // stream synchronisation makes them readable
  HIP_TRY(hipMemcpyAsync(c->ctl_host, ctl, sizeof(FrameCtl), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipGetLastError());
  return RT355_OK;
}
}  // namespace rt355
