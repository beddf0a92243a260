/*
 * rt355.h -- C ABI of the MI355X (gfx950) render backend.
 *
 * This is the drop-in boundary for ONE call of the reference ray tracer:
 *
 *     impl Camera { pub fn render(&self, world: &World) -> Canvas }   (src/camera.rs:74-88)
 *
 * and everything it calls (World::color_at src/world.rs:58-66 and below).  The reference has
 * no FFI today; a Rust `-sys` crate binds exactly the functions declared here (see
 * rust/rt355-sys/src/lib.rs).  Plain C: no exceptions cross this boundary, every function returns 0
 * or a negative RT355_E* code, rt355_last_error() gives a thread-local message.
 *
 * All scalar inputs are f32 because every scalar in the reference is f32 (src/epsilon.rs:1,
 * src/camera.rs:12-18).  The arithmetic type of the device kernels is chosen per context:
 * real_bits = 64 (double) or 32 (float, the reference's own precision).
 */
#ifndef RT355_H
#define RT355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT355_VERSION 1

/* error codes */
#define RT355_OK 0
#define RT355_EINVAL (-1)    /* bad argument, singular / non-affine transform, NaN in the scene   */
#define RT355_EHIP (-2)      /* a HIP runtime call failed                                          */
#define RT355_ECAPACITY (-3) /* secondary-ray pool exhausted (raise it with rt355_ctx_set_pool)    */
#define RT355_ENOMEM (-4)

/* Shape kinds, in the order of `enum Shape` (src/shape.rs:11-20). */
enum rt355_kind {
  RT355_SPHERE = 0,
  RT355_PLANE = 1,
  RT355_CUBE = 2,
  RT355_CYLINDER = 3,
  RT355_CONE = 4,
  RT355_GROUP = 5, /* inert inside a World (src/shape.rs:376, :134, :204): never hit */
  RT355_TRIANGLE = 6,
  RT355_SMOOTH_TRIANGLE = 7
};

/* Pattern kinds, in the order of `enum Pattern` (src/pattern.rs:6-12); 0 = material.pattern None. */
enum rt355_pattern {
  RT355_PATTERN_NONE = 0,
  RT355_PATTERN_STRIPE = 1,
  RT355_PATTERN_GRADIENT = 2,
  RT355_PATTERN_RING = 3,
  RT355_PATTERN_CHECKER = 4,
  RT355_PATTERN_TEST = 5
};

/* `struct Material` (src/material.rs:6-17).  The pattern's own transformation is stored by the
 * reference but never read (src/pattern.rs:16,94), so it is not part of the ABI. */
typedef struct rt355_material {
  float color[3];
  float ambient, diffuse, specular, shininess;
  float reflective, refractive, transparency;
  int32_t pattern_kind;
  float c1[3], c2[3];
} rt355_material;

/* One entry of `World.shapes` (src/world.rs:17-20).  The array index IS the shape id used for
 * nearest-hit ties (lowest index wins: stable sort src/world.rs:42 + first-min
 * src/intersection.rs:151) and reported by rt355_trace_sample. */
typedef struct rt355_shape_desc {
  int32_t kind;      /* enum rt355_kind */
  float m[16];       /* `transformation`, column-major as nalgebra stores Matrix4<f32> */
  rt355_material mat;
  float min, max;    /* cylinder / cone extents (src/cylinder.rs:11-13); +-INFINITY allowed */
  int32_t closed;
  float p1[3], p2[3], p3[3]; /* triangle / smooth triangle vertices (src/triangle.rs:10-12) */
  float n1[3], n2[3], n3[3]; /* smooth triangle vertex normals (src/smooth_triangle.rs:16-18) */
} rt355_shape_desc;

/* `struct PointLight` (src/pointlight.rs:7-10). */
typedef struct rt355_light_desc {
  float position[3];
  float intensity[3];
} rt355_light_desc;

/* The three public arguments of Camera::new plus its transform (src/camera.rs:23-28); half_width,
 * half_height and pixel_size are re-derived inside the library exactly as src/camera.rs:29-41. */
typedef struct rt355_camera_desc {
  float hsize, vsize, fov;
  float transform[16]; /* column-major */
} rt355_camera_desc;

#define RT355_MAX_GEN 8     /* generations 0..max_depth; max_depth <= 7 */
#define RT355_MAX_LAUNCH 48 /* per-frame kernel launches recorded in rt355_stats */

/* Kernel classes recorded per launch in rt355_stats.launch_kind. */
enum rt355_launch_kind {
  RT355_K_TRACE = 1,     /* nearest-hit traversal of one generation; the deepest one also shades its hits */
  RT355_K_SHADE = 2,     /* shadow rays of one generation (is_shadowed)                              */
  RT355_K_COMBINE = 3,   /* Phong lighting + child colours folded into their parent; generation 0
                            writes the canvas */
  RT355_K_SPAWN = 5,     /* n1 / n2, reflect / refract / schlick: writes the rays of the next generation */
  RT355_K_SCAN = 6       /* prefix sums of the child-ray slots of one generation */
};

/* Per-frame statistics.  Ray counts are always filled.  Work counters (box_tests ... flops) are
 * filled only when the context is in counting mode (rt355_ctx_set_counting), which runs
 * instrumented kernels and is not meant to be timed.  Launch times are filled when timing mode
 * is on (rt355_ctx_set_timing): HIP events around every launch, with all launches on ONE stream.
 * Outside these two modes the shadow launches run on a second stream beside the others, so a
 * frame takes less than the sum of its launch times. */
typedef struct rt355_stats {
  uint64_t radiance_rays[RT355_MAX_GEN]; /* one per color_at call (src/world.rs:58)      */
  uint64_t hits[RT355_MAX_GEN];          /* = shadow rays: one per is_shadowed call (:105) */
  uint64_t total_radiance, total_shadow;
  double frame_ms;                       /* host wall time of the call, device-synchronised */
  int32_t n_launch;
  int32_t launch_kind[RT355_MAX_LAUNCH];
  int32_t launch_gen[RT355_MAX_LAUNCH];
  float launch_ms[RT355_MAX_LAUNCH];
  /* counting mode, per launch */
  uint64_t launch_box_tests[RT355_MAX_LAUNCH];  /* child-box slab tests               */
  uint64_t launch_prim_tests[RT355_MAX_LAUNCH]; /* local_intersect calls, all kinds    */
  uint64_t launch_max_box_tests[RT355_MAX_LAUNCH]; /* most box tests done by one lane */
  uint64_t launch_max_wave_clocks[RT355_MAX_LAUNCH]; /* k_trace: longest traversal of one wave, shader clocks */
  double launch_flops[RT355_MAX_LAUNCH];        /* algorithmic FLOPs, see rt355_flop_table */
} rt355_stats;

typedef struct rt355_ctx rt355_ctx;     /* one device, one stream, scratch pools */
typedef struct rt355_scene rt355_scene; /* device-resident shape tables + BVH    */

const char* rt355_last_error(void);
int rt355_version(void);

/* One context per (process, GPU).  real_bits: 64 or 32. */
int rt355_ctx_create(rt355_ctx** out, int device_id, int real_bits);
/* The same with flags.  RT355_CTX_ONE_STREAM: a context for hosts that keep several frames in flight on one
 * GPU, one context per frame (rt355_render_tiles_begin / rt355_render_device_begin / rt355_frame_end): all its
 * launches go to one stream of default priority instead of a high-priority chain plus a low-priority stream
 * for the shadow launches.  A single frame takes longer in such a context (1.18 ms against 0.86 ms for the
 * headline frame), three in flight take less per frame (0.72 ms). */
#define RT355_CTX_ONE_STREAM 1
int rt355_ctx_create_ex(rt355_ctx** out, int device_id, int real_bits, int flags);
void rt355_ctx_destroy(rt355_ctx*);
/* Ray pool capacity in rays (all generations) per primary ray (default 8.0). */
int rt355_ctx_set_pool(rt355_ctx*, double rays_per_pixel);
/* A truncated cone hit by a ray that is parallel to one of its halves (|a| <= EPSILON at
 * src/shape.rs:295-298) is reported by the reference WITHOUT the min < y < max test, i.e. anywhere on the
 * infinite double cone.  Such hits have no finite bound.  mode 1: every truncated cone is tested by
 * every ray (exact, slow for many cones).  mode 0 (default): exact while the scene holds at most 32
 * truncated cones; beyond that cones live in the BVH and a hit of this kind is found only if the ray
 * also passes the cone's bounding box.  Applies to scenes uploaded afterwards. */
int rt355_ctx_set_exact_cones(rt355_ctx*, int mode);
/* Counting mode.  0: off.  1: the frame's own kernels, instrumented (what the launches really tested: in a wave-packet
 * traversal every live lane is charged for every node and triangle its wave visits).  2: PER RAY -- every generation is
 * traced one ray per lane, so a ray is charged only for the boxes and primitives of the nodes its own path through the
 * BVH reaches; the count a roofline numerator is made from, since it does not reward the union work of packets. */
int rt355_ctx_set_counting(rt355_ctx*, int mode);
int rt355_ctx_set_timing(rt355_ctx*, int on);

/* Replaces the per-ray work the reference redoes for every shape and every ray
 * (Ray::inv_transform src/ray.rs:32-37, Shape::normal src/shape.rs:111-136): inverses, triangle
 * edges and flat normals (src/triangle.rs:65-66,83-85) and the BVH are built here, ONCE.
 * Group entries are kept as index placeholders and never hit. */
int rt355_scene_upload(rt355_ctx*, const rt355_shape_desc* shapes, size_t n_shapes,
                       const rt355_light_desc* light, rt355_scene** out);
void rt355_scene_destroy(rt355_scene*);
/* BVH facts for reports: out[0]=nodes, out[1]=bounded prims, out[2]=unbounded prims,
 * out[3]=max depth, out[4]=device bytes. */
int rt355_scene_info(const rt355_scene*, uint64_t out[5]);

/* Camera::render (src/camera.rs:74-88).  rgb_out: hsize*vsize*3 floats, caller-owned HOST memory,
 * row-major from the top-left, unclamped linear RGB, like Canvas.pixels (src/canvas.rs:8-12).
 * max_depth is the constant 5 of src/camera.rs:79. */
int rt355_render(rt355_ctx*, const rt355_scene*, const rt355_camera_desc* cam, int max_depth,
                 float* rgb_out, rt355_stats* stats /* nullable */);

/* Camera::render followed by Color::to_u8 (src/color.rs:97-111) of every channel, on the device: rgb_out is
 * hsize*vsize*3 BYTES of host memory, the body of a binary PPM (the values src/ppm.rs:40-46 prints).  A
 * quarter of the bytes of rt355_render cross PCIe. */
int rt355_render_u8(rt355_ctx*, const rt355_scene*, const rt355_camera_desc* cam, int max_depth,
                    uint8_t* rgb_out, rt355_stats* stats /* nullable */);

/* Page-locked host memory for canvases (Canvas.pixels, src/canvas.rs:8-12): the copy at the end of
 * rt355_render / rt355_multi_render runs at full PCIe rate into such a buffer and about three times slower
 * into pageable memory.  NULL when the allocation fails. */
void* rt355_host_alloc(size_t bytes);
void rt355_host_free(void*);

/* Camera::render again, with the canvas left in DEVICE memory (`rgb_out_dev`: hsize*vsize*3 floats,
 * same layout as rt355_render): for callers that go on working on the GPU (tone mapping, encoding, a
 * single-GPU frame loop).  Synchronous with respect to the host. */
int rt355_render_device(rt355_ctx*, const rt355_scene*, const rt355_camera_desc* cam, int max_depth,
                        float* rgb_out_dev, rt355_stats* stats /* nullable */);

/* rt355_render_device for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc).  It has no counterpart in the reference.  It allocates nothing and is synchronous with respect to the
 * host: it waits for the frame and reads its statistics back, so it is not a call to make while a stream is being
 * captured into a HIP graph.  The frame's ray
 * pool and control block live in a fixed workspace of RT355_NOALLOC_BYTES inside the library (one copy per
 * GPU, held until the process ends) instead of the context's growable buffers: the call allocates and frees
 * nothing.  A frame that needs more -- (tiles x 1024 primary rays) x pool factor x ~122 bytes in f64, ~70 in
 * f32, i.e. about 320x200 at the default factor 8 in f64 -- is refused with RT355_ECAPACITY before any launch.
 * Every call on a GPU shares that GPU's workspace: calls that overlap on one GPU (two contexts, two threads)
 * race; the workspace keeps the last frame's contents, what must start at zero is cleared by the call itself.
 * Same output and statistics as rt355_render_device. */
#define RT355_NOALLOC_BYTES (64u << 20)
int rt355_render_device_noalloc(rt355_ctx*, const rt355_scene*, const rt355_camera_desc* cam, int max_depth,
                                float* rgb_out_dev, rt355_stats* stats /* nullable */);

/* Same frame, rank `rank` of `n_ranks` renders only its 32x32 tiles (tile t belongs to rank
 * t % n_ranks) into DEVICE memory `tiles_out`: rt355_tiles_per_rank() * 1024 * 3 floats, tile-major,
 * pixels of a tile row-major.  With n_ranks == 1 this is the whole frame in tile order.
 * The call is synchronous with respect to the host. */
int rt355_render_tiles(rt355_ctx*, const rt355_scene*, const rt355_camera_desc* cam, int max_depth,
                       int rank, int n_ranks, float* tiles_out_dev, rt355_stats* stats);
/* rt355_render_tiles in two halves, for hosts that keep several frames in flight on one GPU: _begin queues
 * the frame's launches on the context's streams and returns at once; rt355_frame_end waits for them and fills
 * the statistics (and reports RT355_ECAPACITY).  One frame at a time per context: a host that wants several
 * frames in flight uses several contexts (RT355_CTX_ONE_STREAM); an uploaded scene may be rendered by any
 * context of its device and precision.  The launch chain of one frame is a
 * sequence of dependent kernels that each end on their slowest wave; a second frame's kernels fill the
 * device while the first one's tails drain.  The reference renders one frame per call (src/camera.rs:74-88);
 * this is the same call, split. */
int rt355_render_tiles_begin(rt355_ctx*, const rt355_scene*, const rt355_camera_desc* cam, int max_depth,
                             int rank, int n_ranks, float* tiles_out_dev);
int rt355_render_device_begin(rt355_ctx*, const rt355_scene*, const rt355_camera_desc* cam, int max_depth,
                              float* rgb_out_dev); /* rt355_render_device, begun */
int rt355_frame_end(rt355_ctx*, rt355_stats* stats /* nullable */);
/* Several frames of ONE image size in one launch chain: Camera::render (src/camera.rs:74-88) called n_frames
 * times, e.g. along a camera path, with the rays of all the frames traced together.  `cams` points at
 * n_frames cameras (1 <= n_frames <= RT355_MAX_BATCH) that share hsize and vsize.  A rank of many renders
 * few tiles per frame, too few to fill a GPU whose every kernel ends on its slowest wave; the tiles of
 * n_frames frames fill it like one whole frame does.  tiles_out_dev: n_frames x rt355_tiles_per_rank() x 1024
 * x 3 floats, frame after frame, each frame laid out as rt355_render_tiles lays it out.  rgb_out_dev:
 * n_frames canvases of rt355_render_device, one after the other.  Ended by rt355_frame_end, whose statistics
 * are the sums over the frames.  Every frame's pixels are identical to those of a call of its own. */
#define RT355_MAX_BATCH 64
int rt355_render_tiles_batch_begin(rt355_ctx*, const rt355_scene*, const rt355_camera_desc* cams, int n_frames,
                                   int max_depth, int rank, int n_ranks, float* tiles_out_dev);
int rt355_render_device_batch_begin(rt355_ctx*, const rt355_scene*, const rt355_camera_desc* cams, int n_frames,
                                    int max_depth, float* rgb_out_dev);
/* The same chain, waited for, with the canvases copied to HOST memory: rgb_out receives n_frames canvases of
 * rt355_render, one after the other (Camera::render, src/camera.rs:74-88, called once per camera). */
int rt355_render_batch(rt355_ctx*, const rt355_scene*, const rt355_camera_desc* cams, int n_frames, int max_depth,
                       float* rgb_out, rt355_stats* stats /* nullable */);
/* Number of tiles each rank owns (tile count padded up to a multiple of n_ranks). */
int64_t rt355_tiles_per_rank(const rt355_camera_desc* cam, int n_ranks);
/* Rank 0, after the gather: `gathered_dev` holds n_ranks consecutive rank buffers; scatters them
 * to the row-major canvas `rgb_out_dev` (hsize*vsize*3 floats, device memory). */
int rt355_untile(rt355_ctx*, const rt355_camera_desc* cam, int n_ranks, const float* gathered_dev,
                 float* rgb_out_dev);

/* The same for tile buffers already taken through Color::to_u8 (rt355_to_u8_device on each rank's tiles): a
 * quarter of the bytes to gather when the frame is going to be a PPM (src/ppm.rs:40-46) anyway. */
int rt355_untile_u8(rt355_ctx*, const rt355_camera_desc* cam, int n_ranks, const uint8_t* gathered_dev,
                    uint8_t* rgb_out_dev);
/* The two above for the gathered rank buffers of rt355_render_tiles_batch_begin (each holds n_frames frames):
 * rgb_out_dev receives n_frames row-major canvases, one after the other. */
int rt355_untile_batch(rt355_ctx*, const rt355_camera_desc* cam, int n_ranks, int n_frames,
                       const float* gathered_dev, float* rgb_out_dev);
int rt355_untile_batch_u8(rt355_ctx*, const rt355_camera_desc* cam, int n_ranks, int n_frames,
                          const uint8_t* gathered_dev, uint8_t* rgb_out_dev);

/* Camera::render (src/camera.rs:74-88) on SEVERAL GPUs of one node from ONE host process: what a host that
 * is not split into one process per GPU calls.  rt355_multi_create makes one context per listed device
 * (real_bits as rt355_ctx_create); rt355_multi_upload replicates World { pointlight, shapes }
 * (src/world.rs:17-20) on every device; rt355_multi_render cuts the frame into 32x32 tiles, tile t to device
 * t % n_devices, renders them concurrently (one host thread per device), gathers the tile buffers to the
 * first device -- RCCL send / receive over xGMI when librccl can be loaded and the devices are distinct, peer
 * copies otherwise (rt355_multi_gather_kind tells: "rccl", "peer copies" or "none") -- un-tiles there and
 * copies the canvas to rgb_out (hsize*vsize*3 floats of HOST memory, layout of rt355_render).  stats, when
 * given, holds the ray counts summed over the devices and the wall time of the call.  The canvas is
 * bit-identical to rt355_render's for any number of devices. */
typedef struct rt355_multi rt355_multi;
int rt355_multi_create(rt355_multi** out, const int* device_ids, int n_devices, int real_bits);
void rt355_multi_destroy(rt355_multi*);
int rt355_multi_upload(rt355_multi*, const rt355_shape_desc* shapes, size_t n_shapes, const rt355_light_desc* light);
int rt355_multi_render(rt355_multi*, const rt355_camera_desc* cam, int max_depth, float* rgb_out,
                       rt355_stats* stats /* nullable */);
/* n_frames cameras of one image size in one launch chain per device (rt355_render_tiles_batch_begin on each): what
 * keeps 8 GPUs busy on frames of which each renders an eighth.  rgb_out: n_frames canvases, one after the other. */
int rt355_multi_render_batch(rt355_multi*, const rt355_camera_desc* cams, int n_frames, int max_depth, float* rgb_out,
                             rt355_stats* stats /* nullable */);
const char* rt355_multi_gather_kind(const rt355_multi*);

/* World::color_at(ray, max_depth) (src/world.rs:58-66) for caller-supplied rays: rays is n*6
 * floats (origin xyz, direction xyz; direction is used as given, not normalised), rgb_out n*3. */
int rt355_color_at(rt355_ctx*, const rt355_scene*, const float* rays, size_t n, int max_depth,
                   float* rgb_out, rt355_stats* stats);

/* Parity probe: World::intersect + Intersections::hit (src/world.rs:35-44,
 * src/intersection.rs:149-154) for n rays.  t_out[i] = nearest t >= 0 (NaN when there is no hit),
 * shape_index_out[i] = index into the uploaded shape array (-1 when there is no hit). */
int rt355_trace_sample(rt355_ctx*, const rt355_scene*, const float* rays /* n*6 */, size_t n,
                       double* t_out, int32_t* shape_index_out);

/* Parity probe: World::is_shadowed (src/world.rs:45-57) for n points (n*3 floats). */
int rt355_is_shadowed(rt355_ctx*, const rt355_scene*, const float* points, size_t n,
                      uint8_t* out);

/* Color::to_u8 (src/color.rs:97-111) on the device: n floats -> n bytes, both HOST memory. */
int rt355_to_u8(rt355_ctx*, const float* rgb, size_t n, uint8_t* out);
/* The same with both buffers in DEVICE memory: a canvas left in HBM by rt355_render_device / rt355_untile
 * becomes the bytes of a binary PPM body (src/ppm.rs:40-46 per pixel) without leaving the GPU, a quarter
 * of the bytes to copy out.  Synchronous with respect to the host. */
int rt355_to_u8_device(rt355_ctx*, const float* rgb_dev, size_t n, uint8_t* out_dev);

/* Ordering against the caller's own HIP streams, for frame loops that must not stall the host between a
 * collective and the launch that consumes it.  No counterpart in the reference (it has no device).
 * rt355_ctx_wait_stream: whatever the context launches from now on starts after the work queued on `hip_stream`
 * (a hipStream_t; NULL is the legacy default stream) so far -- e.g. the gather that fills the buffer rt355_untile
 * reads.  rt355_stream_wait_ctx: the other direction, work queued on `hip_stream` after the call starts once
 * everything the context has launched is done -- e.g. the consumer of an un-tiled canvas, or the send of converted
 * tiles.  Neither blocks the host.
 * rt355_ctx_set_async(ctx, 1): rt355_untile*, rt355_to_u8_device stop waiting for their own launch (they only queue
 * it); the caller orders with the two calls above or ends with rt355_ctx_sync.  0 (the default) restores the wait. */
int rt355_ctx_wait_stream(rt355_ctx*, void* hip_stream);
int rt355_stream_wait_ctx(rt355_ctx*, void* hip_stream);
int rt355_ctx_set_async(rt355_ctx*, int on);
int rt355_ctx_sync(rt355_ctx*);

/* Host only, needs no GPU: builds the scene tables and the BVH exactly as rt355_scene_upload does
 * (real_bits picks the box padding) and checks their structure: every bounded shape in exactly one
 * leaf, every child box enclosing what lies beneath it, the traversal stack bound.  out[0]=nodes,
 * out[1]=bounded prims, out[2]=unbounded prims, out[3]=stack entries a descent can need,
 * out[4]=distinct materials.  Returns RT355_EINVAL with a message when the input is rejected or the
 * check fails. */
int rt355_scene_check(const rt355_shape_desc* shapes, size_t n_shapes, int real_bits, int exact_cones,
                      uint64_t out[5]);

/* Algorithmic FLOP weights used for launch_flops: out[0]=primary ray, [1]=box test,
 * [2]=ray->object transform, [3..10]=local_intersect per rt355_kind, [11]=shading per hit,
 * [12]=reflect spawn, [13]=refract spawn, [14]=schlick. */
void rt355_flop_table(double out[16]);

#ifdef __cplusplus
}
#endif
#endif /* RT355_H */
