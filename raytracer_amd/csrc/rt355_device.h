// rt355_device.h -- data layout shared by the host scene builder, the C ABI and the HIP kernels.
//
// Everything the kernels read is laid out here once, at scene upload:
//   * prims: every non-Group shape, BVH leaf order first (bounded shapes), then the shapes that have no
//     finite bound (planes, uncapped cylinders / cones).  A "prim id" p indexes all per-prim arrays.
//   * nodes: 4-wide BVH, each node carries the boxes of ALL its children, so one fetch (one memory round
//     trip, which is what a traversal step costs) decides up to four descents.
//   * materials: deduplicated table (material.rs:6-17 + pattern.rs colours).
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/rt355.h"

namespace rt355 {

constexpr int TILE = 32;             // image tile edge; tile t belongs to rank t % n_ranks
constexpr int TILE_PIX = TILE * TILE;
constexpr int STACK_DEPTH = 64;      // traversal stack entries; the builder rejects trees that could need more
constexpr int LDS_STACK = 24;        // of which this many live in LDS (one column per lane); the rest in scratch
constexpr int BLOCK = 64;            // threads per block of the traversal kernels: one wave.  No kernel synchronises across waves, and a
                                     // block's wave slots return to the dispatcher only when its last wave ends
constexpr int GROUP = 64;            // rays per entry of the child-slot bookkeeping: the rays of one wave
// Waves per SIMD the per-lane kernels are compiled for (the register budget is 512 / this per lane).
#ifndef RT355_WPS_LANE
#define RT355_WPS_LANE 3
#endif
#ifndef RT355_WPS_SPAWN
#define RT355_WPS_SPAWN 3
#endif
// the same for the packet kernels (fewer live registers: node and triangle records sit in scalar registers)
#ifndef RT355_WPS_PACKET
#define RT355_WPS_PACKET 4
#endif
#ifndef RT355_WPS_COMBINE
#define RT355_WPS_COMBINE 4
#endif
// the packet shadow kernel gains from a fifth wave although 28 of its registers then live in scratch memory (generation 0
// of the headline: 1.33 -> 1.2 ms per chain of 16 frames; six waves are no better); the packet spawn kernels lose by it
#ifndef RT355_WPS_SHADOW_PACKET
#define RT355_WPS_SHADOW_PACKET 5
#endif
// the deepest generation's trace-and-shade kernel: 128 registers and ten spilled beat 140 and three waves (0.73 -> 0.66 ms)
#ifndef RT355_WPS_PACKET_LAST
#define RT355_WPS_PACKET_LAST 4
#endif
#ifndef RT355_WPS_SPAWN_PACKET
#define RT355_WPS_SPAWN_PACKET 4
#endif
constexpr uint32_t LEAF_FIRST_MASK = 0x1FFFFFFu;  // leaf reference ~c: first | tri-only flag << 25 | count << 26
constexpr uint32_t LEAF_TRI_BIT = 1u << 25;       // every prim of the leaf is a triangle with identity transform
constexpr int MAX_TREE_DEPTH = 40;   // of the binary tree the builder collapses; a 4-wide node pushes up to 3 entries
constexpr int BVH_WIDTH = 4;
constexpr int LEAF_MAX = 4;
constexpr int N_WORK_COUNTERS = 12;  // per launch: [0] box tests, [1] xforms, [2..10) prim tests per kind

// prim_info.y bits
constexpr int PF_KIND_MASK = 7;
constexpr int PF_CLOSED = 8;
constexpr int PF_IDENTITY = 16;  // transformation is exactly the identity: the ray is used as is
constexpr int PF_REFLECTIVE = 32;   // material.reflective != 0: a hit may spawn a reflected ray
constexpr int PF_TRANSPARENT = 64;  // material.transparency != 0: a hit may spawn a refracted ray
constexpr int PF_OTHERS_UNIT = 128;  // every OTHER prim of the scene has refractive index 1: n1 / n2 need less searching
constexpr int PF_SCHLICK = 256;      // reflective > 0 && transparency > 0: the two child colours are mixed by schlick()
constexpr int PF_INSTANCED = 512;    // a triangle of an instanced mesh: identity inside its mesh's own tree, taken to world space by
                                     // the transformation of its instance (inst_of[p] is the instance's prim)
constexpr int PF_ISOLATED = 1024;    // (transparent prims) no other prim can have an intersection inside this prim's bounding box: the n1 / n2
                                     // search between two of the prim's own intersections has nothing to find
constexpr int PF_MAT_SHIFT = 11;
constexpr int64_t MAX_MATERIALS = 1ll << (31 - PF_MAT_SHIFT);  // distinct materials of a scene (the id sits above the flags in 32 bits)
// Device-side prim kind of an instanced mesh: a run of triangles of World.shapes that share one transformation (what
// Scene::to_transformed_shapes leaves of a group with a mesh below it, src/group.rs:117-168) is kept in object space
// under a tree of its own; the top tree holds one prim for it.  Groups themselves never reach the device
// (src/shape.rs:376), so their kind number is free.
constexpr int RT355_INSTANCE = RT355_GROUP;

struct PrimInfo {
  int32_t shape;  // index in World.shapes
  int32_t flags;  // kind | PF_* | material << PF_MAT_SHIFT
};

// One node of the 4-wide BVH, 128 bytes.  Boxes are stored in float, rounded outward, whatever the
// arithmetic type of the intersection tests is.  Unused child slots hold a box whose two x planes are both
// +inf, which no ray overlaps within a finite interval, and the empty leaf reference -1.
struct alignas(16) BvhNode4 {
  float box[BVH_WIDTH][6];   // per child: lo.x, hi.x, lo.y, hi.y, lo.z, hi.z (the two planes of an axis side by side)
  int32_t c[BVH_WIDTH];      // >= 0: inner node index; < 0: leaf reference, see LEAF_*
  int32_t pad[4];
};
static_assert(sizeof(BvhNode4) == 128, "BvhNode4 layout");

template <class R>
struct DevMaterial {
  R color[3];
  R ambient, diffuse, specular, shininess, reflective, refractive, transparency;
  R c1[3], c2[3];
  int32_t pattern_kind;
  int32_t pad;
};

template <class R>
struct DevXform {  // rows 0..2 of the inverse transformation (affine): x' = r0 . (x,y,z,w)
  R r[12];
  R ymin, ymax;  // cylinder / cone extents
  R root;        // instance: root node of the mesh's tree (a whole number)
  R pad1;
};

// Triangle record: the three vertices as the f32 values the host handed over (src/triangle.rs:10-12),
// 48 bytes.  The edges e1 = p2 - p1, e2 = p3 - p1 (src/triangle.rs:83-84) are formed in the kernel, in R:
// in double the difference of two floats is exact, in float it is the reference's own subtraction.
struct alignas(16) DevTri {
  float p1[3], p2[3], p3[3];
  int32_t shape, flags;  // copy of prim_info: one fetch per triangle test
  int32_t pad;
};

template <class R>
struct DevTriNormals {  // smooth triangle: n1,n2,n3; flat triangle: nf (src/triangle.rs:85)
  R n1[3], n2[3], n3[3], nf[3];
};

template <class R>
struct DevScene {
  const BvhNode4* nodes;          // n_nodes (0 when there is no bounded prim)
  const PrimInfo* prim_info;      // n_prims
  const DevXform<R>* xform;       // n_prims
  const DevTri* tri;              // n_prims (valid for triangle kinds)
  const DevTriNormals<R>* trin;   // n_prims (valid for triangle kinds)
  const DevMaterial<R>* materials;
  // prims [0, n_bounded) are the leaves of the top tree, [n_bounded, n_top) the prims every ray tests, [n_top, n_prims)
  // the triangles of the instanced meshes
  int32_t n_nodes, n_bounded, n_top, n_materials;
  float box_abs_max;  // largest |coordinate| of any BVH box: scales the error bound of the slab test
  R light_pos[3], light_int[3];
  const int32_t* inst_of;         // n_prims when the scene has instances: the instance prim of a PF_INSTANCED triangle
  int32_t n_prims;
};

// What differs between the frames of one batch: the view.  (Image size, tiling and rank are shared.)
template <class R>
struct DevView {
  R inv[12];  // rows 0..2 of the inverse camera transformation
  R half_width, half_height, pixel_size;
  R pad;
};

// Camera of a launch chain.  A chain may carry several frames of the same size (a batch): generation 0 then
// holds the frames one after the other, `tiles_per_frame` tiles of this rank each, frame f seen through
// views[f].  With n_frames == 1 the view is `view` and `views` is not read.
template <class R>
struct DevCamera {
  DevView<R> view;
  int32_t width, height, tiles_x, tiles_y, n_tiles;
  int32_t rank, n_ranks;
  int32_t n_frames, tiles_per_frame;
  float inv_tiles_x, inv_tiles_per_frame;  // reciprocals for the index arithmetic (quotients are corrected by one step)
  const DevView<R>* views;
};

// Per-frame ray pool, structure of arrays indexed by ray id.  Generation g occupies the id range
// [gen_start[g], gen_start[g] + gen_count[g]); generation 0 is the primary rays.
template <class R>
struct RayPool {
  R* org;      // 3 per ray
  R* dir;      // 3 per ray
  R* hit_t;    // nearest t >= 0
  R* hit_uv;   // 2 per ray (smooth triangles)
  int32_t* hit_prim;   // 2 per ray: prim id (-1 = miss), the prim's flags
  R* color;    // 3 per ray: surface colour of the hit (k_shadow), then the colour of the ray (k_combine)
  R* mixr;     // schlick reflectance of the hit (< 0: plain sum)
  int32_t* child;      // 2 per ray: reflected, refracted child ray id (-1 = none)
  // Child-ray slots are handed out without atomics: k_trace leaves, per GROUP of rays (one wave's), the number of
  // hits and an upper bound of the children they can spawn; k_scan turns the bounds into exclusive
  // prefixes; k_spawn places its children at prefix + rank inside the wave.
  uint32_t* blk_hits;
  uint32_t* blk_children;
  uint32_t* blk_base;
  // generation 0 of a frame: the rays [0, n_view_rays) start at the camera, rays_per_view of them per frame of the
  // chain; their origin is read from views[frame] instead of org (null: every origin is in org)
  const DevView<R>* views;
  uint32_t n_view_rays, rays_per_view, n_views;
  float inv_rays_per_view;
  uint32_t* blk_top;  // per chunk of SCAN_CHUNK blocks: child total, hit total, exclusive prefix of the child totals
  int64_t capacity;
};

// Small device-resident control block; read back once per frame.  Every counter sits on its own 128-byte
// line: one word saturates at under 100 atomics per microsecond, and reads of a line that is being
// hammered by atomics wait behind them.
struct alignas(128) PaddedCounter {
  uint32_t v;
  uint32_t pad[31];
};
struct FrameCtl {
  PaddedCounter gen_count[RT355_MAX_GEN + 1];  // slots of the generation (children are placed by upper bound)
  PaddedCounter hit_count[RT355_MAX_GEN + 1];
  PaddedCounter holes[RT355_MAX_GEN + 1];      // slots of the generation that hold no ray (refraction that turned
                                               // out to be total internal reflection)
  PaddedCounter overflow;
  PaddedCounter scan_ticket[RT355_MAX_GEN + 1];  // k_scan: blocks of the launch that have finished
  unsigned long long work[RT355_MAX_LAUNCH][N_WORK_COUNTERS];
};

// ---- host scene builder output (scene_build.cpp) -------------------------------------------------
struct HostPrim {
  int32_t shape;   // index in World.shapes
  int32_t flags;   // kind | PF_*
  int32_t material;
  double inv[12];
  double ymin, ymax;
  double p1[3], p2[3], p3[3];
  double n1[3], n2[3], n3[3], nf[3];
  double lo[3], hi[3];
  bool bounded;
  int32_t inst_index;  // instance prim: which instance; triangle of an instanced mesh: its instance; -1 otherwise
  int32_t blas_root;   // instance prim: root node of its mesh's tree
  int32_t inst_prim;   // triangle of an instanced mesh: prim id of its instance
};

struct HostNode {
  double lo[BVH_WIDTH][3], hi[BVH_WIDTH][3];
  int32_t c[BVH_WIDTH];
};

struct HostScene {
  std::vector<HostPrim> prims;  // final order: leaf order of the top tree, the unbounded prims, then mesh after instanced mesh
  int32_t n_bounded = 0;
  int32_t n_top = 0;        // bounded + unbounded
  int32_t n_instances = 0;
  std::vector<HostNode> nodes;
  int32_t max_depth = 0;  // most traversal-stack entries any descent can need
  std::vector<rt355_material> materials;
  bool any_secondary = false;    // some material is reflective or transparent
  bool any_transparent = false;  // some material is transparent
};

// Returns 0 or RT355_EINVAL (message in err).  pad_rel: relative outward padding of every box.
// cone_exact_limit: scenes with at most this many truncated cones keep them in the every-ray list.
int build_host_scene(const rt355_shape_desc* shapes, size_t n, double pad_rel, int64_t cone_exact_limit, HostScene& out, char* err, size_t errlen);

// Host-side structural check of a built scene (scene_build.cpp); 0 or RT355_EINVAL with a message.
int verify_host_scene(const HostScene& hs, char* err, size_t errlen);

}  // namespace rt355
