// scene_build.cpp -- host side of rt355_scene_upload: runs ONCE per scene.
//
// What the reference redoes for every shape and every ray is hoisted here:
//   * Ray::inv_transform inverts the shape's matrix twice per shape per ray (src/ray.rs:32-37) and
//     Shape::normal a third time per hit (src/shape.rs:111-136): the inverse is computed once, in double,
//     from the f32 matrix.  Transforms must be affine (every builder in src/matrix.rs:15-89 and
//     src/tuple.rs:50-60 is); anything else is RT355_EINVAL where the reference would unwrap() a None.
//   * triangle edges e1 = p2-p1, e2 = p3-p1 and the flat normal normalize((p3-p2) x (p2-p1))
//     (src/triangle.rs:65-66, 83-85).
//   * World::intersect is a brute-force loop over all shapes (src/world.rs:35-44); there is no spatial
//     index in the reference (src/bound.rs is an unused AABB struct and Shape::bound() returns a
//     placeholder for triangles, src/shape.rs:679-692), so the BVH below is new: binned SAH, binary,
//     each node storing both child boxes.  Shapes without a finite bound (planes, cylinders / cones with
//     infinite extents, src/cylinder.rs:40-47) go to a side list that every ray tests.
//   * Group entries are inert in a World (src/shape.rs:376): dropped from the prim list, their index
//     stays reserved so that prim -> shape index is the position in World.shapes.
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>

#include "rt355_device.h"

namespace rt355 {
namespace {

constexpr double EPS_F = 1.0e-3;  // src/epsilon.rs:1

struct Box {
  double lo[3], hi[3];
  void reset() {
    for (int k = 0; k < 3; ++k) lo[k] = INFINITY, hi[k] = -INFINITY;
  }
  void grow(const double* p) {
    for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], p[k]), hi[k] = std::max(hi[k], p[k]);
  }
  void grow(const Box& b) {
    for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], b.lo[k]), hi[k] = std::max(hi[k], b.hi[k]);
  }
  double area() const {
    double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0;
    return 2 * (dx * dy + dy * dz + dz * dx);
  }
};

bool affine_inverse(const float* m, double inv[12]) {
  // m is column-major: element (r,c) = m[c*4+r]
  auto at = [&](int r, int c) { return (double)m[c * 4 + r]; };
  if (m[3] != 0.f || m[7] != 0.f || m[11] != 0.f || m[15] != 1.f) return false;
  double a = at(0, 0), b = at(0, 1), c = at(0, 2), d = at(1, 0), e = at(1, 1), f = at(1, 2), g = at(2, 0), h = at(2, 1), i = at(2, 2);
  double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  double det = a * A + b * B + c * C;
  if (det == 0.0 || !std::isfinite(det)) return false;
  double id = 1.0 / det;
  double r[9] = {A * id, -(b * i - c * h) * id, (b * f - c * e) * id,
                 B * id, (a * i - c * g) * id, -(a * f - c * d) * id,
                 C * id, -(a * h - b * g) * id, (a * e - b * d) * id};
  double tx = at(0, 3), ty = at(1, 3), tz = at(2, 3);
  for (int row = 0; row < 3; ++row) {
    inv[row * 4 + 0] = r[row * 3 + 0];
    inv[row * 4 + 1] = r[row * 3 + 1];
    inv[row * 4 + 2] = r[row * 3 + 2];
    inv[row * 4 + 3] = -(r[row * 3 + 0] * tx + r[row * 3 + 1] * ty + r[row * 3 + 2] * tz);
  }
  // The SIGN of a zero entry reaches the cube's slab test through the zero components of inverse * direction
  // (prim_hits).  A general 4x4 inverse by cofactors (what the reference computes, src/ray.rs:34-35) leaves a zero
  // entry as a sum of products of mixed signs, +0.0, times 1 / det: every zero carries the determinant's sign.  The
  // closed forms above leave zeros of either sign, so they are set here.
  for (int k = 0; k < 12; ++k) {
    if (!std::isfinite(inv[k])) return false;
    if (inv[k] == 0.0) inv[k] = std::copysign(0.0, det);
  }
  return true;
}

void xf_point(const float* m, const double* p, double* out) {
  for (int r = 0; r < 3; ++r) out[r] = (double)m[0 * 4 + r] * p[0] + (double)m[1 * 4 + r] * p[1] + (double)m[2 * 4 + r] * p[2] + (double)m[3 * 4 + r];
}

bool is_identity(const float* m) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      if (m[c * 4 + r] != (r == c ? 1.f : 0.f)) return false;
  return true;
}

// World-space box of the object-space box [lx,hx]x[ly,hy]x[lz,hz] under m.
void box_of_local(const float* m, double lx, double hx, double ly, double hy, double lz, double hz, Box& b) {
  b.reset();
  for (int k = 0; k < 8; ++k) {
    double p[3] = {k & 1 ? hx : lx, k & 2 ? hy : ly, k & 4 ? hz : lz}, q[3];
    xf_point(m, p, q);
    b.grow(q);
  }
}

struct BuildNode {
  Box box;
  int left = -1, right = -1;  // children (build nodes)
  int first = 0, count = 0;   // leaf range in the ordered prim list
};

struct Builder {
  std::vector<HostPrim>& prims;  // bounded prims only, reordered in place
  std::vector<Box> boxes;
  std::vector<double> cx, cy, cz;
  std::vector<double> cost;  // of testing the prim, in identity-triangle tests
  std::vector<int> order;
  std::vector<BuildNode> nodes;
  int max_depth = 0;

  explicit Builder(std::vector<HostPrim>& p) : prims(p) {}

  double centroid(int i, int axis) const { return axis == 0 ? cx[i] : axis == 1 ? cy[i] : cz[i]; }

  int build(int first, int count, int depth) {
    max_depth = std::max(max_depth, depth);
    int id = (int)nodes.size();
    nodes.emplace_back();
    Box box, cbox;
    box.reset(), cbox.reset();
    double cost_all = 0;
    for (int k = first; k < first + count; ++k) {
      int i = order[k];
      cost_all += cost[i];
      box.grow(boxes[i]);
      double c[3] = {cx[i], cy[i], cz[i]};
      cbox.grow(c);
    }
    nodes[id].box = box;
    auto make_leaf = [&]() {
      nodes[id].first = first, nodes[id].count = count;
      return id;
    };
    if (count <= 1) return make_leaf();
    // the leaf reference keeps the count in 5 bits (and bit 31 must stay clear): clamp what the environment asks for
    static const int leaf_max = getenv("RT355_LEAF_MAX") ? std::max(1, std::min(31, atoi(getenv("RT355_LEAF_MAX")))) : LEAF_MAX;
    static const double trav_cost = getenv("RT355_TRAV_COST") ? atof(getenv("RT355_TRAV_COST")) : 4.0;
    bool must_split = count > leaf_max;
    if (depth >= MAX_TREE_DEPTH - 1 && count <= 31) return make_leaf();  // leaf count field is 5 bits

    constexpr int NB = 16;
    double best_cost = INFINITY;
    int best_axis = -1, best_bin = -1;
    for (int axis = 0; axis < 3; ++axis) {
      double lo = cbox.lo[axis], hi = cbox.hi[axis];
      if (!(hi > lo)) continue;
      double scale = NB / (hi - lo);
      Box bb[NB];
      int bc[NB] = {0};
      double bw[NB] = {0};
      for (auto& b : bb) b.reset();
      for (int k = first; k < first + count; ++k) {
        int i = order[k];
        int bin = std::min(NB - 1, std::max(0, (int)((centroid(i, axis) - lo) * scale)));
        bb[bin].grow(boxes[i]);
        bc[bin]++;
        bw[bin] += cost[i];
      }
      double la[NB], ra[NB], lw[NB], rw[NB];
      int lc[NB], rc[NB];
      Box acc;
      acc.reset();
      int n = 0;
      double wsum = 0;
      for (int b = 0; b < NB; ++b) {
        acc.grow(bb[b]);
        n += bc[b], wsum += bw[b];
        la[b] = acc.area(), lc[b] = n, lw[b] = wsum;
      }
      acc.reset();
      n = 0, wsum = 0;
      for (int b = NB - 1; b >= 0; --b) {
        acc.grow(bb[b]);
        n += bc[b], wsum += bw[b];
        ra[b] = acc.area(), rc[b] = n, rw[b] = wsum;
      }
      for (int b = 0; b < NB - 1; ++b) {
        if (lc[b] == 0 || rc[b + 1] == 0) continue;
        double split_cost = la[b] * lw[b] + ra[b + 1] * rw[b + 1];
        if (split_cost < best_cost) best_cost = split_cost, best_axis = axis, best_bin = b;
      }
    }
    int mid = -1;
    if (best_axis >= 0) {
      double leaf_cost = box.area() * cost_all;
      // A node visit costs one dependent memory round trip: measured on the teapot scenes, a leaf is
      // worth splitting only when that saves about four triangle tests; a shape that needs the ray taken
      // to its object space and a quadratic solved counts as four triangles (measured on the 100 000
      // instance scene: same optimum).
      if (!must_split && best_cost + box.area() * trav_cost >= leaf_cost) return make_leaf();
      double lo = cbox.lo[best_axis], hi = cbox.hi[best_axis];
      double scale = NB / (hi - lo);
      auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](int i) {
        int bin = std::min(NB - 1, std::max(0, (int)((centroid(i, best_axis) - lo) * scale)));
        return bin <= best_bin;
      });
      mid = (int)(it - order.begin());
    }
    if (mid <= first || mid >= first + count) {
      // all centroids coincide (or binning failed): split the range in half by index
      if (!must_split) return make_leaf();
      mid = first + count / 2;
    }
    int l = build(first, mid - first, depth + 1);
    int r = build(mid, first + count - mid, depth + 1);
    nodes[id].left = l, nodes[id].right = r;
    return id;
  }
};

int32_t encode_leaf(int first, int count, bool tri_only) {
  return ~(int32_t)((uint32_t)first | (tri_only ? LEAF_TRI_BIT : 0u) | ((uint32_t)count << 26));
}

// Builds the 4-wide BVH of `prims` (bounded prims, any order) and appends it to `out`: the prims, reordered into leaf
// order, go to out.prims (their first index is the value of out.prims.size() on entry), the nodes to out.nodes, root
// first.  Returns the index of the root node, or -1 when there is no prim; `need_out` receives the number of stack
// entries a descent through this tree can need.
int append_tree(std::vector<HostPrim>& prims, HostScene& out, int& need_out) {
  need_out = 0;
  if (prims.empty()) return -1;
  const int prim_base = (int)out.prims.size();
  const int node_base = (int)out.nodes.size();
  Builder B(prims);
  size_t nb = prims.size();
  B.boxes.resize(nb), B.cx.resize(nb), B.cy.resize(nb), B.cz.resize(nb), B.order.resize(nb), B.cost.resize(nb);
  for (size_t i = 0; i < nb; ++i) {
    for (int k = 0; k < 3; ++k) B.boxes[i].lo[k] = prims[i].lo[k], B.boxes[i].hi[k] = prims[i].hi[k];
    B.cx[i] = 0.5 * (prims[i].lo[0] + prims[i].hi[0]);
    B.cy[i] = 0.5 * (prims[i].lo[1] + prims[i].hi[1]);
    B.cz[i] = 0.5 * (prims[i].lo[2] + prims[i].hi[2]);
    B.order[i] = (int)i;
    const int kind = prims[i].flags & PF_KIND_MASK;
    const bool plain_tri = kind >= RT355_TRIANGLE && (prims[i].flags & PF_IDENTITY);
    // an instance costs the ray's trip to object space plus a descent into its own tree
    B.cost[i] = plain_tri ? 1.0 : kind == RT355_INSTANCE ? 16.0 : 4.0;
  }
  B.nodes.reserve(2 * nb);
  int root = B.build(0, (int)nb, 0);
  // reorder prims into leaf order
  for (size_t k = 0; k < nb; ++k) out.prims.push_back(prims[B.order[k]]);
  // Collapse the binary tree into 4-wide nodes: a wide node starts from the two children of a binary
  // inner node and keeps replacing its largest inner child by that child's two children until it has
  // four, or only leaves.  Device nodes are emitted in DFS order.
  auto leaf_ref = [&](int first, int count) -> int32_t {
    bool tri_only = count > 0;
    for (int k = first; k < first + count; ++k) {
      const HostPrim& q = out.prims[prim_base + k];
      if ((q.flags & PF_KIND_MASK) < RT355_TRIANGLE || !(q.flags & PF_IDENTITY)) tri_only = false;
    }
    return encode_leaf(prim_base + first, count, tri_only);
  };
  auto kids_of = [&](int bn) {
    std::vector<int> kids{B.nodes[bn].left, B.nodes[bn].right};
    while ((int)kids.size() < BVH_WIDTH) {
      int pick = -1;
      double best = -1.0;
      for (int k = 0; k < (int)kids.size(); ++k)
        if (B.nodes[kids[k]].left >= 0 && B.nodes[kids[k]].box.area() > best) best = B.nodes[kids[k]].box.area(), pick = k;
      if (pick < 0) break;
      const int bn2 = kids[pick];
      kids[pick] = B.nodes[bn2].left;
      kids.insert(kids.begin() + pick + 1, B.nodes[bn2].right);
    }
    return kids;
  };
  auto empty_slot = [](HostNode& hn, int k) {
    for (int a = 0; a < 3; ++a) hn.lo[k][a] = INFINITY, hn.hi[k][a] = -INFINITY;
    hn.c[k] = encode_leaf(0, 0, false);
  };
  if (B.nodes[root].left < 0) {
    // the root is a leaf: wrap it in one node with a single child
    HostNode hn;
    for (int k = 0; k < BVH_WIDTH; ++k) empty_slot(hn, k);
    for (int a = 0; a < 3; ++a) hn.lo[0][a] = B.nodes[root].box.lo[a], hn.hi[0][a] = B.nodes[root].box.hi[a];
    hn.c[0] = leaf_ref(B.nodes[root].first, B.nodes[root].count);
    out.nodes.push_back(hn);
    need_out = 1;
  } else {
    // wide nodes in DFS order: (binary node it was grown from, its children)
    struct Wide { int bn; std::vector<int> kids; };
    std::vector<Wide> wide;
    std::vector<int> map(B.nodes.size(), -1);
    std::vector<int> stack{root};
    while (!stack.empty()) {
      int bn = stack.back();
      stack.pop_back();
      map[bn] = (int)wide.size();
      wide.push_back({bn, kids_of(bn)});
      const std::vector<int>& kids = wide.back().kids;
      for (int k = (int)kids.size() - 1; k >= 0; --k)
        if (B.nodes[kids[k]].left >= 0) stack.push_back(kids[k]);
    }
    std::vector<int> need(wide.size(), 0);  // stack entries a descent through this node can need
    for (int w = (int)wide.size() - 1; w >= 0; --w) {  // children come after their parent in DFS order
      int deepest = 0;
      for (int kid : wide[w].kids)
        if (B.nodes[kid].left >= 0) deepest = std::max(deepest, need[map[kid]]);
      need[w] = (int)wide[w].kids.size() - 1 + deepest;
    }
    need_out = need[0] + 1;
    for (const Wide& w : wide) {
      HostNode hn;
      for (int k = 0; k < BVH_WIDTH; ++k) empty_slot(hn, k);
      for (int k = 0; k < (int)w.kids.size(); ++k) {
        const BuildNode& c = B.nodes[w.kids[k]];
        for (int a = 0; a < 3; ++a) hn.lo[k][a] = c.box.lo[a], hn.hi[k][a] = c.box.hi[a];
        hn.c[k] = c.left < 0 ? leaf_ref(c.first, c.count) : node_base + map[w.kids[k]];
      }
      out.nodes.push_back(hn);
    }
  }
  return node_base;
}

}  // namespace

int build_host_scene(const rt355_shape_desc* shapes, size_t n, double pad_rel, int64_t cone_exact_limit, HostScene& out, char* err, size_t errlen) {
  // A ray parallel to one half of a truncated cone hits it anywhere on the infinite double cone
  // (src/shape.rs:295-298 pushes c / (-2b) without the min < y < max test): no finite box holds that.
  int64_t n_trunc_cones = 0;
  for (size_t i = 0; i < n; ++i)
    if (shapes[i].kind == RT355_CONE && std::isfinite(shapes[i].min) && std::isfinite(shapes[i].max)) ++n_trunc_cones;
  const bool cones_unbounded = n_trunc_cones <= cone_exact_limit;
  auto fail = [&](const char* what, size_t i) {
    snprintf(err, errlen, "shape %zu: %s", i, what);
    return RT355_EINVAL;
  };
  std::map<std::string, int> mat_index;
  std::vector<HostPrim> bounded, unbounded;
  // Instanced meshes: a run of triangles that share one transformation other than the identity -- what the reference's
  // flattening leaves of a group with a mesh under it (src/group.rs:130-168 folds the group's transformation into
  // every triangle).  Such a run stays in object space under a tree of its own; rays are taken there once per
  // instance instead of once per triangle.  Shorter runs are not worth a second descent.
  struct Instance {
    size_t first_shape;
    std::vector<HostPrim> tris;
  };
  std::vector<Instance> instances;
  std::vector<int32_t> instance_of(n, -1);
  {
    const size_t min_run = getenv("RT355_INSTANCE_MIN") ? (size_t)std::max(1, atoi(getenv("RT355_INSTANCE_MIN"))) : 16;
    auto is_tri = [&](size_t i) { return shapes[i].kind == RT355_TRIANGLE || shapes[i].kind == RT355_SMOOTH_TRIANGLE; };
    for (size_t i = 0; i < n;) {
      if (!is_tri(i) || is_identity(shapes[i].m)) {
        ++i;
        continue;
      }
      size_t j = i + 1, count = 1;
      while (j < n && (shapes[j].kind == RT355_GROUP || (is_tri(j) && memcmp(shapes[j].m, shapes[i].m, sizeof(shapes[i].m)) == 0))) count += shapes[j].kind != RT355_GROUP, ++j;
      if (count >= min_run) {
        for (size_t k = i; k < j; ++k)
          if (shapes[k].kind != RT355_GROUP) instance_of[k] = (int32_t)instances.size();
        instances.push_back({i, {}});
      }
      i = j;
    }
  }
  for (size_t i = 0; i < n; ++i) {
    const rt355_shape_desc& d = shapes[i];
    if (d.kind < 0 || d.kind > RT355_SMOOTH_TRIANGLE) return fail("unknown kind", i);
    if (d.kind == RT355_GROUP) continue;
    HostPrim p;
    memset(&p, 0, sizeof(p));
    p.shape = (int32_t)i;
    p.inst_index = instance_of[i], p.blas_root = -1, p.inst_prim = -1;
    for (int k = 0; k < 16; ++k)
      if (!std::isfinite(d.m[k])) return fail("non-finite transformation", i);
    if (!affine_inverse(d.m, p.inv)) return fail("transformation is singular or not affine", i);
    // material table
    const float* mf = &d.mat.color[0];
    for (size_t k = 0; k < sizeof(rt355_material) / 4; ++k) {
      if (k == 10) continue;  // pattern_kind (int)
      if (std::isnan(mf[k])) return fail("NaN in material", i);
    }
    if (d.mat.pattern_kind < 0 || d.mat.pattern_kind > RT355_PATTERN_TEST) return fail("unknown pattern kind", i);
    std::string key((const char*)&d.mat, sizeof(rt355_material));
    auto it = mat_index.find(key);
    if (it == mat_index.end()) {
      it = mat_index.emplace(key, (int)out.materials.size()).first;
      out.materials.push_back(d.mat);
      if (d.mat.reflective != 0.f || d.mat.transparency != 0.f) out.any_secondary = true;
      if (d.mat.transparency != 0.f) out.any_transparent = true;
    }
    p.material = it->second;
    p.flags = d.kind | (d.closed ? PF_CLOSED : 0) | (is_identity(d.m) ? PF_IDENTITY : 0);
    p.ymin = d.min, p.ymax = d.max;
    if (std::isnan(d.min) || std::isnan(d.max)) return fail("NaN extent", i);
    Box b;
    b.reset();
    p.bounded = true;
    switch (d.kind) {
      case RT355_SPHERE:
      case RT355_CUBE:
        box_of_local(d.m, -1, 1, -1, 1, -1, 1, b);
        break;
      case RT355_PLANE:
        p.bounded = false;
        break;
      case RT355_CYLINDER: {
        if (!std::isfinite(d.min) || !std::isfinite(d.max)) {
          p.bounded = false;
          break;
        }
        // caps accept x^2+z^2 <= 1 + EPSILON (src/shape.rs:468-472)
        double r = std::sqrt(1.0 + EPS_F) * (1 + 1e-6);
        box_of_local(d.m, -r, r, std::min(d.min, d.max), std::max(d.min, d.max), -r, r, b);
        break;
      }
      case RT355_CONE: {
        if (cones_unbounded || !std::isfinite(d.min) || !std::isfinite(d.max)) {
          p.bounded = false;
          break;
        }
        // walls: x^2+z^2 = y^2 + q with |q| <= 0.005/(4|a|), |a| > EPSILON, because the discriminant is
        // rounded to 2 decimals (src/shape.rs:300); caps accept r^2 + EPSILON (src/shape.rs:474-478).
        double ym = std::max(std::fabs((double)d.min), std::fabs((double)d.max));
        double r = std::sqrt(ym * ym + 0.005 / (4 * EPS_F) + EPS_F) * (1 + 1e-6);
        box_of_local(d.m, -r, r, std::min(d.min, d.max), std::max(d.min, d.max), -r, r, b);
        break;
      }
      case RT355_TRIANGLE:
      case RT355_SMOOTH_TRIANGLE: {
        for (int k = 0; k < 3; ++k) {
          if (!std::isfinite(d.p1[k]) || !std::isfinite(d.p2[k]) || !std::isfinite(d.p3[k])) return fail("non-finite vertex", i);
          p.p1[k] = d.p1[k], p.p2[k] = d.p2[k], p.p3[k] = d.p3[k];
          p.n1[k] = d.n1[k], p.n2[k] = d.n2[k], p.n3[k] = d.n3[k];
        }
        // flat normal: local e2 = p3 - p2, local e1 = p2 - p1, normal = normalize(e2 x e1)
        double a[3], c[3];
        for (int k = 0; k < 3; ++k) a[k] = (double)d.p3[k] - (double)d.p2[k], c[k] = (double)d.p2[k] - (double)d.p1[k];
        double nx = a[1] * c[2] - a[2] * c[1], ny = a[2] * c[0] - a[0] * c[2], nz = a[0] * c[1] - a[1] * c[0];
        double len = std::sqrt(nx * nx + ny * ny + nz * nz);
        p.nf[0] = nx / len, p.nf[1] = ny / len, p.nf[2] = nz / len;  // zero-area triangle: NaN, as in the reference
        double q1[3] = {d.p1[0], d.p1[1], d.p1[2]}, q2[3] = {d.p2[0], d.p2[1], d.p2[2]}, q3[3] = {d.p3[0], d.p3[1], d.p3[2]}, w[3];
        if (p.inst_index >= 0) {  // kept in object space: the mesh's own tree is built over the vertices as they are
          p.flags |= PF_IDENTITY | PF_INSTANCED;
          b.grow(q1), b.grow(q2), b.grow(q3);
        } else {
          xf_point(d.m, q1, w), b.grow(w);
          xf_point(d.m, q2, w), b.grow(w);
          xf_point(d.m, q3, w), b.grow(w);
        }
        break;
      }
    }
    if (p.bounded) {
      for (int k = 0; k < 3; ++k)
        if (!std::isfinite(b.lo[k]) || !std::isfinite(b.hi[k])) p.bounded = false;
    }
    if (p.bounded) {
      for (int k = 0; k < 3; ++k) {
        double mag = std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k])) + (b.hi[k] - b.lo[k]);
        double pad = pad_rel * mag + 1e-300;
        p.lo[k] = b.lo[k] - pad, p.hi[k] = b.hi[k] + pad;
      }
      if (p.inst_index >= 0) instances[p.inst_index].tris.push_back(p); else bounded.push_back(p);
    } else {
      if (p.inst_index >= 0) return fail("triangle of an instanced mesh has no finite bound", i);
      unbounded.push_back(p);
    }
  }
  // one prim of the top tree per instance: the world-space box of the mesh's object-space box
  for (size_t k = 0; k < instances.size(); ++k) {
    const Instance& in = instances[k];
    const rt355_shape_desc& d = shapes[in.first_shape];
    HostPrim p;
    memset(&p, 0, sizeof(p));
    p.shape = (int32_t)in.first_shape;
    p.flags = RT355_INSTANCE;
    p.inst_index = (int32_t)k, p.blas_root = -1, p.inst_prim = -1;
    if (!affine_inverse(d.m, p.inv)) return fail("transformation is singular or not affine", in.first_shape);
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const HostPrim& t : in.tris)
      for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], t.lo[a]), hi[a] = std::max(hi[a], t.hi[a]);
    Box b;
    box_of_local(d.m, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], b);
    p.bounded = true;
    for (int a = 0; a < 3; ++a) {
      if (!std::isfinite(b.lo[a]) || !std::isfinite(b.hi[a])) return fail("instanced mesh has no finite bound", in.first_shape);
      double mag = std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])) + (b.hi[a] - b.lo[a]);
      double pad = pad_rel * mag + 1e-300;
      p.lo[a] = b.lo[a] - pad, p.hi[a] = b.hi[a] + pad;
    }
    bounded.push_back(p);
  }
  if (out.materials.size() >= (1u << 21)) {
    snprintf(err, errlen, "too many distinct materials");
    return RT355_EINVAL;
  }
  if (n >= LEAF_FIRST_MASK) {
    snprintf(err, errlen, "too many shapes");
    return RT355_EINVAL;
  }

  // ---- BVH over the bounded prims ---------------------------------------------------------------
  // The top tree first (root = node 0, prims [0, n_bounded)), then the prims no box holds, then the tree of every
  // instanced mesh with its triangles behind it.
  out.n_bounded = (int32_t)bounded.size();
  int need_top = 0, need_mesh = 0;
  append_tree(bounded, out, need_top);
  for (auto& p : unbounded) out.prims.push_back(p);
  out.n_top = (int32_t)out.prims.size();
  out.n_instances = (int32_t)instances.size();
  std::vector<int> inst_prim(instances.size(), -1);  // where each instance's prim ended up in the top tree's leaf order
  for (int q = 0; q < out.n_bounded; ++q)
    if ((out.prims[q].flags & PF_KIND_MASK) == RT355_INSTANCE) inst_prim[out.prims[q].inst_index] = q;
  for (size_t k = 0; k < instances.size(); ++k) {
    int need = 0;
    const size_t first = out.prims.size();
    const int root = append_tree(instances[k].tris, out, need);
    need_mesh = std::max(need_mesh, need);
    out.prims[inst_prim[k]].blas_root = root;
    for (size_t q = first; q < out.prims.size(); ++q) out.prims[q].inst_prim = inst_prim[k];
  }
  // a ray inside a mesh's tree still has the top tree's entries below it on the stack
  out.max_depth = need_top + need_mesh;
  if (out.max_depth > STACK_DEPTH - 2) {
    snprintf(err, errlen, "BVH needs a deeper traversal stack than the kernels have (%d)", out.max_depth);
    return RT355_EINVAL;
  }
  return RT355_OK;
}

// Structural check of a built scene, on the host: every bounded prim sits in exactly one leaf of its tree (the top
// tree, or the tree of the instanced mesh it belongs to), every child box encloses the boxes of all prims beneath it,
// inner nodes are reachable exactly once, every instance points at the root of a tree that holds exactly its
// triangles, and no descent can need more stack entries than max_depth says.  Returns 0 or RT355_EINVAL with a message.
int verify_host_scene(const HostScene& hs, char* err, size_t errlen) {
  const int nb = hs.n_bounded, np = (int)hs.prims.size();
  if (hs.nodes.empty()) {
    if (nb != 0 || hs.n_top != np) { snprintf(err, errlen, "bounded prims but no nodes"); return RT355_EINVAL; }
    return RT355_OK;
  }
  std::vector<int> seen_prim(np, 0), seen_node(hs.nodes.size(), 0);
  // walks the tree under `root`, whose leaves must lie in [lo, hi) and, for a mesh's tree, belong to instance prim `owner`
  auto walk_tree = [&](int root, int lo, int hi, int owner, int& worst) -> int {
    struct Item { int node; int stack; };
    if (root < 0 || root >= (int)hs.nodes.size()) { snprintf(err, errlen, "root out of range"); return RT355_EINVAL; }
    std::vector<Item> todo{{root, 0}};
    worst = 0;
    if (seen_node[root]++) { snprintf(err, errlen, "root %d reached twice", root); return RT355_EINVAL; }
    std::vector<std::pair<int, int>> sub;
    while (!todo.empty()) {
      Item it = todo.back();
      todo.pop_back();
      const HostNode& n = hs.nodes[it.node];
      int kids = 0;
      for (int k = 0; k < BVH_WIDTH; ++k) kids += n.c[k] != -1;
      if (kids == 0) { snprintf(err, errlen, "node %d has no children", it.node); return RT355_EINVAL; }
      worst = std::max(worst, it.stack + kids - 1);
      for (int k = 0; k < BVH_WIDTH; ++k) {
        if (n.c[k] == -1) {
          if (!(n.lo[k][0] > n.hi[k][0])) { snprintf(err, errlen, "empty slot of node %d has a box", it.node); return RT355_EINVAL; }
          continue;
        }
        // collect the prims beneath this child
        sub.clear();
        std::vector<int> walk{n.c[k]};
        while (!walk.empty()) {
          int c = walk.back();
          walk.pop_back();
          if (c >= 0) {
            if (c >= (int)hs.nodes.size()) { snprintf(err, errlen, "child index out of range"); return RT355_EINVAL; }
            for (int j = 0; j < BVH_WIDTH; ++j)
              if (hs.nodes[c].c[j] != -1) walk.push_back(hs.nodes[c].c[j]);
          } else {
            const uint32_t leaf = (uint32_t)~c;
            sub.push_back({(int)(leaf & LEAF_FIRST_MASK), (int)(leaf >> 26)});
          }
        }
        for (auto& fc : sub)
          for (int q = fc.first; q < fc.first + fc.second; ++q) {
            if (q < lo || q >= hi) { snprintf(err, errlen, "leaf range out of bounds"); return RT355_EINVAL; }
            const HostPrim& pr = hs.prims[q];
            if (pr.inst_prim != owner) { snprintf(err, errlen, "prim %d sits in the tree of another owner", q); return RT355_EINVAL; }
            for (int a = 0; a < 3; ++a)
              if (pr.lo[a] < n.lo[k][a] || pr.hi[a] > n.hi[k][a]) {
                snprintf(err, errlen, "prim %d sticks out of child %d of node %d", q, k, it.node);
                return RT355_EINVAL;
              }
          }
        if (n.c[k] >= 0) {
          if (seen_node[n.c[k]]++) { snprintf(err, errlen, "node %d reached twice", n.c[k]); return RT355_EINVAL; }
          todo.push_back({n.c[k], it.stack + kids - 1});
        } else {
          const uint32_t leaf = (uint32_t)~n.c[k];
          const int first = (int)(leaf & LEAF_FIRST_MASK), count = (int)(leaf >> 26);
          const bool tri_only = (leaf & LEAF_TRI_BIT) != 0;
          for (int q = first; q < first + count; ++q) {
            seen_prim[q]++;
            const bool is_tri = (hs.prims[q].flags & PF_KIND_MASK) >= RT355_TRIANGLE && (hs.prims[q].flags & PF_IDENTITY);
            if (tri_only && !is_tri) { snprintf(err, errlen, "triangle-only leaf holds prim %d of another kind", q); return RT355_EINVAL; }
            if (owner >= 0 && !tri_only) { snprintf(err, errlen, "a mesh's tree holds a leaf that is not triangles only"); return RT355_EINVAL; }
          }
        }
      }
    }
    return RT355_OK;
  };
  int worst_top = 0, worst_mesh = 0, n_inst = 0;
  if (nb > 0) {
    if (int rc = walk_tree(0, 0, nb, -1, worst_top)) return rc;
    worst_top += 1;
  }
  for (int q = 0; q < nb; ++q) {
    const HostPrim& pr = hs.prims[q];
    if ((pr.flags & PF_KIND_MASK) != RT355_INSTANCE) continue;
    ++n_inst;
    int worst = 0;
    if (int rc = walk_tree(pr.blas_root, hs.n_top, np, q, worst)) return rc;
    worst_mesh = std::max(worst_mesh, worst + 1);
  }
  if (n_inst != hs.n_instances) { snprintf(err, errlen, "%d instance prims, %d instances recorded", n_inst, hs.n_instances); return RT355_EINVAL; }
  for (int q = 0; q < np; ++q) {
    const bool in_tree = q < nb || q >= hs.n_top;
    if (seen_prim[q] != (in_tree ? 1 : 0)) { snprintf(err, errlen, "prim %d is in %d leaves", q, seen_prim[q]); return RT355_EINVAL; }
    if (q >= hs.n_top && !(hs.prims[q].flags & PF_INSTANCED)) { snprintf(err, errlen, "prim %d behind the top prims is not part of an instance", q); return RT355_EINVAL; }
  }
  for (size_t i = 0; i < hs.nodes.size(); ++i)
    if (!seen_node[i]) { snprintf(err, errlen, "node %zu is unreachable", i); return RT355_EINVAL; }
  if (worst_top + worst_mesh > hs.max_depth) { snprintf(err, errlen, "a descent needs %d stack entries, recorded %d", worst_top + worst_mesh, hs.max_depth); return RT355_EINVAL; }
  return RT355_OK;
}

}  // namespace rt355
