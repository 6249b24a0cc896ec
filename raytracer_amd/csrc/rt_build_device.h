// BVH construction on the device (opt-in, rt_hip_scene_build): where the reference builds its tree on the
// host (BVH_SAH::build, /root/reference/src/bvh/bvh_node_sah.rs:147-159, 256-461) and the default path
// of this library takes that tree as input, this builder makes the tree from the primitives alone.
//
// By default it is the reference's own tree: the top-down recursion of BVHNode_SAH::build run level by level
// (rt_build_topdown.h), by the split rule RtScene.split_method names.  Two other ways to a tree are kept for
// comparison (RT_HIP_BUILD=0 / 1), both over the primitives in Morton order:
//
//   k_morton + radix sort   63-bit Morton code of every box centre
//   k_karras                the binary radix tree over the sorted codes (Karras 2012): every internal node
//                           finds its range and split with no communication; or
//   k_ploc_*                locally-ordered clustering (Meister and Bittner 2018): clusters that choose each
//                           other as cheapest neighbour merge, until one is left
//
// Shared by all three:
//
//   k_prim_bounds   true bounds of every element (f32, rounded outward), scene bounds by atomics
//   k_refit_pass    child boxes and subtree heights bottom-up, one tree level per launch (a kernel
//                   boundary is the only synchronisation: no hand-off between workgroups in flight)
//   k_emit_*        the records k_extend reads: 64-byte f32 nodes, 32-byte grid nodes, primitive
//                   records in leaf order
//
// Included by rt_hip.hip (one translation unit with the handle, the kernels and the error plumbing).
// What differs from an uploaded tree: boxes are the primitives' TRUE bounds (the reference boxes a quad by two
// corners, quad.rs:86), so every box bounds its contents and the order-free walk always applies; exact-t ties
// between primitives are decided by the built tree's depth-first leaf order (the reference's, with the default
// method).
#pragma once

namespace build {

struct Box32 {
  float mn[3], mx[3];
};
static_assert(sizeof(Box32) == 24, "box record");

constexpr uint32_t kChildLeaf = 0x80000000u;

// order-preserving map float -> uint32 (for atomicMin / atomicMax)
RT_DEV uint32_t f32_ordered(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
RT_DEV float f32_from_ordered(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

RT_DEV float down32(double x) { return nextafterf(__double2float_rd(x), -INFINITY); }
RT_DEV float up32(double x) { return nextafterf(__double2float_ru(x), INFINITY); }

// `scene_bounds`: 6 ordered uints (min xyz, max xyz), initialised to 0xffffffff x3, 0 x3; `bad`: set if a box is not finite
__global__ void __launch_bounds__(256) k_prim_bounds(int32_t n, const RtElementRef *elements, const RtTriangle *tris, const RtQuad *quads,
                                                      const RtSphere *spheres, Box32 *boxes, uint32_t *scene_bounds, uint32_t *bad) {
  // bounds of what this thread has seen, as ordered uints
  uint32_t smn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, smx[3] = {0u, 0u, 0u};
  for (int32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    {
      auto add = [&](double x, double y, double z) {
        lo[0] = fmin(lo[0], x); lo[1] = fmin(lo[1], y); lo[2] = fmin(lo[2], z);
        hi[0] = fmax(hi[0], x); hi[1] = fmax(hi[1], y); hi[2] = fmax(hi[2], z);
      };
      RtElementRef er = elements[e];
      if (er.kind == RT_PRIM_TRIANGLE) {
        const RtTriangle &t = tris[er.index];
        add(t.v0.x, t.v0.y, t.v0.z);
        add(t.v0.x + t.v0v1.x, t.v0.y + t.v0v1.y, t.v0.z + t.v0v1.z);
        add(t.v0.x + t.v0v2.x, t.v0.y + t.v0v2.y, t.v0.z + t.v0v2.z);
      } else if (er.kind == RT_PRIM_QUAD) {
        const RtQuad &q = quads[er.index];
        for (int cu = 0; cu < 2; cu++)
          for (int cv = 0; cv < 2; cv++) add(q.q.x + cu * q.u.x + cv * q.v.x, q.q.y + cu * q.u.y + cv * q.v.y, q.q.z + cu * q.u.z + cv * q.v.z);
      } else {
        const RtSphere &s = spheres[er.index];
        double r = fabs(s.radius);
        add(s.center.x - r, s.center.y - r, s.center.z - r);
        add(s.center.x + r, s.center.y + r, s.center.z + r);
      }
      bool finite = true;
      Box32 b;
      for (int a = 0; a < 3; a++) {
        if (!(hi[a] - lo[a] >= 0.0001)) {  // aabb.rs:122-152: thin axes grow by 5e-5 on each side
          lo[a] -= 0.00005;
          hi[a] += 0.00005;
        }
        finite = finite && fabs(lo[a]) <= 1e30 && fabs(hi[a]) <= 1e30;
        b.mn[a] = down32(lo[a]);
        b.mx[a] = up32(hi[a]);
      }
      if (!finite) {
        atomicOr(bad, 0x100u);
        for (int a = 0; a < 3; a++) b.mn[a] = b.mx[a] = 0.0f, lo[a] = hi[a] = 0.0;
      }
      boxes[e] = b;
      for (int a = 0; a < 3; a++) smn[a] = min(smn[a], f32_ordered(b.mn[a])), smx[a] = max(smx[a], f32_ordered(b.mx[a]));
    }
  }
  // wave reduction, then the workgroup's waves through LDS: one atomic per workgroup and bound
  __shared__ uint32_t part[4][6];
  for (int a = 0; a < 3; a++) {
    uint32_t mn = smn[a], mx = smx[a];
    for (int off = 32; off > 0; off >>= 1) {
      mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][a] = mn, part[threadIdx.x >> 6][3 + a] = mx;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    atomicMin(&scene_bounds[a], min(min(part[0][a], part[1][a]), min(part[2][a], part[3][a])));
    atomicMax(&scene_bounds[3 + a], max(max(part[0][3 + a], part[1][3 + a]), max(part[2][3 + a], part[3][3 + a])));
  }
}


RT_DEV uint64_t spread21(uint32_t v) {  // bit k of v -> bit 3k
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

__global__ void __launch_bounds__(256) k_morton(int32_t n, const Box32 *boxes, const uint32_t *scene_bounds, uint64_t *keys, uint32_t *ids) {
  float lo[3], inv[3];
  for (int a = 0; a < 3; a++) {
    lo[a] = f32_from_ordered(scene_bounds[a]);
    float ext = f32_from_ordered(scene_bounds[3 + a]) - lo[a];
    inv[a] = ext > 0.0f ? 2097152.0f / ext : 0.0f;
  }
  for (int32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    Box32 b = boxes[e];
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
      float c = 0.5f * (b.mn[a] + b.mx[a]);
      float x = (c - lo[a]) * inv[a];
      q[a] = (uint32_t)fminf(fmaxf(x, 0.0f), 2097151.0f);
    }
    keys[e] = spread21(q[0]) << 2 | spread21(q[1]) << 1 | spread21(q[2]);
    ids[e] = (uint32_t)e;
  }
}

// common-prefix length of the keys at sorted positions i and j (equal keys: continued by the positions themselves)
RT_DEV int prefix_len(const uint64_t *keys, int32_t n, int32_t i, int32_t j) {
  if (j < 0 || j >= n) return -1;
  uint64_t a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz(i ^ j);
  return __clzll((long long)(a ^ b));
}

// Internal node i of the radix tree over n sorted keys (Karras 2012, section 4): children as
// kChildLeaf | sorted position, or an internal node's index.  Node 0 is the root.
__global__ void __launch_bounds__(256) k_karras(int32_t n, const uint64_t *keys, uint32_t *child_l, uint32_t *child_r) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n - 1; i += gridDim.x * blockDim.x) {
    int d = prefix_len(keys, n, i, i + 1) - prefix_len(keys, n, i, i - 1) >= 0 ? 1 : -1;
    int dmin = prefix_len(keys, n, i, i - d);
    int lmax = 2;
    while (prefix_len(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
      if (prefix_len(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = prefix_len(keys, n, i, j);
    int s = 0, t = l;
    do {
      t = (t + 1) >> 1;
      if (prefix_len(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    child_l[i] = (min(i, j) == gamma) ? (kChildLeaf | (uint32_t)gamma) : (uint32_t)gamma;
    child_r[i] = (max(i, j) == gamma + 1) ? (kChildLeaf | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
  }
}

// One level of the bottom-up pass: a node whose children were complete BEFORE this launch gets its
// box (union of its two child boxes) and its height.  Heights ping-pong between two arrays, -1 = not yet.
__global__ void __launch_bounds__(256) k_refit_pass(int32_t n_internal, const uint32_t *child_l, const uint32_t *child_r, const uint32_t *ids,
                                                     const Box32 *prim_boxes, Box32 *node_boxes, const int32_t *height_in, int32_t *height_out) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_internal; i += gridDim.x * blockDim.x) {
    int32_t h = height_in[i];
    if (h < 0) {
      uint32_t c[2] = {child_l[i], child_r[i]};
      int32_t hc[2];
      for (int k = 0; k < 2; k++) hc[k] = (c[k] & kChildLeaf) ? 0 : height_in[c[k]];
      if (hc[0] >= 0 && hc[1] >= 0) {
        Box32 a = (c[0] & kChildLeaf) ? prim_boxes[ids[c[0] & ~kChildLeaf]] : node_boxes[c[0]];
        Box32 b = (c[1] & kChildLeaf) ? prim_boxes[ids[c[1] & ~kChildLeaf]] : node_boxes[c[1]];
        Box32 u;
        for (int ax = 0; ax < 3; ax++) u.mn[ax] = fminf(a.mn[ax], b.mn[ax]), u.mx[ax] = fmaxf(a.mx[ax], b.mx[ax]);
        node_boxes[i] = u;
        h = 1 + max(hc[0], hc[1]);
      }
    }
    height_out[i] = h;
  }
}

// ---------------------------------------------------------------- agglomerative build
// The tree by parallel locally-ordered clustering (Meister and Bittner 2018) instead of the radix tree: the clusters -- at first
// the primitives, in Morton order -- each look for the neighbour within `radius` places whose union with them has the smallest
// surface area; two clusters that choose each other become the children of a new node, which takes the place of the first of
// them; the rest stay; repeated until one cluster is left.  Nearby clusters merge by what their union costs, not by where the
// Morton code's bits happen to split space, which is what the surface-area heuristic of the reference's builder asks for too
// (bvh_node_sah.rs:385-461).  One iteration is three launches and a prefix sum, and its only hand-off is the kernel boundary.
constexpr int kPlocBlock = 256, kPlocMaxRadius = 32;
constexpr uint32_t kNoParent = 0xffffffffu;

RT_DEV float union_half_area(const Box32 &a, const Box32 &b) {  // never NaN: extents are finite and >= 0 (a product may overflow to +inf)
  const float dx = fmaxf(a.mx[0], b.mx[0]) - fminf(a.mn[0], b.mn[0]);
  const float dy = fmaxf(a.mx[1], b.mx[1]) - fminf(a.mn[1], b.mn[1]);
  const float dz = fmaxf(a.mx[2], b.mx[2]) - fminf(a.mn[2], b.mn[2]);
  return dx * dy + dy * dz + dz * dx;
}

__global__ void __launch_bounds__(256) k_ploc_init(int32_t n, const uint32_t *ids, const Box32 *prim_boxes, uint32_t *cl_ref, Box32 *cl_box) {
  for (int32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    cl_ref[j] = kChildLeaf | (uint32_t)j;
    cl_box[j] = prim_boxes[ids[j]];
  }
}

// nn[i] = the cluster within `radius` places of i whose union with i is smallest (the lower place among equals; c >= 2).
// The relation is symmetric and the order total, so the pair with the globally smallest union that involves the lowest place
// always chooses each other: every iteration merges at least one pair.
__global__ void __launch_bounds__(kPlocBlock) k_ploc_nearest(uint32_t c, int radius, const Box32 *cl_box, uint32_t *nn) {
  __shared__ Box32 tile[kPlocBlock + 2 * kPlocMaxRadius];
  for (uint32_t base = blockIdx.x * kPlocBlock; base < c; base += gridDim.x * kPlocBlock) {  // the same trips for every thread of the block
    __syncthreads();
    for (int k = threadIdx.x; k < kPlocBlock + 2 * radius; k += kPlocBlock) {
      const int64_t g = (int64_t)base - radius + k;
      if (g >= 0 && g < (int64_t)c) tile[k] = cl_box[g];
    }
    __syncthreads();
    const uint32_t i = base + threadIdx.x;
    if (i < c) {
      const Box32 a = tile[threadIdx.x + radius];
      float best = 0.0f;
      uint32_t bj = i;
      for (int k = -radius; k <= radius; k++) {
        const int64_t j = (int64_t)i + k;
        if (k == 0 || j < 0 || j >= (int64_t)c) continue;
        const float d = union_half_area(a, tile[(int)threadIdx.x + radius + k]);
        if (bj == i || d < best) {
          best = d;
          bj = (uint32_t)j;
        }
      }
      nn[i] = bj;
    }
  }
}

// flags[i]: low word 1 if place i survives (alone, or as the new node of a pair), high word 1 if it is the first of a pair
__global__ void __launch_bounds__(256) k_ploc_flags(uint32_t c, const uint32_t *nn, unsigned long long *flags) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < c; i += gridDim.x * blockDim.x) {
    const uint32_t j = nn[i];
    const bool mutual = nn[j] == i;
    flags[i] = (mutual && i > j ? 0ull : 1ull) | (mutual && i < j ? 1ull << 32 : 0ull);
  }
}

struct PlocNodes {  // the tree under construction; node ids count up from 0 in the order of the merges
  uint32_t *child_l, *child_r, *parent, *leaf_parent, *leaves;  // leaves[id]: primitives below node id
  int32_t *height;
  Box32 *box;
};
RT_DEV void ploc_adopt(const PlocNodes &t, uint32_t child, uint32_t id) {
  if (child & kChildLeaf) t.leaf_parent[child & ~kChildLeaf] = id;
  else t.parent[child] = id;
}
// `scan` = exclusive prefix sums of `flags`; pair k of this iteration becomes node node_base + k.  counts[0] = clusters left.
__global__ void __launch_bounds__(256) k_ploc_merge(uint32_t c, uint32_t node_base, const uint32_t *nn, const unsigned long long *flags,
                                                     const unsigned long long *scan, const uint32_t *ref_in, const Box32 *box_in, uint32_t *ref_out,
                                                     Box32 *box_out, PlocNodes t, uint32_t *counts) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < c; i += gridDim.x * blockDim.x) {
    const unsigned long long f = flags[i], s = scan[i];
    if (i == c - 1) counts[0] = (uint32_t)s + (uint32_t)f;
    if (!(uint32_t)f) continue;
    const uint32_t dst = (uint32_t)s;
    if (f >> 32) {
      const uint32_t j = nn[i], id = node_base + (uint32_t)(s >> 32);
      const uint32_t l = ref_in[i], r = ref_in[j];
      const Box32 a = box_in[i], b = box_in[j];
      Box32 u;
      for (int ax = 0; ax < 3; ax++) u.mn[ax] = fminf(a.mn[ax], b.mn[ax]), u.mx[ax] = fmaxf(a.mx[ax], b.mx[ax]);
      t.child_l[id] = l;
      t.child_r[id] = r;
      t.box[id] = u;
      t.height[id] = 1 + max((l & kChildLeaf) ? 0 : t.height[l], (r & kChildLeaf) ? 0 : t.height[r]);
      t.leaves[id] = ((l & kChildLeaf) ? 1u : t.leaves[l]) + ((r & kChildLeaf) ? 1u : t.leaves[r]);
      t.parent[id] = kNoParent;
      ploc_adopt(t, l, id);
      ploc_adopt(t, r, id);
      ref_out[dst] = id;
      box_out[dst] = u;
    } else {
      ref_out[dst] = ref_in[i];
      box_out[dst] = box_in[i];
    }
  }
}

// Position of every leaf in the depth-first, left-first order of the finished tree (what decides exact ties between
// primitives): the leaves under the left siblings passed on the way up to the root.
__global__ void __launch_bounds__(256) k_leaf_ranks(int32_t n, PlocNodes t, uint32_t *rank) {
  for (int32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    uint32_t from = kChildLeaf | (uint32_t)j, node = t.leaf_parent[j], before = 0;
    while (node != kNoParent) {
      const uint32_t l = t.child_l[node];
      if (l != from) before += (l & kChildLeaf) ? 1u : t.leaves[l];
      from = node;
      node = t.parent[node];
    }
    rank[j] = before;
  }
}

// Leaf refs and primitive records in leaf order.  `tri_pos[j]` = triangles before sorted position j;
// `slot_in_kind[e]` = position of element e among the quad (sphere) elements, in element order;
// `rank[j]` = place of the leaf in the tree's depth-first order (null: the sorted position itself, as in the radix tree).
__global__ void __launch_bounds__(256) k_emit_prims(int32_t n, const uint32_t *ids, const uint32_t *tri_pos, const uint32_t *slot_in_kind,
                                                     const RtElementRef *elements, const RtTriangle *tris, const RtQuad *quads, const RtSphere *spheres,
                                                     DevTriangle *out_tris, uint32_t tri_stride, DevTriShade *out_shade, DevQuad *out_quads,
                                                     DevSphere *out_spheres, uint32_t *leaf_ref, const uint32_t *rank) {
  for (int32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    uint32_t e = ids[j];
    RtElementRef er = elements[e];
    const uint32_t leaf_rank = rank ? rank[j] : (uint32_t)j;
    uint32_t slot;
    if (er.kind == RT_PRIM_TRIANGLE) {
      slot = tri_pos[j];
      const RtTriangle &t = tris[er.index];
      DevTriangle d;
      d.v0[0] = t.v0.x; d.v0[1] = t.v0.y; d.v0[2] = t.v0.z;
      d.e1[0] = t.v0v1.x; d.e1[1] = t.v0v1.y; d.e1[2] = t.v0v1.z;
      d.e2[0] = t.v0v2.x; d.e2[1] = t.v0v2.y; d.e2[2] = t.v0v2.z;
      d.element = e;
      d.rank = leaf_rank;
      *reinterpret_cast<DevTriangle *>(reinterpret_cast<char *>(out_tris) + (size_t)slot * tri_stride) = d;  // records tri_stride bytes apart
      DevTriShade sh;
      sh.normal[0] = t.normal.x; sh.normal[1] = t.normal.y; sh.normal[2] = t.normal.z;
      sh.material = t.material;
      sh.pad = 0;
      out_shade[slot] = sh;
    } else if (er.kind == RT_PRIM_QUAD) {
      slot = slot_in_kind[e];
      const RtQuad &q = quads[er.index];
      DevQuad d;
      d.q[0] = q.q.x; d.q[1] = q.q.y; d.q[2] = q.q.z;
      d.u[0] = q.u.x; d.u[1] = q.u.y; d.u[2] = q.u.z;
      d.v[0] = q.v.x; d.v[1] = q.v.y; d.v[2] = q.v.z;
      d.normal[0] = q.normal.x; d.normal[1] = q.normal.y; d.normal[2] = q.normal.z;
      d.w[0] = q.w.x; d.w[1] = q.w.y; d.w[2] = q.w.z;
      d.d = q.d;
      d.area = q.area;
      d.element = e;
      d.material = (uint32_t)q.material;
      d.rank = leaf_rank;
      d.pad = 0;
      out_quads[slot] = d;
    } else {
      slot = slot_in_kind[e];
      const RtSphere &s = spheres[er.index];
      DevSphere d;
      d.center[0] = s.center.x; d.center[1] = s.center.y; d.center[2] = s.center.z;
      d.radius = s.radius;
      d.element = e;
      d.material = (uint32_t)s.material;
      d.rank = leaf_rank;
      d.pad = 0;
      out_spheres[slot] = d;
    }
    leaf_ref[j] = kRefLeaf | ((uint32_t)er.kind << kRefKindShift) | slot;
  }
}

__global__ void __launch_bounds__(256) k_flag_triangles(int32_t n, const uint32_t *ids, const RtElementRef *elements, uint32_t *flags) {
  for (int32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) flags[j] = elements[ids[j]].kind == RT_PRIM_TRIANGLE ? 1u : 0u;
}

struct Grid {
  double mn[3], step[3];
};
// grid value at or below x / at or above x on axis a, checked against the value the traversal's constants stand for
RT_DEV uint16_t grid_down(const Grid &g, double x, int a) {
  double q = floor((x - g.mn[a]) / g.step[a]);
  while (q > 0.0 && g.mn[a] + q * g.step[a] > x - fabs(x) * 1e-15) q -= 1.0;
  return (uint16_t)fmin(fmax(q, 0.0), 65535.0);
}
RT_DEV uint16_t grid_up(const Grid &g, double x, int a) {
  double q = ceil((x - g.mn[a]) / g.step[a]);
  while (q < 65535.0 && g.mn[a] + q * g.step[a] < x + fabs(x) * 1e-15) q += 1.0;
  return (uint16_t)fmin(fmax(q, 0.0), 65535.0);
}

// The node records of k_extend: f32 boxes, and the same boxes on the 16-bit grid.  sums[0] += summed edge length of the
// grid boxes, sums[1] += of the f32 boxes (how much the grid inflates them decides whether the grid form is used).
__global__ void __launch_bounds__(256) k_emit_nodes(int32_t n_internal, const uint32_t *child_l, const uint32_t *child_r, const uint32_t *ids,
                                                     const uint32_t *leaf_ref, const Box32 *prim_boxes, const Box32 *node_boxes, Grid grid,
                                                     DevNode *nodes, DevNodeQ *nodes_q, double *sums) {
  double grown = 0.0, size = 0.0;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_internal; i += gridDim.x * blockDim.x) {
    uint32_t c[2] = {child_l[i], child_r[i]};
    Box32 b[2];
    uint32_t ref[2];
    for (int k = 0; k < 2; k++) {
      if (c[k] & kChildLeaf) {
        uint32_t pos = c[k] & ~kChildLeaf;
        b[k] = prim_boxes[ids[pos]];
        ref[k] = leaf_ref[pos];
      } else {
        b[k] = node_boxes[c[k]];
        ref[k] = c[k];
      }
    }
    DevNode d;
    DevNodeQ q;
    for (int a = 0; a < 3; a++) {
      d.lmin[a] = b[0].mn[a]; d.lmax[a] = b[0].mx[a];
      d.rmin[a] = b[1].mn[a]; d.rmax[a] = b[1].mx[a];
      q.lmin[a] = grid_down(grid, b[0].mn[a], a); q.lmax[a] = grid_up(grid, b[0].mx[a], a);
      q.rmin[a] = grid_down(grid, b[1].mn[a], a); q.rmax[a] = grid_up(grid, b[1].mx[a], a);
      grown += ((double)(q.lmax[a] - q.lmin[a]) + (double)(q.rmax[a] - q.rmin[a])) * grid.step[a];
      size += ((double)b[0].mx[a] - (double)b[0].mn[a]) + ((double)b[1].mx[a] - (double)b[1].mn[a]);
    }
    d.left = q.left = ref[0];
    d.right = q.right = ref[1];
    d.pad[0] = d.pad[1] = 0;
    nodes[i] = d;
    nodes_q[i] = q;
  }
  for (int off = 32; off > 0; off >>= 1) {
    grown += __shfl_xor(grown, off);
    size += __shfl_xor(size, off);
  }
  if ((threadIdx.x & 63) == 0 && size > 0.0) {
    atomicAdd(&sums[0], grown);
    atomicAdd(&sums[1], size);
  }
}

}  // namespace build

#include "rt_build_topdown.h"

// Scratch device memory of a build, freed on every way out.
struct BuildTemps {
  std::vector<void *> ptrs;
  ~BuildTemps() {
    for (void *p : ptrs) (void)hipFree(p);
  }
  template <typename T>
  hipError_t alloc(T **out, size_t count) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) {
      ptrs.push_back(p);
      *out = static_cast<T *>(p);
    }
    return e;
  }
  template <typename T>
  hipError_t upload(const T **out, const T *src, size_t count) {
    T *p = nullptr;
    hipError_t e = alloc(&p, count);
    if (e == hipSuccess && count) e = hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
    *out = p;
    return e;
  }
};

static int scene_build_impl(const RtScene *s, int device, RtSceneHandle **out) {
  using namespace build;
  g_error.clear();
  if (!s || !out) return fail(RT_ERR_BAD_ARG, "scene or out is NULL");
  *out = nullptr;
  const Tuning tuning = tuning_from_env();
  const bool verbose = tuning.verbose != 0;
  auto lap_t0 = std::chrono::steady_clock::now();
  int rc = validate_primitives(s);
  if (rc != RT_OK) return rc;
  const int32_t n = s->n_elements;
  // internal-node indices and leaf slots are 28-bit fields of a child ref (elements may name a primitive more than once, so the
  // per-kind counts checked above do not bound n); k_karras' range arithmetic i +- 2 * n also stays inside an int below this
  if ((uint32_t)(n - 1) > kRefIndexMask) return fail(RT_ERR_UNSUPPORTED, "more than 2^28 elements: node indices do not fit a child reference");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(RT_ERR_BAD_ARG, "no such device");
  HIP_TRY(hipSetDevice(device));

  // position of every quad / sphere element among the elements of its kind, and the triangle elements' count
  std::vector<uint32_t> slot_in_kind((size_t)n, 0);
  uint32_t n_tri_el = 0, n_quad_el = 0, n_sph_el = 0;
  for (int32_t e = 0; e < n; e++) {
    int k = s->elements[e].kind;
    slot_in_kind[(size_t)e] = k == RT_PRIM_TRIANGLE ? n_tri_el++ : k == RT_PRIM_QUAD ? n_quad_el++ : n_sph_el++;
  }

  RtSceneHandle *h = new (std::nothrow) RtSceneHandle;
  if (!h) return fail(RT_ERR_OOM, "out of host memory");
  struct FreeOnUnwind {
    RtSceneHandle *h;
    ~FreeOnUnwind() {
      if (h) rt_hip_scene_free(h);
    }
  } unwind{h};
  h->device = device;
  h->tuning = tuning;
  h->n_elements = n;
  if (hipStreamCreate(&h->stream) != hipSuccess) return fail(RT_ERR_HIP, "hipStreamCreate failed");
  hipStream_t st = h->stream;
  rc = noalloc_scratch_of(device, &h->noalloc_scratch);
  if (rc != RT_OK) return rc;

  BuildTemps tmp;
  const RtElementRef *d_elements;
  const RtTriangle *d_tris;
  const RtQuad *d_quads;
  const RtSphere *d_spheres;
  const uint32_t *d_slot;
  HIP_TRY(tmp.upload(&d_elements, s->elements, (size_t)n));
  HIP_TRY(tmp.upload(&d_tris, s->triangles, (size_t)s->n_triangles));
  HIP_TRY(tmp.upload(&d_quads, s->quads, (size_t)s->n_quads));
  HIP_TRY(tmp.upload(&d_spheres, s->spheres, (size_t)s->n_spheres));
  HIP_TRY(tmp.upload(&d_slot, slot_in_kind.data(), (size_t)n));
  Box32 *d_boxes, *d_node_boxes;
  uint32_t *d_scene, *d_bad, *d_ids[2], *d_child_l, *d_child_r, *d_flags, *d_tri_pos, *d_leaf_ref;
  uint64_t *d_keys[2];
  int32_t *d_height[2];
  double *d_sums;
  const int32_t n_internal = n - 1;
  HIP_TRY(tmp.alloc(&d_boxes, (size_t)n));
  HIP_TRY(tmp.alloc(&d_node_boxes, (size_t)n_internal));
  HIP_TRY(tmp.alloc(&d_scene, 8));
  d_bad = d_scene + 6;
  for (int k = 0; k < 2; k++) {
    HIP_TRY(tmp.alloc(&d_ids[k], (size_t)n));
    HIP_TRY(tmp.alloc(&d_keys[k], (size_t)n));
    HIP_TRY(tmp.alloc(&d_height[k], (size_t)n_internal));
  }
  HIP_TRY(tmp.alloc(&d_child_l, (size_t)n_internal));
  HIP_TRY(tmp.alloc(&d_child_r, (size_t)n_internal));
  HIP_TRY(tmp.alloc(&d_flags, (size_t)n));
  HIP_TRY(tmp.alloc(&d_tri_pos, (size_t)n));
  HIP_TRY(tmp.alloc(&d_leaf_ref, (size_t)n));
  HIP_TRY(tmp.alloc(&d_sums, 2));
  const uint32_t scene_init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
  HIP_TRY(hipMemcpyAsync(d_scene, scene_init, sizeof scene_init, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(d_sums, 0, 2 * sizeof(double), st));
  auto t_upload = std::chrono::steady_clock::now();

  const int block = 256;
  const int grid = grid_for(n, block, 256 * 8);
  const int gi = grid_for(std::max(n_internal, 1), block, 256 * 8);
  // how the tree is made: 0 radix tree over Morton codes, 1 locally-ordered clustering, 2 / 3 the reference's own top-down
  // build with its bucketed surface-area heuristic / its midpoint split (rt_build_topdown.h)
  if (s->split_method != RT_SPLIT_SAH && s->split_method != RT_SPLIT_MID) return fail(RT_ERR_BAD_ARG, "RtScene.split_method must be RT_SPLIT_SAH or RT_SPLIT_MID");
  const int method = n_internal == 0 ? 0 : tuning.build_method >= 0 ? tuning.build_method : (s->split_method == RT_SPLIT_MID ? 3 : 2);
  const bool topdown = method >= 2, ploc = method == 1;
  hipLaunchKernelGGL(k_prim_bounds, dim3(grid), dim3(block), 0, st, n, d_elements, d_tris, d_quads, d_spheres, d_boxes, d_scene, d_bad);
  size_t sort_bytes = 0, scan_bytes = 0;
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_keys[0], d_keys[1], d_ids[0], d_ids[1], n, 0, 63, st));
  HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_flags, d_tri_pos, n, st));
  char *d_lib_tmp;
  HIP_TRY(tmp.alloc(&d_lib_tmp, std::max(sort_bytes, scan_bytes)));
  const uint64_t *keys = d_keys[1];
  const uint32_t *ids = d_ids[1];  // the elements in leaf order
  int td_levels = 0;
  if (!topdown) {
    hipLaunchKernelGGL(k_morton, dim3(grid), dim3(block), 0, st, n, d_boxes, d_scene, d_keys[0], d_ids[0]);
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(d_lib_tmp, sort_bytes, d_keys[0], d_keys[1], d_ids[0], d_ids[1], n, 0, 63, st));
  } else {
    // ---- the reference's top-down build, one level of the tree per trip
    const int sah = method == 2 ? 1 : 0;
    const uint32_t chunks = ((uint32_t)n + 63u) / 64u, n_items = 2u * chunks;
    Rec64 *d_recs;
    uint32_t *d_seg_of[2], *d_class_key, *d_n_left, *d_new_count, *d_base, *d_counts, *d_item_key;
    uint8_t *d_bins;
    Cnt16 *d_item_cnt, *d_carry_cnt, *d_seg_total;
    Box12 *d_class_box, *d_item_box, *d_carry_box;
    double *d_child_cb;
    TdSegments sg[2];
    HIP_TRY(tmp.alloc(&d_recs, (size_t)n));
    HIP_TRY(tmp.alloc(&d_bins, (size_t)n));
    HIP_TRY(tmp.alloc(&d_class_box, (size_t)n));
    HIP_TRY(tmp.alloc(&d_class_key, (size_t)n));
    HIP_TRY(tmp.alloc(&d_seg_total, (size_t)n));
    HIP_TRY(tmp.alloc(&d_n_left, (size_t)n));
    HIP_TRY(tmp.alloc(&d_new_count, (size_t)n));
    HIP_TRY(tmp.alloc(&d_base, (size_t)n));
    HIP_TRY(tmp.alloc(&d_child_cb, 12 * (size_t)n));
    HIP_TRY(tmp.alloc(&d_counts, 4));
    HIP_TRY(tmp.alloc(&d_item_key, (size_t)n_items));
    HIP_TRY(tmp.alloc(&d_item_cnt, (size_t)n_items));
    HIP_TRY(tmp.alloc(&d_carry_cnt, (size_t)n_items));
    HIP_TRY(tmp.alloc(&d_item_box, (size_t)n_items));
    HIP_TRY(tmp.alloc(&d_carry_box, (size_t)n_items));
    for (int k = 0; k < 2; k++) {
      HIP_TRY(tmp.alloc(&d_seg_of[k], (size_t)n));
      HIP_TRY(tmp.alloc(&sg[k].lo, (size_t)n));
      HIP_TRY(tmp.alloc(&sg[k].cnt, (size_t)n));
      HIP_TRY(tmp.alloc(&sg[k].node, (size_t)n));
      HIP_TRY(tmp.alloc(&sg[k].active, (size_t)n));
      HIP_TRY(tmp.alloc(&sg[k].cb, 6 * (size_t)n));
    }
    const TdTree tree{d_child_l, d_child_r};
    Cnt16 zero_cnt;
    for (int k = 0; k < kTdBuckets; k++) zero_cnt.c[k] = 0u;
    Box12 no_box;
    for (int k = 0; k < 3; k++) no_box.bmn[k] = no_box.cmn[k] = INFINITY, no_box.bmx[k] = no_box.cmx[k] = -INFINITY;
    size_t b1 = 0, b2 = 0, b3 = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveScanByKey(nullptr, b1, d_item_key, d_item_cnt, d_carry_cnt, Cnt16Sum(), zero_cnt, n_items, hipcub::Equality(), st));
    HIP_TRY(hipcub::DeviceScan::ExclusiveScanByKey(nullptr, b2, d_item_key, d_item_box, d_carry_box, Box12Union(), no_box, n_items, hipcub::Equality(), st));
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, b3, d_new_count, d_base, n, st));
    char *d_td_tmp;
    HIP_TRY(tmp.alloc(&d_td_tmp, std::max(b1, std::max(b2, b3))));
    const int gw = grid_for((int64_t)chunks, kTdWavesPerBlock, 256 * 8);  // one wave per 64 positions
    int cur = 0;  // d_ids[cur], d_seg_of[cur], sg[cur] describe the level about to be split
    hipLaunchKernelGGL(k_td_element_recs, dim3(grid), dim3(block), 0, st, n, d_elements, d_tris, d_quads, d_spheres, d_recs, d_ids[0], d_seg_of[0]);
    HIP_TRY(hipMemsetAsync(d_class_key, 0, (size_t)n * sizeof(uint32_t), st));  // every element in one class: the root's boxes
    hipLaunchKernelGGL(k_td_class_boxes, dim3(gw), dim3(block), 0, st, n, d_ids[0], d_class_key, d_recs, d_class_box, d_item_key, d_item_box);
    HIP_TRY(hipcub::DeviceScan::ExclusiveScanByKey(d_td_tmp, b2, d_item_key, d_item_box, d_carry_box, Box12Union(), no_box, n_items, hipcub::Equality(), st));
    hipLaunchKernelGGL(k_td_root, dim3(1), dim3(64), 0, st, n, d_class_box, d_carry_box, sg[0], tree);
    uint32_t n_seg = 1, n_active = n >= 3 ? 1u : 0u;
    while (n_active > 0) {
      if (++td_levels > 4 * kMaxStackDepth) return fail(RT_ERR_UNSUPPORTED, "the top-down device build does not terminate: use the host build for this scene");
      const int gs = grid_for((int64_t)n_seg, block, 256 * 8);
      hipLaunchKernelGGL(k_td_count, dim3(gw), dim3(block), 0, st, n, sah, d_ids[cur], d_seg_of[cur], sg[cur], d_recs, d_bins, d_item_key, d_item_cnt, d_bad);
      HIP_TRY(hipcub::DeviceScan::ExclusiveScanByKey(d_td_tmp, b1, d_item_key, d_item_cnt, d_carry_cnt, Cnt16Sum(), zero_cnt, n_items, hipcub::Equality(), st));
      hipLaunchKernelGGL(k_td_totals, dim3(gw), dim3(block), 0, st, n, d_seg_of[cur], sg[cur], d_bins, d_carry_cnt, d_seg_total);
      hipLaunchKernelGGL(k_td_scatter, dim3(gw), dim3(block), 0, st, n, d_ids[cur], d_seg_of[cur], sg[cur], d_bins, d_carry_cnt, d_seg_total, d_ids[cur ^ 1],
                         d_class_key);
      hipLaunchKernelGGL(k_td_class_boxes, dim3(gw), dim3(block), 0, st, n, d_ids[cur ^ 1], d_class_key, d_recs, d_class_box, d_item_key, d_item_box);
      HIP_TRY(hipcub::DeviceScan::ExclusiveScanByKey(d_td_tmp, b2, d_item_key, d_item_box, d_carry_box, Box12Union(), no_box, n_items, hipcub::Equality(), st));
      hipLaunchKernelGGL(k_td_split, dim3(gs), dim3(block), 0, st, n_seg, sah, sg[cur], d_seg_total, d_class_box, d_carry_box, tree, d_n_left, d_child_cb,
                         d_new_count, d_bad);
      HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_td_tmp, b3, d_new_count, d_base, (int)n_seg, st));
      HIP_TRY(hipMemsetAsync(d_counts, 0, 2 * sizeof(uint32_t), st));
      hipLaunchKernelGGL(k_td_segments, dim3(gs), dim3(block), 0, st, n_seg, sg[cur], d_n_left, d_child_cb, d_base, sg[cur ^ 1], d_counts);
      hipLaunchKernelGGL(k_td_seg_of, dim3(grid), dim3(block), 0, st, n, d_seg_of[cur], sg[cur], d_n_left, d_base, d_seg_of[cur ^ 1]);
      uint32_t counts[2] = {0, 0};
      HIP_TRY(hipMemcpyAsync(counts, d_counts, sizeof counts, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      HIP_TRY(hipGetLastError());
      if (counts[1] <= n_seg || counts[1] > (uint32_t)n) return fail(RT_ERR_HIP, "the top-down device build lost track of its ranges");
      n_active = counts[0];
      n_seg = counts[1];
      cur ^= 1;
    }
    ids = d_ids[cur];
  }
  hipLaunchKernelGGL(k_flag_triangles, dim3(grid), dim3(block), 0, st, n, ids, d_elements, d_flags);
  HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_lib_tmp, scan_bytes, d_flags, d_tri_pos, n, st));

  // device records of the primitives
  DevTriangle *o_tris;
  DevTriShade *o_shade;
  DevQuad *o_quads;
  DevSphere *o_spheres;
  // triangle records beyond the caches one per 128-byte line, as in an upload
  const bool padded = tuning.tri_pad >= 0 ? tuning.tri_pad != 0 : n_tri_el * sizeof(DevTriangle) > ((size_t)256 << 20);
  const uint32_t tri_stride = padded ? 128u : (uint32_t)sizeof(DevTriangle);
  {
    char *bytes = nullptr;
    if ((rc = dev_alloc(h, &bytes, n_tri_el * tri_stride)) != RT_OK) return rc;
    o_tris = reinterpret_cast<DevTriangle *>(bytes);
  }
  if ((rc = dev_alloc(h, &o_shade, n_tri_el)) != RT_OK || (rc = dev_alloc(h, &o_quads, n_quad_el)) != RT_OK ||
      (rc = dev_alloc(h, &o_spheres, n_sph_el)) != RT_OK)
    return rc;
  h->dev.triangles = o_tris;
  h->dev.tri_stride = tri_stride;
  h->dev.tri_shade = o_shade;
  h->dev.quads = o_quads;
  h->dev.spheres = o_spheres;

  // scene bounds (and the finiteness flag) are needed on the host for the grid
  uint32_t scene_host[8];
  HIP_TRY(hipMemcpyAsync(scene_host, d_scene, sizeof scene_host, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (scene_host[6] & 0x100u) return fail(RT_ERR_UNSUPPORTED, "a primitive has a non-finite or enormous coordinate: the device build needs finite bounds");
  if (scene_host[6]) return fail(RT_ERR_UNSUPPORTED, "the top-down device build met a bucket index out of range or an empty child (a NaN coordinate?): use the host build for this scene");
  auto unordered = [](uint32_t u) {
    uint32_t bits = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &bits, 4);
    return (double)f;
  };
  Grid gridc;
  for (int a = 0; a < 3; a++) {
    double lo = unordered(scene_host[a]), hi = unordered(scene_host[3 + a]);
    gridc.mn[a] = lo;
    gridc.step[a] = (hi - lo) / 65535.0;
    if (!(gridc.step[a] > 0.0)) gridc.step[a] = 1.0;
    gridc.step[a] *= 1.0 + 1e-12;
  }

  int32_t tree_height = 0;
  uint32_t root_ref;
  DevNode *o_nodes = nullptr;
  DevNodeQ *o_nodes_q = nullptr;
  if ((rc = dev_alloc(h, &o_nodes, (size_t)std::max(n_internal, 1))) != RT_OK) return rc;
  uint32_t root_node = 0;
  uint32_t *d_rank = nullptr;  // depth-first place of every leaf where it is not its sorted position
  int ploc_iterations = 0;
  // ---- the tree: children of every internal node, its box and its height
  if (n_internal == 0) {
    // a single element: the root is its leaf
  } else if (ploc) {
    PlocNodes t;
    t.child_l = d_child_l;
    t.child_r = d_child_r;
    t.box = d_node_boxes;
    t.height = d_height[0];
    uint32_t *cl_ref[2], *d_nn, *d_counts;
    Box32 *cl_box[2];
    unsigned long long *d_pflags, *d_pscan;
    HIP_TRY(tmp.alloc(&t.parent, (size_t)n_internal));
    HIP_TRY(tmp.alloc(&t.leaf_parent, (size_t)n));
    HIP_TRY(tmp.alloc(&t.leaves, (size_t)n_internal));
    for (int k = 0; k < 2; k++) {
      HIP_TRY(tmp.alloc(&cl_ref[k], (size_t)n));
      HIP_TRY(tmp.alloc(&cl_box[k], (size_t)n));
    }
    HIP_TRY(tmp.alloc(&d_nn, (size_t)n));
    HIP_TRY(tmp.alloc(&d_pflags, (size_t)n));
    HIP_TRY(tmp.alloc(&d_pscan, (size_t)n));
    HIP_TRY(tmp.alloc(&d_counts, 4));
    HIP_TRY(tmp.alloc(&d_rank, (size_t)n));
    size_t pscan_bytes = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, pscan_bytes, d_pflags, d_pscan, n, st));
    char *d_pscan_tmp;
    HIP_TRY(tmp.alloc(&d_pscan_tmp, pscan_bytes));
    hipLaunchKernelGGL(k_ploc_init, dim3(grid), dim3(block), 0, st, n, ids, d_boxes, cl_ref[0], cl_box[0]);
    const int radius = std::max(1, std::min(tuning.ploc_radius, kPlocMaxRadius));
    uint32_t c = (uint32_t)n, node_base = 0;
    int cur = 0;
    while (c > 1) {
      const int gc = grid_for((int64_t)c, block, 256 * 8);
      hipLaunchKernelGGL(k_ploc_nearest, dim3(gc), dim3(kPlocBlock), 0, st, c, radius, cl_box[cur], d_nn);
      hipLaunchKernelGGL(k_ploc_flags, dim3(gc), dim3(block), 0, st, c, d_nn, d_pflags);
      HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_pscan_tmp, pscan_bytes, d_pflags, d_pscan, (int)c, st));
      hipLaunchKernelGGL(k_ploc_merge, dim3(gc), dim3(block), 0, st, c, node_base, d_nn, d_pflags, d_pscan, cl_ref[cur], cl_box[cur], cl_ref[cur ^ 1],
                         cl_box[cur ^ 1], t, d_counts);
      uint32_t left = 0;
      HIP_TRY(hipMemcpyAsync(&left, d_counts, sizeof left, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (left == 0 || left >= c) return fail(RT_ERR_HIP, "the clustering pass of the device build merged nothing");
      node_base += c - left;  // every pair takes one cluster away and adds one node
      c = left;
      cur ^= 1;
      ploc_iterations++;
    }
    if (node_base != (uint32_t)n_internal) return fail(RT_ERR_HIP, "the clustering pass of the device build lost a node");
    root_node = (uint32_t)n_internal - 1u;  // the last pair merged
    HIP_TRY(hipMemcpyAsync(&tree_height, t.height + root_node, sizeof tree_height, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(k_leaf_ranks, dim3(grid), dim3(block), 0, st, n, t, d_rank);
    HIP_TRY(hipStreamSynchronize(st));
  } else {
    if (!topdown) hipLaunchKernelGGL(k_karras, dim3(gi), dim3(block), 0, st, n, keys, d_child_l, d_child_r);
    HIP_TRY(hipMemsetAsync(d_height[0], 0xff, (size_t)n_internal * sizeof(int32_t), st));
    int cur = 0, passes = 0;
    int32_t root_height = -1;
    while (root_height < 0) {
      for (int k = 0; k < 8; k++, passes++, cur ^= 1)
        hipLaunchKernelGGL(k_refit_pass, dim3(gi), dim3(block), 0, st, n_internal, d_child_l, d_child_r, ids, d_boxes, d_node_boxes, d_height[cur],
                           d_height[cur ^ 1]);
      HIP_TRY(hipMemcpyAsync(&root_height, d_height[cur], sizeof root_height, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (passes > 4096) return fail(RT_ERR_HIP, "the bottom-up pass of the device build does not terminate");
    }
    tree_height = root_height;
  }
  if (tree_height > kMaxStackDepth)
    return fail(RT_ERR_UNSUPPORTED, "the device-built BVH is " + std::to_string(tree_height) + " levels deep (limit 64): use the host build for this scene");
  // ---- the records the kernels read
  hipLaunchKernelGGL(k_emit_prims, dim3(grid), dim3(block), 0, st, n, ids, d_tri_pos, d_slot, d_elements, d_tris, d_quads, d_spheres, o_tris, tri_stride,
                     o_shade, o_quads, o_spheres, d_leaf_ref, d_rank);
  if (n_internal == 0) {
    HIP_TRY(hipMemcpyAsync(&root_ref, d_leaf_ref, sizeof root_ref, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemset(o_nodes, 0, sizeof(DevNode)));
  } else {
    if ((rc = dev_alloc(h, &o_nodes_q, (size_t)n_internal)) != RT_OK) return rc;
    hipLaunchKernelGGL(k_emit_nodes, dim3(gi), dim3(block), 0, st, n_internal, d_child_l, d_child_r, ids, d_leaf_ref, d_boxes, d_node_boxes, gridc, o_nodes,
                       o_nodes_q, d_sums);
    root_ref = root_node;
  }
  double sums[2] = {0.0, 0.0};
  HIP_TRY(hipMemcpyAsync(sums, d_sums, sizeof sums, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  h->dev.nodes = o_nodes;
  // the grid form only where it inflates the boxes by less than a tenth (summed edge lengths), as for uploaded trees
  const bool use_grid = o_nodes_q && !tuning.no_grid && sums[0] <= 1.10 * sums[1];
  h->dev.nodes_q = use_grid ? o_nodes_q : nullptr;
  h->dev.nodes_d = nullptr;
  h->dev.nodes_w = nullptr;
  h->dev.root_w = root_ref;
  h->dev.stack_depth_w = 1;
  for (int a = 0; a < 3; a++) h->dev.grid_min[a] = gridc.mn[a], h->dev.grid_step[a] = gridc.step[a];
  // Trees beyond the caches also get the wide records of an upload, made the same way: the binary grid records come back to
  // the host, collapse_to_wide opens them four at a time, the result goes up (10 M triangles: 0.3 GB down, 0.3 GB up and one
  // pass of one host thread over the nodes: most of the preparation then, and still a fraction of a host build's).
  if (use_grid && wide_nodes_wanted(tuning, (size_t)n_internal)) {
    auto t_w0 = std::chrono::steady_clock::now();
    std::vector<DevNodeQ> host_q((size_t)n_internal);
    HIP_TRY(hipMemcpy(host_q.data(), o_nodes_q, host_q.size() * sizeof(DevNodeQ), hipMemcpyDeviceToHost));
    std::vector<DevNodeW> nodes_w;
    int32_t stack_depth_w = 1;
    collapse_to_wide(host_q, root_ref, gridc.step, nodes_w, stack_depth_w);
    if (!nodes_w.empty()) {
      if ((rc = dev_upload(h, &h->dev.nodes_w, nodes_w)) != RT_OK) return rc;
      h->dev.root_w = 0;
      h->dev.stack_depth_w = stack_depth_w;
    }
    if (verbose)
      fprintf(stderr, "rt_hip build: %zu wide nodes for %d binary ones, stack need %d, %.1f ms\n", nodes_w.size(), n_internal, stack_depth_w,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_w0).count());
  }

  // materials and attractors as in an upload
  std::vector<DevMaterial> mats((size_t)s->n_materials);
  for (int i = 0; i < s->n_materials; i++) {
    mats[(size_t)i].albedo[0] = s->materials[i].albedo[0];
    mats[(size_t)i].albedo[1] = s->materials[i].albedo[1];
    mats[(size_t)i].albedo[2] = s->materials[i].albedo[2];
    mats[(size_t)i].param = s->materials[i].param;
    mats[(size_t)i].kind = s->materials[i].kind;
    mats[(size_t)i].pad = 0;
  }
  std::vector<uint32_t> attractors((size_t)s->n_attractors);
  for (int i = 0; i < s->n_attractors; i++) {
    int e = s->attractors[i];
    attractors[(size_t)i] = kRefLeaf | ((uint32_t)s->elements[e].kind << kRefKindShift) | slot_in_kind[(size_t)e];
  }
  if ((rc = dev_upload(h, &h->dev.materials, mats)) != RT_OK || (rc = dev_upload(h, &h->dev.attractors, attractors)) != RT_OK) return rc;
  h->dev.n_attractors = s->n_attractors;
  h->dev.n_materials = s->n_materials;
  h->dev.root = root_ref;
  h->dev.stack_depth = std::max(tree_height, 1);
  if ((rc = alloc_stack_spill(h)) != RT_OK) return rc;
  h->dev.quirk = 0;  // true bounds everywhere: every box bounds its contents
  h->dev.danger = nullptr;
  h->dev.n_danger = 0;
  h->dev.camera = s->camera;
  h->bvh_depth = tree_height;
  h->n_internal = n_internal;
  h->device_built = true;
  resolve_packet_default(h);
  if ((rc = noalloc_warm_up(h)) != RT_OK) return rc;
  if (verbose) {
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "rt_hip build: %d elements, %s, height %d, copies in %.1f ms, build %.1f ms, grid %s (boxes %.1f %% larger)\n", n,
            ploc ? ("clustered in " + std::to_string(ploc_iterations) + " iterations").c_str()
            : topdown ? ((method == 2 ? "the reference's bucketed top-down build, " : "the reference's midpoint top-down build, ") + std::to_string(td_levels) + " levels").c_str()
                      : "radix tree",
            tree_height,
            std::chrono::duration<double, std::milli>(t_upload - lap_t0).count(), std::chrono::duration<double, std::milli>(now - t_upload).count(),
            use_grid ? "on" : "off", sums[1] > 0 ? 100.0 * (sums[0] / sums[1] - 1.0) : 0.0);
  }
  unwind.h = nullptr;
  *out = h;
  return RT_OK;
}

// The tree a handle holds, as RtNode records: internal nodes first (node 0 = the root; device index order),
// then one leaf per child slot that names a primitive.  Child boxes are the f32 boxes the device tests.
static int scene_download_tree_impl(RtSceneHandle *h, RtNode *nodes, int32_t capacity, int32_t *n_nodes) {
  g_error.clear();
  if (!h || !n_nodes) return fail(RT_ERR_BAD_ARG, "NULL argument");
  const int32_t n_internal = h->n_internal, total = 2 * n_internal + 1;
  *n_nodes = total;
  if (!nodes) return RT_OK;
  if (capacity < total) return fail(RT_ERR_BAD_ARG, "node buffer too small: " + std::to_string(total) + " nodes");
  HIP_TRY(hipSetDevice(h->device));
  auto element_of_ref = [&](uint32_t ref, int32_t *element) -> int {
    uint32_t kind = (ref >> kRefKindShift) & 3u, idx = ref & kRefIndexMask, e = 0;
    const char *src = kind == RT_PRIM_TRIANGLE ? (const char *)&tri_record(h->dev, idx)->element
                      : kind == RT_PRIM_QUAD   ? (const char *)&h->dev.quads[idx].element
                                               : (const char *)&h->dev.spheres[idx].element;
    HIP_TRY(hipMemcpy(&e, src, sizeof e, hipMemcpyDeviceToHost));
    *element = (int32_t)e;
    return RT_OK;
  };
  if (n_internal == 0) {
    memset(&nodes[0], 0, sizeof(RtNode));
    nodes[0].is_leaf = 1;
    return element_of_ref(h->dev.root, &nodes[0].element_index);
  }
  std::vector<DevNode> dn((size_t)n_internal);
  HIP_TRY(hipMemcpy(dn.data(), h->dev.nodes, dn.size() * sizeof(DevNode), hipMemcpyDeviceToHost));
  // element ids of the leaves: one strided copy per kind would do; the records are small next to the nodes, copy them whole
  std::vector<DevTriangle> tris;
  std::vector<DevQuad> quads;
  std::vector<DevSphere> spheres;
  uint32_t max_idx[3] = {0, 0, 0};
  bool any[3] = {false, false, false};
  for (const DevNode &d : dn)
    for (uint32_t ref : {d.left, d.right})
      if (ref & kRefLeaf) {
        uint32_t kind = (ref >> kRefKindShift) & 3u;
        any[kind] = true;
        max_idx[kind] = std::max(max_idx[kind], ref & kRefIndexMask);
      }
  if (any[RT_PRIM_TRIANGLE]) {
    tris.resize((size_t)max_idx[RT_PRIM_TRIANGLE] + 1);
    HIP_TRY(hipMemcpy2D(tris.data(), sizeof(DevTriangle), h->dev.triangles, h->dev.tri_stride, sizeof(DevTriangle), tris.size(), hipMemcpyDeviceToHost));
  }
  if (any[RT_PRIM_QUAD]) {
    quads.resize((size_t)max_idx[RT_PRIM_QUAD] + 1);
    HIP_TRY(hipMemcpy(quads.data(), h->dev.quads, quads.size() * sizeof(DevQuad), hipMemcpyDeviceToHost));
  }
  if (any[RT_PRIM_SPHERE]) {
    spheres.resize((size_t)max_idx[RT_PRIM_SPHERE] + 1);
    HIP_TRY(hipMemcpy(spheres.data(), h->dev.spheres, spheres.size() * sizeof(DevSphere), hipMemcpyDeviceToHost));
  }
  // the caller's root is nodes[0]: the device's root record trades places with record 0 if it is another one
  const int32_t root = (int32_t)(h->dev.root & kRefIndexMask);
  auto place = [&](int32_t i) { return i == root ? 0 : i == 0 ? root : i; };
  int32_t next_leaf = n_internal;
  for (int32_t i = 0; i < n_internal; i++) {
    const DevNode &d = dn[(size_t)i];
    RtNode &o = nodes[place(i)];
    memset(&o, 0, sizeof o);
    for (int a = 0; a < 3; a++) {
      o.child_l_min[a] = d.lmin[a]; o.child_l_max[a] = d.lmax[a];
      o.child_r_min[a] = d.rmin[a]; o.child_r_max[a] = d.rmax[a];
    }
    const uint32_t refs[2] = {d.left, d.right};
    int32_t child[2];
    for (int k = 0; k < 2; k++) {
      if (refs[k] & kRefLeaf) {
        uint32_t kind = (refs[k] >> kRefKindShift) & 3u, idx = refs[k] & kRefIndexMask;
        RtNode &leaf = nodes[next_leaf];
        memset(&leaf, 0, sizeof leaf);
        leaf.is_leaf = 1;
        leaf.element_index = (int32_t)(kind == RT_PRIM_TRIANGLE ? tris[idx].element : kind == RT_PRIM_QUAD ? quads[idx].element : spheres[idx].element);
        child[k] = next_leaf++;
      } else {
        child[k] = place((int32_t)refs[k]);
      }
    }
    o.child_l_index = child[0];
    o.child_r_index = child[1];
  }
  return RT_OK;
}
