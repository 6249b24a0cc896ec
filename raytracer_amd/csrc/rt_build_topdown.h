// The reference's own tree, built on the device (rt_hip_scene_build, the default method): BVHNode_SAH::build
// (/root/reference/src/bvh/bvh_node_sah.rs:256-461) is a top-down recursion -- box of the element-box centroids,
// its longest axis, sixteen buckets along it (or the midpoint, BVHSplitMethod::Mid), Buckets::split_objects (:28-79),
// children from the concatenated buckets -- in which the children of a node depend on nothing but the node's own element
// list.  All nodes of one depth are therefore split together, one LEVEL per trip of a host loop:
//
//   k_td_bins      the bucket of every element of every node of the level (one lane per element)
//   scan by key    per node, how many elements of each bucket precede each element (sixteen counters per element)
//   k_td_scatter   the stable counting sort of every node's range by bucket = the concatenation of the reference's
//                  per-bucket vectors (:68-74), so the element order inside a child is the reference's
//   scan by key    per (node, bucket), the union of the element boxes and of their centroids
//   k_td_split     per node: bucket areas, the fifteen plane costs summed in the reference's order, the first cheapest
//                  plane, the two children (leaves, two-element nodes, or ranges for the next level)
//   k_td_segments  the next level's table of ranges
//
// A subtree over c elements has c - 1 internal nodes, so a node's place in the reference's preorder is known when its
// parent splits: left child right after the parent, right child n_left places after it.  The leaves come out in the
// reference's depth-first order (position in the final element order = leaf rank), which is what decides exact ties.
// The element boxes are recomputed from the primitive records (v0 + v0v1 for a triangle's second vertex): a centroid may
// differ from the host's in its last bit, which moves an element to another bucket only when it sits on a bucket
// boundary to within that bit.
#pragma once

namespace build {

struct Rec64 {  // what the reference's build reads of an element: its box (Element::bounding_box) and the box's centroid
  double mn[3], mx[3], c[3];
};

RT_DEV void rec_from_corners(const double *p, const double *q, bool pad, Rec64 &r) {
  for (int a = 0; a < 3; a++) {
    double lo = fmin(q[a], p[a]), hi = fmax(q[a], p[a]);  // Aabb::new_from_points, aabb.rs:59-75
    if (pad && !(hi - lo >= 0.0001)) {                    // Aabb::pad, aabb.rs:122-152
      lo = lo - 0.0001 / 2.0;
      hi = hi + 0.0001 / 2.0;
    }
    r.mn[a] = lo;
    r.mx[a] = hi;
    r.c[a] = lo + (hi - lo) / 2.;  // aabb.rs:214-220
  }
}

__global__ void __launch_bounds__(256) k_td_element_recs(int32_t n, const RtElementRef *elements, const RtTriangle *tris, const RtQuad *quads,
                                                          const RtSphere *spheres, Rec64 *recs, uint32_t *idx, uint32_t *seg_of) {
  for (int32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const RtElementRef er = elements[e];
    Rec64 r;
    if (er.kind == RT_PRIM_TRIANGLE) {  // triangle.rs:55-70
      const RtTriangle &t = tris[er.index];
      const double v0[3] = {t.v0.x, t.v0.y, t.v0.z};
      const double v1[3] = {t.v0.x + t.v0v1.x, t.v0.y + t.v0v1.y, t.v0.z + t.v0v1.z};
      const double v2[3] = {t.v0.x + t.v0v2.x, t.v0.y + t.v0v2.y, t.v0.z + t.v0v2.z};
      double lo[3], hi[3];
      for (int a = 0; a < 3; a++) lo[a] = fmin(v0[a], fmin(v1[a], v2[a])), hi[a] = fmax(v0[a], fmax(v1[a], v2[a]));
      rec_from_corners(lo, hi, true, r);
    } else if (er.kind == RT_PRIM_QUAD) {  // quad.rs:86: the box of two opposite corners
      const RtQuad &q = quads[er.index];
      const double p[3] = {q.q.x, q.q.y, q.q.z};
      const double o[3] = {q.q.x + q.u.x + q.v.x, q.q.y + q.u.y + q.v.y, q.q.z + q.u.z + q.v.z};
      rec_from_corners(p, o, true, r);
    } else {  // sphere.rs: centre -+ radius, not padded
      const RtSphere &s = spheres[er.index];
      const double p[3] = {s.center.x - s.radius, s.center.y - s.radius, s.center.z - s.radius};
      const double o[3] = {s.center.x + s.radius, s.center.y + s.radius, s.center.z + s.radius};
      rec_from_corners(p, o, false, r);
    }
    recs[e] = r;
    idx[e] = (uint32_t)e;  // bvh_node_sah.rs:149
    seg_of[e] = 0u;
  }
}

constexpr int kTdBuckets = 16;  // bvh_node_sah.rs:385
struct Cnt16 {
  uint32_t c[kTdBuckets];
};
struct Cnt16Sum {
  __host__ __device__ Cnt16 operator()(const Cnt16 &a, const Cnt16 &b) const {
    Cnt16 r;
    for (int k = 0; k < kTdBuckets; k++) r.c[k] = a.c[k] + b.c[k];
    return r;
  }
};
struct OneHot16 {
  __host__ __device__ Cnt16 operator()(uint8_t b) const {
    Cnt16 r;
    for (int k = 0; k < kTdBuckets; k++) r.c[k] = k == (int)b ? 1u : 0u;
    return r;
  }
};
struct Box12 {  // union of element boxes, union of their centroids
  double bmn[3], bmx[3], cmn[3], cmx[3];
};
struct Box12Union {  // Aabb::merge (aabb.rs:81-97) is min / max per axis: exact in any grouping
  __host__ __device__ Box12 operator()(const Box12 &a, const Box12 &b) const {
    Box12 r;
    for (int k = 0; k < 3; k++) {
      r.bmn[k] = fmin(a.bmn[k], b.bmn[k]);
      r.bmx[k] = fmax(a.bmx[k], b.bmx[k]);
      r.cmn[k] = fmin(a.cmn[k], b.cmn[k]);
      r.cmx[k] = fmax(a.cmx[k], b.cmx[k]);
    }
    return r;
  }
};
struct RecBox12 {
  const Rec64 *recs;
  __host__ __device__ Box12 operator()(uint32_t e) const {
    const Rec64 &r = recs[e];
    Box12 b;
    for (int k = 0; k < 3; k++) b.bmn[k] = r.mn[k], b.bmx[k] = r.mx[k], b.cmn[k] = b.cmx[k] = r.c[k];
    return b;
  }
};

struct TdSegments {  // the ranges of one level, in position order; every position belongs to exactly one
  uint32_t *lo, *cnt, *node;  // node: the internal node a range of two or more elements becomes
  uint32_t *active;           // 1: three or more elements, split at this level
  double *cb;                 // 6 per range: box of the centroids (min xyz, max xyz), valid for active ranges
};

// How the elements of an active range are dealt to buckets (bvh_node_sah.rs:346-406).
struct TdPlan {
  int axis, n_bins;
  bool halves;  // all centroids coincide on the longest axis: halves by position (:351-362)
  double k0, k1, splitpoint;
};
RT_DEV TdPlan td_plan(const double *cb, bool sah) {
  TdPlan p;
  const double sx = cb[3] - cb[0], sy = cb[4] - cb[1], sz = cb[5] - cb[2];
  p.axis = (sx > sy && sx > sz) ? 0 : (sy > sz ? 1 : 2);  // Aabb::largest_axis, aabb.rs:165-175
  const double extent = cb[3 + p.axis] - cb[p.axis];
  p.halves = extent < 2.220446049250313e-16;
  p.n_bins = (sah && !p.halves) ? kTdBuckets : 2;
  p.k0 = cb[p.axis];
  p.k1 = ((double)kTdBuckets * (1.0 - 0.0001)) / (cb[3 + p.axis] - cb[p.axis]);
  p.splitpoint = cb[p.axis] + extent / 2.0;
  return p;
}

__global__ void __launch_bounds__(256) k_td_bins(int32_t n, int sah, const uint32_t *idx, const uint32_t *seg_of, TdSegments sg, const Rec64 *recs,
                                                  uint8_t *bins, uint32_t *error) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t s = seg_of[i];
    uint32_t bin = 0;
    if (sg.active[s]) {
      const TdPlan p = td_plan(sg.cb + 6 * (size_t)s, sah != 0);
      if (p.halves) {
        bin = (uint32_t)i - sg.lo[s] < sg.cnt[s] / 2u ? 0u : 1u;
      } else if (sah) {
        const double b = p.k1 * (recs[idx[i]].c[p.axis] - p.k0);
        // `as usize`: truncation, saturating, NaN -> 0
        bin = (b != b || b <= 0.0) ? 0u : (b >= 4e9 ? 0xffffffffu : (uint32_t)b);
        if (bin >= (uint32_t)kTdBuckets) {
          atomicOr(error, 1u);
          bin = kTdBuckets - 1;
        }
      } else {
        bin = recs[idx[i]].c[p.axis] < p.splitpoint ? 0u : 1u;  // :366-381
      }
    }
    bins[i] = (uint8_t)bin;
  }
}

// `before[i]` = inclusive counts per bucket over the positions lo..i of i's range.  Writes the element to its place in the
// stable order by bucket and gives every (range, bucket) class a key of its own: the position at which it starts.
__global__ void __launch_bounds__(256) k_td_scatter(int32_t n, const uint32_t *idx, const uint32_t *seg_of, TdSegments sg, const uint8_t *bins,
                                                     const Cnt16 *before, uint32_t *idx_out, uint32_t *class_key) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t s = seg_of[i];
    if (!sg.active[s]) {
      idx_out[i] = idx[i];
      class_key[i] = (uint32_t)i;
      continue;
    }
    const uint32_t lo = sg.lo[s], b = bins[i];
    const Cnt16 &total = before[lo + sg.cnt[s] - 1u];
    uint32_t start = lo;
    for (uint32_t k = 0; k < b; k++) start += total.c[k];
    const uint32_t dest = start + before[i].c[b] - 1u;
    idx_out[dest] = idx[i];
    class_key[dest] = start;
  }
}

struct TdTree {
  uint32_t *child_l, *child_r;
};
RT_DEV uint32_t td_child(const TdTree &t, uint32_t lo, uint32_t cnt, uint32_t node) {  // ref of the child over [lo, lo + cnt); a pair is complete at once
  if (cnt == 1u) return kChildLeaf | lo;
  if (cnt == 2u) {  // bvh_node_sah.rs:283-321
    t.child_l[node] = kChildLeaf | lo;
    t.child_r[node] = kChildLeaf | (lo + 1u);
  }
  return node;
}

// Buckets::split_objects (bvh_node_sah.rs:28-79) for every active range; `class_box[p]` = union over the positions of p's
// class up to p (so a class's last position holds the class's boxes).  Per range: n_left, and the children's centroid boxes.
__global__ void __launch_bounds__(256) k_td_split(uint32_t n_seg, int sah, TdSegments sg, const Cnt16 *before, const Box12 *class_box, TdTree tree,
                                                   uint32_t *n_left_out, double *child_cb, uint32_t *new_count, uint32_t *error) {
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_seg; s += gridDim.x * blockDim.x) {
    if (!sg.active[s]) {
      new_count[s] = 1u;
      n_left_out[s] = 0u;
      continue;
    }
    const uint32_t lo = sg.lo[s], cnt = sg.cnt[s], node = sg.node[s];
    const TdPlan p = td_plan(sg.cb + 6 * (size_t)s, sah != 0);
    const Cnt16 total = before[lo + cnt - 1u];
    double area[kTdBuckets];
    uint32_t end = lo;
    uint32_t last[kTdBuckets];
    for (int i = 0; i < kTdBuckets; i++) {
      area[i] = 0.0;
      last[i] = 0u;
      if (i < p.n_bins && total.c[i]) {
        end += total.c[i];
        last[i] = end - 1u;
        const Box12 &b = class_box[last[i]];
        const double x = b.bmx[0] - b.bmn[0], y = b.bmx[1] - b.bmn[1], z = b.bmx[2] - b.bmn[2];
        area[i] = x * y + y * z + z * x;  // aabb.rs:115-119
      }
    }
    int best_split = 1;
    if (p.n_bins == kTdBuckets) {
      best_split = 0;
      double best_cost = INFINITY;
      for (int split = 1; split < kTdBuckets; split++) {
        double left_area = 0.0, left_count = 0.0, right_area = 0.0, right_count = 0.0;
        for (int i = 0; i < kTdBuckets; i++) {
          if (!total.c[i]) continue;
          if (i < split) {
            left_area += area[i];
            left_count += (double)total.c[i];
          } else {
            right_area += area[i];
            right_count += (double)total.c[i];
          }
        }
        const double cost = left_area * left_count + right_area * right_count;
        if (cost < best_cost) {
          best_split = split;
          best_cost = cost;
        }
      }
    }
    uint32_t n_left = 0;
    double lcb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}, rcb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < p.n_bins; i++) {
      if (!total.c[i]) continue;
      const Box12 &b = class_box[last[i]];
      double *cb = i < best_split ? lcb : rcb;
      if (i < best_split) n_left += total.c[i];
      for (int a = 0; a < 3; a++) cb[a] = fmin(cb[a], b.cmn[a]), cb[3 + a] = fmax(cb[3 + a], b.cmx[a]);
    }
    if (n_left == 0u || n_left == cnt) {  // cannot happen: the first and the last bucket are never empty
      atomicOr(error, 2u);
      n_left = cnt / 2u;
    }
    tree.child_l[node] = td_child(tree, lo, n_left, node + 1u);
    tree.child_r[node] = td_child(tree, lo + n_left, cnt - n_left, node + n_left);
    n_left_out[s] = n_left;
    for (int a = 0; a < 6; a++) child_cb[12 * (size_t)s + a] = lcb[a], child_cb[12 * (size_t)s + 6 + a] = rcb[a];
    new_count[s] = 2u;
  }
}

// The next level's table: an active range becomes its two children, the others stay.  counts[0] += active ranges made.
__global__ void __launch_bounds__(256) k_td_segments(uint32_t n_seg, TdSegments sg, const uint32_t *n_left, const double *child_cb, const uint32_t *base,
                                                      TdSegments out, uint32_t *counts) {
  uint32_t made = 0;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_seg; s += gridDim.x * blockDim.x) {
    const uint32_t at = base[s], lo = sg.lo[s], cnt = sg.cnt[s], node = sg.node[s];
    if (s == n_seg - 1u) counts[1] = at + (sg.active[s] ? 2u : 1u);  // ranges in the new table
    if (!sg.active[s]) {
      out.lo[at] = lo; out.cnt[at] = cnt; out.node[at] = node; out.active[at] = 0u;
      continue;
    }
    const uint32_t nl = n_left[s];
    out.lo[at] = lo; out.cnt[at] = nl; out.node[at] = node + 1u; out.active[at] = nl >= 3u ? 1u : 0u;
    out.lo[at + 1] = lo + nl; out.cnt[at + 1] = cnt - nl; out.node[at + 1] = node + nl; out.active[at + 1] = cnt - nl >= 3u ? 1u : 0u;
    for (int a = 0; a < 6; a++) out.cb[6 * (size_t)at + a] = child_cb[12 * (size_t)s + a], out.cb[6 * (size_t)(at + 1) + a] = child_cb[12 * (size_t)s + 6 + a];
    made += (nl >= 3u ? 1u : 0u) + (cnt - nl >= 3u ? 1u : 0u);
  }
  for (int off = 32; off > 0; off >>= 1) made += __shfl_xor(made, off);
  if ((threadIdx.x & 63) == 0 && made) atomicAdd(&counts[0], made);
}

__global__ void __launch_bounds__(256) k_td_seg_of(int32_t n, const uint32_t *seg_of, TdSegments sg, const uint32_t *n_left, const uint32_t *base,
                                                    uint32_t *seg_of_out) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t s = seg_of[i];
    seg_of_out[i] = base[s] + ((sg.active[s] && (uint32_t)i >= sg.lo[s] + n_left[s]) ? 1u : 0u);
  }
}

// the root range: every element, node 0; its centroid box is the last entry of a scan over all elements under one key
__global__ void k_td_root(int32_t n, const Box12 *all, TdSegments sg, TdTree tree) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sg.lo[0] = 0u;
    sg.cnt[0] = (uint32_t)n;
    sg.node[0] = 0u;
    sg.active[0] = n >= 3 ? 1u : 0u;
    const Box12 &b = all[n - 1];
    for (int a = 0; a < 3; a++) sg.cb[a] = b.cmn[a], sg.cb[3 + a] = b.cmx[a];
    if (n == 2) td_child(tree, 0u, 2u, 0u);
  }
}

}  // namespace build
