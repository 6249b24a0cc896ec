// The reference's own tree, built on the device (rt_hip_scene_build, the default method): BVHNode_SAH::build
// (/root/reference/src/bvh/bvh_node_sah.rs:256-461) is a top-down recursion -- box of the element-box centroids,
// its longest axis, sixteen buckets along it (or the midpoint, BVHSplitMethod::Mid), Buckets::split_objects (:28-79),
// children from the concatenated buckets -- in which the children of a node depend on nothing but the node's own element
// list.  All nodes of one depth are therefore split together, one LEVEL per trip of a host loop:
//
//   k_td_count        the bucket of every element of every node of the level; per 64 positions, elements per bucket
//   k_td_totals       per node, its elements per bucket
//   k_td_scatter      the stable counting sort of every node's range by bucket = the concatenation of the reference's
//                     per-bucket vectors (:68-74), so the element order inside a child is the reference's
//   k_td_class_boxes  per (node, bucket), the union of the element boxes and of their centroids
//   k_td_split        per node: bucket areas, the fifteen plane costs summed in the reference's order, the first cheapest
//                     plane, the two children (leaves, two-element nodes, or ranges for the next level)
//   k_td_segments     the next level's table of ranges
// (the element passes are wave-wide ballots and shuffles over 64 positions each; what crosses those chunks goes through
// two small items per chunk and a library scan by key over the items)
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

// ---- per level, the element passes.  One wave takes 64 consecutive positions (a "chunk").  Ranges are runs of positions, so
// the lanes of one range form a run of lanes; what a range needs from beyond the chunk arrives through two items per chunk --
// one for the run its first lane is in, one for the run its last lane is in (a chunk inside one range: the first item holds
// everything, the second nothing) -- and an exclusive scan by key over those items (64 times fewer than elements).
RT_DEV uint32_t td_bin_of(const TdSegments &sg, uint32_t s, uint32_t p, int sah, const Rec64 *recs, const uint32_t *idx, uint32_t *error) {
  const TdPlan plan = td_plan(sg.cb + 6 * (size_t)s, sah != 0);
  if (plan.halves) return p - sg.lo[s] < sg.cnt[s] / 2u ? 0u : 1u;
  const double c = recs[idx[p]].c[plan.axis];
  if (!sah) return c < plan.splitpoint ? 0u : 1u;  // :366-381
  const double b = plan.k1 * (c - plan.k0);
  // `as usize`: truncation, saturating, NaN -> 0
  uint32_t bin = (b != b || b <= 0.0) ? 0u : (b >= 4e9 ? 0xffffffffu : (uint32_t)b);
  if (bin >= (uint32_t)kTdBuckets) {
    atomicOr(error, 1u);
    bin = kTdBuckets - 1;
  }
  return bin;
}
struct TdChunk {  // what every lane of a wave knows about its chunk
  uint32_t p, s, s_first, s_last, n_valid;
  bool valid, act;
  unsigned long long head, tail;  // lanes of the first / the last run
  int run_start;                  // first lane of this lane's run
};
RT_DEV TdChunk td_chunk(uint32_t chunk, int32_t n, const uint32_t *seg_of, const TdSegments &sg) {
  TdChunk c;
  const uint32_t lane = threadIdx.x & 63u;
  c.p = chunk * 64u + lane;
  c.valid = c.p < (uint32_t)n;
  c.s = c.valid ? seg_of[c.p] : 0xffffffffu;
  c.act = c.valid && sg.active[c.s] != 0u;
  c.n_valid = min(64u, (uint32_t)n - chunk * 64u);
  c.s_first = (uint32_t)__shfl((int)c.s, 0);
  c.s_last = (uint32_t)__shfl((int)c.s, (int)c.n_valid - 1);
  c.head = __ballot(c.valid && c.s == c.s_first);
  c.tail = __ballot(c.valid && c.s == c.s_last);
  const uint32_t s_prev = (uint32_t)__shfl_up((int)c.s, 1);
  const unsigned long long starts = __ballot(lane == 0u || c.s != s_prev);
  c.run_start = 63 - __clzll((long long)(starts & ((2ull << lane) - 1ull)));
  return c;
}
constexpr int kTdWavesPerBlock = 4;
RT_DEV uint32_t td_wave() { return blockIdx.x * kTdWavesPerBlock + (threadIdx.x >> 6); }

// The bucket of every element (kept in `bins`), and the chunk's two items: elements per bucket in its first and in its last run.
__global__ void __launch_bounds__(256) k_td_count(int32_t n, int sah, const uint32_t *idx, const uint32_t *seg_of, TdSegments sg, const Rec64 *recs,
                                                   uint8_t *bins, uint32_t *item_key, Cnt16 *item_val, uint32_t *error) {
  const uint32_t lane = threadIdx.x & 63u, chunks = ((uint32_t)n + 63u) / 64u;
  for (uint32_t chunk = td_wave(); chunk < chunks; chunk += gridDim.x * kTdWavesPerBlock) {
    const TdChunk c = td_chunk(chunk, n, seg_of, sg);
    uint32_t bin = 0;
    if (c.act) bin = td_bin_of(sg, c.s, c.p, sah, recs, idx, error);
    if (c.valid) bins[c.p] = (uint8_t)bin;
    unsigned long long mine = 0;
#pragma unroll
    for (uint32_t b = 0; b < (uint32_t)kTdBuckets; b++) {
      const unsigned long long m = __ballot(c.act && bin == b);
      if (lane == b) mine = m;
    }
    const bool single = c.s_first == c.s_last;
    if (lane < (uint32_t)kTdBuckets) {
      item_val[2 * (size_t)chunk].c[lane] = (uint32_t)__popcll(mine & c.head);
      item_val[2 * (size_t)chunk + 1].c[lane] = single ? 0u : (uint32_t)__popcll(mine & c.tail);
    }
    if (lane == 0u) {
      item_key[2 * (size_t)chunk] = c.s_first;
      item_key[2 * (size_t)chunk + 1] = c.s_last;
    }
  }
}

// `carry[2 * chunk]` = elements per bucket of the chunk's first run's range in earlier chunks.  The lane that holds the last
// element of an active range adds up the range's elements per bucket.
__global__ void __launch_bounds__(256) k_td_totals(int32_t n, const uint32_t *seg_of, TdSegments sg, const uint8_t *bins, const Cnt16 *carry,
                                                    Cnt16 *seg_total) {
  const uint32_t lane = threadIdx.x & 63u, chunks = ((uint32_t)n + 63u) / 64u;
  for (uint32_t chunk = td_wave(); chunk < chunks; chunk += gridDim.x * kTdWavesPerBlock) {
    const TdChunk c = td_chunk(chunk, n, seg_of, sg);
    const uint32_t bin = c.valid ? bins[c.p] : 0u;
    const bool closes = c.act && c.p == sg.lo[c.s] + sg.cnt[c.s] - 1u;
    const unsigned long long run = ((2ull << lane) - 1ull) & ~((1ull << c.run_start) - 1ull);  // lanes run_start..lane
    Cnt16 t;
#pragma unroll
    for (uint32_t b = 0; b < (uint32_t)kTdBuckets; b++) t.c[b] = (uint32_t)__popcll(__ballot(c.act && bin == b) & run);
    if (closes) {
      if (c.s == c.s_first) {
        const Cnt16 &in = carry[2 * (size_t)chunk];
#pragma unroll
        for (int b = 0; b < kTdBuckets; b++) t.c[b] += in.c[b];
      }
      seg_total[c.s] = t;
    }
  }
}

// The stable counting sort of every active range by bucket: the element's new place = start of its bucket's class + elements of
// the class before it (in earlier chunks: the carry; in this chunk: the lanes before it).  Every class gets a key of its own,
// the position at which it starts; the elements of the other ranges stay and carry no key.
constexpr uint32_t kTdNoClass = 0xffffffffu;
__global__ void __launch_bounds__(256) k_td_scatter(int32_t n, const uint32_t *idx, const uint32_t *seg_of, TdSegments sg, const uint8_t *bins,
                                                     const Cnt16 *carry, const Cnt16 *seg_total, uint32_t *idx_out, uint32_t *class_key) {
  const uint32_t lane = threadIdx.x & 63u, chunks = ((uint32_t)n + 63u) / 64u;
  for (uint32_t chunk = td_wave(); chunk < chunks; chunk += gridDim.x * kTdWavesPerBlock) {
    const TdChunk c = td_chunk(chunk, n, seg_of, sg);
    const uint32_t bin = c.valid ? bins[c.p] : 0u;
    unsigned long long same = 0;
#pragma unroll
    for (uint32_t b = 0; b < (uint32_t)kTdBuckets; b++) {
      const unsigned long long m = __ballot(c.act && bin == b);
      if (bin == b) same = m;
    }
    if (!c.valid) continue;
    if (!c.act) {
      idx_out[c.p] = idx[c.p];
      class_key[c.p] = kTdNoClass;
      continue;
    }
    const unsigned long long before = ((1ull << lane) - 1ull) & ~((1ull << c.run_start) - 1ull);  // lanes run_start..lane-1
    const Cnt16 &total = seg_total[c.s];
    uint32_t start = sg.lo[c.s];
#pragma unroll
    for (uint32_t b = 0; b < (uint32_t)kTdBuckets; b++) start += b < bin ? total.c[b] : 0u;
    uint32_t dest = start + (uint32_t)__popcll(same & before);
    if (c.s == c.s_first) dest += carry[2 * (size_t)chunk].c[bin];
    idx_out[dest] = idx[c.p];
    class_key[dest] = start;
  }
}

RT_DEV Box12 box12_identity() {
  Box12 b;
  for (int k = 0; k < 3; k++) b.bmn[k] = b.cmn[k] = INFINITY, b.bmx[k] = b.cmx[k] = -INFINITY;
  return b;
}
RT_DEV Box12 box12_shfl_up(const Box12 &b, int d) {
  Box12 r;
  for (int k = 0; k < 3; k++) {
    r.bmn[k] = __shfl_up(b.bmn[k], d);
    r.bmx[k] = __shfl_up(b.bmx[k], d);
    r.cmn[k] = __shfl_up(b.cmn[k], d);
    r.cmx[k] = __shfl_up(b.cmx[k], d);
  }
  return r;
}
// After the sort a class is a run of positions.  class_box[p], written by the last lane of every run of a chunk = the union
// over the run's lanes; the chunk's two items = the unions over its first and over its last run.  A class that began in an
// earlier chunk is completed by the reader: class_box[its last position] united with carry[2 * that position's chunk].
__global__ void __launch_bounds__(256) k_td_class_boxes(int32_t n, const uint32_t *idx, const uint32_t *class_key, const Rec64 *recs, Box12 *class_box,
                                                         uint32_t *item_key, Box12 *item_box) {
  const uint32_t lane = threadIdx.x & 63u, chunks = ((uint32_t)n + 63u) / 64u;
  for (uint32_t chunk = td_wave(); chunk < chunks; chunk += gridDim.x * kTdWavesPerBlock) {
    const uint32_t p = chunk * 64u + lane, n_valid = min(64u, (uint32_t)n - chunk * 64u);
    const bool valid = p < (uint32_t)n;
    const uint32_t key = valid ? class_key[p] : kTdNoClass;
    const bool live = key != kTdNoClass;
    Box12 box = box12_identity();
    if (live) box = RecBox12{recs}(idx[p]);
    const uint32_t key_prev = (uint32_t)__shfl_up((int)key, 1), key_next = (uint32_t)__shfl_down((int)key, 1);
    const unsigned long long starts = __ballot(lane == 0u || key != key_prev);
    const int run_start = 63 - __clzll((long long)(starts & ((2ull << lane) - 1ull)));
#pragma unroll
    for (int d = 1; d < 64; d *= 2) {
      const Box12 other = box12_shfl_up(box, d);
      if ((int)lane - d >= run_start) box = Box12Union()(box, other);
    }
    const bool run_last = lane == n_valid - 1u || key != key_next;
    if (live && run_last) class_box[p] = box;
    const uint32_t key_first = (uint32_t)__shfl((int)key, 0), key_last = (uint32_t)__shfl((int)key, (int)n_valid - 1);
    const uint32_t head_last = (uint32_t)__popcll(__ballot(valid && key == key_first && run_start == 0)) - 1u;
    if (lane == head_last) {
      item_key[2 * (size_t)chunk] = key_first;
      item_box[2 * (size_t)chunk] = box;
    }
    if (lane == n_valid - 1u) {
      item_key[2 * (size_t)chunk + 1] = key_last;
      item_box[2 * (size_t)chunk + 1] = run_start == 0 ? box12_identity() : box;  // one run over the whole chunk: the first item holds it
    }
  }
}
RT_DEV Box12 td_class_total(const Box12 *class_box, const Box12 *carry, uint32_t start, uint32_t last) {
  Box12 b = class_box[last];
  if (start < (last & ~63u)) b = Box12Union()(b, carry[2 * (size_t)(last >> 6)]);
  return b;
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

// Buckets::split_objects (bvh_node_sah.rs:28-79) for every active range, from its elements per bucket (seg_total) and its
// classes' boxes (td_class_total).  Per range: n_left, and the children's centroid boxes.
__global__ void __launch_bounds__(256) k_td_split(uint32_t n_seg, int sah, TdSegments sg, const Cnt16 *seg_total, const Box12 *class_box,
                                                   const Box12 *carry_box, TdTree tree, uint32_t *n_left_out, double *child_cb, uint32_t *new_count,
                                                   uint32_t *error) {
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_seg; s += gridDim.x * blockDim.x) {
    if (!sg.active[s]) {
      new_count[s] = 1u;
      n_left_out[s] = 0u;
      continue;
    }
    const uint32_t lo = sg.lo[s], cnt = sg.cnt[s], node = sg.node[s];
    const TdPlan p = td_plan(sg.cb + 6 * (size_t)s, sah != 0);
    const Cnt16 total = seg_total[s];
    double area[kTdBuckets];
    uint32_t end = lo;
    uint32_t last[kTdBuckets];
    for (int i = 0; i < kTdBuckets; i++) {
      area[i] = 0.0;
      last[i] = 0u;
      if (i < p.n_bins && total.c[i]) {
        end += total.c[i];
        last[i] = end - 1u;
        const Box12 b = td_class_total(class_box, carry_box, end - total.c[i], last[i]);
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
      const Box12 b = td_class_total(class_box, carry_box, last[i] + 1u - total.c[i], last[i]);
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

// the root range: every element, node 0; its centroid box comes from k_td_class_boxes over all elements under one key
__global__ void k_td_root(int32_t n, const Box12 *class_box, const Box12 *carry_box, TdSegments sg, TdTree tree) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sg.lo[0] = 0u;
    sg.cnt[0] = (uint32_t)n;
    sg.node[0] = 0u;
    sg.active[0] = n >= 3 ? 1u : 0u;
    const Box12 b = td_class_total(class_box, carry_box, 0u, (uint32_t)n - 1u);
    for (int a = 0; a < 3; a++) sg.cb[a] = b.cmn[a], sg.cb[3 + a] = b.cmx[a];
    if (n == 2) td_child(tree, 0u, 2u, 0u);
  }
}

}  // namespace build
