// BVH construction on the host, restating BVH_SAH::build / BVHNode_SAH::build
// (/root/reference/src/bvh/bvh_node_sah.rs:147-159, 256-461) and
// Buckets::split_objects (:28-79) so that the node vector -- preorder, one
// element per leaf, 2N-1 nodes -- and therefore the traversal order and every
// tie-break of the hot path are those of the reference.
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <exception>
#include <limits>
#include <stdexcept>
#include <system_error>
#include <thread>

#include "scene.hpp"

namespace rt {

// Worker threads for the host-side preparation: RT_HOST_THREADS, else OMP_NUM_THREADS, else the
// hardware concurrency, never more than 64.
int host_threads() {
  for (const char *name : {"RT_HOST_THREADS", "OMP_NUM_THREADS"}) {
    if (const char *v = std::getenv(name)) {
      int n = std::atoi(v);
      if (n >= 1) return n > 64 ? 64 : n;
    }
  }
  unsigned hw = std::thread::hardware_concurrency();
  return hw == 0 ? 1 : (hw > 64 ? 64 : (int)hw);
}

namespace {

constexpr double kEpsilon = std::numeric_limits<double>::epsilon();  // f64::EPSILON
constexpr int kBuckets = 16;                                         // bvh_node_sah.rs:385
constexpr size_t kParallelSubtree = 8192;  // smallest subtree worth a thread of its own
constexpr size_t kParallelPass = 1 << 17;  // smallest node whose element passes are split over threads

// What the build reads of an element, packed: its box and the box centroid.
struct Rec {
  double mn[3], mx[3], c[3];
};

struct Box {
  double mn[3], mx[3];
  void reset() {
    for (int a = 0; a < 3; a++) mn[a] = std::numeric_limits<double>::infinity(), mx[a] = -std::numeric_limits<double>::infinity();
  }
  // Aabb::merge (aabb.rs:81-97) is min/max per axis -- exact and order-free, so a box of many
  // elements may be accumulated in any grouping.  fmin/fmax drop a NaN like f64::min/max.
  void grow(const double *lo, const double *hi) {
    for (int a = 0; a < 3; a++) mn[a] = std::fmin(mn[a], lo[a]), mx[a] = std::fmax(mx[a], hi[a]);
  }
  double half_area() const {  // aabb.rs:115-119
    double x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2];
    return x * y + y * z + z * x;
  }
};

// Runs fn(chunk) for chunk = 0..n_chunks-1, one thread each (the caller takes the last one).
template <typename F>
void run_chunks(int n_chunks, F &&fn) {
  if (n_chunks <= 1) {
    fn(0);
    return;
  }
  std::vector<std::thread> pool;
  std::vector<std::exception_ptr> failures((size_t)n_chunks);
  auto chunk = [&](int c) {
    try {
      fn(c);
    } catch (...) {
      failures[(size_t)c] = std::current_exception();
    }
  };
  pool.reserve((size_t)n_chunks);
  for (int c = 0; c + 1 < n_chunks; c++) {
    try {
      pool.emplace_back(chunk, c);
    } catch (const std::system_error &) {  // no more threads to be had: the caller does this chunk itself
      chunk(c);
    }
  }
  chunk(n_chunks - 1);
  for (auto &t : pool) t.join();
  for (auto &f : failures)
    if (f) std::rethrow_exception(f);
}

// A subtree over n elements always has 2n-1 nodes (one element per leaf), so the position of
// every node in the reference's preorder vector is known before the subtree is built: a node at
// `pos` has its left subtree at pos+1 and its right subtree at pos+2*n_left.  That makes the
// recursion order-free: large left subtrees are built on their own threads, and the element
// passes of the largest nodes are split over threads as well.  The element list of a node is
// the range [lo, hi) of one shared index array; a split is a stable counting sort of that range
// by bucket, which yields exactly the reference's concatenation of its per-bucket vectors
// (bvh_node_sah.rs:68-74, 366-381).
struct Builder {
  const std::vector<Rec> &recs;
  std::vector<RtNode> &nodes;
  std::vector<int32_t> &idx, &tmp;
  std::vector<uint8_t> &bins;
  SplitMethod method;
  int threads;
  std::atomic<int> max_depth{0};
  std::atomic<int> spare_threads{0};

  Builder(const std::vector<Rec> &r, std::vector<RtNode> &n, std::vector<int32_t> &i, std::vector<int32_t> &t, std::vector<uint8_t> &b,
          SplitMethod m, int th)
      : recs(r), nodes(n), idx(i), tmp(t), bins(b), method(m), threads(th) {
    spare_threads = th - 1;
  }

  void note_depth(int depth) {
    int seen = max_depth.load(std::memory_order_relaxed);
    while (depth > seen && !max_depth.compare_exchange_weak(seen, depth, std::memory_order_relaxed)) {
    }
  }

  static void store_boxes(RtNode &n, const double *lmn, const double *lmx, const double *rmn, const double *rmx) {
    for (int a = 0; a < 3; a++) {
      n.child_l_min[a] = lmn[a];
      n.child_l_max[a] = lmx[a];
      n.child_r_min[a] = rmn[a];
      n.child_r_max[a] = rmx[a];
    }
  }

  void put_leaf(size_t pos, int32_t element) {
    RtNode n{};
    n.is_leaf = 1;
    n.element_index = element;
    nodes[pos] = n;
  }

  // How many threads the passes over a node of n elements may use.
  int pass_threads(size_t n) const {
    if (n < kParallelPass || threads <= 1) return 1;
    size_t share = (size_t)threads * n / recs.size() + 1;
    return (int)std::min<size_t>({share, (size_t)threads, n / (kParallelPass / 4)});
  }

  // Builds the subtree over idx[lo, hi) into nodes[pos, pos + 2*(hi-lo) - 1).
  void build(size_t lo, size_t hi, size_t pos, int depth) {
    note_depth(depth);
    const size_t n = hi - lo;
    if (n == 1) {
      put_leaf(pos, idx[lo]);
      return;
    }
    if (n == 2) {
      RtNode node{};
      node.child_l_index = (int32_t)pos + 1;
      node.child_r_index = (int32_t)pos + 2;
      const Rec &l = recs[(size_t)idx[lo]], &r = recs[(size_t)idx[lo + 1]];
      store_boxes(node, l.mn, l.mx, r.mn, r.mx);
      nodes[pos] = node;
      put_leaf(pos + 1, idx[lo]);
      put_leaf(pos + 2, idx[lo + 1]);
      note_depth(depth + 1);
      return;
    }
    const int T = pass_threads(n);
    auto chunk_lo = [&](int c) { return lo + n * (size_t)c / (size_t)T; };

    // box of the element-box centroids (bvh_node_sah.rs:329-340)
    Box cb;
    cb.reset();
    {
      Box one_part;
      std::vector<Box> many_parts(T > 1 ? (size_t)T : 0);  // heap only when the pass is shared
      Box *part = T > 1 ? many_parts.data() : &one_part;
      run_chunks(T, [&](int c) {
        Box b;
        b.reset();
        for (size_t k = chunk_lo(c), e = chunk_lo(c + 1); k < e; k++) {
          const Rec &r = recs[(size_t)idx[k]];
          b.grow(r.c, r.c);
        }
        part[(size_t)c] = b;
      });
      for (int c = 0; c < T; c++) cb.grow(part[c].mn, part[c].mx);
    }
    // Aabb::largest_axis, aabb.rs:165-175
    const double sx = cb.mx[0] - cb.mn[0], sy = cb.mx[1] - cb.mn[1], sz = cb.mx[2] - cb.mn[2];
    const int axis = (sx > sy && sx > sz) ? 0 : (sy > sz ? 1 : 2);
    const double extent = cb.mx[axis] - cb.mn[axis];

    size_t n_left;
    Box lbox, rbox;
    lbox.reset();
    rbox.reset();
    if (extent < kEpsilon) {
      // all centroids coincide on the longest axis: halves by position (bvh_node_sah.rs:351-362)
      n_left = n / 2;
      for (size_t k = lo; k < hi; k++) {
        const Rec &r = recs[(size_t)idx[k]];
        (k < lo + n_left ? lbox : rbox).grow(r.mn, r.mx);
      }
    } else {
      const bool sah = method == SplitMethod::SAH;
      const int n_bins = sah ? kBuckets : 2;
      const double k0 = cb.mn[axis];
      const double k1 = ((double)kBuckets * (1.0 - 0.0001)) / (cb.mx[axis] - cb.mn[axis]);
      const double splitpoint = cb.mn[axis] + extent / 2.0;
      // pass 1: bucket of every element, per-chunk bucket counts and boxes
      std::array<size_t, kBuckets> one_count, one_offset;
      std::array<Box, kBuckets> one_boxes;
      std::vector<std::array<size_t, kBuckets>> many_counts(T > 1 ? (size_t)T : 0), many_offsets(T > 1 ? (size_t)T : 0);
      std::vector<std::array<Box, kBuckets>> many_boxes(T > 1 ? (size_t)T : 0);
      std::array<size_t, kBuckets> *count = T > 1 ? many_counts.data() : &one_count;
      std::array<size_t, kBuckets> *offset = T > 1 ? many_offsets.data() : &one_offset;
      std::array<Box, kBuckets> *boxes = T > 1 ? many_boxes.data() : &one_boxes;
      run_chunks(T, [&](int c) {
        auto &cnt = count[(size_t)c];
        auto &bx = boxes[(size_t)c];
        cnt.fill(0);
        for (auto &b : bx) b.reset();
        for (size_t k = chunk_lo(c), e = chunk_lo(c + 1); k < e; k++) {
          const Rec &r = recs[(size_t)idx[k]];
          size_t bin;
          if (sah) {
            double b = k1 * (r.c[axis] - k0);
            // `as usize`: truncation, saturating, NaN -> 0
            bin = (b != b || b <= 0.0) ? 0 : (b >= 1e18 ? (size_t)1e18 : (size_t)b);
            if (bin >= (size_t)kBuckets) throw std::runtime_error("BVH build: bucket index out of range");
          } else {
            bin = r.c[axis] < splitpoint ? 0 : 1;  // bvh_node_sah.rs:366-381
          }
          bins[k] = (uint8_t)bin;
          cnt[bin]++;
          bx[bin].grow(r.mn, r.mx);
        }
      });
      std::array<size_t, kBuckets> total{};
      std::array<Box, kBuckets> bbox;
      for (auto &b : bbox) b.reset();
      for (int c = 0; c < T; c++)
        for (int i = 0; i < n_bins; i++) {
          total[(size_t)i] += count[(size_t)c][(size_t)i];
          if (count[(size_t)c][(size_t)i]) bbox[(size_t)i].grow(boxes[(size_t)c][(size_t)i].mn, boxes[(size_t)c][(size_t)i].mx);
        }
      int best_split = 1;
      if (sah) {
        // Buckets::split_objects (bvh_node_sah.rs:28-79): cost of a plane = (sum of the per-bucket
        // box areas on the left) * count_left + (same on the right) * count_right, summed in bucket
        // order; the first strictly cheaper plane wins.
        double area[kBuckets];
        for (int i = 0; i < kBuckets; i++) area[i] = total[(size_t)i] ? bbox[(size_t)i].half_area() : 0.0;
        best_split = 0;
        double best_cost = std::numeric_limits<double>::infinity();
        for (int split = 1; split < kBuckets; split++) {
          double left_area = 0.0, left_count = 0.0, right_area = 0.0, right_count = 0.0;
          for (int i = 0; i < kBuckets; i++) {
            if (!total[(size_t)i]) continue;
            if (i < split) {
              left_area += area[i];
              left_count += (double)total[(size_t)i];
            } else {
              right_area += area[i];
              right_count += (double)total[(size_t)i];
            }
          }
          double cost = left_area * left_count + right_area * right_count;
          if (cost < best_cost) {
            best_split = split;
            best_cost = cost;
          }
        }
      }
      n_left = 0;
      for (int i = 0; i < n_bins; i++) {
        if (!total[(size_t)i]) continue;
        if (i < best_split) {
          n_left += total[(size_t)i];
          lbox.grow(bbox[(size_t)i].mn, bbox[(size_t)i].mx);
        } else {
          rbox.grow(bbox[(size_t)i].mn, bbox[(size_t)i].mx);
        }
      }
      // pass 2: stable scatter by bucket (chunk c's share of bucket i follows chunk c-1's)
      size_t at = lo;
      for (int i = 0; i < n_bins; i++)
        for (int c = 0; c < T; c++) {
          offset[(size_t)c][(size_t)i] = at;
          at += count[(size_t)c][(size_t)i];
        }
      run_chunks(T, [&](int c) {
        auto off = offset[(size_t)c];
        for (size_t k = chunk_lo(c), e = chunk_lo(c + 1); k < e; k++) tmp[off[bins[k]]++] = idx[k];
      });
      run_chunks(T, [&](int c) { std::copy(tmp.begin() + (long)chunk_lo(c), tmp.begin() + (long)chunk_lo(c + 1), idx.begin() + (long)chunk_lo(c)); });
    }
    if (n_left == 0 || n_left == n) throw std::runtime_error("BVH build: empty child");

    RtNode node{};
    node.child_l_index = (int32_t)pos + 1;
    node.child_r_index = (int32_t)(pos + 2 * n_left);
    store_boxes(node, lbox.mn, lbox.mx, rbox.mn, rbox.mx);  // bvh_node_sah.rs:444-445
    nodes[pos] = node;

    const size_t mid = lo + n_left, right_pos = pos + 2 * n_left;
    if (n_left >= kParallelSubtree && spare_threads.fetch_sub(1, std::memory_order_relaxed) > 0) {
      std::exception_ptr failure;
      auto left = [&, this] {
        try {
          build(lo, mid, pos + 1, depth + 1);
        } catch (...) {
          failure = std::current_exception();
        }
      };
      std::thread worker;
      try {
        worker = std::thread(left);
      } catch (const std::system_error &) {  // no more threads to be had: both halves on this one
        left();
      }
      std::exception_ptr mine;
      try {
        build(mid, hi, right_pos, depth + 1);
      } catch (...) {
        mine = std::current_exception();
      }
      if (worker.joinable()) worker.join();
      spare_threads.fetch_add(1, std::memory_order_relaxed);
      if (mine) std::rethrow_exception(mine);
      if (failure) std::rethrow_exception(failure);
    } else {
      if (n_left >= kParallelSubtree) spare_threads.fetch_add(1, std::memory_order_relaxed);  // undo the probe
      build(lo, mid, pos + 1, depth + 1);
      build(mid, hi, right_pos, depth + 1);
    }
  }
};

}  // namespace

void build_bvh(SceneData &scene, SplitMethod method) {
  const size_t n = scene.elements.size();
  if (n == 0) throw std::runtime_error("BVH build: no elements");
  if (n > (size_t)0x3fffffff) throw std::runtime_error("BVH build: too many elements");
  std::vector<Rec> recs(n);
  for (size_t i = 0; i < n; i++) {
    const Aabb &b = scene.elements[i].bbox;
    recs[i] = Rec{{b.min.x, b.min.y, b.min.z}, {b.max.x, b.max.y, b.max.z}, {b.centroid.x, b.centroid.y, b.centroid.z}};
  }
  scene.nodes.assign(2 * n - 1, RtNode{});
  std::vector<int32_t> idx(n), tmp(n);
  std::vector<uint8_t> bins(n);
  for (size_t i = 0; i < n; i++) idx[i] = (int32_t)i;  // bvh_node_sah.rs:149
  Builder b(recs, scene.nodes, idx, tmp, bins, method, host_threads());
  b.build(0, n, 0, 0);
  scene.bvh_depth = b.max_depth;
}

}  // namespace rt
