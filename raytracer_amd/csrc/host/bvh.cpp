// BVH construction on the host, restating BVH_SAH::build / BVHNode_SAH::build
// (/root/reference/src/bvh/bvh_node_sah.rs:147-159, 256-461) and
// Buckets::split_objects (:28-79) so that the node vector -- preorder, one
// element per leaf, 2N-1 nodes -- and therefore the traversal order and every
// tie-break of the hot path are those of the reference.
#include <array>
#include <limits>
#include <stdexcept>

#include "scene.hpp"

namespace rt {
namespace {

constexpr double kEpsilon = std::numeric_limits<double>::epsilon();  // f64::EPSILON
constexpr int kBuckets = 16;                                         // bvh_node_sah.rs:385

struct Builder {
  const std::vector<Element> &objects;
  std::vector<RtNode> &nodes;
  SplitMethod method;
  int max_depth = 0;

  // Aabb::new_from_objects, aabb.rs:99-112
  Aabb bounds_of(const std::vector<int32_t> &indices) const {
    Aabb box = objects[indices[0]].bbox;
    if (indices.size() > 1)
      for (int32_t i : indices) box = Aabb::merge(box, objects[i].bbox);
    return box;
  }

  static void store_boxes(RtNode &n, const Aabb &l, const Aabb &r) {
    n.child_l_min[0] = l.min.x; n.child_l_min[1] = l.min.y; n.child_l_min[2] = l.min.z;
    n.child_l_max[0] = l.max.x; n.child_l_max[1] = l.max.y; n.child_l_max[2] = l.max.z;
    n.child_r_min[0] = r.min.x; n.child_r_min[1] = r.min.y; n.child_r_min[2] = r.min.z;
    n.child_r_max[0] = r.max.x; n.child_r_max[1] = r.max.y; n.child_r_max[2] = r.max.z;
  }

  int32_t push_leaf(int32_t element) {
    RtNode n{};
    n.is_leaf = 1;
    n.element_index = element;
    nodes.push_back(n);
    return (int32_t)nodes.size() - 1;
  }

  // Buckets::split_objects: cost of a plane = (sum of per-bucket box areas on
  // the left) * count_left + (same on the right) * count_right; the first
  // strictly cheaper plane wins.  A bucket's box does not depend on the plane,
  // so it is computed once; the sums keep the reference's order.
  void sah_split(const std::array<std::vector<int32_t>, kBuckets> &buckets, std::vector<int32_t> &left,
                 std::vector<int32_t> &right) const {
    double area[kBuckets];
    for (int i = 0; i < kBuckets; i++) area[i] = buckets[i].empty() ? 0.0 : bounds_of(buckets[i]).area();
    int best_split = 0;
    double best_cost = std::numeric_limits<double>::infinity();
    for (int split = 1; split < kBuckets; split++) {
      double left_area = 0.0, left_count = 0.0, right_area = 0.0, right_count = 0.0;
      for (int i = 0; i < kBuckets; i++) {
        if (buckets[i].empty()) continue;
        if (i < split) {
          left_area += area[i];
          left_count += (double)buckets[i].size();
        } else {
          right_area += area[i];
          right_count += (double)buckets[i].size();
        }
      }
      double cost = left_area * left_count + right_area * right_count;
      if (cost < best_cost) {
        best_split = split;
        best_cost = cost;
      }
    }
    for (int i = 0; i < kBuckets; i++) {
      auto &dst = (i < best_split) ? left : right;
      dst.insert(dst.end(), buckets[i].begin(), buckets[i].end());
    }
  }

  int32_t build(const std::vector<int32_t> &indices, int depth) {
    if (depth > max_depth) max_depth = depth;
    if (indices.empty()) return 0;
    if (indices.size() == 1) return push_leaf(indices[0]);
    if (indices.size() == 2) {
      int32_t self = (int32_t)nodes.size();
      RtNode n{};
      n.child_l_index = self + 1;
      n.child_r_index = self + 2;
      store_boxes(n, objects[indices[0]].bbox, objects[indices[1]].bbox);
      nodes.push_back(n);
      push_leaf(indices[0]);
      push_leaf(indices[1]);
      if (depth + 1 > max_depth) max_depth = depth + 1;
      return self;
    }

    int32_t self = (int32_t)nodes.size();
    nodes.push_back(RtNode{});  // placeholder, filled once both subtrees exist

    // box of the element-box centroids (bvh_node_sah.rs:329-340)
    Aabb cb = Aabb::from_points(objects[indices[0]].bbox.centroid, objects[indices[1]].bbox.centroid);
    for (int32_t i : indices) cb = Aabb::merge(cb, Aabb::from_point(objects[i].bbox.centroid));

    std::vector<int32_t> left, right;
    left.reserve(indices.size() / 2);
    right.reserve(indices.size() / 2);
    int axis = cb.largest_axis();
    double extent = cb.max[axis] - cb.min[axis];

    if (extent < kEpsilon) {
      size_t mid = indices.size() / 2;
      left.assign(indices.begin(), indices.begin() + (long)mid);
      right.assign(indices.begin() + (long)mid, indices.end());
    } else if (method == SplitMethod::Mid) {
      double splitpoint = cb.min[axis] + extent / 2.0;
      for (int32_t i : indices) (objects[i].bbox.centroid[axis] < splitpoint ? left : right).push_back(i);
    } else {
      std::array<std::vector<int32_t>, kBuckets> buckets;
      double k0 = cb.min[axis];
      double k1 = ((double)kBuckets * (1.0 - 0.0001)) / (cb.max[axis] - cb.min[axis]);
      for (int32_t i : indices) {
        double b = k1 * (objects[i].bbox.centroid[axis] - k0);
        // `as usize`: truncation, saturating, NaN -> 0
        size_t bin = (b != b || b <= 0.0) ? 0 : (b >= 1e18 ? (size_t)1e18 : (size_t)b);
        if (bin >= (size_t)kBuckets) throw std::runtime_error("BVH build: bucket index out of range");
        buckets[bin].push_back(i);
      }
      sah_split(buckets, left, right);
    }
    if (left.empty() || right.empty()) throw std::runtime_error("BVH build: empty child");

    int32_t l = build(left, depth + 1);
    int32_t r = build(right, depth + 1);
    RtNode n{};
    n.child_l_index = l;
    n.child_r_index = r;
    store_boxes(n, bounds_of(left), bounds_of(right));
    nodes[self] = n;
    return self;
  }
};

}  // namespace

void build_bvh(SceneData &scene, SplitMethod method) {
  scene.nodes.clear();
  scene.nodes.reserve(scene.elements.size() * 2);
  std::vector<int32_t> indices(scene.elements.size());
  for (size_t i = 0; i < indices.size(); i++) indices[i] = (int32_t)i;
  Builder b{scene.elements, scene.nodes, method};
  b.build(indices, 0);
  scene.bvh_depth = b.max_depth;
}

}  // namespace rt
