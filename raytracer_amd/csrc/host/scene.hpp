// Host-side scene model: what the reference's Rust host builds before the hot
// loop (integrator.rs:163-192).  Everything here runs once per render on the
// CPU; the per-ray work lives behind rt_hip.h.
#pragma once

#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/rt_hip.h"
#include "json.hpp"

namespace rt {

// ---------------------------------------------------------------- Vec3
// linalg/vec3.rs:11-15; arithmetic keeps the reference's operation order.
struct Vec3 {
  double x = 0, y = 0, z = 0;
  Vec3() = default;
  Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
  double operator[](int a) const { return a == 0 ? x : (a == 1 ? y : z); }  // vec3.rs:196-208
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, double q) { return {a.x * q, a.y * q, a.z * q}; }
inline Vec3 operator/(Vec3 a, double q) { return {a.x / q, a.y / q, a.z / q}; }
inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double length(Vec3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // vec3.rs:116-128
inline Vec3 normalized(Vec3 a) {                                                       // vec3.rs:163-166
  double l = length(a);
  return {a.x / l, a.y / l, a.z / l};
}
inline Vec3 vmin(Vec3 a, Vec3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline Vec3 vmax(Vec3 a, Vec3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }
inline RtVec3 abi(Vec3 a) { return RtVec3{a.x, a.y, a.z}; }

// ---------------------------------------------------------------- Aabb
// bvh/aabb.rs:13-20.  The x/y/z intervals of the reference duplicate min/max,
// so only min/max/centroid are kept.
struct Aabb {
  Vec3 min, max, centroid;

  static Aabb from_points(Vec3 p, Vec3 q);  // aabb.rs:59-75
  static Aabb from_point(Vec3 p) { return from_points(p, p); }
  static Aabb merge(const Aabb &a, const Aabb &b);  // aabb.rs:81-97
  Aabb pad() const;                                 // aabb.rs:122-152
  double area() const;                              // aabb.rs:115-119 (half the surface area)
  int largest_axis() const;                         // aabb.rs:165-175
};

// -------------------------------------------------------------- Config
enum class SplitMethod { Mid, SAH };  // bvh_node_sah.rs:13-19

struct Config {  // render/config.rs:8-20
  double img_width = 0, img_height = 0;
  int64_t sample_batch_size = 0, max_sample_batches = 0, min_sample_batches = 0, max_depth = 0;
  double sky_color[3] = {0, 0, 0};  // parsed, never used (integrator.rs:465-468)
  double pixel_radius = 0;
  bool has_split_method = false;
  SplitMethod split_method = SplitMethod::Mid;
  double gamma_correction = 0;

  static Config from_json(const Json &j);
};

struct JsonCamera {  // camera.rs:6-13
  Vec3 center, look_at;
  double fov_vertical = 0, defocus_angle = 0, focus_dist = 0;
};

// ------------------------------------------------------------- Elements
// One entry of the reference's Vec<Element> (elements/mod.rs:59-63) with the
// bounding box its constructor computes.
struct Element {
  int kind = RT_PRIM_TRIANGLE;
  int index = 0;  // slot in the per-kind array
  Aabb bbox;
};

struct SceneData {
  RtCamera camera{};
  std::vector<Element> elements;
  std::vector<RtTriangle> triangles;
  std::vector<RtQuad> quads;
  std::vector<RtSphere> spheres;
  std::vector<RtMaterial> materials;
  std::vector<int32_t> attractors;
  std::vector<RtNode> nodes;
  std::vector<RtElementRef> element_refs;
  int bvh_depth = 0;

  void add_triangle(Vec3 v0, Vec3 v1, Vec3 v2, int material);                 // triangle.rs:55-70
  void add_quad(Vec3 q, Vec3 u, Vec3 v, int material, bool attractor);        // quad.rs:69-89
  void add_sphere(Vec3 center, double radius, int material, bool attractor);  // sphere.rs:72-83
};

// camera.rs:35-91
RtCamera make_camera(const Config &cfg, const JsonCamera &jc);

// integrator.rs:167-177 + elements/{triangle,quad,sphere,cube,obj}.rs
void expand_elements(const Json &scene, const std::string &base_dir, SceneData &out);

// bvh_node_sah.rs:147-159, 256-461
void build_bvh(SceneData &scene, SplitMethod method);
int host_threads();

}  // namespace rt
