// Device-side math, sampling and intersection routines of the MI355X path
// tracer (gfx950 only).  All radiometric and geometric arithmetic is f64 and
// keeps the reference's operation order (this translation unit is compiled
// with -ffp-contract=off: the Rust reference never fuses a multiply-add), so
// that a primitive test returns the same t bits as the reference's.  Only the
// BVH child boxes are stored and tested in f32, rounded outward, which can
// add box visits but never removes one.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_hip.h"

#define RT_DEV __device__ __forceinline__
#ifndef RT_PK_FMA
#define RT_PK_FMA 0
#endif

namespace rtd {

constexpr double kEpsilon = 2.220446049250313e-16;  // f64::EPSILON
constexpr double kPi = 3.14159265358979323846;      // std::f64::consts::PI
constexpr double kF64Max = 1.7976931348623157e308;
constexpr double kTMin = 0.0001;  // integrator.rs:359

// ------------------------------------------------------------ child refs
// A child reference / stack entry packs: bit31 leaf, bits30-29 primitive kind
// (RT_PRIM_*), bit28 "box test passed" (stack entries only), bits27-0 index
// (internal-node index, or slot in the per-kind primitive array).
constexpr uint32_t kRefLeaf = 0x80000000u;
constexpr uint32_t kRefFlag = 0x10000000u;
constexpr uint32_t kRefIndexMask = 0x0fffffffu;
constexpr uint32_t kRefNone = 0xffffffffu;
constexpr int kRefKindShift = 29;
constexpr int kMaxStackDepth = 64;
#ifndef RT_SPILL_ABOVE
#define RT_SPILL_ABOVE 22
#endif
#ifndef RT_OVF_LEVELS
#define RT_OVF_LEVELS 16
#endif
constexpr int kLdsStackLevels = 24;               // default of the LDSL argument of the stack functions (the wide walk passes its own)
constexpr int kSpillAboveDepth = RT_SPILL_ABOVE;  // trees up to this height keep the whole stack of the binary order-free walks in LDS: six
                                     // waves per SIMD (24 stacks per CU) have room for 22 levels + the leaf queue each.  Higher trees take the
                                     // block-overflow stack (StackRef below), whose LDS footprint does not depend on the tree.  Measured on a 2 M-triangle
                                     // scene at 2048 x 2048: the reference-order SAH tree (height 26) all in LDS at five waves 1124 ms, overflow stack at
                                     // six waves 1070 ms; the device-built linear BVH (height 38) 1255 ms with the levels above 24 in global memory
                                     // slot by slot at five waves, 1116 ms with the overflow stack.  On a tree of height 21 that has the choice the
                                     // overflow stack is 2 % slower than the whole stack in LDS (650 against 635 ms).

// 64-byte internal node: both child boxes (f32, rounded outward) + child refs.
struct __attribute__((aligned(64))) DevNode {
  float lmin[3], lmax[3], rmin[3], rmax[3];
  uint32_t left, right;
  uint32_t pad[2];
};
static_assert(sizeof(DevNode) == 64, "node must be one 64-byte record");

// 112-byte internal node with the child boxes as the reference holds them (f64).  Uploaded only for trees in
// which some box does not bound its contents: there the result of a query depends on exactly which boxes the
// ray hits (bvh_node_sah.rs:206-250), so the walk that follows the reference's visit rule evaluates
// Aabb::hit (aabb.rs:187-204) on the reference's own numbers instead of a conservative f32 test.
struct __attribute__((aligned(16))) DevNodeD {
  double lmin[3], lmax[3], rmin[3], rmax[3];
  uint32_t left, right;
  uint32_t pad[2];
};
static_assert(sizeof(DevNodeD) == 112, "f64 node record");

// 32-byte internal node: the same two child boxes on a 16-bit grid spanning the scene's
// bounds (min corners rounded down, max corners up), + child refs.  Half the bytes of DevNode:
// the node set of a 100 000-triangle scene (3.2 MB) then fits one XCD's L2.
struct __attribute__((aligned(32))) DevNodeQ {
  uint16_t lmin[3], lmax[3], rmin[3], rmax[3];
  uint32_t left, right;
};
static_assert(sizeof(DevNodeQ) == 32, "quantised node must be one 32-byte record");

// 64-byte wide node: up to four child boxes on the same 16-bit grid, stored plane by plane (word c of `px`
// holds child c's x planes, min in the low half, max in the high half), + four child refs.  Made at upload by
// pulling grandchildren up into their grandparent (the child with the largest box is opened first), so a ray
// takes about half as many dependent node fetches as in the binary tree.  An unused slot has ref kRefNone.
struct __attribute__((aligned(64))) DevNodeW {
  uint32_t px[4], py[4], pz[4];
  uint32_t child[4];
};
static_assert(sizeof(DevNodeW) == 64, "wide node must be one 64-byte record");
constexpr int kWideLdsLevels = 22;  // stack entries per lane the walk on wide nodes keeps in LDS: with the four leaf-queue
                                    // levels 26 x 256 B per wave, what 24 waves (six per SIMD) can share of a CU's 160 KB

// 80-byte triangle record read by the leaf test (triangle.rs:76-117).
struct __attribute__((aligned(16))) DevTriangle {
  double v0[3], e1[3], e2[3];
  uint32_t element;
  uint32_t rank;  // position of the triangle's leaf in the reference's depth-first, left-first leaf order
};
static_assert(sizeof(DevTriangle) == 80, "triangle record");

// 32-byte record read when a triangle hit is shaded: Triangle::new's unit
// normal (triangle.rs:58) and the index of the triangle's Material.
struct __attribute__((aligned(32))) DevTriShade {
  double normal[3];
  int32_t material;
  int32_t pad;
};
static_assert(sizeof(DevTriShade) == 32, "triangle shade record");

struct DevQuad {
  double q[3], u[3], v[3], normal[3], w[3];
  double d, area;
  uint32_t element, material;
  uint32_t rank, pad;
};
struct DevSphere {
  double center[3];
  double radius;
  uint32_t element, material;
  uint32_t rank, pad;
};
struct DevMaterial {
  double albedo[3];
  double param;
  int32_t kind;
  int32_t pad;
};

struct DevScene {
  const DevNode *nodes;
  const DevNodeQ *nodes_q;  // non-null when the 16-bit grid is fine enough for this scene
  const DevNodeD *nodes_d;  // non-null when some box does not bound its contents (see DevNodeD)
  const DevNodeW *nodes_w;  // non-null when the wide form of the grid nodes was made (order-free walk only)
  uint32_t root_w;          // ref of the wide root (a leaf ref when the tree is a single leaf)
  int32_t stack_depth_w;    // stack entries a lane can need on the wide nodes
  double grid_min[3], grid_step[3];  // grid value q on axis a stands for grid_min[a] + q * grid_step[a]
  const DevTriangle *triangles;  // record i starts i * tri_stride bytes in: see tri_record
  uint32_t tri_stride;           // 80, or 128 (one record per line, none across two) when the records do not fit the caches
  const DevTriShade *tri_shade;
  const DevQuad *quads;
  const DevSphere *spheres;
  const DevMaterial *materials;
  const uint32_t *attractors;  // leaf refs (kind + index)
  int32_t n_attractors;
  int32_t n_materials;
  uint32_t root;        // ref of nodes[0] of the reference array
  int32_t stack_depth;  // stack entries a lane can need (the height of the tree)
  // Global-memory part of the traversal stacks, [level][wave][lane]: the blocks of entries the binary order-free walks move out
  // of LDS on trees too high for it, and the wide walk's levels beyond kWideLdsLevels; null when no walk of the scene needs it.
  uint32_t *stack_spill;
  int32_t quirk;        // 1 when the walk must follow the reference's visit rule step by step: some child box
                        // does not contain the primitives below it, or the nodes do not form a tree
  // When the only offenders are a few primitives that stick out of their own leaf boxes, `danger` holds
  // their true bounds (min xyz, max xyz per box): a ray that misses all of them cannot hit any of those
  // primitives, so for it the reference's result is the plain closest hit and the fast walk applies.
  const double *danger;
  int32_t n_danger;
  RtCamera camera;
};

__host__ __device__ inline const DevTriangle *tri_record(const DevScene &sc, uint32_t idx) {
  return reinterpret_cast<const DevTriangle *>(reinterpret_cast<const char *>(sc.triangles) + (size_t)idx * sc.tri_stride);
}

// ----------------------------------------------------------------- vec3
struct d3 {
  double x, y, z;
};
RT_DEV d3 D3(double x, double y, double z) { return d3{x, y, z}; }
RT_DEV d3 ld3(const double *p) { return d3{p[0], p[1], p[2]}; }
RT_DEV d3 ld3(const RtVec3 &p) { return d3{p.x, p.y, p.z}; }
RT_DEV d3 operator+(d3 a, d3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV d3 operator-(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV d3 operator*(d3 a, double q) { return {a.x * q, a.y * q, a.z * q}; }
RT_DEV d3 operator*(d3 a, d3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV d3 operator/(d3 a, double q) { return {a.x / q, a.y / q, a.z / q}; }
RT_DEV d3 neg(d3 a) { return {-a.x, -a.y, -a.z}; }
RT_DEV double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV d3 cross(d3 a, d3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RT_DEV double length_squared(d3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
RT_DEV double length(d3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // vec3.rs:116-128
RT_DEV d3 normalized(d3 a) {                                                    // vec3.rs:163-166
  double l = length(a);
  return {a.x / l, a.y / l, a.z / l};
}
RT_DEV bool has_nan(d3 a) { return isnan(a.x) || isnan(a.y) || isnan(a.z); }

// ------------------------------------------------------------------ RNG
// One SplitMix64 stream per path; identical to the CPU oracle's RT_RNG_CTR
// mode.  The conversions are rand 0.8.5's (gen_range, Standard f64, index).
RT_DEV uint64_t splitmix64_mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
// the stream's key is mixed from the pixel first: that part is the same for every batch and sample of the pixel
RT_DEV uint64_t rng_seed_pixel(uint64_t pixel_index) { return splitmix64_mix(pixel_index + 0x9e3779b97f4a7c15ULL); }
RT_DEV uint64_t rng_seed_path_of(uint64_t pixel_key, uint64_t batch, uint64_t sample, uint64_t seed) {
  uint64_t k = splitmix64_mix(pixel_key ^ (batch * 0xd1342543de82ef95ULL + 0x632be59bd9b4e019ULL));
  k = splitmix64_mix(k ^ (sample * 0xaf251af3b0f025b5ULL + 0x9e3779b97f4a7c15ULL));
  return splitmix64_mix(k ^ seed);
}
RT_DEV uint64_t rng_seed_path(uint64_t pixel_index, uint64_t batch, uint64_t sample, uint64_t seed) {
  return rng_seed_path_of(rng_seed_pixel(pixel_index), batch, sample, seed);
}
RT_DEV uint64_t rng_next_u64(uint64_t &s) {
  s += 0x9e3779b97f4a7c15ULL;
  return splitmix64_mix(s);
}
RT_DEV double rng_gen_range(uint64_t &s, double low, double high) {
  double scale = high - low;
  for (;;) {
    uint64_t bits = (rng_next_u64(s) >> 12) | 0x3ff0000000000000ULL;
    double value0_1 = __longlong_as_double((long long)bits) - 1.0;
    double res = value0_1 * scale + low;
    if (res < high) return res;
  }
}
RT_DEV double rng_gen_f64(uint64_t &s) { return (double)(rng_next_u64(s) >> 11) * (1.0 / 9007199254740992.0); }
RT_DEV uint32_t rng_gen_index(uint64_t &s, uint32_t n) {
  uint32_t zone = (n << __clz((int)n)) - 1u;
  for (;;) {
    uint32_t v = (uint32_t)(rng_next_u64(s) >> 32);
    uint64_t m = (uint64_t)v * (uint64_t)n;
    if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
  }
}
RT_DEV d3 random_unit_vector(uint64_t &s) {  // rand_distr UnitSphere, vec3.rs:35-38
  for (;;) {
    double x1 = rng_gen_range(s, -1.0, 1.0);
    double x2 = rng_gen_range(s, -1.0, 1.0);
    double sum = x1 * x1 + x2 * x2;
    if (sum >= 1.0) continue;
    double factor = 2.0 * sqrt(1.0 - sum);
    return {x1 * factor, x2 * factor, 1.0 - 2.0 * sum};
  }
}
RT_DEV d3 random_in_unit_disk(uint64_t &s) {  // vec3.rs:41-48
  for (;;) {
    double a = rng_gen_range(s, -1.0, 1.0);
    double b = rng_gen_range(s, -1.0, 1.0);
    if (a * a + b * b + 0.0 * 0.0 < 1.0) return {a, b, 0.0};
  }
}
RT_DEV d3 random_cosine_direction(uint64_t &s) {  // vec3.rs:69-79
  double r1 = rng_gen_range(s, 0.0, 1.0);
  double r2 = rng_gen_range(s, 0.0, 1.0);
  double phi = 2. * kPi * r1;
  double sn, cs;
  sincos(phi, &sn, &cs);
  double sq = sqrt(r2);
  return {cs * sq, sn * sq, sqrt(1. - r2)};
}

// ---------------------------------------------------------------- Sobol
// Owen-scrambled, index-shuffled Sobol (Burley 2020), dims 0 and 1; the same
// restatement of sobol_burley::sample as the CPU oracle's.
RT_DEV uint32_t hash_u32(uint32_t n) {
  n ^= 0xe6fe3beb;
  n ^= n >> 16;
  n *= 0x7feb352d;
  n ^= n >> 15;
  n *= 0x846ca68b;
  n ^= n >> 16;
  return n;
}
RT_DEV uint32_t owen_scramble_rev(uint32_t n, uint32_t key) {
  n ^= n * 0x3d20adea;
  n += key;
  n *= (key >> 16) | 1;
  n ^= n * 0x05526c56;
  n ^= n * 0x53a22864;
  return n;
}
RT_DEV float sobol_sample(uint32_t index, uint32_t dim, uint32_t seed) {
  uint32_t shuffled = __brev(owen_scramble_rev(__brev(index), hash_u32(seed)));
  uint32_t idx = shuffled & 0xffffu;
  uint32_t x = 0;
  if (dim == 0) {
    x = __brev(idx);
  } else {
    uint32_t v = 0x80000000u;
    for (int k = 0; k < 16; k++) {
      if (idx & (1u << k)) x ^= v;
      v ^= v >> 1;
    }
  }
  uint32_t key = hash_u32(seed ^ hash_u32(dim + 1u));
  x = __brev(owen_scramble_rev(__brev(x), key));
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// --------------------------------------------------------------- filter
RT_DEV double mitchell_1d(double x, double b, double c) {  // filter.rs:45-61
  double fx = fabs(x) * 2.0;
  if (fx < 1.0) {
    return ((12.0 - 9.0 * b - 6.0 * c) * fx * fx * fx + (-18.0 + 12.0 * b + 6.0 * c) * fx * fx + (6.0 - 2.0 * b)) * (1.0 / 6.0);
  } else if (fx < 2.0) {
    return ((-b - 6.0 * c) * fx * fx * fx + (6.0 * b + 30.0 * c) * fx * fx + (-12.0 * b - 48.0 * c) * fx + (8.0 * b + 24.0 * c)) *
           (1.0 / 6.0);
  }
  return 0.0;
}

// ------------------------------------------------------------------ ONB
struct Onb {
  d3 u, v, w;
};
RT_DEV Onb onb_build_from_w(d3 w) {  // onb.rs:31-45
  d3 unit_w = normalized(w);
  d3 a = (fabs(unit_w.x) > 0.9) ? D3(0., 1., 0.) : D3(1.0, 0., 0.);
  d3 v = normalized(cross(unit_w, a));
  d3 u = cross(unit_w, v);
  return {u, v, unit_w};
}
RT_DEV d3 onb_local_vec(const Onb &o, d3 vec) { return o.u * vec.x + o.v * vec.y + o.w * vec.z; }  // onb.rs:27-29

// ---------------------------------------------------------------- a ray
struct Ray {
  d3 o, d, inv;  // Ray::new, ray.rs:35-48 (d already normalised)
};
RT_DEV Ray make_ray_normalized(d3 o, d3 dn) { return {o, dn, D3(1.0 / dn.x, 1.0 / dn.y, 1.0 / dn.z)}; }
RT_DEV Ray make_ray(d3 o, d3 dir) { return make_ray_normalized(o, normalized(dir)); }

// ----------------------------------------------------- primitive tests
// triangle.rs:76-117; returns true and t when the hit lies in [tmin, tmax].
RT_DEV bool triangle_hit(d3 v0, d3 e1, d3 e2, const Ray &r, double tmin, double tmax, double &t_out) {
  d3 pvec = cross(r.d, e2);
  double det = dot(e1, pvec);
  if (det < kEpsilon) return false;
  double inv_det = 1. / det;
  d3 tvec = r.o - v0;
  double u = dot(tvec, pvec) * inv_det;
  if (!(0.0 <= u && u <= 1.0)) return false;
  d3 qvec = cross(tvec, e1);
  double v = dot(r.d, qvec) * inv_det;
  if (v < 0.0 || u + v > 1.0) return false;
  double t = dot(e2, qvec) * inv_det;
  if (!(tmin <= t && t <= tmax)) return false;
  t_out = t;
  return true;
}
// quad.rs:108-145
RT_DEV bool quad_hit(const DevQuad &q, const Ray &r, double tmin, double tmax, double &t_out) {
  d3 normal = ld3(q.normal);
  double denom = dot(normal, r.d);
  if (fabs(denom) < kEpsilon) return false;
  double t = (q.d - dot(normal, r.o)) / denom;
  if (!(tmin <= t && t <= tmax)) return false;
  d3 p = (r.o + r.d * t) - ld3(q.q);
  d3 w = ld3(q.w);
  double alpha = dot(w, cross(p, ld3(q.v)));
  double beta = dot(w, cross(ld3(q.u), p));
  if (!(0.0 <= alpha && alpha <= 1.0) || !(0.0 <= beta && beta <= 1.0)) return false;
  t_out = t;
  return true;
}
// sphere.rs:94-145 (strict interval; only the numerically smaller root)
RT_DEV bool sphere_hit(const DevSphere &s, const Ray &r, double tmin, double tmax, double &t_out) {
  d3 l = r.o - ld3(s.center);
  double a = dot(r.d, r.d);
  double b = 2.0 * dot(r.d, l);
  double c = dot(l, l) - (s.radius * s.radius);
  double discr = b * b - 4.0 * a * c;
  if (discr >= 0.0) {
    double q = (b > 0.0) ? -0.5 * (b + sqrt(discr)) : -0.5 * (b - sqrt(discr));
    double first_root = q / a;
    double second_root = c / q;
    if (first_root < 0.0 && second_root < 0.0) return false;
    double nearest_root = (first_root < second_root) ? first_root : second_root;
    if (nearest_root < tmax && nearest_root > tmin) {
      t_out = nearest_root;
      return true;
    }
  }
  return false;
}

// ------------------------------------------------ BVH closest-hit query
// Restates BVH_SAH::hit (bvh_node_sah.rs:162-253).  The reference's recursion
// is an ordered depth-first walk, left child first, that keeps the best t so
// far in ray_t.interval_max; a later leaf replaces the current hit when its t
// lies in [tmin, best] (inclusive for triangles and quads, strict for
// spheres).  Child boxes are never culled against t (aabb.rs:187-204 ignores
// ray_t), so which leaves get tested depends on the boxes alone -- with one
// irregularity: after the left subtree of a node returned a hit, the right
// child is entered WITHOUT testing its box (bvh_node_sah.rs:215-216).
//
// That irregularity can only change a result when a primitive sticks out of
// the boxes above it (the reference's Quad bounding box is wrong for quads
// that are not axis-aligned, quad.rs:86).  The scene upload checks whether
// every child box contains the primitives below it:
//   QUIRK == false  all boxes bound their contents: a right child whose box
//                   the ray misses is never entered (it cannot hold a hit);
//   QUIRK == true   some box does not: the right child is pushed with a
//                   "box passed" bit and `hitmask` bit s records whether a hit
//                   was accepted since stack entry s was pushed, exactly
//                   reproducing the reference's visit rule.
struct Hit {
  double t;
  uint32_t ref;  // leaf ref of the primitive hit, kRefNone on a miss
};

constexpr uint32_t kDone = 0xffffffffu;  // no node left (has the leaf bit set on purpose)
constexpr uint32_t kWait = 0xfffffffeu;  // QUIRK: a pop is waiting for deferred leaf tests
constexpr uint32_t kPending = 0xfffffffdu;  // hit buffer: the ray is left to the reference-rule pass
constexpr int kMaxLeafQueue = 4;         // deferred leaf tests per lane

// f32 copy of a ray for the box tests.  The origin is widened by its own
// rounding error: o_up >= o >= o_dn, so that with boxes rounded outward the
// computed slab interval contains the exact one.
struct RayF {
  float oux, ouy, ouz;  // origin rounded up
  float odx, ody, odz;  // origin rounded down
  float ix, iy, iz;     // 1/direction
};
RT_DEV RayF make_rayf(d3 o, d3 d) {
  RayF f;
  float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
  const float u2 = 1.1920929e-7f;  // 2^-23: twice the rounding error of the conversion
  float ex = fabsf(ox) * u2 + 1e-30f, ey = fabsf(oy) * u2 + 1e-30f, ez = fabsf(oz) * u2 + 1e-30f;
  f.oux = ox + ex; f.ouy = oy + ey; f.ouz = oz + ez;
  f.odx = ox - ex; f.ody = oy - ey; f.odz = oz - ez;
  // 1/d to within two f32 roundings; a zero component gives +-inf like the reference's 1.0 / d
  f.ix = 1.0f / (float)d.x; f.iy = 1.0f / (float)d.y; f.iz = 1.0f / (float)d.z;
  // a component too small for f32 must keep its sign and become +-inf, not flip through zero
  if ((float)d.x == 0.0f) f.ix = copysignf(INFINITY, (float)copysign(1.0, d.x));
  if ((float)d.y == 0.0f) f.iy = copysignf(INFINITY, (float)copysign(1.0, d.y));
  if ((float)d.z == 0.0f) f.iz = copysignf(INFINITY, (float)copysign(1.0, d.z));
  return f;
}
// Conservative slab test (aabb.rs:187-204 evaluated in f32): the min corner is
// always paired with the rounded-up origin and the max corner with the
// rounded-down one, which makes the near plane nearer and the far plane
// farther for either sign of the direction; the far side is widened by the
// relative rounding error of the f32 operations.  fminf/fmaxf return the
// non-NaN operand like f64::min/max.  Passes whenever the reference's test does.
RT_DEV bool slab_hit_f32(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, const RayF &f) {
  float tlx = (bminx - f.oux) * f.ix, thx = (bmaxx - f.odx) * f.ix;
  float tly = (bminy - f.ouy) * f.iy, thy = (bmaxy - f.ody) * f.iy;
  float tlz = (bminz - f.ouz) * f.iz, thz = (bmaxz - f.odz) * f.iz;
  float tn = fmaxf(fmaxf(fminf(tlx, thx), fminf(tly, thy)), fmaxf(fminf(tlz, thz), 0.0f));
  float tf = fminf(fminf(fmaxf(tlx, thx), fmaxf(tly, thy)), fmaxf(tlz, thz));
  return tn <= tf * 1.000001f;
}
// The same test that also rejects a box lying wholly beyond `tmax` (an upper bound of the
// closest hit so far) and reports the entry distance.  The entry distance carries the same
// relative rounding error as the exit distance, hence the same widening of the bound.
RT_DEV bool slab_hit_f32(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, const RayF &f, float tmax,
                         float &tn_out) {
  float tlx = (bminx - f.oux) * f.ix, thx = (bmaxx - f.odx) * f.ix;
  float tly = (bminy - f.ouy) * f.iy, thy = (bmaxy - f.ody) * f.iy;
  float tlz = (bminz - f.ouz) * f.iz, thz = (bmaxz - f.odz) * f.iz;
  float tn = fmaxf(fmaxf(fminf(tlx, thx), fminf(tly, thy)), fmaxf(fminf(tlz, thz), 0.0f));
  float tf = fminf(fminf(fmaxf(tlx, thx), fmaxf(tly, thy)), fmaxf(tlz, thz));
  tn_out = tn;
  return tn <= fminf(tf, tmax) * 1.000001f;
}

// Ray constants for boxes on the 16-bit grid: the slab distance of grid value q on axis a is
// (grid_min + q*step - o) / d = q * A + B with A = step/d and B = (grid_min - o)/d, one fma.
// A and B are computed in f32 and the fma rounds once more; per axis, e bounds the sum of those
// errors (rayq_axis), and it is folded into the constants: the plane the ray meets first uses B - e, the
// other one B + e (which of the two grid values that is depends on the sign of A only), so the
// computed interval contains the exact one.  The widening must be per axis: for a ray almost
// parallel to an axis |A| and e are huge there, and a single bound would pass every box.
struct RayQ {
  float ax, ay, az;
  float blx, bly, blz;  // added to q_min * A (SORTED: to the grid value of the plane the ray meets first)
  float bhx, bhy, bhz;  // added to q_max * A (SORTED: to the grid value of the other plane)
  uint32_t selx, sely, selz;  // SORTED only: byte selectors that bring that first plane's half of a (min | max << 16) word into the low half
};
// SORTED: the constants belong to the near and the far plane instead of the min and the max plane, and a selector says which
// half of a packed (min | max << 16) word is the near one: the node step swaps the halves with one v_perm_b32 and needs no min / max per axis.
template <bool SORTED = false>
RT_DEV void rayq_axis(double step, double gmin, double o, double d, float &a_out, float &bl_out, float &bh_out, uint32_t &sel_out) {
  // Everything but the subtraction gmin - o (which would cancel in f32) is f32 arithmetic: a ray is set up once
  // per traversal by the lanes that refill, and three f64 divisions there cost more than the node steps they feed.
  // Errors, relative to |A| resp. |B|: d -> f32 2^-24, the reciprocal instruction 2^-23, step resp. (gmin - o) -> f32
  // 2^-24, the product 2^-24, i.e. 2.5 * 2^-23 = 3.0e-7 on each constant; the kernel's fma adds 2^-24 of its
  // result, |q*A + B| <= 65535 |A| + |B|; rounding B -+ e to f32 another 2^-24 |B|.  Sum < 4.2e-7: e uses 4.8e-7.
  // A direction component that is zero (or below 1e-25) counts as +-1e-25: with an infinite reciprocal q*A + B would be
  // inf - inf for every box and the axis would drop out of the test altogether -- still conservative, but a ray parallel
  // to two axes would then enter almost every box of the tree.  With the finite stand-in a box whose slab does not
  // contain the origin is 1e25 x its distance away (beyond any hit), one that does contains t = 0, and e below, 4.8e-7 of
  // 1e25 x the scene's extent, covers what the stand-in and the rounding change.
  float inv = __builtin_amdgcn_rcpf((float)d);
  inv = fminf(fmaxf(inv, -1e25f), 1e25f);
  float a = (float)step * inv, b = (float)(gmin - o) * inv;
  float e = (65535.0f * fabsf(a) + fabsf(b)) * 4.8e-7f + 1e-30f;
  if (!(e <= 3.0e38f)) e = 0.0f;  // a scene wider than 1e8 units with such a ray: the products are inf/NaN and the axis drops out of min/max
  a_out = a;
  // a >= 0: q_min gives the near plane; a < 0: q_max does.  The near plane moves nearer, the far plane farther.
  if (SORTED) {
    bl_out = b - e;
    bh_out = b + e;
    sel_out = a >= 0.0f ? 0x03020100u : 0x01000302u;
  } else {
    bl_out = a >= 0.0f ? b - e : b + e;
    bh_out = a >= 0.0f ? b + e : b - e;
    sel_out = 0;
  }
}
template <bool SORTED = false>
RT_DEV RayQ make_rayq(const DevScene &sc, d3 o, d3 d) {
  RayQ r;
  rayq_axis<SORTED>(sc.grid_step[0], sc.grid_min[0], o.x, d.x, r.ax, r.blx, r.bhx, r.selx);
  rayq_axis<SORTED>(sc.grid_step[1], sc.grid_min[1], o.y, d.y, r.ay, r.bly, r.bhy, r.sely);
  rayq_axis<SORTED>(sc.grid_step[2], sc.grid_min[2], o.z, d.z, r.az, r.blz, r.bhz, r.selz);
  return r;
}
// The same test on packed words (min | max << 16 per axis) with a SORTED ray: the halves are put in near / far order first.
RT_DEV bool slab_hit_qs(uint32_t wx, uint32_t wy, uint32_t wz, const RayQ &r, float tmax, float &tn_out) {
  const uint32_t px = __builtin_amdgcn_perm(wx, wx, r.selx), py = __builtin_amdgcn_perm(wy, wy, r.sely), pz = __builtin_amdgcn_perm(wz, wz, r.selz);
  float nx = fmaf((float)(px & 0xffffu), r.ax, r.blx), fx = fmaf((float)(px >> 16), r.ax, r.bhx);
  float ny = fmaf((float)(py & 0xffffu), r.ay, r.bly), fy = fmaf((float)(py >> 16), r.ay, r.bhy);
  float nz = fmaf((float)(pz & 0xffffu), r.az, r.blz), fz = fmaf((float)(pz >> 16), r.az, r.bhz);
  float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
  float tf = fminf(fminf(fx, fy), fminf(fz, tmax));
  tn_out = tn;
  return tn <= tf;
}
// Conservative slab test on the grid: passes whenever the reference's f64 test on the original
// box does, and never rejects a box that begins at or before `tmax`.
// Both planes of an axis in one packed fma: (q_min, q_max) * (A, A) + (B_lo, B_hi).
typedef float f32x2 __attribute__((ext_vector_type(2)));
RT_DEV bool slab_hit_q2(f32x2 qx, f32x2 qy, f32x2 qz, const RayQ &r, float tmax, float &tn_out) {
  f32x2 tx = __builtin_elementwise_fma(qx, f32x2{r.ax, r.ax}, f32x2{r.blx, r.bhx});
  f32x2 ty = __builtin_elementwise_fma(qy, f32x2{r.ay, r.ay}, f32x2{r.bly, r.bhy});
  f32x2 tz = __builtin_elementwise_fma(qz, f32x2{r.az, r.az}, f32x2{r.blz, r.bhz});
  float tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), 0.0f));
  float tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fmaxf(tz.x, tz.y));
  tn_out = tn;
  return tn <= fminf(tf, tmax);
}
RT_DEV bool slab_hit_q(float qminx, float qminy, float qminz, float qmaxx, float qmaxy, float qmaxz, const RayQ &r, float tmax, float &tn_out) {
  float tlx = fmaf(qminx, r.ax, r.blx), thx = fmaf(qmaxx, r.ax, r.bhx);
  float tly = fmaf(qminy, r.ay, r.bly), thy = fmaf(qmaxy, r.ay, r.bhy);
  float tlz = fmaf(qminz, r.az, r.blz), thz = fmaf(qmaxz, r.az, r.bhz);
  float tn = fmaxf(fmaxf(fminf(tlx, thx), fminf(tly, thy)), fmaxf(fminf(tlz, thz), 0.0f));
  float tf = fminf(fminf(fmaxf(tlx, thx), fmaxf(tly, thy)), fmaxf(tlz, thz));
  tn_out = tn;
  return tn <= fminf(tf, tmax);
}

// Ray constants for the f32 boxes in the form one fma per plane: the slab distance of plane b on an axis is
// (b - o) / d = b * I + C with I = 1/d and C = -o * I.  I carries a relative error (conversion of d, the reciprocal
// instruction), but the SAME I multiplies b and o, so that error is relative to the distance itself and the factor on the
// far side in slab_hit_p absorbs it, as in slab_hit_f32; the product b * I is exact inside the fma.  What is absolute is the
// rounding of C (computed in f64, rounded once: 2^-24 |C|): E = 2^-22 |C| is folded into the constants, C - E for the plane
// the ray meets first and C + E for the other (which of the two box planes that is depends on the sign of I only).  A
// direction component of zero (or below 1e-25) counts as +-1e-25, as in rayq_axis and for the same reason.
struct RayP {
  float ix, iy, iz;
  float clx, cly, clz;  // added to b_min * I
  float chx, chy, chz;  // added to b_max * I
};
RT_DEV void rayp_axis(double o, double d, float &i_out, float &cl_out, float &ch_out) {
  float inv = __builtin_amdgcn_rcpf((float)d);
  inv = fminf(fmaxf(inv, -1e25f), 1e25f);
  if (!(inv == inv)) inv = 1e25f;
  const float c = (float)(-o * (double)inv);
  const float e = fabsf(c) * 2.4e-7f + 1e-30f;
  i_out = inv;
  cl_out = inv >= 0.0f ? c - e : c + e;
  ch_out = inv >= 0.0f ? c + e : c - e;
}
RT_DEV RayP make_rayp(d3 o, d3 d) {
  RayP r;
  rayp_axis(o.x, d.x, r.ix, r.clx, r.chx);
  rayp_axis(o.y, d.y, r.iy, r.cly, r.chy);
  rayp_axis(o.z, d.z, r.iz, r.clz, r.chz);
  return r;
}
// Passes whenever the reference's f64 test on a box inside (bmin, bmax) does, and never rejects a box that begins at or before `tmax`.
RT_DEV bool slab_hit_p(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, const RayP &r, float tmax, float &tn_out) {
  float tlx = fmaf(bminx, r.ix, r.clx), thx = fmaf(bmaxx, r.ix, r.chx);
  float tly = fmaf(bminy, r.iy, r.cly), thy = fmaf(bmaxy, r.iy, r.chy);
  float tlz = fmaf(bminz, r.iz, r.clz), thz = fmaf(bmaxz, r.iz, r.chz);
  float tn = fmaxf(fmaxf(fminf(tlx, thx), fminf(tly, thy)), fmaxf(fminf(tlz, thz), 0.0f));
  float tf = fminf(fminf(fmaxf(tlx, thx), fmaxf(tly, thy)), fmaxf(tlz, thz));
  tn_out = tn;
  return tn <= fminf(tf, tmax) * 1.000001f;
}

// Does the ray come near one of the out-of-box primitives?  Slab test in f64 on their true bounds
// (padded at upload); passes whenever the ray could touch the box.
RT_DEV bool hits_danger_box(const DevScene &sc, d3 o, d3 d) {
  d3 inv = D3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
  bool any = false;
  for (int i = 0; i < sc.n_danger; i++) {
    const double *b = sc.danger + 6 * i;
    double t1 = (b[0] - o.x) * inv.x, t2 = (b[3] - o.x) * inv.x;
    double t3 = (b[1] - o.y) * inv.y, t4 = (b[4] - o.y) * inv.y;
    double t5 = (b[2] - o.z) * inv.z, t6 = (b[5] - o.z) * inv.z;
    double tn = fmax(fmax(fmin(t1, t2), fmin(t3, t4)), fmin(t5, t6));
    double tf = fmin(fmin(fmax(t1, t2), fmax(t3, t4)), fmax(t5, t6));
    any = any || !(tn > tf * (1.0 + 1e-12) + 1e-300) || tn != tn || tf != tf;  // NaN (0 * inf) counts as a hit
  }
  return any;
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Per-lane traversal state.  `stack` / `leafq` point at this lane's slot of
// LDS arrays laid out [level][lane]; consecutive levels are `stride` entries
// apart, so a wave's accesses to one level touch 64 consecutive dwords.
struct Traversal {
  uint32_t cur;      // node ref, leaf ref, kDone or kWait
  int sp;            // stack entries
  int qn;            // deferred leaves
  uint64_t hitmask;  // QUIRK: bit s = a hit was accepted since stack entry s was pushed
  u16x2 m01, m23;    // QUIRK: per deferred leaf, the lowest stack height since it was queued
  float tcull;       // !QUIRK: best.t rounded up to f32 (+inf before the first hit)
  Hit best;
  uint32_t moves;    // blocks of stack entries moved between LDS and global memory (counted over the lane's whole life)
};

// A direction component of exactly -0.0: Ray::new takes its sign as positive (`-0.0 < 0.0` is false, ray.rs:44-46) and its
// reciprocal as -inf (ray.rs:38), so Aabb::hit pairs the wrong planes (aabb.rs:187-204): inside the slab the interval is
// (+inf, -inf), beside it both ends are -inf or both +inf, and with at least one finite axis every box fails.  Such a ray
// misses everything below an internal root.
RT_DEV bool ray_has_negative_zero(d3 d) {
  return (d.x == 0.0 && __builtin_signbit(d.x)) || (d.y == 0.0 && __builtin_signbit(d.y)) || (d.z == 0.0 && __builtin_signbit(d.z));
}

RT_DEV void trav_begin(uint32_t root, Traversal &tv) {
  tv.cur = root;
  tv.sp = 0;
  tv.qn = 0;
  tv.hitmask = 0;
  tv.m01 = u16x2{0, 0};
  tv.m23 = u16x2{0, 0};
  tv.tcull = INFINITY;
  tv.best.t = kF64Max;
  tv.best.ref = kRefNone;
}

// A lane's stack.  Three ways to hold it (template argument SPILL of the functions below):
//   0  every level in LDS;
//   1  the first LDSL levels in LDS, level LDSL and above in global memory, one slot per level;
//   2  the newest entries in kOvfLevels levels of LDS.  A push that finds them full moves the oldest kOvfBlock entries out to
//      global memory as a block and slides the rest down; a pop that finds LDS empty fetches the newest block back.  The number
//      of blocks outside is kept in the LDS row after the stack's (zero whenever a walk starts: a walk ends with its stack
//      drained).  Both moves are rare -- a walk that enters the nearer child first seldom holds more than a dozen entries however
//      high the tree -- and sit in cold branches, so the step costs one compare more than mode 0 and the LDS footprint no longer
//      depends on the tree.
constexpr int kOvfLevels = RT_OVF_LEVELS, kOvfBlock = 8;
static_assert(kOvfLevels == 2 * kOvfBlock, "stack_block_out slides exactly one block down: the LDS part must hold two blocks");
struct StackRef {
  uint32_t *lds;     // this lane's slot of level 0; levels are 64 entries apart
  uint32_t *spill;   // mode 1: this lane's slot of the first spilled level; mode 2: the wave's first slot of the buffer
  uint32_t spill_stride;  // entries between two levels of the buffer
};
template <int SPILL, int LDSL = kLdsStackLevels>
RT_DEV void stack_store(const StackRef &s, int level, uint32_t v) {
  if (SPILL != 1 || level < LDSL) s.lds[level * 64] = v;
  else s.spill[(size_t)(level - LDSL) * s.spill_stride] = v;
}
template <int SPILL, int LDSL = kLdsStackLevels>
RT_DEV uint32_t stack_load(const StackRef &s, int level) {
  if (SPILL != 1 || level < LDSL) return s.lds[level * 64];
  return s.spill[(size_t)(level - LDSL) * s.spill_stride];
}
// mode 2: this lane's slot of level `level` of the global buffer.  The lane number passes through an empty asm statement
// so that the address is formed here, in the cold branch, and does not occupy registers through the whole walk.
RT_DEV uint32_t *stack_block_slot(const StackRef &s, uint32_t level) {
  uint32_t l = threadIdx.x;
  asm volatile("" : "+v"(l));
  return s.spill + (size_t)level * s.spill_stride + l;
}
RT_DEV void stack_block_out(Traversal &tv, const StackRef &s) {
  const uint32_t nb = s.lds[kOvfLevels * 64];
#pragma unroll
  for (int k = 0; k < kOvfBlock; k++) {
    *stack_block_slot(s, nb * kOvfBlock + k) = s.lds[k * 64];
    s.lds[k * 64] = s.lds[(k + kOvfBlock) * 64];
  }
  s.lds[kOvfLevels * 64] = nb + 1u;
  tv.sp = kOvfLevels - kOvfBlock;
  tv.moves++;
}
template <int SPILL, int LDSL = kLdsStackLevels>
RT_DEV void stack_push(Traversal &tv, const StackRef &s, uint32_t v) {
  if (SPILL == 2 && __builtin_expect(tv.sp == kOvfLevels, 0)) stack_block_out(tv, s);
  stack_store<SPILL, LDSL>(s, tv.sp, v);
  tv.sp++;
}
// the order-free walk's pop: entries are children whose box the ray hits
template <int SPILL, int LDSL = kLdsStackLevels>
RT_DEV void trav_pop_fast(Traversal &tv, const StackRef &s) {
  if (tv.sp == 0) {
    if (SPILL == 2) {
      uint32_t nb = s.lds[kOvfLevels * 64];
      if (nb != 0u) {  // the newest block comes back
        nb--;
#pragma unroll
        for (int k = 0; k < kOvfBlock; k++) s.lds[k * 64] = *stack_block_slot(s, nb * kOvfBlock + k);
        s.lds[kOvfLevels * 64] = nb;
        tv.moves++;
        tv.sp = kOvfBlock - 1;
        tv.cur = s.lds[tv.sp * 64];
        return;
      }
    }
    tv.cur = kDone;
  } else {
    tv.sp--;
    tv.cur = stack_load<SPILL, LDSL>(s, tv.sp);
  }
}

// Take the next node off the stack.
//   !QUIRK: entries are right children whose box the ray hits.
//   QUIRK : every right child of a node whose left box was hit is on the stack
//           with a "box passed" bit; an entry without the bit is entered only
//           if a hit was accepted since it was pushed (bvh_node_sah.rs:211-216).
//           While leaf tests queued since then are still pending that is not
//           known yet, and the lane waits (kWait) for the next leaf phase.
template <bool QUIRK>
RT_DEV void trav_pop(Traversal &tv, const uint32_t *stack, int stride) {
  if (!QUIRK) {
    if (tv.sp == 0) {
      tv.cur = kDone;
    } else {
      tv.sp--;
      tv.cur = stack[tv.sp * stride];
    }
  } else {
    for (;;) {
      if (tv.sp == 0) {
        tv.cur = kDone;
        return;
      }
      uint32_t e = stack[(tv.sp - 1) * stride];
      bool enter = (e & kRefFlag) || ((tv.hitmask >> (tv.sp - 1)) & 1ull);
      if (!enter && tv.qn > 0) {
        tv.cur = kWait;
        return;
      }
      tv.sp--;
      unsigned short h = (unsigned short)tv.sp;
      tv.m01 = __builtin_elementwise_min(tv.m01, u16x2{h, h});
      tv.m23 = __builtin_elementwise_min(tv.m23, u16x2{h, h});
      if (enter) {
        tv.cur = e & ~kRefFlag;
        return;
      }
    }
  }
}

// One internal-node step of the order-free walk on the 64-byte f32 records: boxes bound their
// contents, so which leaves are tested, and in which order, cannot change the result
// (trav_test_leaf breaks ties by the reference's leaf order).  Enter the nearer child first and
// drop children that begin beyond the closest hit so far.
template <int SPILL>
RT_DEV void trav_node_fast(const DevScene &sc, const RayP &f, Traversal &tv, const StackRef &stack) {
  const float4 *p = reinterpret_cast<const float4 *>(&sc.nodes[tv.cur & kRefIndexMask]);
  float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
  uint32_t left = __float_as_uint(q3.x), right = __float_as_uint(q3.y);
  float tl, tr;
  bool hl = slab_hit_p(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, f, tv.tcull, tl);
  bool hr = slab_hit_p(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, f, tv.tcull, tr);
  if (hl && hr) {
    bool swap = tr < tl;
    stack_push<SPILL>(tv, stack, swap ? left : right);
    tv.cur = swap ? right : left;
  } else if (hl || hr) {
    tv.cur = hl ? left : right;
  } else {
    trav_pop_fast<SPILL>(tv, stack);
  }
}

// Aabb::hit, aabb.rs:187-204, on the reference's f64 box: per axis the plane the ray meets first is picked by the
// sign of the direction (ray.rs:44-46: `dir_norm < 0.0`, so -0.0 counts as positive although its reciprocal is -inf), f64::max / min return the non-NaN operand like fmax / fmin.
RT_DEV bool slab_hit_exact(const double *mn, const double *mx, d3 o, d3 d, d3 inv) {
  const bool sx = d.x < 0.0, sy = d.y < 0.0, sz = d.z < 0.0;
  double ray_min = ((sx ? mx[0] : mn[0]) - o.x) * inv.x;
  double ray_max = ((sx ? mn[0] : mx[0]) - o.x) * inv.x;
  double y_min = ((sy ? mx[1] : mn[1]) - o.y) * inv.y;
  double y_max = ((sy ? mn[1] : mx[1]) - o.y) * inv.y;
  ray_min = fmax(ray_min, y_min);
  ray_max = fmin(ray_max, y_max);
  double z_min = ((sz ? mx[2] : mn[2]) - o.z) * inv.z;
  double z_max = ((sz ? mn[2] : mx[2]) - o.z) * inv.z;
  ray_min = fmax(ray_min, z_min);
  ray_max = fmin(ray_max, z_max);
  return fmax(ray_min, 0.0) <= ray_max;
}
// The same step on the f64 records: exactly the boxes the reference enters.
RT_DEV void trav_node_ref_exact(const DevScene &sc, d3 o, d3 d, d3 inv, Traversal &tv, uint32_t *stack, int stride) {
  const DevNodeD &n = sc.nodes_d[tv.cur & kRefIndexMask];
  bool hl = slab_hit_exact(n.lmin, n.lmax, o, d, inv);
  bool hr = slab_hit_exact(n.rmin, n.rmax, o, d, inv);
  uint32_t left = n.left, right = n.right;
  if (hl) {
    stack[tv.sp * stride] = right | (hr ? kRefFlag : 0u);
    tv.hitmask &= ~(1ull << tv.sp);
    tv.sp++;
    tv.cur = left;
  } else if (hr) {
    tv.cur = right;
  } else {
    trav_pop<true>(tv, stack, stride);
  }
}

// One internal-node step of the walk that follows the reference's visit rule: fetch the 64-byte
// node, test both child boxes, descend / push / pop.
RT_DEV void trav_node_ref(const DevScene &sc, const RayF &f, Traversal &tv, uint32_t *stack, int stride) {
  const float4 *p = reinterpret_cast<const float4 *>(&sc.nodes[tv.cur & kRefIndexMask]);
  float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
  uint32_t left = __float_as_uint(q3.x), right = __float_as_uint(q3.y);
  bool hl = slab_hit_f32(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, f);
  bool hr = slab_hit_f32(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, f);
  if (hl) {
    stack[tv.sp * stride] = right | (hr ? kRefFlag : 0u);
    tv.hitmask &= ~(1ull << tv.sp);
    tv.sp++;
    tv.cur = left;
  } else if (hr) {
    tv.cur = right;
  } else {
    trav_pop<true>(tv, stack, stride);
  }
}

// The node step on the 32-byte grid records (order-free walk only).
template <int SPILL>
RT_DEV void trav_node_q(const DevScene &sc, const RayQ &r, Traversal &tv, const StackRef &stack) {
  const uint4 *p = reinterpret_cast<const uint4 *>(&sc.nodes_q[tv.cur & kRefIndexMask]);
  uint4 w0 = p[0], w1 = p[1];
  // halves of the words, in the order of DevNodeQ
  float lminx = (float)(w0.x & 0xffffu), lminy = (float)(w0.x >> 16), lminz = (float)(w0.y & 0xffffu);
  float lmaxx = (float)(w0.y >> 16), lmaxy = (float)(w0.z & 0xffffu), lmaxz = (float)(w0.z >> 16);
  float rminx = (float)(w0.w & 0xffffu), rminy = (float)(w0.w >> 16), rminz = (float)(w1.x & 0xffffu);
  float rmaxx = (float)(w1.x >> 16), rmaxy = (float)(w1.y & 0xffffu), rmaxz = (float)(w1.y >> 16);
  uint32_t left = w1.z, right = w1.w;
  float tl, tr;
#if RT_PK_FMA
  bool hl = slab_hit_q2(f32x2{lminx, lmaxx}, f32x2{lminy, lmaxy}, f32x2{lminz, lmaxz}, r, tv.tcull, tl);
  bool hr = slab_hit_q2(f32x2{rminx, rmaxx}, f32x2{rminy, rmaxy}, f32x2{rminz, rmaxz}, r, tv.tcull, tr);
#else
  bool hl = slab_hit_q(lminx, lminy, lminz, lmaxx, lmaxy, lmaxz, r, tv.tcull, tl);
  bool hr = slab_hit_q(rminx, rminy, rminz, rmaxx, rmaxy, rmaxz, r, tv.tcull, tr);
#endif
  if (hl && hr) {
    bool swap = tr < tl;
    stack_push<SPILL>(tv, stack, swap ? left : right);
    tv.cur = swap ? right : left;
  } else if (hl || hr) {
    tv.cur = hl ? left : right;
  } else {
    trav_pop_fast<SPILL>(tv, stack);
  }
}

// The node step on the 64-byte wide records (order-free walk only): four box tests behind one memory round trip.
// A child that is a leaf goes straight into the leaf queue while there is room; of the others the nearest is
// entered and the rest are pushed.  Which children are entered in which order cannot change the result.
template <int SPILL>
RT_DEV void trav_node_w(const DevScene &sc, const RayQ &r, Traversal &tv, const StackRef &stack, uint32_t *leafq, int leafq_cap) {
  const uint4 *p = reinterpret_cast<const uint4 *>(&sc.nodes_w[tv.cur & kRefIndexMask]);
  const uint4 wx = p[0], wy = p[1], wz = p[2], rf = p[3];
  float t0, t1, t2, t3;
#define RT_WIDE_CHILD(X, Y, Z, T) slab_hit_qs(X, Y, Z, r, tv.tcull, T)
  bool h0 = RT_WIDE_CHILD(wx.x, wy.x, wz.x, t0);
  bool h1 = RT_WIDE_CHILD(wx.y, wy.y, wz.y, t1);
  bool h2 = RT_WIDE_CHILD(wx.z, wy.z, wz.z, t2) && rf.z != kRefNone;  // a node has at least two children
  bool h3 = RT_WIDE_CHILD(wx.w, wy.w, wz.w, t3) && rf.w != kRefNone;
#undef RT_WIDE_CHILD
  // leaves into the queue
#define RT_WIDE_LEAF(H, REF)                               \
  if (H && ((REF) & kRefLeaf) && tv.qn < leafq_cap) {      \
    leafq[tv.qn * 64] = (REF);                             \
    tv.qn++;                                               \
    H = false;                                             \
  }
  RT_WIDE_LEAF(h0, rf.x)
  RT_WIDE_LEAF(h1, rf.y)
  RT_WIDE_LEAF(h2, rf.z)
  RT_WIDE_LEAF(h3, rf.w)
#undef RT_WIDE_LEAF
  // the nearest of the rest: entry distances are >= 0, so their bit patterns order like unsigned integers; the two
  // lowest bits carry the slot
  const uint32_t k0 = h0 ? ((__float_as_uint(t0) & ~3u) | 0u) : 0xffffffffu;
  const uint32_t k1 = h1 ? ((__float_as_uint(t1) & ~3u) | 1u) : 0xffffffffu;
  const uint32_t k2 = h2 ? ((__float_as_uint(t2) & ~3u) | 2u) : 0xffffffffu;
  const uint32_t k3 = h3 ? ((__float_as_uint(t3) & ~3u) | 3u) : 0xffffffffu;
  const uint32_t kmin = min(min(k0, k1), min(k2, k3));
  if (kmin == 0xffffffffu) {
    trav_pop_fast<SPILL, kWideLdsLevels>(tv, stack);
    return;
  }
  const uint32_t ni = kmin & 3u;
  if (h0 && ni != 0u) { stack_push<SPILL, kWideLdsLevels>(tv, stack, rf.x); }
  if (h1 && ni != 1u) { stack_push<SPILL, kWideLdsLevels>(tv, stack, rf.y); }
  if (h2 && ni != 2u) { stack_push<SPILL, kWideLdsLevels>(tv, stack, rf.z); }
  if (h3 && ni != 3u) { stack_push<SPILL, kWideLdsLevels>(tv, stack, rf.w); }
  tv.cur = ni == 0u ? rf.x : ni == 1u ? rf.y : ni == 2u ? rf.z : rf.w;
}

// Queue the leaf in `cur` for the next leaf phase and move on (order-free walk).
template <int SPILL, int LDSL = kLdsStackLevels>
RT_DEV void trav_defer_leaf_fast(Traversal &tv, const StackRef &stack, uint32_t *leafq) {
  leafq[tv.qn * 64] = tv.cur;
  tv.qn++;
  trav_pop_fast<SPILL, LDSL>(tv, stack);
}

// Queue the leaf in `cur` for the next leaf phase and move on.
template <bool QUIRK>
RT_DEV void trav_defer_leaf(Traversal &tv, const uint32_t *stack, uint32_t *leafq, int stride) {
  leafq[tv.qn * stride] = tv.cur;
  if (QUIRK) {
    unsigned short h = (unsigned short)tv.sp;
    if (tv.qn == 0) tv.m01.x = h;
    else if (tv.qn == 1) tv.m01.y = h;
    else if (tv.qn == 2) tv.m23.x = h;
    else tv.m23.y = h;
  }
  tv.qn++;
  trav_pop<QUIRK>(tv, stack, stride);
}

// Two primitives hit at exactly the same t.  The reference tests leaves in depth-first,
// left-first order and lets a later triangle/quad replace an equal hit (inclusive interval)
// but not a later sphere (strict interval): the winner is the last triangle/quad in that
// order if there is one, else the first sphere.  `cand` is being tested now, `best` was
// accepted before -- in whatever order this traversal happened to reach them.
RT_DEV uint32_t leaf_rank(const DevScene &sc, uint32_t leaf) {
  uint32_t kind = (leaf >> kRefKindShift) & 3u, idx = leaf & kRefIndexMask;
  if (kind == RT_PRIM_TRIANGLE) return tri_record(sc, idx)->rank;
  if (kind == RT_PRIM_QUAD) return sc.quads[idx].rank;
  return sc.spheres[idx].rank;
}
RT_DEV bool tie_goes_to_later_test(const DevScene &sc, uint32_t cand, uint32_t best) {
  bool cand_sphere = ((cand >> kRefKindShift) & 3u) == RT_PRIM_SPHERE;
  bool best_sphere = ((best >> kRefKindShift) & 3u) == RT_PRIM_SPHERE;
  if (cand_sphere != best_sphere) return best_sphere;
  uint32_t rc = leaf_rank(sc, cand), rb = leaf_rank(sc, best);
  return cand_sphere ? rc < rb : rc > rb;
}

// The primitive test of the reference with the interval [kTMin, best.t]
// (triangle/quad inclusive, sphere strict).  `height` = how many of the
// current stack entries were already there when the leaf was reached.
template <bool QUIRK>
RT_DEV void trav_test_leaf(const DevScene &sc, d3 o, d3 d, Traversal &tv, uint32_t leaf, int height) {
  Ray r = {o, d, D3(0, 0, 0)};
  uint32_t kind = (leaf >> kRefKindShift) & 3u;
  uint32_t idx = leaf & kRefIndexMask;
  double t;
  bool ok;
  if (kind == RT_PRIM_TRIANGLE) {
    const double2 *p = reinterpret_cast<const double2 *>(tri_record(sc, idx));
    double2 a = p[0], b = p[1], c = p[2], dd = p[3], e = p[4];
    ok = triangle_hit(D3(a.x, a.y, b.x), D3(b.y, c.x, c.y), D3(dd.x, dd.y, e.x), r, kTMin, tv.best.t, t);
  } else if (kind == RT_PRIM_QUAD) {
    ok = quad_hit(sc.quads[idx], r, kTMin, tv.best.t, t);
  } else if (QUIRK) {
    ok = sphere_hit(sc.spheres[idx], r, kTMin, tv.best.t, t);
  } else {
    // leaves arrive out of the reference's order: let an equal t through to the tie rule below
    ok = sphere_hit(sc.spheres[idx], r, kTMin, kF64Max, t) && t <= tv.best.t;
  }
  if (!QUIRK && ok && t == tv.best.t && tv.best.ref != kRefNone) ok = tie_goes_to_later_test(sc, leaf, tv.best.ref);
  if (ok) {
    tv.best.t = t;
    tv.best.ref = leaf;
    if (QUIRK) tv.hitmask |= (height >= 64) ? ~0ull : ((1ull << height) - 1ull);
    else tv.tcull = __double2float_ru(t);
  }
}

// The order-free walk's leaf phase tests one kind of primitive at a time: a wave pays for the quad and
// sphere code once per phase, not in every queue slot in which some lane happens to hold one.
RT_DEV void trav_accept_fast(const DevScene &sc, Traversal &tv, uint32_t leaf, bool ok, double t) {
  if (ok && t == tv.best.t && tv.best.ref != kRefNone) ok = tie_goes_to_later_test(sc, leaf, tv.best.ref);
  if (ok) {
    tv.best.t = t;
    tv.best.ref = leaf;
    tv.tcull = __double2float_ru(t);
  }
}
RT_DEV void trav_test_triangle_fast(const DevScene &sc, d3 o, d3 d, Traversal &tv, uint32_t leaf) {
  Ray r = {o, d, D3(0, 0, 0)};
  const double2 *p = reinterpret_cast<const double2 *>(tri_record(sc, leaf & kRefIndexMask));
  double2 a = p[0], b = p[1], c = p[2], dd = p[3], e = p[4];
  double t;
  bool ok = triangle_hit(D3(a.x, a.y, b.x), D3(b.y, c.x, c.y), D3(dd.x, dd.y, e.x), r, kTMin, tv.best.t, t);
  trav_accept_fast(sc, tv, leaf, ok, t);
}
RT_DEV void trav_test_quad_fast(const DevScene &sc, d3 o, d3 d, Traversal &tv, uint32_t leaf) {
  Ray r = {o, d, D3(0, 0, 0)};
  double t;
  bool ok = quad_hit(sc.quads[leaf & kRefIndexMask], r, kTMin, tv.best.t, t);
  trav_accept_fast(sc, tv, leaf, ok, t);
}
RT_DEV void trav_test_sphere_fast(const DevScene &sc, d3 o, d3 d, Traversal &tv, uint32_t leaf) {
  Ray r = {o, d, D3(0, 0, 0)};
  double t;
  // leaves arrive out of the reference's order: let an equal t through to the tie rule
  bool ok = sphere_hit(sc.spheres[leaf & kRefIndexMask], r, kTMin, kF64Max, t) && t <= tv.best.t;
  trav_accept_fast(sc, tv, leaf, ok, t);
}

}  // namespace rtd
