/*
 * rt_oracle.c -- CPU restatement of scott223/raytracer's per-pixel
 * Monte-Carlo rendering path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker for the HIP path.  Only tests/, the smoke check
 * and the benchmark's CPU-baseline leg may load it; nothing in the product
 * library links, imports or executes it.
 *
 * The reference is a Rust crate and no Rust toolchain exists where this was
 * written, so the reference itself could not be run: each function below
 * restates one reference function line by line (file:line cited) in plain
 * f64 C, compiled with -ffp-contract=off so that no multiply-add is fused
 * (Rust never fuses implicitly).
 *
 * Pinned by the reference's own known-answer tests (see tests/): the AABB
 * slab test (aabb.rs:321-335), Ray::new / at (ray.rs:73-92), Color clamp /
 * gamma / to_rgb (color.rs:256-286), Vec3 algebra (vec3.rs:465-526) and the
 * sphere hit (elements/tests.rs:101-115).
 *
 * PARITY UNPINNED (third-party crates whose sources are not in the
 * reference tree; restated from their published algorithms, no reference
 * test fixes their outputs):
 *   - sobol_burley 0.5.0  `sample(index, dim, seed)`       -> sobol_sample()
 *   - rand 0.8.5 SmallRng (xoshiro256++, SplitMix64 seeding), gen_range,
 *     gen::<f64>, SliceRandom::choose                       -> rng_*()
 *   - rand_distr 0.4.2 UnitSphere                           -> random_unit_vector()
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/rt_hip.h"

#define RT_EPSILON 2.220446049250313e-16 /* f64::EPSILON */
#define RT_PI 3.14159265358979323846     /* std::f64::consts::PI */
#define RT_F64_MAX 1.7976931348623157e308

/* ===================================================== linalg/vec3.rs */
typedef struct { double x, y, z; } v3;

static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 fromabi(RtVec3 a) { return V(a.x, a.y, a.z); }
static inline v3 v_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }   /* vec3.rs:224-234 */
static inline v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }   /* vec3.rs:237-247 */
static inline v3 v_mulf(v3 a, double q) { return V(a.x * q, a.y * q, a.z * q); }     /* vec3.rs:263-273 */
static inline v3 v_divf(v3 a, double q) { return V(a.x / q, a.y / q, a.z / q); }     /* vec3.rs:289-299 */
static inline v3 v_neg(v3 a) { return V(-a.x, -a.y, -a.z); }                         /* vec3.rs:211-221 */
static inline double v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* vec3.rs:146-148 */
static inline v3 v_cross(v3 a, v3 b) {                                               /* vec3.rs:152-158 */
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double v_length_squared(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; } /* vec3.rs:133-135 */
/* vec3.rs:116-128: distance to the origin; powf(d, 2.0) is d*d */
static inline double v_length(v3 a) {
  double dx = (a.x - 0.0) * (a.x - 0.0), dy = (a.y - 0.0) * (a.y - 0.0), dz = (a.z - 0.0) * (a.z - 0.0);
  return sqrt(dx + dy + dz);
}
static inline v3 v_normalized(v3 a) { /* vec3.rs:163-166 */
  double l = v_length(a);
  return V(a.x / l, a.y / l, a.z / l);
}

/* ================================================== render/color.rs */
typedef v3 col; /* r,g,b in x,y,z */

static inline int c_has_nan(col c) { return isnan(c.x) || isnan(c.y) || isnan(c.z); } /* color.rs:56-58 */
static inline col c_mul(col a, col b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }  /* color.rs:155-165 */
/* color.rs:29-35: r.min(1.0).max(0.0); Rust min/max return the non-NaN operand, as fmin/fmax do */
static inline col c_clamp(col c) {
  return V(fmax(fmin(c.x, 1.0), 0.0), fmax(fmin(c.y, 1.0), 0.0), fmax(fmin(c.z, 1.0), 0.0));
}
static inline col c_linear_to_gamma(col c, double correction) { /* color.rs:38-44 */
  return V(pow(c.x, 1.0 / correction), pow(c.y, 1.0 / correction), pow(c.z, 1.0 / correction));
}
static inline double c_illuminance(col c) { return 0.2126 * c.x + 0.7152 * c.y + 0.0722 * c.z; } /* color.rs:66-69 */
static inline double c_illuminance_approx(col c) { /* color.rs:72-74 */
  return (c.x + c.x + c.z + c.y + c.y + c.y) / 6.0;
}

/* =================================================== render/ray.rs */
typedef struct {
  v3 origin, direction, inv_direction;
  int sign_x, sign_y, sign_z;
} ray_t;

static inline ray_t ray_new(v3 origin, v3 direction) { /* ray.rs:35-48 */
  ray_t r;
  v3 d = v_normalized(direction);
  r.origin = origin;
  r.direction = d;
  r.inv_direction = V(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
  r.sign_x = d.x < 0.0;
  r.sign_y = d.y < 0.0;
  r.sign_z = d.z < 0.0;
  return r;
}
static inline v3 ray_at(const ray_t *r, double t) { return v_add(r->origin, v_mulf(r->direction, t)); } /* ray.rs:51-53 */

typedef struct { double min, max; } interval_t;
static inline int interval_contains(const interval_t *i, double x) { return i->min <= x && x <= i->max; } /* interval.rs:31-33 */

/* ===================================================== bvh/aabb.rs */
/* aabb.rs:187-204; self[0]=min, self[1]=max (aabb.rs:231-241).  f64::max/min
 * ignore a NaN operand exactly like fmax/fmin. */
static inline int aabb_hit(const double *bmin, const double *bmax, const ray_t *ray) {
  const double *b[2] = {bmin, bmax};
  double ray_min = (b[ray->sign_x][0] - ray->origin.x) * ray->inv_direction.x;
  double ray_max = (b[1 - ray->sign_x][0] - ray->origin.x) * ray->inv_direction.x;
  double y_min = (b[ray->sign_y][1] - ray->origin.y) * ray->inv_direction.y;
  double y_max = (b[1 - ray->sign_y][1] - ray->origin.y) * ray->inv_direction.y;
  ray_min = fmax(ray_min, y_min);
  ray_max = fmin(ray_max, y_max);
  double z_min = (b[ray->sign_z][2] - ray->origin.z) * ray->inv_direction.z;
  double z_max = (b[1 - ray->sign_z][2] - ray->origin.z) * ray->inv_direction.z;
  ray_min = fmax(ray_min, z_min);
  ray_max = fmin(ray_max, z_max);
  return fmax(ray_min, 0.0) <= ray_max;
}

/* ============================================= elements: hit records */
typedef struct { /* elements/mod.rs:31-37 */
  double t;
  v3 point, normal;
  int front_face;
  int material;
  int element; /* not in the reference: which objects[] entry was hit */
} hit_t;

typedef struct {
  uint64_t rays, node_visits, box_tests, leaf_tests;
  uint64_t rays_by_depth[RT_MAX_DEPTH_STATS];
  uint64_t samples;
} counters_t;

/* triangle.rs:76-117 (Moller-Trumbore with back-face culling) */
static int triangle_hit(const RtTriangle *tr, const ray_t *ray, const interval_t *ray_t_, hit_t *out) {
  v3 v0v1 = fromabi(tr->v0v1), v0v2 = fromabi(tr->v0v2), nrm = fromabi(tr->normal);
  v3 pvec = v_cross(ray->direction, v0v2);
  double det = v_dot(v0v1, pvec);
  if (det < RT_EPSILON) return 0;
  double inv_det = 1. / det;
  v3 tvec = v_sub(ray->origin, fromabi(tr->v0));
  double u = v_dot(tvec, pvec) * inv_det;
  if (!(0.0 <= u && u <= 1.0)) return 0;
  v3 qvec = v_cross(tvec, v0v1);
  double v = v_dot(ray->direction, qvec) * inv_det;
  if (v < 0.0 || u + v > 1.0) return 0;
  double t = v_dot(v0v2, qvec) * inv_det;
  if (!interval_contains(ray_t_, t)) return 0;
  out->t = t;
  out->normal = nrm;
  out->point = ray_at(ray, t);
  out->front_face = v_dot(ray->direction, nrm) < 0.0;
  out->material = tr->material;
  return 1;
}

/* quad.rs:108-145 */
static int quad_hit(const RtQuad *q, const ray_t *ray, const interval_t *ray_t_, hit_t *out) {
  v3 normal = fromabi(q->normal);
  double denom = v_dot(normal, ray->direction);
  if (fabs(denom) < RT_EPSILON) return 0;
  double t = (q->d - v_dot(normal, ray->origin)) / denom;
  if (!interval_contains(ray_t_, t)) return 0;
  v3 intersection = ray_at(ray, t);
  v3 p = v_sub(intersection, fromabi(q->q));
  v3 w = fromabi(q->w);
  double alpha = v_dot(w, v_cross(p, fromabi(q->v)));
  double beta = v_dot(w, v_cross(fromabi(q->u), p));
  if (!(0.0 <= alpha && alpha <= 1.0) || !(0.0 <= beta && beta <= 1.0)) return 0; /* quad.rs:92-99 */
  out->t = t;
  out->normal = normal;
  out->point = intersection;
  out->front_face = v_dot(ray->direction, normal) < 0.0;
  out->material = q->material;
  return 1;
}

/* sphere.rs:94-145 */
static int sphere_hit(const RtSphere *s, const ray_t *ray, const interval_t *ray_t_, hit_t *out) {
  v3 center = fromabi(s->center);
  v3 l = v_sub(ray->origin, center);
  double a = v_dot(ray->direction, ray->direction);
  double b = 2.0 * v_dot(ray->direction, l);
  double c = v_dot(l, l) - (s->radius * s->radius);
  double discr = b * b - 4.0 * a * c;
  if (discr >= 0.0) {
    double q = (b > 0.0) ? -0.5 * (b + sqrt(discr)) : -0.5 * (b - sqrt(discr));
    double first_root = q / a;
    double second_root = c / q;
    if (first_root < 0.0 && second_root < 0.0) return 0;
    double nearest_root = (first_root < second_root) ? first_root : second_root;
    if (nearest_root < ray_t_->max && nearest_root > ray_t_->min) {
      v3 p = ray_at(ray, nearest_root);
      v3 outward_normal = v_normalized(v_divf(v_sub(p, center), s->radius));
      int front_face = v_dot(ray->direction, outward_normal) < 0.0;
      out->t = nearest_root;
      out->normal = front_face ? outward_normal : v_neg(outward_normal);
      out->point = p;
      out->front_face = front_face;
      out->material = s->material;
      return 1;
    }
  }
  return 0;
}

/* elements/mod.rs:100-107 */
static int element_hit(const RtScene *sc, int element, const ray_t *ray, const interval_t *ray_t_, hit_t *out) {
  RtElementRef e = sc->elements[element];
  int r;
  switch (e.kind) {
    case RT_PRIM_SPHERE: r = sphere_hit(&sc->spheres[e.index], ray, ray_t_, out); break;
    case RT_PRIM_QUAD: r = quad_hit(&sc->quads[e.index], ray, ray_t_, out); break;
    default: r = triangle_hit(&sc->triangles[e.index], ray, ray_t_, out); break;
  }
  if (r) out->element = element;
  return r;
}

/* ======================================== bvh/bvh_node_sah.rs:180-253 */
static int node_hit(const RtScene *sc, int node, const ray_t *ray, interval_t *ray_t_, hit_t *out, counters_t *cn) {
  const RtNode *n = &sc->nodes[node];
  if (n->is_leaf) {
    cn->leaf_tests++;
    return element_hit(sc, n->element_index, ray, ray_t_, out);
  }
  cn->node_visits++;
  cn->box_tests++;
  if (aabb_hit(n->child_l_min, n->child_l_max, ray)) {
    hit_t lh;
    if (node_hit(sc, n->child_l_index, ray, ray_t_, &lh, cn)) {
      ray_t_->max = lh.t;
      hit_t rh;
      /* the right child's box is NOT tested after a left hit (bvh_node_sah.rs:215-216) */
      if (node_hit(sc, n->child_r_index, ray, ray_t_, &rh, cn)) {
        *out = rh;
      } else {
        *out = lh;
      }
      return 1;
    }
    cn->box_tests++;
    if (aabb_hit(n->child_r_min, n->child_r_max, ray)) return node_hit(sc, n->child_r_index, ray, ray_t_, out, cn);
    return 0;
  }
  cn->box_tests++;
  if (aabb_hit(n->child_r_min, n->child_r_max, ray)) return node_hit(sc, n->child_r_index, ray, ray_t_, out, cn);
  return 0;
}

/* bvh_node_sah.rs:162-165 */
static int bvh_hit(const RtScene *sc, const ray_t *ray, interval_t *ray_t_, hit_t *out, counters_t *cn) {
  return node_hit(sc, 0, ray, ray_t_, out, cn);
}

/* ============================================================== RNG
 * rand 0.8.5 SmallRng on 64-bit targets = xoshiro256++; seed_from_u64 fills
 * the state with SplitMix64 outputs.  PARITY UNPINNED (see file header). */
typedef struct {
  int mode;      /* RT_RNG_REF: xoshiro state s[]; RT_RNG_CTR: SplitMix64 counter in s[0] */
  uint64_t s[4];
} rng_t;

static inline uint64_t splitmix64_mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

static void rng_seed_from_u64(rng_t *r, uint64_t state) { /* integrator.rs:263 */
  r->mode = RT_RNG_REF;
  for (int i = 0; i < 4; i++) {
    state += 0x9e3779b97f4a7c15ULL;
    r->s[i] = splitmix64_mix(state);
  }
}

/* Counter mode: one SplitMix64 stream per path.  The start of the stream is
 * a hash of (pixel index in the full image, batch, sample, seed) so the
 * stream does not depend on tiling, thread count or device count. */
static void rng_seed_path(rng_t *r, uint64_t pixel_index, uint64_t batch, uint64_t sample, uint64_t seed) {
  r->mode = RT_RNG_CTR;
  uint64_t k = splitmix64_mix(pixel_index + 0x9e3779b97f4a7c15ULL);
  k = splitmix64_mix(k ^ (batch * 0xd1342543de82ef95ULL + 0x632be59bd9b4e019ULL));
  k = splitmix64_mix(k ^ (sample * 0xaf251af3b0f025b5ULL + 0x9e3779b97f4a7c15ULL));
  r->s[0] = splitmix64_mix(k ^ seed);
  r->s[1] = r->s[2] = r->s[3] = 0;
}

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static inline uint64_t rng_next_u64(rng_t *r) {
  if (r->mode == RT_RNG_CTR) {
    r->s[0] += 0x9e3779b97f4a7c15ULL;
    return splitmix64_mix(r->s[0]);
  }
  uint64_t *s = r->s;
  uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
  uint64_t t = s[1] << 17;
  s[2] ^= s[0];
  s[3] ^= s[1];
  s[1] ^= s[2];
  s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl64(s[3], 45);
  return result;
}
static inline uint32_t rng_next_u32(rng_t *r) { return (uint32_t)(rng_next_u64(r) >> 32); }

/* rand UniformFloat<f64>::sample_single: 52 random mantissa bits -> [1,2) - 1,
 * then value*scale + low, retried while the result is >= high. */
static inline double rng_gen_range(rng_t *r, double low, double high) {
  double scale = high - low;
  for (;;) {
    uint64_t bits = (rng_next_u64(r) >> 12) | 0x3ff0000000000000ULL;
    double value1_2;
    memcpy(&value1_2, &bits, 8);
    double value0_1 = value1_2 - 1.0;
    double res = value0_1 * scale + low;
    if (res < high) return res;
  }
}
/* rand Standard for f64: 53 bits * 2^-53 */
static inline double rng_gen_f64(rng_t *r) { return (double)(rng_next_u64(r) >> 11) * (1.0 / 9007199254740992.0); }
/* rand gen_range(0..n) for u32 (SliceRandom::choose): widening multiply with
 * the rejection zone (n << lzcnt(n)) - 1 */
static inline uint32_t rng_gen_index(rng_t *r, uint32_t n) {
  uint32_t zone = (n << __builtin_clz(n)) - 1u;
  for (;;) {
    uint32_t v = rng_next_u32(r);
    uint64_t m = (uint64_t)v * (uint64_t)n;
    uint32_t lo = (uint32_t)m;
    if (lo <= zone) return (uint32_t)(m >> 32);
  }
}

/* rand_distr UnitSphere (Marsaglia 1972), vec3.rs:35-38, 51-53 */
static v3 random_unit_vector(rng_t *r) {
  for (;;) {
    double x1 = rng_gen_range(r, -1.0, 1.0);
    double x2 = rng_gen_range(r, -1.0, 1.0);
    double sum = x1 * x1 + x2 * x2;
    if (sum >= 1.0) continue;
    double factor = 2.0 * sqrt(1.0 - sum);
    return V(x1 * factor, x2 * factor, 1.0 - 2.0 * sum);
  }
}
/* vec3.rs:41-48 */
static v3 random_in_unit_disk(rng_t *r) {
  for (;;) {
    double a = rng_gen_range(r, -1.0, 1.0);
    double b = rng_gen_range(r, -1.0, 1.0);
    v3 v = V(a, b, 0.0);
    if (v_length_squared(v) < 1.0) return v;
  }
}
/* vec3.rs:69-79 */
static v3 random_cosine_direction(rng_t *r) {
  double r1 = rng_gen_range(r, 0.0, 1.0);
  double r2 = rng_gen_range(r, 0.0, 1.0);
  double phi = 2. * RT_PI * r1;
  double x = cos(phi) * sqrt(r2);
  double y = sin(phi) * sqrt(r2);
  double z = sqrt(1. - r2);
  return V(x, y, z);
}

/* ===================================================== linalg/onb.rs */
typedef struct { v3 u, v, w; } onb_t;
static onb_t onb_build_from_w(v3 w) { /* onb.rs:31-45 */
  onb_t o;
  v3 unit_w = v_normalized(w);
  v3 a = (fabs(unit_w.x) > 0.9) ? V(0., 1., 0.) : V(1.0, 0., 0.);
  v3 v = v_normalized(v_cross(unit_w, a));
  v3 u = v_cross(unit_w, v);
  o.u = u;
  o.v = v;
  o.w = unit_w;
  return o;
}
static inline v3 onb_local_vec(const onb_t *o, v3 vec) { /* onb.rs:27-29 */
  return v_add(v_add(v_mulf(o->u, vec.x), v_mulf(o->v, vec.y)), v_mulf(o->w, vec.z));
}

/* ============================ elements: pdf_value / random (attractors) */
/* quad.rs:155-167 */
static double quad_pdf_value(const RtQuad *q, v3 origin, v3 direction) {
  ray_t r = ray_new(origin, direction);
  interval_t iv = {0.001, INFINITY};
  hit_t h;
  if (quad_hit(q, &r, &iv, &h)) {
    double distance_squared = h.t * h.t * v_length_squared(direction);
    double cosine = fabs(v_dot(direction, h.normal)) / v_length(direction);
    return distance_squared / (cosine * q->area);
  }
  return 0.0;
}
/* quad.rs:172-178 */
static v3 quad_random(const RtQuad *q, v3 origin, rng_t *rng) {
  double r0 = rng_gen_range(rng, 0.0, 1.0);
  double r1 = rng_gen_range(rng, 0.0, 1.0);
  v3 p = v_add(v_add(fromabi(q->q), v_mulf(fromabi(q->u), r0)), v_mulf(fromabi(q->v), r1));
  return v_sub(p, origin);
}
/* sphere.rs:154-169 */
static double sphere_pdf_value(const RtSphere *s, v3 origin, v3 direction) {
  ray_t r = ray_new(origin, direction);
  interval_t iv = {0.001, INFINITY};
  hit_t h;
  if (sphere_hit(s, &r, &iv, &h)) {
    double cos_theta_max = sqrt(1. - s->radius * s->radius / v_length_squared(v_sub(fromabi(s->center), origin)));
    double solid_angle = 2.0 * RT_PI * (1.0 - cos_theta_max);
    return 1. / solid_angle;
  }
  return 0.;
}
/* sphere.rs:184-194 */
static v3 random_to_sphere(double radius, double distance_squared, rng_t *rng) {
  double r1 = rng_gen_range(rng, 0.0, 1.0);
  double r2 = rng_gen_range(rng, 0.0, 1.0);
  double z = 1.0 + r2 * (sqrt(1.0 - radius * radius / distance_squared) - 1.0);
  double phi = 2.0 * RT_PI * r1;
  double x = cos(phi) * sqrt(1.0 - z * z);
  double y = sin(phi) * sqrt(1.0 - z * z);
  return V(x, y, z);
}
/* sphere.rs:173-181 */
static v3 sphere_random(const RtSphere *s, v3 origin, rng_t *rng) {
  v3 direction = v_sub(fromabi(s->center), origin);
  onb_t uvw = onb_build_from_w(direction);
  return onb_local_vec(&uvw, random_to_sphere(s->radius, v_length_squared(direction), rng));
}
/* elements/mod.rs:117-131; triangle.rs:123-129 */
static double element_pdf_value(const RtScene *sc, int element, v3 origin, v3 direction) {
  RtElementRef e = sc->elements[element];
  switch (e.kind) {
    case RT_PRIM_SPHERE: return sphere_pdf_value(&sc->spheres[e.index], origin, direction);
    case RT_PRIM_QUAD: return quad_pdf_value(&sc->quads[e.index], origin, direction);
    default: return 0.0;
  }
}
static v3 element_random(const RtScene *sc, int element, v3 origin, rng_t *rng) {
  RtElementRef e = sc->elements[element];
  switch (e.kind) {
    case RT_PRIM_SPHERE: return sphere_random(&sc->spheres[e.index], origin, rng);
    case RT_PRIM_QUAD: return quad_random(&sc->quads[e.index], origin, rng);
    default: return V(1.0, 0.0, 0.0);
  }
}

/* ==================================================== render/pdf.rs */
/* CosinePDF::value, pdf.rs:159-162 */
static double cosine_pdf_value(const onb_t *uvw, v3 direction) {
  double cosine_theta = v_dot(v_normalized(direction), uvw->w);
  return fmax(0.0, cosine_theta / RT_PI);
}
/* CosinePDF::generate, pdf.rs:164-166 */
static v3 cosine_pdf_generate(const onb_t *uvw, rng_t *rng) { return onb_local_vec(uvw, random_cosine_direction(rng)); }
/* HittablePDF::value, pdf.rs:98-107 */
static double hittable_pdf_value(const RtScene *sc, v3 origin, v3 direction) {
  double weight = 1.0 / (double)sc->n_attractors;
  double sum = 0.0;
  for (int i = 0; i < sc->n_attractors; i++) sum += weight * element_pdf_value(sc, sc->attractors[i], origin, direction);
  return sum;
}
/* HittablePDF::generate, pdf.rs:110-115 */
static v3 hittable_pdf_generate(const RtScene *sc, v3 origin, rng_t *rng) {
  uint32_t k = rng_gen_index(rng, (uint32_t)sc->n_attractors);
  return element_random(sc, sc->attractors[k], origin, rng);
}
/* MixedPDF, pdf.rs:57-69 with ratio 0.5 (integrator.rs:384-389) */
static double mixed_pdf_value(const RtScene *sc, v3 origin, const onb_t *uvw, v3 direction) {
  const double ratio = 0.5;
  return ratio * hittable_pdf_value(sc, origin, direction) + (1. - ratio) * cosine_pdf_value(uvw, direction);
}
static v3 mixed_pdf_generate(const RtScene *sc, v3 origin, const onb_t *uvw, rng_t *rng) {
  const double ratio = 0.5;
  double r = rng_gen_range(rng, 0.0, 1.0);
  if (r < ratio) return hittable_pdf_generate(sc, origin, rng);
  return cosine_pdf_generate(uvw, rng);
}

/* ================================================ materials/metal.rs */
static inline v3 reflect_vector(v3 v, v3 n) { return v_sub(v, v_mulf(n, 2.0 * v_dot(v, n))); } /* metal.rs:56-58 */
static v3 refract_ray(v3 uv, v3 n, double etai_over_etat) {                                    /* metal.rs:62-68 */
  v3 minus_uv = v_mulf(uv, -1.0);
  double cos_theta = fmin(v_dot(n, minus_uv), 1.0);
  v3 r_out_perp = v_mulf(v_add(uv, v_mulf(n, cos_theta)), etai_over_etat);
  v3 r_out_parallel = v_mulf(n, sqrt(fabs(1.0 - v_length_squared(r_out_perp))) * -1.0);
  return v_add(r_out_perp, r_out_parallel);
}
static double reflectance(double cosine, double ref_idx) { /* metal.rs:72-76; powi(5) */
  double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
  r0 = r0 * r0;
  double x = 1.0 - cosine;
  double x2 = x * x;
  return r0 + (1.0 - r0) * (x2 * x2 * x);
}

/* ==================================== render/integrator.rs:344-470 */
static col ray_color(const RtScene *sc, const ray_t *ray, int depth, int max_depth, rng_t *rng, counters_t *cn) {
  if (depth == 0) return V(0.0, 0.0, 0.0);
  interval_t iv = {0.0001, RT_F64_MAX};
  hit_t hit;
  cn->rays++;
  {
    int bounce = max_depth - depth;
    if (bounce < RT_MAX_DEPTH_STATS) cn->rays_by_depth[bounce]++;
  }
  if (!bvh_hit(sc, ray, &iv, &hit, cn)) return V(0.0, 0.0, 0.0);

  const RtMaterial *m = &sc->materials[hit.material];
  col albedo = V(m->albedo[0], m->albedo[1], m->albedo[2]);

  /* emitted: diffuse_light.rs:20-29 */
  if (m->kind == RT_MAT_DIFFUSE_LIGHT && hit.front_face) return albedo;

  /* scatter: lambertian.rs:26-43 */
  if (m->kind == RT_MAT_LAMBERTIAN) {
    onb_t uvw = onb_build_from_w(hit.normal); /* CosinePDF::new, pdf.rs:150-154 */
    ray_t scattered;
    double pdf_val;
    if (sc->n_attractors > 0) {
      scattered = ray_new(hit.point, mixed_pdf_generate(sc, hit.point, &uvw, rng));
      pdf_val = mixed_pdf_value(sc, hit.point, &uvw, scattered.direction);
    } else {
      scattered = ray_new(hit.point, cosine_pdf_generate(&uvw, rng));
      pdf_val = cosine_pdf_value(&uvw, scattered.direction);
    }
    col target = ray_color(sc, &scattered, depth - 1, max_depth, rng, cn);
    double scattering_pdf = cosine_pdf_value(&uvw, scattered.direction);
    col c = v_divf(v_mulf(c_mul(albedo, target), scattering_pdf), pdf_val);
    if (c_has_nan(c)) c = V(0., 0., 0.);
    return c;
  }

  /* reflect: metal.rs:21-53 */
  if (m->kind == RT_MAT_METAL) {
    v3 new_direction =
        v_add(reflect_vector(v_normalized(ray->direction), hit.normal), v_mulf(random_unit_vector(rng), m->param));
    if (v_dot(hit.normal, new_direction) > 0.0) {
      ray_t reflected = ray_new(hit.point, new_direction);
      col target = ray_color(sc, &reflected, depth - 1, max_depth, rng, cn);
      return c_mul(albedo, target);
    }
    return V(0.0, 0.0, 0.0);
  }

  /* refract: dielectric.rs:24-66 */
  if (m->kind == RT_MAT_DIELECTRIC) {
    col white = V(1.0, 1.0, 1.0);
    double refraction_ratio = hit.front_face ? 1.0 / m->param : m->param;
    v3 unit_direction = v_normalized(ray->direction);
    v3 minus_unit_direction = v_mulf(unit_direction, -1.0);
    double cos_theta = fmin(v_dot(hit.normal, minus_unit_direction), 1.0);
    double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    int cannot_refract = refraction_ratio * sin_theta > 1.0;
    v3 direction;
    /* the random number is drawn only when refraction is possible (|| short-circuit) */
    if (cannot_refract || reflectance(cos_theta, refraction_ratio) > rng_gen_f64(rng)) {
      direction = reflect_vector(unit_direction, hit.normal);
    } else {
      direction = refract_ray(unit_direction, hit.normal, refraction_ratio);
    }
    ray_t next = ray_new(hit.point, direction);
    col target = ray_color(sc, &next, depth - 1, max_depth, rng, cn);
    return c_mul(white, target);
  }

  return V(0.0, 0.0, 0.0); /* e.g. the back face of a light */
}

/* ============================================== sobol_burley::sample
 * PARITY UNPINNED.  Burley 2020, "Practical Hash-based Owen Scrambling":
 * the sample index is shuffled by a seed-keyed nested-uniform (Owen) hash
 * applied in bit-reversed space, the low 16 bits of the shuffled index pick
 * direction vectors, and the result is Owen-scrambled per dimension.  Dim 0
 * is van der Corput, dim 1 the standard second Sobol dimension
 * (v_k = v_{k-1} ^ (v_{k-1} >> 1)). */
static inline uint32_t reverse_bits32(uint32_t x) {
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
  x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
  return (x >> 16) | (x << 16);
}
static inline uint32_t hash_u32(uint32_t n) { /* avalanche used to derive keys */
  n ^= 0xe6fe3beb;
  n ^= n >> 16;
  n *= 0x7feb352d;
  n ^= n >> 15;
  n *= 0x846ca68b;
  n ^= n >> 16;
  return n;
}
/* Owen scramble of a value held in bit-reversed order */
static inline uint32_t owen_scramble_rev(uint32_t n, uint32_t key) {
  n ^= n * 0x3d20adea;
  n += key;
  n *= (key >> 16) | 1;
  n ^= n * 0x05526c56;
  n ^= n * 0x53a22864;
  return n;
}
static float sobol_sample(uint32_t index, uint32_t dim, uint32_t seed) {
  uint32_t shuffled = reverse_bits32(owen_scramble_rev(reverse_bits32(index), hash_u32(seed)));
  uint32_t idx = shuffled & 0xffffu;
  uint32_t x = 0;
  if (dim == 0) {
    x = reverse_bits32(idx);
  } else {
    uint32_t v = 0x80000000u;
    for (int k = 0; k < 16; k++) {
      if (idx & (1u << k)) x ^= v;
      v ^= v >> 1;
    }
  }
  uint32_t key = hash_u32(seed ^ hash_u32(dim + 1u));
  x = reverse_bits32(owen_scramble_rev(reverse_bits32(x), key));
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

/* ================================================= render/filter.rs */
static double mitchell_1d(double x, double b, double c) { /* filter.rs:45-61 */
  double fx = fabs(x) * 2.0;
  if (fx < 1.0) {
    return ((12.0 - 9.0 * b - 6.0 * c) * fx * fx * fx + (-18.0 + 12.0 * b + 6.0 * c) * fx * fx + (6.0 - 2.0 * b)) *
           (1.0 / 6.0);
  } else if (fx < 2.0) {
    return ((-b - 6.0 * c) * fx * fx * fx + (6.0 * b + 30.0 * c) * fx * fx + (-12.0 * b - 48.0 * c) * fx +
            (8.0 * b + 24.0 * c)) *
           (1.0 / 6.0);
  }
  return 0.0;
}
static double filter_evaluate(const RtConfig *cfg, double sx, double sy) { /* filter.rs:33-42, 65-67 */
  double inv_width = 1.0 / cfg->pixel_radius, inv_height = 1.0 / cfg->pixel_radius;
  return mitchell_1d(sx * inv_width, cfg->filter_b, cfg->filter_c) * mitchell_1d(sy * inv_height, cfg->filter_b, cfg->filter_c);
}

/* ================================================= render/camera.rs */
static ray_t get_prime_ray(const RtCamera *cam, int x, int y, rng_t *rng, double sx, double sy) { /* camera.rs:95-120 */
  v3 center = fromabi(cam->camera_center);
  v3 ray_origin;
  if (cam->defocus_angle <= 0.0) {
    ray_origin = center;
  } else {
    v3 p = random_in_unit_disk(rng);
    ray_origin = v_add(v_add(center, v_mulf(fromabi(cam->defocus_disk_u), p.x)), v_mulf(fromabi(cam->defocus_disk_v), p.y));
  }
  v3 pixel_loc = v_add(v_add(fromabi(cam->pixel00_loc), v_mulf(fromabi(cam->pixel_delta_u), (double)x)),
                       v_mulf(fromabi(cam->pixel_delta_v), (double)y));
  v3 pixel_sample = v_add(pixel_loc, V(sx, sy, 0.));
  v3 ray_direction = v_sub(pixel_sample, ray_origin);
  return ray_new(ray_origin, ray_direction);
}

/* Sobol offset for sample i of the pixel, integrator.rs:275, 286-291 */
static void sample_offset(const RtConfig *cfg, int x, int y, int i, double *sx, double *sy) {
  uint64_t pixel_num = (uint64_t)x * (uint64_t)y; /* sic: product */
  uint32_t seed = (uint32_t)pixel_num;
  *sx = ((double)sobol_sample((uint32_t)i, 0, seed) * cfg->pixel_radius) - cfg->pixel_radius / 2.;
  *sy = ((double)sobol_sample((uint32_t)i, 1, seed) * cfg->pixel_radius) - cfg->pixel_radius / 2.;
}

/* ================================================== render/stats.rs */
typedef struct {
  col *colors;
  double *weights;
  int n;
} stats_t;
static col stats_mean(const stats_t *s) { /* stats.rs:21-30 */
  col sum = V(0.0, 0.0, 0.0);
  for (int k = 0; k < s->n; k++) sum = v_add(sum, v_mulf(s->colors[k], s->weights[k]));
  double w = 0.0;
  for (int k = 0; k < s->n; k++) w += s->weights[k];
  return v_divf(sum, w);
}
static double stats_variance(const stats_t *s) { /* stats.rs:32-47 */
  if (s->n == 1) return 0.0;
  double mean = c_illuminance_approx(stats_mean(s));
  double nominator = 0.0;
  for (int k = 0; k < s->n; k++) {
    double d = c_illuminance_approx(s->colors[k]) - mean;
    nominator += s->weights[k] * (d * d);
  }
  double sum_of_weights = 0.0;
  for (int k = 0; k < s->n; k++) sum_of_weights += s->weights[k];
  return nominator / sum_of_weights;
}
static double stats_interval(const stats_t *s) { /* stats.rs:49-51 */
  return 1.96 * (sqrt(stats_variance(s)) / sqrt((double)s->n));
}

/* ==================================== render/integrator.rs:253-339 */
static void render_line(const RtScene *sc, const RtConfig *cfg, int y, double *band, double *sample_band,
                        double *linear_band, uint32_t *count_band, counters_t *cn) {
  rng_t rng;
  rng_seed_from_u64(&rng, (uint64_t)y);
  int width = cfg->img_width;
  uint64_t max_samples = (uint64_t)cfg->sample_batch_size * (uint64_t)cfg->max_sample_batches;
  stats_t stats;
  stats.colors = (col *)malloc(sizeof(col) * (size_t)(cfg->max_sample_batches > 0 ? cfg->max_sample_batches : 1));
  stats.weights = (double *)malloc(sizeof(double) * (size_t)(cfg->max_sample_batches > 0 ? cfg->max_sample_batches : 1));

  for (int x = 0; x < width; x++) {
    uint64_t actual_samples = 0;
    stats.n = 0;
    for (int b = 0; b < cfg->max_sample_batches; b++) {
      col batch_color = V(0.0, 0.0, 0.0);
      double batch_sum_sample_weight = 0.0;
      for (int i = 0; i < cfg->sample_batch_size; i++) {
        double sx, sy;
        sample_offset(cfg, x, y, i, &sx, &sy);
        double sample_weight = filter_evaluate(cfg, sx, sy);
        if (cfg->rng_mode == RT_RNG_CTR)
          rng_seed_path(&rng, (uint64_t)y * (uint64_t)width + (uint64_t)x, (uint64_t)b, (uint64_t)i, cfg->seed);
        ray_t ray = get_prime_ray(&sc->camera, x, y, &rng, sx, sy);
        col c = ray_color(sc, &ray, cfg->max_depth, cfg->max_depth, &rng, cn);
        batch_color = v_add(batch_color, v_mulf(c, sample_weight));
        batch_sum_sample_weight += sample_weight;
        actual_samples += 1;
        cn->samples++;
      }
      stats.colors[stats.n] = v_divf(batch_color, batch_sum_sample_weight);
      stats.weights[stats.n] = batch_sum_sample_weight;
      stats.n++;
      if (b > cfg->min_sample_batches) {
        if (stats_interval(&stats) < 0.05 * c_illuminance(stats_mean(&stats))) break;
      }
    }
    col mean = stats_mean(&stats);
    col px = c_linear_to_gamma(c_clamp(mean), cfg->gamma_correction);
    band[3 * x + 0] = px.x;
    band[3 * x + 1] = px.y;
    band[3 * x + 2] = px.z;
    /* usize arithmetic wraps in the reference's release profile */
    uint64_t num = actual_samples - (uint64_t)cfg->min_sample_batches * (uint64_t)cfg->sample_batch_size;
    double factor = (double)num / (double)max_samples;
    if (sample_band) {
      sample_band[3 * x + 0] = factor * 1.0;
      sample_band[3 * x + 1] = (1.0 - factor) * 1.0;
      sample_band[3 * x + 2] = 0.0;
    }
    if (linear_band) {
      linear_band[3 * x + 0] = mean.x;
      linear_band[3 * x + 1] = mean.y;
      linear_band[3 * x + 2] = mean.z;
    }
    if (count_band) count_band[x] = (uint32_t)actual_samples;
  }
  free(stats.colors);
  free(stats.weights);
}

/* The row shard of RtConfig: blocks of row_block rows (0 or 1: single rows), the blocks row_first,
 * row_first + row_stride, ... of the image. */
static int row_block_of(const RtConfig *cfg) { return cfg->row_block > 1 ? cfg->row_block : 1; }
static int shard_rows(const RtConfig *cfg) {
  int B = row_block_of(cfg), n_blocks = (cfg->img_height + B - 1) / B;
  if (cfg->row_stride <= 0 || cfg->row_first < 0 || cfg->row_first >= n_blocks) return 0;
  int mine = (n_blocks - cfg->row_first + cfg->row_stride - 1) / cfg->row_stride;
  int rows = mine * B;
  if (cfg->row_first + (mine - 1) * cfg->row_stride == n_blocks - 1 && cfg->img_height % B) rows -= B - cfg->img_height % B;
  return rows;
}
static int image_row(const RtConfig *cfg, int k) {
  int B = row_block_of(cfg);
  return (cfg->row_first + (k / B) * cfg->row_stride) * B + k % B;
}

/* ============================ driver: rows fan out over a thread pool,
 * mirroring the Rayon par_iter of integrator.rs:226-243 (one task = one row,
 * dynamic hand-out). */
typedef struct {
  const RtScene *sc;
  const RtConfig *cfg;
  RtImage *out;
  int n_rows;
  volatile int next_row;
  pthread_mutex_t mu;
  counters_t total;
} job_t;

static void *worker(void *arg) {
  job_t *job = (job_t *)arg;
  counters_t cn;
  memset(&cn, 0, sizeof cn);
  int width = job->cfg->img_width;
  for (;;) {
    int k = __sync_fetch_and_add(&job->next_row, 1);
    if (k >= job->n_rows) break;
    int y = image_row(job->cfg, k);
    size_t off = (size_t)k * (size_t)width;
    render_line(job->sc, job->cfg, y, job->out->pixels + 3 * off,
                job->out->sample_pixels ? job->out->sample_pixels + 3 * off : NULL,
                job->out->linear ? job->out->linear + 3 * off : NULL, job->out->samples ? job->out->samples + off : NULL, &cn);
  }
  pthread_mutex_lock(&job->mu);
  job->total.rays += cn.rays;
  job->total.node_visits += cn.node_visits;
  job->total.box_tests += cn.box_tests;
  job->total.leaf_tests += cn.leaf_tests;
  job->total.samples += cn.samples;
  for (int i = 0; i < RT_MAX_DEPTH_STATS; i++) job->total.rays_by_depth[i] += cn.rays_by_depth[i];
  pthread_mutex_unlock(&job->mu);
  return NULL;
}

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int rt_oracle_render(const RtScene *sc, const RtConfig *cfg, RtImage *out, RtStats *stats, int n_threads) {
  if (!sc || !cfg || !out || !out->pixels) return RT_ERR_BAD_ARG;
  if (cfg->img_width <= 0 || cfg->img_height <= 0 || cfg->sample_batch_size <= 0 || cfg->max_sample_batches <= 0)
    return RT_ERR_BAD_ARG;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  job_t job;
  memset(&job, 0, sizeof job);
  job.sc = sc;
  job.cfg = cfg;
  job.out = out;
  job.n_rows = shard_rows(cfg);
  pthread_mutex_init(&job.mu, NULL);
  double t0 = now_ms();
  pthread_t th[256];
  for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, worker, &job);
  for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
  double t1 = now_ms();
  pthread_mutex_destroy(&job.mu);
  if (stats) {
    memset(stats, 0, sizeof *stats);
    stats->rays = job.total.rays;
    stats->rays_traced = job.total.rays;
    stats->samples = job.total.samples;
    stats->node_visits = job.total.node_visits;
    stats->box_tests = job.total.box_tests;
    stats->leaf_tests = job.total.leaf_tests;
    for (int i = 0; i < RT_MAX_DEPTH_STATS; i++) stats->rays_by_depth[i] = job.total.rays_by_depth[i];
    stats->wall_ms = t1 - t0;
  }
  return RT_OK;
}

/* Primary-ray closest hits for every pixel of the shard and every sample of
 * batch 0 (camera.rs:95-120 -> bvh_node_sah.rs:162-165 with the interval of
 * integrator.rs:359).  Layout as rt_hip_primary_hits. */
int rt_oracle_primary_hits(const RtScene *sc, const RtConfig *cfg, int32_t *prim_id, double *t) {
  if (!sc || !cfg || !prim_id || !t) return RT_ERR_BAD_ARG;
  int n_rows = shard_rows(cfg);
  counters_t cn;
  memset(&cn, 0, sizeof cn);
  for (int k = 0; k < n_rows; k++) {
    int y = image_row(cfg, k);
    rng_t rng;
    rng_seed_from_u64(&rng, (uint64_t)y);
    for (int x = 0; x < cfg->img_width; x++) {
      for (int i = 0; i < cfg->sample_batch_size; i++) {
        double sx, sy;
        sample_offset(cfg, x, y, i, &sx, &sy);
        if (cfg->rng_mode == RT_RNG_CTR)
          rng_seed_path(&rng, (uint64_t)y * (uint64_t)cfg->img_width + (uint64_t)x, 0, (uint64_t)i, cfg->seed);
        ray_t ray = get_prime_ray(&sc->camera, x, y, &rng, sx, sy);
        interval_t iv = {0.0001, RT_F64_MAX};
        hit_t h;
        size_t o = ((size_t)k * (size_t)cfg->img_width + (size_t)x) * (size_t)cfg->sample_batch_size + (size_t)i;
        if (bvh_hit(sc, &ray, &iv, &h, &cn)) {
          prim_id[o] = h.element;
          t[o] = h.t;
        } else {
          prim_id[o] = -1;
          t[o] = 0.0;
        }
      }
    }
  }
  return RT_OK;
}

/* Closest hits for caller-supplied rays; layout as rt_hip_trace_rays.
 * `visits` (optional, 3 per ray) receives node_visits, box_tests, leaf_tests. */
int rt_oracle_trace_rays(const RtScene *sc, int64_t n, const double *origins, const double *directions, int32_t *prim_id,
                         double *t, uint32_t *visits) {
  if (!sc || !origins || !directions || !prim_id || !t) return RT_ERR_BAD_ARG;
  for (int64_t r = 0; r < n; r++) {
    counters_t cn;
    memset(&cn, 0, sizeof cn);
    ray_t ray = ray_new(V(origins[3 * r], origins[3 * r + 1], origins[3 * r + 2]),
                        V(directions[3 * r], directions[3 * r + 1], directions[3 * r + 2]));
    interval_t iv = {0.0001, RT_F64_MAX};
    hit_t h;
    if (bvh_hit(sc, &ray, &iv, &h, &cn)) {
      prim_id[r] = h.element;
      t[r] = h.t;
    } else {
      prim_id[r] = -1;
      t[r] = 0.0;
    }
    if (visits) {
      visits[3 * r] = (uint32_t)cn.node_visits;
      visits[3 * r + 1] = (uint32_t)cn.box_tests;
      visits[3 * r + 2] = (uint32_t)cn.leaf_tests;
    }
  }
  return RT_OK;
}

/* =================== unit-level entry points for the known-answer tests */
int rt_oracle_aabb_hit(const double *bmin, const double *bmax, const double *origin, const double *direction) {
  ray_t r = ray_new(V(origin[0], origin[1], origin[2]), V(direction[0], direction[1], direction[2]));
  return aabb_hit(bmin, bmax, &r);
}
void rt_oracle_ray_new(const double *origin, const double *direction, double *out9, int *signs3) {
  ray_t r = ray_new(V(origin[0], origin[1], origin[2]), V(direction[0], direction[1], direction[2]));
  out9[0] = r.origin.x; out9[1] = r.origin.y; out9[2] = r.origin.z;
  out9[3] = r.direction.x; out9[4] = r.direction.y; out9[5] = r.direction.z;
  out9[6] = r.inv_direction.x; out9[7] = r.inv_direction.y; out9[8] = r.inv_direction.z;
  signs3[0] = r.sign_x; signs3[1] = r.sign_y; signs3[2] = r.sign_z;
}
void rt_oracle_ray_at(const double *origin, const double *direction, double t, double *out3) {
  ray_t r = ray_new(V(origin[0], origin[1], origin[2]), V(direction[0], direction[1], direction[2]));
  v3 p = ray_at(&r, t);
  out3[0] = p.x; out3[1] = p.y; out3[2] = p.z;
}
void rt_oracle_vec3_ops(const double *a, const double *b, double *out) {
  v3 A = V(a[0], a[1], a[2]), B = V(b[0], b[1], b[2]);
  v3 c = v_cross(A, B), n = v_normalized(A);
  out[0] = v_dot(A, B);
  out[1] = c.x; out[2] = c.y; out[3] = c.z;
  out[4] = v_length(A);
  out[5] = n.x; out[6] = n.y; out[7] = n.z;
  out[8] = v_length(v_sub(A, B)); /* distance, vec3.rs:116-121 */
  out[9] = v_length_squared(A);
}
void rt_oracle_color_ops(const double *c, double gamma, double *clamped3, double *gamma3, uint8_t *rgb3, double *lum2) {
  col C = V(c[0], c[1], c[2]);
  col cl = c_clamp(C), g = c_linear_to_gamma(C, gamma);
  clamped3[0] = cl.x; clamped3[1] = cl.y; clamped3[2] = cl.z;
  gamma3[0] = g.x; gamma3[1] = g.y; gamma3[2] = g.z;
  /* to_rgb, color.rs:47-54: clamp then truncating cast */
  rgb3[0] = (uint8_t)(cl.x * 255.0); rgb3[1] = (uint8_t)(cl.y * 255.0); rgb3[2] = (uint8_t)(cl.z * 255.0);
  lum2[0] = c_illuminance(C);
  lum2[1] = c_illuminance_approx(C);
}
int rt_oracle_sphere_hit(const double *center, double radius, const double *origin, const double *direction, double tmin,
                         double tmax, double *t, double *point3, double *normal3, int *front_face) {
  RtSphere s;
  memset(&s, 0, sizeof s);
  s.center.x = center[0]; s.center.y = center[1]; s.center.z = center[2];
  s.radius = radius;
  ray_t r = ray_new(V(origin[0], origin[1], origin[2]), V(direction[0], direction[1], direction[2]));
  interval_t iv = {tmin, tmax};
  hit_t h;
  if (!sphere_hit(&s, &r, &iv, &h)) return 0;
  *t = h.t;
  point3[0] = h.point.x; point3[1] = h.point.y; point3[2] = h.point.z;
  normal3[0] = h.normal.x; normal3[1] = h.normal.y; normal3[2] = h.normal.z;
  *front_face = h.front_face;
  return 1;
}
int rt_oracle_triangle_hit(const RtTriangle *tr, const double *origin, const double *direction, double tmin, double tmax,
                           double *t) {
  ray_t r = ray_new(V(origin[0], origin[1], origin[2]), V(direction[0], direction[1], direction[2]));
  interval_t iv = {tmin, tmax};
  hit_t h;
  if (!triangle_hit(tr, &r, &iv, &h)) return 0;
  *t = h.t;
  return 1;
}
int rt_oracle_quad_hit(const RtQuad *q, const double *origin, const double *direction, double tmin, double tmax, double *t) {
  ray_t r = ray_new(V(origin[0], origin[1], origin[2]), V(direction[0], direction[1], direction[2]));
  interval_t iv = {tmin, tmax};
  hit_t h;
  if (!quad_hit(q, &r, &iv, &h)) return 0;
  *t = h.t;
  return 1;
}
double rt_oracle_quad_pdf_value(const RtQuad *q, const double *origin, const double *direction) {
  return quad_pdf_value(q, V(origin[0], origin[1], origin[2]), V(direction[0], direction[1], direction[2]));
}
double rt_oracle_sphere_pdf_value(const RtSphere *s, const double *origin, const double *direction) {
  return sphere_pdf_value(s, V(origin[0], origin[1], origin[2]), V(direction[0], direction[1], direction[2]));
}
float rt_oracle_sobol_sample(uint32_t index, uint32_t dim, uint32_t seed) { return sobol_sample(index, dim, seed); }
double rt_oracle_filter_evaluate(double sx, double sy, double radius, double b, double c) {
  RtConfig cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.pixel_radius = radius;
  cfg.filter_b = b;
  cfg.filter_c = c;
  return filter_evaluate(&cfg, sx, sy);
}
/* mode 0: xoshiro256++ seeded by seed_from_u64(seed); mode 1: counter stream
 * for (pixel=seed, batch=0, sample=0, seed=0).  kind 0: next_u64 -> out_u64;
 * 1: gen_range(lo,hi) -> out_f64; 2: gen::<f64>() -> out_f64; 3: choose
 * index in 0..n -> out_u64. */
void rt_oracle_rng_stream(int mode, uint64_t seed, int kind, double lo, double hi, uint32_t n, int count, uint64_t *out_u64,
                          double *out_f64) {
  rng_t r;
  if (mode == RT_RNG_REF) rng_seed_from_u64(&r, seed);
  else rng_seed_path(&r, seed, 0, 0, 0);
  for (int i = 0; i < count; i++) {
    switch (kind) {
      case 0: out_u64[i] = rng_next_u64(&r); break;
      case 1: out_f64[i] = rng_gen_range(&r, lo, hi); break;
      case 2: out_f64[i] = rng_gen_f64(&r); break;
      default: out_u64[i] = rng_gen_index(&r, n); break;
    }
  }
}
void rt_oracle_stats(const double *colors, const double *weights, int n, double *mean3, double *variance, double *interval) {
  stats_t s;
  s.colors = (col *)malloc(sizeof(col) * (size_t)n);
  s.weights = (double *)malloc(sizeof(double) * (size_t)n);
  s.n = n;
  for (int k = 0; k < n; k++) {
    s.colors[k] = V(colors[3 * k], colors[3 * k + 1], colors[3 * k + 2]);
    s.weights[k] = weights[k];
  }
  col m = stats_mean(&s);
  mean3[0] = m.x; mean3[1] = m.y; mean3[2] = m.z;
  *variance = stats_variance(&s);
  *interval = stats_interval(&s);
  free(s.colors);
  free(s.weights);
}
void rt_oracle_sample_offset(double pixel_radius, int x, int y, int i, double *sx, double *sy) {
  RtConfig cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.pixel_radius = pixel_radius;
  sample_offset(&cfg, x, y, i, sx, sy);
}
void rt_oracle_onb(const double *w, double *out9) {
  onb_t o = onb_build_from_w(V(w[0], w[1], w[2]));
  out9[0] = o.u.x; out9[1] = o.u.y; out9[2] = o.u.z;
  out9[3] = o.v.x; out9[4] = o.v.y; out9[5] = o.v.z;
  out9[6] = o.w.x; out9[7] = o.w.y; out9[8] = o.w.z;
}
void rt_oracle_material_vectors(const double *v, const double *n, double eta, double cosine, double *reflect3, double *refract3,
                                double *refl) {
  v3 r = reflect_vector(V(v[0], v[1], v[2]), V(n[0], n[1], n[2]));
  v3 f = refract_ray(V(v[0], v[1], v[2]), V(n[0], n[1], n[2]), eta);
  reflect3[0] = r.x; reflect3[1] = r.y; reflect3[2] = r.z;
  refract3[0] = f.x; refract3[1] = f.y; refract3[2] = f.z;
  *refl = reflectance(cosine, eta);
}
