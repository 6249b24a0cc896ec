// rt_hip.hip -- MI355X (gfx950) implementation of the C ABI in
// include/rt_hip.h: a wavefront path tracer that replaces the reference's
// per-row loop (/root/reference/src/render/integrator.rs:226-339).
//
// Pipeline per batch round (one round = one `b` iteration of
// integrator.rs:279 for every still-active pixel):
//
//   k_generate    one lane per (active pixel, sample i): Sobol offset,
//                 Mitchell weight, camera ray, per-path RNG stream
//                 (integrator.rs:284-296, camera.rs:95-120)
//   per bounce:
//     k_extend    BVH closest-hit query (bvh_node_sah.rs:162-253), LDS stacks
//     k_shade     emit / Lambertian mixture-PDF / Metal / Dielectric
//                 (integrator.rs:361-469), wave64 ballot + prefix-sum
//                 compaction of the surviving paths into the next queue
//   k_accumulate  per pixel: ordered batch sum, Stats::push, z-test
//                 (integrator.rs:305-326, stats.rs), compaction of the
//                 unconverged pixels, final clamp + gamma + heat-map
//                 (integrator.rs:331-337)
//
// Path state lives in SoA queues in HBM; every array is indexed so that a
// wave touches consecutive addresses.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rt_device.h"

using namespace rtd;

// ===================================================================== errors
static thread_local std::string g_error;
const char *rt_hip_last_error(void) { return g_error.c_str(); }

static int fail(int code, const std::string &msg) {
  g_error = msg;
  return code;
}
#define HIP_TRY(expr)                                                                                \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) {                                                                          \
      int code_ = (e_ == hipErrorOutOfMemory) ? RT_ERR_OOM : RT_ERR_HIP;                             \
      return fail(code_, std::string(#expr) + ": " + hipGetErrorString(e_));                         \
    }                                                                                                \
  } while (0)

constexpr int32_t kFlagTimeKernels = RT_FLAG_TIME_KERNELS;
constexpr int32_t kFlagCountVisits = RT_FLAG_COUNT_VISITS;
constexpr int32_t kFlagEarlyTerminate = RT_FLAG_EARLY_TERMINATE;

// ==================================================================== queues
// SoA path state.  One "queue" holds the rays about to be traced at a bounce.
struct PathQueue {
  double *ox, *oy, *oz;  // origin
  double *dx, *dy, *dz;  // normalised direction
  double *tx, *ty, *tz;  // throughput
  uint64_t *rng;         // SplitMix64 counter
  uint32_t *pid;         // sample * n_slots + slot
};

struct HitBuffer {
  double *t;
  uint32_t *ref;
};

struct RenderParams {
  int32_t width, height, row_first, row_stride;
  int32_t S;  // sample_batch_size
  int32_t max_batches, min_batches, max_depth;
  double pixel_radius, gamma, filter_b, filter_c;
  uint64_t seed;
  int32_t n_pix;  // pixels in the shard
  int32_t flags;
};

// XCD-aware block order: the dispatcher deals consecutive workgroups
// round-robin over the 8 XCDs, so workgroups b and b+8 share an L2.  Map the
// hardware block id to a logical id such that each XCD owns a contiguous run
// of logical blocks; consecutive logical blocks hold neighbouring pixels and
// therefore walk the same BVH nodes out of the same L2.  Bijective for any
// grid size.
__device__ __forceinline__ uint32_t xcd_logical_block(uint32_t b, uint32_t nblocks) {
  uint32_t xcd = b & 7u, q = nblocks >> 3, rem = nblocks & 7u;
  uint32_t base = (xcd < rem) ? xcd * (q + 1u) : rem * (q + 1u) + (xcd - rem) * q;
  return base + (b >> 3);
}

// ================================================================== generate
__global__ void __launch_bounds__(256) k_generate(DevScene sc, RenderParams rp, const uint32_t *active, int32_t n_slots,
                                                   int32_t batch, PathQueue q, double *weights, uint32_t *qcount0) {
  int64_t n = (int64_t)n_slots * rp.S;
  if (blockIdx.x == 0 && threadIdx.x == 0) *qcount0 = (uint32_t)n;
  uint32_t lb = xcd_logical_block(blockIdx.x, gridDim.x);
  for (int64_t j = (int64_t)lb * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    int32_t slot = (int32_t)(j / rp.S), i = (int32_t)(j % rp.S);
    uint32_t p = active[slot];
    int32_t x = (int32_t)(p % (uint32_t)rp.width);
    int32_t y = rp.row_first + (int32_t)(p / (uint32_t)rp.width) * rp.row_stride;
    // integrator.rs:275: pixel_num = x * y (product), truncated to u32 as the Sobol seed
    uint32_t sobol_seed = (uint32_t)((uint64_t)x * (uint64_t)y);
    double sx = ((double)sobol_sample((uint32_t)i, 0, sobol_seed) * rp.pixel_radius) - rp.pixel_radius / 2.;
    double sy = ((double)sobol_sample((uint32_t)i, 1, sobol_seed) * rp.pixel_radius) - rp.pixel_radius / 2.;
    double inv_r = 1.0 / rp.pixel_radius;  // filter.rs:33-42
    double w = mitchell_1d(sx * inv_r, rp.filter_b, rp.filter_c) * mitchell_1d(sy * inv_r, rp.filter_b, rp.filter_c);
    uint64_t rng = rng_seed_path((uint64_t)y * (uint64_t)rp.width + (uint64_t)x, (uint64_t)batch, (uint64_t)i, rp.seed);
    // camera.rs:95-120
    d3 center = ld3(sc.camera.camera_center);
    d3 origin = center;
    if (!(sc.camera.defocus_angle <= 0.0)) {
      d3 pd = random_in_unit_disk(rng);
      origin = center + (ld3(sc.camera.defocus_disk_u) * pd.x) + (ld3(sc.camera.defocus_disk_v) * pd.y);
    }
    d3 pixel_loc = ld3(sc.camera.pixel00_loc) + (ld3(sc.camera.pixel_delta_u) * (double)x) + (ld3(sc.camera.pixel_delta_v) * (double)y);
    d3 pixel_sample = pixel_loc + D3(sx, sy, 0.);
    d3 dir = normalized(pixel_sample - origin);
    uint32_t pid = (uint32_t)i * (uint32_t)n_slots + (uint32_t)slot;
    q.ox[j] = origin.x; q.oy[j] = origin.y; q.oz[j] = origin.z;
    q.dx[j] = dir.x; q.dy[j] = dir.y; q.dz[j] = dir.z;
    q.tx[j] = 1.0; q.ty[j] = 1.0; q.tz[j] = 1.0;
    q.rng[j] = rng;
    q.pid[j] = pid;
    weights[pid] = w;
  }
}

// ==================================================================== extend
extern __shared__ uint32_t g_stack[];

template <bool COUNT>
__global__ void __launch_bounds__(256) k_extend(DevScene sc, PathQueue q, const uint32_t *count_ptr, HitBuffer hits,
                                                 unsigned long long *visit_counters) {
  uint32_t n = *count_ptr;
  uint32_t lb = xcd_logical_block(blockIdx.x, gridDim.x);
  uint32_t *stack = g_stack + threadIdx.x;
  uint32_t nn = 0, nl = 0;
  for (uint32_t j = lb * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    Ray r = make_ray_normalized(D3(q.ox[j], q.oy[j], q.oz[j]), D3(q.dx[j], q.dy[j], q.dz[j]));
    Hit h = bvh_closest_hit<COUNT>(sc, r, stack, (int)blockDim.x, &nn, &nl);
    hits.t[j] = h.t;
    hits.ref[j] = h.ref;
  }
  if (COUNT) {
    atomicAdd(&visit_counters[0], (unsigned long long)nn);
    atomicAdd(&visit_counters[1], (unsigned long long)nl);
  }
}

// ===================================================================== shade
// HittablePDF::value over the attractor list (pdf.rs:98-107) with
// Quad::pdf_value (quad.rs:155-167) / Sphere::pdf_value (sphere.rs:154-169).
__device__ double attractor_pdf_value(const DevScene &sc, d3 origin, d3 direction) {
  double weight = 1.0 / (double)sc.n_attractors;
  double sum = 0.0;
  Ray probe = make_ray(origin, direction);  // Ray::new re-normalises
  for (int a = 0; a < sc.n_attractors; a++) {
    uint32_t ref = sc.attractors[a];
    uint32_t kind = (ref >> kRefKindShift) & 3u, idx = ref & kRefIndexMask;
    double v = 0.0, t;
    if (kind == RT_PRIM_QUAD) {
      const DevQuad &q = sc.quads[idx];
      if (quad_hit(q, probe, 0.001, INFINITY, t)) {
        double distance_squared = t * t * length_squared(direction);
        double cosine = fabs(dot(direction, ld3(q.normal))) / length(direction);
        v = distance_squared / (cosine * q.area);
      }
    } else if (kind == RT_PRIM_SPHERE) {
      const DevSphere &s = sc.spheres[idx];
      if (sphere_hit(s, probe, 0.001, INFINITY, t)) {
        double cos_theta_max = sqrt(1. - s.radius * s.radius / length_squared(ld3(s.center) - origin));
        double solid_angle = 2.0 * kPi * (1.0 - cos_theta_max);
        v = 1. / solid_angle;
      }
    }
    sum += weight * v;
  }
  return sum;
}

// HittablePDF::generate (pdf.rs:110-115): choose one attractor, then its
// random() (quad.rs:172-178, sphere.rs:173-194).
__device__ d3 attractor_generate(const DevScene &sc, d3 origin, uint64_t &rng) {
  uint32_t k = rng_gen_index(rng, (uint32_t)sc.n_attractors);
  uint32_t ref = sc.attractors[k];
  uint32_t kind = (ref >> kRefKindShift) & 3u, idx = ref & kRefIndexMask;
  if (kind == RT_PRIM_QUAD) {
    const DevQuad &q = sc.quads[idx];
    double r0 = rng_gen_range(rng, 0.0, 1.0);
    double r1 = rng_gen_range(rng, 0.0, 1.0);
    d3 p = ld3(q.q) + (ld3(q.u) * r0) + (ld3(q.v) * r1);
    return p - origin;
  }
  const DevSphere &s = sc.spheres[idx];
  d3 direction = ld3(s.center) - origin;
  Onb uvw = onb_build_from_w(direction);
  double distance_squared = length_squared(direction);
  double r1 = rng_gen_range(rng, 0.0, 1.0);
  double r2 = rng_gen_range(rng, 0.0, 1.0);
  double z = 1.0 + r2 * (sqrt(1.0 - s.radius * s.radius / distance_squared) - 1.0);
  double phi = 2.0 * kPi * r1;
  double sn, cs;
  sincos(phi, &sn, &cs);
  double sq = sqrt(1.0 - z * z);
  return onb_local_vec(uvw, D3(cs * sq, sn * sq, z));
}

RT_DEV double cosine_pdf_value(const Onb &uvw, d3 direction) {  // pdf.rs:159-162
  double cosine_theta = dot(normalized(direction), uvw.w);
  return fmax(0.0, cosine_theta / kPi);
}

__global__ void __launch_bounds__(256) k_shade(DevScene sc, PathQueue q, const uint32_t *count_ptr, HitBuffer hits, PathQueue out,
                                                uint32_t *out_count, double *radiance, int32_t n_slots, int32_t S, int32_t last_bounce,
                                                int32_t flags) {
  uint32_t n = *count_ptr;
  uint32_t lb = xcd_logical_block(blockIdx.x, gridDim.x);
  uint32_t stride = gridDim.x * blockDim.x;
  // round the trip count up to whole waves so every lane reaches the ballot
  uint32_t n_round = (n + 63u) & ~63u;
  int64_t plane = (int64_t)n_slots * S;  // radiance is stored as 3 planes
  for (uint32_t j = lb * blockDim.x + threadIdx.x; j < n_round; j += stride) {
    bool alive = false;
    d3 no = D3(0, 0, 0), nd = D3(0, 0, 0), T = D3(0, 0, 0);
    uint64_t rng = 0;
    uint32_t pid = 0;
    if (j < n) {
      pid = q.pid[j];
      T = D3(q.tx[j], q.ty[j], q.tz[j]);
      uint32_t ref = hits.ref[j];
      d3 rad = D3(0.0, 0.0, 0.0);
      if (ref != kRefNone) {
        double t = hits.t[j];
        d3 o = D3(q.ox[j], q.oy[j], q.oz[j]), d = D3(q.dx[j], q.dy[j], q.dz[j]);
        rng = q.rng[j];
        d3 point = o + d * t;  // Ray::at
        uint32_t kind = (ref >> kRefKindShift) & 3u, idx = ref & kRefIndexMask;
        d3 normal;
        uint32_t mat;
        bool front_face;
        if (kind == RT_PRIM_TRIANGLE) {
          normal = ld3(sc.tri_normals + 3 * (size_t)idx);
          mat = sc.triangles[idx].material;
          front_face = dot(d, normal) < 0.0;
        } else if (kind == RT_PRIM_QUAD) {
          normal = ld3(sc.quads[idx].normal);
          mat = sc.quads[idx].material;
          front_face = dot(d, normal) < 0.0;
        } else {  // sphere.rs:127-136
          const DevSphere &s = sc.spheres[idx];
          d3 outward = normalized((point - ld3(s.center)) / s.radius);
          front_face = dot(d, outward) < 0.0;
          normal = front_face ? outward : neg(outward);
          mat = s.material;
        }
        const DevMaterial &m = sc.materials[mat];
        d3 albedo = ld3(m.albedo);
        if (m.kind == RT_MAT_DIFFUSE_LIGHT) {
          if (front_face) rad = T * albedo;  // diffuse_light.rs:20-29
        } else if (last_bounce) {
          // the next ray_color call would run at depth 0 and return black
        } else if (m.kind == RT_MAT_LAMBERTIAN) {
          Onb uvw = onb_build_from_w(normal);  // CosinePDF::new
          d3 dir;
          double pdf_val, spdf;
          if (sc.n_attractors > 0) {  // integrator.rs:381-391, MixedPDF ratio 0.5
            double pick = rng_gen_range(rng, 0.0, 1.0);
            if (pick < 0.5) dir = attractor_generate(sc, point, rng);
            else dir = onb_local_vec(uvw, random_cosine_direction(rng));
            nd = normalized(dir);
            spdf = cosine_pdf_value(uvw, nd);
            pdf_val = 0.5 * attractor_pdf_value(sc, point, nd) + (1. - 0.5) * spdf;
          } else {
            dir = onb_local_vec(uvw, random_cosine_direction(rng));
            nd = normalized(dir);
            spdf = cosine_pdf_value(uvw, nd);
            pdf_val = spdf;
          }
          // (attenuation * target * scattering_pdf) / pdf_val, NaN -> black (integrator.rs:413-419)
          d3 Tn = ((T * albedo) * spdf) / pdf_val;
          if (has_nan(Tn)) Tn = D3(0., 0., 0.);
          bool zero = (Tn.x == 0.0 && Tn.y == 0.0 && Tn.z == 0.0);
          if (!(zero && (flags & kFlagEarlyTerminate))) {
            T = Tn;
            no = point;
            alive = true;
          }
        } else if (m.kind == RT_MAT_METAL) {  // metal.rs:21-53
          d3 v = normalized(d);
          d3 reflected = v - normal * (2.0 * dot(v, normal));
          d3 dir = reflected + random_unit_vector(rng) * m.param;
          if (dot(normal, dir) > 0.0) {
            T = albedo * T;
            no = point;
            nd = normalized(dir);
            alive = true;
          }
        } else {  // dielectric.rs:24-66
          double ratio = front_face ? 1.0 / m.param : m.param;
          d3 unit_direction = normalized(d);
          d3 minus = unit_direction * -1.0;
          double cos_theta = fmin(dot(normal, minus), 1.0);
          double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
          bool cannot_refract = ratio * sin_theta > 1.0;
          bool reflect = cannot_refract;
          if (!reflect) {  // the draw happens only when refraction is possible
            double r0 = (1.0 - ratio) / (1.0 + ratio);
            r0 = r0 * r0;
            double x1 = 1.0 - cos_theta, x2 = x1 * x1;
            double refl = r0 + (1.0 - r0) * (x2 * x2 * x1);
            reflect = refl > rng_gen_f64(rng);
          }
          d3 dir;
          if (reflect) {
            dir = unit_direction - normal * (2.0 * dot(unit_direction, normal));
          } else {  // refract_ray, metal.rs:62-68
            double ct = fmin(dot(normal, unit_direction * -1.0), 1.0);
            d3 r_out_perp = (unit_direction + normal * ct) * ratio;
            d3 r_out_parallel = normal * (sqrt(fabs(1.0 - length_squared(r_out_perp))) * -1.0);
            dir = r_out_perp + r_out_parallel;
          }
          no = point;
          nd = normalized(dir);
          alive = true;  // attenuation is white
        }
      }
      if (!alive) {
        radiance[pid] = rad.x;
        radiance[plane + pid] = rad.y;
        radiance[2 * plane + pid] = rad.z;
      }
    }
    // wave64 stream compaction: ballot, prefix popcount, one atomic per wave
    unsigned long long mask = __ballot(alive);
    if (mask) {
      uint32_t lane = threadIdx.x & 63u;
      uint32_t prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      uint32_t base = 0;
      if (lane == (uint32_t)(__ffsll((long long)mask) - 1)) base = atomicAdd(out_count, (uint32_t)__popcll(mask));
      base = __shfl(base, __ffsll((long long)mask) - 1);
      if (alive) {
        uint32_t k = base + prefix;
        out.ox[k] = no.x; out.oy[k] = no.y; out.oz[k] = no.z;
        out.dx[k] = nd.x; out.dy[k] = nd.y; out.dz[k] = nd.z;
        out.tx[k] = T.x; out.ty[k] = T.y; out.tz[k] = T.z;
        out.rng[k] = rng;
        out.pid[k] = pid;
      }
    }
  }
}

// ================================================================ accumulate
struct PixelOutputs {
  double *pixels, *sample_pixels, *linear;
  uint32_t *samples;
};

__global__ void __launch_bounds__(256) k_accumulate(RenderParams rp, const uint32_t *active, int32_t n_slots, int32_t batch,
                                                     const double *radiance, const double *weights, double *hist, uint32_t *next_active,
                                                     uint32_t *next_count, PixelOutputs out, const uint32_t *qcounts,
                                                     unsigned long long *rays_by_depth) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int k = 0; k < rp.max_depth; k++) {
      int slot = k < RT_MAX_DEPTH_STATS ? k : RT_MAX_DEPTH_STATS - 1;
      rays_by_depth[slot] += qcounts[k];
    }
  }
  int32_t n_round = (n_slots + 63) & ~63;
  int64_t plane = (int64_t)n_slots * rp.S;
  size_t np = (size_t)rp.n_pix;
  for (int32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n_round; slot += gridDim.x * blockDim.x) {
    bool again = false;
    uint32_t p = 0;
    if (slot < n_slots) {
      p = active[slot];
      // integrator.rs:280-317: batch_color += ray_color * w, in sample order
      double br = 0.0, bg = 0.0, bb = 0.0, bw = 0.0;
      for (int i = 0; i < rp.S; i++) {
        int64_t pid = (int64_t)i * n_slots + slot;
        double w = weights[pid];
        br = br + radiance[pid] * w;
        bg = bg + radiance[plane + pid] * w;
        bb = bb + radiance[2 * plane + pid] * w;
        bw += w;
      }
      // stats.push(batch_color / batch_sum_sample_weight, batch_sum_sample_weight)
      double *h = hist + (size_t)batch * 4 * np + p;
      h[0] = br / bw;
      h[np] = bg / bw;
      h[2 * np] = bb / bw;
      h[3 * np] = bw;
      int nb = batch + 1;
      // Stats::mean, stats.rs:21-30
      double mr = 0.0, mg = 0.0, mb = 0.0, sw = 0.0;
      for (int k = 0; k < nb; k++) {
        const double *hk = hist + (size_t)k * 4 * np + p;
        double w = hk[3 * np];
        mr = mr + hk[0] * w;
        mg = mg + hk[np] * w;
        mb = mb + hk[2 * np] * w;
      }
      for (int k = 0; k < nb; k++) sw += hist[(size_t)k * 4 * np + 3 * np + p];
      mr = mr / sw;
      mg = mg / sw;
      mb = mb / sw;
      bool stop = false;
      if (batch > rp.min_batches) {  // integrator.rs:321-326
        double variance = 0.0;      // Stats::variance, stats.rs:32-47
        if (nb != 1) {
          double mean_l = (mr + mr + mb + mg + mg + mg) / 6.0;
          double nominator = 0.0;
          for (int k = 0; k < nb; k++) {
            const double *hk = hist + (size_t)k * 4 * np + p;
            double r = hk[0], g = hk[np], b = hk[2 * np];
            double dl = (r + r + b + g + g + g) / 6.0 - mean_l;
            nominator += hk[3 * np] * (dl * dl);
          }
          variance = nominator / sw;
        }
        double interval = 1.96 * (sqrt(variance) / sqrt((double)nb));
        double illum = 0.2126 * mr + 0.7152 * mg + 0.0722 * mb;
        stop = interval < 0.05 * illum;
      }
      if (stop || batch == rp.max_batches - 1) {
        // integrator.rs:331: mean.clamp().linear_to_gamma(gamma)
        double inv_g = 1.0 / rp.gamma;
        out.pixels[3 * (size_t)p + 0] = pow(fmax(fmin(mr, 1.0), 0.0), inv_g);
        out.pixels[3 * (size_t)p + 1] = pow(fmax(fmin(mg, 1.0), 0.0), inv_g);
        out.pixels[3 * (size_t)p + 2] = pow(fmax(fmin(mb, 1.0), 0.0), inv_g);
        uint64_t actual = (uint64_t)nb * (uint64_t)rp.S;
        if (out.sample_pixels) {  // integrator.rs:333-337 (usize arithmetic wraps)
          uint64_t num = actual - (uint64_t)rp.min_batches * (uint64_t)rp.S;
          double factor = (double)num / (double)((uint64_t)rp.S * (uint64_t)rp.max_batches);
          out.sample_pixels[3 * (size_t)p + 0] = factor * 1.0;
          out.sample_pixels[3 * (size_t)p + 1] = (1.0 - factor) * 1.0;
          out.sample_pixels[3 * (size_t)p + 2] = 0.0;
        }
        if (out.linear) {
          out.linear[3 * (size_t)p + 0] = mr;
          out.linear[3 * (size_t)p + 1] = mg;
          out.linear[3 * (size_t)p + 2] = mb;
        }
        if (out.samples) out.samples[p] = (uint32_t)actual;
      } else {
        again = true;
      }
    }
    unsigned long long mask = __ballot(again);
    if (mask) {
      uint32_t lane = threadIdx.x & 63u;
      int leader = __ffsll((long long)mask) - 1;
      uint32_t prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      uint32_t base = 0;
      if ((int)lane == leader) base = atomicAdd(next_count, (uint32_t)__popcll(mask));
      base = __shfl(base, leader);
      if (again) next_active[base + prefix] = p;
    }
  }
}

__global__ void k_iota(uint32_t *a, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) a[i] = i;
}

// ============================================== primary hits / caller rays
RT_DEV int32_t element_of(const DevScene &sc, uint32_t ref) {
  if (ref == kRefNone) return -1;
  uint32_t kind = (ref >> kRefKindShift) & 3u, idx = ref & kRefIndexMask;
  if (kind == RT_PRIM_TRIANGLE) return (int32_t)sc.triangles[idx].element;
  if (kind == RT_PRIM_QUAD) return (int32_t)sc.quads[idx].element;
  return (int32_t)sc.spheres[idx].element;
}

__global__ void __launch_bounds__(256) k_primary(DevScene sc, RenderParams rp, int32_t *prim_id, double *t_out) {
  int64_t n = (int64_t)rp.n_pix * rp.S;
  uint32_t *stack = g_stack + threadIdx.x;
  uint32_t nn = 0, nl = 0;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    int32_t p = (int32_t)(j / rp.S), i = (int32_t)(j % rp.S);
    int32_t x = p % rp.width;
    int32_t y = rp.row_first + (p / rp.width) * rp.row_stride;
    uint32_t sobol_seed = (uint32_t)((uint64_t)x * (uint64_t)y);
    double sx = ((double)sobol_sample((uint32_t)i, 0, sobol_seed) * rp.pixel_radius) - rp.pixel_radius / 2.;
    double sy = ((double)sobol_sample((uint32_t)i, 1, sobol_seed) * rp.pixel_radius) - rp.pixel_radius / 2.;
    uint64_t rng = rng_seed_path((uint64_t)y * (uint64_t)rp.width + (uint64_t)x, 0, (uint64_t)i, rp.seed);
    d3 center = ld3(sc.camera.camera_center);
    d3 origin = center;
    if (!(sc.camera.defocus_angle <= 0.0)) {
      d3 pd = random_in_unit_disk(rng);
      origin = center + (ld3(sc.camera.defocus_disk_u) * pd.x) + (ld3(sc.camera.defocus_disk_v) * pd.y);
    }
    d3 pixel_loc = ld3(sc.camera.pixel00_loc) + (ld3(sc.camera.pixel_delta_u) * (double)x) + (ld3(sc.camera.pixel_delta_v) * (double)y);
    d3 pixel_sample = pixel_loc + D3(sx, sy, 0.);
    Ray r = make_ray(origin, pixel_sample - origin);
    Hit h = bvh_closest_hit<false>(sc, r, stack, (int)blockDim.x, &nn, &nl);
    prim_id[j] = element_of(sc, h.ref);
    t_out[j] = (h.ref == kRefNone) ? 0.0 : h.t;
  }
}

__global__ void __launch_bounds__(256) k_trace(DevScene sc, int64_t n, const double *origins, const double *directions,
                                                int32_t *prim_id, double *t_out) {
  uint32_t *stack = g_stack + threadIdx.x;
  uint32_t nn = 0, nl = 0;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    Ray r = make_ray(ld3(origins + 3 * j), ld3(directions + 3 * j));
    Hit h = bvh_closest_hit<false>(sc, r, stack, (int)blockDim.x, &nn, &nl);
    prim_id[j] = element_of(sc, h.ref);
    t_out[j] = (h.ref == kRefNone) ? 0.0 : h.t;
  }
}

// ================================================================ host side
struct RtSceneHandle {
  int device = 0;
  DevScene dev{};
  hipStream_t stream = nullptr;
  std::vector<void *> allocations;
  int32_t n_elements = 0;
  int32_t bvh_depth = 0;
};

namespace {

template <typename T>
int dev_alloc(RtSceneHandle *h, T **out, size_t count) {
  void *p = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  HIP_TRY(hipMalloc(&p, bytes));
  h->allocations.push_back(p);
  *out = static_cast<T *>(p);
  return RT_OK;
}

template <typename T>
int dev_upload(RtSceneHandle *h, const T **out, const std::vector<T> &v) {
  T *p = nullptr;
  int rc = dev_alloc(h, &p, v.size());
  if (rc != RT_OK) return rc;
  if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = p;
  return RT_OK;
}

float round_down(double x) {
  float f = (float)x;
  if ((double)f > x) f = std::nextafterf(f, -INFINITY);
  return f;
}
float round_up(double x) {
  float f = (float)x;
  if ((double)f < x) f = std::nextafterf(f, INFINITY);
  return f;
}

struct Scratch {  // per-render device buffers, freed on every exit path
  std::vector<void *> ptrs;
  std::vector<hipEvent_t> events;
  ~Scratch() {
    for (void *p : ptrs) (void)hipFree(p);
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
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
};

int shard_rows(const RtConfig *cfg) {
  return (cfg->img_height - cfg->row_first + cfg->row_stride - 1) / cfg->row_stride;
}

int check_config(const RtConfig *cfg) {
  if (!cfg) return fail(RT_ERR_BAD_ARG, "config is NULL");
  if (cfg->img_width <= 0 || cfg->img_height <= 0) return fail(RT_ERR_BAD_ARG, "image size must be positive");
  if (cfg->sample_batch_size <= 0 || cfg->max_sample_batches <= 0 || cfg->min_sample_batches < 0 || cfg->max_depth < 0)
    return fail(RT_ERR_BAD_ARG, "sample_batch_size and max_sample_batches must be >= 1; min_sample_batches, max_depth >= 0");
  if (cfg->row_stride <= 0 || cfg->row_first < 0 || cfg->row_first >= cfg->img_height)
    return fail(RT_ERR_BAD_ARG, "row shard out of range");
  if (!(cfg->pixel_radius > 0.0) || !(cfg->gamma_correction > 0.0)) return fail(RT_ERR_BAD_ARG, "pixel_radius and gamma_correction must be > 0");
  if (cfg->rng_mode != RT_RNG_CTR)
    return fail(RT_ERR_UNSUPPORTED, "the device path implements RT_RNG_CTR only (RT_RNG_REF is serial per image row)");
  int64_t n_pix = (int64_t)shard_rows(cfg) * cfg->img_width;
  if (n_pix > 0x7fffffff) return fail(RT_ERR_UNSUPPORTED, "more than 2^31-1 pixels in one shard");
  return RT_OK;
}

RenderParams make_params(const RtConfig *cfg) {
  RenderParams rp{};
  rp.width = cfg->img_width;
  rp.height = cfg->img_height;
  rp.row_first = cfg->row_first;
  rp.row_stride = cfg->row_stride;
  rp.S = cfg->sample_batch_size;
  rp.max_batches = cfg->max_sample_batches;
  rp.min_batches = cfg->min_sample_batches;
  rp.max_depth = cfg->max_depth;
  rp.pixel_radius = cfg->pixel_radius;
  rp.gamma = cfg->gamma_correction;
  rp.filter_b = cfg->filter_b;
  rp.filter_c = cfg->filter_c;
  rp.seed = cfg->seed;
  rp.n_pix = shard_rows(cfg) * cfg->img_width;
  rp.flags = cfg->flags;
  return rp;
}

int grid_for(int64_t n, int block, int max_blocks) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

size_t stack_bytes(const RtSceneHandle *h, int block) { return (size_t)std::max(h->dev.stack_depth, 1) * (size_t)block * sizeof(uint32_t); }

}  // namespace

int rt_hip_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e == hipErrorNoDevice) return 0;
  if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  return n;
}

int rt_hip_scene_free(RtSceneHandle *h) {
  if (!h) return RT_OK;
  (void)hipSetDevice(h->device);
  for (void *p : h->allocations) (void)hipFree(p);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return RT_OK;
}

int rt_hip_scene_upload(const RtScene *s, int device, RtSceneHandle **out) {
  g_error.clear();
  if (!s || !out) return fail(RT_ERR_BAD_ARG, "scene or out is NULL");
  *out = nullptr;
  // ---- validate every index before anything reaches a kernel
  if (s->n_nodes <= 0 || s->n_elements <= 0 || !s->nodes || !s->elements) return fail(RT_ERR_BAD_ARG, "scene has no nodes or no elements");
  if (s->n_triangles < 0 || s->n_quads < 0 || s->n_spheres < 0 || s->n_materials <= 0 || s->n_attractors < 0 || !s->materials)
    return fail(RT_ERR_BAD_ARG, "negative count or no materials");
  if ((s->n_triangles && !s->triangles) || (s->n_quads && !s->quads) || (s->n_spheres && !s->spheres) || (s->n_attractors && !s->attractors))
    return fail(RT_ERR_BAD_ARG, "count without array");
  if ((uint32_t)s->n_nodes > kRefIndexMask || (uint32_t)s->n_triangles > kRefIndexMask || (uint32_t)s->n_quads > kRefIndexMask ||
      (uint32_t)s->n_spheres > kRefIndexMask)
    return fail(RT_ERR_UNSUPPORTED, "more than 2^28-1 nodes or primitives of one kind");
  for (int i = 0; i < s->n_elements; i++) {
    RtElementRef e = s->elements[i];
    int lim = e.kind == RT_PRIM_TRIANGLE ? s->n_triangles : e.kind == RT_PRIM_QUAD ? s->n_quads : e.kind == RT_PRIM_SPHERE ? s->n_spheres : -1;
    if (e.index < 0 || e.index >= lim) return fail(RT_ERR_BAD_ARG, "element " + std::to_string(i) + ": bad kind or index");
  }
  auto bad_mat = [&](int m) { return m < 0 || m >= s->n_materials; };
  for (int i = 0; i < s->n_triangles; i++)
    if (bad_mat(s->triangles[i].material)) return fail(RT_ERR_BAD_ARG, "triangle material index out of range");
  for (int i = 0; i < s->n_quads; i++)
    if (bad_mat(s->quads[i].material)) return fail(RT_ERR_BAD_ARG, "quad material index out of range");
  for (int i = 0; i < s->n_spheres; i++)
    if (bad_mat(s->spheres[i].material)) return fail(RT_ERR_BAD_ARG, "sphere material index out of range");
  for (int i = 0; i < s->n_materials; i++)
    if (s->materials[i].kind < RT_MAT_LAMBERTIAN || s->materials[i].kind > RT_MAT_DIFFUSE_LIGHT) return fail(RT_ERR_BAD_ARG, "bad material kind");
  for (int i = 0; i < s->n_attractors; i++) {
    int e = s->attractors[i];
    if (e < 0 || e >= s->n_elements) return fail(RT_ERR_BAD_ARG, "attractor index out of range");
  }
  // nodes: children must lie after their parent (the reference's vector is
  // preorder), which also rules out cycles; leaves must name a valid element.
  std::vector<int32_t> internal_index((size_t)s->n_nodes, -1);
  int32_t n_internal = 0;
  for (int i = 0; i < s->n_nodes; i++) {
    const RtNode &n = s->nodes[i];
    if (n.is_leaf) {
      if (n.element_index < 0 || n.element_index >= s->n_elements) return fail(RT_ERR_BAD_ARG, "leaf element index out of range");
    } else {
      if (n.child_l_index <= i || n.child_l_index >= s->n_nodes || n.child_r_index <= i || n.child_r_index >= s->n_nodes)
        return fail(RT_ERR_BAD_ARG, "node " + std::to_string(i) + ": child index must be in (self, n_nodes)");
      internal_index[(size_t)i] = n_internal++;
    }
  }
  // depth of every node reachable from the root (children come later in the array)
  std::vector<int32_t> depth((size_t)s->n_nodes, -1);
  depth[0] = 0;
  int32_t max_depth = 0;
  for (int i = 0; i < s->n_nodes; i++) {
    if (depth[(size_t)i] < 0) continue;
    const RtNode &n = s->nodes[i];
    if (n.is_leaf) continue;
    int32_t d = depth[(size_t)i] + 1;
    depth[(size_t)n.child_l_index] = std::max(depth[(size_t)n.child_l_index], d);
    depth[(size_t)n.child_r_index] = std::max(depth[(size_t)n.child_r_index], d);
    max_depth = std::max(max_depth, d);
  }
  if (max_depth > kMaxStackDepth) return fail(RT_ERR_UNSUPPORTED, "BVH deeper than 64 levels");

  // ---- flatten
  auto ref_of_element = [&](int e) -> uint32_t {
    RtElementRef er = s->elements[e];
    return kRefLeaf | ((uint32_t)er.kind << kRefKindShift) | (uint32_t)er.index;
  };
  auto ref_of_node = [&](int i) -> uint32_t {
    const RtNode &n = s->nodes[i];
    return n.is_leaf ? ref_of_element(n.element_index) : (uint32_t)internal_index[(size_t)i];
  };
  std::vector<DevNode> nodes((size_t)n_internal);
  for (int i = 0; i < s->n_nodes; i++) {
    const RtNode &n = s->nodes[i];
    if (n.is_leaf) continue;
    DevNode &d = nodes[(size_t)internal_index[(size_t)i]];
    for (int a = 0; a < 3; a++) {
      d.lmin[a] = round_down(n.child_l_min[a]);
      d.lmax[a] = round_up(n.child_l_max[a]);
      d.rmin[a] = round_down(n.child_r_min[a]);
      d.rmax[a] = round_up(n.child_r_max[a]);
    }
    d.left = ref_of_node(n.child_l_index);
    d.right = ref_of_node(n.child_r_index);
    d.pad[0] = d.pad[1] = 0;
  }
  std::vector<DevTriangle> tris((size_t)s->n_triangles);
  std::vector<double> tri_normals((size_t)s->n_triangles * 3);
  std::vector<DevQuad> quads((size_t)s->n_quads);
  std::vector<DevSphere> spheres((size_t)s->n_spheres);
  for (int e = 0; e < s->n_elements; e++) {
    RtElementRef er = s->elements[e];
    if (er.kind == RT_PRIM_TRIANGLE) {
      const RtTriangle &t = s->triangles[er.index];
      DevTriangle &d = tris[(size_t)er.index];
      d.v0[0] = t.v0.x; d.v0[1] = t.v0.y; d.v0[2] = t.v0.z;
      d.e1[0] = t.v0v1.x; d.e1[1] = t.v0v1.y; d.e1[2] = t.v0v1.z;
      d.e2[0] = t.v0v2.x; d.e2[1] = t.v0v2.y; d.e2[2] = t.v0v2.z;
      d.element = (uint32_t)e;
      d.material = (uint32_t)t.material;
      tri_normals[3 * (size_t)er.index + 0] = t.normal.x;
      tri_normals[3 * (size_t)er.index + 1] = t.normal.y;
      tri_normals[3 * (size_t)er.index + 2] = t.normal.z;
    } else if (er.kind == RT_PRIM_QUAD) {
      const RtQuad &q = s->quads[er.index];
      DevQuad &d = quads[(size_t)er.index];
      d.q[0] = q.q.x; d.q[1] = q.q.y; d.q[2] = q.q.z;
      d.u[0] = q.u.x; d.u[1] = q.u.y; d.u[2] = q.u.z;
      d.v[0] = q.v.x; d.v[1] = q.v.y; d.v[2] = q.v.z;
      d.normal[0] = q.normal.x; d.normal[1] = q.normal.y; d.normal[2] = q.normal.z;
      d.w[0] = q.w.x; d.w[1] = q.w.y; d.w[2] = q.w.z;
      d.d = q.d;
      d.area = q.area;
      d.element = (uint32_t)e;
      d.material = (uint32_t)q.material;
    } else {
      const RtSphere &sp = s->spheres[er.index];
      DevSphere &d = spheres[(size_t)er.index];
      d.center[0] = sp.center.x; d.center[1] = sp.center.y; d.center[2] = sp.center.z;
      d.radius = sp.radius;
      d.element = (uint32_t)e;
      d.material = (uint32_t)sp.material;
    }
  }
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
  for (int i = 0; i < s->n_attractors; i++) attractors[(size_t)i] = ref_of_element(s->attractors[i]);

  // ---- upload
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(RT_ERR_BAD_ARG, "no such device");
  HIP_TRY(hipSetDevice(device));
  RtSceneHandle *h = new (std::nothrow) RtSceneHandle;
  if (!h) return fail(RT_ERR_OOM, "out of host memory");
  h->device = device;
  h->n_elements = s->n_elements;
  h->bvh_depth = max_depth;
  int rc = RT_OK;
  auto step = [&](int r) {
    if (rc == RT_OK) rc = r;
  };
  step(dev_upload(h, &h->dev.nodes, nodes));
  step(dev_upload(h, &h->dev.triangles, tris));
  step(dev_upload(h, &h->dev.tri_normals, tri_normals));
  step(dev_upload(h, &h->dev.quads, quads));
  step(dev_upload(h, &h->dev.spheres, spheres));
  step(dev_upload(h, &h->dev.materials, mats));
  step(dev_upload(h, &h->dev.attractors, attractors));
  if (rc == RT_OK && hipStreamCreate(&h->stream) != hipSuccess) rc = fail(RT_ERR_HIP, "hipStreamCreate failed");
  if (rc != RT_OK) {
    std::string keep = g_error;
    rt_hip_scene_free(h);
    g_error = keep;
    return rc;
  }
  h->dev.n_attractors = s->n_attractors;
  h->dev.root = ref_of_node(0);
  h->dev.stack_depth = std::max(max_depth, 1);
  h->dev.camera = s->camera;
  *out = h;
  return RT_OK;
}

int rt_hip_render_resident(RtSceneHandle *h, const RtConfig *cfg, RtImage *out, RtStats *stats) {
  g_error.clear();
  if (!h || !out || !out->pixels) return fail(RT_ERR_BAD_ARG, "handle, out or out->pixels is NULL");
  int rc = check_config(cfg);
  if (rc != RT_OK) return rc;
  if (out->memory != RT_MEM_HOST && out->memory != RT_MEM_DEVICE) return fail(RT_ERR_BAD_ARG, "bad RtImage.memory");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  RenderParams rp = make_params(cfg);
  const int S = rp.S;
  const int32_t n_pix = rp.n_pix;
  const bool timed = (rp.flags & kFlagTimeKernels) != 0;
  const bool counted = (rp.flags & kFlagCountVisits) != 0;
  auto wall0 = std::chrono::steady_clock::now();

  // paths per chunk: bounded so the SoA queues stay a few GB
  int64_t max_paths = 1ll << 24;
  if (const char *env = getenv("RT_HIP_MAX_PATHS")) {
    long long v = atoll(env);
    if (v >= 64) max_paths = v;
  }
  int32_t chunk_slots = (int32_t)std::max<int64_t>(1, std::min<int64_t>(n_pix, max_paths / S));
  int64_t chunk_paths = (int64_t)chunk_slots * S;
  if (chunk_paths > 0xfffffff0ll) return fail(RT_ERR_UNSUPPORTED, "sample_batch_size too large for one chunk");

  Scratch sc;
  PathQueue q[2];
  for (int k = 0; k < 2; k++) {
    HIP_TRY(sc.alloc(&q[k].ox, chunk_paths)); HIP_TRY(sc.alloc(&q[k].oy, chunk_paths)); HIP_TRY(sc.alloc(&q[k].oz, chunk_paths));
    HIP_TRY(sc.alloc(&q[k].dx, chunk_paths)); HIP_TRY(sc.alloc(&q[k].dy, chunk_paths)); HIP_TRY(sc.alloc(&q[k].dz, chunk_paths));
    HIP_TRY(sc.alloc(&q[k].tx, chunk_paths)); HIP_TRY(sc.alloc(&q[k].ty, chunk_paths)); HIP_TRY(sc.alloc(&q[k].tz, chunk_paths));
    HIP_TRY(sc.alloc(&q[k].rng, chunk_paths));
    HIP_TRY(sc.alloc(&q[k].pid, chunk_paths));
  }
  HitBuffer hits;
  HIP_TRY(sc.alloc(&hits.t, chunk_paths));
  HIP_TRY(sc.alloc(&hits.ref, chunk_paths));
  double *radiance, *weights, *hist;
  HIP_TRY(sc.alloc(&radiance, 3 * chunk_paths));
  HIP_TRY(sc.alloc(&weights, chunk_paths));
  HIP_TRY(sc.alloc(&hist, (size_t)rp.max_batches * 4 * (size_t)n_pix));
  uint32_t *active[2], *counters;
  HIP_TRY(sc.alloc(&active[0], n_pix));
  HIP_TRY(sc.alloc(&active[1], n_pix));
  // counters: [0] next_count, [1 .. max_depth+1] queue sizes per bounce
  const int n_counters = rp.max_depth + 2;
  HIP_TRY(sc.alloc(&counters, n_counters));
  unsigned long long *totals;  // [0..15] rays by depth, [16] node visits, [17] leaf tests
  HIP_TRY(sc.alloc(&totals, RT_MAX_DEPTH_STATS + 2));
  HIP_TRY(hipMemsetAsync(totals, 0, (RT_MAX_DEPTH_STATS + 2) * sizeof(unsigned long long), st));

  PixelOutputs po{};
  if (out->memory == RT_MEM_DEVICE) {
    po.pixels = out->pixels;
    po.sample_pixels = out->sample_pixels;
    po.linear = out->linear;
    po.samples = out->samples;
  } else {
    HIP_TRY(sc.alloc(&po.pixels, 3 * (size_t)n_pix));
    if (out->sample_pixels) HIP_TRY(sc.alloc(&po.sample_pixels, 3 * (size_t)n_pix));
    if (out->linear) HIP_TRY(sc.alloc(&po.linear, 3 * (size_t)n_pix));
    if (out->samples) HIP_TRY(sc.alloc(&po.samples, (size_t)n_pix));
  }

  const int block = 256;
  const int max_blocks = 256 * 8;  // 256 CUs x 8 resident workgroups
  size_t lds = stack_bytes(h, block);
  hipLaunchKernelGGL(k_iota, dim3(grid_for(n_pix, block, max_blocks)), dim3(block), 0, st, active[0], (uint32_t)n_pix);

  struct TimedLaunch {
    hipEvent_t a, b;
    int kind;  // 0 extend, 1 shade, 2 other
  };
  std::vector<TimedLaunch> timeline;
  auto mark_begin = [&](int kind) -> hipError_t {
    if (!timed) return hipSuccess;
    TimedLaunch t{};
    t.kind = kind;
    hipError_t e = hipEventCreate(&t.a);
    if (e != hipSuccess) return e;
    sc.events.push_back(t.a);
    e = hipEventCreate(&t.b);
    if (e != hipSuccess) return e;
    sc.events.push_back(t.b);
    timeline.push_back(t);
    return hipEventRecord(t.a, st);
  };
  auto mark_end = [&]() -> hipError_t { return timed ? hipEventRecord(timeline.back().b, st) : hipSuccess; };

  uint32_t n_active = (uint32_t)n_pix;
  int cur = 0;
  uint64_t batches = 0, samples = 0;
  for (int b = 0; b < rp.max_batches && n_active > 0; b++) {
    HIP_TRY(hipMemsetAsync(counters, 0, sizeof(uint32_t), st));
    for (uint32_t base = 0; base < n_active; base += (uint32_t)chunk_slots) {
      int32_t ns = (int32_t)std::min<uint32_t>((uint32_t)chunk_slots, n_active - base);
      int64_t np = (int64_t)ns * S;
      HIP_TRY(hipMemsetAsync(counters + 1, 0, (size_t)(rp.max_depth + 1) * sizeof(uint32_t), st));
      int g = grid_for(np, block, max_blocks);
      HIP_TRY(mark_begin(2));
      hipLaunchKernelGGL(k_generate, dim3(g), dim3(block), 0, st, h->dev, rp, active[cur] + base, ns, b, q[0], weights, counters + 1);
      HIP_TRY(mark_end());
      for (int k = 0; k < rp.max_depth; k++) {
        HIP_TRY(mark_begin(0));
        if (counted)
          hipLaunchKernelGGL(k_extend<true>, dim3(g), dim3(block), lds, st, h->dev, q[k & 1], counters + 1 + k, hits, totals + RT_MAX_DEPTH_STATS);
        else
          hipLaunchKernelGGL(k_extend<false>, dim3(g), dim3(block), lds, st, h->dev, q[k & 1], counters + 1 + k, hits, totals + RT_MAX_DEPTH_STATS);
        HIP_TRY(mark_end());
        HIP_TRY(mark_begin(1));
        hipLaunchKernelGGL(k_shade, dim3(g), dim3(block), 0, st, h->dev, q[k & 1], counters + 1 + k, hits, q[(k + 1) & 1], counters + 2 + k,
                           radiance, ns, S, (int32_t)(k == rp.max_depth - 1), rp.flags);
        HIP_TRY(mark_end());
      }
      HIP_TRY(mark_begin(2));
      if (rp.max_depth == 0) HIP_TRY(hipMemsetAsync(radiance, 0, 3 * (size_t)np * sizeof(double), st));
      hipLaunchKernelGGL(k_accumulate, dim3(grid_for(ns, block, max_blocks)), dim3(block), 0, st, rp, active[cur] + base, ns, b, radiance, weights,
                         hist, active[cur ^ 1], counters, po, counters + 1, totals);
      HIP_TRY(mark_end());
      samples += (uint64_t)np;
    }
    HIP_TRY(hipGetLastError());
    uint32_t next = 0;
    HIP_TRY(hipMemcpyAsync(&next, counters, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    n_active = next;
    cur ^= 1;
    batches++;
  }
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());

  if (out->memory == RT_MEM_HOST) {
    HIP_TRY(hipMemcpy(out->pixels, po.pixels, 3 * (size_t)n_pix * sizeof(double), hipMemcpyDeviceToHost));
    if (out->sample_pixels) HIP_TRY(hipMemcpy(out->sample_pixels, po.sample_pixels, 3 * (size_t)n_pix * sizeof(double), hipMemcpyDeviceToHost));
    if (out->linear) HIP_TRY(hipMemcpy(out->linear, po.linear, 3 * (size_t)n_pix * sizeof(double), hipMemcpyDeviceToHost));
    if (out->samples) HIP_TRY(hipMemcpy(out->samples, po.samples, (size_t)n_pix * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  if (stats) {
    memset(stats, 0, sizeof *stats);
    unsigned long long host_totals[RT_MAX_DEPTH_STATS + 2];
    HIP_TRY(hipMemcpy(host_totals, totals, sizeof host_totals, hipMemcpyDeviceToHost));
    for (int k = 0; k < RT_MAX_DEPTH_STATS; k++) {
      stats->rays_by_depth[k] = host_totals[k];
      stats->rays += host_totals[k];
    }
    stats->node_visits = host_totals[RT_MAX_DEPTH_STATS];
    stats->leaf_tests = host_totals[RT_MAX_DEPTH_STATS + 1];
    stats->box_tests = 2 * stats->node_visits;
    stats->samples = samples;
    stats->batches = batches;
    for (const TimedLaunch &t : timeline) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
      if (t.kind == 0) {
        stats->trace_kernel_ms += ms;
        stats->trace_launches++;
      } else {
        stats->other_kernel_ms += ms;
      }
    }
    stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  }
  return RT_OK;
}

int rt_hip_render(const RtScene *scene, const RtConfig *cfg, RtImage *out, RtStats *stats) {
  RtSceneHandle *h = nullptr;
  int rc = rt_hip_scene_upload(scene, 0, &h);
  if (rc != RT_OK) return rc;
  rc = rt_hip_render_resident(h, cfg, out, stats);
  std::string keep = g_error;
  rt_hip_scene_free(h);
  g_error = keep;
  return rc;
}

int rt_hip_primary_hits(RtSceneHandle *h, const RtConfig *cfg, int32_t *prim_id, double *t) {
  g_error.clear();
  if (!h || !prim_id || !t) return fail(RT_ERR_BAD_ARG, "NULL argument");
  int rc = check_config(cfg);
  if (rc != RT_OK) return rc;
  HIP_TRY(hipSetDevice(h->device));
  RenderParams rp = make_params(cfg);
  int64_t n = (int64_t)rp.n_pix * rp.S;
  Scratch sc;
  int32_t *d_id;
  double *d_t;
  HIP_TRY(sc.alloc(&d_id, (size_t)n));
  HIP_TRY(sc.alloc(&d_t, (size_t)n));
  const int block = 256;
  hipLaunchKernelGGL(k_primary, dim3(grid_for(n, block, 2048)), dim3(block), stack_bytes(h, block), h->stream, h->dev, rp, d_id, d_t);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(prim_id, d_id, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(t, d_t, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_hip_trace_rays(RtSceneHandle *h, int64_t n, const double *origins, const double *directions, int32_t *prim_id, double *t) {
  g_error.clear();
  if (!h || n < 0 || (n > 0 && (!origins || !directions || !prim_id || !t))) return fail(RT_ERR_BAD_ARG, "NULL argument or negative count");
  if (n == 0) return RT_OK;
  HIP_TRY(hipSetDevice(h->device));
  Scratch sc;
  double *d_o, *d_d, *d_t;
  int32_t *d_id;
  HIP_TRY(sc.alloc(&d_o, 3 * (size_t)n));
  HIP_TRY(sc.alloc(&d_d, 3 * (size_t)n));
  HIP_TRY(sc.alloc(&d_t, (size_t)n));
  HIP_TRY(sc.alloc(&d_id, (size_t)n));
  HIP_TRY(hipMemcpy(d_o, origins, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_d, directions, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  const int block = 256;
  hipLaunchKernelGGL(k_trace, dim3(grid_for(n, block, 2048)), dim3(block), stack_bytes(h, block), h->stream, h->dev, n, d_o, d_d, d_id, d_t);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(prim_id, d_id, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(t, d_t, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return RT_OK;
}
