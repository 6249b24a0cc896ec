// rt_hip.hip -- MI355X (gfx950) implementation of the C ABI in
// include/rt_hip.h: a wavefront path tracer that replaces the reference's
// per-row loop (/root/reference/src/render/integrator.rs:226-339).
//
// Pipeline per round (one round = several consecutive `b` iterations of
// integrator.rs:279 for every still-active pixel; batches traced beyond a
// pixel's stopping batch are dropped):
//
//   k_generate    one lane per (active pixel, batch, sample i): Sobol offset,
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
//   radix sort    of the surviving pixels' keys: the active list stays in its
//                 scattered-tile order (pixel_of_key)
//
// Path state lives in SoA queues in HBM; every array is indexed so that a
// wave touches consecutive addresses.
#include <hip/hip_runtime.h>
#include <hipcub/device/device_radix_sort.hpp>
#include <hipcub/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
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

// After a kernel launch: a launch that was refused (too much LDS, bad grid) is reported at once, by the kernel's name.
#define LAUNCHED(name)                                                                               \
  do {                                                                                               \
    hipError_t e_ = hipGetLastError();                                                               \
    if (e_ != hipSuccess) return fail(RT_ERR_HIP, std::string("launch of " name ": ") + hipGetErrorString(e_)); \
  } while (0)

constexpr int32_t kFlagTimeKernels = RT_FLAG_TIME_KERNELS;
constexpr int32_t kFlagCountVisits = RT_FLAG_COUNT_VISITS;
constexpr int32_t kFlagEarlyTerminate = RT_FLAG_EARLY_TERMINATE;
constexpr int32_t kFlagReferenceVisitRule = RT_FLAG_REFERENCE_VISIT_RULE;
constexpr int32_t kFlagReusePrimaryHits = RT_FLAG_REUSE_PRIMARY_HITS;

// ==================================================================== queues
// SoA path state.  One "queue" holds the rays about to be traced at a bounce.
// A queue is split into kRegions regions of `cap` entries each; region r holds
// entries [r*cap, r*cap + count[r]).  A path stays in its region for its whole
// life, every region has its own append counter (one global atomic word serves
// only ~90 M appends/s, far less than the kernels produce), and region r is
// always processed by workgroups with blockIdx % kRegions == r, i.e. by one XCD.
#ifndef RT_TILE_LOG
#define RT_TILE_LOG 3
#endif
constexpr int kRegions = 32;
constexpr int kCounterStride = 32;  // uint32 words between two counters: one 128-byte line each

struct PathQueue {
  double *ox, *oy, *oz;  // origin
  double *dx, *dy, *dz;  // normalised direction
  double *tx, *ty, *tz;  // throughput
  uint64_t *rng;         // SplitMix64 counter
  uint32_t *pid;         // path id within the round
  uint32_t cap;          // entries per region
};

// Where entry j of n goes when a kernel fills a queue from scratch: regions get
// equal contiguous shares (whole waves), and their counts follow from n alone.
__device__ __forceinline__ uint32_t region_share(int64_t n) { return (uint32_t)(((n + kRegions - 1) / kRegions + 63) / 64 * 64); }
__device__ __forceinline__ uint32_t region_place(int64_t j, uint32_t share, uint32_t cap) {
  uint32_t r = (uint32_t)(j / share);
  return r * cap + (uint32_t)(j - (int64_t)r * share);
}
__device__ __forceinline__ void region_publish_counts(int64_t n, uint32_t share, uint32_t *counts) {
  // called by one workgroup: thread r < kRegions publishes region r's count
  if (threadIdx.x < kRegions) {
    int64_t left = n - (int64_t)threadIdx.x * share;
    counts[threadIdx.x * kCounterStride] = (uint32_t)(left < 0 ? 0 : (left > share ? share : left));
  }
}

struct HitBuffer {
  double *t;
  uint32_t *ref;
};

struct RenderParams {
  int32_t width, height, row_first, row_stride, row_block;
  int32_t S;  // sample_batch_size
  int32_t max_batches, min_batches, max_depth;
  double pixel_radius, gamma, filter_b, filter_c;
  uint64_t seed;
  int32_t n_pix;  // pixels in the shard
  int32_t flags;
  // order of the pixels in the active list (see pixel_of_key)
  uint32_t tiles_x, n_tiles, tile_mult, tile_inv;
  uint32_t tile_wlog, tile_hlog;  // a tile is 2^wlog pixels wide and 2^hlog rows of the shard high (64 pixels)
  // The per-pixel working arrays (filter weights, running sums, per-batch history) are indexed by KEY, not by pixel: neighbouring
  // entries of the active list have neighbouring keys, so a wave's 64 lanes read 512 consecutive bytes of them whatever the shape
  // of a tile in the image (indexed by pixel, an 8 x 8 tile is eight runs of 64 bytes: generate + accumulate 51 -> 42 ms).
  uint64_t n_keys;  // n_tiles * 64 >= n_pix (tiles at the image's edge hold keys no pixel has)
};

// The active list holds sort keys, not pixel numbers, and is kept sorted.  A key names a 64-pixel
// tile of the shard and a pixel in it (8x8 for a whole image; wider and flatter when the shard
// holds every n-th row only, so that the tile stays compact in the image):
// key = rank(tile) * 64 + (row % h) * w + (x % w), where
// rank(tile) = tile * tile_mult mod n_tiles scatters consecutive tiles over the whole list.
// Neighbouring entries are therefore neighbouring pixels (coherent rays within a wave and a
// claim), while every contiguous slice of the list -- a queue region, served by one XCD and by a
// fixed set of k_shade workgroups -- is a fair sample of the image: cheap and costly parts of
// the picture do not pile up in different regions.
constexpr uint32_t kTileLog = RT_TILE_LOG;  // log2 of the side of a square tile; a tile has 4^kTileLog pixels
// image row of row `k` of the shard (RtConfig.row_first / row_stride / row_block)
__device__ __forceinline__ int32_t image_row(const RenderParams &rp, uint32_t k) {
  uint32_t blk = k / (uint32_t)rp.row_block;
  return (rp.row_first + (int32_t)blk * rp.row_stride) * rp.row_block + (int32_t)(k - blk * (uint32_t)rp.row_block);
}
__device__ __forceinline__ uint32_t pixel_of_key(const RenderParams &rp, uint32_t key) {
  uint32_t within = key & ((1u << (2 * kTileLog)) - 1u), rank = key >> (2 * kTileLog);
  uint32_t tile = (uint32_t)(((uint64_t)rank * rp.tile_inv) % rp.n_tiles);
  uint32_t ty = tile / rp.tiles_x, tx = tile - ty * rp.tiles_x;
  return ((ty << rp.tile_hlog) + (within >> rp.tile_wlog)) * (uint32_t)rp.width + (tx << rp.tile_wlog) + (within & ((1u << rp.tile_wlog) - 1u));
}
__device__ __forceinline__ uint32_t key_of_pixel(const RenderParams &rp, uint32_t p) {
  uint32_t row = p / (uint32_t)rp.width, x = p - row * (uint32_t)rp.width;
  uint32_t tile = (row >> rp.tile_hlog) * rp.tiles_x + (x >> rp.tile_wlog);
  uint32_t rank = (uint32_t)(((uint64_t)tile * rp.tile_mult) % rp.n_tiles);
  return (rank << (2 * kTileLog)) + ((row & ((1u << rp.tile_hlog) - 1u)) << rp.tile_wlog) + (x & ((1u << rp.tile_wlog) - 1u));
}

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
// The path id is (slot * G + g) * S + i, so the samples of one batch of one pixel are adjacent: coherent primary
// rays, and paths that end at the same bounce store neighbouring result records.  The Sobol index restarts with
// every batch (integrator.rs:285): sample i of a pixel has the same offset, filter weight and -- without a lens --
// the same direction in every batch; the paths of its G batches in a round differ in their random-number stream only
// (and, with a lens, in the origin it draws).  The weight is stored once, by batch 0, as weights[i * n_keys + key].
struct GenSample {  // what the batches of one (pixel, sample i) share
  d3 pixel_sample, dir;
  uint64_t pixel_key;
};
RT_DEV GenSample generate_sample(const DevScene &sc, const RenderParams &rp, const uint32_t *active, uint32_t slot, uint32_t i, int32_t batch0, double *weights) {
  const uint32_t key = active[slot];
  const uint32_t p = pixel_of_key(rp, key);
  const int32_t x = (int32_t)(p % (uint32_t)rp.width);
  const int32_t y = image_row(rp, p / (uint32_t)rp.width);
  // integrator.rs:275: pixel_num = x * y (product), truncated to u32 as the Sobol seed
  const uint32_t sobol_seed = (uint32_t)((uint64_t)x * (uint64_t)y);
  const double sx = ((double)sobol_sample(i, 0, sobol_seed) * rp.pixel_radius) - rp.pixel_radius / 2.;
  const double sy = ((double)sobol_sample(i, 1, sobol_seed) * rp.pixel_radius) - rp.pixel_radius / 2.;
  if (batch0 == 0) {
    const double inv_r = 1.0 / rp.pixel_radius;  // filter.rs:33-42
    weights[(size_t)i * (size_t)rp.n_keys + key] = mitchell_1d(sx * inv_r, rp.filter_b, rp.filter_c) * mitchell_1d(sy * inv_r, rp.filter_b, rp.filter_c);
  }
  // camera.rs:95-120
  const d3 pixel_loc = ld3(sc.camera.pixel00_loc) + (ld3(sc.camera.pixel_delta_u) * (double)x) + (ld3(sc.camera.pixel_delta_v) * (double)y);
  GenSample gs;
  gs.pixel_sample = pixel_loc + D3(sx, sy, 0.);
  gs.dir = normalized(gs.pixel_sample - ld3(sc.camera.camera_center));
  gs.pixel_key = rng_seed_pixel((uint64_t)y * (uint64_t)rp.width + (uint64_t)x);
  return gs;
}
// Primary rays traced once per round (RT_FLAG_REUSE_PRIMARY_HITS, cameras without a lens): the G batches of a round send the SAME
// S rays through a pixel, so the rays go into a compact queue of their own -- one entry per (pixel slot, sample), entry
// slot * S + i dealt over the regions like any queue -- and every path only remembers which compact entry is its ray.
struct PrimaryReuse {
  double *dx, *dy, *dz;  // the compact queue's directions (origins: the camera centre)
  uint32_t *entry;       // per path (indexed like the path queue): its ray's entry in the compact queue
  uint32_t *counts;      // the compact queue's entries per region
  uint32_t cap;          // its entries per region; 0: no reuse, every path carries its own ray
};
RT_DEV void generate_store(const DevScene &sc, const RenderParams &rp, const GenSample &gs, uint32_t batch, uint32_t i, uint32_t j, uint32_t e, const PathQueue &q,
                           const PrimaryReuse &pr, uint32_t entry) {
  uint64_t rng = rng_seed_path_of(gs.pixel_key, (uint64_t)batch, (uint64_t)i, rp.seed);
  if (pr.cap) {
    pr.entry[e] = entry;
  } else {
    d3 dir = gs.dir;
    // without a lens every primary ray starts at the camera centre: the first bounce's kernels know
    if (!(sc.camera.defocus_angle <= 0.0)) {
      d3 pd = random_in_unit_disk(rng);
      d3 origin = ld3(sc.camera.camera_center) + (ld3(sc.camera.defocus_disk_u) * pd.x) + (ld3(sc.camera.defocus_disk_v) * pd.y);
      dir = normalized(gs.pixel_sample - origin);
      q.ox[e] = origin.x; q.oy[e] = origin.y; q.oz[e] = origin.z;
    }
    q.dx[e] = dir.x; q.dy[e] = dir.y; q.dz[e] = dir.z;
  }
  // throughput starts at one: bounce 0 of k_shade knows, nothing is stored
  q.rng[e] = rng;
  q.pid[e] = j;
}
// A workgroup takes `spb` consecutive pixel slots: its first spb * S lanes do the shared arithmetic of one
// (slot, sample) each and leave it in LDS, then all lanes write the slots' spb * G * S paths, which are contiguous
// in the queue, in order: whole lines per store instruction.
constexpr int kGenMaxShared = 512;  // (slot, sample) pairs per workgroup; S above it takes k_generate_strided
__global__ void __launch_bounds__(256) k_generate(DevScene sc, RenderParams rp, const uint32_t *active, int32_t n_slots, int32_t spb,
                                                   int32_t batch0, int32_t G, PathQueue q, double *weights, uint32_t *qcounts, PrimaryReuse pr) {
  __shared__ double sh_f64[3][kGenMaxShared];
  __shared__ uint64_t sh_key[kGenMaxShared];
  __shared__ uint32_t sh_entry[kGenMaxShared];
  // n fits 32 bits (the host caps a round below 2^32 / 3 paths): 32-bit arithmetic throughout
  const int64_t n = (int64_t)n_slots * G * rp.S;
  const uint32_t share = region_share(n);
  const int64_t n_rays = (int64_t)n_slots * rp.S;  // entries of the compact queue
  const uint32_t share_rays = region_share(n_rays);
  if (blockIdx.x == 0) {
    region_publish_counts(n, share, qcounts);
    if (pr.cap) region_publish_counts(n_rays, share_rays, pr.counts);
  }
  const uint32_t S = (uint32_t)rp.S, GS = (uint32_t)G * S;
  const bool lens = !(sc.camera.defocus_angle <= 0.0);
  const uint32_t n_chunks = ((uint32_t)n_slots + (uint32_t)spb - 1u) / (uint32_t)spb;
  for (uint32_t chunk = xcd_logical_block(blockIdx.x, gridDim.x); chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t slot0 = chunk * (uint32_t)spb, slots = min((uint32_t)spb, (uint32_t)n_slots - slot0);
    __syncthreads();  // the previous chunk's readers are done
    for (uint32_t k = threadIdx.x; k < slots * S; k += blockDim.x) {
      const uint32_t sl = k / S, i = k - sl * S;
      GenSample gs = generate_sample(sc, rp, active, slot0 + sl, i, batch0, weights);
      const d3 v = lens ? gs.pixel_sample : gs.dir;
      sh_f64[0][k] = v.x; sh_f64[1][k] = v.y; sh_f64[2][k] = v.z;
      sh_key[k] = gs.pixel_key;
      if (pr.cap) {  // this lane's (slot, sample) is one ray of the compact queue
        const uint32_t entry = region_place((int64_t)slot0 * S + k, share_rays, pr.cap);
        sh_entry[k] = entry;
        pr.dx[entry] = v.x; pr.dy[entry] = v.y; pr.dz[entry] = v.z;
      }
    }
    __syncthreads();
    const uint32_t j0 = slot0 * GS, count = slots * GS;
    const uint32_t r0 = j0 / share;
    const bool one_boundary = count <= share;  // the chunk's paths lie in region r0 and at most the next one
    for (uint32_t idx = threadIdx.x; idx < count; idx += blockDim.x) {
      const uint32_t sl = idx / GS, rem = idx - sl * GS, g = rem / S, i = rem - g * S, k = sl * S + i;
      const uint32_t j = j0 + idx;
      uint32_t r = r0;
      if (one_boundary) r += (j >= (r0 + 1u) * share) ? 1u : 0u;
      else r = j / share;
      const uint32_t e = r * q.cap + (j - r * share);
      GenSample gs;
      gs.pixel_sample = gs.dir = D3(sh_f64[0][k], sh_f64[1][k], sh_f64[2][k]);
      gs.pixel_key = sh_key[k];
      generate_store(sc, rp, gs, (uint32_t)batch0 + g, i, j, e, q, pr, pr.cap ? sh_entry[k] : 0u);
    }
  }
}
// The same for batches of more than kGenMaxShared samples: one lane takes (active pixel slot, sample i), does the shared
// arithmetic once and writes the G paths of the round's batches (runs of S entries, G * S apart).
__global__ void __launch_bounds__(256) k_generate_strided(DevScene sc, RenderParams rp, const uint32_t *active, int32_t n_slots,
                                                           int32_t batch0, int32_t G, PathQueue q, double *weights, uint32_t *qcounts, PrimaryReuse pr) {
  const int64_t n = (int64_t)n_slots * G * rp.S;
  const uint32_t share = region_share(n);
  const int64_t n_rays = (int64_t)n_slots * rp.S;
  const uint32_t share_rays = region_share(n_rays);
  if (blockIdx.x == 0) {
    region_publish_counts(n, share, qcounts);
    if (pr.cap) region_publish_counts(n_rays, share_rays, pr.counts);
  }
  const uint32_t lb = xcd_logical_block(blockIdx.x, gridDim.x);
  const uint32_t S = (uint32_t)rp.S, n_si = (uint32_t)n_slots * S;
  for (uint32_t k = lb * blockDim.x + threadIdx.x; k < n_si; k += gridDim.x * blockDim.x) {
    const uint32_t slot = k / S, i = k - slot * S;
    const GenSample gs = generate_sample(sc, rp, active, slot, i, batch0, weights);
    uint32_t entry = 0;
    if (pr.cap) {
      entry = region_place((int64_t)k, share_rays, pr.cap);
      pr.dx[entry] = gs.dir.x; pr.dy[entry] = gs.dir.y; pr.dz[entry] = gs.dir.z;
    }
    // entry of the path of batch g: j = (slot * G + g) * S + i, region by region (region_place)
    uint32_t j = slot * (uint32_t)G * S + i;
    uint32_t r = j / share, off = j - r * share;
    for (int32_t g = 0; g < G; g++) {
      generate_store(sc, rp, gs, (uint32_t)(batch0 + g), i, j, r * q.cap + off, q, pr, entry);
      j += S;
      off += S;
      while (off >= share) {
        off -= share;
        r++;
      }
    }
  }
}

// ==================================================================== extend
// BVH closest-hit queries for one ray queue: the dominant kernel.
//
// * One wave per workgroup, persistent.  A wave claims rays in chunks of
//   kClaim from the regions of the queue (its own region first -- workgroups
//   with equal blockIdx % 8 share an XCD and so an L2 -- then the others), one
//   device atomic per chunk, and refills its idle lanes from the
//   chunk whenever `refill` of them have finished: a long traversal in one
//   lane does not idle the other 63, and there is a single drain per wave.
// * Traversal is split into node steps (two f32 slab tests on one 64-byte
//   record) and leaf steps (the reference's f64 primitive test).  A lane that
//   reaches a leaf queues it (up to `leafq` per lane) and keeps walking; the
//   queued tests run together once fewer than `node_min` lanes still have node
//   work, so neither kind of step runs with most lanes masked off.  Leaf tests
//   run in depth-first order, which is all the reference's result depends on.
// * Per-lane stacks and leaf queues live in LDS, laid out [level][lane].
extern __shared__ uint32_t g_lds[];

constexpr uint32_t kClaim = 512;
#ifndef RT_EXTRA_STEPS
#define RT_EXTRA_STEPS 2  // node steps a wave takes between two votes, minus one (C3 at six waves: 650 / 645 / 633 / 643 ms for 0..3)
#endif
#ifndef RT_EXTRA_STEPS_WIDE
#define RT_EXTRA_STEPS_WIDE 1  // the same for the walk on wide nodes
#endif
#ifndef RT_WIDE_STACK_MODE
#define RT_WIDE_STACK_MODE 1
#endif
#ifndef RT_SPLIT_LEAF
#define RT_SPLIT_LEAF 1
#endif
constexpr uint32_t kNoRay = 0xffffffffu;

struct ExtendTuning {
  int refill;    // refill idle lanes when at least this many are idle
  int node_min;  // run the queued leaf tests when fewer lanes than this have node work
  int leafq;     // deferred leaves per lane (1..kMaxLeafQueue)
  int claim;     // rays a wave takes from a region's counter at a time
  int tail;      // shrink the last claims of a region
  int shared_origin;  // the queue holds primary rays of a camera without a lens: origins are not stored
  int pass;      // 0: every ray; 1: leave rays near an out-of-box primitive to pass 2 (mark them kPending); 2: only those
};

template <bool QUIRK, bool COUNT, bool GRID, bool SPILL, int WAVES, bool WIDE = false>
__attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
__global__ void __launch_bounds__(64) k_extend(DevScene sc, PathQueue q, const uint32_t *counts, HitBuffer hits, uint32_t *work,
                                                ExtendTuning tune, unsigned long long *visit_counters) {
  const uint32_t lane = threadIdx.x;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int stride = 64;
  uint32_t *stack = g_lds + lane;
  static_assert(!WIDE || (GRID && !QUIRK), "wide nodes exist on the grid, for the order-free walk");
  // how the stack is held (rt_device.h, StackRef): all in LDS; the wide walk's levels beyond kWideLdsLevels in global memory; the
  // binary walks' newest kOvfLevels entries in LDS with blocks of older ones moved out
  constexpr int SM = !SPILL ? 0 : (WIDE ? RT_WIDE_STACK_MODE : 2);
  constexpr int kLdsl = SM == 1 ? kWideLdsLevels : kOvfLevels + 1;
  const uint32_t root = WIDE ? sc.root_w : sc.root;
  const int lds_levels = SPILL ? kLdsl : (WIDE ? sc.stack_depth_w : sc.stack_depth);
  uint32_t *leafq = g_lds + (size_t)lds_levels * 64 + lane;
  const StackRef sref = {stack, SM == 0 ? nullptr : sc.stack_spill + (size_t)blockIdx.x * 64 + (SM == 1 ? lane : 0u), gridDim.x * 64u};
  if (SM == 2) stack[kOvfLevels * 64] = 0u;  // blocks of this lane's stack outside LDS
  const uint32_t home = blockIdx.x % kRegions;
  const uint32_t waves_per_region = gridDim.x / kRegions;
  uint32_t region_try = 0;         // regions tried so far (wave-uniform)
  uint32_t w_next = 0, w_end = 0;  // the wave's claimed range of queue entries (wave-uniform)
  bool dry = false;                // every region is exhausted (wave-uniform)

  uint32_t j = kNoRay;
  d3 ro = D3(0, 0, 0), rd = D3(0, 0, 1);
  RayF f = {};
  RayP fp = {};
  RayQ fq = {};
  d3 rinv = D3(0, 0, 0);  // 1 / direction in f64: the walk on the reference's own boxes (QUIRK with DevNodeD records)
  Traversal tv;
  trav_begin(root, tv);
  tv.cur = kDone;
  tv.moves = 0;
  uint32_t nn = 0, nl = 0, wn = 0, wl = 0;  // lane and wave step counts (COUNT only)
  uint32_t d_idle = 0, d_done = 0, d_block = 0;
  unsigned long long cyc_refill = 0, cyc_node = 0, cyc_leaf = 0, cyc_mark = 0;  // shader cycles of the wave per phase (COUNT only)
  if (COUNT) cyc_mark = __builtin_amdgcn_s_memtime();
#define RT_PHASE_END(acc)                                         \
  if (COUNT) {                                                    \
    unsigned long long now_ = __builtin_amdgcn_s_memtime();       \
    acc += now_ - cyc_mark;                                       \
    cyc_mark = now_;                                              \
  }

  for (;;) {
    // ---------------------------------------------------------------- refill
    unsigned long long idle = __ballot(j == kNoRay);
    int n_idle = __popcll(idle);
    if (!dry && n_idle >= tune.refill) {
      while (w_next >= w_end && !dry) {  // claim a new chunk: own region first, then the others
        uint32_t r = (home + region_try) % kRegions;
        uint32_t r_count = counts[r * kCounterStride];
        uint32_t off = r_count;
        uint32_t taken = *(volatile uint32_t *)&work[r * kCounterStride];
        uint32_t claim = (uint32_t)tune.claim;
        if (taken < r_count) {
          // the last claims of a region are smaller: a launch ends when its slowest wave does, and
          // a wave that starts a full claim when the others run dry keeps the chip waiting
          uint32_t left = r_count - taken;
          if (tune.tail == 1 && left < waves_per_region * claim) claim = left < waves_per_region * (claim / 4) ? max(64u, claim / 8) : max(64u, claim / 4);
          if (tune.tail >= 2) claim = max(64u, min(claim, (left / (waves_per_region * (uint32_t)(tune.tail - 1))) & ~63u));
          if (lane == 0) off = atomicAdd(&work[r * kCounterStride], claim);
          off = __shfl(off, 0);
        }
        if (off < r_count) {
          w_next = r * q.cap + off;
          w_end = r * q.cap + min(off + claim, r_count);
        } else if (++region_try == kRegions) {
          dry = true;
        }
      }
      if (w_next < w_end) {
        uint32_t cand = w_next + (uint32_t)__popcll(idle & lt_mask);
        if (j == kNoRay && cand < w_end) {
          j = cand;
          // queue entries are read once: streaming loads leave the L2 to the nodes and triangles
          if (tune.shared_origin) ro = ld3(sc.camera.camera_center);
          else ro = D3(__builtin_nontemporal_load(&q.ox[j]), __builtin_nontemporal_load(&q.oy[j]), __builtin_nontemporal_load(&q.oz[j]));
          rd = D3(__builtin_nontemporal_load(&q.dx[j]), __builtin_nontemporal_load(&q.dy[j]), __builtin_nontemporal_load(&q.dz[j]));
          if (WIDE) fq = make_rayq<true>(sc, ro, rd);
          else if (GRID) fq = make_rayq(sc, ro, rd);
          else if (QUIRK) f = make_rayf(ro, rd);
          else fp = make_rayp(ro, rd);
          if (QUIRK && sc.nodes_d) rinv = D3(1.0 / rd.x, 1.0 / rd.y, 1.0 / rd.z);  // ray.rs:38-41
          trav_begin(root, tv);
          if (!(root & kRefLeaf) && ray_has_negative_zero(rd)) tv.cur = kDone;  // retires below as a miss
          if (tune.pass == 1 && hits_danger_box(sc, ro, rd)) {
            hits.ref[j] = kPending;
            j = kNoRay;
            tv.cur = kDone;
          } else if (tune.pass == 2 && hits.ref[j] != kPending) {
            j = kNoRay;
            tv.cur = kDone;
          }
        }
        w_next = min(w_end, w_next + (uint32_t)n_idle);
        n_idle = __popcll(__ballot(j == kNoRay));
      }
    }
    RT_PHASE_END(cyc_refill)
    if (n_idle == 64) {
      if (dry) break;
      continue;  // nothing in flight yet: claim again
    }
    // ------------------------------------------------------------ node steps
    // keep stepping while enough lanes have node work; a lane that reaches a leaf queues it and
    // moves on (or waits once its queue is full)
    bool pending;
    do {
      if (COUNT) wn++;
      if (COUNT) { d_idle += j == kNoRay; d_done += j != kNoRay && tv.cur == kDone; d_block += (tv.cur & kRefLeaf) && tv.cur < kWait; }
      if (!(tv.cur & kRefLeaf)) {
        if (COUNT) nn++;
        if (QUIRK && sc.nodes_d) trav_node_ref_exact(sc, ro, rd, rinv, tv, stack, stride);
        else if (QUIRK) trav_node_ref(sc, f, tv, stack, stride);
        else if (WIDE) trav_node_w<SM>(sc, fq, tv, sref, leafq, tune.leafq);
        else if (GRID) trav_node_q<SM>(sc, fq, tv, sref);
        else trav_node_fast<SM>(sc, fp, tv, sref);
      }
      if ((tv.cur & kRefLeaf) && tv.cur < kWait && tv.qn < tune.leafq) {
        if (QUIRK) trav_defer_leaf<true>(tv, stack, leafq, stride);
        else trav_defer_leaf_fast<SM, kLdsl>(tv, sref, leafq);
      }
#pragma unroll
      for (int extra = 0; extra < (QUIRK ? 0 : WIDE ? RT_EXTRA_STEPS_WIDE : RT_EXTRA_STEPS); extra++) {  // more steps before the wave votes again
        if (COUNT) wn++;
        if (!(tv.cur & kRefLeaf)) {
          if (COUNT) nn++;
          if (WIDE) trav_node_w<SM>(sc, fq, tv, sref, leafq, tune.leafq);
          else if (GRID) trav_node_q<SM>(sc, fq, tv, sref);
          else trav_node_fast<SM>(sc, fp, tv, sref);
        }
        if ((tv.cur & kRefLeaf) && tv.cur < kWait && tv.qn < tune.leafq) trav_defer_leaf_fast<SM, kLdsl>(tv, sref, leafq);
      }
    } while (__popcll(__ballot(!(tv.cur & kRefLeaf))) >= tune.node_min);
    RT_PHASE_END(cyc_node)
    // ------------------------------------------------------------ leaf phase
    pending = __ballot(tv.qn > 0) != 0;
    if (pending && !QUIRK && RT_SPLIT_LEAF) {
      uint32_t quads = 0, spheres = 0;  // queue slots that hold one
      for (int k = 0; k < kMaxLeafQueue; k++) {
        if (!__ballot(k < tv.qn)) break;
        if (COUNT) wl++;
        if (k < tv.qn) {
          if (COUNT) nl++;
          uint32_t leaf = leafq[k * stride];
          uint32_t kind = (leaf >> kRefKindShift) & 3u;
          if (kind == RT_PRIM_TRIANGLE) trav_test_triangle_fast(sc, ro, rd, tv, leaf);
          else if (kind == RT_PRIM_QUAD) quads |= 1u << k;
          else spheres |= 1u << k;
        }
      }
      while (__ballot(quads != 0)) {
        if (COUNT) wl++;
        if (quads) {
          int k = __ffs(quads) - 1;
          quads &= quads - 1;
          trav_test_quad_fast(sc, ro, rd, tv, leafq[k * stride]);
        }
      }
      while (__ballot(spheres != 0)) {
        if (COUNT) wl++;
        if (spheres) {
          int k = __ffs(spheres) - 1;
          spheres &= spheres - 1;
          trav_test_sphere_fast(sc, ro, rd, tv, leafq[k * stride]);
        }
      }
      tv.qn = 0;
    } else if (pending) {
      for (int k = 0; k < kMaxLeafQueue; k++) {
        if (!__ballot(k < tv.qn)) break;
        if (COUNT) wl++;
        if (k < tv.qn) {
          if (COUNT) nl++;
          int height = 64;
          if (QUIRK) height = k == 0 ? tv.m01.x : k == 1 ? tv.m01.y : k == 2 ? tv.m23.x : tv.m23.y;
          trav_test_leaf<QUIRK>(sc, ro, rd, tv, leafq[k * stride], height);
        }
      }
      tv.qn = 0;
      if (QUIRK && tv.cur == kWait) trav_pop<QUIRK>(tv, stack, stride);
    }
    // ---------------------------------------------------------------- retire
    if (tv.cur == kDone && tv.qn == 0 && j != kNoRay) {
      __builtin_nontemporal_store(tv.best.t, &hits.t[j]);
      __builtin_nontemporal_store(tv.best.ref, &hits.ref[j]);
      j = kNoRay;
    }
    RT_PHASE_END(cyc_leaf)
  }
#undef RT_PHASE_END
  if (COUNT) {
    atomicAdd(&visit_counters[0], (unsigned long long)nn);
    atomicAdd(&visit_counters[1], (unsigned long long)nl);
    atomicAdd(&visit_counters[8], (unsigned long long)d_idle);
    atomicAdd(&visit_counters[9], (unsigned long long)d_done);
    atomicAdd(&visit_counters[10], (unsigned long long)d_block);
    if (SM == 2) atomicAdd(&visit_counters[11], (unsigned long long)tv.moves);
    if (lane == 0) {
      atomicAdd(&visit_counters[3], (unsigned long long)wn);
      atomicAdd(&visit_counters[4], (unsigned long long)wl);
      atomicAdd(&visit_counters[5], cyc_refill);
      atomicAdd(&visit_counters[6], cyc_node);
      atomicAdd(&visit_counters[7], cyc_leaf);
    }
  }
}

// ------------------------------------------------------------- packet walk
// The closest-hit queries of a queue whose neighbouring entries are NEARLY THE SAME RAY: primary rays (64 consecutive
// entries are the samples of one or two pixels, each of them G times when the batches of a round repeat the ray).  The 64
// rays of a wave walk the tree TOGETHER: one node at a time for the whole wave, its record fetched once through the scalar
// unit, every lane testing its own ray against the two child boxes, the wave entering a child if any lane hits it (the child
// more lanes meet first goes first, the other waits on ONE stack, kept in LDS with the set of lanes that hit it).  A leaf is
// tested by the lanes that hit its box.  Per lane this is the order-free walk of k_extend on the 64-byte f32 records -- the
// same conservative box test, the same primitive tests on the same candidates or fewer, ties by leaf rank -- so results are
// those of k_extend; what changes is the price: no per-lane stack, no divergence between lanes that are at different nodes,
// one record fetch per node instead of sixty-four.  For rays that do not travel together it would be ruinous (the wave
// visits the union of what its lanes need): the host uses it for bounce 0 only.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
RT_DEV u32x16 scalar_load_node(const DevNode *nodes, uint32_t index) {  // `index` is the same in every lane
  const uint64_t address = reinterpret_cast<uint64_t>(nodes) + (uint64_t)__builtin_amdgcn_readfirstlane(index) * sizeof(DevNode);
  u32x16 v;
  asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(address) : "memory");
  return v;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// The 72 bytes of a triangle record that the test reads (v0, e1, e2), the same for every lane.
RT_DEV void scalar_load_triangle(const DevScene &sc, uint32_t index, d3 &v0, d3 &e1, d3 &e2) {
  const uint64_t address = reinterpret_cast<uint64_t>(sc.triangles) + (uint64_t)__builtin_amdgcn_readfirstlane(index) * sc.tri_stride;
  u32x16 a;
  u32x4 b;
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(address) : "memory");
  auto f64 = [](uint32_t lo, uint32_t hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); };
  v0 = D3(f64(a[0], a[1]), f64(a[2], a[3]), f64(a[4], a[5]));
  e1 = D3(f64(a[6], a[7]), f64(a[8], a[9]), f64(a[10], a[11]));
  e2 = D3(f64(a[12], a[13]), f64(a[14], a[15]), f64(b[0], b[1]));
}
constexpr int kPacketStack = kMaxStackDepth + 2;
template <bool COUNT, int WAVES>
__attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
__global__ void __launch_bounds__(64) k_extend_packet(DevScene sc, PathQueue q, const uint32_t *counts, HitBuffer hits, uint32_t *work, ExtendTuning tune,
                                                       unsigned long long *visit_counters) {
  __shared__ uint32_t st_ref[kPacketStack], st_lo[kPacketStack], st_hi[kPacketStack];
  const uint32_t lane = threadIdx.x;
  const uint32_t root = sc.root;
  const uint32_t home = blockIdx.x % kRegions;
  const uint32_t waves_per_region = gridDim.x / kRegions;
  uint32_t nn = 0, nl = 0, wn = 0, wl = 0;
  for (uint32_t region_try = 0; region_try < kRegions; region_try++) {  // own region first, then the others
    const uint32_t r = (home + region_try) % kRegions;
    const uint32_t r_count = counts[r * kCounterStride];
    for (;;) {
      // ---- claim (as k_extend does: the last claims of a region are smaller)
      uint32_t claim = (uint32_t)tune.claim, off = r_count;
      const uint32_t taken = *(volatile uint32_t *)&work[r * kCounterStride];
      if (taken < r_count) {
        const uint32_t left = r_count - taken;
        if (tune.tail >= 1 && left < waves_per_region * claim) claim = left < waves_per_region * (claim / 4) ? max(64u, claim / 8) : max(64u, claim / 4);
        if (lane == 0) off = atomicAdd(&work[r * kCounterStride], claim);
        off = __shfl(off, 0);
      }
      if (off >= r_count) break;
      const uint32_t w_end = r * q.cap + min(off + claim, r_count);
      for (uint32_t p0 = r * q.cap + off; p0 < w_end; p0 += 64) {
        // ---- one packet: the lane's ray
        const uint32_t j = p0 + lane;
        bool mine = j < w_end;
        d3 ro = D3(0, 0, 0), rd = D3(0, 0, 1);
        if (mine) {
          if (tune.shared_origin) ro = ld3(sc.camera.camera_center);
          else ro = D3(__builtin_nontemporal_load(&q.ox[j]), __builtin_nontemporal_load(&q.oy[j]), __builtin_nontemporal_load(&q.oz[j]));
          rd = D3(__builtin_nontemporal_load(&q.dx[j]), __builtin_nontemporal_load(&q.dy[j]), __builtin_nontemporal_load(&q.dz[j]));
        }
        const RayP f = make_rayp(ro, rd);
        Traversal tv;
        trav_begin(root, tv);
        bool walks = mine;
        if (mine && !(root & kRefLeaf) && ray_has_negative_zero(rd)) walks = false;  // a miss (see ray_has_negative_zero)
        if (mine && tune.pass == 1 && hits_danger_box(sc, ro, rd)) {  // left to the pass that follows the reference's rule
          hits.ref[j] = kPending;
          mine = walks = false;
        }
        uint32_t cur = root;
        unsigned long long cmask = __ballot(walks);
        int sp = 0;
        while (cmask != 0ull) {
          const bool in = (cmask >> lane) & 1ull;
          bool pop = false;
          if (!(cur & kRefLeaf)) {
            const u32x16 n = scalar_load_node(sc.nodes, cur & kRefIndexMask);
            float tl = 0.f, tr = 0.f;
            const bool hl = in && slab_hit_p(__uint_as_float(n[0]), __uint_as_float(n[1]), __uint_as_float(n[2]), __uint_as_float(n[3]),
                                             __uint_as_float(n[4]), __uint_as_float(n[5]), f, tv.tcull, tl);
            const bool hr = in && slab_hit_p(__uint_as_float(n[6]), __uint_as_float(n[7]), __uint_as_float(n[8]), __uint_as_float(n[9]),
                                             __uint_as_float(n[10]), __uint_as_float(n[11]), f, tv.tcull, tr);
            const unsigned long long ml = __ballot(hl), mr = __ballot(hr);
            const uint32_t left = n[12], right = n[13];
            if (COUNT) { nn += in; wn++; }
            if (ml != 0ull && mr != 0ull) {
              // the child that more lanes meet first is entered first; the other waits with the lanes that hit it
              const int want_left = __popcll(__ballot(hl && (!hr || tl <= tr))), want_right = __popcll(__ballot(hr && (!hl || tr < tl)));
              const bool left_first = want_left >= want_right;
              const unsigned long long later = left_first ? mr : ml;
              if (lane == 0) {
                st_ref[sp] = left_first ? right : left;
                st_lo[sp] = (uint32_t)later;
                st_hi[sp] = (uint32_t)(later >> 32);
              }
              sp++;
              cur = left_first ? left : right;
              cmask = left_first ? ml : mr;
            } else if ((ml | mr) != 0ull) {
              cur = ml != 0ull ? left : right;
              cmask = ml | mr;
            } else {
              pop = true;
            }
          } else {
            // a leaf, the same for every lane of the packet: the lanes that hit its box test it
            const uint32_t kind = (cur >> kRefKindShift) & 3u;
            if (COUNT) { nl += in; wl++; }
            if (kind == RT_PRIM_TRIANGLE) {
              d3 v0, e1, e2;
              scalar_load_triangle(sc, cur & kRefIndexMask, v0, e1, e2);
              if (in) {
                const Ray ray = {ro, rd, D3(0, 0, 0)};
                double t;
                const bool ok = triangle_hit(v0, e1, e2, ray, kTMin, tv.best.t, t);
                trav_accept_fast(sc, tv, cur, ok, t);
              }
            } else if (in) {
              if (kind == RT_PRIM_QUAD) trav_test_quad_fast(sc, ro, rd, tv, cur);
              else trav_test_sphere_fast(sc, ro, rd, tv, cur);
            }
            pop = true;
          }
          if (pop) {
            if (sp == 0) break;
            sp--;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)st_ref[sp]);
            cmask = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)st_lo[sp]) |
                    ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)st_hi[sp]) << 32);
          } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          }
        }
        if (mine) {
          __builtin_nontemporal_store(tv.best.t, &hits.t[j]);
          __builtin_nontemporal_store(tv.best.ref, &hits.ref[j]);
        }
      }
    }
  }
  if (COUNT) {
    atomicAdd(&visit_counters[0], (unsigned long long)nn);
    atomicAdd(&visit_counters[1], (unsigned long long)nl);
    if (lane == 0) {
      atomicAdd(&visit_counters[3], (unsigned long long)wn);
      atomicAdd(&visit_counters[4], (unsigned long long)wl);
    }
  }
}

// Tuning of a scene handle.  Read from the environment ONCE PER HANDLE, when rt_hip_scene_upload / rt_hip_scene_build make it
// (the variable names are in the table below), changed afterwards with rt_hip_scene_tune; no render and no query reads the
// environment, so what a handle does never depends on what ran before it in the process.
struct Tuning {
  ExtendTuning extend{40, 20, kMaxLeafQueue, (int)kClaim, 1, 0, 0};
  int wide_waves = 5;    // waves per SIMD of the walk on wide nodes (100 000 triangles: 690 ms at five, 719 at six)
  int ovf_waves = 6;     // waves per SIMD of the binary walk with the block-overflow stack
  int shade_blocks = 256 * 64;  // most workgroups of a k_shade launch
  long long max_paths = 0;      // paths in flight per round; 0: what the memory allows, at most 2^29
  int max_group = 64;    // most batches traced per round
  int first_group = 8;   // jump after a quiet first round
  int tile_hlog = -1;    // log2 of the tile height in shard rows; -1: by the row stride
  int spill_above = kSpillAboveDepth;  // tree height above which the binary walks keep only the newest kOvfLevels stack entries in LDS
  int verbose = 0;
  int packet_waves = 7;  // waves per SIMD of the packet walk: the walk waits for one record after another (headline image, bounce 0:
                         // 90.8 / 84.7 / 81.8 / 95.3 ms at 5 / 6 / 7 / 8 waves; eight spill inside the loop)
  int packet = -1;       // bounce 0 by the packet walk (k_extend_packet): 64 neighbouring primary rays walk the tree together;
                         // 2 (for measurements): the rays of rt_hip_trace_rays too, whatever they are; -1: decided by the size of
                         // the tree when the handle is made (resolve_packet_default)
  // layout of the device copy, used while the handle is made only
  int wide = -1, tri_pad = -1;  // -1: by size; 0 / 1: forced off / on
  int node_order = -1;          // -1: by size; 0: the reference's preorder; 1: line-sized blocks
  int no_grid = 0, no_exact_boxes = 0;
  // rt_hip_scene_build: -1 = the reference's top-down build by the rule RtScene.split_method names; 0 = the radix tree over Morton codes,
  // 1 = locally-ordered clustering (neighbours within ploc_radius places of the Morton order), 2 / 3 = the reference's top-down build
  // with its bucketed surface-area heuristic / its midpoint split whatever the scene says
  int build_method = -1, ploc_radius = 16;
};
struct TuningName {
  const char *env, *name;
  int Tuning::*field;
  int lo, hi;
  bool layout;
};
static const TuningName kTuningNames[] = {
    {"RT_HIP_WIDE_WAVES", "wide_waves", &Tuning::wide_waves, 4, 6, false},
    {"RT_HIP_OVF_WAVES", "ovf_waves", &Tuning::ovf_waves, 5, 6, false},
    {"RT_HIP_SHADE_BLOCKS", "shade_blocks", &Tuning::shade_blocks, kRegions, 1 << 20, false},
    {"RT_HIP_MAX_GROUP", "max_group", &Tuning::max_group, 1, 1 << 20, false},
    {"RT_HIP_FIRST_GROUP", "first_group", &Tuning::first_group, 1, 1 << 20, false},
    {"RT_HIP_TILE_HLOG", "tile_hlog", &Tuning::tile_hlog, 0, 2 * RT_TILE_LOG, false},
    {"RT_HIP_VERBOSE", "verbose", &Tuning::verbose, 0, 1, false},
    {"RT_HIP_PACKET", "packet", &Tuning::packet, 0, 2, false},  // (unset: by the size of the tree)
    {"RT_HIP_PACKET_WAVES", "packet_waves", &Tuning::packet_waves, 5, 8, false},
    {"RT_HIP_SPILL_ABOVE", "spill_above", &Tuning::spill_above, kOvfLevels, kMaxStackDepth, true},
    {"RT_HIP_WIDE", "wide", &Tuning::wide, 0, 1, true},
    {"RT_HIP_TRI_PAD", "tri_pad", &Tuning::tri_pad, 0, 1, true},
    {"RT_HIP_NO_GRID", "no_grid", &Tuning::no_grid, 0, 1, true},
    {"RT_HIP_NO_EXACT_BOXES", "no_exact_boxes", &Tuning::no_exact_boxes, 0, 1, true},
    {"RT_HIP_BUILD", "build_method", &Tuning::build_method, -1, 3, true},
    {"RT_HIP_PLOC_RADIUS", "ploc_radius", &Tuning::ploc_radius, 1, 32, true},
};
struct ExtendTuningName {
  const char *env, *name;
  int ExtendTuning::*field;
  int lo, hi;
};
static const ExtendTuningName kExtendTuningNames[] = {
    {"RT_HIP_TAIL", "tail", &ExtendTuning::tail, 0, 1 << 16},
    {"RT_HIP_REFILL", "refill", &ExtendTuning::refill, 1, 64},
    {"RT_HIP_NODE_MIN", "node_min", &ExtendTuning::node_min, 1, 65},
    {"RT_HIP_LEAFQ", "leafq", &ExtendTuning::leafq, 1, kMaxLeafQueue},
    {"RT_HIP_CLAIM", "claim", &ExtendTuning::claim, 64, 1 << 16},
};
static int clamp_int(long long v, int lo, int hi) { return (int)std::max<long long>(lo, std::min<long long>(hi, v)); }
static Tuning tuning_from_env() {
  Tuning t;
  for (const TuningName &n : kTuningNames)
    if (const char *e = getenv(n.env)) t.*n.field = clamp_int(atoll(e), n.lo, n.hi);
  for (const ExtendTuningName &n : kExtendTuningNames)
    if (const char *e = getenv(n.env)) t.extend.*n.field = clamp_int(atoll(e), n.lo, n.hi);
  if (const char *e = getenv("RT_HIP_MAX_PATHS")) t.max_paths = std::max(0ll, atoll(e));
  if (const char *e = getenv("RT_HIP_NODE_ORDER")) t.node_order = !strcmp(e, "blocks") ? 1 : 0;
  // switches that count as set when the variable merely exists
  if (getenv("RT_HIP_VERBOSE")) t.verbose = 1;
  if (getenv("RT_HIP_NO_GRID")) t.no_grid = 1;
  if (getenv("RT_HIP_NO_EXACT_BOXES")) t.no_exact_boxes = 1;
  t.shade_blocks = std::max((int)kRegions, t.shade_blocks / (int)kRegions * (int)kRegions);
  return t;
}

// The walk that follows the reference's rule keeps its whole stack in LDS.
static bool extend_wide(const DevScene &sc, bool quirk) { return !quirk && sc.nodes_w != nullptr; }
// The binary order-free walks move blocks of stack entries out of LDS (rt_device.h, StackRef) exactly when upload gave the scene a
// buffer for them: trees higher than Tuning::spill_above.
static bool extend_spills(const DevScene &sc, bool quirk) {
  if (extend_wide(sc, quirk)) return sc.stack_depth_w > kWideLdsLevels;
  return !quirk && sc.stack_spill != nullptr;
}
static size_t extend_lds_bytes(const DevScene &sc, bool quirk) {
  const bool wide = extend_wide(sc, quirk);
  const int levels = extend_spills(sc, quirk) ? (wide && RT_WIDE_STACK_MODE == 1 ? kWideLdsLevels : kOvfLevels + 1) : (wide ? sc.stack_depth_w : sc.stack_depth);
  return (size_t)(levels + kMaxLeafQueue) * 64 * sizeof(uint32_t);
}
constexpr int kMaxExtendWaves = 256 * 32;  // 256 CUs x 32 wave slots
constexpr size_t kLdsPerCu = 160 * 1024;
// Waves per SIMD the kernel's registers are budgeted for (template argument WAVES): the walk is bound by VALU issue with
// part of its memory latency exposed, and a sixth wave (80 instead of 88 registers, a few loop-invariant values in
// scratch) hides more of it than the tighter budget costs (C3: 688 -> 654 ms).  Seven waves (72 registers) spill inside the loops
// (1076 ms).  Six waves need 24 stacks per CU in LDS: trees up to height 22 keep the whole stack there, higher ones take the
// block-overflow stack (17 rows of LDS whatever the height); the counting and reference-rule variants take what they need.

// `counts` = entries per region, `work` = claim counters per region (zeroed by the caller); both
// kCounterStride words apart.
// Which walk a query takes: 0 the order-free one, 1 the reference's visit rule for every ray,
// 2 both -- the order-free walk for the rays that stay clear of the out-of-box primitives and
// the reference's rule for the others.
static void launch_extend_pass(const DevScene &sc, const Tuning &tuning, bool quirk, bool counted, int pass, bool shared_origin, int64_t n_upper,
                               hipStream_t st, PathQueue q, const uint32_t *count_ptr, HitBuffer hits, uint32_t *work,
                               unsigned long long *visit_counters) {
  ExtendTuning tune = tuning.extend;
  tune.pass = pass;
  tune.shared_origin = shared_origin;
  // persistent waves: enough to fill 256 CUs x 32 wave slots, never more than one per claim,
  // a whole number of them per region
  int64_t claims = (n_upper + tune.claim - 1) / tune.claim;
  int grid = (int)std::max<int64_t>(kRegions, std::min<int64_t>((claims + kRegions - 1) / kRegions * kRegions, kMaxExtendWaves));
  dim3 g(grid), b(64);
  size_t lds = extend_lds_bytes(sc, quirk);
#define RT_LAUNCH_EXTEND(Q, C, G, S, W) hipLaunchKernelGGL((k_extend<Q, C, G, S, W>), g, b, lds, st, sc, q, count_ptr, hits, work, tune, visit_counters)
  const bool spill = extend_spills(sc, quirk);
  if (quirk) {
    if (counted) RT_LAUNCH_EXTEND(true, true, false, false, 4);
    else RT_LAUNCH_EXTEND(true, false, false, false, 5);  // 96 registers, none in scratch (README scene, every ray by this rule: 1933 ms at four waves, 1697 at five, 1718 at six with 60 bytes of scratch)
  } else if (sc.nodes_w) {
    const int wide_waves = tuning.wide_waves;
#define RT_LAUNCH_WIDE(C, S, W) hipLaunchKernelGGL((k_extend<false, C, true, S, W, true>), g, b, lds, st, sc, q, count_ptr, hits, work, tune, visit_counters)
    // (eight-wide, 128-byte records were measured too: 10 M-triangle soup 40.8 instead of 57.1 records per ray, but k_extend
    // 126 ms at five waves per SIMD, 140 at four, against 101 on the four-wide ones)
    if (counted) { if (spill) RT_LAUNCH_WIDE(true, true, 4); else RT_LAUNCH_WIDE(true, false, 4); }
    else if (wide_waves >= 6 && lds * 24 <= kLdsPerCu) { if (spill) RT_LAUNCH_WIDE(false, true, 6); else RT_LAUNCH_WIDE(false, false, 6); }
    else if (wide_waves == 5) { if (spill) RT_LAUNCH_WIDE(false, true, 5); else RT_LAUNCH_WIDE(false, false, 5); }
    else { if (spill) RT_LAUNCH_WIDE(false, true, 4); else RT_LAUNCH_WIDE(false, false, 4); }
#undef RT_LAUNCH_WIDE
  } else if (sc.nodes_q) {
    if (counted) { if (spill) RT_LAUNCH_EXTEND(false, true, true, true, 4); else RT_LAUNCH_EXTEND(false, true, true, false, 4); }
    else if (spill) {
      if (tuning.ovf_waves >= 6) RT_LAUNCH_EXTEND(false, false, true, true, 6);
      else RT_LAUNCH_EXTEND(false, false, true, true, 5);
    } else if (lds * 24 <= kLdsPerCu) RT_LAUNCH_EXTEND(false, false, true, false, 6);  // the production kernel on trees up to height 22
    else RT_LAUNCH_EXTEND(false, false, true, false, 5);
  } else {
    if (counted) { if (spill) RT_LAUNCH_EXTEND(false, true, false, true, 4); else RT_LAUNCH_EXTEND(false, true, false, false, 4); }
    else if (lds * 24 <= kLdsPerCu) {  // the walk on the 64-byte f32 records too (README scene forced onto them: 727 ms at five waves, 708 at six)
      if (spill) RT_LAUNCH_EXTEND(false, false, false, true, 6); else RT_LAUNCH_EXTEND(false, false, false, false, 6);
    } else { if (spill) RT_LAUNCH_EXTEND(false, false, false, true, 5); else RT_LAUNCH_EXTEND(false, false, false, false, 5); }
  }
#undef RT_LAUNCH_EXTEND
}

// `counts` = entries per region; `work`, `work2` = claim counters per region of the (up to) two
// passes, zeroed by the caller; all kCounterStride words apart.
// `shared_origin`: the queue holds the primary rays k_generate wrote for a camera without a lens (no origins stored).
// `coherent`: neighbouring entries of the queue are nearly the same ray (primary rays): the order-free walk is then done by
// k_extend_packet, 64 rays at a time together (same results; the reference-rule walk has no such form).
static void launch_extend_packet(const DevScene &sc, const Tuning &tuning, bool counted, int pass, bool shared_origin, int64_t n_upper, hipStream_t st,
                                 PathQueue q, const uint32_t *count_ptr, HitBuffer hits, uint32_t *work, unsigned long long *visit_counters) {
  ExtendTuning tune = tuning.extend;
  tune.pass = pass;
  tune.shared_origin = shared_origin;
  int64_t claims = (n_upper + tune.claim - 1) / tune.claim;
  int grid = (int)std::max<int64_t>(kRegions, std::min<int64_t>((claims + kRegions - 1) / kRegions * kRegions, kMaxExtendWaves));
#define RT_LAUNCH_PACKET(C, W) hipLaunchKernelGGL((k_extend_packet<C, W>), dim3(grid), dim3(64), 0, st, sc, q, count_ptr, hits, work, tune, visit_counters)
  if (counted) RT_LAUNCH_PACKET(true, 5);
  else if (tuning.packet_waves >= 8) RT_LAUNCH_PACKET(false, 8);
  else if (tuning.packet_waves == 7) RT_LAUNCH_PACKET(false, 7);
  else if (tuning.packet_waves == 6) RT_LAUNCH_PACKET(false, 6);
  else RT_LAUNCH_PACKET(false, 5);
#undef RT_LAUNCH_PACKET
}
static void launch_extend(const DevScene &sc, const Tuning &tuning, int mode, bool counted, bool shared_origin, int64_t n_upper, hipStream_t st,
                          PathQueue q, const uint32_t *count_ptr, HitBuffer hits, uint32_t *work, uint32_t *work2,
                          unsigned long long *visit_counters, bool coherent = false) {
  const bool packet = coherent && tuning.packet && mode != 1;
  if (mode == 2) {
    if (packet) launch_extend_packet(sc, tuning, counted, 1, shared_origin, n_upper, st, q, count_ptr, hits, work, visit_counters);
    else launch_extend_pass(sc, tuning, false, counted, 1, shared_origin, n_upper, st, q, count_ptr, hits, work, visit_counters);
    launch_extend_pass(sc, tuning, true, counted, 2, shared_origin, n_upper, st, q, count_ptr, hits, work2, visit_counters);
  } else if (packet) {
    launch_extend_packet(sc, tuning, counted, 0, shared_origin, n_upper, st, q, count_ptr, hits, work, visit_counters);
  } else {
    launch_extend_pass(sc, tuning, mode == 1, counted, 0, shared_origin, n_upper, st, q, count_ptr, hits, work, visit_counters);
  }
}

// ===================================================================== shade
// HittablePDF::value over the attractor list (pdf.rs:98-107) with
// Quad::pdf_value (quad.rs:155-167) / Sphere::pdf_value (sphere.rs:154-169).
// `renormalized` = normalized(direction): what Ray::new makes of it again inside every pdf_value
// (quad.rs:157, sphere.rs:157); the caller has it already for the cosine density.
__device__ double attractor_pdf_value(const DevScene &sc, d3 origin, d3 direction, d3 renormalized) {
  double weight = 1.0 / (double)sc.n_attractors;
  double sum = 0.0;
  Ray probe = {origin, renormalized, D3(0, 0, 0)};  // the tests below never read inv
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

RT_DEV double cosine_pdf_value(const Onb &uvw, d3 renormalized) {  // pdf.rs:159-162, argument = direction.normalized()
  double cosine_theta = dot(renormalized, uvw.w);
  return fmax(0.0, cosine_theta / kPi);
}

// Per-path result, written once when the path ends: one aligned 32-byte record
// (a single sector for the scattered store) holding the radiance and the
// number of BVH queries the path made.
struct alignas(32) PathRecord {
  double r, g, b;
  uint64_t n_rays;
};
struct PathResults {
  PathRecord *rec;
};

// One bounce of every path in a queue: the material's response at the hit, the next ray, stream compaction into the next queue.
// The continuing paths are SORTED BY WHAT THEY DO NEXT before the sampling arithmetic.  Left as the queue has them, a wave pays for
// every material's sampling code that any of its lanes needs while the lanes of paths that just ended idle (measured that way: 38 %
// of the lanes enabled in the kernel's VALU instructions).  So a workgroup of four waves classifies 256 paths (phase A), counts the
// continuing ones by category -- Lambertian by cosine, Lambertian towards an attractor, Metal, Dielectric -- hands each one's state
// through LDS to the lane at its place in that order, and samples (phase B) with the continuing paths packed into as few waves as
// they fill, one category per wave except where two meet.  Per-path results do not depend on which lane computes them, and the next
// queue comes out grouped the same way, which the walk likes: rays aimed at an attractor from neighbouring pixels travel together.
//  * the hit reference of the NEXT trip is fetched one trip ahead: the shading record it names is then requested together with the
//    path state instead of one memory latency after it;
//  * a small material table is staged in LDS: the lookup sits on the dependent chain hit reference -> shading record -> material ->
//    "does the path go on?";
//  * a path that ends stores its result at once; wave64 compaction of the survivors: ballot, prefix popcount, one atomic per wave.
constexpr int kShadeCategories = 4;
#ifndef RT_SHADE_BLOCK
#define RT_SHADE_BLOCK 256
#endif
constexpr int kShadeBlock = RT_SHADE_BLOCK, kShadeWaves = kShadeBlock / 64;  // paths a workgroup sorts together
// REUSE (bounce 0 of a round whose primary rays were traced once, PrimaryReuse above): a path's ray and hit are those of the
// compact queue's entry pr.entry[j]; `hits` is that queue's hit buffer.
template <bool REUSE>
__attribute__((amdgpu_waves_per_eu(4, 4)))
__global__ void __launch_bounds__(kShadeBlock) k_shade(DevScene sc, PathQueue q, const uint32_t *count_ptr, HitBuffer hits, PathQueue out,
                                                       uint32_t *out_count, PathResults res, int32_t bounce, int32_t last_bounce, int32_t flags,
                                                       PrimaryReuse pr) {
  constexpr int kLdsMaterials = 64;
  __shared__ DevMaterial lds_materials[kLdsMaterials];
  __shared__ double lds_f64[13][kShadeBlock];  // point, normal, throughput, direction (3 each), material parameter
  __shared__ uint64_t lds_rng[kShadeBlock];
  __shared__ uint32_t lds_pid[kShadeBlock], lds_bits[kShadeBlock];
  // continuing paths per category and wave, one set per parity of the trip: a wave writes the next trip's counts before that trip's
  // first barrier, while a slower wave may still be reading this trip's after its second one
  __shared__ uint32_t lds_counts_all[2][kShadeCategories][kShadeWaves];
  uint32_t trip_parity = 0;
  const bool staged = sc.n_materials <= kLdsMaterials;
  if (staged) {
    const uint32_t words = (uint32_t)sc.n_materials * (uint32_t)(sizeof(DevMaterial) / 4);
    for (uint32_t w = threadIdx.x; w < words; w += blockDim.x)
      reinterpret_cast<uint32_t *>(lds_materials)[w] = reinterpret_cast<const uint32_t *>(sc.materials)[w];
    __syncthreads();
  }
  const uint32_t region = blockIdx.x % kRegions;
  const uint32_t n = count_ptr[region * kCounterStride];
  const uint32_t stride = (gridDim.x / kRegions) * (uint32_t)kShadeBlock;
  const uint32_t n_round = (n + (uint32_t)kShadeBlock - 1u) / (uint32_t)kShadeBlock * (uint32_t)kShadeBlock;  // whole workgroups: every thread reaches every barrier
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const uint32_t in0 = region * q.cap, out0 = region * out.cap;
  uint32_t *out_counter = out_count + region * kCounterStride;
  const uint32_t jj0 = (blockIdx.x / kRegions) * (uint32_t)kShadeBlock + threadIdx.x;
  uint32_t ref_ahead = (!REUSE && jj0 < n) ? hits.ref[in0 + jj0] : kRefNone;
  for (uint32_t jj = jj0; jj - threadIdx.x < n_round; jj += stride) {
    const uint32_t j = in0 + jj;
    uint32_t (*lds_counts)[kShadeWaves] = lds_counts_all[trip_parity];
    trip_parity ^= 1u;
    const uint32_t ref_now = ref_ahead;
    if (!REUSE) ref_ahead = (jj + stride < n) ? hits.ref[j + stride] : kRefNone;
    // ---- phase A: load, classify, decide whether the path goes on
    int cat = -1;  // 0 Lambertian by cosine, 1 Lambertian towards an attractor, 2 Metal, 3 Dielectric
    {
      int32_t mkind = -1;
      bool front_face = false;
      d3 normal = D3(0, 0, 1), albedo = D3(0, 0, 0), point = D3(0, 0, 0), d = D3(0, 0, 1), T = D3(0, 0, 0), rad = D3(0.0, 0.0, 0.0);
      double param = 0.0;
      uint64_t rng = 0;
      uint32_t pid = 0;
      if (jj < n) {
        const uint32_t hj = REUSE ? pr.entry[j] : j;  // where the path's ray and hit are
        uint32_t ref = REUSE ? hits.ref[hj] : ref_now;
        double t = hits.t[hj];
        pid = q.pid[j];
        rng = q.rng[j];
        T = bounce ? D3(q.tx[j], q.ty[j], q.tz[j]) : D3(1.0, 1.0, 1.0);
        d3 o = (REUSE || (bounce == 0 && sc.camera.defocus_angle <= 0.0)) ? ld3(sc.camera.camera_center) : D3(q.ox[j], q.oy[j], q.oz[j]);
        d = REUSE ? D3(pr.dx[hj], pr.dy[hj], pr.dz[hj]) : D3(q.dx[j], q.dy[j], q.dz[j]);
        bool dead_weight = (flags & kFlagEarlyTerminate) && T.x == 0.0 && T.y == 0.0 && T.z == 0.0;
        if (ref != kRefNone && !dead_weight) {
          point = o + d * t;  // Ray::at
          uint32_t kind = (ref >> kRefKindShift) & 3u, idx = ref & kRefIndexMask;
          uint32_t mat;
          if (kind == RT_PRIM_TRIANGLE) {
            const double2 *p = reinterpret_cast<const double2 *>(&sc.tri_shade[idx]);
            double2 a = p[0], b = p[1];
            normal = D3(a.x, a.y, b.x);
            mat = (uint32_t)(__double_as_longlong(b.y) & 0xffffffffll);
            front_face = dot(d, normal) < 0.0;
          } else if (kind == RT_PRIM_QUAD) {
            normal = ld3(sc.quads[idx].normal);
            mat = sc.quads[idx].material;
            front_face = dot(d, normal) < 0.0;
          } else {  // sphere.rs:127-136
            const DevSphere &sp = sc.spheres[idx];
            d3 outward = normalized((point - ld3(sp.center)) / sp.radius);
            front_face = dot(d, outward) < 0.0;
            normal = front_face ? outward : neg(outward);
            mat = sp.material;
          }
          if (staged) {
            const DevMaterial &m = lds_materials[mat];
            albedo = ld3(m.albedo);
            param = m.param;
            mkind = m.kind;
          } else {
            const DevMaterial &m = sc.materials[mat];
            albedo = ld3(m.albedo);
            param = m.param;
            mkind = m.kind;
          }
          if (mkind != RT_MAT_DIELECTRIC) T = T * albedo;
          if (mkind == RT_MAT_DIFFUSE_LIGHT) {
            if (front_face) rad = T;  // diffuse_light.rs:20-29
          } else if (last_bounce) {
            // the next ray_color call would run at depth 0 and return black
          } else if (mkind == RT_MAT_METAL) {
            cat = 2;
          } else if (mkind == RT_MAT_DIELECTRIC) {
            cat = 3;
          } else {
            cat = 0;
            // integrator.rs:381-391, MixedPDF ratio 0.5: the draw that picks the density is the path's next one
            if (sc.n_attractors > 0 && rng_gen_range(rng, 0.0, 1.0) < 0.5) cat = 1;
          }
        }
        if (cat < 0) {  // the path ends here
          double4 rec = {rad.x, rad.y, rad.z, __longlong_as_double((long long)(bounce + 1))};
          double2 *dst = reinterpret_cast<double2 *>(res.rec + pid);
          dst[0] = make_double2(rec.x, rec.y);
          dst[1] = make_double2(rec.z, rec.w);
        }
      }
      // ---- the continuing paths take their places: category by category, wave by wave, lane by lane
      unsigned long long mine = 0;
#pragma unroll
      for (int c = 0; c < kShadeCategories; c++) {
        unsigned long long m = __ballot(cat == c);
        if (cat == c) mine = m;
        if (lane == 0) lds_counts[c][wave] = (uint32_t)__popcll(m);
      }
      __syncthreads();
      uint32_t pos = (uint32_t)__popcll(mine & lt_mask);
#pragma unroll
      for (int c = 0; c < kShadeCategories; c++)
#pragma unroll
        for (uint32_t w = 0; w < (uint32_t)kShadeWaves; w++) {
          uint32_t cnt = lds_counts[c][w];
          if (c < cat || (c == cat && w < wave)) pos += cnt;
        }
      if (cat >= 0) {
        lds_f64[0][pos] = point.x; lds_f64[1][pos] = point.y; lds_f64[2][pos] = point.z;
        lds_f64[3][pos] = normal.x; lds_f64[4][pos] = normal.y; lds_f64[5][pos] = normal.z;
        lds_f64[6][pos] = T.x; lds_f64[7][pos] = T.y; lds_f64[8][pos] = T.z;
        lds_f64[9][pos] = d.x; lds_f64[10][pos] = d.y; lds_f64[11][pos] = d.z;
        lds_f64[12][pos] = param;
        lds_rng[pos] = rng;
        lds_pid[pos] = pid;
        lds_bits[pos] = (uint32_t)cat | (front_face ? 4u : 0u);
      }
    }
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (int c = 0; c < kShadeCategories; c++)
#pragma unroll
      for (uint32_t w = 0; w < (uint32_t)kShadeWaves; w++) total += lds_counts[c][w];
    // ---- phase B: thread i samples the i-th continuing path of the workgroup
    const uint32_t i = threadIdx.x;
    bool alive = i < total;
    d3 point = D3(0, 0, 0), nd = D3(0, 0, 0), T = D3(0, 0, 0);
    uint64_t rng = 0;
    uint32_t pid = 0;
    if (alive) {
      point = D3(lds_f64[0][i], lds_f64[1][i], lds_f64[2][i]);
      d3 normal = D3(lds_f64[3][i], lds_f64[4][i], lds_f64[5][i]);
      T = D3(lds_f64[6][i], lds_f64[7][i], lds_f64[8][i]);
      d3 d = D3(lds_f64[9][i], lds_f64[10][i], lds_f64[11][i]);
      double param = lds_f64[12][i];
      rng = lds_rng[i];
      pid = lds_pid[i];
      const uint32_t bits = lds_bits[i];
      const int c = (int)(bits & 3u);
      const bool front_face = (bits & 4u) != 0;
      if (c <= 1) {  // Lambertian
        // CosinePDF::new builds the whole basis (onb.rs:31-45); a path aimed at an attractor only ever reads its w axis
        Onb uvw;
        uvw.w = normalized(normal);
        uvw.u = uvw.v = D3(0, 0, 0);
        if (c == 0) {
          d3 a = (fabs(uvw.w.x) > 0.9) ? D3(0., 1., 0.) : D3(1.0, 0., 0.);
          uvw.v = normalized(cross(uvw.w, a));
          uvw.u = cross(uvw.w, uvw.v);
        }
        d3 dir;
        double pdf_val, spdf;
        if (sc.n_attractors > 0) {
          if (c == 1) dir = attractor_generate(sc, point, rng);
          else dir = onb_local_vec(uvw, random_cosine_direction(rng));
          nd = normalized(dir);
          d3 nn = normalized(nd);  // both densities normalise the (already unit) direction once more
          spdf = cosine_pdf_value(uvw, nn);
          pdf_val = 0.5 * attractor_pdf_value(sc, point, nd, nn) + (1. - 0.5) * spdf;
        } else {
          dir = onb_local_vec(uvw, random_cosine_direction(rng));
          nd = normalized(dir);
          spdf = cosine_pdf_value(uvw, normalized(nd));
          pdf_val = spdf;
        }
        // (attenuation * target * scattering_pdf) / pdf_val, NaN -> black (integrator.rs:413-419)
        d3 Tn = (T * spdf) / pdf_val;  // T already holds throughput * albedo
        if (has_nan(Tn)) Tn = D3(0., 0., 0.);
        T = Tn;
      } else if (c == 2) {  // metal.rs:21-53
        d3 v = normalized(d);
        d3 reflected = v - normal * (2.0 * dot(v, normal));
        d3 dir = reflected + random_unit_vector(rng) * param;
        if (dot(normal, dir) > 0.0) {
          nd = normalized(dir);
        } else {  // absorbed: the path ends black
          alive = false;
          double4 rec = {0.0, 0.0, 0.0, __longlong_as_double((long long)(bounce + 1))};
          double2 *dst = reinterpret_cast<double2 *>(res.rec + pid);
          dst[0] = make_double2(rec.x, rec.y);
          dst[1] = make_double2(rec.z, rec.w);
        }
      } else {  // dielectric.rs:24-66
        double ratio = front_face ? 1.0 / param : param;
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
        nd = normalized(dir);  // attenuation is white
      }
    }
    // ---- wave64 stream compaction and stores
    unsigned long long mask = __ballot(alive);
    if (mask) {
      uint32_t base = 0;
      int leader = __ffsll((long long)mask) - 1;
      if ((int)lane == leader) base = atomicAdd(out_counter, (uint32_t)__popcll(mask));
      base = __shfl(base, leader);
      if (alive) {
        uint32_t k = out0 + base + (uint32_t)__popcll(mask & lt_mask);
        out.ox[k] = point.x; out.oy[k] = point.y; out.oz[k] = point.z;
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
  uint8_t *rgb8, *sample_rgb8;
};
// Color::to_rgb, color.rs:47-54: clamp to [0,1], then a truncating cast of c * 255
__device__ __forceinline__ uint8_t to_rgb8(double c) { return (uint8_t)(fmax(fmin(c, 1.0), 0.0) * 255.0); }
constexpr int kAccChunk = 4;  // batches k_accumulate evaluates together

// One lane per active pixel.  Processes the G batches of the round in order,
// exactly as the `'batch` loop of integrator.rs:279-328 would: ordered sum of
// the batch, Stats::push, z-test, stop.  Batches after the stopping one were
// traced speculatively and are ignored.  totals[0] accumulates the BVH queries
// of the accepted samples.
__global__ void __launch_bounds__(256) k_accumulate(RenderParams rp, const uint32_t *active, int32_t n_slots, int32_t batch0, int32_t G,
                                                     PathResults res, const double *weights, double *sums, double *hist,
                                                     uint32_t *next_active, uint32_t *next_count, PixelOutputs out, const uint32_t *qcounts,
                                                     const uint32_t *qcounts_primary, unsigned long long *totals) {
  if (blockIdx.x == 0 && threadIdx.x < 64) {  // queue sizes = queries traced per bounce (first wave)
    for (int k = 0; k < rp.max_depth; k++) {
      unsigned long long c = 0;
      // (the primary rays of a round that traces them once per pixel and sample have a queue of their own)
      const uint32_t *qc = (k == 0 && qcounts_primary) ? qcounts_primary : qcounts + (size_t)k * kRegions * kCounterStride;
      for (int r = threadIdx.x; r < kRegions; r += 64) c += qc[(size_t)r * kCounterStride];
      for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
      if (threadIdx.x == 0) totals[1 + (k < RT_MAX_DEPTH_STATS ? k : RT_MAX_DEPTH_STATS - 1)] += c;
    }
  }
  int32_t n_round = (n_slots + 63) & ~63;
  const size_t np = (size_t)rp.n_keys;  // stride of the per-pixel working arrays, which are indexed by key
  const uint32_t lane = threadIdx.x & 63u;
  // per-wave staging tile: 64 slots x 128 bytes, rows padded to 144 bytes against bank conflicts
  __shared__ double2 tile_all[4][64 * 9];
  double2 *tile = tile_all[threadIdx.x >> 6];
  for (int32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n_round; slot += gridDim.x * blockDim.x) {
    bool again = false;
    uint32_t p = 0, key = 0;
    unsigned long long my_rays = 0, my_samples = 0;
    const bool valid = slot < n_slots;
    const int32_t slot0 = slot - (int32_t)lane;  // first slot of this wave
    double sr = 0.0, sg = 0.0, sb = 0.0, sw = 0.0;
    double *ps = sums;
    if (valid) {
      key = active[slot];
      p = pixel_of_key(rp, key);
      ps = sums + key;
      if (batch0) {
        sr = ps[0]; sg = ps[np]; sb = ps[2 * np]; sw = ps[3 * np];
      }
    }
    bool done = !valid;
    double mr = 0.0, mg = 0.0, mb = 0.0;
    int nb = batch0;
    const double *ww = weights + key;
    // The batches of the round are handled kAccChunk at a time, in three passes that together
    // perform, per pixel, exactly the operations of the reference's batch loop in its order:
    //   1. per batch: the ordered sum of its samples, Stats::push, the running mean;
    //   2. Stats::variance for all batches of the chunk at once -- every history entry is
    //      loaded once and folded into each batch's own nominator (each fold still runs over
    //      k = 0..nb-1 in order with that batch's mean);
    //   3. the z-test decisions in batch order; batches after a pixel's stopping batch were
    //      speculative and leave no trace.
    for (int g0 = 0; g0 < G; g0 += kAccChunk) {
      const unsigned long long busy = __ballot(!done);
      if (!busy) break;
      const int gc = G - g0 < kAccChunk ? G - g0 : kAccChunk;
      const int kb = batch0 + g0;  // first batch of the chunk = number of history entries before it
      double mrv[kAccChunk], mgv[kAccChunk], mbv[kAccChunk], swv[kAccChunk], nom[kAccChunk];
      uint32_t rays_v[kAccChunk];
      // ---- pass 1
#pragma unroll
      for (int gi = 0; gi < kAccChunk; gi++) {
        mrv[gi] = mgv[gi] = mbv[gi] = swv[gi] = nom[gi] = 0.0;
        rays_v[gi] = 0;
        if (gi < gc) {
          const int g = g0 + gi;
          // integrator.rs:280-317: batch_color += ray_color * w, in sample order
          double br = 0.0, bg = 0.0, bb = 0.0, bw = 0.0;
          uint32_t nr = 0;
          for (int i0 = 0; i0 < rp.S; i0 += 4) {
            const int nrec = rp.S - i0 < 4 ? rp.S - i0 : 4;
            // The 4 records of one slot are 128 contiguous bytes.  Load instruction k fetches those
            // of slots 8k..8k+7, 16 bytes per lane (whole lines, each fetched once), and the tile in
            // LDS hands every lane the records of its own slot.
            double2 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
              const int sl = 8 * k + (int)(lane >> 3), piece = (int)(lane & 7u);
              v[k] = make_double2(0.0, 0.0);
              if (((busy >> sl) & 1ull) && piece < 2 * nrec)
                v[k] = reinterpret_cast<const double2 *>(res.rec + ((size_t)(slot0 + sl) * G + g) * rp.S + i0)[piece];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) tile[(8 * k + (lane >> 3)) * 9 + (lane & 7u)] = v[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double2 m[8];
#pragma unroll
            for (int k = 0; k < 8; k++) m[k] = tile[lane * 9 + k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!done) {
#pragma unroll
              for (int k = 0; k < 4; k++) {
                if (k < nrec) {
                  double w = ww[(size_t)(i0 + k) * np];
                  br = br + m[2 * k].x * w;
                  bg = bg + m[2 * k].y * w;
                  bb = bb + m[2 * k + 1].x * w;
                  bw += w;
                  nr += (uint32_t)__double_as_longlong(m[2 * k + 1].y);
                }
              }
            }
          }
          if (!done) {
            // stats.push(batch_color / batch_sum_sample_weight, batch_sum_sample_weight)
            double cr = br / bw, cg = bg / bw, cb = bb / bw;
            // Stats::mean (stats.rs:21-30) folds colors[k] * weights[k] and weights[k] from zero in
            // push order; keeping the two partial folds per pixel and adding this batch's term
            // performs exactly the same additions in the same order.
            sr = sr + cr * bw;
            sg = sg + cg * bw;
            sb = sb + cb * bw;
            sw += bw;
            mrv[gi] = sr / sw; mgv[gi] = sg / sw; mbv[gi] = sb / sw;
            swv[gi] = sw;
            rays_v[gi] = nr;
            // the variance needs every batch again (the mean moved): keep its illuminance_approx and weight
            if (hist) {
              reinterpret_cast<double2 *>(hist)[(size_t)(kb + gi) * np + key] = make_double2((cr + cr + cb + cg + cg + cg) / 6.0, bw);
            }
          }
        }
      }
      // ---- pass 2: Stats::variance nominators (stats.rs:32-47); only batches > min_batches are ever tested
      if (hist && !done && kb + gc - 1 > rp.min_batches) {
        double ml[kAccChunk];
#pragma unroll
        for (int gi = 0; gi < kAccChunk; gi++) ml[gi] = (mrv[gi] + mrv[gi] + mbv[gi] + mgv[gi] + mgv[gi] + mgv[gi]) / 6.0;
        const double2 *hp = reinterpret_cast<const double2 *>(hist) + key;  // per batch and key: (illuminance_approx, weight), one 16-byte load
        int k = 0;
        for (; k + 4 <= kb; k += 4) {  // entries every batch of the chunk folds
          double hl[4], hw[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const double2 e = hp[(size_t)(k + u) * np];
            hl[u] = e.x;
            hw[u] = e.y;
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int gi = 0; gi < kAccChunk; gi++) {
              double dl = hl[u] - ml[gi];
              nom[gi] += hw[u] * (dl * dl);
            }
          }
        }
        for (; k < kb + gc; k++) {  // the rest, and the chunk's own entries: batch gi folds entries 0..kb+gi
          const double2 e = hp[(size_t)k * np];
          const double hl = e.x, hw = e.y;
#pragma unroll
          for (int gi = 0; gi < kAccChunk; gi++) {
            if (k <= kb + gi) {
              double dl = hl - ml[gi];
              nom[gi] += hw * (dl * dl);
            }
          }
        }
      }
      // ---- pass 3
#pragma unroll
      for (int gi = 0; gi < kAccChunk; gi++) {
        if (gi < gc && !done) {
          const int batch = kb + gi;
          nb = batch + 1;
          mr = mrv[gi]; mg = mgv[gi]; mb = mbv[gi];
          my_rays += rays_v[gi];
          bool stop = false;
          if (batch > rp.min_batches) {  // integrator.rs:321-326
            double variance = nb != 1 ? nom[gi] / swv[gi] : 0.0;
            double interval = 1.96 * (sqrt(variance) / sqrt((double)nb));
            double illum = 0.2126 * mr + 0.7152 * mg + 0.0722 * mb;
            stop = interval < 0.05 * illum;
          }
          done = stop || batch == rp.max_batches - 1;
        }
      }
    }
    if (valid) {
      // here `done` means the pixel is finished (it was forced on for lanes without a slot only)
      my_samples = (unsigned long long)(nb - batch0) * (unsigned long long)rp.S;
      if (done) {
        // integrator.rs:331: mean.clamp().linear_to_gamma(gamma)
        double inv_g = 1.0 / rp.gamma;
        double pr = pow(fmax(fmin(mr, 1.0), 0.0), inv_g), pg = pow(fmax(fmin(mg, 1.0), 0.0), inv_g), pb = pow(fmax(fmin(mb, 1.0), 0.0), inv_g);
        out.pixels[3 * (size_t)p + 0] = pr;
        out.pixels[3 * (size_t)p + 1] = pg;
        out.pixels[3 * (size_t)p + 2] = pb;
        if (out.rgb8) {
          out.rgb8[3 * (size_t)p + 0] = to_rgb8(pr);
          out.rgb8[3 * (size_t)p + 1] = to_rgb8(pg);
          out.rgb8[3 * (size_t)p + 2] = to_rgb8(pb);
        }
        uint64_t actual = (uint64_t)nb * (uint64_t)rp.S;
        if (out.sample_pixels || out.sample_rgb8) {  // integrator.rs:333-337 (usize arithmetic wraps)
          uint64_t num = actual - (uint64_t)rp.min_batches * (uint64_t)rp.S;
          double factor = (double)num / (double)((uint64_t)rp.S * (uint64_t)rp.max_batches);
          if (out.sample_pixels) {
            out.sample_pixels[3 * (size_t)p + 0] = factor * 1.0;
            out.sample_pixels[3 * (size_t)p + 1] = (1.0 - factor) * 1.0;
            out.sample_pixels[3 * (size_t)p + 2] = 0.0;
          }
          if (out.sample_rgb8) {
            out.sample_rgb8[3 * (size_t)p + 0] = to_rgb8(factor * 1.0);
            out.sample_rgb8[3 * (size_t)p + 1] = to_rgb8((1.0 - factor) * 1.0);
            out.sample_rgb8[3 * (size_t)p + 2] = 0;
          }
        }
        if (out.linear) {
          out.linear[3 * (size_t)p + 0] = mr;
          out.linear[3 * (size_t)p + 1] = mg;
          out.linear[3 * (size_t)p + 2] = mb;
        }
        if (out.samples) out.samples[p] = (uint32_t)actual;
      } else {
        ps[0] = sr; ps[np] = sg; ps[2 * np] = sb; ps[3 * np] = sw;
        again = true;
      }
    }
    // BVH queries of the accepted samples: wave reduction, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
      my_rays += __shfl_down(my_rays, off);
      my_samples += __shfl_down(my_samples, off);
    }
    if (lane == 0 && my_samples) {
      atomicAdd(&totals[0], my_rays);
      atomicAdd(&totals[RT_MAX_DEPTH_STATS + 3], my_samples);
    }
    unsigned long long mask = __ballot(again);
    if (mask) {
      int leader = __ffsll((long long)mask) - 1;
      uint32_t prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      uint32_t base = 0;
      if ((int)lane == leader) base = atomicAdd(next_count, (uint32_t)__popcll(mask));
      base = __shfl(base, leader);
      if (again) next_active[base + prefix] = key;
    }
  }
}

__global__ void k_all_keys(RenderParams rp, uint32_t *a, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) a[i] = key_of_pixel(rp, i);
}

// ============================================== primary hits / caller rays
// Both entry points write a ray queue and run the production k_extend on it.
RT_DEV int32_t element_of(const DevScene &sc, uint32_t ref) {
  if (ref == kRefNone) return -1;
  uint32_t kind = (ref >> kRefKindShift) & 3u, idx = ref & kRefIndexMask;
  if (kind == RT_PRIM_TRIANGLE) return (int32_t)tri_record(sc, idx)->element;
  if (kind == RT_PRIM_QUAD) return (int32_t)sc.quads[idx].element;
  return (int32_t)sc.spheres[idx].element;
}

__global__ void __launch_bounds__(256) k_primary_rays(DevScene sc, RenderParams rp, PathQueue q, uint32_t *counts) {
  int64_t n = (int64_t)rp.n_pix * rp.S;
  const uint32_t share = region_share(n);
  if (blockIdx.x == 0) region_publish_counts(n, share, counts);
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    int32_t p = (int32_t)(j / rp.S), i = (int32_t)(j % rp.S);
    int32_t x = p % rp.width;
    int32_t y = image_row(rp, (uint32_t)(p / rp.width));
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
    d3 dir = normalized(pixel_sample - origin);
    uint32_t e = region_place(j, share, q.cap);
    q.ox[e] = origin.x; q.oy[e] = origin.y; q.oz[e] = origin.z;
    q.dx[e] = dir.x; q.dy[e] = dir.y; q.dz[e] = dir.z;
  }
}

__global__ void __launch_bounds__(256) k_caller_rays(int64_t n, const double *origins, const double *directions, PathQueue q, uint32_t *counts) {
  const uint32_t share = region_share(n);
  if (blockIdx.x == 0) region_publish_counts(n, share, counts);
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    d3 dir = normalized(ld3(directions + 3 * j));  // Ray::new, ray.rs:35-48
    uint32_t e = region_place(j, share, q.cap);
    q.ox[e] = origins[3 * j]; q.oy[e] = origins[3 * j + 1]; q.oz[e] = origins[3 * j + 2];
    q.dx[e] = dir.x; q.dy[e] = dir.y; q.dz[e] = dir.z;
  }
}

__global__ void __launch_bounds__(256) k_hits_to_elements(DevScene sc, int64_t n, uint32_t cap, HitBuffer hits, int32_t *prim_id, double *t_out) {
  const uint32_t share = region_share(n);
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    uint32_t e = region_place(j, share, cap);
    uint32_t ref = hits.ref[e];
    prim_id[j] = element_of(sc, ref);
    t_out[j] = (ref == kRefNone) ? 0.0 : hits.t[e];
  }
}

// ================================================================ host side
struct RtSceneHandle;
// Working memory of a call, carved out of an arena owned by the scene handle: the first render
// pays for hipMalloc, repeated renders (benchmark loops, one render per frame) reuse the block.
// A render states its needs first (a planning pass over the same allocation code, which only adds
// up sizes) and the arena is then ONE block of at least that size: a block that is too small is
// freed and replaced -- never kept beside a second one -- so a handle holds no more than its
// largest render needed.  The small query entry points allocate without a plan and may add blocks;
// the next planned call folds them into one again.  One call in flight per handle.
struct Scratch {
  RtSceneHandle *h;
  bool planning = false;
  size_t planned = 0;
  explicit Scratch(RtSceneHandle *handle);
  template <typename T>
  hipError_t alloc(T **out, size_t count) {
    void *p = nullptr;
    hipError_t e = raw(&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) *out = static_cast<T *>(p);
    return e;
  }
  hipError_t raw(void **out, size_t bytes);
  hipError_t reserve(size_t bytes);  // after a planning pass: make the arena one block of >= bytes
  hipError_t event(hipEvent_t *out);
};

struct ArenaBlock {
  char *base;
  size_t size, used;
};

struct RtSceneHandle {
  int device = 0;
  DevScene dev{};
  hipStream_t stream = nullptr;
  std::vector<void *> allocations;
  int32_t n_elements = 0;
  int32_t bvh_depth = 0;
  std::vector<ArenaBlock> arena;  // scratch blocks, kept until the scene is freed
  std::vector<hipEvent_t> events;
  size_t events_used = 0;
  RtProgressFn progress = nullptr;
  void *progress_user = nullptr;
  int32_t n_internal = 0;           // internal nodes resident on the device
  bool device_built = false;        // the tree was built by rt_hip_scene_build
  void *noalloc_scratch = nullptr;  // this device's address of the file-scope scratch array (rt_hip_trace_rays_noalloc)
  Tuning tuning;                    // read from the environment when the handle was made, then rt_hip_scene_tune
};

Scratch::Scratch(RtSceneHandle *handle) : h(handle) {
  for (ArenaBlock &b : h->arena) b.used = 0;
  h->events_used = 0;
}
hipError_t Scratch::raw(void **out, size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (planning) {
    planned += bytes;
    *out = nullptr;
    return hipSuccess;
  }
  for (ArenaBlock &b : h->arena) {
    if (b.size - b.used >= bytes) {
      *out = b.base + b.used;
      b.used += bytes;
      return hipSuccess;
    }
  }
  size_t size = std::max<size_t>(bytes, (size_t)64 << 20);
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, size);
  if (e != hipSuccess) return e;
  h->arena.push_back(ArenaBlock{static_cast<char *>(p), size, bytes});
  *out = p;
  return hipSuccess;
}
hipError_t Scratch::reserve(size_t bytes) {
  planning = false;
  if (h->arena.size() == 1 && h->arena[0].size >= bytes) return hipSuccess;
  // no kernel of an earlier call is still running (calls are blocking), so the blocks can go
  hipError_t e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return e;
  for (ArenaBlock &b : h->arena) (void)hipFree(b.base);
  h->arena.clear();
  void *p = nullptr;
  e = hipMalloc(&p, std::max<size_t>(bytes, 256));
  if (e != hipSuccess) return e;
  h->arena.push_back(ArenaBlock{static_cast<char *>(p), std::max<size_t>(bytes, 256), 0});
  return hipSuccess;
}
hipError_t Scratch::event(hipEvent_t *out) {
  if (h->events_used == h->events.size()) {
    hipEvent_t e;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) return rc;
    h->events.push_back(e);
  }
  *out = h->events[h->events_used++];
  return hipSuccess;
}

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

int row_block_of(const RtConfig *cfg) { return cfg->row_block > 1 ? cfg->row_block : 1; }
int shard_rows(const RtConfig *cfg) {  // rows in the blocks row_first, row_first + row_stride, ... of the image
  const int B = row_block_of(cfg), n_blocks = (cfg->img_height + B - 1) / B;
  if (cfg->row_first >= n_blocks) return 0;
  const int mine = (n_blocks - cfg->row_first + cfg->row_stride - 1) / cfg->row_stride;
  int rows = mine * B;
  // only the image's last block can be short, and if the shard owns it, it is the shard's last too
  if (cfg->row_first + (mine - 1) * cfg->row_stride == n_blocks - 1 && cfg->img_height % B) rows -= B - cfg->img_height % B;
  return rows;
}

int check_config(const RtConfig *cfg) {
  if (!cfg) return fail(RT_ERR_BAD_ARG, "config is NULL");
  if (cfg->img_width <= 0 || cfg->img_height <= 0) return fail(RT_ERR_BAD_ARG, "image size must be positive");
  if (cfg->sample_batch_size <= 0 || cfg->max_sample_batches <= 0 || cfg->min_sample_batches < 0 || cfg->max_depth < 0)
    return fail(RT_ERR_BAD_ARG, "sample_batch_size and max_sample_batches must be >= 1; min_sample_batches, max_depth >= 0");
  if (cfg->row_stride <= 0 || cfg->row_first < 0 || cfg->row_block < 0 || (int64_t)cfg->row_first * row_block_of(cfg) >= cfg->img_height)
    return fail(RT_ERR_BAD_ARG, "row shard out of range");
  if (!(cfg->pixel_radius > 0.0) || !(cfg->gamma_correction > 0.0)) return fail(RT_ERR_BAD_ARG, "pixel_radius and gamma_correction must be > 0");
  if (cfg->work_mib < 0) return fail(RT_ERR_BAD_ARG, "work_mib must be >= 0");
  if (cfg->rng_mode != RT_RNG_CTR)
    return fail(RT_ERR_UNSUPPORTED, "the device path implements RT_RNG_CTR only (RT_RNG_REF is serial per image row)");
  int64_t n_pix = (int64_t)shard_rows(cfg) * cfg->img_width;
  if (n_pix > 0x7fffffff) return fail(RT_ERR_UNSUPPORTED, "more than 2^31-1 pixels in one shard");
  return RT_OK;
}

RenderParams make_params(const RtConfig *cfg, const Tuning &tuning) {
  RenderParams rp{};
  rp.width = cfg->img_width;
  rp.height = cfg->img_height;
  rp.row_first = cfg->row_first;
  rp.row_stride = cfg->row_stride;
  rp.row_block = row_block_of(cfg);
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
  // tile order of the active list: a multiplier near 0.618 * n_tiles, coprime to n_tiles, and its inverse
  // Rows of the shard are neighbours in the image only inside a block of row_block rows, and blocks are
  // row_stride blocks apart: a tile is as high as the largest power of two that divides the block (at most
  // 2^kTileLog; single rows: trade tile height for width by the stride) and correspondingly wider.
  rp.tile_hlog = kTileLog;
  if (cfg->row_stride > 1) {
    if (rp.row_block > 1) {
      rp.tile_hlog = 0;
      while (rp.tile_hlog < kTileLog && rp.row_block % (2 << rp.tile_hlog) == 0) rp.tile_hlog++;
    } else {
      for (int s = cfg->row_stride; s > 1 && rp.tile_hlog > 0; s >>= 1) rp.tile_hlog--;
    }
  }
  if (tuning.tile_hlog >= 0) rp.tile_hlog = (uint32_t)std::min<int>(2 * (int)kTileLog, tuning.tile_hlog);
  rp.tile_wlog = 2 * kTileLog - rp.tile_hlog;
  rp.tiles_x = ((uint32_t)cfg->img_width + (1u << rp.tile_wlog) - 1u) >> rp.tile_wlog;
  rp.n_tiles = rp.tiles_x * (((uint32_t)shard_rows(cfg) + (1u << rp.tile_hlog) - 1u) >> rp.tile_hlog);
  rp.n_keys = (uint64_t)rp.n_tiles << (2 * kTileLog);
  auto gcd = [](uint64_t a, uint64_t b) {
    while (b) {
      uint64_t t = a % b;
      a = b;
      b = t;
    }
    return a;
  };
  uint64_t m = std::max<uint64_t>(1, (uint64_t)(0.6180339887 * rp.n_tiles));
  while (gcd(m, rp.n_tiles) != 1) m++;
  m %= rp.n_tiles;
  if (rp.n_tiles == 1) m = 0;
  // inverse of m modulo n_tiles (extended Euclid)
  int64_t t0 = 0, t1 = 1, r0 = rp.n_tiles, r1 = (int64_t)m;
  while (r1 != 0) {
    int64_t qq = r0 / r1, tt = t0 - qq * t1, rr = r0 - qq * r1;
    t0 = t1; t1 = tt; r0 = r1; r1 = rr;
  }
  if (t0 < 0) t0 += rp.n_tiles;
  rp.tile_mult = (uint32_t)m;
  rp.tile_inv = rp.n_tiles == 1 ? 0u : (uint32_t)t0;
  return rp;
}

int grid_for(int64_t n, int block, int max_blocks) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

// Children whose box the ray misses are skipped whenever that cannot change a result (every box
// bounds its contents); otherwise, or on request, the reference's visit rule is followed exactly.
int walk_mode(const RtSceneHandle *h, int32_t flags) {
  if (flags & kFlagReferenceVisitRule) return 1;
  if (!h->dev.quirk) return 0;
  return h->dev.n_danger > 0 ? 2 : 1;
}

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
  for (ArenaBlock &b : h->arena) (void)hipFree(b.base);
  for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return RT_OK;
}

// Host threads for the per-element loops of an upload: RT_HOST_THREADS, else OMP_NUM_THREADS,
// else the hardware concurrency; never more than 64.
static int upload_threads() {
  for (const char *name : {"RT_HOST_THREADS", "OMP_NUM_THREADS"})
    if (const char *v = getenv(name)) {
      int n = atoi(v);
      if (n >= 1) return std::min(n, 64);
    }
  unsigned hw = std::thread::hardware_concurrency();
  return hw == 0 ? 1 : (int)std::min(hw, 64u);
}
template <typename F>
static void parallel_for(size_t n, F &&fn) {  // fn(begin, end) over [0, n); a worker's exception is rethrown on the caller
  int threads = upload_threads();
  if (threads <= 1 || n < 65536) {
    fn((size_t)0, n);
    return;
  }
  std::vector<std::thread> pool;
  std::vector<std::exception_ptr> failures((size_t)threads);
  auto chunk = [&](int c) {
    try {
      fn(n * (size_t)c / (size_t)threads, n * (size_t)(c + 1) / (size_t)threads);
    } catch (...) {
      failures[(size_t)c] = std::current_exception();
    }
  };
  pool.reserve((size_t)threads);
  for (int c = 0; c < threads; c++) {
    try {
      pool.emplace_back(chunk, c);
    } catch (const std::system_error &) {  // no more threads to be had: the caller does this share itself
      chunk(c);
    }
  }
  for (auto &t : pool) t.join();
  for (auto &f : failures)
    if (f) std::rethrow_exception(f);
}

// No exception may cross the C boundary: every entry point runs its body through this.
template <typename F>
static int guarded(F &&body) {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    return fail(RT_ERR_OOM, "out of host memory");
  } catch (const std::exception &e) {
    return fail(RT_ERR_HIP, std::string("unexpected exception: ") + e.what());
  } catch (...) {
    return fail(RT_ERR_HIP, "unexpected exception");
  }
}

// Element, material and attractor indices of a scene (everything but its nodes).
static int validate_primitives(const RtScene *s) {
  if (s->n_elements <= 0 || !s->elements) return fail(RT_ERR_BAD_ARG, "scene has no elements");
  if (s->n_triangles < 0 || s->n_quads < 0 || s->n_spheres < 0 || s->n_materials <= 0 || s->n_attractors < 0 || !s->materials)
    return fail(RT_ERR_BAD_ARG, "negative count or no materials");
  if ((s->n_triangles && !s->triangles) || (s->n_quads && !s->quads) || (s->n_spheres && !s->spheres) || (s->n_attractors && !s->attractors))
    return fail(RT_ERR_BAD_ARG, "count without array");
  if ((uint32_t)s->n_triangles > kRefIndexMask || (uint32_t)s->n_quads > kRefIndexMask || (uint32_t)s->n_spheres > kRefIndexMask)
    return fail(RT_ERR_UNSUPPORTED, "more than 2^28-1 primitives of one kind");
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
    // only spheres and quads can be attractors (elements/mod.rs:68-75); the sampling code has no triangle branch
    if (s->elements[e].kind != RT_PRIM_QUAD && s->elements[e].kind != RT_PRIM_SPHERE)
      return fail(RT_ERR_BAD_ARG, "attractor " + std::to_string(i) + " is not a quad or a sphere");
  }
  return RT_OK;
}

// Wide nodes: up to four children per node, made from the grid nodes by opening, largest box first, the internal children of a
// node until four boxes are held (or only leaves are left).
// Measured (one MI355X): the walk fetches half as many node records on them (100 000 triangles: 14.7 -> 7.0 per ray) but the
// same bytes, and is no faster while the binary records stay in the caches (k_extend 633 -> 690 ms; 2 M triangles 1115 -> 1180 ms);
// on a 10 M-triangle soup, whose 320 MB of node records do not, it is 6 % faster.  Hence: made for trees above 256 MB of binary
// records, RT_HIP_WIDE=1 / 0 forces them on / off.
static bool wide_nodes_wanted(const Tuning &tuning, size_t n_internal) {
  return tuning.wide >= 0 ? tuning.wide != 0 : n_internal * sizeof(DevNodeQ) > ((size_t)256 << 20);
}
// `nodes_q`: the binary grid records, internal children named by their index in it; `root`: the root's index.  Leaves `nodes_w`
// empty when a lane's stack could need more entries on them than the walk provides for.
static void collapse_to_wide(const std::vector<DevNodeQ> &nodes_q, uint32_t root, const double grid_step[3], std::vector<DevNodeW> &nodes_w,
                             int32_t &stack_depth_w) {
  struct Child {
    uint16_t mn[3], mx[3];
    uint32_t ref;
  };
  auto left_of = [](const DevNodeQ &q) { return Child{{q.lmin[0], q.lmin[1], q.lmin[2]}, {q.lmax[0], q.lmax[1], q.lmax[2]}, q.left}; };
  auto right_of = [](const DevNodeQ &q) { return Child{{q.rmin[0], q.rmin[1], q.rmin[2]}, {q.rmax[0], q.rmax[1], q.rmax[2]}, q.right}; };
  auto area_of = [&](const Child &c) {
    double dx = (c.mx[0] - c.mn[0]) * grid_step[0], dy = (c.mx[1] - c.mn[1]) * grid_step[1], dz = (c.mx[2] - c.mn[2]) * grid_step[2];
    return dx * dy + dy * dz + dz * dx;
  };
  // the wide node that stands for binary node `bin`: boxes and leaf refs filled in, internal children left as kRefNone with
  // the binary node each stands for in `below`
  auto open_node = [&](uint32_t bin, DevNodeW &w, uint32_t below[4]) {
    Child ch[4];
    int n = 2;
    ch[0] = left_of(nodes_q[bin]);
    ch[1] = right_of(nodes_q[bin]);
    while (n < 4) {
      int best = -1;
      double best_area = -1.0;
      for (int i = 0; i < n; i++)
        if (!(ch[i].ref & kRefLeaf)) {
          double a = area_of(ch[i]);
          if (a > best_area) best_area = a, best = i;
        }
      if (best < 0) break;
      const DevNodeQ &open = nodes_q[ch[best].ref];
      ch[best] = left_of(open);
      ch[n++] = right_of(open);
    }
    for (int i = 0; i < 4; i++) {
      below[i] = kRefNone;
      if (i >= n) {
        w.px[i] = w.py[i] = w.pz[i] = 0;
        w.child[i] = kRefNone;
        continue;
      }
      w.px[i] = (uint32_t)ch[i].mn[0] | ((uint32_t)ch[i].mx[0] << 16);
      w.py[i] = (uint32_t)ch[i].mn[1] | ((uint32_t)ch[i].mx[1] << 16);
      w.pz[i] = (uint32_t)ch[i].mn[2] | ((uint32_t)ch[i].mx[2] << 16);
      w.child[i] = (ch[i].ref & kRefLeaf) ? ch[i].ref : kRefNone;
      if (!(ch[i].ref & kRefLeaf)) below[i] = ch[i].ref;
    }
  };
  struct Item {
    uint32_t bin, wide;
  };
  // everything below binary node `bin`, depth-first; out[0] stands for `bin`, internal children are named by their index in `out`
  auto collapse_subtree = [&](uint32_t bin, std::vector<DevNodeW> &out) {
    std::vector<Item> todo;
    out.emplace_back();
    todo.push_back({bin, 0u});
    while (!todo.empty()) {
      Item it = todo.back();
      todo.pop_back();
      DevNodeW w;
      uint32_t below[4];
      open_node(it.bin, w, below);
      for (int i = 0; i < 4; i++)
        if (below[i] != kRefNone) {
          w.child[i] = (uint32_t)out.size();
          todo.push_back({below[i], (uint32_t)out.size()});
          out.emplace_back();
        }
      out[it.wide] = w;
    }
  };
  nodes_w.clear();
  const int threads = upload_threads();
  if (threads <= 1 || nodes_q.size() < ((size_t)1 << 18)) {
    nodes_w.reserve(nodes_q.size() / 2 + 1);
    collapse_subtree(root, nodes_w);
  } else {
    // the top of the tree breadth-first until there are subtrees enough to share out among the threads; then every subtree
    // on its own, and the pieces one after the other
    struct Pending {
      uint32_t bin, parent, slot;  // the wide node `parent` names it in child[slot] once its index is known
    };
    std::vector<Pending> fifo;
    size_t head = 0;
    fifo.push_back({root, kRefNone, 0u});
    const size_t want = (size_t)threads * 16;
    while (head < fifo.size() && fifo.size() - head < want) {
      Pending p = fifo[head++];
      const uint32_t me = (uint32_t)nodes_w.size();
      if (p.parent != kRefNone) nodes_w[p.parent].child[p.slot] = me;
      DevNodeW w;
      uint32_t below[4];
      open_node(p.bin, w, below);
      nodes_w.push_back(w);
      for (uint32_t i = 0; i < 4; i++)
        if (below[i] != kRefNone) fifo.push_back({below[i], me, i});
    }
    const size_t n_sub = fifo.size() - head;
    std::vector<std::vector<DevNodeW>> sub(n_sub);
    std::atomic<size_t> next{0};
    std::vector<std::exception_ptr> failures((size_t)threads);
    std::vector<std::thread> pool;
    auto worker = [&](int c) {
      try {
        for (size_t k; (k = next.fetch_add(1)) < n_sub;) collapse_subtree(fifo[head + k].bin, sub[k]);
      } catch (...) {
        failures[(size_t)c] = std::current_exception();
      }
    };
    pool.reserve((size_t)threads);
    for (int c = 1; c < threads; c++) {
      try {
        pool.emplace_back(worker, c);
      } catch (const std::system_error &) {  // no more threads to be had: the ones there are share the work
        break;
      }
    }
    worker(0);
    for (auto &t : pool) t.join();
    for (auto &f : failures)
      if (f) std::rethrow_exception(f);
    size_t total = nodes_w.size();
    for (const auto &v : sub) total += v.size();
    nodes_w.reserve(total);
    for (size_t k = 0; k < n_sub; k++) {
      const uint32_t offset = (uint32_t)nodes_w.size();
      const Pending &p = fifo[head + k];
      nodes_w[p.parent].child[p.slot] = offset;
      for (DevNodeW w : sub[k]) {
        for (int i = 0; i < 4; i++)
          if (!(w.child[i] & kRefLeaf)) w.child[i] += offset;  // (an unused slot, kRefNone, carries the leaf bit)
        nodes_w.push_back(w);
      }
      std::vector<DevNodeW>().swap(sub[k]);
    }
  }
  // stack entries a lane can need below a node: all but one of its children pushed, then the deepest child
  // (children are stored after their parent, so one backward pass does it)
  std::vector<int32_t> need(nodes_w.size(), 0);
  for (size_t i = nodes_w.size(); i-- > 0;) {
    int32_t deepest = 0, k = 0;
    for (int c = 0; c < 4; c++) {
      uint32_t r = nodes_w[i].child[c];
      if (r == kRefNone) continue;
      k++;
      if (!(r & kRefLeaf)) deepest = std::max(deepest, need[r]);
    }
    need[i] = deepest + k - 1;
  }
  stack_depth_w = std::max(need[0], 1);
  if (stack_depth_w > kMaxStackDepth + kWideLdsLevels) nodes_w.clear(), stack_depth_w = 1;
}

// Global-memory part of the traversal stacks of a tree higher than the LDS part holds: [level][wave][lane].
static int alloc_stack_spill(RtSceneHandle *h) {
  h->dev.stack_spill = nullptr;
  // binary walks: blocks of kOvfBlock entries leave LDS when a push finds kOvfLevels there, i.e. with 16 + 8 b entries on the stack
  // and b blocks outside, and the stack never holds more entries than the tree is high
  int levels = h->dev.stack_depth > h->tuning.spill_above ? ((h->dev.stack_depth - kOvfLevels - 1) / kOvfBlock + 1) * kOvfBlock : 0;
  if (h->dev.nodes_w && h->dev.stack_depth_w > kWideLdsLevels)
    levels = std::max(levels, RT_WIDE_STACK_MODE == 1 ? h->dev.stack_depth_w - kWideLdsLevels : ((h->dev.stack_depth_w - kOvfLevels - 1) / kOvfBlock + 1) * kOvfBlock);
  if (levels <= 0) return RT_OK;
  return dev_alloc(h, &h->dev.stack_spill, (size_t)levels * kMaxExtendWaves * 64);
}

// The packet walk reads the 64-byte f32 records and visits, per wave, the union of what its 64 rays need: it pays while that node set
// is small against the caches and the rays of neighbouring pixels still share their leaves.  Measured on the generated mesh at
// 1200 x 1200, bounce 0, packet walk against k_extend: 100 000 triangles 76 / 105 ms, 300 000: 48 / 56, 600 000: 55 / 57, 1 M: 61 / 58,
// 2 M (2048 x 2048): 174 / 166; 10 M-triangle soup 12.5 / 11.2.  Hence: on for trees up to 32 MB of such records (500 000 nodes).
static void resolve_packet_default(RtSceneHandle *h) {
  if (h->tuning.packet < 0) h->tuning.packet = (size_t)h->n_internal * sizeof(DevNode) <= ((size_t)32 << 20) ? 1 : 0;
}
static int noalloc_scratch_of(int device, void **out);  // per-device address of the scratch array of rt_hip_trace_rays_noalloc
static int noalloc_warm_up(RtSceneHandle *h);            // one query through that entry point's kernels, at set-up

static int scene_upload_impl(const RtScene *s, int device, RtSceneHandle **out) {
  g_error.clear();
  if (!s || !out) return fail(RT_ERR_BAD_ARG, "scene or out is NULL");
  *out = nullptr;
  const Tuning tuning = tuning_from_env();
  const bool verbose = tuning.verbose != 0;
  auto lap_t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    auto now = std::chrono::steady_clock::now();
    if (verbose) fprintf(stderr, "rt_hip upload: %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - lap_t0).count());
    lap_t0 = now;
  };
  // ---- validate every index before anything reaches a kernel
  if (s->n_nodes <= 0 || !s->nodes) return fail(RT_ERR_BAD_ARG, "scene has no nodes");
  if ((uint32_t)s->n_nodes > kRefIndexMask) return fail(RT_ERR_UNSUPPORTED, "more than 2^28-1 nodes");
  {
    int vrc = validate_primitives(s);
    if (vrc != RT_OK) return vrc;
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
  lap("validation + depth");

  // Does every child box contain the primitives below it?  (The reference's quad box, quad.rs:86,
  // does not for quads that are not axis-aligned.)  If not, k_extend must reproduce the
  // reference's irregular visit rule.  Checked locally, node by node: the box of a leaf child must
  // contain its primitive and the box of an internal child must contain both boxes stored in that
  // child -- by induction every box then contains everything below it.  (A tree whose boxes
  // bound their contents without being nested fails this test and is merely walked the slow way.)
  std::atomic<bool> nested{true};
  // Leaf children whose primitive sticks out of the child box, kept apart from boxes that are not
  // nested: if they are few, only the rays that come near them need the reference's rule.
  std::vector<int32_t> stray_leaves;
  std::mutex stray_mutex;
  {
    struct B { double mn[3], mx[3]; };
    auto grow = [](B &b, double x, double y, double z) {
      const double p[3] = {x, y, z};
      for (int a = 0; a < 3; a++) {
        if (!(p[a] >= b.mn[a])) b.mn[a] = p[a] != p[a] ? -INFINITY : p[a];
        if (!(p[a] <= b.mx[a])) b.mx[a] = p[a] != p[a] ? INFINITY : p[a];
      }
    };
    auto content_of = [&](int32_t child) {  // what the box of `child` has to contain
      const RtNode &n = s->nodes[child];
      B b;
      for (int a = 0; a < 3; a++) b.mn[a] = INFINITY, b.mx[a] = -INFINITY;
      if (n.is_leaf) {
        RtElementRef e = s->elements[n.element_index];
        if (e.kind == RT_PRIM_TRIANGLE) {
          const RtTriangle &t = s->triangles[e.index];
          grow(b, t.v0.x, t.v0.y, t.v0.z);
          grow(b, t.v0.x + t.v0v1.x, t.v0.y + t.v0v1.y, t.v0.z + t.v0v1.z);
          grow(b, t.v0.x + t.v0v2.x, t.v0.y + t.v0v2.y, t.v0.z + t.v0v2.z);
        } else if (e.kind == RT_PRIM_QUAD) {
          const RtQuad &qd = s->quads[e.index];
          for (int cu = 0; cu < 2; cu++)
            for (int cv = 0; cv < 2; cv++)
              grow(b, qd.q.x + cu * qd.u.x + cv * qd.v.x, qd.q.y + cu * qd.u.y + cv * qd.v.y, qd.q.z + cu * qd.u.z + cv * qd.v.z);
        } else {
          const RtSphere &sp = s->spheres[e.index];
          double rad = std::fabs(sp.radius);
          grow(b, sp.center.x - rad, sp.center.y - rad, sp.center.z - rad);
          grow(b, sp.center.x + rad, sp.center.y + rad, sp.center.z + rad);
        }
        // triangle corners are re-derived from v0 + edge, which can round one ulp past the stored box
        for (int a = 0; a < 3; a++) {
          double m = std::fmax(std::fabs(b.mn[a]), std::fabs(b.mx[a])) * 4.5e-16;
          b.mn[a] += m;
          b.mx[a] -= m;
        }
      } else {
        for (int a = 0; a < 3; a++) {
          b.mn[a] = std::fmin(n.child_l_min[a], n.child_r_min[a]);
          b.mx[a] = std::fmax(n.child_l_max[a], n.child_r_max[a]);
        }
      }
      return b;
    };
    auto inside = [](const B &b, const double *mn, const double *mx) {
      for (int a = 0; a < 3; a++)
        if (!(mn[a] <= b.mn[a] && b.mx[a] <= mx[a])) return false;
      return true;
    };
    parallel_for((size_t)s->n_nodes, [&](size_t begin, size_t end) {
      bool ok = true;
      std::vector<int32_t> mine;
      for (size_t i = begin; i < end && ok; i++) {
        const RtNode &n = s->nodes[i];
        if (n.is_leaf || depth[i] < 0) continue;
        const int32_t child[2] = {n.child_l_index, n.child_r_index};
        const double *mn[2] = {n.child_l_min, n.child_r_min}, *mx[2] = {n.child_l_max, n.child_r_max};
        for (int c = 0; c < 2 && ok; c++) {
          if (inside(content_of(child[c]), mn[c], mx[c])) continue;
          if (s->nodes[child[c]].is_leaf && mine.size() < 64) mine.push_back(child[c]);
          else ok = false;
        }
      }
      if (!ok) nested.store(false, std::memory_order_relaxed);
      if (!mine.empty()) {
        std::lock_guard<std::mutex> lock(stray_mutex);
        stray_leaves.insert(stray_leaves.end(), mine.begin(), mine.end());
      }
    });
  }
  // true bounds of the stray primitives (a root that is a leaf has no box to stick out of)
  constexpr size_t kMaxDanger = 16;
  std::vector<double> danger;
  if (nested.load() && !stray_leaves.empty() && stray_leaves.size() <= kMaxDanger) {
    for (int32_t leaf : stray_leaves) {
      RtElementRef e = s->elements[s->nodes[leaf].element_index];
      double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      auto add = [&](double x, double y, double z) {
        const double p[3] = {x, y, z};
        for (int a = 0; a < 3; a++) lo[a] = std::fmin(lo[a], p[a]), hi[a] = std::fmax(hi[a], p[a]);
      };
      if (e.kind == RT_PRIM_TRIANGLE) {
        const RtTriangle &t = s->triangles[e.index];
        add(t.v0.x, t.v0.y, t.v0.z);
        add(t.v0.x + t.v0v1.x, t.v0.y + t.v0v1.y, t.v0.z + t.v0v1.z);
        add(t.v0.x + t.v0v2.x, t.v0.y + t.v0v2.y, t.v0.z + t.v0v2.z);
      } else if (e.kind == RT_PRIM_QUAD) {
        const RtQuad &qd = s->quads[e.index];
        for (int cu = 0; cu < 2; cu++)
          for (int cv = 0; cv < 2; cv++) add(qd.q.x + cu * qd.u.x + cv * qd.v.x, qd.q.y + cu * qd.u.y + cv * qd.v.y, qd.q.z + cu * qd.u.z + cv * qd.v.z);
      } else {
        const RtSphere &sp = s->spheres[e.index];
        double rad = std::fabs(sp.radius);
        add(sp.center.x - rad, sp.center.y - rad, sp.center.z - rad);
        add(sp.center.x + rad, sp.center.y + rad, sp.center.z + rad);
      }
      bool finite = true;
      for (int a = 0; a < 3; a++) {
        double pad = 1e-9 * (std::fabs(lo[a]) + std::fabs(hi[a])) + 1e-9;  // the hit test accepts points a rounding off the plane
        lo[a] -= pad;
        hi[a] += pad;
        finite = finite && std::isfinite(lo[a]) && std::isfinite(hi[a]);
      }
      if (!finite) {
        danger.clear();
        break;
      }
      danger.insert(danger.end(), lo, lo + 3);
      danger.insert(danger.end(), hi, hi + 3);
    }
  }
  bool single_position = true;  // every reachable node has one parent, every element one leaf
  lap("box containment");
  // The reference's leaf order (depth-first, left child first) decides exact ties between
  // primitives; the reordering traversal needs it per element.  An element named by more than
  // one reachable leaf has no single position: such a tree is walked in the reference's order.
  std::vector<uint32_t> leaf_rank_of_element((size_t)s->n_elements, 0xffffffffu);
  // Triangle records are stored in that leaf order too: triangles that are close in the tree,
  // hence in space, share cache lines whatever their order in the caller's array.
  std::vector<uint32_t> tri_slot((size_t)s->n_triangles, 0xffffffffu);
  uint32_t next_tri_slot = 0;
  auto place_triangle = [&](int32_t element) {
    RtElementRef er = s->elements[element];
    if (er.kind == RT_PRIM_TRIANGLE && tri_slot[(size_t)er.index] == 0xffffffffu) tri_slot[(size_t)er.index] = next_tri_slot++;
  };
  {
    std::vector<int32_t> todo{0};
    uint32_t next_rank = 0;
    int64_t visited = 0;
    while (!todo.empty()) {
      if (++visited > (int64_t)s->n_nodes) {  // some node has two parents
        single_position = false;
        break;
      }
      int32_t i = todo.back();
      todo.pop_back();
      const RtNode &n = s->nodes[i];
      if (n.is_leaf) {
        uint32_t &r = leaf_rank_of_element[(size_t)n.element_index];
        if (r != 0xffffffffu) single_position = false;
        r = next_rank++;
        place_triangle(n.element_index);
      } else {
        todo.push_back(n.child_r_index);
        todo.push_back(n.child_l_index);
      }
    }
  }

  bool boxes_bound_contents = single_position && nested.load() && stray_leaves.empty();
  if (!single_position) danger.clear();
  // every box bounds what lies below it, or the few primitives that do not are fenced by `danger`:
  // the order-free walk (and its grid nodes) exists for this tree
  bool fast_walk_exists = boxes_bound_contents || !danger.empty();
  for (int e = 0; e < s->n_elements; e++) place_triangle(e);  // elements no reachable leaf names
  for (uint32_t &slot : tri_slot)
    if (slot == 0xffffffffu) slot = next_tri_slot++;            // triangles no element names
  lap("leaf order");
  // Storage order of the internal nodes.  The reference's preorder puts a node next to its left
  // child only; once the node array outgrows the caches every visit is a 128-byte line fetched for
  // one 32-byte record.  For very large trees (the ordering pass is serial: 0.1 s per 2 M nodes, worth
  // it only where it saves more) the records are laid out in blocks that share a line: a
  // block starts at a node and takes in, largest box first (the child a ray is likeliest to enter),
  // descendants until the line is full; what is left over starts the next blocks.
  {
    bool blocks = tuning.node_order >= 0 ? tuning.node_order == 1 : (size_t)n_internal * sizeof(DevNodeQ) > ((size_t)64 << 20);
    if (blocks && single_position) {
      constexpr int kPerLine = 128 / (int)sizeof(DevNodeQ);
      struct Cand { int32_t node; float area; };
      auto push_children = [&](std::vector<Cand> &f, int32_t i) {
        const RtNode &n = s->nodes[i];
        const int32_t c[2] = {n.child_l_index, n.child_r_index};
        const double *mn[2] = {n.child_l_min, n.child_r_min}, *mx[2] = {n.child_l_max, n.child_r_max};
        for (int k = 0; k < 2; k++) {
          if (s->nodes[c[k]].is_leaf) continue;
          double dx = mx[k][0] - mn[k][0], dy = mx[k][1] - mn[k][1], dz = mx[k][2] - mn[k][2];
          f.push_back({c[k], (float)(dx * dy + dy * dz + dz * dx)});
        }
      };
      std::fill(internal_index.begin(), internal_index.end(), -1);
      int32_t next = 0;
      std::vector<int32_t> roots;
      std::vector<Cand> frontier;
      if (!s->nodes[0].is_leaf) roots.push_back(0);
      while (!roots.empty()) {
        int32_t r = roots.back();
        roots.pop_back();
        int room = kPerLine - next % kPerLine;
        frontier.clear();
        internal_index[(size_t)r] = next++;
        push_children(frontier, r);
        for (room--; room > 0 && !frontier.empty(); room--) {
          size_t best = 0;
          for (size_t k = 1; k < frontier.size(); k++)
            if (frontier[k].area > frontier[best].area) best = k;
          int32_t c = frontier[best].node;
          frontier.erase(frontier.begin() + (ptrdiff_t)best);
          internal_index[(size_t)c] = next++;
          push_children(frontier, c);
        }
        // the largest of what is left is laid out next
        std::sort(frontier.begin(), frontier.end(), [](const Cand &x, const Cand &y) { return x.area < y.area; });
        for (const Cand &c : frontier) roots.push_back(c.node);
      }
      for (int i = 0; i < s->n_nodes; i++)  // internal nodes the root does not reach
        if (!s->nodes[i].is_leaf && internal_index[(size_t)i] < 0) internal_index[(size_t)i] = next++;
    }
  }
  lap("node order");
  // ---- flatten
  auto ref_of_element = [&](int e) -> uint32_t {
    RtElementRef er = s->elements[e];
    uint32_t index = er.kind == RT_PRIM_TRIANGLE ? tri_slot[(size_t)er.index] : (uint32_t)er.index;
    return kRefLeaf | ((uint32_t)er.kind << kRefKindShift) | index;
  };
  auto ref_of_node = [&](int i) -> uint32_t {
    const RtNode &n = s->nodes[i];
    return n.is_leaf ? ref_of_element(n.element_index) : (uint32_t)internal_index[(size_t)i];
  };
  std::vector<DevNode> nodes((size_t)n_internal);
  parallel_for((size_t)s->n_nodes, [&](size_t begin, size_t end) {
  for (size_t i = begin; i < end; i++) {
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
  });
  // ---- the reference's own f64 boxes, for trees whose result depends on exactly which boxes a ray hits
  std::vector<DevNodeD> nodes_d;
  if (!boxes_bound_contents && !tuning.no_exact_boxes) {  // (the switch exists for the test that shows why they are needed)
    nodes_d.resize((size_t)n_internal);
    parallel_for((size_t)s->n_nodes, [&](size_t begin, size_t end) {
      for (size_t i = begin; i < end; i++) {
        const RtNode &n = s->nodes[i];
        if (n.is_leaf) continue;
        DevNodeD &d = nodes_d[(size_t)internal_index[i]];
        for (int a = 0; a < 3; a++) {
          d.lmin[a] = n.child_l_min[a]; d.lmax[a] = n.child_l_max[a];
          d.rmin[a] = n.child_r_min[a]; d.rmax[a] = n.child_r_max[a];
        }
        d.left = ref_of_node(n.child_l_index);
        d.right = ref_of_node(n.child_r_index);
        d.pad[0] = d.pad[1] = 0;
      }
    });
  }
  // ---- the same boxes on a 16-bit grid over the bounds of all child boxes.  Used only by the
  // order-free walk, and only if the grid is fine compared with the boxes: a cell much larger
  // than a typical box would make every box test pass.
  std::vector<DevNodeQ> nodes_q;
  double grid_min[3] = {0, 0, 0}, grid_step[3] = {1, 1, 1};
  if (fast_walk_exists && !tuning.no_grid) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool finite = true;
    for (int i = 0; i < s->n_nodes; i++) {
      const RtNode &n = s->nodes[i];
      if (n.is_leaf || depth[(size_t)i] < 0) continue;
      for (int a = 0; a < 3; a++) {
        const double v[4] = {n.child_l_min[a], n.child_l_max[a], n.child_r_min[a], n.child_r_max[a]};
        for (double x : v) {
          if (!(std::fabs(x) <= 1e30)) finite = false;
          lo[a] = std::fmin(lo[a], x);
          hi[a] = std::fmax(hi[a], x);
        }
      }
    }
    if (finite && n_internal > 0) {
      for (int a = 0; a < 3; a++) {
        grid_min[a] = lo[a];
        grid_step[a] = (hi[a] - lo[a]) / 65535.0;
        if (!(grid_step[a] > 0.0)) grid_step[a] = 1.0;
        grid_step[a] *= 1.0 + 1e-12;  // the top grid value must reach hi[a] in spite of rounding
      }
      nodes_q.resize((size_t)n_internal);
      // grid value at or below x / at or above x, checked against the value the kernel's constants stand for
      auto q_down = [&](double x, int a) {
        double q = std::floor((x - grid_min[a]) / grid_step[a]);
        while (q > 0.0 && grid_min[a] + q * grid_step[a] > x - std::fabs(x) * 1e-15) q -= 1.0;
        return (uint16_t)std::fmin(std::fmax(q, 0.0), 65535.0);
      };
      auto q_up = [&](double x, int a) {
        double q = std::ceil((x - grid_min[a]) / grid_step[a]);
        while (q < 65535.0 && grid_min[a] + q * grid_step[a] < x + std::fabs(x) * 1e-15) q += 1.0;
        return (uint16_t)std::fmin(std::fmax(q, 0.0), 65535.0);
      };
      std::vector<double> growth((size_t)upload_threads() + 1, 0.0), size((size_t)upload_threads() + 1, 0.0);
      std::atomic<int> slot_counter{0};
      parallel_for((size_t)s->n_nodes, [&](size_t begin, size_t end) {
        double my_growth = 0.0, my_size = 0.0;
        for (size_t i = begin; i < end; i++) {
          const RtNode &n = s->nodes[i];
          if (n.is_leaf) continue;
          DevNodeQ &d = nodes_q[(size_t)internal_index[i]];
          for (int a = 0; a < 3; a++) {
            d.lmin[a] = q_down(n.child_l_min[a], a);
            d.lmax[a] = q_up(n.child_l_max[a], a);
            d.rmin[a] = q_down(n.child_r_min[a], a);
            d.rmax[a] = q_up(n.child_r_max[a], a);
            my_growth += ((double)(d.lmax[a] - d.lmin[a]) + (double)(d.rmax[a] - d.rmin[a])) * grid_step[a];
            my_size += (n.child_l_max[a] - n.child_l_min[a]) + (n.child_r_max[a] - n.child_r_min[a]);
          }
          d.left = ref_of_node(n.child_l_index);
          d.right = ref_of_node(n.child_r_index);
        }
        int slot = slot_counter.fetch_add(1);
        growth[(size_t)slot] = my_growth;
        size[(size_t)slot] = my_size;
      });
      double total_growth = 0.0, total_size = 0.0;
      for (double g : growth) total_growth += g;
      for (double z : size) total_size += z;
      // summed edge lengths of the grid boxes vs the original ones: accept up to 10 % more
      if (verbose) fprintf(stderr, "rt_hip upload: boxes on the 16-bit grid are %.1f %% larger (summed edges)\n", 100.0 * (total_growth / total_size - 1.0));
      if (!(total_growth <= 1.10 * total_size)) nodes_q.clear();
    }
  }
  // ---- wide nodes (collapse_to_wide above)
  std::vector<DevNodeW> nodes_w;
  int32_t stack_depth_w = 1;
  if (!nodes_q.empty() && !s->nodes[0].is_leaf && wide_nodes_wanted(tuning, (size_t)n_internal)) {
    collapse_to_wide(nodes_q, (uint32_t)internal_index[0], grid_step, nodes_w, stack_depth_w);
    if (verbose) fprintf(stderr, "rt_hip upload: %zu wide nodes for %d binary ones, stack need %d\n", nodes_w.size(), n_internal, stack_depth_w);
  }
  std::vector<DevTriangle> tris((size_t)s->n_triangles);
  std::vector<DevTriShade> tri_shade((size_t)s->n_triangles);
  std::vector<DevQuad> quads((size_t)s->n_quads);
  std::vector<DevSphere> spheres((size_t)s->n_spheres);
  parallel_for((size_t)s->n_elements, [&](size_t begin, size_t end) {
  for (size_t e = begin; e < end; e++) {
    RtElementRef er = s->elements[e];
    if (er.kind == RT_PRIM_TRIANGLE) {
      const RtTriangle &t = s->triangles[er.index];
      DevTriangle &d = tris[(size_t)tri_slot[(size_t)er.index]];
      d.v0[0] = t.v0.x; d.v0[1] = t.v0.y; d.v0[2] = t.v0.z;
      d.e1[0] = t.v0v1.x; d.e1[1] = t.v0v1.y; d.e1[2] = t.v0v1.z;
      d.e2[0] = t.v0v2.x; d.e2[1] = t.v0v2.y; d.e2[2] = t.v0v2.z;
      d.element = (uint32_t)e;
      d.rank = leaf_rank_of_element[(size_t)e];
      DevTriShade &sh = tri_shade[(size_t)tri_slot[(size_t)er.index]];
      sh.normal[0] = t.normal.x; sh.normal[1] = t.normal.y; sh.normal[2] = t.normal.z;
      sh.material = t.material;
      sh.pad = 0;
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
      d.rank = leaf_rank_of_element[(size_t)e];
      d.pad = 0;
    } else {
      const RtSphere &sp = s->spheres[er.index];
      DevSphere &d = spheres[(size_t)er.index];
      d.center[0] = sp.center.x; d.center[1] = sp.center.y; d.center[2] = sp.center.z;
      d.radius = sp.radius;
      d.element = (uint32_t)e;
      d.material = (uint32_t)sp.material;
      d.rank = leaf_rank_of_element[(size_t)e];
      d.pad = 0;
    }
  }
  });
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

  lap("flatten");
  // ---- upload
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(RT_ERR_BAD_ARG, "no such device");
  HIP_TRY(hipSetDevice(device));
  RtSceneHandle *h = new (std::nothrow) RtSceneHandle;
  if (!h) return fail(RT_ERR_OOM, "out of host memory");
  struct FreeOnUnwind {  // an exception below (host allocation) must not leak the device copies made so far
    RtSceneHandle *h;
    ~FreeOnUnwind() {
      if (h) rt_hip_scene_free(h);
    }
  } unwind{h};
  h->device = device;
  h->tuning = tuning;
  h->n_elements = s->n_elements;
  h->bvh_depth = max_depth;
  h->n_internal = n_internal;
  int rc = RT_OK;
  auto step = [&](int r) {
    if (rc == RT_OK) rc = r;
  };
  step(dev_upload(h, &h->dev.nodes, nodes));
  h->dev.nodes_q = nullptr;
  if (!nodes_q.empty()) step(dev_upload(h, &h->dev.nodes_q, nodes_q));
  h->dev.nodes_d = nullptr;
  if (!nodes_d.empty()) step(dev_upload(h, &h->dev.nodes_d, nodes_d));
  h->dev.nodes_w = nullptr;
  if (!nodes_w.empty()) step(dev_upload(h, &h->dev.nodes_w, nodes_w));
  h->dev.root_w = nodes_w.empty() ? ref_of_node(0) : 0u;
  h->dev.stack_depth_w = stack_depth_w;
  for (int a = 0; a < 3; a++) h->dev.grid_min[a] = grid_min[a], h->dev.grid_step[a] = grid_step[a];
  // Triangle records that do not fit the caches are fetched line by line from HBM: at 80 bytes apart every second
  // record lies across two 128-byte lines, at 128 bytes apart none does.
  const bool padded = tuning.tri_pad >= 0 ? tuning.tri_pad != 0 : tris.size() * sizeof(DevTriangle) > ((size_t)256 << 20);
  h->dev.tri_stride = padded ? 128u : (uint32_t)sizeof(DevTriangle);
  if (padded) {
    struct alignas(128) Padded {
      DevTriangle t;
    };
    static_assert(sizeof(Padded) == 128, "one record per line");
    std::vector<Padded> wide_tris(tris.size());
    parallel_for(tris.size(), [&](size_t begin, size_t end) {
      for (size_t i = begin; i < end; i++) wide_tris[i].t = tris[i];
    });
    const Padded *dst = nullptr;
    step(dev_upload(h, &dst, wide_tris));
    h->dev.triangles = reinterpret_cast<const DevTriangle *>(dst);
  } else {
    step(dev_upload(h, &h->dev.triangles, tris));
  }
  step(dev_upload(h, &h->dev.tri_shade, tri_shade));
  step(dev_upload(h, &h->dev.quads, quads));
  step(dev_upload(h, &h->dev.spheres, spheres));
  step(dev_upload(h, &h->dev.materials, mats));
  step(dev_upload(h, &h->dev.attractors, attractors));
  if (rc == RT_OK && hipStreamCreate(&h->stream) != hipSuccess) rc = fail(RT_ERR_HIP, "hipStreamCreate failed");
  if (rc == RT_OK) rc = noalloc_scratch_of(device, &h->noalloc_scratch);
  if (rc != RT_OK) return rc;  // `unwind` frees the handle; the message stays
  h->dev.n_attractors = s->n_attractors;
  h->dev.n_materials = s->n_materials;
  h->dev.root = ref_of_node(0);
  h->dev.stack_depth = std::max(max_depth, 1);
  if ((rc = alloc_stack_spill(h)) != RT_OK) return rc;
  h->dev.quirk = boxes_bound_contents ? 0 : 1;
  h->dev.danger = nullptr;
  h->dev.n_danger = 0;
  if (fast_walk_exists && !boxes_bound_contents) {
    rc = dev_upload(h, &h->dev.danger, danger);
    if (rc != RT_OK) return rc;
    h->dev.n_danger = (int32_t)(danger.size() / 6);
  }
  h->dev.camera = s->camera;
  resolve_packet_default(h);
  if ((rc = noalloc_warm_up(h)) != RT_OK) return rc;
  lap("device copies");
  unwind.h = nullptr;
  *out = h;
  return RT_OK;
}

static int render_resident_impl(RtSceneHandle *h, const RtConfig *cfg, RtImage *out, RtStats *stats) {
  g_error.clear();
  if (!h || !out || !out->pixels) return fail(RT_ERR_BAD_ARG, "handle, out or out->pixels is NULL");
  int rc = check_config(cfg);
  if (rc != RT_OK) return rc;
  if (out->memory != RT_MEM_HOST && out->memory != RT_MEM_DEVICE) return fail(RT_ERR_BAD_ARG, "bad RtImage.memory");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  RenderParams rp = make_params(cfg, h->tuning);
  const int S = rp.S;
  const int32_t n_pix = rp.n_pix;
  const bool timed = (rp.flags & kFlagTimeKernels) != 0;
  const bool counted = (rp.flags & kFlagCountVisits) != 0;
  auto wall0 = std::chrono::steady_clock::now();

  // Paths per round.  A round traces G consecutive batches of every active pixel at once (the
  // batches after a pixel's stopping batch are speculative and dropped by k_accumulate), which
  // keeps launches large late in a render and rounds few (every launch ends with a tail in which
  // the chip waits for its slowest wave); HBM is big enough for 2^29 paths in flight (~125 GB).
  const int64_t bytes_per_path = 2 * (9 * 8 + 8 + 4) + (8 + 4) + (int64_t)sizeof(PathRecord);
  int64_t max_paths = 1ll << 29;
  {
    size_t free_b = 0, total_b = 0;
    size_t held = 0;  // the arena of earlier renders is reused, so it counts as available
    for (const ArenaBlock &blk : h->arena) held += blk.size;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) max_paths = std::min<int64_t>(max_paths, (int64_t)((free_b + held) / 2) / bytes_per_path);
  }
  if (cfg->work_mib > 0) max_paths = std::min<int64_t>(max_paths, ((int64_t)cfg->work_mib << 20) / bytes_per_path);
  if (h->tuning.max_paths >= 1) max_paths = h->tuning.max_paths;
  const int max_group = h->tuning.max_group;
  max_paths = std::max<int64_t>(max_paths, S);
  max_paths = std::min<int64_t>(max_paths, 0xfffffff0ll / 3);
  const int64_t cap_paths = std::min<int64_t>(max_paths / S * S, (int64_t)n_pix * S * std::min(rp.max_batches, max_group));
  if (S > cap_paths) return fail(RT_ERR_UNSUPPORTED, "sample_batch_size too large for one round");

  Scratch sc(h);
  // queue entries per region (whole waves) and in total
  const uint32_t region_cap = (uint32_t)(((cap_paths + kRegions - 1) / kRegions + 63) / 64 * 64);
  const size_t q_entries = (size_t)region_cap * kRegions;
  PathQueue q[2];
  HitBuffer hits;
  PathResults results;
  double *weights, *sums, *hist = nullptr;
  uint32_t *active[2], *counters;
  // Primary rays once per round (RT_FLAG_REUSE_PRIMARY_HITS): possible when no lens moves the origin from batch to batch
  const bool reuse_primary = (rp.flags & kFlagReusePrimaryHits) && h->dev.camera.defocus_angle <= 0.0 && rp.max_depth > 0;
  PrimaryReuse primary{};
  HitBuffer primary_hits{};
  const uint32_t primary_cap = (uint32_t)((((int64_t)n_pix * S + kRegions - 1) / kRegions + 63) / 64 * 64);
  // counters (each on its own 128-byte line): [0] next_count; then per bounce 0..max_depth the
  // kRegions queue sizes; then per bounce the kRegions work-claim counters of k_extend, twice
  // (a tree with stray primitives is walked in two launches per bounce, each claiming the whole queue)
  const size_t per_bounce = (size_t)kRegions * kCounterStride;
  // ... and last the compact queue of primary rays traced once per round: its sizes and its two sets of claim counters
  const size_t n_counters = kCounterStride + (size_t)(3 * rp.max_depth + 1 + 3) * per_bounce;
  // totals: [0] BVH queries of accepted samples, [1..16] queries traced per bounce, [17] node visits,
  // [18] leaf tests, [19] accepted samples, [20] wave node steps, [21] wave leaf steps
  constexpr int kTotals = RT_MAX_DEPTH_STATS + 13;  // ... [22..24] wave cycles in refill / node steps / leaf tests and retire, [25..27] lane states, [28] stack block moves
  unsigned long long *totals;
  PixelOutputs po{};
  // k_accumulate appends the surviving pixels wave by wave in whatever order the waves finish;
  // the list is sorted by key again after every round that drops pixels, because the order of
  // the pixels is the order of the rays in every queue: consecutive keys are the pixels of one
  // tile, whose rays pass through the same nodes.
  size_t sort_bytes = 0;
  int sort_bits = 1;
  while (sort_bits < 32 && (1ll << sort_bits) < ((int64_t)rp.n_tiles << (2 * kTileLog))) sort_bits++;
  HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, n_pix, 0, sort_bits, st));
  void *sort_tmp = nullptr;
  // every buffer of the render; run once to add up the sizes, then for real on the reserved block
  auto allocate = [&]() -> int {
    for (int k = 0; k < 2; k++) {
      HIP_TRY(sc.alloc(&q[k].ox, q_entries)); HIP_TRY(sc.alloc(&q[k].oy, q_entries)); HIP_TRY(sc.alloc(&q[k].oz, q_entries));
      HIP_TRY(sc.alloc(&q[k].dx, q_entries)); HIP_TRY(sc.alloc(&q[k].dy, q_entries)); HIP_TRY(sc.alloc(&q[k].dz, q_entries));
      HIP_TRY(sc.alloc(&q[k].tx, q_entries)); HIP_TRY(sc.alloc(&q[k].ty, q_entries)); HIP_TRY(sc.alloc(&q[k].tz, q_entries));
      HIP_TRY(sc.alloc(&q[k].rng, q_entries));
      HIP_TRY(sc.alloc(&q[k].pid, q_entries));
      q[k].cap = region_cap;
    }
    HIP_TRY(sc.alloc(&hits.t, q_entries));
    HIP_TRY(sc.alloc(&hits.ref, q_entries));
    if (reuse_primary) {
      const size_t pe = (size_t)primary_cap * kRegions;
      HIP_TRY(sc.alloc(&primary.dx, pe)); HIP_TRY(sc.alloc(&primary.dy, pe)); HIP_TRY(sc.alloc(&primary.dz, pe));
      HIP_TRY(sc.alloc(&primary.entry, q_entries));
      HIP_TRY(sc.alloc(&primary_hits.t, pe));
      HIP_TRY(sc.alloc(&primary_hits.ref, pe));
    }
    HIP_TRY(sc.alloc(&results.rec, cap_paths));
    HIP_TRY(sc.alloc(&weights, (size_t)S * (size_t)rp.n_keys));
    HIP_TRY(sc.alloc(&sums, 4 * (size_t)rp.n_keys));
    // per-batch (illuminance_approx, weight) history, needed only if the z-test can ever run
    if (rp.max_batches - 1 > rp.min_batches) HIP_TRY(sc.alloc(&hist, (size_t)rp.max_batches * 2 * (size_t)rp.n_keys));
    HIP_TRY(sc.alloc(&active[0], n_pix));
    HIP_TRY(sc.alloc(&active[1], n_pix));
    HIP_TRY(sc.alloc(&counters, n_counters));
    HIP_TRY(sc.alloc(&totals, kTotals));
    if (out->memory == RT_MEM_DEVICE) {
      po.pixels = out->pixels;
      po.sample_pixels = out->sample_pixels;
      po.linear = out->linear;
      po.samples = out->samples;
      po.rgb8 = out->rgb8;
      po.sample_rgb8 = out->sample_rgb8;
    } else {
      HIP_TRY(sc.alloc(&po.pixels, 3 * (size_t)n_pix));
      if (out->sample_pixels) HIP_TRY(sc.alloc(&po.sample_pixels, 3 * (size_t)n_pix));
      if (out->linear) HIP_TRY(sc.alloc(&po.linear, 3 * (size_t)n_pix));
      if (out->samples) HIP_TRY(sc.alloc(&po.samples, (size_t)n_pix));
      if (out->rgb8) HIP_TRY(sc.alloc(&po.rgb8, 3 * (size_t)n_pix));
      if (out->sample_rgb8) HIP_TRY(sc.alloc(&po.sample_rgb8, 3 * (size_t)n_pix));
    }
    HIP_TRY(sc.raw(&sort_tmp, sort_bytes));
    return RT_OK;
  };
  sc.planning = true;
  rc = allocate();
  if (rc != RT_OK) return rc;
  HIP_TRY(sc.reserve(sc.planned));
  rc = allocate();
  if (rc != RT_OK) return rc;
  uint32_t *qcounts = counters + kCounterStride;
  uint32_t *work = qcounts + (size_t)(rp.max_depth + 1) * per_bounce;
  uint32_t *work2 = work + (size_t)rp.max_depth * per_bounce;
  uint32_t *primary_counts = work2 + (size_t)rp.max_depth * per_bounce, *primary_work = primary_counts + per_bounce, *primary_work2 = primary_work + per_bounce;
  primary.counts = primary_counts;
  const int walk = walk_mode(h, rp.flags);
  HIP_TRY(hipMemsetAsync(totals, 0, kTotals * sizeof(unsigned long long), st));

  const int block = 256;
  const int max_blocks = 256 * 8;  // 256 CUs x 8 resident workgroups
  hipLaunchKernelGGL(k_all_keys, dim3(grid_for(n_pix, block, max_blocks)), dim3(block), 0, st, rp, active[1], (uint32_t)n_pix);
  LAUNCHED("k_all_keys");
  HIP_TRY(hipcub::DeviceRadixSort::SortKeys(sort_tmp, sort_bytes, active[1], active[0], (int)n_pix, 0, sort_bits, st));

  struct TimedLaunch {
    hipEvent_t a, b;
    int kind;  // 0 extend, 1 shade, 2 other
    int depth;
  };
  std::vector<TimedLaunch> timeline;
  auto mark_begin = [&](int kind, int depth = 0) -> hipError_t {
    if (!timed) return hipSuccess;
    TimedLaunch t{};
    t.kind = kind;
    t.depth = depth;
    hipError_t e = sc.event(&t.a);
    if (e != hipSuccess) return e;
    e = sc.event(&t.b);
    if (e != hipSuccess) return e;
    timeline.push_back(t);
    return hipEventRecord(t.a, st);
  };
  auto mark_end = [&]() -> hipError_t { return timed ? hipEventRecord(timeline.back().b, st) : hipSuccess; };

  uint32_t n_active = (uint32_t)n_pix;
  int cur = 0;
  uint64_t rounds = 0;
  int b = 0;
  int group = 1;
  // after the first round: jump straight to this many batches if (almost) no pixel stopped in it
  const int first_jump = h->tuning.first_group;
  while (b < rp.max_batches && n_active > 0) {
    // how many batches this round traces at once
    int G;
    if (b == 0) {
      // no pixel can stop before batch min+1 (integrator.rs:321): those batches are never speculative
      G = std::min(rp.max_batches, rp.min_batches + 2);
    } else {
      G = std::min(group, rp.max_batches - b);
    }
    G = (int)std::max<int64_t>(1, std::min<int64_t>(G, cap_paths / ((int64_t)n_active * S)));
    // if even one batch of every active pixel exceeds the buffers, walk the pixels in chunks
    const uint32_t chunk_slots = (uint32_t)std::min<int64_t>(n_active, cap_paths / ((int64_t)S * G));
    HIP_TRY(hipMemsetAsync(counters, 0, sizeof(uint32_t), st));
    for (uint32_t base = 0; base < n_active; base += chunk_slots) {
      int32_t ns = (int32_t)std::min<uint32_t>(chunk_slots, n_active - base);
      int64_t np = (int64_t)ns * S * G;
      HIP_TRY(hipMemsetAsync(qcounts, 0, (n_counters - kCounterStride) * sizeof(uint32_t), st));
      // k_shade: a whole number of workgroups per region, and many more of them than fit the
      // chip at once -- short workgroups even out the uneven cost of the material branches
      const int shade_blocks = h->tuning.shade_blocks;
      int g_shade = (int)std::min<int64_t>(shade_blocks, (np / kShadeBlock / kRegions + 1) * kRegions);
      HIP_TRY(mark_begin(2));
      // one batch per round has nothing to share: its paths carry their rays as usual
      PrimaryReuse pr = primary;
      pr.cap = (reuse_primary && G > 1) ? primary_cap : 0u;
      PathQueue primary_q{};
      primary_q.dx = pr.dx; primary_q.dy = pr.dy; primary_q.dz = pr.dz;
      primary_q.cap = primary_cap;
      if (S <= kGenMaxShared) {
        const int spb = std::max(1, std::min(block, kGenMaxShared) / S);  // pixel slots per workgroup: about one (slot, sample) per lane
        hipLaunchKernelGGL(k_generate, dim3(grid_for(((int64_t)ns + spb - 1) / spb, 1, 1 << 20)), dim3(block), 0, st, h->dev, rp, active[cur] + base, ns, spb, b, G, q[0], weights, qcounts, pr);
      } else {
        hipLaunchKernelGGL(k_generate_strided, dim3(grid_for((int64_t)ns * S, block, max_blocks)), dim3(block), 0, st, h->dev, rp, active[cur] + base, ns, b, G, q[0], weights, qcounts, pr);
      }
      LAUNCHED("k_generate");
      HIP_TRY(mark_end());
      for (int k = 0; k < rp.max_depth; k++) {
        const bool shared_rays = k == 0 && pr.cap != 0;  // this bounce's rays are the compact queue's
        HIP_TRY(mark_begin(0, k));
        if (shared_rays)
          launch_extend(h->dev, h->tuning, walk, counted, true, (int64_t)ns * S, st, primary_q, primary_counts, primary_hits, primary_work, primary_work2,
                        totals + RT_MAX_DEPTH_STATS + 1, true);
        else
          launch_extend(h->dev, h->tuning, walk, counted, k == 0 && h->dev.camera.defocus_angle <= 0.0, np, st, q[k & 1], qcounts + k * per_bounce, hits, work + k * per_bounce,
                        work2 + k * per_bounce, totals + RT_MAX_DEPTH_STATS + 1, k == 0);
        LAUNCHED("k_extend");
        HIP_TRY(mark_end());
        HIP_TRY(mark_begin(1, k));
        if (shared_rays)
          hipLaunchKernelGGL(k_shade<true>, dim3(g_shade), dim3(kShadeBlock), 0, st, h->dev, q[0], qcounts, primary_hits, q[1], qcounts + per_bounce, results, 0,
                             (int32_t)(rp.max_depth == 1), rp.flags, pr);
        else
          hipLaunchKernelGGL(k_shade<false>, dim3(g_shade), dim3(kShadeBlock), 0, st, h->dev, q[k & 1], qcounts + k * per_bounce, hits, q[(k + 1) & 1],
                             qcounts + (k + 1) * per_bounce, results, k, (int32_t)(k == rp.max_depth - 1), rp.flags, pr);
        LAUNCHED("k_shade");
        HIP_TRY(mark_end());
      }
      HIP_TRY(mark_begin(2));
      if (rp.max_depth == 0) {
        HIP_TRY(hipMemsetAsync(results.rec, 0, (size_t)np * sizeof(PathRecord), st));
      }
      hipLaunchKernelGGL(k_accumulate, dim3(grid_for(ns, block, max_blocks)), dim3(block), 0, st, rp, active[cur] + base, ns, b, G, results, weights,
                         sums, hist, active[cur ^ 1], counters, po, qcounts, pr.cap ? primary_counts : nullptr, totals);
      LAUNCHED("k_accumulate");
      HIP_TRY(mark_end());
    }
    HIP_TRY(hipGetLastError());
    uint32_t next = 0;
    HIP_TRY(hipMemcpyAsync(&next, counters, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // A pixel that stops inside a round wastes on average half a round of its own samples, so the
    // share of wasted work is about half the share of pixels that stopped, whatever the group size:
    // speculate further ahead while at most 5 % stop, back off when more than 20 % do.
    uint32_t stopped = n_active - next;
    if (b + G < rp.max_batches) {
      if ((uint64_t)stopped * 20 <= n_active) {
        // after a quiet first round: at least first_jump batches, and enough of them to half-fill the
        // buffers -- a small shard of the image would otherwise spend its first rounds on short launches
        int jump = rounds == 0 ? (int)std::max<int64_t>(first_jump, cap_paths / 2 / ((int64_t)n_active * S)) : 1;
        group = std::min(max_group, std::max(jump, group * 2));
      }
      else if ((uint64_t)stopped * 5 > n_active) group = std::max(1, group / 2);
    }
    if (stopped == 0) {
      // nothing dropped: the list in use is still right, and still sorted
    } else if (next > 0) {
      HIP_TRY(hipcub::DeviceRadixSort::SortKeys(sort_tmp, sort_bytes, active[cur ^ 1], active[cur], (int)next, 0, sort_bits, st));
    }
    n_active = next;
    b += G;
    rounds++;
    if (h->progress) {
      // batches still to come for the pixels still active, over all batches of all pixels
      double left = (double)n_active * (double)(rp.max_batches - b);
      h->progress(h->progress_user, n_active == 0 || b >= rp.max_batches ? 1.0 : 1.0 - left / ((double)n_pix * (double)rp.max_batches));
    }
  }
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());

  if (out->memory == RT_MEM_HOST) {
    HIP_TRY(hipMemcpy(out->pixels, po.pixels, 3 * (size_t)n_pix * sizeof(double), hipMemcpyDeviceToHost));
    if (out->sample_pixels) HIP_TRY(hipMemcpy(out->sample_pixels, po.sample_pixels, 3 * (size_t)n_pix * sizeof(double), hipMemcpyDeviceToHost));
    if (out->linear) HIP_TRY(hipMemcpy(out->linear, po.linear, 3 * (size_t)n_pix * sizeof(double), hipMemcpyDeviceToHost));
    if (out->samples) HIP_TRY(hipMemcpy(out->samples, po.samples, (size_t)n_pix * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (out->rgb8) HIP_TRY(hipMemcpy(out->rgb8, po.rgb8, 3 * (size_t)n_pix, hipMemcpyDeviceToHost));
    if (out->sample_rgb8) HIP_TRY(hipMemcpy(out->sample_rgb8, po.sample_rgb8, 3 * (size_t)n_pix, hipMemcpyDeviceToHost));
  }
  if (stats) {
    memset(stats, 0, sizeof *stats);
    unsigned long long host_totals[kTotals];
    HIP_TRY(hipMemcpy(host_totals, totals, sizeof host_totals, hipMemcpyDeviceToHost));
    stats->rays = host_totals[0];
    for (int k = 0; k < RT_MAX_DEPTH_STATS; k++) {
      stats->rays_by_depth[k] = host_totals[1 + k];
      stats->rays_traced += host_totals[1 + k];
    }
    stats->node_visits = host_totals[RT_MAX_DEPTH_STATS + 1];
    stats->leaf_tests = host_totals[RT_MAX_DEPTH_STATS + 2];
    stats->box_tests = 2 * stats->node_visits;
    stats->samples = host_totals[RT_MAX_DEPTH_STATS + 3];
    stats->wave_node_steps = host_totals[RT_MAX_DEPTH_STATS + 4];
    stats->wave_leaf_steps = host_totals[RT_MAX_DEPTH_STATS + 5];
    stats->wave_cycles_refill = host_totals[RT_MAX_DEPTH_STATS + 6];
    stats->wave_cycles_node = host_totals[RT_MAX_DEPTH_STATS + 7];
    stats->wave_cycles_leaf = host_totals[RT_MAX_DEPTH_STATS + 8];
    stats->stack_block_moves = host_totals[RT_MAX_DEPTH_STATS + 12];
    if (h->tuning.verbose && host_totals[RT_MAX_DEPTH_STATS + 4]) {
      // sampled at the first of the 1 + RT_EXTRA_STEPS node steps between two votes (the reference-rule walk votes after every step)
      const double sampled_slots = 64.0 * host_totals[RT_MAX_DEPTH_STATS + 4] / (walk == 1 ? 1.0 : 1.0 + RT_EXTRA_STEPS);
      fprintf(stderr, "[rt_hip] lanes of a node step that are not visiting a node: no ray %.3f, walk over but leaf tests pending %.3f, holding a leaf %.3f\n",
              host_totals[RT_MAX_DEPTH_STATS + 9] / sampled_slots, host_totals[RT_MAX_DEPTH_STATS + 10] / sampled_slots,
              host_totals[RT_MAX_DEPTH_STATS + 11] / sampled_slots);
    }
    stats->batches = rounds;
    double shade_ms_by_depth[RT_MAX_DEPTH_STATS] = {};
    for (const TimedLaunch &t : timeline) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
      if (t.kind == 0) {
        stats->trace_kernel_ms += ms;
        stats->trace_launches++;
        stats->trace_ms_by_depth[t.depth < RT_MAX_DEPTH_STATS ? t.depth : RT_MAX_DEPTH_STATS - 1] += ms;
      } else if (t.kind == 1) {
        stats->shade_kernel_ms += ms;
        shade_ms_by_depth[t.depth < RT_MAX_DEPTH_STATS ? t.depth : RT_MAX_DEPTH_STATS - 1] += ms;
      } else {
        stats->other_kernel_ms += ms;
      }
    }
    if (h->tuning.verbose && timed) {
      fprintf(stderr, "[rt_hip] k_shade ms by bounce:");
      for (int k = 0; k < rp.max_depth && k < RT_MAX_DEPTH_STATS; k++) fprintf(stderr, " %.1f", shade_ms_by_depth[k]);
      fprintf(stderr, "\n");
    }
    stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  }
  return RT_OK;
}

int rt_hip_set_progress(RtSceneHandle *h, RtProgressFn fn, void *user) {
  g_error.clear();
  if (!h) return fail(RT_ERR_BAD_ARG, "handle is NULL");
  h->progress = fn;
  h->progress_user = fn ? user : nullptr;
  return RT_OK;
}

static int scene_tune_impl(RtSceneHandle *h, const char *name, int64_t value) {
  g_error.clear();
  if (!h || !name) return fail(RT_ERR_BAD_ARG, "handle or name is NULL");
  auto out_of_range = [&](long long lo, long long hi) {
    return fail(RT_ERR_BAD_ARG, std::string(name) + " must lie in [" + std::to_string(lo) + ", " + std::to_string(hi) + "]");
  };
  if (!strcmp(name, "max_paths")) {
    if (value < 0) return out_of_range(0, INT64_MAX);
    h->tuning.max_paths = value;
    return RT_OK;
  }
  for (const TuningName &n : kTuningNames)
    if (!strcmp(name, n.name)) {
      if (n.layout) return fail(RT_ERR_BAD_ARG, std::string(name) + " shapes the device copy of the scene: set " + n.env + " before the handle is made");
      if (value < n.lo || value > n.hi) return out_of_range(n.lo, n.hi);
      h->tuning.*n.field = (int)value;
      if (n.field == &Tuning::shade_blocks) h->tuning.shade_blocks = std::max((int)kRegions, (int)value / (int)kRegions * (int)kRegions);
      return RT_OK;
    }
  for (const ExtendTuningName &n : kExtendTuningNames)
    if (!strcmp(name, n.name)) {
      if (value < n.lo || value > n.hi) return out_of_range(n.lo, n.hi);
      h->tuning.extend.*n.field = (int)value;
      return RT_OK;
    }
  return fail(RT_ERR_BAD_ARG, std::string("no tuning parameter named ") + name);
}

// `coherent`: the queue holds primary rays in pixel order (see launch_extend)
static int trace_queue(RtSceneHandle *h, Scratch &sc, int64_t n, PathQueue q, uint32_t *d_counts, int32_t flags, int32_t *prim_id, double *t,
                       bool coherent = false) {
  const size_t entries = (size_t)q.cap * kRegions;
  HitBuffer hits;
  int32_t *d_id;
  double *d_t;
  HIP_TRY(sc.alloc(&hits.t, entries));
  HIP_TRY(sc.alloc(&hits.ref, entries));
  HIP_TRY(sc.alloc(&d_id, (size_t)n));
  HIP_TRY(sc.alloc(&d_t, (size_t)n));
  const int block = 256;
  uint32_t *work = d_counts + (size_t)kRegions * kCounterStride;
  launch_extend(h->dev, h->tuning, walk_mode(h, flags), false, false, n, h->stream, q, d_counts, hits, work, work + (size_t)kRegions * kCounterStride, nullptr,
                coherent || h->tuning.packet == 2);
  LAUNCHED("k_extend");
  hipLaunchKernelGGL(k_hits_to_elements, dim3(grid_for(n, block, 2048)), dim3(block), 0, h->stream, h->dev, n, q.cap, hits, d_id, d_t);
  LAUNCHED("k_hits_to_elements");
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(prim_id, d_id, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(t, d_t, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return RT_OK;
}

// Ray-only queue (origin + direction) for n rays, plus its region counts and claim counters (zeroed).
static int alloc_ray_queue(RtSceneHandle *h, Scratch &sc, int64_t n, PathQueue *q, uint32_t **d_counts) {
  if (n > 0xfffffff0ll / 2) return fail(RT_ERR_UNSUPPORTED, "more than 2^31 rays in one call");
  memset(q, 0, sizeof *q);
  q->cap = (uint32_t)(((n + kRegions - 1) / kRegions + 63) / 64 * 64);
  const size_t entries = (size_t)q->cap * kRegions;
  HIP_TRY(sc.alloc(&q->ox, entries)); HIP_TRY(sc.alloc(&q->oy, entries)); HIP_TRY(sc.alloc(&q->oz, entries));
  HIP_TRY(sc.alloc(&q->dx, entries)); HIP_TRY(sc.alloc(&q->dy, entries)); HIP_TRY(sc.alloc(&q->dz, entries));
  const size_t words = 3 * (size_t)kRegions * kCounterStride;
  HIP_TRY(sc.alloc(d_counts, words));
  HIP_TRY(hipMemsetAsync(*d_counts, 0, words * sizeof(uint32_t), h->stream));
  return RT_OK;
}

static int primary_hits_impl(RtSceneHandle *h, const RtConfig *cfg, int32_t *prim_id, double *t) {
  g_error.clear();
  if (!h || !prim_id || !t) return fail(RT_ERR_BAD_ARG, "NULL argument");
  int rc = check_config(cfg);
  if (rc != RT_OK) return rc;
  HIP_TRY(hipSetDevice(h->device));
  RenderParams rp = make_params(cfg, h->tuning);
  int64_t n = (int64_t)rp.n_pix * rp.S;
  Scratch sc(h);
  PathQueue q;
  uint32_t *d_count;
  rc = alloc_ray_queue(h, sc, n, &q, &d_count);
  if (rc != RT_OK) return rc;
  hipLaunchKernelGGL(k_primary_rays, dim3(grid_for(n, 256, 2048)), dim3(256), 0, h->stream, h->dev, rp, q, d_count);
  LAUNCHED("k_primary_rays");
  return trace_queue(h, sc, n, q, d_count, cfg->flags, prim_id, t, true);  // the walk a render's first bounce takes
}

static int trace_rays_impl(RtSceneHandle *h, int64_t n, const double *origins, const double *directions, int32_t *prim_id, double *t) {
  g_error.clear();
  if (!h || n < 0 || (n > 0 && (!origins || !directions || !prim_id || !t))) return fail(RT_ERR_BAD_ARG, "NULL argument or negative count");
  if (n == 0) return RT_OK;
  HIP_TRY(hipSetDevice(h->device));
  Scratch sc(h);
  double *d_o, *d_d;
  HIP_TRY(sc.alloc(&d_o, 3 * (size_t)n));
  HIP_TRY(sc.alloc(&d_d, 3 * (size_t)n));
  HIP_TRY(hipMemcpy(d_o, origins, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_d, directions, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  PathQueue q;
  uint32_t *d_count;
  int rc = alloc_ray_queue(h, sc, n, &q, &d_count);
  if (rc != RT_OK) return rc;
  hipLaunchKernelGGL(k_caller_rays, dim3(grid_for(n, 256, 2048)), dim3(256), 0, h->stream, n, d_o, d_d, q, d_count);
  LAUNCHED("k_caller_rays");
  return trace_queue(h, sc, n, q, d_count, 0, prim_id, t);
}

// ------------------------------------------- closest-hit queries that allocate nothing
// rt_hip_trace_rays_noalloc: the queries of rt_hip_trace_rays for callers that cannot allocate
// device memory.  Everything the kernels need beyond the caller's inputs and outputs -- the SoA
// ray queue, the hit buffer, the region counts and claim counters -- lives in this file-scope
// array of the code object.  Every GPU the library runs on holds its own copy at its own address
// until the process ends; host code never passes the array's name to a kernel or a memset, only
// the per-device address hipGetSymbolAddress returned for it (noalloc_scratch_of).
constexpr int64_t kNoAllocMaxRays = 1 << 17;
static_assert(kNoAllocMaxRays == RT_NOALLOC_MAX_RAYS, "the header states the limit");
constexpr size_t kNoAllocRegionCap = (size_t)(((kNoAllocMaxRays + kRegions - 1) / kRegions + 63) / 64 * 64);
constexpr size_t kNoAllocEntries = kNoAllocRegionCap * kRegions;
constexpr size_t kNoAllocCounterWords = 3 * (size_t)kRegions * kCounterStride;  // region counts + two sets of claim counters
constexpr size_t kNoAllocCounterBytes = (kNoAllocCounterWords * sizeof(uint32_t) + 255) / 256 * 256;
constexpr size_t kNoAllocBytes = kNoAllocCounterBytes + kNoAllocEntries * (6 * sizeof(double) + sizeof(double) + sizeof(uint32_t));
static __device__ __align__(256) unsigned char scratch[kNoAllocBytes];

constexpr int kMaxDevices = 64;
static std::mutex g_noalloc_mutex;
static void *g_noalloc_ptr[kMaxDevices];  // device address of `scratch` on each device, once known
static std::mutex g_noalloc_busy[kMaxDevices];  // one query at a time per device: they all use that device's one array

// Device address of `scratch` on `device`, which must be the current device.  Looked up once per
// device (at scene upload: set-up time, where loading the code object may allocate).
static int noalloc_scratch_of(int device, void **out) {
  if (device < 0 || device >= kMaxDevices) return fail(RT_ERR_UNSUPPORTED, "device index beyond the table of scratch addresses");
  std::lock_guard<std::mutex> lock(g_noalloc_mutex);
  if (!g_noalloc_ptr[device]) {
    void *ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(RT_ERR_HIP, "hipGetSymbolAddress returned a null address for the scratch array");
    g_noalloc_ptr[device] = ptr;
  }
  *out = g_noalloc_ptr[device];
  return RT_OK;
}

// The query itself; the caller holds g_noalloc_busy[h->device] and has made h->device current.
static int trace_rays_noalloc_locked(RtSceneHandle *h, int64_t n, const double *d_origins, const double *d_directions, int32_t *d_prim_id,
                                     double *d_t) {
  // carve the device's copy of the array: counters first, then the queue and the hit buffer
  unsigned char *base = static_cast<unsigned char *>(h->noalloc_scratch);
  uint32_t *d_counts = reinterpret_cast<uint32_t *>(base);
  double *cols = reinterpret_cast<double *>(base + kNoAllocCounterBytes);
  PathQueue q{};
  q.cap = (uint32_t)(((n + kRegions - 1) / kRegions + 63) / 64 * 64);  // <= kNoAllocRegionCap
  const size_t entries = (size_t)q.cap * kRegions;
  q.ox = cols; q.oy = cols + entries; q.oz = cols + 2 * entries;
  q.dx = cols + 3 * entries; q.dy = cols + 4 * entries; q.dz = cols + 5 * entries;
  HitBuffer hits;
  hits.t = cols + 6 * entries;
  hits.ref = reinterpret_cast<uint32_t *>(cols + 7 * entries);
  // the array keeps the last call's contents: the counters must start at zero
  HIP_TRY(hipMemsetAsync(d_counts, 0, kNoAllocCounterBytes, h->stream));
  hipLaunchKernelGGL(k_caller_rays, dim3(grid_for(n, 256, 2048)), dim3(256), 0, h->stream, n, d_origins, d_directions, q, d_counts);
  LAUNCHED("k_caller_rays");
  uint32_t *work = d_counts + (size_t)kRegions * kCounterStride;
  launch_extend(h->dev, h->tuning, walk_mode(h, 0), false, false, n, h->stream, q, d_counts, hits, work, work + (size_t)kRegions * kCounterStride, nullptr);
  LAUNCHED("k_extend");
  hipLaunchKernelGGL(k_hits_to_elements, dim3(grid_for(n, 256, 2048)), dim3(256), 0, h->stream, h->dev, n, q.cap, hits, d_prim_id, d_t);
  LAUNCHED("k_hits_to_elements");
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return RT_OK;
}

static int trace_rays_noalloc_impl(RtSceneHandle *h, int64_t n, const double *d_origins, const double *d_directions, int32_t *d_prim_id,
                                   double *d_t) {
  g_error.clear();
  if (!h || n < 0 || (n > 0 && (!d_origins || !d_directions || !d_prim_id || !d_t))) return fail(RT_ERR_BAD_ARG, "NULL argument or negative count");
  if (n > kNoAllocMaxRays)
    return fail(RT_ERR_UNSUPPORTED, "rt_hip_trace_rays_noalloc takes at most " + std::to_string(kNoAllocMaxRays) + " rays per call, got " +
                                        std::to_string(n) + ": split the batch or use rt_hip_trace_rays");
  if (n == 0) return RT_OK;
  if (!h->noalloc_scratch) return fail(RT_ERR_HIP, "the scene handle has no scratch address");
  HIP_TRY(hipSetDevice(h->device));
  std::lock_guard<std::mutex> one_at_a_time(g_noalloc_busy[h->device]);
  return trace_rays_noalloc_locked(h, n, d_origins, d_directions, d_prim_id, d_t);
}

// Set-up for rt_hip_trace_rays_noalloc: trace one ray through it now.  Whatever the runtime sets up lazily at a kernel's first
// launch on a queue (loading the code object, private-segment memory for the few registers the walk keeps there) happens
// here, where allocating is allowed, and not inside a caller's first query.  The ray and its result live in the last 256 bytes
// of the scratch array, which a one-ray query does not use otherwise -- but a full-size query of another thread does (they are
// the last hit references of its last region), so the device's lock is held from the copy of the ray to the end of the query:
// a set-up never writes into a query in flight on the same GPU.
static int noalloc_warm_up(RtSceneHandle *h) {
  static_assert(kNoAllocBytes > kNoAllocCounterBytes + 64 * kRegions * 60 + 256, "room for the warm-up ray behind a one-ray queue");
  unsigned char *tail = static_cast<unsigned char *>(h->noalloc_scratch) + kNoAllocBytes - 256;
  const double ray[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 1.0};
  std::lock_guard<std::mutex> one_at_a_time(g_noalloc_busy[h->device]);
  HIP_TRY(hipMemcpy(tail, ray, sizeof ray, hipMemcpyHostToDevice));
  double *d = reinterpret_cast<double *>(tail);
  return trace_rays_noalloc_locked(h, 1, d, d + 3, reinterpret_cast<int32_t *>(d + 7), d + 6);
}

#include "rt_build_device.h"
#include "rt_comm.h"

// ------------------------------------------------------- the C entry points
int rt_hip_comm_id(unsigned char *id) {
  return guarded([&] { return comm_id_impl(id); });
}
int rt_hip_comm_init(const unsigned char *id, int rank, int world, int device, RtComm **out) {
  return guarded([&] { return comm_init_impl(id, rank, world, device, out); });
}
int rt_hip_comm_free(RtComm *comm) {
  return guarded([&] { return comm_free_impl(comm); });
}
int rt_hip_render_ranks(RtSceneHandle *handle, RtComm *comm, const RtConfig *cfg, RtImage *out, RtStats *stats) {
  return guarded([&] { return render_ranks_impl(handle, comm, cfg, out, stats); });
}
int rt_hip_scene_build(const RtScene *scene, int device, RtSceneHandle **out) {
  return guarded([&] { return scene_build_impl(scene, device, out); });
}
int rt_hip_scene_download_tree(RtSceneHandle *handle, RtNode *nodes, int32_t capacity, int32_t *n_nodes) {
  return guarded([&] { return scene_download_tree_impl(handle, nodes, capacity, n_nodes); });
}
int rt_hip_scene_tune(RtSceneHandle *handle, const char *name, int64_t value) {
  return guarded([&] { return scene_tune_impl(handle, name, value); });
}
int rt_hip_scene_upload(const RtScene *scene, int device, RtSceneHandle **out) {
  return guarded([&] { return scene_upload_impl(scene, device, out); });
}
int rt_hip_render_resident(RtSceneHandle *handle, const RtConfig *cfg, RtImage *out, RtStats *stats) {
  return guarded([&] { return render_resident_impl(handle, cfg, out, stats); });
}
int rt_hip_render(const RtScene *scene, const RtConfig *cfg, RtImage *out, RtStats *stats) {
  return guarded([&] {
    RtSceneHandle *h = nullptr;
    int rc = scene_upload_impl(scene, 0, &h);
    if (rc != RT_OK) return rc;
    rc = render_resident_impl(h, cfg, out, stats);
    std::string keep = g_error;
    rt_hip_scene_free(h);
    g_error = keep;
    return rc;
  });
}
int rt_hip_primary_hits(RtSceneHandle *handle, const RtConfig *cfg, int32_t *prim_id, double *t) {
  return guarded([&] { return primary_hits_impl(handle, cfg, prim_id, t); });
}
int rt_hip_trace_rays(RtSceneHandle *handle, int64_t n, const double *origins, const double *directions, int32_t *prim_id, double *t) {
  return guarded([&] { return trace_rays_impl(handle, n, origins, directions, prim_id, t); });
}
int rt_hip_trace_rays_noalloc(RtSceneHandle *handle, int64_t n, const double *d_origins, const double *d_directions, int32_t *d_prim_id,
                              double *d_t) {
  return guarded([&] { return trace_rays_noalloc_impl(handle, n, d_origins, d_directions, d_prim_id, d_t); });
}
