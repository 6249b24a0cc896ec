/*
 * rt_hip.h -- C ABI of the MI355X per-pixel Monte-Carlo rendering path.
 *
 * This is the drop-in boundary for the reference's row fan-out
 *   /root/reference/src/render/integrator.rs:226-243
 *   (bands.into_par_iter()...for_each(render_line))
 * i.e. everything `RenderIntegrator::render` does AFTER it has built the
 * camera, the element list, the BVH and the attractor list
 * (integrator.rs:161-206).  The host keeps scene loading and the BVH build;
 * the library takes plain-old-data copies of those structures and fills the
 * two pixel buffers `render_line` fills (integrator.rs:331,337).
 *
 * Plain C, no C++ types, no exceptions/unwinding across the boundary.  All
 * floating point on the ABI is f64, as in the reference (vec3.rs:11-15,
 * color.rs:16-20).  Every function returns RT_OK (0) or a negative RT_ERR_*
 * code; the message for the last error on the calling thread is available
 * from rt_hip_last_error().
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status */
#define RT_OK 0
#define RT_ERR_BAD_ARG (-1)     /* NULL pointer, inconsistent sizes, bad index */
#define RT_ERR_HIP (-2)         /* a HIP runtime call failed */
#define RT_ERR_OOM (-3)         /* device or host allocation failed */
#define RT_ERR_UNSUPPORTED (-4) /* valid input the device path cannot run */

/* ------------------------------------------------------------ enum tags */
/* Element enum order, elements/mod.rs:59-63 */
#define RT_PRIM_SPHERE 0
#define RT_PRIM_QUAD 1
#define RT_PRIM_TRIANGLE 2

/* Material enum order, materials/mod.rs:23-28 */
#define RT_MAT_LAMBERTIAN 0
#define RT_MAT_METAL 1
#define RT_MAT_DIELECTRIC 2
#define RT_MAT_DIFFUSE_LIGHT 3

/* Random-number modes.
 * RT_RNG_REF : one xoshiro256++ per image row seeded with the row index and
 *              consumed sequentially across the row (integrator.rs:263).
 *              Inherently serial per row: CPU oracle only.
 * RT_RNG_CTR : one counter-based stream per path keyed by
 *              (pixel, batch, sample); the order of draws inside a path is
 *              the reference's.  Identical on the CPU oracle and the GPU, and
 *              independent of tiling / GPU count. */
#define RT_RNG_REF 0
#define RT_RNG_CTR 1

/* RtConfig.flags (device path; none of them changes a pixel value).
 * RT_FLAG_TIME_KERNELS    bracket every kernel launch with events and report
 *                         the device durations in RtStats.
 * RT_FLAG_COUNT_VISITS    count BVH node visits / leaf tests in RtStats.
 * RT_FLAG_EARLY_TERMINATE stop a path as soon as its contribution is exactly
 *                         zero (scattering pdf 0, NaN pdf: integrator.rs:413-419
 *                         zeroes those anyway).  The reference still traces
 *                         such sub-paths, so ray counts drop below its count. */
#define RT_FLAG_TIME_KERNELS 1
#define RT_FLAG_COUNT_VISITS 2
#define RT_FLAG_EARLY_TERMINATE 4
/* RT_FLAG_REFERENCE_VISIT_RULE  The reference enters the right child of a
 *                         BVH node WITHOUT testing its box after a hit in the
 *                         left subtree (bvh_node_sah.rs:215-216).  A child
 *                         whose box the ray misses can only hold a hit if a
 *                         primitive sticks out of its box (the reference's box
 *                         for a quad that is not axis-aligned, quad.rs:86), so
 *                         the device skips such children when every box of the
 *                         uploaded scene bounds its contents and follows the
 *                         reference's rule otherwise (for every ray, or, when
 *                         only a few primitives stick out of their leaf boxes,
 *                         for the rays that pass the true bounds of one of
 *                         them).  This flag forces the reference's rule for
 *                         every ray (same pixels, more node visits). */
#define RT_FLAG_REFERENCE_VISIT_RULE 8
/* RT_FLAG_REUSE_PRIMARY_HITS  The reference restarts the Sobol index with every
 *                         batch (integrator.rs:284-289), so without a lens
 *                         (camera_defocus_angle <= 0, camera.rs:104-105) sample i
 *                         of a pixel sends the SAME primary ray into the scene
 *                         in every batch, and the reference traces it again
 *                         each time.  With this flag the device traces each
 *                         such ray once per round of batches and lets the
 *                         batches share its hit; what follows the first hit
 *                         (random streams per batch) is unchanged.  Pixels,
 *                         sample counts and RtStats.rays (the queries the
 *                         reference makes) are identical to a render without
 *                         the flag; RtStats.rays_traced and rays_by_depth[0]
 *                         report the queries really executed, which are fewer.
 *                         Ignored for a camera with a lens. */
#define RT_FLAG_REUSE_PRIMARY_HITS 16

/* RtImage.memory */
#define RT_MEM_HOST 0
#define RT_MEM_DEVICE 1

/* RtScene.split_method: BVHSplitMethod (bvh_node_sah.rs:11-19) */
#define RT_SPLIT_SAH 0
#define RT_SPLIT_MID 1

/* ------------------------------------------------------------------ PODs */
typedef struct RtVec3 {
  double x, y, z;
} RtVec3;

/* The Camera fields get_prime_ray reads (camera.rs:16-30, 95-120). */
typedef struct RtCamera {
  RtVec3 camera_center;
  RtVec3 pixel00_loc;
  RtVec3 pixel_delta_u;
  RtVec3 pixel_delta_v;
  RtVec3 defocus_disk_u;
  RtVec3 defocus_disk_v;
  double defocus_angle;
} RtCamera;

/* Material (materials/mod.rs:23-28): albedo for Lambertian / Metal /
 * DiffuseLight; param = fuzz (Metal, metal.rs:10-13) or index_of_refraction
 * (Dielectric, dielectric.rs:9-13). */
typedef struct RtMaterial {
  int32_t kind;
  int32_t reserved;
  double albedo[3];
  double param;
} RtMaterial;

/* Triangle with the fields Triangle::new precomputes (triangle.rs:24-33,
 * 55-70).  The bounding box is not needed: boxes live in the nodes. */
typedef struct RtTriangle {
  RtVec3 v0, v0v1, v0v2, normal;
  int32_t material;
  int32_t reserved;
} RtTriangle;

/* Quad with the fields Quad::new precomputes (quad.rs:41-55, 69-89). */
typedef struct RtQuad {
  RtVec3 q, u, v, normal, w;
  double d, area;
  int32_t material;
  int32_t attractor; /* 1 iff attractor == Some(true) (quad.rs:102-104) */
} RtQuad;

/* Sphere (sphere.rs:51-57). */
typedef struct RtSphere {
  RtVec3 center;
  double radius;
  int32_t material;
  int32_t attractor;
} RtSphere;

/* objects[i] of the reference's Vec<Element> (integrator.rs:167-177):
 * which per-kind array holds it and where.  The element index i is the
 * "primitive id" reported by rt_hip_primary_hits. */
typedef struct RtElementRef {
  int32_t kind;
  int32_t index;
} RtElementRef;

/* One BVHNode_SAH in the reference's preorder vector
 * (bvh_node_sah.rs:102-132, 137-143): either a Leaf{element_index} or a
 * Node{child_l_index, child_l_aabb, child_r_index, child_r_aabb}.  Only the
 * min/max corners of each Aabb are read on the hot path (aabb.rs:187-204). */
typedef struct RtNode {
  int32_t is_leaf;
  int32_t element_index; /* leaf only */
  int32_t child_l_index; /* node only */
  int32_t child_r_index; /* node only */
  double child_l_min[3], child_l_max[3];
  double child_r_min[3], child_r_max[3];
} RtNode;

typedef struct RtScene {
  RtCamera camera;
  int32_t n_nodes;
  int32_t n_elements;
  int32_t n_triangles;
  int32_t n_quads;
  int32_t n_spheres;
  int32_t n_materials;
  int32_t n_attractors;
  int32_t split_method; /* RT_SPLIT_*: Config.bvh_split_method (config.rs:18), the rule the tree in `nodes` was built by;
                           read by rt_hip_scene_build only, which builds the tree by the same rule */
  const RtNode *nodes;          /* nodes[0] is the root (bvh_node_sah.rs:164) */
  const RtElementRef *elements; /* Vec<Element> order */
  const RtTriangle *triangles;
  const RtQuad *quads;
  const RtSphere *spheres;
  const RtMaterial *materials;
  const int32_t *attractors; /* element indices, objects order (integrator.rs:192) */
} RtScene;

/* Config fields render_line reads (config.rs:8-20, integrator.rs:253-339)
 * plus what the reference fixes in RenderIntegrator::new (filter B, C:
 * integrator.rs:47-52) and the row shard this call renders. */
typedef struct RtConfig {
  int32_t img_width;  /* config.img_width as usize */
  int32_t img_height; /* config.img_height as usize */
  int32_t sample_batch_size;
  int32_t max_sample_batches;
  int32_t min_sample_batches;
  int32_t max_depth;
  double pixel_radius;
  double gamma_correction;
  double filter_b; /* 1/3 in the reference */
  double filter_c; /* 1/3 in the reference */
  int32_t rng_mode; /* RT_RNG_REF | RT_RNG_CTR */
  int32_t flags;    /* RT_FLAG_* bits, 0 for the reference's behaviour */
  uint64_t seed; /* extra key for RT_RNG_CTR; ignored by RT_RNG_REF */
  /* Row shard.  The image is cut into blocks of row_block consecutive rows
   * (0 or 1: single rows) and this call renders the blocks
   * b = row_first + j*row_stride (j = 0,1,...) with b*row_block < img_height;
   * (0,1) is the whole image.  Output buffers hold only those rows: the k-th
   * row of the shard, at offset k*img_width, is image row
   *   (row_first + (k / row_block) * row_stride) * row_block + k % row_block. */
  int32_t row_first;
  int32_t row_stride;
  int32_t row_block;
  /* Device working memory this render may take for its path queues, hit buffers and per-path results, in MiB
   * (no counterpart in the reference, whose working state is the call stack).  0: up to half of the device
   * memory that is free when the render starts, at most 2^29 paths in flight (about 125 GB).  A smaller
   * budget means more, smaller rounds -- the image does not change.  The handle keeps one working block,
   * sized by its largest render, until rt_hip_scene_free. */
  int32_t work_mib;
} RtConfig;

/* Output buffers, caller-allocated, library-filled.  n_rows = number of rows
 * in the shard.  Any pointer may be NULL (that output is skipped) except
 * `pixels`.  With memory == RT_MEM_DEVICE the pointers are device pointers
 * on the device the scene was uploaded to. */
typedef struct RtImage {
  double *pixels;        /* n_rows*width*3: clamped, gamma-encoded (integrator.rs:331) */
  double *sample_pixels; /* n_rows*width*3: heat-map (factor,1-factor,0) (integrator.rs:333-337) */
  double *linear;        /* n_rows*width*3: stats.mean() before clamp/gamma (not in the reference) */
  uint32_t *samples;     /* n_rows*width: actual_samples per pixel (integrator.rs:315) */
  uint8_t *rgb8;         /* n_rows*width*3: `pixels` as Color::to_rgb makes them for the PNG (color.rs:47-54:
                            clamp, truncating cast of c*255), converted on the device; optional */
  uint8_t *sample_rgb8;  /* n_rows*width*3: `sample_pixels` likewise (integrator.rs:119-153); optional */
  int32_t memory;        /* RT_MEM_HOST | RT_MEM_DEVICE */
  int32_t reserved;
} RtImage;

#define RT_MAX_DEPTH_STATS 16

/* Counters the reference lacks (it has no ray counter).  A "ray" is one BVH
 * closest-hit query, i.e. one call of BVH_SAH::hit at integrator.rs:359. */
typedef struct RtStats {
  uint64_t rays;        /* BVH closest-hit queries of the samples that count (what the reference would trace) */
  uint64_t rays_traced; /* queries actually executed: the device also traces batches speculatively
                           beyond a pixel's stopping batch and drops them (oracle: == rays) */
  uint64_t rays_by_depth[RT_MAX_DEPTH_STATS]; /* executed queries per bounce number, 0 = primary */
  uint64_t samples;     /* primary samples (paths) that count */
  uint64_t batches;     /* device: rounds executed (a round traces several batches); oracle: 0 */
  uint64_t node_visits;  /* internal nodes entered (oracle; device if instrumented) */
  uint64_t box_tests;    /* Aabb::hit calls */
  uint64_t leaf_tests;   /* Element::hit calls from BVH leaves */
  double wall_ms;        /* call wall time, scene already resident */
  /* device durations, filled when RT_FLAG_TIME_KERNELS is set */
  double trace_kernel_ms;    /* sum over launches of the closest-hit kernel (k_extend) */
  uint64_t trace_launches;   /* number of launches of that kernel */
  double shade_kernel_ms;    /* sum over launches of the shading kernel */
  double other_kernel_ms;    /* generate + accumulate/z-test kernels */
  double trace_ms_by_depth[RT_MAX_DEPTH_STATS]; /* k_extend time per bounce number */
  /* device, with RT_FLAG_COUNT_VISITS: how often a 64-lane wave executed a node step / a leaf test
   * (64 * wave_node_steps / node_visits = average lanes idle-or-busy ratio of the SIMD) */
  uint64_t wave_node_steps;
  uint64_t wave_leaf_steps;
  /* device, with RT_FLAG_COUNT_VISITS: shader clock cycles the waves of the closest-hit kernel spent, summed over
   * waves, claiming and loading rays / in node steps / in leaf tests and storing results (waiting included) */
  uint64_t wave_cycles_refill;
  uint64_t wave_cycles_node;
  uint64_t wave_cycles_leaf;
  /* device, with RT_FLAG_COUNT_VISITS: on trees too high for every lane's traversal stack to stay in on-chip memory, how
   * often a block of stack entries was moved out to device memory or fetched back (0 on lower trees) */
  uint64_t stack_block_moves;
} RtStats;

typedef struct RtSceneHandle RtSceneHandle;

/* ------------------------------------------------------------- functions */

/* Message for the last error on this thread ("" if none). */
const char *rt_hip_last_error(void);

/* Number of visible HIP devices, or a negative RT_ERR_* code. */
int rt_hip_device_count(void);

/* Copy a scene to `device` (validates every index first).  The host arrays
 * are not referenced after return.  Criterion-style repeated renders
 * (benches/default_benchmark.rs:119-120) reuse the handle. */
int rt_hip_scene_upload(const RtScene *scene, int device, RtSceneHandle **out);
int rt_hip_scene_free(RtSceneHandle *handle);

/* Replaces the Rayon loop integrator.rs:226-243 for the rows of the shard:
 * fills out->pixels / sample_pixels exactly as render_line does for each of
 * those rows (RT_RNG_CTR streams).  Blocking; one call in flight per handle. */
int rt_hip_render_resident(RtSceneHandle *handle, const RtConfig *cfg, RtImage *out,
                           RtStats *stats);

/* Progress of the renders on this handle, where the reference advances its
 * progress bar once per finished row (pb.inc(1), integrator.rs:242).  The
 * device finishes pixels in rounds of sample batches, not in rows, so the
 * callback gets the share of the shard's sample batches that are done or no
 * longer needed (0..1, non-decreasing, 1.0 last), after every round, on the
 * thread that called rt_hip_render_resident.  It must return quickly: the GPU
 * idles meanwhile.  fn = NULL removes it. */
typedef void (*RtProgressFn)(void *user, double fraction_done);
int rt_hip_set_progress(RtSceneHandle *handle, RtProgressFn fn, void *user);

/* One-shot convenience: upload to device 0, render, free. */
int rt_hip_render(const RtScene *scene, const RtConfig *cfg, RtImage *out, RtStats *stats);

/* Parity gate: for every pixel of the shard and every sample i of batch 0,
 * the element index hit by the primary ray (camera.rs:95-120 ->
 * bvh_node_sah.rs:162-165) or -1, and its t (0 if no hit).  Arrays hold
 * n_rows*width*sample_batch_size entries, sample index fastest.  Host
 * pointers. */
int rt_hip_primary_hits(RtSceneHandle *handle, const RtConfig *cfg, int32_t *prim_id, double *t);

/* Closest-hit queries for caller-supplied rays (origin, un-normalised
 * direction; Ray::new normalises, ray.rs:35-48) with the interval
 * (0.0001, f64::MAX) of integrator.rs:359.  Host pointers, n rays. */
int rt_hip_trace_rays(RtSceneHandle *handle, int64_t n, const double *origins,
                      const double *directions, int32_t *prim_id, double *t);

/* Opt-in: build the BVH on the device instead of uploading the host's.  The
 * reference builds its tree in BVH_SAH::build (bvh_node_sah.rs:147-159,
 * 256-461: midpoint or 16-bucket surface-area split, a recursion on one host
 * thread).  This entry point takes the primitives alone (scene->nodes and
 * n_nodes are ignored) and runs the same top-down build on the GPU, one level
 * of the tree per pass, by the rule scene->split_method names: the tree has
 * the reference's topology and the reference's depth-first leaf order
 * (element boxes are recomputed from the primitive records, so a centroid can
 * differ from the host's in its last bit: an element moves to another bucket
 * only if it lies on a bucket boundary to within that bit).  Everything else
 * is as after rt_hip_scene_upload, except that the boxes stored are the
 * primitives' TRUE bounds: the reference boxes a quad by two corners only
 * (quad.rs:86) and can lose hits on a tilted quad's overhang; this tree finds
 * them.  Hit distances are those of the reference's primitive tests either
 * way.  RT_HIP_BUILD=0 / 1 selects, for comparison, a linear BVH over Morton
 * codes / a tree by locally-ordered clustering instead (ties between
 * primitives hit at exactly the same t then follow that tree's leaf order). */
int rt_hip_scene_build(const RtScene *scene, int device, RtSceneHandle **out);

/* ---- one image, several processes (one per GPU).  The reference deals the rows of the image to the threads of one
 * process (integrator.rs:226-243, Rayon); here rank r of `world` renders the row blocks r, r + world, ... on its own GPU
 * and rank 0 receives every rank's rows over RCCL (xGMI between the GPUs of a node) and assembles the image on its
 * device.  Pixels are independent given the scene, so this one exchange is all the traffic there is.
 *   rt_hip_comm_id      rank 0 makes an id (ncclGetUniqueId) and hands a copy to every rank by the caller's own means
 *                       (a file, an environment variable, MPI, a socket);
 *   rt_hip_comm_init    every rank, with the same id: collective (ncclCommInitRank), binds the communicator to `device`;
 *   rt_hip_render_ranks every rank, with its own handle of the same scene on that device and the same cfg (its
 *                       row_first / row_stride are set by the call; row_block is kept): collective.  On rank 0 `out`
 *                       receives the WHOLE image (pixels and, if not NULL, sample_pixels; host or device pointers by
 *                       out->memory; the other outputs must be NULL); on the other ranks `out` may be NULL.  stats:
 *                       rays, samples, rays_traced and rays_by_depth are the job's totals on every rank, the rest is
 *                       the calling rank's.
 * With id == NULL (and rank 0) rt_hip_comm_init makes a communicator that needs no RCCL and no second process: the one
 * caller of rt_hip_render_ranks renders the shards of all `world` ranks in turn on its device and assembles them the same
 * way -- for tests, and for trying a sharding out on one GPU.
 * The calls are collective and blocking: a rank that returns an error before the exchange (bad arguments, out of
 * memory) leaves the others waiting for it, so a caller that sees an error tears the job down, as with any RCCL program.
 * RCCL is loaded (dlopen) when the first id or communicator is made; a process that never calls these never loads it. */
typedef struct RtComm RtComm;
#define RT_COMM_ID_BYTES 128
int rt_hip_comm_id(unsigned char id[RT_COMM_ID_BYTES]);
int rt_hip_comm_init(const unsigned char id[RT_COMM_ID_BYTES], int rank, int world, int device, RtComm **out);
int rt_hip_comm_free(RtComm *comm);
int rt_hip_render_ranks(RtSceneHandle *handle, RtComm *comm, const RtConfig *cfg, RtImage *out, RtStats *stats);

/* Debug / test aid: the tree a handle holds, as RtNode records (2*k+1 nodes
 * for k internal nodes: the root first, every leaf after the internal nodes;
 * child boxes are the f32 boxes the device tests).  With nodes == NULL only
 * *n_nodes is written.  Works for uploaded and device-built scenes. */
int rt_hip_scene_download_tree(RtSceneHandle *handle, RtNode *nodes, int32_t capacity, int32_t *n_nodes);

/* Tuning aid; no counterpart in the reference.  A handle's launch and scheduling parameters are read from the
 * environment once, when rt_hip_scene_upload / rt_hip_scene_build make the handle (RT_HIP_REFILL, RT_HIP_CLAIM, ...:
 * the variable of parameter `x` is RT_HIP_X); no render or query reads the environment afterwards.  This call changes
 * one of them on an existing handle, for sweeps and tests:
 *   refill, node_min, leafq, claim, tail     closest-hit kernel: idle lanes before a refill, lanes with node work below
 *                                            which queued leaf tests run, leaves queued per lane, rays per claim, shrinking
 *                                            last claims
 *   wide_waves, ovf_waves                    waves per SIMD of the walks on wide nodes / with the block-overflow stack
 *   shade_blocks                             most workgroups of a shading launch
 *   max_paths, max_group, first_group        paths in flight per round (0: by memory), most batches per round, jump
 *                                            after a quiet first round
 *   packet                                   1 / 0: bounce 0 by the packet walk (64 neighbouring primary rays walk the
 *                                            tree together) or by the general one
 *   tile_hlog, verbose
 * None of them changes a pixel, a hit or a count.  Parameters that shape the device copy of the scene (wide, tri_pad,
 * spill_above, no_grid, node order) are fixed when the handle is made and are refused here, as are unknown names and
 * values out of range (RT_ERR_BAD_ARG). */
int rt_hip_scene_tune(RtSceneHandle *handle, const char *name, int64_t value);

/* The queries of rt_hip_trace_rays for callers that cannot allocate device
 * memory (tight serving loops, harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference.  The call makes no hipMalloc / hipFree and uses no allocating
 * library routine: the ray queue, hit buffer and counters live in a fixed
 * array inside the library's code object (about 7.5 MiB per GPU, held until
 * the process ends).
 *   - d_origins, d_directions (3*n doubles each), d_prim_id, d_t (n each) are
 *     DEVICE pointers on the handle's device.
 *   - n <= RT_NOALLOC_MAX_RAYS, otherwise RT_ERR_UNSUPPORTED (split the batch).
 *   - Stream contract: the call is SYNCHRONOUS and runs on a stream of the
 *     library's own.  The inputs must be complete before the call -- work that
 *     produces them on a stream of the caller's is not waited for: synchronise
 *     that stream (or an event recorded on it) first -- and the results are
 *     complete when it returns.  It takes a host lock and ends with a stream
 *     synchronisation, so it cannot itself be recorded into a graph capture.
 *   - All calls on one GPU share that GPU's array, which keeps the previous
 *     call's contents (the library clears what must start at zero).  Within a
 *     process the library lets one call at a time through per GPU, so calls
 *     from two threads or on two handles wait for each other instead of
 *     racing; rt_hip_scene_upload / rt_hip_scene_build trace one ray through
 *     the same path at set-up (under the same lock), where whatever the
 *     runtime prepares at a kernel's first launch may allocate. */
#define RT_NOALLOC_MAX_RAYS 131072
int rt_hip_trace_rays_noalloc(RtSceneHandle *handle, int64_t n, const double *d_origins,
                              const double *d_directions, int32_t *d_prim_id, double *d_t);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
