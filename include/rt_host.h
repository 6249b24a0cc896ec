/*
 * rt_host.h -- C ABI of the host side that sits ABOVE rt_hip.h.
 *
 * The reference's host is Rust and keeps JSON scene/config loading, element
 * expansion, the BVH build and the render entry point.  No Rust toolchain is
 * available to this build, so that host is restated in C++ and exported with
 * the same operator surface as the reference's RenderIntegrator
 * (/root/reference/src/render/integrator.rs:38-249):
 *
 *   RenderIntegrator::new_from_json(scene, config)  -> rt_host_new_from_json
 *   RenderIntegrator::render()                      -> rt_host_render
 *   RenderIntegrator::save_to_png(path)             -> rt_host_save_to_png
 *
 * rt_host_prepare() performs steps 1-4 of render() (integrator.rs:163-192:
 * Camera::new, JSON -> Vec<Element>, BVH_SAH::build, attractor list) and
 * exposes the result as the RtScene POD that rt_hip.h consumes; that POD is
 * exactly what a Rust maintainer's FFI shim would assemble from its own
 * structures.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RtHostIntegrator RtHostIntegrator;

const char *rt_host_last_error(void);

/* integrator.rs:63-82.  Paths inside the scene ("filepath" of JSONObj,
 * obj.rs:31) are opened relative to the process working directory, as in the
 * reference, unless `base_dir` is non-NULL, in which case relative paths are
 * resolved against it. */
int rt_host_new_from_json(const char *scene_path, const char *config_path, const char *base_dir,
                          RtHostIntegrator **out);
/* Same, from in-memory JSON text. */
int rt_host_new_from_strings(const char *scene_json, const char *config_json, const char *base_dir,
                             RtHostIntegrator **out);
void rt_host_free(RtHostIntegrator *h);

/* integrator.rs:163-192.  Idempotent. */
int rt_host_prepare(RtHostIntegrator *h);

/* Opt-in, before rt_host_prepare: leave BVH_SAH::build (integrator.rs:183-189) to the GPU.  rt_host_prepare then
 * makes no tree (RtScene.nodes is NULL, n_nodes 0; rt_host_bvh_depth returns -1) and rt_host_render /
 * rt_host_render_devices set their scenes up with rt_hip_scene_build, which runs the same top-down build by the
 * same split rule on the device (2 M triangles: 10 ms instead of 0.24 s of build and 0.25 s of upload). */
int rt_host_set_device_build(RtHostIntegrator *h, int on);

/* Valid after rt_host_prepare; owned by the integrator. */
const RtScene *rt_host_scene(RtHostIntegrator *h);
/* The Config as the RtConfig POD (row shard = whole image, RT_RNG_CTR).
 * The caller may edit it before rt_host_render. */
RtConfig *rt_host_config(RtHostIntegrator *h);
/* Depth of the BVH (root = 0) and the element's bounding box as built by the
 * element constructors (min xyz, max xyz), for tests. */
int rt_host_bvh_depth(RtHostIntegrator *h);
int rt_host_element_bbox(RtHostIntegrator *h, int element, double *min3, double *max3);

/* integrator.rs:161-249 with the Rayon loop replaced by rt_hip_render on
 * `device`.  Fills the integrator's `pixels` and `sample_pixels`. */
int rt_host_render(RtHostIntegrator *h, int device, RtStats *stats);
/* The same render on several GPUs of this process: entry r of `devices` gets
 * the rows r, r+n, r+2n, ... (one scene copy and one host thread per entry;
 * the reference's rows are independent tasks, integrator.rs:226-243), and the
 * rows are interleaved back into `pixels` / `sample_pixels`.  The image equals
 * rt_host_render's bit for bit.  Stats: counts are summed, durations are those
 * of the slowest shard. */
int rt_host_render_devices(RtHostIntegrator *h, const int *devices, int n_devices, RtStats *stats);
const double *rt_host_pixels(RtHostIntegrator *h);        /* width*height*3 */
const double *rt_host_sample_pixels(RtHostIntegrator *h); /* width*height*3 */
/* Overwrite the pixel buffers (e.g. with tiles gathered from several GPUs). */
int rt_host_set_pixels(RtHostIntegrator *h, const double *pixels, const double *sample_pixels);

/* integrator.rs:84-156: writes `png_path` (image) and, next to the working
 * directory like the reference, `samples_path` (heat-map; "samples.png" if
 * NULL): 8-bit RGB via Color::to_rgb (color.rs:47-54), gAMA 1/2.2 and the
 * reference's cHRM chunk. */
int rt_host_save_to_png(RtHostIntegrator *h, const char *png_path, const char *samples_path);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
