// RenderIntegrator: the host driver above the C ABI.  Mirrors the reference's
// operator surface (/root/reference/src/render/integrator.rs:38-249):
// new_from_json -> render -> save_to_png, with the Rayon row loop
// (integrator.rs:226-243) replaced by one call into rt_hip.h.
#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../../include/rt_host.h"
#include "scene.hpp"

namespace rt {

class RenderIntegrator {
 public:
  Json scene_json;
  Config config;
  std::string base_dir;
  std::vector<double> pixels, sample_pixels;  // integrator.rs:41-45
  SceneData scene;
  RtScene pod{};
  RtConfig pod_config{};
  bool prepared = false;
  bool device_build = false;  // leave BVH_SAH::build to the GPU (rt_hip_scene_build): prepare() then makes no tree
  RtSceneHandle *resident = nullptr;
  int resident_device = -1;
  // one copy of the scene per entry of the device list of render_devices()
  std::vector<RtSceneHandle *> shard_scenes;
  std::vector<int> shard_devices;

  // integrator.rs:39-61
  RenderIntegrator(Json scene_, Config config_, std::string base_dir_)
      : scene_json(std::move(scene_)), config(config_), base_dir(std::move(base_dir_)) {
    size_t n = (size_t)(config.img_width * config.img_height);
    pixels.assign(n * 3, 0.0);
    sample_pixels.assign(n * 3, 0.0);
    pod_config.img_width = (int32_t)config.img_width;
    pod_config.img_height = (int32_t)config.img_height;
    pod_config.sample_batch_size = (int32_t)config.sample_batch_size;
    pod_config.max_sample_batches = (int32_t)config.max_sample_batches;
    pod_config.min_sample_batches = (int32_t)config.min_sample_batches;
    pod_config.max_depth = (int32_t)config.max_depth;
    pod_config.pixel_radius = config.pixel_radius;
    pod_config.gamma_correction = config.gamma_correction;
    pod_config.filter_b = 1. / 3.;  // integrator.rs:47-52
    pod_config.filter_c = 1. / 3.;
    pod_config.rng_mode = RT_RNG_CTR;
    pod_config.seed = 0;
    pod_config.row_first = 0;
    pod_config.row_stride = 1;
  }

  ~RenderIntegrator() {
    if (resident) rt_hip_scene_free(resident);
    for (RtSceneHandle *s : shard_scenes)
      if (s) rt_hip_scene_free(s);
  }

  // integrator.rs:163-192
  void prepare() {
    if (prepared) return;
    const Json &jc = scene_json.at("camera");
    JsonCamera cam;
    cam.center = {jc.at("camera_center").at("x").as_number(), jc.at("camera_center").at("y").as_number(),
                  jc.at("camera_center").at("z").as_number()};
    cam.look_at = {jc.at("camera_look_at").at("x").as_number(), jc.at("camera_look_at").at("y").as_number(),
                   jc.at("camera_look_at").at("z").as_number()};
    cam.fov_vertical = jc.at("camera_fov_vertical").as_number();
    cam.defocus_angle = jc.at("camera_defocus_angle").as_number();
    cam.focus_dist = jc.at("camera_focus_dist").as_number();
    scene.camera = make_camera(config, cam);
    auto t0 = std::chrono::steady_clock::now();
    expand_elements(scene_json, base_dir, scene);
    if (scene.elements.empty()) throw std::runtime_error("scene has no elements");
    auto t1 = std::chrono::steady_clock::now();
    if (!device_build) build_bvh(scene, config.has_split_method ? config.split_method : SplitMethod::Mid);  // integrator.rs:183-189
    auto t2 = std::chrono::steady_clock::now();
    if (std::getenv("RT_HOST_VERBOSE"))
      std::fprintf(stderr, "rt_host: %zu elements expanded in %.3f s, BVH (%zu nodes, depth %d) built in %.3f s on up to %d threads\n",
                   scene.elements.size(), std::chrono::duration<double>(t1 - t0).count(), scene.nodes.size(), scene.bvh_depth,
                   std::chrono::duration<double>(t2 - t1).count(), host_threads());

    pod.camera = scene.camera;
    pod.n_nodes = (int32_t)scene.nodes.size();
    pod.n_elements = (int32_t)scene.element_refs.size();
    pod.n_triangles = (int32_t)scene.triangles.size();
    pod.n_quads = (int32_t)scene.quads.size();
    pod.n_spheres = (int32_t)scene.spheres.size();
    pod.n_materials = (int32_t)scene.materials.size();
    pod.n_attractors = (int32_t)scene.attractors.size();
    pod.split_method = (config.has_split_method ? config.split_method : SplitMethod::Mid) == SplitMethod::SAH ? RT_SPLIT_SAH : RT_SPLIT_MID;
    pod.nodes = scene.nodes.data();
    pod.elements = scene.element_refs.data();
    pod.triangles = scene.triangles.data();
    pod.quads = scene.quads.data();
    pod.spheres = scene.spheres.data();
    pod.materials = scene.materials.data();
    pod.attractors = scene.attractors.data();
    prepared = true;
  }

  // the scene on a device: the host's tree uploaded, or (device_build) the same tree built there from the primitives
  int make_scene(int device, RtSceneHandle **out) {
    return device_build ? rt_hip_scene_build(&pod, device, out) : rt_hip_scene_upload(&pod, device, out);
  }

  // integrator.rs:161-249
  void render(int device, RtStats *stats) {
    prepare();
    if (resident && resident_device != device) {
      rt_hip_scene_free(resident);
      resident = nullptr;
    }
    if (!resident) {
      if (make_scene(device, &resident) != RT_OK) throw std::runtime_error(rt_hip_last_error());
      resident_device = device;
    }
    RtConfig cfg = pod_config;
    cfg.row_first = 0;
    cfg.row_stride = 1;
    RtImage img{};
    img.pixels = pixels.data();
    img.sample_pixels = sample_pixels.data();
    img.memory = RT_MEM_HOST;
    if (rt_hip_render_resident(resident, &cfg, &img, stats) != RT_OK) throw std::runtime_error(rt_hip_last_error());
  }

  // The same render split over several GPUs of this process: entry r of `devices` renders the rows
  // r, r+n, r+2n, ... of the image (pixels are independent given the scene, integrator.rs:226-243, and
  // interleaving evens out the cost of adaptive sampling) on a thread of its own; the rows are
  // interleaved back into `pixels` / `sample_pixels` on the host.  The image is the one render() makes.
  void render_devices(const int *devices, int n, RtStats *stats) {
    prepare();
    if (n <= 0 || !devices) throw std::runtime_error("empty device list");
    const int H = pod_config.img_height, W = pod_config.img_width;
    if (n > H) n = H;
    std::vector<int> want(devices, devices + n);
    if (want != shard_devices) {
      for (RtSceneHandle *s : shard_scenes)
        if (s) rt_hip_scene_free(s);
      shard_scenes.assign((size_t)n, nullptr);
      shard_devices = want;
    }
    std::vector<std::string> errors((size_t)n);
    std::vector<RtStats> part((size_t)n);
    std::vector<std::vector<double>> px((size_t)n), heat((size_t)n);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    pool.reserve((size_t)n);
    for (int r = 0; r < n; r++) {
      auto shard = [&, r] {
        try {
          if (!shard_scenes[(size_t)r] && make_scene(want[(size_t)r], &shard_scenes[(size_t)r]) != RT_OK) {
            errors[(size_t)r] = rt_hip_last_error();
            return;
          }
          RtConfig cfg = pod_config;
          cfg.row_first = r;
          cfg.row_stride = n;
          size_t rows = (size_t)((H - r + n - 1) / n);
          px[(size_t)r].resize(rows * (size_t)W * 3);
          heat[(size_t)r].resize(rows * (size_t)W * 3);
          RtImage img{};
          img.pixels = px[(size_t)r].data();
          img.sample_pixels = heat[(size_t)r].data();
          img.memory = RT_MEM_HOST;
          if (rt_hip_render_resident(shard_scenes[(size_t)r], &cfg, &img, &part[(size_t)r]) != RT_OK) errors[(size_t)r] = rt_hip_last_error();
        } catch (const std::exception &e) {  // nothing may leave a worker thread
          errors[(size_t)r] = e.what();
        } catch (...) {
          errors[(size_t)r] = "unknown failure";
        }
      };
      try {
        pool.emplace_back(shard);
      } catch (const std::system_error &) {  // no more threads to be had: this shard on the caller's
        shard();
      }
    }
    for (std::thread &t : pool) t.join();
    for (int r = 0; r < n; r++)
      if (!errors[(size_t)r].empty()) throw std::runtime_error("device " + std::to_string(want[(size_t)r]) + ": " + errors[(size_t)r]);
    const size_t row_bytes = (size_t)W * 3 * sizeof(double);
    for (int y = 0; y < H; y++) {
      size_t r = (size_t)(y % n), local = (size_t)(y / n);
      std::memcpy(pixels.data() + (size_t)y * W * 3, px[r].data() + local * W * 3, row_bytes);
      std::memcpy(sample_pixels.data() + (size_t)y * W * 3, heat[r].data() + local * W * 3, row_bytes);
    }
    if (stats) {
      *stats = RtStats{};
      for (const RtStats &p : part) {
        stats->rays += p.rays;
        stats->samples += p.samples;
        stats->rays_traced += p.rays_traced;
        stats->node_visits += p.node_visits;
        stats->leaf_tests += p.leaf_tests;
        stats->box_tests += p.box_tests;
        stats->wave_node_steps += p.wave_node_steps;
        stats->wave_leaf_steps += p.wave_leaf_steps;
        stats->trace_launches += p.trace_launches;
        for (int k = 0; k < RT_MAX_DEPTH_STATS; k++) stats->rays_by_depth[k] += p.rays_by_depth[k];
        // device durations: the slowest shard bounds the render
        stats->trace_kernel_ms = std::max(stats->trace_kernel_ms, p.trace_kernel_ms);
        stats->shade_kernel_ms = std::max(stats->shade_kernel_ms, p.shade_kernel_ms);
        stats->other_kernel_ms = std::max(stats->other_kernel_ms, p.other_kernel_ms);
        stats->batches = std::max(stats->batches, p.batches);
      }
      stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  }

  // ----------------------------------------------------------- PNG output
  static void put_u32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
  }
  static void chunk(std::ofstream &f, const char *tag, const std::vector<uint8_t> &data) {
    std::vector<uint8_t> head;
    put_u32(head, (uint32_t)data.size());
    f.write((const char *)head.data(), 4);
    f.write(tag, 4);
    if (!data.empty()) f.write((const char *)data.data(), (std::streamsize)data.size());
    uLong crc = crc32(0L, (const Bytef *)tag, 4);
    if (!data.empty()) crc = crc32(crc, data.data(), (uInt)data.size());
    std::vector<uint8_t> tail;
    put_u32(tail, (uint32_t)crc);
    f.write((const char *)tail.data(), 4);
  }
  // Color::to_rgb, color.rs:47-54: clamp to [0,1] (NaN-ignoring min/max) then
  // a truncating cast of c*255.
  static uint8_t to_u8(double c) {
    double cl = std::fmax(std::fmin(c, 1.0), 0.0);
    return (uint8_t)(cl * 255.0);
  }
  void write_png(const std::string &path, const std::vector<double> &img) const {
    uint32_t w = (uint32_t)config.img_width, h = (uint32_t)config.img_height;
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot create `" + path + "`");
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    f.write((const char *)sig, 8);
    std::vector<uint8_t> ihdr;
    put_u32(ihdr, w);
    put_u32(ihdr, h);
    ihdr.push_back(8);  // BitDepth::Eight
    ihdr.push_back(2);  // ColorType::Rgb
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    chunk(f, "IHDR", ihdr);
    std::vector<uint8_t> gama;  // ScaledFloat::new(1.0 / 2.2), integrator.rs:105
    put_u32(gama, (uint32_t)std::llround((1.0 / 2.2) * 100000.0));
    chunk(f, "gAMA", gama);
    std::vector<uint8_t> chrm;  // integrator.rs:106-113: white, red, green, blue
    const double xy[8] = {0.31270, 0.32900, 0.64000, 0.33000, 0.30000, 0.60000, 0.15000, 0.06000};
    for (double v : xy) put_u32(chrm, (uint32_t)std::llround(v * 100000.0));
    chunk(f, "cHRM", chrm);

    std::vector<uint8_t> raw((size_t)h * (1 + (size_t)w * 3));
    size_t o = 0;
    for (uint32_t y = 0; y < h; y++) {
      raw[o++] = 0;  // filter: None
      for (uint32_t x = 0; x < w; x++)
        for (int c = 0; c < 3; c++) raw[o++] = to_u8(img[((size_t)y * w + x) * 3 + (size_t)c]);
    }
    uLongf bound = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(bound);
    if (compress2(z.data(), &bound, raw.data(), (uLong)raw.size(), Z_DEFAULT_COMPRESSION) != Z_OK)
      throw std::runtime_error("zlib compress failed");
    z.resize(bound);
    chunk(f, "IDAT", z);
    chunk(f, "IEND", {});
    if (!f) throw std::runtime_error("write to `" + path + "` failed");
  }
  // integrator.rs:84-156
  void save_to_png(const std::string &png_path, const std::string &samples_path) const {
    write_png(png_path, pixels);
    write_png(samples_path, sample_pixels);
  }
};

}  // namespace rt

// ------------------------------------------------------------------ C ABI
struct RtHostIntegrator {
  std::unique_ptr<rt::RenderIntegrator> impl;
};

static thread_local std::string g_host_error;

const char *rt_host_last_error(void) { return g_host_error.c_str(); }

template <typename F>
static int guarded(F &&f) {
  try {
    g_host_error.clear();
    f();
    return RT_OK;
  } catch (const std::bad_alloc &) {
    g_host_error = "out of memory";
    return RT_ERR_OOM;
  } catch (const std::exception &e) {
    g_host_error = e.what();
    return RT_ERR_BAD_ARG;
  } catch (...) {
    g_host_error = "unknown error";
    return RT_ERR_BAD_ARG;
  }
}

static std::string slurp(const char *path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error(std::string("cannot open `") + path + "`");
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

int rt_host_new_from_strings(const char *scene_json, const char *config_json, const char *base_dir, RtHostIntegrator **out) {
  return guarded([&] {
    if (!scene_json || !config_json || !out) throw std::runtime_error("null argument");
    rt::Json scene = rt::Json::parse(scene_json);
    rt::Config cfg = rt::Config::from_json(rt::Json::parse(config_json));
    if (!(cfg.img_width >= 1) || !(cfg.img_height >= 1) || cfg.img_width * cfg.img_height > 2.0e9)
      throw std::runtime_error("image size out of range");
    auto *h = new RtHostIntegrator;
    h->impl = std::make_unique<rt::RenderIntegrator>(std::move(scene), cfg, base_dir ? base_dir : "");
    *out = h;
  });
}

int rt_host_new_from_json(const char *scene_path, const char *config_path, const char *base_dir, RtHostIntegrator **out) {
  std::string s, c;
  int rc = guarded([&] {
    if (!scene_path || !config_path || !out) throw std::runtime_error("null argument");
    s = slurp(scene_path);
    c = slurp(config_path);
  });
  if (rc != RT_OK) return rc;
  return rt_host_new_from_strings(s.c_str(), c.c_str(), base_dir, out);
}

void rt_host_free(RtHostIntegrator *h) { delete h; }

int rt_host_prepare(RtHostIntegrator *h) {
  return guarded([&] {
    if (!h) throw std::runtime_error("null integrator");
    h->impl->prepare();
  });
}

const RtScene *rt_host_scene(RtHostIntegrator *h) { return (h && h->impl->prepared) ? &h->impl->pod : nullptr; }
RtConfig *rt_host_config(RtHostIntegrator *h) { return h ? &h->impl->pod_config : nullptr; }
int rt_host_bvh_depth(RtHostIntegrator *h) { return (h && h->impl->prepared && !h->impl->device_build) ? h->impl->scene.bvh_depth : -1; }

int rt_host_set_device_build(RtHostIntegrator *h, int on) {
  return guarded([&] {
    if (!h) throw std::runtime_error("null integrator");
    if (h->impl->prepared) throw std::runtime_error("rt_host_set_device_build must be called before rt_host_prepare / rt_host_render");
    h->impl->device_build = on != 0;
  });
}

int rt_host_element_bbox(RtHostIntegrator *h, int element, double *min3, double *max3) {
  return guarded([&] {
    if (!h || !h->impl->prepared || !min3 || !max3) throw std::runtime_error("integrator not prepared");
    if (element < 0 || (size_t)element >= h->impl->scene.elements.size()) throw std::runtime_error("element index out of range");
    const rt::Aabb &b = h->impl->scene.elements[(size_t)element].bbox;
    min3[0] = b.min.x; min3[1] = b.min.y; min3[2] = b.min.z;
    max3[0] = b.max.x; max3[1] = b.max.y; max3[2] = b.max.z;
  });
}

int rt_host_render(RtHostIntegrator *h, int device, RtStats *stats) {
  return guarded([&] {
    if (!h) throw std::runtime_error("null integrator");
    h->impl->render(device, stats);
  });
}

int rt_host_render_devices(RtHostIntegrator *h, const int *devices, int n_devices, RtStats *stats) {
  return guarded([&] {
    if (!h) throw std::runtime_error("null integrator");
    h->impl->render_devices(devices, n_devices, stats);
  });
}

const double *rt_host_pixels(RtHostIntegrator *h) { return h ? h->impl->pixels.data() : nullptr; }
const double *rt_host_sample_pixels(RtHostIntegrator *h) { return h ? h->impl->sample_pixels.data() : nullptr; }

int rt_host_set_pixels(RtHostIntegrator *h, const double *pixels, const double *sample_pixels) {
  return guarded([&] {
    if (!h) throw std::runtime_error("null integrator");
    if (pixels) std::memcpy(h->impl->pixels.data(), pixels, h->impl->pixels.size() * sizeof(double));
    if (sample_pixels) std::memcpy(h->impl->sample_pixels.data(), sample_pixels, h->impl->sample_pixels.size() * sizeof(double));
  });
}

int rt_host_save_to_png(RtHostIntegrator *h, const char *png_path, const char *samples_path) {
  return guarded([&] {
    if (!h || !png_path) throw std::runtime_error("null argument");
    h->impl->save_to_png(png_path, samples_path ? samples_path : "samples.png");
  });
}
