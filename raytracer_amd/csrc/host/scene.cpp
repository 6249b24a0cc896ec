// Scene ingestion on the host: Config / camera / element expansion.
// Restates the load-time half of the reference (config.rs, camera.rs:35-91,
// elements/{triangle,quad,sphere,cube,obj}.rs, linalg/mat4.rs transforms) so
// that the POD scene handed to rt_hip.h is the one the Rust host would build.
#include "scene.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <exception>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <system_error>
#include <thread>

namespace rt {

// ------------------------------------------------------------------ Aabb
static Vec3 centroid_of(Vec3 mn, Vec3 mx) {  // aabb.rs:214-220
  return {mn.x + (mx.x - mn.x) / 2., mn.y + (mx.y - mn.y) / 2., mn.z + (mx.z - mn.z) / 2.};
}

Aabb Aabb::from_points(Vec3 p, Vec3 q) {
  Aabb b;
  b.min = {std::fmin(q.x, p.x), std::fmin(q.y, p.y), std::fmin(q.z, p.z)};
  b.max = {std::fmax(q.x, p.x), std::fmax(q.y, p.y), std::fmax(q.z, p.z)};
  b.centroid = centroid_of(b.min, b.max);
  return b;
}

Aabb Aabb::merge(const Aabb &a, const Aabb &b) {  // interval.rs:15-20 per axis
  Aabb r;
  r.min = vmin(a.min, b.min);
  r.max = vmax(a.max, b.max);
  r.centroid = centroid_of(r.min, r.max);
  return r;
}

Aabb Aabb::pad() const {
  const double delta = 0.0001;
  Aabb r;
  double mn[3] = {min.x, min.y, min.z}, mx[3] = {max.x, max.y, max.z};
  for (int a = 0; a < 3; a++) {
    if (!(mx[a] - mn[a] >= delta)) {  // interval.rs:26-29: expand by delta/2 each side
      double padding = delta / 2.0;
      mn[a] = mn[a] - padding;
      mx[a] = mx[a] + padding;
    }
  }
  r.min = {mn[0], mn[1], mn[2]};
  r.max = {mx[0], mx[1], mx[2]};
  r.centroid = centroid_of(r.min, r.max);
  return r;
}

double Aabb::area() const {
  return (max.x - min.x) * (max.y - min.y) + (max.y - min.y) * (max.z - min.z) + (max.z - min.z) * (max.x - min.x);
}

int Aabb::largest_axis() const {
  Vec3 size = max - min;
  if (size.x > size.y && size.x > size.z) return 0;
  if (size.y > size.z) return 1;
  return 2;
}

// ---------------------------------------------------------------- Config
static Vec3 vec3_from_json(const Json &j) { return {j.at("x").as_number(), j.at("y").as_number(), j.at("z").as_number()}; }

static int64_t usize_from_json(const Json &j, const char *name) {
  double d = j.at(name).as_number();
  if (!(d >= 0) || d != std::floor(d)) throw std::runtime_error(std::string("field `") + name + "` must be a non-negative integer");
  return (int64_t)d;
}

Config Config::from_json(const Json &j) {
  Config c;
  c.img_width = j.at("img_width").as_number();
  c.img_height = j.at("img_height").as_number();
  c.sample_batch_size = usize_from_json(j, "sample_batch_size");
  c.max_sample_batches = usize_from_json(j, "max_sample_batches");
  c.min_sample_batches = usize_from_json(j, "min_sample_batches");
  c.max_depth = usize_from_json(j, "max_depth");
  const Json &sky = j.at("sky_color");
  c.sky_color[0] = sky.at("r").as_number();
  c.sky_color[1] = sky.at("g").as_number();
  c.sky_color[2] = sky.at("b").as_number();
  c.pixel_radius = j.at("pixel_radius").as_number();
  c.gamma_correction = j.at("gamma_correction").as_number();
  const Json *sm = j.find("bvh_split_method");  // Option<BVHSplitMethod>
  if (sm && !sm->is_null()) {
    const std::string &s = sm->as_string();
    c.has_split_method = true;
    if (s == "SAH") c.split_method = SplitMethod::SAH;
    else if (s == "Mid") c.split_method = SplitMethod::Mid;
    else throw std::runtime_error("unknown variant `" + s + "`, expected `Mid` or `SAH`");
  }
  return c;
}

// ---------------------------------------------------------------- Camera
RtCamera make_camera(const Config &cfg, const JsonCamera &jc) {
  const double PI = 3.14159265358979323846;
  Vec3 relative_up{0.0, 1.0, 0.0};
  double ratio = cfg.img_width / cfg.img_height;
  double theta = jc.fov_vertical * (PI / 180.0);  // f64::to_radians
  double h = std::tan(theta / 2.0);
  double vv = 2.0 * h * jc.focus_dist;
  double vu = vv * ratio;
  Vec3 w = normalized(jc.center - jc.look_at);
  Vec3 u = normalized(cross(relative_up, w));
  Vec3 v = cross(w, u);
  Vec3 viewport_u = u * vu;
  Vec3 viewport_v = v * -1.0 * vv;
  Vec3 pixel_delta_u = viewport_u / cfg.img_width;
  Vec3 pixel_delta_v = viewport_v / (cfg.img_width / ratio);
  Vec3 viewport_upper_left = jc.center - (w * jc.focus_dist) - viewport_u / 2.0 - viewport_v / 2.0;
  Vec3 pixel00_loc = viewport_upper_left + (pixel_delta_u + pixel_delta_v) * 0.5;
  double defocus_radius = jc.focus_dist * std::tan((jc.defocus_angle / 2.0) * (PI / 180.0));
  RtCamera c{};
  c.camera_center = abi(jc.center);
  c.pixel00_loc = abi(pixel00_loc);
  c.pixel_delta_u = abi(pixel_delta_u);
  c.pixel_delta_v = abi(pixel_delta_v);
  c.defocus_disk_u = abi(u * defocus_radius);
  c.defocus_disk_v = abi(v * defocus_radius);
  c.defocus_angle = jc.defocus_angle;
  return c;
}

// ------------------------------------------------------- element builders
// Triangle::new, triangle.rs:55-70: the record and the element (with its padded box) for slot `index`.
static void make_triangle(Vec3 v0, Vec3 v1, Vec3 v2, int material, int index, RtTriangle &t, Element &e) {
  Vec3 v0v1 = v1 - v0, v0v2 = v2 - v0;
  Vec3 normal = normalized(cross(v0v1, v0v2));
  t = RtTriangle{};
  t.v0 = abi(v0);
  t.v0v1 = abi(v0v1);
  t.v0v2 = abi(v0v2);
  t.normal = abi(normal);
  t.material = material;
  e.kind = RT_PRIM_TRIANGLE;
  e.index = index;
  e.bbox = Aabb::from_points(vmin(v0, vmin(v1, v2)), vmax(v0, vmax(v1, v2))).pad();
}

void SceneData::add_triangle(Vec3 v0, Vec3 v1, Vec3 v2, int material) {
  RtTriangle t;
  Element e;
  make_triangle(v0, v1, v2, material, (int)triangles.size(), t, e);
  triangles.push_back(t);
  elements.push_back(e);
}

void SceneData::add_quad(Vec3 q, Vec3 u, Vec3 v, int material, bool attractor) {
  Vec3 n = cross(u, v);
  Vec3 normal = normalized(n);
  RtQuad r{};
  r.q = abi(q);
  r.u = abi(u);
  r.v = abi(v);
  r.normal = abi(normal);
  r.d = dot(normal, q);
  r.w = abi(n / dot(n, n));
  r.area = length(n);
  r.material = material;
  r.attractor = attractor ? 1 : 0;
  Element e;
  e.kind = RT_PRIM_QUAD;
  e.index = (int)quads.size();
  e.bbox = Aabb::from_points(q, q + u + v).pad();  // exact only for axis-aligned quads, as in the reference
  quads.push_back(r);
  elements.push_back(e);
}

void SceneData::add_sphere(Vec3 center, double radius, int material, bool attractor) {
  RtSphere s{};
  s.center = abi(center);
  s.radius = radius;
  s.material = material;
  s.attractor = attractor ? 1 : 0;
  Vec3 rvec{radius, radius, radius};
  Element e;
  e.kind = RT_PRIM_SPHERE;
  e.index = (int)spheres.size();
  e.bbox = Aabb::from_points(center - rvec, center + rvec);
  spheres.push_back(s);
  elements.push_back(e);
}

// --------------------------------------------------------------- Mat4
// linalg/mat4.rs:28-99 (constructors) and 281-304 (Mat4 * Vec4 with w = 1;
// only x,y,z of the product are kept, vec4.rs:28-30).
struct Mat4 {
  double m[4][4];
  Vec3 apply(Vec3 v) const {
    const double w = 1.0;
    return {m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z + m[0][3] * w,
            m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z + m[1][3] * w,
            m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z + m[2][3] * w};
  }
};
static Mat4 mat_translate(double dx, double dy, double dz) {  // the reference calls this "transpose"
  return Mat4{{{1., 0., 0., dx}, {0., 1., 0., dy}, {0., 0., 1., dz}, {0., 0., 0., 1.}}};
}
static Mat4 mat_scale(double sx, double sy, double sz) {
  return Mat4{{{sx, 0., 0., 0.}, {0., sy, 0., 0.}, {0., 0., sz, 0.}, {0., 0., 0., 0.}}};
}
static Mat4 mat_rotate_x(double t) {
  return Mat4{{{1., 0., 0., 0.}, {0., std::cos(t), std::sin(t), 0.}, {0., -std::sin(t), std::cos(t), 0.}, {0., 0., 0., 1.}}};
}
static Mat4 mat_rotate_y(double t) {
  return Mat4{{{std::cos(t), 0., -std::sin(t), 0.}, {0., 1., 0., 0.}, {std::sin(t), 0., std::cos(t), 0.}, {0., 0., 0., 1.}}};
}
static Mat4 mat_rotate_z(double t) {
  return Mat4{{{std::cos(t), -std::sin(t), 0., 0.}, {std::sin(t), std::cos(t), 0., 0.}, {0., 0., 1., 0.}, {0., 0., 0., 1.}}};
}
static void apply_all(const Mat4 &m, std::vector<Vec3> &vs) {
  for (auto &v : vs) v = m.apply(v);
}

struct Transform {  // elements/mod.rs:78-97
  bool has_translate = false, has_rotate = false, has_scale = false;
  Vec3 translate, rotate, scale;
};
static Transform transform_from_json(const Json &j) {
  Transform t;
  if (const Json *p = j.find("transpose"); p && !p->is_null()) {
    t.has_translate = true;
    t.translate = vec3_from_json(*p);
  }
  if (const Json *p = j.find("rotate"); p && !p->is_null()) {
    t.has_rotate = true;
    t.rotate = {p->at("theta_x").as_number(), p->at("theta_y").as_number(), p->at("theta_z").as_number()};
  }
  if (const Json *p = j.find("scale"); p && !p->is_null()) {
    t.has_scale = true;
    t.scale = vec3_from_json(*p);
  }
  return t;
}

// materials/mod.rs:22-28 (externally tagged enum)
static int material_from_json(const Json &j, SceneData &out) {
  if (j.kind != Json::Object || j.object.size() != 1) throw std::runtime_error("material: expected a single-variant object");
  const std::string &tag = j.object[0].first;
  const Json &body = j.object[0].second;
  RtMaterial m{};
  auto albedo = [&](const Json &b) {
    const Json &a = b.at("albedo");
    m.albedo[0] = a.at("r").as_number();
    m.albedo[1] = a.at("g").as_number();
    m.albedo[2] = a.at("b").as_number();
  };
  if (tag == "Lambertian") {
    m.kind = RT_MAT_LAMBERTIAN;
    albedo(body);
  } else if (tag == "Metal") {
    m.kind = RT_MAT_METAL;
    albedo(body);
    m.param = body.at("fuzz").as_number();
  } else if (tag == "Dielectric") {
    m.kind = RT_MAT_DIELECTRIC;
    m.param = body.at("index_of_refraction").as_number();
  } else if (tag == "DiffuseLight") {
    m.kind = RT_MAT_DIFFUSE_LIGHT;
    albedo(body);
  } else {
    throw std::runtime_error("unknown material variant `" + tag + "`");
  }
  out.materials.push_back(m);
  return (int)out.materials.size() - 1;
}

static bool option_true(const Json &j, const char *name) {  // Option<bool> == Some(true)
  const Json *p = j.find(name);
  return p && p->kind == Json::Bool && p->boolean;
}

// elements/cube.rs:26-144
static void add_box(const Json &j, SceneData &out) {
  Vec3 a = vec3_from_json(j.at("a")), b = vec3_from_json(j.at("b"));
  int material = material_from_json(j.at("material"), out);
  Transform tf = transform_from_json(j);
  Vec3 mn = vmin(a, b), mx = vmax(a, b);
  double dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
  std::vector<Vec3> v = {{1., -1., 1.},  {1., -1., -1.},  {1., 1., -1.},  {1., 1., 1.},
                         {-1., -1., 1.}, {-1., -1., -1.}, {-1., 1., -1.}, {-1., 1., 1.}};
  static const int tris[12][3] = {{1, 5, 6}, {1, 6, 2}, {5, 7, 6}, {5, 4, 7}, {0, 2, 3}, {0, 1, 2},
                                  {2, 7, 3}, {2, 6, 7}, {0, 7, 4}, {0, 3, 7}, {4, 5, 1}, {1, 0, 4}};
  apply_all(mat_scale(dx / 2., dy / 2., dz / 2.), v);
  if (tf.has_scale) apply_all(mat_scale(tf.scale.x, tf.scale.y, tf.scale.z), v);
  if (tf.has_rotate) {
    apply_all(mat_rotate_x(tf.rotate.x), v);
    apply_all(mat_rotate_y(tf.rotate.y), v);
    apply_all(mat_rotate_z(tf.rotate.z), v);
  }
  apply_all(mat_translate(mn.x + dx / 2.0, mn.y + dy / 2.0, mn.z + dz / 2.0), v);
  if (tf.has_translate) apply_all(mat_translate(tf.translate.x, tf.translate.y, tf.translate.z), v);
  for (const auto &t : tris) out.add_triangle(v[t[0]], v[t[1]], v[t[2]], material);
}

// elements/obj.rs:27-103.  The reference parses with the wavefront_obj crate
// (10.0.0, source not in the reference tree): per-object vertex lists with
// object-local 0-based indices, polygons fan-triangulated, only triangles
// consumed.  Restated here for `v`, `f` (with v/vt/vn forms), `o`.
//
// Large meshes dominate the load time, so the file is read whole and parsed by several threads,
// each on a run of complete lines; what depends on everything before a line (the running vertex
// count for relative indices, object boundaries) is resolved once the per-thread counts are known.
// Vertices are transformed and triangles built in parallel; their order is the file's.
template <typename F>
static void parallel_ranges(size_t n, int threads, size_t min_items, F &&fn) {  // fn(begin, end) over [0, n)
  if (threads <= 1 || n < min_items) {
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

static void add_obj(const Json &j, const std::string &base_dir, SceneData &out) {
  std::string path = j.at("filepath").as_string();
  if (!base_dir.empty() && !path.empty() && path[0] != '/') path = base_dir + "/" + path;
  int material = material_from_json(j.at("material"), out);
  Transform tf = transform_from_json(j);

  std::string text;
  {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open obj file `" + path + "`");
    f.seekg(0, std::ios::end);
    std::streamoff size = f.tellg();
    f.seekg(0, std::ios::beg);
    text.resize(size > 0 ? (size_t)size : 0);
    if (size > 0 && !f.read(&text[0], size)) throw std::runtime_error("cannot read obj file `" + path + "`");
  }

  // ---- parse: each chunk starts after a newline and ends after one
  struct Chunk {
    std::vector<Vec3> vertices;
    std::vector<int64_t> tri;        // 3 per triangle: a global 1-based index, or for a relative
    std::vector<uint8_t> tri_rel;    //   one (bit k set) its value + 1 + vertices seen so far in the chunk
    struct Marker { size_t vertices_before, tris_before; };
    std::vector<Marker> objects;     // `o` lines
  };
  size_t chunk_bytes = 1 << 20;  // least text worth a thread
  if (const char *v = std::getenv("RT_HOST_PARSE_CHUNK"))
    if (std::atoll(v) >= 1) chunk_bytes = (size_t)std::atoll(v);
  const int threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), text.size() / chunk_bytes + 1));
  std::vector<size_t> cut((size_t)threads + 1, text.size());
  cut[0] = 0;
  for (int c = 1; c < threads; c++) {
    size_t at = text.find('\n', text.size() * (size_t)c / (size_t)threads);
    cut[(size_t)c] = at == std::string::npos ? text.size() : at + 1;
  }
  std::vector<Chunk> chunks((size_t)threads);
  parallel_ranges((size_t)threads, threads, 2, [&](size_t c0, size_t c1) {
    for (size_t c = c0; c < c1; c++) {
      Chunk &ch = chunks[c];
      std::vector<int64_t> poly;
      std::vector<uint8_t> poly_rel;
      const char *p = text.data() + cut[c], *stop = text.data() + cut[c + 1];
      while (p < stop) {
        const char *eol = (const char *)memchr(p, '\n', (size_t)(stop - p));
        if (!eol) eol = stop;
        // strtod/strtoll stop at the newline themselves; the last line of the file is NUL-terminated by std::string
        while (p < eol && (*p == ' ' || *p == '\t')) p++;
        if (eol - p >= 2 && p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
          // strtod skips newlines like any blank: a line with fewer than three numbers must not
          // borrow from the next one (missing coordinates read as 0, as they would from the line alone)
          double xyz[3] = {0.0, 0.0, 0.0};
          const char *q = p + 1;
          for (int k = 0; k < 3; k++) {
            while (q < eol && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
            if (q >= eol) break;
            char *end;
            double value = std::strtod(q, &end);
            if (end == q || end > eol) break;
            xyz[k] = value;
            q = end;
          }
          ch.vertices.push_back({xyz[0], xyz[1], xyz[2]});
        } else if (eol - p >= 2 && p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
          poly.clear();
          poly_rel.clear();
          const char *q = p + 1;
          while (q < eol) {
            while (q < eol && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
            if (q >= eol) break;
            char *end;
            long long idx = std::strtoll(q, &end, 10);
            if (end == q) throw std::runtime_error("obj: bad face index in `" + path + "`");
            if (idx < 0) {  // relative to the vertices read so far
              poly.push_back((int64_t)ch.vertices.size() + idx + 1);
              poly_rel.push_back(1);
            } else {
              poly.push_back((int64_t)idx);
              poly_rel.push_back(0);
            }
            q = end;
            while (q < eol && *q != ' ' && *q != '\t' && *q != '\r') q++;  // skip /vt/vn
          }
          for (size_t k = 1; k + 1 < poly.size(); k++) {  // fan
            ch.tri.push_back(poly[0]);
            ch.tri.push_back(poly[k]);
            ch.tri.push_back(poly[k + 1]);
            ch.tri_rel.push_back((uint8_t)(poly_rel[0] | (poly_rel[k] << 1) | (poly_rel[k + 1] << 2)));
          }
        } else if (eol - p >= 2 && p[0] == 'o' && (p[1] == ' ' || p[1] == '\t')) {
          ch.objects.push_back({ch.vertices.size(), ch.tri_rel.size()});
        }
        p = eol < stop ? eol + 1 : stop;
      }
    }
  });

  // ---- merge: global vertex / triangle numbering and the objects' ranges
  std::vector<size_t> vbase((size_t)threads + 1, 0), tbase((size_t)threads + 1, 0);
  for (int c = 0; c < threads; c++) {
    vbase[(size_t)c + 1] = vbase[(size_t)c] + chunks[(size_t)c].vertices.size();
    tbase[(size_t)c + 1] = tbase[(size_t)c] + chunks[(size_t)c].tri_rel.size();
  }
  const size_t n_vertices = vbase[(size_t)threads], n_tris = tbase[(size_t)threads];
  struct Range { size_t first_vertex, first_tri; };
  std::vector<Range> objects{{0, 0}};
  for (int c = 0; c < threads; c++)
    for (const auto &m : chunks[(size_t)c].objects) {
      size_t gv = vbase[(size_t)c] + m.vertices_before, gt = tbase[(size_t)c] + m.tris_before;
      if (gv > objects.back().first_vertex || gt > objects.back().first_tri) objects.push_back({gv, gt});  // a non-empty object ends here
    }
  objects.push_back({n_vertices, n_tris});  // sentinel

  std::vector<Vec3> vertices(n_vertices);
  std::vector<int64_t> tri(3 * n_tris);
  parallel_ranges((size_t)threads, threads, 2, [&](size_t c0, size_t c1) {
    for (size_t c = c0; c < c1; c++) {
      const Chunk &ch = chunks[c];
      std::copy(ch.vertices.begin(), ch.vertices.end(), vertices.begin() + (long)vbase[c]);
      for (size_t t = 0; t < ch.tri_rel.size(); t++)
        for (int k = 0; k < 3; k++) tri[3 * (tbase[c] + t) + (size_t)k] = ch.tri[3 * t + (size_t)k] + (((ch.tri_rel[t] >> k) & 1) ? (int64_t)vbase[c] : 0);
    }
  });
  std::vector<Chunk>().swap(chunks);

  // ---- transform (obj.rs:48-78: scale, rotate x, y, z, translate -- one matrix at a time)
  std::vector<Mat4> steps;
  if (tf.has_scale) steps.push_back(mat_scale(tf.scale.x, tf.scale.y, tf.scale.z));
  if (tf.has_rotate) {
    steps.push_back(mat_rotate_x(tf.rotate.x));
    steps.push_back(mat_rotate_y(tf.rotate.y));
    steps.push_back(mat_rotate_z(tf.rotate.z));
  }
  if (tf.has_translate) steps.push_back(mat_translate(tf.translate.x, tf.translate.y, tf.translate.z));
  const int workers = host_threads();
  parallel_ranges(n_vertices, workers, 4096, [&](size_t b, size_t e) {
    for (size_t i = b; i < e; i++)
      for (const Mat4 &m : steps) vertices[i] = m.apply(vertices[i]);
  });

  // ---- triangles, in file order; indices are local to their object (obj.rs:83-99)
  const size_t first_triangle = out.triangles.size(), first_element = out.elements.size();
  out.triangles.resize(first_triangle + n_tris);
  out.elements.resize(first_element + n_tris);
  for (size_t o = 0; o + 1 < objects.size(); o++) {
    const size_t v0 = objects[o].first_vertex, nv = objects[o + 1].first_vertex - v0;
    const size_t t0 = objects[o].first_tri, nt = objects[o + 1].first_tri - t0;
    parallel_ranges(nt, workers, 4096, [&](size_t b, size_t e) {
      for (size_t t = t0 + b; t < t0 + e; t++) {
        int64_t ia = tri[3 * t] - 1 - (int64_t)v0, ib = tri[3 * t + 1] - 1 - (int64_t)v0, ic = tri[3 * t + 2] - 1 - (int64_t)v0;
        if (ia < 0 || ib < 0 || ic < 0 || ia >= (int64_t)nv || ib >= (int64_t)nv || ic >= (int64_t)nv)
          throw std::runtime_error("obj: face index out of range in `" + path + "`");
        make_triangle(vertices[v0 + (size_t)ia], vertices[v0 + (size_t)ib], vertices[v0 + (size_t)ic], material, (int)(first_triangle + t),
                      out.triangles[first_triangle + t], out.elements[first_element + t]);
      }
    });
  }
}

void expand_elements(const Json &scene, const std::string &base_dir, SceneData &out) {
  const Json &elements = scene.at("elements");
  if (elements.kind != Json::Array) throw std::runtime_error("`elements` must be an array");
  for (const Json &je : elements.array) {
    if (je.kind != Json::Object || je.object.size() != 1) throw std::runtime_error("element: expected a single-variant object");
    const std::string &tag = je.object[0].first;
    const Json &body = je.object[0].second;
    if (tag == "JSONQuad") {  // quad.rs:26-37
      int material = material_from_json(body.at("material"), out);
      out.add_quad(vec3_from_json(body.at("q")), vec3_from_json(body.at("u")), vec3_from_json(body.at("v")), material,
                   option_true(body, "attractor"));
    } else if (tag == "JSONSphere") {  // sphere.rs:29-47
      int material = material_from_json(body.at("material"), out);
      Vec3 center = vec3_from_json(body.at("center"));
      Transform tf = transform_from_json(body);
      if (tf.has_translate) center = mat_translate(tf.translate.x, tf.translate.y, tf.translate.z).apply(center);
      out.add_sphere(center, body.at("radius").as_number(), material, option_true(body, "attractor"));
    } else if (tag == "JSONTriangle") {  // triangle.rs:35-41
      int material = material_from_json(body.at("material"), out);
      out.add_triangle(vec3_from_json(body.at("v0")), vec3_from_json(body.at("v1")), vec3_from_json(body.at("v2")), material);
    } else if (tag == "JSONBox") {
      add_box(body, out);
    } else if (tag == "JSONObj") {
      add_obj(body, base_dir, out);
    } else {
      throw std::runtime_error("unknown element variant `" + tag + "`");
    }
  }
  // integrator.rs:192 / elements/mod.rs:68-75: spheres and quads flagged Some(true)
  for (size_t i = 0; i < out.elements.size(); i++) {
    const Element &e = out.elements[i];
    bool a = (e.kind == RT_PRIM_SPHERE && out.spheres[e.index].attractor) || (e.kind == RT_PRIM_QUAD && out.quads[e.index].attractor);
    if (a) out.attractors.push_back((int32_t)i);
  }
  out.element_refs.resize(out.elements.size());
  for (size_t i = 0; i < out.elements.size(); i++) out.element_refs[i] = RtElementRef{out.elements[i].kind, out.elements[i].index};
}

}  // namespace rt
