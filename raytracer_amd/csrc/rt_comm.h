// One image rendered by several processes, one per GPU, inside the C library: where the reference deals the image's rows to
// the threads of one process (/root/reference/src/render/integrator.rs:226-243), rank r of `world` renders the row blocks
// r, r + world, ... on its GPU (RtConfig.row_first / row_stride / row_block) and rank 0 receives every rank's rows over RCCL
// (xGMI between the GPUs of a node) and puts the image together on its device: one group of sends and receives per render, no
// other traffic -- pixels are independent given the replicated scene.  RCCL is loaded when the first communicator is made
// (dlopen), so a process that renders on one GPU never loads it.
//
// Included by rt_hip.hip (one translation unit with the handle, the render and the error plumbing).
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace comm {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static int rccl_load(const Rccl **out) {
  static std::mutex m;
  static Rccl r;
  static std::string why;
  std::lock_guard<std::mutex> lock(m);
  if (!r.lib && why.empty()) {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      why = std::string("RCCL could not be loaded: ") + (dlerror() ? dlerror() : "librccl.so not found");
    } else {
      bool ok = true;
      auto sym = [&](const char *name) {
        void *p = dlsym(r.lib, name);
        if (!p) {
          ok = false;
          why = std::string("RCCL lacks ") + name;
        }
        return p;
      };
      r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
      r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
      r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
      r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
      r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
      r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
      r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
      r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
      r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
      if (!ok) {
        dlclose(r.lib);
        r.lib = nullptr;
      }
    }
  }
  if (!r.lib) return fail(RT_ERR_HIP, why);
  *out = &r;
  return RT_OK;
}

}  // namespace comm

struct RtComm {
  const comm::Rccl *rccl = nullptr;
  ncclComm_t nccl = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  // device buffers, grown as needed and kept: this rank's rows (two planes: image, heat-map); on rank 0 every rank's rows one
  // after the other, and the assembled image; the counters that are summed over the ranks
  double *mine = nullptr, *all = nullptr, *image = nullptr;
  size_t mine_cap = 0, all_cap = 0, image_cap = 0;
  unsigned long long *sums = nullptr;
};

#define RCCL_TRY(c, expr)                                                                                       \
  do {                                                                                                          \
    ncclResult_t r_ = (expr);                                                                                   \
    if (r_ != ncclSuccess) return fail(RT_ERR_HIP, std::string(#expr) + ": " + (c)->rccl->GetErrorString(r_)); \
  } while (0)

static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "the header states the size of a communicator id");
constexpr int kCommSums = 3 + RT_MAX_DEPTH_STATS;  // rays, samples, rays_traced, rays_by_depth[]

static int comm_id_impl(unsigned char *id) {
  g_error.clear();
  if (!id) return fail(RT_ERR_BAD_ARG, "id is NULL");
  const comm::Rccl *rccl;
  int rc = comm::rccl_load(&rccl);
  if (rc != RT_OK) return rc;
  ncclUniqueId u;
  ncclResult_t r = rccl->GetUniqueId(&u);
  if (r != ncclSuccess) return fail(RT_ERR_HIP, std::string("ncclGetUniqueId: ") + rccl->GetErrorString(r));
  memcpy(id, &u, sizeof u);
  return RT_OK;
}

static int comm_free_impl(RtComm *c) {
  if (!c) return RT_OK;
  (void)hipSetDevice(c->device);
  if (c->nccl) (void)c->rccl->CommDestroy(c->nccl);
  for (void *p : {(void *)c->mine, (void *)c->all, (void *)c->image, (void *)c->sums})
    if (p) (void)hipFree(p);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return RT_OK;
}

static int comm_init_impl(const unsigned char *id, int rank, int world, int device, RtComm **out) {
  g_error.clear();
  if (!out) return fail(RT_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail(RT_ERR_BAD_ARG, "rank must lie in [0, world)");
  if (!id && rank != 0) return fail(RT_ERR_BAD_ARG, "a communicator without an id plays every rank itself: its rank is 0");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(RT_ERR_BAD_ARG, "no such device");
  const comm::Rccl *rccl = nullptr;
  int rc = id ? comm::rccl_load(&rccl) : RT_OK;  // id == NULL: no RCCL, the one process renders the ranks' shards in turn
  if (rc != RT_OK) return rc;
  HIP_TRY(hipSetDevice(device));
  RtComm *c = new (std::nothrow) RtComm;
  if (!c) return fail(RT_ERR_OOM, "out of host memory");
  c->rccl = rccl;
  c->rank = rank;
  c->world = world;
  c->device = device;
  struct FreeOnUnwind {
    RtComm *c;
    ~FreeOnUnwind() {
      if (c) comm_free_impl(c);
    }
  } unwind{c};
  HIP_TRY(hipStreamCreate(&c->stream));
  HIP_TRY(hipMalloc(&c->sums, 2 * kCommSums * sizeof(unsigned long long)));
  if (id) {
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    RCCL_TRY(c, rccl->CommInitRank(&c->nccl, world, u, rank));
  }
  unwind.c = nullptr;
  *out = c;
  return RT_OK;
}

// rows of the image that rank r of `world` renders when blocks of B rows are dealt (the shard_rows of its RtConfig)
static int64_t comm_rows_of(int H, int B, int r, int world) {
  int64_t rows = 0;
  for (int64_t blk = r; blk * B < H; blk += world) rows += std::min<int64_t>(B, H - blk * B);
  return rows;
}

// image row y <- the rank that rendered it, at its place among that rank's rows; `offset[r]` = where rank r's two planes start
// in `all` (in doubles), `rows[r]` = its rows.  One lane per double.
__global__ void __launch_bounds__(256) k_interleave_rows(const double *all, const int64_t *offset, const int64_t *rows, double *image, int H, int64_t w3, int world,
                                                          int B) {
  const int64_t n = 2 * (int64_t)H * w3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t plane = i / ((int64_t)H * w3), rem = i - plane * (int64_t)H * w3, y = rem / w3, x = rem - y * w3;
    const int64_t blk = y / B, r = blk % world, local = (blk / world) * B + y % B;
    image[i] = all[offset[r] + (plane * rows[r] + local) * w3 + x];
  }
}

static int comm_grow(double **buf, size_t *cap, size_t need) {
  if (*cap >= need) return RT_OK;
  if (*buf) HIP_TRY(hipFree(*buf));
  *buf = nullptr;
  *cap = 0;
  HIP_TRY(hipMalloc(buf, std::max<size_t>(need, 1) * sizeof(double)));
  *cap = need;
  return RT_OK;
}

static int render_ranks_impl(RtSceneHandle *h, RtComm *c, const RtConfig *cfg_in, RtImage *out, RtStats *stats) {
  g_error.clear();
  if (!h || !c || !cfg_in) return fail(RT_ERR_BAD_ARG, "handle, comm or cfg is NULL");
  if (h->device != c->device) return fail(RT_ERR_BAD_ARG, "the scene and the communicator live on different devices");
  const bool root = c->rank == 0;
  if (root && (!out || !out->pixels)) return fail(RT_ERR_BAD_ARG, "rank 0 needs out->pixels");
  if (out && (out->linear || out->samples || out->rgb8 || out->sample_rgb8))
    return fail(RT_ERR_UNSUPPORTED, "only the two buffers render_line fills (pixels, sample_pixels) are gathered");
  if (root && out->memory != RT_MEM_HOST && out->memory != RT_MEM_DEVICE) return fail(RT_ERR_BAD_ARG, "bad RtImage.memory");
  RtConfig cfg = *cfg_in;
  cfg.row_first = c->rank;
  cfg.row_stride = c->world;
  int rc = check_config(&cfg);
  if (rc != RT_OK) return rc;
  auto wall0 = std::chrono::steady_clock::now();
  HIP_TRY(hipSetDevice(c->device));
  const int H = cfg.img_height, B = row_block_of(&cfg);
  const int64_t w3 = 3 * (int64_t)cfg.img_width;
  const int64_t my_rows = comm_rows_of(H, B, c->rank, c->world);
  if (my_rows != shard_rows(&cfg)) return fail(RT_ERR_HIP, "row bookkeeping of the gather disagrees with the render's");
  const bool loopback = c->nccl == nullptr && c->world > 1;  // this process plays every rank in turn
  std::vector<int64_t> rows((size_t)c->world), offset((size_t)c->world);
  int64_t total = 0;
  for (int r = 0; r < c->world; r++) {
    rows[(size_t)r] = comm_rows_of(H, B, r, c->world);
    offset[(size_t)r] = total;
    total += 2 * rows[(size_t)r] * w3;
  }
  if (root) {
    if ((rc = comm_grow(&c->all, &c->all_cap, (size_t)total + 2 * (size_t)c->world)) != RT_OK) return rc;  // (+ room for the offsets and row counts)
    if ((rc = comm_grow(&c->image, &c->image_cap, (size_t)(2 * (int64_t)H * w3))) != RT_OK) return rc;
  }
  RtStats local{};
  if (loopback) {
    for (int r = 0; r < c->world; r++) {
      if (rows[(size_t)r] == 0) continue;
      RtConfig cr = cfg;
      cr.row_first = r;
      RtImage img{};
      img.pixels = c->all + offset[(size_t)r];
      img.sample_pixels = img.pixels + rows[(size_t)r] * w3;
      img.memory = RT_MEM_DEVICE;
      RtStats part{};
      if ((rc = render_resident_impl(h, &cr, &img, &part)) != RT_OK) return rc;
      local.rays += part.rays;
      local.samples += part.samples;
      local.rays_traced += part.rays_traced;
      for (int k = 0; k < RT_MAX_DEPTH_STATS; k++) local.rays_by_depth[k] += part.rays_by_depth[k];
      local.trace_kernel_ms += part.trace_kernel_ms;
      local.shade_kernel_ms += part.shade_kernel_ms;
      local.other_kernel_ms += part.other_kernel_ms;
      local.trace_launches += part.trace_launches;
      local.batches = std::max(local.batches, part.batches);
    }
  } else {
    if ((rc = comm_grow(&c->mine, &c->mine_cap, (size_t)(2 * my_rows * w3))) != RT_OK) return rc;
    if (my_rows > 0) {
      RtImage img{};
      img.pixels = c->mine;
      img.sample_pixels = c->mine + my_rows * w3;
      img.memory = RT_MEM_DEVICE;
      if ((rc = render_resident_impl(h, &cfg, &img, &local)) != RT_OK) return rc;
    }
    if (root && my_rows > 0)
      HIP_TRY(hipMemcpyAsync(c->all + offset[0], c->mine, (size_t)(2 * my_rows * w3) * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  // ---- every rank's rows to rank 0: one group of sends and receives
  if (c->world > 1 && !loopback) {
    RCCL_TRY(c, c->rccl->GroupStart());
    if (root) {
      for (int r = 1; r < c->world; r++)
        if (rows[(size_t)r] > 0) RCCL_TRY(c, c->rccl->Recv(c->all + offset[(size_t)r], (size_t)(2 * rows[(size_t)r] * w3), ncclFloat64, r, c->nccl, c->stream));
    } else if (my_rows > 0) {
      RCCL_TRY(c, c->rccl->Send(c->mine, (size_t)(2 * my_rows * w3), ncclFloat64, 0, c->nccl, c->stream));
    }
    RCCL_TRY(c, c->rccl->GroupEnd());
  }
  // ---- the counters of the whole job on every rank
  unsigned long long host_sums[kCommSums] = {local.rays, local.samples, local.rays_traced};
  for (int k = 0; k < RT_MAX_DEPTH_STATS; k++) host_sums[3 + k] = local.rays_by_depth[k];
  HIP_TRY(hipMemcpyAsync(c->sums, host_sums, sizeof host_sums, hipMemcpyHostToDevice, c->stream));
  if (c->world > 1 && !loopback) RCCL_TRY(c, c->rccl->AllReduce(c->sums, c->sums + kCommSums, kCommSums, ncclUint64, ncclSum, c->nccl, c->stream));
  else HIP_TRY(hipMemcpyAsync(c->sums + kCommSums, c->sums, sizeof host_sums, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(host_sums, c->sums + kCommSums, sizeof host_sums, hipMemcpyDeviceToHost, c->stream));
  // ---- rank 0 puts the image together and hands it over
  if (root) {
    int64_t *d_meta = reinterpret_cast<int64_t *>(c->all + total);
    std::vector<int64_t> meta(offset);
    meta.insert(meta.end(), rows.begin(), rows.end());
    HIP_TRY(hipMemcpyAsync(d_meta, meta.data(), meta.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    const int64_t n = 2 * (int64_t)H * w3;
    hipLaunchKernelGGL(k_interleave_rows, dim3(grid_for(n, 256, 256 * 8)), dim3(256), 0, c->stream, c->all, d_meta, d_meta + c->world, c->image, H, w3, c->world, B);
    LAUNCHED("k_interleave_rows");
    const hipMemcpyKind kind = out->memory == RT_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    HIP_TRY(hipMemcpyAsync(out->pixels, c->image, (size_t)((int64_t)H * w3) * sizeof(double), kind, c->stream));
    if (out->sample_pixels) HIP_TRY(hipMemcpyAsync(out->sample_pixels, c->image + (int64_t)H * w3, (size_t)((int64_t)H * w3) * sizeof(double), kind, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (stats) {
    *stats = local;  // kernel times and visit counters are this rank's; the query counts are the job's
    stats->rays = host_sums[0];
    stats->samples = host_sums[1];
    stats->rays_traced = host_sums[2];
    for (int k = 0; k < RT_MAX_DEPTH_STATS; k++) stats->rays_by_depth[k] = host_sums[3 + k];
    stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  }
  return RT_OK;
}
