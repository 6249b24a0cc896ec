// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of the renderer:
// how many bytes the counter reports for a KNOWN number of distinct bytes, each read once, from a
// buffer far larger than the Infinity Cache (1 GiB).
//   k_stream16    every lane reads 16 consecutive bytes (the wide streaming read the guide calibrates)
//   k_scatter32   every lane reads one 32-byte record (two 16-byte loads) at a pseudo-random position
//                 (the BVH node fetch of k_extend); every record is read exactly once
//   k_scatter80   every lane reads one 80-byte record (five 16-byte loads), likewise (the triangle fetch)
// Run under `rocprofv3 --pmc FETCH_SIZE` and compare the counter (KiB) with the bytes printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_stream16(const float4 *p, size_t n, float *sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = p[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) *sink = acc;
}
// records of `words16` 16-byte pieces; record r of n (n a power of two) is visited by thread perm(r) = r * odd mod n
template <int PIECES>
__global__ void k_scatter(const float4 *p, size_t n_records, float *sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_records; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = (i * 0x9E3779B1ull + 12345ull) & (n_records - 1);
    const float4 *q = p + r * PIECES;
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
      float4 v = q[k];
      acc += v.x + v.w;
    }
  }
  if (acc == 12345.678f) *sink = acc;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  float4 *buf;
  float *sink;
  CHECK(hipMalloc(&buf, bytes + 4096));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(buf, 1, bytes));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  auto timed = [&](const char *name, size_t useful, auto launch) {
    launch();                         // first launch of a kernel pays for loading its code
    CHECK(hipMemset(buf, 1, bytes));  // evicts what the last kernel left in the caches (1 GiB > 256 MiB)
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("{\"kernel\": \"%s\", \"useful_bytes\": %zu, \"ms\": %.4f, \"useful_GBs\": %.1f}\n", name, useful, ms, useful / (ms * 1e-3) / 1e9);
  };
  const int grid = 256 * 16, block = 256;
  timed("k_stream16", bytes, [&] { hipLaunchKernelGGL(k_stream16, dim3(grid), dim3(block), 0, 0, buf, bytes / 16, sink); });
  timed("k_scatter<2>", bytes, [&] { hipLaunchKernelGGL((k_scatter<2>), dim3(grid), dim3(block), 0, 0, buf, bytes / 32, sink); });
  // 80-byte records: 2^23 of them = 640 MiB
  timed("k_scatter<5>", ((size_t)1 << 23) * 80, [&] { hipLaunchKernelGGL((k_scatter<5>), dim3(grid), dim3(block), 0, 0, buf, (size_t)1 << 23, sink); });
  CHECK(hipDeviceSynchronize());
  return 0;
}
