// hbm_read.hip -- streaming-read ceiling of the box: the number the small-Q scan is compared with beside the
// 8 TB/s vendor peak (SURVEY.md 8d asks for both).  A grid-stride sum of uint4 loads over a buffer far larger than the 256 MiB
// Infinity Cache, for several grid sizes and two access shapes: contiguous per block ("tile", the shape the scan uses:
// one 64-lane x 16 B row after the other) and fully interleaved ("stride").
// Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_read.hip -o tools/hbm_read ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every block walks its own contiguous chunk, 256 threads x 16 B x UNROLL per trip
template <int UNROLL>
__global__ __launch_bounds__(256) void k_tile(const u32x4 *__restrict__ in, size_t n16, uint32_t *out)
{
  size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  size_t lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
  uint32_t acc = 0;
  size_t i = lo + threadIdx.x;
  for (; i + (size_t) 256 * (UNROLL - 1) < hi; i += (size_t) 256 * UNROLL) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = __builtin_nontemporal_load(in + i + (size_t) 256 * u);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  for (; i < hi; i += 256) { u32x4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;    // keeps the loads alive, practically never taken
}

// classic grid-stride: consecutive blocks read consecutive 4 KiB
template <int UNROLL>
__global__ __launch_bounds__(256) void k_stride(const u32x4 *__restrict__ in, size_t n16, uint32_t *out)
{
  size_t step = (size_t) gridDim.x * 256;
  uint32_t acc = 0;
  size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  for (; i + step * (UNROLL - 1) < n16; i += step * UNROLL) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = __builtin_nontemporal_load(in + i + step * u);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  for (; i < n16; i += step) { u32x4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
static double time_ms (F launch, int reps)
{
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  launch(); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int r = 0; r < reps; r++) launch();
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
  return ms / reps;
}

int main (int argc, char **argv)
{
  setvbuf(stdout, NULL, _IONBF, 0);
  size_t gib = argc > 1 ? (size_t) atoi(argv[1]) : 4;
  size_t bytes = gib << 30, n16 = bytes / 16;
  u32x4 *buf; uint32_t *out;
  CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&out, 4));
  CHECK(hipMemset(buf, 1, bytes));
  printf("streaming read of %zu GiB (uint4 loads, 256 threads/block); GB/s = 1e9 bytes/s\n", gib);
  printf("%-10s %8s %10s %10s\n", "shape", "blocks", "ms", "GB/s");
  const int grids[] = {256, 512, 1024, 2048, 4096, 8192, 16384, 65536};
  double best = 0;
  for (int g : grids) {
    double ms = time_ms([&] { k_tile<4><<<g, 256>>>(buf, n16, out); }, 5);
    double gbs = bytes / ms / 1e6; if (gbs > best) best = gbs;
    printf("%-10s %8d %10.3f %10.1f\n", "tile x4", g, ms, gbs);
  }
  for (int g : grids) {
    double ms = time_ms([&] { k_tile<8><<<g, 256>>>(buf, n16, out); }, 5);
    double gbs = bytes / ms / 1e6; if (gbs > best) best = gbs;
    printf("%-10s %8d %10.3f %10.1f\n", "tile x8", g, ms, gbs);
  }
  for (int g : grids) {
    double ms = time_ms([&] { k_stride<4><<<g, 256>>>(buf, n16, out); }, 5);
    double gbs = bytes / ms / 1e6; if (gbs > best) best = gbs;
    printf("%-10s %8d %10.3f %10.1f\n", "stride x4", g, ms, gbs);
  }
  printf("best %.1f GB/s = %.3f of the 8000 GB/s vendor peak\n", best, best / 8000.0);
  CHECK(hipFree(buf)); CHECK(hipFree(out));
  return 0;
}
