// load_latency.hip -- what a dependent load costs a lone wave while a streaming kernel saturates HBM: the replay of a handful of
// queries is one wave per query next to a scan that runs at HBM speed, and every memory round trip in its chain
// costs this.  A pointer chase (one 64-lane wave, each step one dword per lane from a line picked by the previous step) runs alone
// and next to a grid-stride streaming read with a given number of resident waves per CU; reported: ns per step and the stream's GB/s.
// Build: hipcc --offload-arch=gfx950 -O3 tools/load_latency.hip -o tools/load_latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int UNROLL>
__global__ __launch_bounds__(256) void k_stream(const u32x4 *__restrict__ in, size_t n16, int passes, uint32_t *out)
{
  uint32_t acc = 0;
  const size_t step = (size_t)gridDim.x * 256;
  for (int p = 0; p < passes; p++) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + step * (UNROLL - 1) < n16; i += step * UNROLL) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; u++) v[u] = __builtin_nontemporal_load(in + i + step * u);
#pragma unroll
      for (int u = 0; u < UNROLL; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// lines of 256 B (64 dwords); next[line * 64 + lane] = index of the next line (same for every lane)
__global__ __launch_bounds__(64) void k_chase(const uint32_t *__restrict__ next, int steps, unsigned long long *ticks, uint32_t *sink)
{
  __builtin_amdgcn_s_setprio(3);
  uint32_t line = sink[0];                                    // goes on where the last run stopped: lines nobody has touched for a long time
  const unsigned long long t0 = wall_clock64();
  for (int s = 0; s < steps; s++) line = next[(size_t)line * 64 + threadIdx.x];
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) { ticks[0] = t1 - t0; sink[0] = line; }
}

int main(int argc, char **argv)
{
  setvbuf(stdout, NULL, _IONBF, 0);
  const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 4, bytes = gib << 30, n16 = bytes / 16;
  const int steps = 4000;
  u32x4 *buf; uint32_t *out, *next, *sink; unsigned long long *ticks;
  CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&out, 4)); CHECK(hipMemset(buf, 1, bytes));
  const size_t n_lines = (size_t)1 << 20;                     // 256 MiB of chase lines: larger than one XCD's L2, as large as the Infinity Cache
  {
    std::vector<uint32_t> perm(n_lines), h(n_lines * 64);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(7); std::shuffle(perm.begin() + 1, perm.end(), rng);
    for (size_t i = 0; i < n_lines; i++) { const uint32_t from = perm[i], to = perm[(i + 1) % n_lines]; for (int l = 0; l < 64; l++) h[(size_t)from * 64 + l] = to; }
    CHECK(hipMalloc(&next, h.size() * 4)); CHECK(hipMemcpy(next, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&ticks, 8)); CHECK(hipMemset(sink, 0, 4));
  int lo, hi; CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t s_hi, s_lo;
  CHECK(hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, hi)); CHECK(hipStreamCreateWithPriority(&s_lo, hipStreamNonBlocking, lo));
  int clk_khz = 100000; (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0);
  auto chase_ns = [&]() { unsigned long long t = 0; CHECK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost)); return (double)t / steps * 1e6 / clk_khz; };
  k_chase<<<1, 64, 0, s_hi>>>(next, steps, ticks, sink); CHECK(hipDeviceSynchronize());
  k_chase<<<1, 64, 0, s_hi>>>(next, steps, ticks, sink); CHECK(hipDeviceSynchronize());
  printf("wall clock %d kHz; pointer chase alone: %.0f ns per dependent load\n", clk_khz, chase_ns());
  printf("%-28s %10s %12s\n", "stream (blocks of 256 thr)", "GB/s", "chase ns/load");
  struct Cfg { int blocks, unroll; };
  const Cfg cfgs[] = {{256, 1}, {256, 4}, {512, 4}, {1024, 4}, {2048, 4}, {2048, 8}, {4096, 4}, {8192, 4}};
  for (const Cfg &c : cfgs) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int passes = 6;
    CHECK(hipEventRecord(a, s_lo));
    if (c.unroll == 1) k_stream<1><<<c.blocks, 256, 0, s_lo>>>(buf, n16, passes, out);
    else if (c.unroll == 4) k_stream<4><<<c.blocks, 256, 0, s_lo>>>(buf, n16, passes, out);
    else k_stream<8><<<c.blocks, 256, 0, s_lo>>>(buf, n16, passes, out);
    CHECK(hipEventRecord(b, s_lo));
    k_chase<<<1, 64, 0, s_hi>>>(next, steps, ticks, sink);
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    printf("%6d blocks x%d loads in flight %10.1f %12.0f\n", c.blocks, c.unroll, (double)bytes * passes / ms / 1e6, chase_ns());
  }
  return 0;
}
