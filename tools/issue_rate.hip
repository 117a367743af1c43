// issue_rate.hip -- sustained issue rates of the NON-vector instruction classes the dirty-word loop of the scan is made of:
// scalar ALU ops, scalar compare+branch pairs, scalar loads that hit the scalar cache, LDS atomics, alone and mixed with VALU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/issue_rate.hip -o tools/issue_rate ; run on the GPU box.
// Output unit: G wave-instructions per second over the whole chip (256 CUs), for 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X X X X X X X X

// 8 scalar adds on 8 independent chains
__global__ __launch_bounds__(256) void k_salu(uint32_t *out, const uint32_t *in, int iters)
{
  uint32_t s0 = __builtin_amdgcn_readfirstlane(in[0]), s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("s_add_u32 %0, %0, 3\ns_add_u32 %1, %1, 3\ns_add_u32 %2, %2, 3\ns_add_u32 %3, %3, 3\ns_add_u32 %4, %4, 3\ns_add_u32 %5, %5, 3\ns_add_u32 %6, %6, 3\ns_add_u32 %7, %7, 3\n"
                      : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)::"scc");)
  }
  out[blockIdx.x * 256 + threadIdx.x] = s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;
}
// 4 x (s_bitcmp1 + not-taken s_cbranch) = 8 instructions
__global__ __launch_bounds__(256) void k_cmpbr(uint32_t *out, const uint32_t *in, int iters)
{
  uint32_t s0 = __builtin_amdgcn_readfirstlane(in[0]) & 0xFFFF0000u;   // low bits clear: branches never taken
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("s_bitcmp1_b32 %0, 0\ns_cbranch_scc1 1f\n1:\ns_bitcmp1_b32 %0, 1\ns_cbranch_scc1 2f\n2:\ns_bitcmp1_b32 %0, 2\ns_cbranch_scc1 3f\n3:\ns_bitcmp1_b32 %0, 3\ns_cbranch_scc1 4f\n4:\n"
                      : "+s"(s0)::"scc");)
  }
  out[blockIdx.x * 256 + threadIdx.x] = s0 ^ acc;
}
// 8 VALU only (reference)
__global__ __launch_bounds__(256) void k_valu(uint32_t *out, const uint32_t *in, int iters)
{
  uint32_t a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = in[threadIdx.x + 256];
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("v_and_b32 %0, %0, %4\nv_and_b32 %1, %1, %4\nv_and_b32 %2, %2, %4\nv_and_b32 %3, %3, %4\nv_and_b32 %0, %0, %4\nv_and_b32 %1, %1, %4\nv_and_b32 %2, %2, %4\nv_and_b32 %3, %3, %4\n"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
// 8 VALU + 8 SALU interleaved (16 instructions): do they share issue slots?
__global__ __launch_bounds__(256) void k_mix(uint32_t *out, const uint32_t *in, int iters)
{
  uint32_t a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = in[threadIdx.x + 256];
  uint32_t s0 = __builtin_amdgcn_readfirstlane(in[0]), s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("v_and_b32 %0, %0, %8\ns_add_u32 %4, %4, 3\nv_and_b32 %1, %1, %8\ns_add_u32 %5, %5, 3\nv_and_b32 %2, %2, %8\ns_add_u32 %6, %6, 3\nv_and_b32 %3, %3, %8\ns_add_u32 %7, %7, 3\n"
                      "v_and_b32 %0, %0, %8\ns_add_u32 %4, %4, 3\nv_and_b32 %1, %1, %8\ns_add_u32 %5, %5, 3\nv_and_b32 %2, %2, %8\ns_add_u32 %6, %6, 3\nv_and_b32 %3, %3, %8\ns_add_u32 %7, %7, 3\n"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b) : "scc");)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ s0 ^ s1 ^ s2 ^ s3;
}
// 8 VALU + 16 SALU (the ratio of the dirty-word loop before this experiment)
__global__ __launch_bounds__(256) void k_mix2(uint32_t *out, const uint32_t *in, int iters)
{
  uint32_t a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = in[threadIdx.x + 256];
  uint32_t s0 = __builtin_amdgcn_readfirstlane(in[0]), s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("v_and_b32 %0, %0, %8\ns_add_u32 %4, %4, 3\ns_add_u32 %5, %5, 3\nv_and_b32 %1, %1, %8\ns_add_u32 %6, %6, 3\ns_add_u32 %7, %7, 3\nv_and_b32 %2, %2, %8\ns_add_u32 %4, %4, 3\ns_add_u32 %5, %5, 3\nv_and_b32 %3, %3, %8\ns_add_u32 %6, %6, 3\ns_add_u32 %7, %7, 3\n"
                      "v_and_b32 %0, %0, %8\ns_add_u32 %4, %4, 3\ns_add_u32 %5, %5, 3\nv_and_b32 %1, %1, %8\ns_add_u32 %6, %6, 3\ns_add_u32 %7, %7, 3\nv_and_b32 %2, %2, %8\ns_add_u32 %4, %4, 3\ns_add_u32 %5, %5, 3\nv_and_b32 %3, %3, %8\ns_add_u32 %6, %6, 3\ns_add_u32 %7, %7, 3\n"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b) : "scc");)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ s0 ^ s1 ^ s2 ^ s3;
}
// 8 LDS atomic adds, every lane its own dword (no bank conflict)
__global__ __launch_bounds__(256) void k_dsadd(uint32_t *out, const uint32_t *in, int iters)
{
  __shared__ uint32_t acc[256 * 8];
  for (int k = 0; k < 8; k++) acc[threadIdx.x + k * 256] = 0;
  __syncthreads();
  uint32_t addr = (uint32_t)(uintptr_t)(&acc[threadIdx.x]) , v = in[threadIdx.x] & 3;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("ds_add_u32 %0, %1\nds_add_u32 %0, %1 offset:1024\nds_add_u32 %0, %1 offset:2048\nds_add_u32 %0, %1 offset:3072\nds_add_u32 %0, %1 offset:4096\nds_add_u32 %0, %1 offset:5120\nds_add_u32 %0, %1 offset:6144\nds_add_u32 %0, %1 offset:7168\n"
                      :: "v"(addr), "v"(v) : "memory");)
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x];
}
// 8 scalar loads of 32 bytes from a 4 KiB table (scalar-cache hits), consumed by a wait every 8
__global__ __launch_bounds__(256) void k_sload(uint32_t *out, const uint32_t *in, int iters)
{
  uint32_t acc = 0;
  const uint32_t *p = in;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("s_load_dwordx8 s[36:43], %1, 0x0\ns_load_dwordx8 s[44:51], %1, 0x20\ns_load_dwordx8 s[52:59], %1, 0x40\ns_load_dwordx8 s[60:67], %1, 0x60\n"
                      "s_load_dwordx8 s[36:43], %1, 0x80\ns_load_dwordx8 s[44:51], %1, 0xa0\ns_load_dwordx8 s[52:59], %1, 0xc0\ns_load_dwordx8 s[60:67], %1, 0xe0\ns_waitcnt lgkmcnt(0)\ns_add_u32 %0, %0, s36\n"
                      : "+s"(acc) : "s"(p) : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",
                        "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "memory", "scc");)
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef void (*kfn)(uint32_t *, const uint32_t *, int);
int main()
{
  uint32_t *in, *out;
  int blocks_per_cu[] = {1, 2, 4, 8};
  hipMalloc(&in, 8192); hipMemset(in, 0x5a, 8192); hipMalloc(&out, 256 * 8 * 256 * 4);
  struct { const char *name; kfn f; double per_rep; } ks[] = {
    {"v_and only (8)", k_valu, 8}, {"s_add only (8)", k_salu, 8}, {"s_bitcmp+branch (8)", k_cmpbr, 8}, {"8 valu + 8 salu", k_mix, 16}, {"8 valu + 16 salu", k_mix2, 24},
    {"ds_add_u32 (8)", k_dsadd, 8}, {"s_load_dwordx8 (8)", k_sload, 8}};
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000;
  setvbuf(stdout, nullptr, _IONBF, 0);
  printf("%-22s", "class \\ waves/SIMD");
  for (int bpc : blocks_per_cu) printf(" %10d", bpc);
  printf("   (G wave-instructions/s on 256 CUs; cycles per instruction per CU at 8 waves in the last column, 2.4 GHz nominal)\n");
  for (auto &k : ks) {
    printf("%-22s", k.name);
    double last = 0;
    for (int bpc : blocks_per_cu) {
      int grid = 256 * bpc;
      hipLaunchKernelGGL(k.f, dim3(grid), dim3(256), 0, 0, out, in, 50);
      hipDeviceSynchronize();
      hipEventRecord(a); hipLaunchKernelGGL(k.f, dim3(grid), dim3(256), 0, 0, out, in, iters); hipEventRecord(b);
      hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
      double winstr = (double)grid * 4 * iters * 8.0 * k.per_rep;     // 4 waves per block, 8 asm blocks per iteration
      last = winstr / (ms * 1e-3) / 1e9;
      printf(" %10.1f", last);
    }
    printf("   %6.2f\n", 256 * 2.4 / last);
    }
  return 0;
}
