// valu_rate.hip -- measures sustained wave64 issue rates of the integer VALU ops the scan kernel is made of.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

#define KERNEL(NAME, ASM)                                                                       \
__global__ __launch_bounds__(256) void NAME(uint32_t* out, const uint32_t* in, int iters) {      \
  uint32_t a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
  uint32_t b = in[threadIdx.x + 256];                                                             \
  uint32_t s = in[blockIdx.x & 7];  s = __builtin_amdgcn_readfirstlane(s);                        \
  for (int i = 0; i < iters; i++) {                                                               \
    REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(s));) \
  }                                                                                               \
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                   \
}
// each asm block = 8 independent instructions (one per chain)
#define EIGHT(OP, TAIL) OP " %0, %0, " TAIL "\n" OP " %1, %1, " TAIL "\n" OP " %2, %2, " TAIL "\n" OP " %3, %3, " TAIL "\n" \
                        OP " %4, %4, " TAIL "\n" OP " %5, %5, " TAIL "\n" OP " %6, %6, " TAIL "\n" OP " %7, %7, " TAIL "\n"
KERNEL(k_and_vv,   EIGHT("v_and_b32", "%8"))
KERNEL(k_xor_sv,   "v_xor_b32 %0, %9, %0\nv_xor_b32 %1, %9, %1\nv_xor_b32 %2, %9, %2\nv_xor_b32 %3, %9, %3\nv_xor_b32 %4, %9, %4\nv_xor_b32 %5, %9, %5\nv_xor_b32 %6, %9, %6\nv_xor_b32 %7, %9, %7\n")
KERNEL(k_add_vv,   EIGHT("v_add_u32", "%8"))
KERNEL(k_bcnt,     "v_bcnt_u32_b32 %0, %8, %0\nv_bcnt_u32_b32 %1, %8, %1\nv_bcnt_u32_b32 %2, %8, %2\nv_bcnt_u32_b32 %3, %8, %3\nv_bcnt_u32_b32 %4, %8, %4\nv_bcnt_u32_b32 %5, %8, %5\nv_bcnt_u32_b32 %6, %8, %6\nv_bcnt_u32_b32 %7, %8, %7\n")
KERNEL(k_bitop3_vvv, "v_bitop3_b32 %0, %0, %8, %1 bitop3:0xbe\nv_bitop3_b32 %1, %1, %8, %2 bitop3:0xbe\nv_bitop3_b32 %2, %2, %8, %3 bitop3:0xbe\nv_bitop3_b32 %3, %3, %8, %4 bitop3:0xbe\nv_bitop3_b32 %4, %4, %8, %5 bitop3:0xbe\nv_bitop3_b32 %5, %5, %8, %6 bitop3:0xbe\nv_bitop3_b32 %6, %6, %8, %7 bitop3:0xbe\nv_bitop3_b32 %7, %7, %8, %0 bitop3:0xbe\n")
KERNEL(k_bitop3_vsv, "v_bitop3_b32 %0, %8, %9, %0 bitop3:0xbe\nv_bitop3_b32 %1, %8, %9, %1 bitop3:0xbe\nv_bitop3_b32 %2, %8, %9, %2 bitop3:0xbe\nv_bitop3_b32 %3, %8, %9, %3 bitop3:0xbe\nv_bitop3_b32 %4, %8, %9, %4 bitop3:0xbe\nv_bitop3_b32 %5, %8, %9, %5 bitop3:0xbe\nv_bitop3_b32 %6, %8, %9, %6 bitop3:0xbe\nv_bitop3_b32 %7, %8, %9, %7 bitop3:0xbe\n")
KERNEL(k_andor_vsv,  "v_and_or_b32 %0, %8, %9, %0\nv_and_or_b32 %1, %8, %9, %1\nv_and_or_b32 %2, %8, %9, %2\nv_and_or_b32 %3, %8, %9, %3\nv_and_or_b32 %4, %8, %9, %4\nv_and_or_b32 %5, %8, %9, %5\nv_and_or_b32 %6, %8, %9, %6\nv_and_or_b32 %7, %8, %9, %7\n")
KERNEL(k_fma_f32,    "v_fma_f32 %0, %0, %8, %0\nv_fma_f32 %1, %1, %8, %1\nv_fma_f32 %2, %2, %8, %2\nv_fma_f32 %3, %3, %8, %3\nv_fma_f32 %4, %4, %8, %4\nv_fma_f32 %5, %5, %8, %5\nv_fma_f32 %6, %6, %8, %6\nv_fma_f32 %7, %7, %8, %7\n")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %8\nv_pk_add_u16 %1, %1, %8\nv_pk_add_u16 %2, %2, %8\nv_pk_add_u16 %3, %3, %8\nv_pk_add_u16 %4, %4, %8\nv_pk_add_u16 %5, %5, %8\nv_pk_add_u16 %6, %6, %8\nv_pk_add_u16 %7, %7, %8\n")
KERNEL(k_mix_scan,   "v_xor_b32 %0, %9, %0\nv_bitop3_b32 %1, %8, %9, %0 bitop3:0xbe\nv_bitop3_b32 %2, %8, %9, %1 bitop3:0xbe\nv_and_b32 %3, %9, %2\nv_bcnt_u32_b32 %4, %3, %4\nv_bcnt_u32_b32 %5, %2, %5\nv_bitop3_b32 %6, %8, %9, %6 bitop3:0xea\nv_bcnt_u32_b32 %7, %6, %7\n")

typedef void (*kfn)(uint32_t*, const uint32_t*, int);
int main() {
  uint32_t *in, *out; int blocks_per_cu[] = {1, 2, 4, 8};
  hipMalloc(&in, 4096); hipMemset(in, 0x5a, 4096); hipMalloc(&out, 256 * 8 * 256 * 4);
  struct { const char* name; kfn f; } ks[] = {{"v_and_b32 v,v", k_and_vv}, {"v_xor_b32 s,v", k_xor_sv}, {"v_add_u32 v,v", k_add_vv}, {"v_bcnt_u32_b32", k_bcnt},
    {"v_bitop3 v,v,v", k_bitop3_vvv}, {"v_bitop3 v,s,v", k_bitop3_vsv}, {"v_and_or v,s,v", k_andor_vsv}, {"v_fma_f32", k_fma_f32}, {"v_pk_add_u16", k_pk_add_u16}, {"scan-like mix", k_mix_scan}};
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 20000;
  printf("%-18s", "op \\ waves/SIMD");
  for (int bpc : blocks_per_cu) printf(" %10d", bpc);
  printf("   (T lane-ops/s, 256 CUs; blocks of 256 threads = 1 wave per SIMD each)\n");
  for (auto& k : ks) {
    printf("%-18s", k.name);
    for (int bpc : blocks_per_cu) {
      int grid = 256 * bpc;
      hipLaunchKernelGGL(k.f, dim3(grid), dim3(256), 0, 0, out, in, 100);
      hipDeviceSynchronize();
      hipEventRecord(a); hipLaunchKernelGGL(k.f, dim3(grid), dim3(256), 0, 0, out, in, iters); hipEventRecord(b);
      hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
      double ops = (double)grid * 256 * iters * 64.0;   // 8 asm blocks x 8 instructions
      printf(" %10.2f", ops / (ms * 1e-3) / 1e12);
    }
    printf("\n");
  }
  return 0;
}
