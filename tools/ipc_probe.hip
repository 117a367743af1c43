// ipc_probe: can one process map another's device allocation (hipIpc*)?  usage: ipc_probe A <file> | ipc_probe B <file>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(int *p, int n, int v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + i; }
__global__ void sum(const int *p, int n, unsigned long long *out) { unsigned long long s = 0; for (int i = threadIdx.x; i < n; i += blockDim.x) s += (unsigned)p[i]; atomicAdd(out, s); }
int main(int argc, char **argv)
{
  if (argc < 3) return 2;
  const int n = 1 << 20;
  if (argv[1][0] == 'A') {
    int *d; CK(hipMalloc(&d, n * sizeof(int)));
    fill<<<n / 256, 256>>>(d, n, 7); CK(hipDeviceSynchronize());
    hipIpcMemHandle_t h; CK(hipIpcGetMemHandle(&h, d));
    FILE *f = fopen(argv[2], "wb"); fwrite(&h, sizeof h, 1, f); fclose(f);
    printf("A: handle written\n"); fflush(stdout);
    for (int i = 0; i < 100; i++) { if (access((std::string(argv[2]) + ".done").c_str(), F_OK) == 0) break; usleep(100000); }
    printf("A: done\n");
    return 0;
  }
  for (int i = 0; i < 100; i++) { if (access(argv[2], F_OK) == 0) break; usleep(100000); }
  usleep(200000);
  hipIpcMemHandle_t h; FILE *f = fopen(argv[2], "rb"); if (!f || fread(&h, sizeof h, 1, f) != 1) { printf("B: no handle\n"); return 1; } fclose(f);
  int *p; CK(hipIpcOpenMemHandle((void **)&p, h, hipIpcMemLazyEnablePeerAccess));
  unsigned long long *o; CK(hipMalloc(&o, 8)); CK(hipMemset(o, 0, 8));
  sum<<<1, 256>>>(p, n, o); CK(hipDeviceSynchronize());
  unsigned long long r; CK(hipMemcpy(&r, o, 8, hipMemcpyDeviceToHost));
  unsigned long long want = 0; for (int i = 0; i < n; i++) want += (unsigned)(7 + i);
  printf("B: sum %llu want %llu %s\n", r, want, r == want ? "OK" : "MISMATCH");
  CK(hipIpcCloseMemHandle(p));
  f = fopen((std::string(argv[2]) + ".done").c_str(), "wb"); fclose(f);
  return r == want ? 0 : 1;
}
