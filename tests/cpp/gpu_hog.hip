// TEST INFRASTRUCTURE.  A "foreign" kernel for tests/test_gpu_parity.py: holds K compute units of device 0 for T ms -- K
// workgroups of one wavefront with 140 KiB of LDS each (one per CU; nothing that needs more than 20 KiB of LDS fits beside
// one), spinning on the 100 MHz clock.  Stands for what the library cannot know about: another process's kernels, the host
// application's own.
//   hipcc --offload-arch=gfx950 -O2 tests/cpp/gpu_hog.hip -o gpu_hog && ./gpu_hog 200 4000
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(64) void hog(unsigned long long ticks, unsigned* sink) {
  __shared__ unsigned lds[140 * 256];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned x = lds[(threadIdx.x * 7) & 63];
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(64); x = x * 1664525u + 1013904223u; }
  if (x == 0x12345678u) sink[0] = x;
}

int main(int argc, char** argv) {
  const int k = argc > 1 ? atoi(argv[1]) : 64, ms = argc > 2 ? atoi(argv[2]) : 2000;
  unsigned* sink = nullptr;
  if (hipMalloc((void**)&sink, 64) != hipSuccess) { printf("no device\n"); return 1; }
  hipLaunchKernelGGL(hog, dim3(k), dim3(64), 0, 0, (unsigned long long)ms * 100000ull, sink);
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
  printf("holding %d compute units for %d ms\n", k, ms);
  fflush(stdout);
  if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed\n"); return 1; }
  printf("released\n");
  return 0;
}
