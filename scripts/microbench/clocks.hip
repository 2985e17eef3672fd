// What do s_memtime (__builtin_readcyclecounter, clock64) and s_memrealtime (wall_clock64) count on gfx950?  (the spin
// watchdogs of the persistent kernel counted s_memtime ticks, "between 2 and 50 ms".)  Three kernels, each timed by HIP events on the host:
//   sleep   one wave polls s_memrealtime in an s_sleep loop for a fixed number of its ticks — the rest of the chip idle
//   sleepN  the same in every wave of a full grid (what a stalled persistent grid looks like: every wave asleep)
//   busy    one wave brackets a grid that keeps every SIMD issuing VALU work
// and the two counters' advance per microsecond of event time in each.
// hipcc --offload-arch=gfx950 -O3 scripts/microbench/clocks.hip -o scripts/microbench/clocks && scripts/microbench/clocks
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void k_sleep(unsigned long long* o, unsigned long long rt_ticks) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
  while (__builtin_amdgcn_s_memrealtime() - r0 < rt_ticks) __builtin_amdgcn_s_sleep(8);
  if (blockIdx.x == 0 && threadIdx.x == 0) { o[0] = __builtin_amdgcn_s_memrealtime() - r0; o[1] = __builtin_readcyclecounter() - c0; }
}
__global__ void k_busy(unsigned long long* o, float* sink, int iters) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f;
  for (int i = 0; i < iters; i++) { x = x * 1.0001f + y; y = y * 0.9999f + x; }
  if (x + y == 12345.678f) sink[0] = x;
  if (blockIdx.x == 0 && threadIdx.x == 0) { o[0] = __builtin_amdgcn_s_memrealtime() - r0; o[1] = __builtin_readcyclecounter() - c0; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  unsigned long long* o; float* sink;
  CK(hipMalloc(&o, 64)); CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("hipDeviceAttributeWallClockRate %d kHz, ClockRate %d kHz\n", rate, clk);
  for (int leg = 0; leg < 4; leg++) {
    for (int rep = 0; rep < 3; rep++) {
      unsigned long long h[2] = {0, 0};
      CK(hipEventRecord(e0, 0));
      if (leg == 0) hipLaunchKernelGGL(k_sleep, dim3(1), dim3(64), 0, 0, o, 1000000ull);
      else if (leg == 1) hipLaunchKernelGGL(k_sleep, dim3(1024), dim3(256), 0, 0, o, 1000000ull);
      else if (leg == 2) hipLaunchKernelGGL(k_sleep, dim3(1024), dim3(256), 0, 0, o, 5000000ull);
      else hipLaunchKernelGGL(k_busy, dim3(4096), dim3(256), 0, 0, o, sink, 400000);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h, o, 16, hipMemcpyDeviceToHost));
      const char* name[4] = {"sleep, one wave, 1 M realtime ticks", "sleep, 4096 waves, 1 M realtime ticks", "sleep, 4096 waves, 5 M realtime ticks", "busy, 16384 waves of VALU work"};
      printf("%-40s event %9.3f ms   s_memrealtime %10llu ticks = %7.2f per us   s_memtime %12llu ticks = %8.2f per us\n", name[leg], ms, h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3));
    }
  }
  return 0;
}
