// VALU issue rate on gfx950: N independent 32-bit integer ops per lane, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* o, uint32_t seed, int iters) {
  uint32_t a0 = threadIdx.x ^ seed, a1 = a0 * 3u, a2 = a0 + 7u, a3 = a0 ^ 99u, a4 = a0 + 1u, a5 = a0 + 2u, a6 = a0 + 3u, a7 = a0 + 4u;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      if (MODE == 0) { a0 = (a0 ^ a1) + 0x76767676u; a1 = (a1 ^ a2) + 0x01010101u; a2 = (a2 ^ a3) + 0x11u; a3 = (a3 ^ a4) + 0x13u;
                       a4 = (a4 ^ a5) + 0x17u; a5 = (a5 ^ a6) + 0x19u; a6 = (a6 ^ a7) + 0x23u; a7 = (a7 ^ a0) + 0x29u; }
      else { a0 = __builtin_amdgcn_udot4(a0, 0x08040201u, a1, false); a1 = __builtin_amdgcn_udot4(a1, 0x08040201u, a2, false);
             a2 = __builtin_amdgcn_udot4(a2, 0x08040201u, a3, false); a3 = __builtin_amdgcn_udot4(a3, 0x08040201u, a4, false);
             a4 = __builtin_amdgcn_udot4(a4, 0x08040201u, a5, false); a5 = __builtin_amdgcn_udot4(a5, 0x08040201u, a6, false);
             a6 = __builtin_amdgcn_udot4(a6, 0x08040201u, a7, false); a7 = __builtin_amdgcn_udot4(a7, 0x08040201u, a0, false); }
    }
  }
  o[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int MODE>
void run(const char* name, int ops_per_iter) {
  uint32_t* d; hipMalloc(&d, 256 * 2048 * 4 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int bpc = 1; bpc <= 8; bpc *= 2) {            // blocks per CU -> waves per SIMD
    const int grid = 256 * bpc, iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1u, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1u, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)grid * 4 * iters * ops_per_iter;       // wave-instructions
    printf("%s waves/SIMD=%d: %.3f ms, %.2f wave-instr/cycle/SIMD @2.4GHz (cycles per wave-instr per SIMD = %.2f)\n", name, bpc, ms,
           winstr / 1024.0 / (ms * 1e-3 * 2.4e9), (ms * 1e-3 * 2.4e9) / (winstr / 1024.0));
  }
}
int main() { run<0>("xor+add", 16 * 16); run<1>("dot4", 16 * 8); return 0; }
