// Checks the 64-lane reversal built from DPP row_mirror + v_permlane16_swap + v_permlane32_swap (gfx950).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t lane_reverse(uint32_t v, int lane) {
  const uint32_t m = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140 /*row_mirror*/, 0xF, 0xF, false);
  auto r16 = __builtin_amdgcn_permlane16_swap(m, m, false, false);
  const uint32_t s16 = ((lane >> 4) & 1) ? r16[0] : r16[1];
  auto r32 = __builtin_amdgcn_permlane32_swap(s16, s16, false, false);
  return (lane & 32) ? r32[0] : r32[1];
}
__global__ void k(uint32_t* o, const uint32_t* a) { o[threadIdx.x] = lane_reverse(a[threadIdx.x], threadIdx.x & 63); }
int main() {
  uint32_t h[64], r[64], *di, *dout;
  for (int i = 0; i < 64; i++) h[i] = 1000 + i;
  hipMalloc(&di, 256); hipMalloc(&dout, 256);
  hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, di);
  hipMemcpy(r, dout, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; i++) if (r[i] != h[63 - i]) bad++;
  printf("lane_reverse: %s (lane0=%u lane1=%u lane16=%u lane63=%u)\n", bad ? "WRONG" : "ok", r[0], r[1], r[16], r[63]);
  return bad != 0;
}
