// Micro-benchmark: what does a plain streaming read of 1 GiB cost on this box with the launch geometry
// of the scan kernels (one 256-thread workgroup per 16 KiB tile) vs a persistent grid?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_tile_read(const uint4* p, uint64_t ntiles, unsigned long long* out) {
  const uint64_t tile = blockIdx.x;
  const uint4* t = p + tile * 1024;                    // 16 KiB = 1024 x 16 B
  uint4 a = t[threadIdx.x], b = t[threadIdx.x + 256], c = t[threadIdx.x + 512], d = t[threadIdx.x + 768];
  unsigned s = a.x ^ b.y ^ c.z ^ d.w ^ a.w ^ b.x ^ c.y ^ d.z;
  if (s == 0x12345678u) atomicAdd(out, 1ull);
}
__global__ void k_tile_read_ticket(const uint4* p, uint64_t ntiles, unsigned long long* out, unsigned* ticket) {
  __shared__ unsigned s_t;
  if (threadIdx.x == 0) s_t = atomicAdd(ticket + (blockIdx.x & 7), 1u) * 8 + (blockIdx.x & 7);
  __syncthreads();
  const uint64_t tile = s_t;
  if (tile >= ntiles) return;
  const uint4* t = p + tile * 1024;
  uint4 a = t[threadIdx.x], b = t[threadIdx.x + 256], c = t[threadIdx.x + 512], d = t[threadIdx.x + 768];
  unsigned s = a.x ^ b.y ^ c.z ^ d.w ^ a.w ^ b.x ^ c.y ^ d.z;
  if (s == 0x12345678u) atomicAdd(out, 1ull);
}
__global__ void k_persist_read(const uint4* p, uint64_t nvec, unsigned long long* out) {
  unsigned s = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nvec; i += (uint64_t)gridDim.x * blockDim.x) {
    uint4 a = p[i]; s ^= a.x ^ a.y ^ a.z ^ a.w;
  }
  if (s == 0x12345678u) atomicAdd(out, 1ull);
}
__global__ void k_tile_lds(const uint4* p, uint64_t ntiles, unsigned long long* out) {
  __shared__ uint4 buf[1024 + 64];
  const uint64_t tile = blockIdx.x;
  const uint4* t = p + tile * 1024;
  uint4 a = t[threadIdx.x], b = t[threadIdx.x + 256], c = t[threadIdx.x + 512], d = t[threadIdx.x + 768];
  buf[threadIdx.x] = a; buf[threadIdx.x + 256] = b; buf[threadIdx.x + 512] = c; buf[threadIdx.x + 768] = d;
  __syncthreads();
  uint4 e = buf[(threadIdx.x * 4 + 1) & 1023];
  if ((e.x ^ e.y) == 0x12345678u) atomicAdd(out, 1ull);
}

int main() {
  const uint64_t bytes = 1ull << 30, ntiles = bytes / 16384, nvec = bytes / 16;
  uint4* d; unsigned long long* out; unsigned* ticket;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 8)); CK(hipMalloc(&ticket, 64));
  CK(hipMemset(d, 1, bytes)); CK(hipMemset(out, 0, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    float best = 1e9;
    for (int r = 0; r < 6; r++) {
      hipMemset(ticket, 0, 64);
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
    }
    printf("%-28s %.3f ms  %.1f GB/s\n", name, best, bytes / best / 1e6);
  };
  time("tile_read grid=ntiles", [&] { hipLaunchKernelGGL(k_tile_read, dim3(ntiles), dim3(256), 0, 0, d, ntiles, out); });
  time("tile_read + ticket", [&] { hipLaunchKernelGGL(k_tile_read_ticket, dim3(ntiles), dim3(256), 0, 0, d, ntiles, out, ticket); });
  time("tile_read -> LDS", [&] { hipLaunchKernelGGL(k_tile_lds, dim3(ntiles), dim3(256), 0, 0, d, ntiles, out); });
  for (int g : {1024, 2048, 4096, 8192})
    time(g == 1024 ? "persistent grid=1024" : g == 2048 ? "persistent grid=2048" : g == 4096 ? "persistent grid=4096" : "persistent grid=8192",
         [&] { hipLaunchKernelGGL(k_persist_read, dim3(g), dim3(256), 0, 0, d, nvec, out); });
  return 0;
}
