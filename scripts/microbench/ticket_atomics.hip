// Micro-benchmark (round 6): how to hand out tickets to the waves of a persistent streaming kernel.
//   mode 0: no tickets (static units)            mode 1: vector atomic with return by lane 0, consumed 8 "tiles" later
//   mode 2: scalar atomic (s_atomic_add ... glc), consumed 8 tiles later — returns through lgkmcnt, not through the in-order VMEM queue
// Every wave streams "units" of 9 x 4 KiB with a one-tile prefetch, as k_scan_fields_pers does.  Prints ms and, for modes 1 / 2, whether every
// ticket of every counter was handed out exactly once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 6) void k_stream(const uint8_t* hay, uint64_t len, uint32_t* ctr, uint32_t nunits, uint32_t* seen, unsigned long long* out) {
  const int lane = threadIdx.x & 63;
  const uint32_t W = gridDim.x * 4u, wv = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 7u;
  uint32_t* const myctr = ctr + xcc * 32u;
  const uint32_t per = (nunits + 7u - xcc) / 8u;
  uint32_t acc = 0;
  auto claim_v = [&]() -> uint32_t { uint32_t t = 0; if (lane == 0) t = __hip_atomic_fetch_add(myctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return t; };
  auto claim_s = [&]() -> uint32_t {
    uint32_t t = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(t) : "s"(myctr) : "memory");
    return t;                                                   // valid after s_waitcnt lgkmcnt(0)
  };
  const uint32_t c64 = blockIdx.x & 63u;
  uint32_t* const ctr64 = ctr + c64 * 32u;
  const uint32_t per64 = (nunits + 63u - c64) / 64u;
  auto claim_64 = [&]() -> uint32_t { uint32_t t = 0; if (lane == 0) t = __hip_atomic_fetch_add(ctr64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return t; };
  uint32_t u = wv, t_raw = 0;
  if (MODE == 5) { t_raw = __builtin_amdgcn_readfirstlane(claim_64()); u = t_raw < per64 ? t_raw * 64u + c64 : nunits; }
  // mode 3: per-XCD rank of this wave from a one-time atomic, then static steps: the ticket modes' address pattern without their atomics
  uint32_t rank = 0, nx = 0;
  if (MODE == 3) { uint32_t t = 0; if (lane == 0) t = __hip_atomic_fetch_add(myctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); rank = __builtin_amdgcn_readfirstlane(t); nx = W / 8u; u = rank * 8u + xcc; }
  if (MODE == 1) { t_raw = claim_v(); t_raw = __builtin_amdgcn_readfirstlane(t_raw); u = t_raw < per ? t_raw * 8u + xcc : nunits; }
  if (MODE == 2) { t_raw = claim_s(); asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t_raw)); u = t_raw < per ? t_raw * 8u + xcc : nunits; }
  while (u < nunits) {
    if (MODE != 0 && seen != nullptr && lane == 0) atomicAdd(seen + u, 1u);        // (checked on the host; a plain non-returning atomic)
    const uint64_t base = static_cast<uint64_t>(u) * 9u * 4096u;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(hay) + base, 0, 9 * 4096, 0x00020000);
    u32x4 x[4];
    for (int k = 0; k < 4; k++) x[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (lane + 64 * k) << 4, 0, 2);
    uint32_t nt = 0;
    if (MODE == 1 || MODE == 4) nt = claim_v();
    if (MODE == 2) nt = claim_s();
    if (MODE == 5) nt = claim_64();
    for (int j = 0; j < 9; j++) {
      u32x4 y[4];
      for (int k = 0; k < 4; k++) { y[k] = x[k]; if (j < 8) x[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((lane + 64 * k) << 4) + 4096 * (j + 1), 0, 2); }
      for (int k = 0; k < 4; k++) acc += __builtin_popcount(y[k].x ^ y[k].y) + __builtin_popcount(y[k].z & y[k].w);
      for (int r = 0; r < 24; r++) acc = acc * 1664525u + 1013904223u;      // some arithmetic per tile
    }
    if (MODE == 0) u += W;
    if (MODE == 3) { rank += nx; u = rank * 8u + xcc; }
    if (MODE == 5) { nt = __builtin_amdgcn_readfirstlane(nt); u = nt < per64 ? nt * 64u + c64 : nunits; }
    if (MODE == 4) { nt = __builtin_amdgcn_readfirstlane(nt); u += W; if (nt == 0xFFFFFFFFu) u = nunits; }
    if (MODE == 1) { nt = __builtin_amdgcn_readfirstlane(nt); u = nt < per ? nt * 8u + xcc : nunits; }
    if (MODE == 2) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(nt)); u = nt < per ? nt * 8u + xcc : nunits; }
  }
  if (acc == 0x12345678u) atomicAdd(out, 1ull);
}

int main(int argc, char** argv) {
  const uint64_t len = (argc > 1 ? atoll(argv[1]) : 4) << 30;
  const bool check = argc > 2 && atoi(argv[2]) != 0;
  uint8_t* hay; CK(hipMalloc(&hay, len)); CK(hipMemset(hay, 0x5A, len));
  const uint32_t nunits = static_cast<uint32_t>(len / (9 * 4096));
  uint32_t *ctr, *seen; unsigned long long* out;
  CK(hipMalloc(&ctr, 64 * 32 * 4)); CK(hipMalloc(&seen, nunits * 4ull)); CK(hipMalloc(&out, 8));
  int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int G = cus * 6;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 6; mode++) {
    float best = 1e9;
    bool ok = true;
    for (int it = 0; it < 5; it++) {
      CK(hipMemset(ctr, 0, 64 * 32 * 4)); CK(hipMemset(seen, 0, nunits * 4ull));
      CK(hipDeviceSynchronize());
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k_stream<0>, dim3(G), dim3(256), 0, 0, hay, len, ctr, nunits, check ? seen : nullptr, out);
      if (mode == 1) hipLaunchKernelGGL(k_stream<1>, dim3(G), dim3(256), 0, 0, hay, len, ctr, nunits, check ? seen : nullptr, out);
      if (mode == 2) hipLaunchKernelGGL(k_stream<2>, dim3(G), dim3(256), 0, 0, hay, len, ctr, nunits, check ? seen : nullptr, out);
      if (mode == 3) hipLaunchKernelGGL(k_stream<3>, dim3(G), dim3(256), 0, 0, hay, len, ctr, nunits, check ? seen : nullptr, out);
      if (mode == 5) hipLaunchKernelGGL(k_stream<5>, dim3(G), dim3(256), 0, 0, hay, len, ctr, nunits, check ? seen : nullptr, out);
      if (mode == 4) hipLaunchKernelGGL(k_stream<4>, dim3(G), dim3(256), 0, 0, hay, len, ctr, nunits, check ? seen : nullptr, out);
      hipEventRecord(e1); CK(hipEventSynchronize(e1));
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      if ((mode == 1 || mode == 2 || mode == 5) && check) {
        std::vector<uint32_t> h(nunits); CK(hipMemcpy(h.data(), seen, nunits * 4ull, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < nunits; i++) if (h[i] != 1) { if (ok) printf("  mode %d: unit %u seen %u times\n", mode, i, h[i]); ok = false; }
      }
    }
    printf("mode %d (%s): best %.4f ms = %.1f GB/s, every unit exactly once: %s\n", mode, mode == 0 ? "static" : mode == 1 ? "vector atomic" : mode == 2 ? "scalar atomic" : mode == 3 ? "static, XCD-interleaved units" : mode == 4 ? "static + a dummy vector atomic per unit" : "vector atomic, 64 counters by workgroup", best, len / best / 1e6, mode == 0 ? "-" : ok ? "yes" : "NO");
  }
  return 0;
}
