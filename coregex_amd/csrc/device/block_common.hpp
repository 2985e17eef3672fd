// Block-level building blocks shared by the scan kernels: exclusive scan of per-lane match counts and
// the decoupled look-back that turns a tile's count into its global output base.
//
// Look-back protocol: one 8-byte status word per tile, {flag:2, value:62}, written and read with
// agent-scope relaxed atomics (sc1: served by L2 / memory, never a stale per-CU L1 line).  The word
// is self-contained, so no release/acquire pair is needed (MI355X_MICROARCH.md "granule").  Tiles are
// claimed through an atomic ticket, so every predecessor of a running tile has already started and
// the wait is bounded by their run time; a spin watchdog turns a protocol bug into an error flag
// instead of a hung GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cxgdev {

constexpr uint64_t kFlagAggregate = 1ull << 62;
constexpr uint64_t kFlagInclusive = 2ull << 62;
constexpr uint64_t kFlagMask = 3ull << 62;
constexpr uint32_t kSpinLimit = 1u << 22;

// Exclusive prefix of `mine` over the 256 threads of the block; `total` = block sum.
// s_wsum: 4 uint32 of LDS.  Contains one __syncthreads().
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t mine, uint32_t* s_wsum, uint32_t& total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  uint32_t wave_off = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const uint32_t ws = s_wsum[w];
    if (w < wave) wave_off += ws;
    total += ws;
  }
  return wave_off + incl - mine;
}

// Publishes this tile's count and resolves its exclusive global base into *s_base (LDS).
// Executed by wave 0; ends with __syncthreads() for the whole block.
__device__ __forceinline__ void tile_lookback(uint64_t* status, uint64_t* total_out, uint32_t* err, uint64_t tile,
                                              uint64_t ntiles, uint32_t total, uint64_t* s_base) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) {
    if (lane == 0) {
      const uint64_t word = (tile == 0 ? kFlagInclusive : kFlagAggregate) | static_cast<uint64_t>(total);
      __hip_atomic_store(status + tile, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint64_t base = 0;
    if (tile > 0) {
      int64_t look = static_cast<int64_t>(tile) - 1;   // lane l inspects tile look - l
      uint32_t spins = 0;
      for (;;) {
        const int64_t idx = look - lane;
        uint64_t w = kFlagInclusive;                   // "tiles" before 0 contribute an inclusive 0
        if (idx >= 0) w = __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ready = (w & kFlagMask) != 0;
        if (!__all(ready)) {
          if (++spins > kSpinLimit) { if (lane == 0) atomicOr(err, 2u); break; }
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        const unsigned long long incl_mask = __ballot((w & kFlagMask) == kFlagInclusive);
        const int first_incl = incl_mask ? __builtin_ctzll(incl_mask) : 64;
        uint64_t v = (lane <= first_incl) ? (w & ~kFlagMask) : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        base += v;
        if (first_incl < 64) break;
        look -= 64;
      }
      if (lane == 0)
        __hip_atomic_store(status + tile, kFlagInclusive | (base + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) {
      *s_base = base;
      if (tile == ntiles - 1) *total_out = base + total;
    }
  }
  __syncthreads();
}

}  // namespace cxgdev
