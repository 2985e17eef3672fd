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

// Error / fallback bits: the word may live in pinned host memory (wave kernels: the host reads it without a copy),
// so the OR is a system-scope atomic.  Rare path.
__device__ __forceinline__ void raise_err(uint32_t* err, uint32_t bits) {
  __hip_atomic_fetch_or(err, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// A spin watchdog ran out: error bit 1, and WHICH wait it was in bits 24..27 (capi_ladder.hip reads it to decide what to demote:
// 1 grouped look-back, 2 delimiter look-back, 3 / 4 transducer hand-off / group entry, 5 / 6 persistent kernel duty / record).
constexpr uint32_t kWdLookback = 1, kWdDelim = 2, kWdFsmExit = 3, kWdFsmEntry = 4, kWdPersDuty = 5, kWdPersRecord = 6;
__device__ __forceinline__ void raise_watchdog(uint32_t* err, uint32_t origin) { raise_err(err, 2u | (origin << 24)); }

// One output pair (16 bytes, 16-byte aligned) with a NONTEMPORAL store.  Rows are written once and not read again by the
// kernel that writes them; stored through the default write-back path of the XCD's L2 the 160 MB of rows of the headline
// workload cost 0.037 ms of a 0.26 ms launch, nontemporal 0.007 (round 4, profiles/r04_pers_64g.txt: 1 GiB 0.2626 -> 0.2323 ms,
// 64 GiB 0.646 -> 0.7015 of the HBM roofline).
__device__ __forceinline__ void store_pair_nt(int64_t* p, int64_t x, int64_t y) {
  typedef long long ll2 __attribute__((ext_vector_type(2)));
  ll2 o; o.x = x; o.y = y;
#ifdef CXG_NO_NT_ROWS
  *reinterpret_cast<ll2*>(p) = o;
#else
  __builtin_nontemporal_store(o, reinterpret_cast<ll2*>(p));
#endif
}

// The same for a compact row (cxg_find_all_device_u32): two uint32, 8 bytes.
__device__ __forceinline__ void store_pair32_nt(uint32_t* p, uint32_t x, uint32_t y) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  u2 o; o.x = x; o.y = y;
  __builtin_nontemporal_store(o, reinterpret_cast<u2*>(p));
}

constexpr uint64_t kFlagAggregate = 1ull << 62;
constexpr uint64_t kFlagInclusive = 2ull << 62;
constexpr uint64_t kFlagMask = 3ull << 62;
// Launch epoch (bits 61..52): a word counts as published only if it carries the epoch of the running launch, so
// the wave kernels need no memset of the status array between launches (epoch 0 = legacy: zeroed array).
constexpr int kEpochShift = 52;
constexpr uint64_t kEpochMask = 0x3FFull << kEpochShift;
constexpr uint64_t kValueMask = (1ull << kEpochShift) - 1ull;
constexpr uint32_t kSpinLimit = 1u << 22;

// Tile tickets, sharded per XCD.  One counter serialises at ~88 atomics/us (MI355X_MICROARCH "dequeue"),
// 0.7 ms for the 65 536 tiles of a 1 GiB scan; eight counters (one per XCD L2) cut that by 8.  Counter x
// hands out tiles x, x+8, x+16, ...; a workgroup whose counter is exhausted steals from the next one.
// Every workgroup claims exactly one tile (grid == ntiles).  Deadlock-free for the look-back: the
// smallest unclaimed tile can always be claimed by the next workgroup that starts, and every tile
// already claimed has all its predecessors' claims in progress or done... more precisely, a waiting
// workgroup only waits for smaller tiles, and the smallest unfinished tile never waits.
__device__ __forceinline__ uint64_t claim_tile(uint32_t* tickets, uint64_t ntiles) {
  const uint32_t xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11)) & 7u;
  for (uint32_t k = 0; k < 8; k++) {
    const uint32_t x = (xcc + k) & 7u;
    const uint64_t per = (ntiles + 7 - x) / 8;          // tiles owned by counter x
    const uint32_t t = atomicAdd(tickets + x, 1u);
    if (t < per) return static_cast<uint64_t>(t) * 8 + x;
  }
  return ntiles;   // cannot happen when grid == ntiles
}

// Group of a workgroup in the wave kernels.  Static mode: group = blockIdx.x, no atomic — the ~9 000 ticket atomics
// of a 1 GiB scan queue at the eight L2 counters and cost ~10 % of the kernel.  The look-back only needs "every
// smaller group is resident or finished"; workgroups are handed to each XCD in index order, so the smallest
// unfinished group always finds a free slot on its XCD (slots there are only ever held by smaller, i.e. finished
// or running, groups) and never waits.  Should a device dispatch differently, the look-back's spin watchdog raises
// error bit 1 and the host reruns the scan with tickets (capi_ladder.hip).
__device__ __forceinline__ uint64_t claim_group(bool static_groups, uint32_t* tickets, uint64_t ngroups) {
  return static_groups ? static_cast<uint64_t>(blockIdx.x) : claim_tile(tickets, ngroups);
}

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
// limit / stop (optional): FindAll with n > 0 (meta/findall.go:196).  The tile whose inclusive sum reaches `limit` stores
// epoch + 1 into *stop; kernels that look at it let later workgroups skip their scan (they publish zero rows), so a call
// for the first ten matches of 8 GiB costs the groups that were resident when the tenth was counted, not the haystack.
__device__ __forceinline__ void tile_lookback(uint64_t* status, uint64_t* total_out, uint32_t* err, uint64_t tile,
                                              uint64_t ntiles, uint32_t total, uint64_t* s_base, uint32_t epoch = 0,
                                              uint64_t limit = 0, uint32_t* stop = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t etag = static_cast<uint64_t>(epoch) << kEpochShift;
  if (wave == 0) {
    if (lane == 0) {
      const uint64_t word = (tile == 0 ? kFlagInclusive : kFlagAggregate) | etag | static_cast<uint64_t>(total);
      __hip_atomic_store(status + tile, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint64_t base = 0;
    if (tile > 0) {
      int64_t look = static_cast<int64_t>(tile) - 1;   // lane l inspects tile look - l
      uint32_t spins = 0;
      for (;;) {
        const int64_t idx = look - lane;
        uint64_t w = kFlagInclusive | etag;            // "tiles" before 0 contribute an inclusive 0
        if (idx >= 0) w = __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ready = (w & kFlagMask) != 0 && (w & kEpochMask) == etag;
        if (!__all(ready)) {
          if (++spins > kSpinLimit) { if (lane == 0) raise_watchdog(err, kWdLookback); break; }
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        const unsigned long long incl_mask = __ballot((w & kFlagMask) == kFlagInclusive);
        const int first_incl = incl_mask ? __builtin_ctzll(incl_mask) : 64;
        uint64_t v = (lane <= first_incl) ? (w & kValueMask) : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        base += v;
        if (first_incl < 64) break;
        look -= 64;
      }
      if (lane == 0)
        __hip_atomic_store(status + tile, kFlagInclusive | etag | (base + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) {
      *s_base = base;
      if (tile == ntiles - 1) *total_out = base + total;
      if (limit != 0 && stop != nullptr && base + total >= limit) __hip_atomic_store(stop, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
}


// FindAll with n > 0, at the start of a wave kernel's workgroup (kernel-argument-uniform branch, one barrier): true when the
// stop word says that `limit` rows were counted before this group started.  The group that set the word had seen every
// group in front of it published, i.e. started — so a group that sees it set at its start lies BEHIND the setter and at
// least `limit` rows precede it: it publishes exactly that as its inclusive sum (no scan, no look-back) and ends; groups
// behind it find an inclusive word at distance one.
template <class Args>
__device__ __forceinline__ bool limit_reached_skip(const Args& a, uint64_t group, uint64_t* s_tmp) {
  if (a.limit == 0) return false;
  if (threadIdx.x == 0) *s_tmp = __hip_atomic_load(a.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const bool skip = static_cast<uint32_t>(*s_tmp) == a.epoch + 1u;
  if (skip && threadIdx.x == 0) {
    __hip_atomic_store(a.status + group, kFlagInclusive | (static_cast<uint64_t>(a.epoch) << kEpochShift) | a.limit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (group == a.ngroups - 1) *a.total = a.limit;
  }
  return skip;
}

}  // namespace cxgdev
