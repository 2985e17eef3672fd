// stream_common.hpp — output ordering for PERSISTENT streaming kernels (scan_fields_wave.hip k_scan_fields_stream):
// a "scan server" workgroup turns per-wave-tile row counts into global row bases; the producing waves never wait on each other.
//
// Why not the decoupled look-back of block_common.hpp here.  Measured on the MI355X (profiles/r03_fields_ablation.txt):
// (1) reading 1 GiB with every workgroup streaming its own contiguous 120 KiB (the grouped wave kernels) tops out at
//     3.7 TB/s, the same loads over ONE dense moving window (round r, wave W reads wave-tile r * NW + W) reach 6.2-6.4 TB/s —
//     the resident workgroups must read neighbouring addresses at the same time;
// (2) with a dense window a workgroup's tiles are no longer consecutive, so rows are ordered per WAVE-TILE (280 000 units
//     per GiB), and a look-back in which every unit walks back over its predecessors reads O(units x resident waves) status
//     words when units finish together (they do: same work, same start) — the epilogue of an otherwise empty grouped
//     kernel already cost 0.13 ms per GiB.
// Here every word is written once and read once.  A producer wave publishes the row count of its tile as a 16-bit word
// cnt16[t] = {epoch:4, rows:11} and goes on with its next tile.  The four waves of a workgroup hold four consecutive tiles
// (a QUAD) in every round, so one aligned 64-bit load shows the scan server four tiles.  The server (workgroup 0, four
// waves) walks the quads in order: wave v takes batches v, v+4, ... of 1 024 quads (16 coalesced 512-byte loads), sums
// them with DPP prefix sums once all are published, receives the running base of its batch from the wave in front through
// LDS (a ~0.2 us hop: the only serial step, 4 096 tiles each), passes it on, and publishes one 64-bit word per quad
// base4[q] = {ready, epoch:10, rows in front of the quad}.  A producer reads its quad's base a few tiles later, adds the
// counts of the lower waves of its own workgroup (kept in LDS) and flushes the rows it held back in its LDS ring.
// (First version: 64-bit count words, one per tile, one server wave: 1.39 ms per GiB — the server managed 200 tiles/us
// where 1 900 are needed.)
// All words move with relaxed agent-scope atomics (sc1: served by L2 / memory, never a stale per-CU line) and are
// self-contained; base words carry the launch epoch (block_common.hpp kEpochShift), count words a 4-bit epoch of their
// own array (capi.hip zeroes it every 15 launches), so no memset runs between launches.
// Deadlock freedom needs every producer to be resident (persistent grid = the device's capacity, capi.hip sizes it) — the
// server waits for producers in tile order, producers wait for the server only when their row ring is full or at the end.
// Every wait is bounded by a spin watchdog that raises error bit 1 (the host reruns the scan on the grouped kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_common.hpp"
#include "wave_common.hpp"

namespace cxgdev {

constexpr uint64_t kStreamReady = 1ull << 62;
#ifndef CXG_SCAN_GROUPS
#define CXG_SCAN_GROUPS 8
#endif
constexpr int kScanGroups = CXG_SCAN_GROUPS;                       // server: groups of 64 quads per batch and wave (16: spills into scratch under the 64-VGPR budget of the producers)
constexpr uint32_t kCntRowsMask = 0x7FFu;             // cnt16: rows in bits 0..10, epoch (1..15) in bits 11..14

__device__ __forceinline__ void stream_publish_count(uint16_t* cnt16, uint64_t t, uint32_t rows, uint32_t epoch4) {
  __hip_atomic_store(cnt16 + t, static_cast<uint16_t>((epoch4 << 11) | rows), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Wave-uniform: the base of unit q if the server has published it.
__device__ __forceinline__ bool stream_try_base(const uint64_t* baseu, uint64_t q, uint32_t epoch, uint64_t& base) {
  const uint64_t w = __hip_atomic_load(baseu + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w)));
  const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w >> 32)));
  const uint64_t u = (static_cast<uint64_t>(hi) << 32) | lo;
  base = u & kValueMask;
  return (u & kFlagMask) == kStreamReady && (u & kEpochMask) == (static_cast<uint64_t>(epoch) << kEpochShift);
}

struct StreamChain { uint32_t seq[8]; uint64_t base[8]; };   // LDS of a server workgroup: running base handed from batch to batch
constexpr int kScanWorkgroups = 4;                    // server workgroups (blockIdx.x < kScanWorkgroups): 16 waves
constexpr uint32_t kScanFastPolls = 256;              // full-batch polls before a wave publishes its batch piecemeal

// The scan server: kScanWorkgroups workgroups of four waves.  nunits units (a unit = the consecutive wave-tiles one producer
// wave takes per round; its 16-bit count word holds the rows of all of them), quads q = unit / 4, batches of
// kScanGroups * 64 quads, SUPER-BATCHES of four batches.  Workgroup s takes super-batches s, s + S, ...; its wave v the
// v-th batch of each.  The running base travels from batch to batch through LDS inside a super-batch (~0.2 us per hop) and
// from super-batch to super-batch through one epoch-tagged word in HBM (gchain[], ~1-2 us per hop).
// A wave first polls its whole batch (all loads in flight at once) and, when every count is there, publishes the batch
// total before anything else — the waves behind it never wait for its stores.  When the batch does not complete (a small
// grid: the producers themselves wait for bases of this very batch before they can publish more), the wave takes its base
// first and publishes group by group: the earliest unpublished quad never waits on anything later, so the server cannot
// deadlock with the producers.  Output: baseu[unit] = {ready, epoch, rows in front of the unit}, four words per lane
// (two 16-byte stores).
__device__ __forceinline__ void stream_scanner(const uint16_t* cnt16, uint64_t* baseu, uint64_t* gchain, uint64_t nunits, uint32_t epoch4, uint32_t epoch,
                                               uint64_t* total_out, uint32_t* err, StreamChain* chain) {
  const int lane = threadIdx.x & 63;
  const int v = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const uint64_t etag = kStreamReady | (static_cast<uint64_t>(epoch) << kEpochShift);
  const uint64_t* cw = reinterpret_cast<const uint64_t*>(cnt16);
  const uint64_t nquads = (nunits + 3) >> 2;
  constexpr uint64_t kBatch = static_cast<uint64_t>(kScanGroups) * 64;
  const uint64_t nbatches = (nquads + kBatch - 1) / kBatch;
  if (threadIdx.x < 8) { chain->seq[threadIdx.x] = 0u; chain->base[threadIdx.x] = 0ull; }
  __syncthreads();
#ifndef CXG_SCAN_NOPRIO
  __builtin_amdgcn_s_setprio(3);                                      // the server's waves share their SIMDs with seven producer waves each: they go first
#endif
  const uint64_t e4x4 = static_cast<uint64_t>(epoch4) * 0x0800080008000800ull;   // the epoch in all four fields
  constexpr uint64_t kEpochBits = 0x7800780078007800ull;
  auto quad_rows = [](uint64_t q) -> uint32_t {
    const uint32_t lo = static_cast<uint32_t>(q), hi = static_cast<uint32_t>(q >> 32);
    return (lo & kCntRowsMask) + ((lo >> 16) & kCntRowsMask) + (hi & kCntRowsMask) + ((hi >> 16) & kCntRowsMask);
  };
  // the four unit bases of quad q (counts w, rows in front of the quad b): two 16-byte stores, words behind nunits dropped
  const __amdgpu_buffer_rsrc_t rbase = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t*>(baseu), 0,
                                                                         static_cast<int>(nunits < 0x0FFFFFFFull ? nunits * 8 : 0x7FFFFFF8ull), 0x00020000);
  auto put_quad = [&](uint64_t q, uint64_t w, uint64_t b) {
    const uint32_t lo = static_cast<uint32_t>(w), hi = static_cast<uint32_t>(w >> 32);
    const uint64_t b0 = etag | b, b1 = b0 + (lo & kCntRowsMask), b2 = b1 + ((lo >> 16) & kCntRowsMask), b3 = b2 + (hi & kCntRowsMask);
    const u32x4 s0 = {static_cast<uint32_t>(b0), static_cast<uint32_t>(b0 >> 32), static_cast<uint32_t>(b1), static_cast<uint32_t>(b1 >> 32)};
    const u32x4 s1 = {static_cast<uint32_t>(b2), static_cast<uint32_t>(b2 >> 32), static_cast<uint32_t>(b3), static_cast<uint32_t>(b3 >> 32)};
    const uint32_t off = static_cast<uint32_t>(q) * 32u;              // (q < 2^26: checked by the host through nunits)
    __builtin_amdgcn_raw_buffer_store_b128(s0, rbase, off, 0, 16 /*sc1*/);
    __builtin_amdgcn_raw_buffer_store_b128(s1, rbase, off + 16u, 0, 16);
  };
  // running base in front of batch b: 0, the word of its super-batch (first batch of a super-batch), else this workgroup's LDS
  auto get_base = [&](uint64_t b, uint64_t& bbase) -> bool {
    uint32_t spins = 0;
    if (b == 0) { bbase = 0; return true; }
    if ((b & 3ull) == 0) {
      for (;;) {
        const uint64_t w = __hip_atomic_load(gchain + (b >> 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t u = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w >> 32)))) << 32) |
                           static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w)));
        if ((u & (kFlagMask | kEpochMask)) == etag) { bbase = u & kValueMask; return true; }
        if (++spins > kSpinLimit) { if (lane == 0) raise_err(err, 2u); return false; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    while (__hip_atomic_load(&chain->seq[b & 7], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != static_cast<uint32_t>(b + 1)) {
      if (++spins > kSpinLimit) { if (lane == 0) raise_err(err, 2u); return false; }
      __builtin_amdgcn_s_sleep(1);
    }
    bbase = chain->base[b & 7];
    return true;
  };
  auto put_base = [&](uint64_t b, uint64_t bbase) {                  // ... in front of batch b (the end of batch b - 1)
    if (lane != 0) return;
    if ((b & 3ull) == 0) __hip_atomic_store(gchain + (b >> 2), etag | bbase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else {
      chain->base[b & 7] = bbase;
      __hip_atomic_store(&chain->seq[b & 7], static_cast<uint32_t>(b + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  for (uint64_t sb = blockIdx.x; sb * 4 < nbatches; sb += kScanWorkgroups) {
    const uint64_t b = sb * 4 + static_cast<uint64_t>(v);
    if (b >= nbatches) break;
    const uint64_t pos = b * kBatch;
    bool piecemeal = b + 1 == nbatches;                              // the last batch: bounds, a partial last quad
    if (!piecemeal) {
      // ---- a full batch inside the array: all groups at once
      const uint64_t* src = cw + pos + lane;
      uint64_t w[kScanGroups];
      uint32_t polls = 0;
      for (;;) {
#pragma unroll
        for (int i = 0; i < kScanGroups; i++) w[i] = __hip_atomic_load(src + i * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t diff = 0;
#pragma unroll
        for (int i = 0; i < kScanGroups; i++) diff |= w[i] ^ e4x4;
        if (__ballot((diff & kEpochBits) == 0ull) == ~0ull) break;    // every count of the batch is published
        if (++polls > kScanFastPolls) { piecemeal = true; break; }
        __builtin_amdgcn_s_sleep(8);
      }
      if (!piecemeal) {
        // rows of the whole batch first (one wave-wide sum): the wave behind can start as soon as this one knows its base
        uint32_t lane_rows = 0;
#pragma unroll
        for (int i = 0; i < kScanGroups; i++) lane_rows += quad_rows(w[i]);
        const uint32_t btotal = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wave_inclusive_sum(lane_rows)), 63));   // < 2^23
        uint64_t bbase = 0;
        if (!get_base(b, bbase)) return;
        put_base(b + 1, bbase + btotal);
        // the bases of the batch's units: one prefix sum per group of 64 quads
        uint64_t run = bbase;
#pragma unroll
        for (int i = 0; i < kScanGroups; i++) {
          const uint32_t sq = quad_rows(w[i]);
          const uint32_t incl = wave_inclusive_sum(sq);
          put_quad(pos + static_cast<uint64_t>(i) * 64 + lane, w[i], run + incl - sq);
          run += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
        }
      }
    }
    if (piecemeal) {
      // ---- group by group, with bounds (the last quad may hold fewer than four units)
      uint64_t run = 0;
      if (!get_base(b, run)) return;
      const uint64_t end = pos + kBatch < nquads ? pos + kBatch : nquads;
      for (uint64_t g = pos; g < end; g += 64) {
        const uint64_t q = g + lane;
        uint64_t w = e4x4;                                           // behind the last quad: published, no rows
        uint32_t spins = 0;
        for (;;) {
          uint64_t m = kEpochBits;
          if (q < nquads) {
            w = __hip_atomic_load(cw + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (q * 4 + 4 > nunits) { const uint32_t nv = static_cast<uint32_t>(nunits - q * 4); m >>= 16 * (4 - nv); w &= (1ull << (16 * nv)) - 1ull; }
          }
          if (__ballot(((w ^ e4x4) & m) == 0ull) == ~0ull) break;
          if (++spins > (kSpinLimit >> 2)) { if (lane == 0) raise_err(err, 2u); return; }
          __builtin_amdgcn_s_sleep(8);
        }
        const uint32_t sq = quad_rows(w);
        const uint32_t incl = wave_inclusive_sum(sq);
        if (q < nquads) put_quad(q, w, run + incl - sq);
        run += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
      }
      if (b + 1 < nbatches) put_base(b + 1, run);
      else if (lane == 0) *total_out = run;
    }
  }
}

}  // namespace cxgdev
