// scan_digit_flat.hip — UseDigitPrefilter FindAll, second kernel generation.
//
// Same ownership rule, tables and output protocol as k_scan_dfa<digit> (scan_dfa.hip); what changes
// is how a wave spends its cycles.  In the nested-loop form every candidate costs the wave
// max-over-lanes(scan length) + max-over-lanes(verify length) dependent LDS round trips.  Here
//   * the digit prefilter is computed once per tile, bit-parallel: while a thread stages its
//     16-byte vectors it also packs "is ASCII digit" into a 16-bit mask (SWAR compare + multiply
//     gather), giving a bitmap with one bit per haystack byte in LDS.  "Next digit at >= pos" and
//     "end of this digit run" are then a 64-bit load, a shift and a count-trailing-zeros
//     (the GPU counterpart of memchrDigitAVX2's compare/movemask/BSF, simd/memchr_digit_amd64.s:26);
//   * the lane walk (walk.hpp lane_digit_flat) is one flat loop, one DFA transition per live lane
//     per iteration, with the next haystack byte fetched speculatively so the only serial dependency
//     is the transition-table read.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_common.hpp"
#include "scan_dfa.h"
#include "walk.hpp"

namespace cxgdev {

namespace {

constexpr int kLdsStride = kChunk + 4;
constexpr int kLdsTileBytes = (kThreads + kHaloChunks) * kLdsStride;
constexpr int kRowStride = 260;
constexpr int kWords = kThreads + kHaloChunks;
constexpr int kRecCap2 = 512;

__device__ __forceinline__ int lds_pad2(int r) { return r + (r >> 6) * 4; }

struct FlatMem {
  const uint8_t* lds;
  const uint64_t* bits;
  const uint8_t* g;
  int32_t lim;
  int32_t flag_at = 0x7FFFFFFF;   // serial-walk cut (scan_dfa.h walk_limit)
  mutable uint32_t over = 0;
  mutable uint32_t slow = 0;    // one-byte reads beyond the staged window (scan_dfa.h kSerialReads)
  __device__ __forceinline__ uint32_t byte(int32_t r) const {
    if (static_cast<uint32_t>(r) < static_cast<uint32_t>(lim)) return lds[lds_pad2(r)];
    over |= static_cast<uint32_t>(r >= flag_at);
    if (++slow > kSerialReads) { over = 1u; return 0u; }
    return g[r];
  }
  __device__ __forceinline__ uint64_t digits(int32_t w) const { return bits[w]; }
  __device__ __forceinline__ int32_t bitmap_limit() const { return lim; }
};

struct RecSink2 {
  uint32_t* recs;
  uint32_t* rec_count;
  uint32_t lane;
  uint32_t n;
  __device__ __forceinline__ void emit(int32_t s, int32_t e) {
    const uint32_t j = n++;
    const uint32_t slot = atomicAdd(rec_count, 1u);
    if (slot < static_cast<uint32_t>(kRecCap2)) {
      recs[slot * 3 + 0] = static_cast<uint32_t>(s);
      recs[slot * 3 + 1] = static_cast<uint32_t>(e);
      recs[slot * 3 + 2] = (lane << 16) | (j & 0xFFFFu);
    }
  }
};

struct DirectSink2 {
  int64_t* out;
  uint64_t cap;
  uint64_t first;
  int64_t origin;
  uint32_t n;
  __device__ __forceinline__ void emit(int32_t s, int32_t e) {
    const uint64_t row = first + n++;
    if (row < cap) {
      longlong2 v; v.x = origin + s; v.y = origin + e;
      store_pair_nt(out + row * 2, v.x, v.y);
    }
  }
};

// 4 bytes -> 4 bits (bit k set iff byte k is '0'..'9')
__device__ __forceinline__ uint32_t digit_bits4(uint32_t x) {
  const uint32_t t = digit_mask4(x) >> 7;          // bits 0, 8, 16, 24
  return ((t * 0x00204081u) >> 21) & 0xFu;         // gather (no carries: partial products are disjoint)
}

}  // namespace

__global__ __launch_bounds__(kThreads) void k_scan_digit_flat(ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];   // DFA table + info
  __shared__ __attribute__((aligned(16))) uint8_t s_tile[kLdsTileBytes];
  __shared__ __attribute__((aligned(16))) uint64_t s_bits[kWords];
  __shared__ uint32_t s_recs[kRecCap2 * 3];
  __shared__ uint32_t s_cnt[kThreads];
  __shared__ uint32_t s_wsum[4];
  __shared__ uint32_t s_rec_count;
  __shared__ uint32_t s_tile_id;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x;
  const uint64_t t0 = a.prof ? clock64() : 0;
  if (tid == 0) {
    s_tile_id = static_cast<uint32_t>(claim_tile(a.ticket, a.ntiles));
    s_rec_count = 0;
  }
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(a.blob);
  const uint32_t fwd_states = h->fwd_states;
  uint8_t* s_fwd = s_dyn;
  uint8_t* s_info = s_fwd + fwd_states * kRowStride;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.blob + h->fwd_off);
    for (uint32_t i = tid; i < fwd_states * 64u; i += kThreads)
      *reinterpret_cast<uint32_t*>(s_fwd + (i >> 6) * kRowStride + (i & 63u) * 4u) = src[i];
    if (tid < 64) reinterpret_cast<uint32_t*>(s_info)[tid] = reinterpret_cast<const uint32_t*>(a.blob + h->info_off)[tid];
  }
  __syncthreads();
  const uint64_t t1 = a.prof ? clock64() : 0;
  const uint64_t tile = s_tile_id;
  if (tile >= a.ntiles) return;
  const uint64_t tile_lo = tile * static_cast<uint64_t>(kTile);
  const uint64_t remaining = a.len - tile_lo;
  const WalkLimit wl = walk_limit(remaining, kTile + kHalo);   // serial-walk budget, scan_dfa.h
  const int32_t rend = wl.rend;
  const int32_t stage = rend < kTile + kHalo ? rend : kTile + kHalo;
  const uint8_t* g = a.hay + tile_lo;
  {
    // All of a thread's global loads are issued before any of them is consumed, so a wave keeps
    // 5 x 1 KiB in flight instead of one (HBM latency ~2 us; MI355X_MICROARCH "keep >= 8 loads per lane").
    uint16_t* pieces = reinterpret_cast<uint16_t*>(s_bits);
    const int nfull = stage >> 4;
    constexpr int kIter = (kWords * 4 + kThreads - 1) / kThreads;   // 5
    uint4 x[kIter];
#pragma unroll
    for (int k = 0; k < kIter; k++) {
      const int v = tid + k * kThreads;
      x[k] = (v < nfull) ? *reinterpret_cast<const uint4*>(g + (static_cast<size_t>(v) << 4)) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < kIter; k++) {
      const int v = tid + k * kThreads;
      if (v >= kWords * 4) break;
      uint32_t mask = 0;
      if (v < nfull) {
        uint32_t* d = reinterpret_cast<uint32_t*>(s_tile + lds_pad2(v << 4));
        d[0] = x[k].x; d[1] = x[k].y; d[2] = x[k].z; d[3] = x[k].w;
        mask = digit_bits4(x[k].x) | (digit_bits4(x[k].y) << 4) | (digit_bits4(x[k].z) << 8) | (digit_bits4(x[k].w) << 12);
      } else if (v == nfull) {
        const int base = v << 4;
        for (int j = 0; base + j < stage; j++) {
          const uint32_t b = g[base + j];
          s_tile[lds_pad2(base + j)] = static_cast<uint8_t>(b);
          mask |= (is_digit(b) ? 1u : 0u) << j;
        }
      }
      pieces[v] = static_cast<uint16_t>(mask);
    }
  }
  __syncthreads();
  const uint64_t t2 = a.prof ? clock64() : 0;

  FlatMem m{s_tile, s_bits, g, stage};
  m.flag_at = wl.flag_at;
  DfaView fv{s_fwd, kRowStride, h->fwd_start, h->fwd_first_accept};
  const bool skip_safe = (h->flags & kFlagRunSkip) != 0;
  const int32_t c0 = tid * kChunk, c1 = c0 + kChunk;
  const bool at_origin = (tile_lo == 0 && tid == 0);

  RecSink2 sink{s_recs, &s_rec_count, static_cast<uint32_t>(tid), 0u};
  if (!(a.dbg & 1u)) lane_digit_flat(m, fv, s_info, skip_safe, c0, c1, rend, at_origin, sink);
  if (m.over) raise_err(a.err, kErrSerialLimit);
  // a lane may emit more than 65 535 matches (no synchronising byte for a long stretch); the 16-bit rank in a
  // buffered record is only read when the whole tile emitted <= the record capacity, so that is not an error
  const uint64_t t3 = a.prof ? clock64() : 0;

  uint32_t total;
  const uint32_t excl = block_exclusive_scan(sink.n, s_wsum, total);
  s_cnt[tid] = excl;
  const uint64_t t4 = a.prof ? clock64() : 0;
  if (a.dbg & 2u) {
    if (tid == 0) { s_base = atomicAdd(reinterpret_cast<unsigned long long*>(a.total), static_cast<unsigned long long>(total)); }
    __syncthreads();
  } else {
    tile_lookback(a.status, a.total, a.err, tile, a.ntiles, total, &s_base);
  }
  const uint64_t t5 = a.prof ? clock64() : 0;
  if (a.prof && (tid & 63) == 0) {   // per wave: phase cycles (stage tables, stage tile, walk, scan, look-back)
    atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 0), static_cast<unsigned long long>(t1 - t0));
    atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 1), static_cast<unsigned long long>(t2 - t1));
    atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 2), static_cast<unsigned long long>(t3 - t2));
    atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 3), static_cast<unsigned long long>(t4 - t3));
    atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 4), static_cast<unsigned long long>(t5 - t4));
    atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 5), 1ull);
  }
  const uint64_t base = s_base;
  const int64_t origin = a.base + static_cast<int64_t>(tile_lo);
  if (a.out == nullptr) return;
  if (total <= static_cast<uint32_t>(kRecCap2)) {
    for (uint32_t i = tid; i < total; i += kThreads) {
      const uint32_t key = s_recs[i * 3 + 2];
      const uint64_t row = base + s_cnt[key >> 16] + (key & 0xFFFFu);
      if (row < a.cap) {
        longlong2 v;
        v.x = origin + static_cast<int32_t>(s_recs[i * 3 + 0]);
        v.y = origin + static_cast<int32_t>(s_recs[i * 3 + 1]);
        store_pair_nt(a.out + row * 2, v.x, v.y);
      }
    }
  } else {
    DirectSink2 ds{a.out, a.cap, base + excl, origin, 0u};
    lane_digit_flat(m, fv, s_info, skip_safe, c0, c1, rend, at_origin, ds);
  }
}

hipError_t launch_scan_digit_flat(const ScanArgs& a, uint32_t fwd_states, hipStream_t stream) {
  const size_t dyn = static_cast<size_t>(fwd_states) * kRowStride + 256;
  hipLaunchKernelGGL(k_scan_digit_flat, dim3(static_cast<unsigned>(a.ntiles)), dim3(kThreads), dyn, stream, a);
  return hipGetLastError();
}

}  // namespace cxgdev
