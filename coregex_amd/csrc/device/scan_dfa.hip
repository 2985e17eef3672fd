// scan_dfa.hip — FindAll kernels for the table-DFA strategies (UseDigitPrefilter, UseDFA direct).
//
// Geometry (gfx950, wave64): one 256-thread workgroup scans one 16 KiB tile; lane t owns the 64-byte
// chunk [64t, 64t+64) of the tile (walk.hpp explains ownership).  The tile plus a 256-byte halo is
// pulled from HBM once with coalesced 16-byte loads (1 KiB per wave instruction) and staged in LDS
// with a 68-byte lane stride — an odd number of dwords, so the 32 lanes of a ds_read_u8/b32 lane
// group that sit at the same offset of their chunks land on 32 different banks.  The DFA table
// (state x 256 bytes -> next state, u8, row stride 260 B for the same reason) and the 256-byte
// byte-info table are copied to LDS beside it; every per-byte step of the reference's table walk
// (dfa/lazy/lazy.go:261-268) is then two LDS reads and no HBM traffic.
//
// Output order: FindAll results must be sorted by start.  Each lane counts its matches, a block scan
// turns counts into ranks, and the tile's base comes from a decoupled look-back over 8-byte
// {flag,value} status words (single-word publish: no fence needed, MI355X_MICROARCH "granule").
// Tiles are handed out by an atomic ticket so a tile's predecessors have always started.
// Matches are buffered in LDS and scattered as int64 pairs (16-byte stores) at base+rank.
//
// Roofline: HBM-bound.  Algorithmic bytes per tile = 16 KiB read + 16 B per match.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_common.hpp"
#include "scan_dfa.h"
#include "walk.hpp"

namespace cxgdev {

namespace {

constexpr int kLdsStride = kChunk + 4;                        // 68 B: 17 dwords
constexpr int kLdsTileBytes = (kThreads + kHaloChunks) * kLdsStride;
constexpr int kRowStride = 260;                               // DFA table row stride in LDS

__device__ __forceinline__ int lds_pad(int r) { return r + (r >> 6) * 4; }

struct LdsMem {
  const uint8_t* lds;   // staged tile (+halo)
  const uint8_t* g;     // hay + tile_lo
  int32_t lim;          // bytes [0, lim) are staged
  int32_t flag_at = 0x7FFFFFFF;   // reading this byte or beyond means the walk ran into the serial-walk cut
  mutable uint32_t over = 0;
  mutable uint32_t slow = 0;    // bytes this lane read from HBM / L2 one at a time (~1.6 us each)
  __device__ __forceinline__ uint32_t byte(int32_t r) const {
    if (static_cast<uint32_t>(r) < static_cast<uint32_t>(lim)) return lds[lds_pad(r)];
    over |= static_cast<uint32_t>(r >= flag_at);
    // Total budget of a lane (scan_dfa.h kSerialReads): a program whose every search runs to the end of a stretch without
    // synchronising bytes (`\D+?xyz|bc` on text without digits: the lazy alternative stays alive, as in the reference) walks
    // matches x stretch bytes — minutes for 80 KB, with the other workgroups spinning in the look-back until their watchdog
    // fires.  Past the budget the call is refused (kErrSerialLimit -> CXG_E_INPUT) and the remaining reads cost nothing.
    if (++slow > kSerialReads) { over = 1u; return 0u; }
    return g[r];
  }
  __device__ __forceinline__ uint32_t dword(int32_t r) const {
    return *reinterpret_cast<const uint32_t*>(lds + lds_pad(r));
  }
  __device__ __forceinline__ int32_t wide_limit(int32_t x) const { return x < lim ? x : lim; }
};

struct RecSink {
  uint32_t* recs;        // [kRecCap][3]
  uint32_t* rec_count;
  uint32_t lane;
  uint32_t n;
  uint32_t max_len = 0, long_hit = 0;
  __device__ __forceinline__ void emit(int32_t s, int32_t e) {
    long_hit |= static_cast<uint32_t>(max_len != 0 && static_cast<uint32_t>(e - s) > max_len);
    const uint32_t j = n++;
    const uint32_t slot = atomicAdd(rec_count, 1u);
    if (slot < static_cast<uint32_t>(kRecCap)) {
      recs[slot * 3 + 0] = static_cast<uint32_t>(s);
      recs[slot * 3 + 1] = static_cast<uint32_t>(e);
      recs[slot * 3 + 2] = (lane << 16) | (j & 0xFFFFu);
    }
  }
};

struct DirectSink {   // overflow path: ranks are known, write straight to HBM
  int64_t* out;
  uint64_t cap;
  uint32_t width;
  uint64_t first;       // tile base + this lane's exclusive rank
  int64_t origin;       // absolute offset of relative position 0
  uint32_t n;
  __device__ __forceinline__ void emit(int32_t s, int32_t e) {
    const uint64_t row = first + n++;
    if (out && row < cap) {
      out[row * width + 0] = origin + s;
      out[row * width + 1] = origin + e;
    }
  }
};

template <int KIND, class Sink>
__device__ __forceinline__ void run_lane(const LdsMem& m, const DfaView& f, const DfaView& r, const uint8_t* info,
                                         bool skip_safe, int32_t c0, int32_t c1, int32_t rend, bool at_origin,
                                         Sink& sink) {
  if (KIND == kKindDigit) lane_digit(m, f, info, skip_safe, c0, c1, rend, at_origin, sink);
  else lane_bidir(m, f, r, info, c0, c1, rend, at_origin, sink);
}

}  // namespace

template <int KIND>
__global__ __launch_bounds__(kThreads) void k_scan_dfa(ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];   // DFA tables + info
  __shared__ __attribute__((aligned(16))) uint8_t s_tile[kLdsTileBytes];
  __shared__ uint32_t s_recs[kRecCap * 3];
  __shared__ uint32_t s_cnt[kThreads];
  __shared__ uint32_t s_wsum[kThreads / 64];
  __shared__ uint32_t s_rec_count;
  __shared__ uint32_t s_tile_id;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x;
  if (tid == 0) {
    s_tile_id = static_cast<uint32_t>(claim_tile(a.ticket, a.ntiles));
    s_rec_count = 0;
  }
  // ---- stage the program: tables with skewed rows, then the info bytes
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(a.blob);
  const uint32_t fwd_states = h->fwd_states, rev_states = (KIND == kKindBidir) ? h->rev_states : 0u;
  uint8_t* s_fwd = s_dyn;
  uint8_t* s_rev = s_fwd + fwd_states * kRowStride;
  uint8_t* s_info = s_rev + rev_states * kRowStride;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.blob + h->fwd_off);
    for (uint32_t i = tid; i < fwd_states * 64u; i += kThreads)
      *reinterpret_cast<uint32_t*>(s_fwd + (i >> 6) * kRowStride + (i & 63u) * 4u) = src[i];
    if (KIND == kKindBidir) {
      const uint32_t* srcr = reinterpret_cast<const uint32_t*>(a.blob + h->rev_off);
      for (uint32_t i = tid; i < rev_states * 64u; i += kThreads)
        *reinterpret_cast<uint32_t*>(s_rev + (i >> 6) * kRowStride + (i & 63u) * 4u) = srcr[i];
    }
    if (tid < 64) reinterpret_cast<uint32_t*>(s_info)[tid] = reinterpret_cast<const uint32_t*>(a.blob + h->info_off)[tid];
  }
  __syncthreads();
  const uint64_t tile = s_tile_id;
  if (tile >= a.ntiles) return;   // uniform
  const uint64_t tile_lo = tile * static_cast<uint64_t>(kTile);
  const uint64_t remaining = a.len - tile_lo;
  const WalkLimit wl = walk_limit(remaining, kTile + kHalo);   // serial-walk budget, scan_dfa.h
  const int32_t rend = wl.rend;
  const int32_t stage = rend < kTile + kHalo ? rend : kTile + kHalo;
  const uint8_t* g = a.hay + tile_lo;
  // ---- stage the tile: coalesced 16-byte loads, 4 ds_write_b32 each
  {
    const int nvec = stage >> 4;
    for (int v = tid; v < nvec; v += kThreads) {
      const uint4 x = *reinterpret_cast<const uint4*>(g + (static_cast<size_t>(v) << 4));
      uint32_t* d = reinterpret_cast<uint32_t*>(s_tile + lds_pad(v << 4));
      d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
    }
    const int tail0 = nvec << 4;
    if (tid < stage - tail0) s_tile[lds_pad(tail0 + tid)] = g[tail0 + tid];
  }
  __syncthreads();

  LdsMem m{s_tile, g, stage};
  m.flag_at = wl.flag_at;
  DfaView fv{s_fwd, kRowStride, h->fwd_start, h->fwd_first_accept};
  DfaView rv{s_rev, kRowStride, h->rev_start, h->rev_first_accept};
  const bool skip_safe = (h->flags & kFlagRunSkip) != 0;
  const int32_t c0 = tid * kChunk, c1 = c0 + kChunk;
  const bool at_origin = (tile_lo == 0 && tid == 0);

  RecSink sink{s_recs, &s_rec_count, static_cast<uint32_t>(tid), 0u};
  sink.max_len = a.max_len;
  run_lane<KIND>(m, fv, rv, s_info, skip_safe, c0, c1, rend, at_origin, sink);
  if (m.over) raise_err(a.err, kErrSerialLimit);
  if (sink.long_hit) raise_err(a.err, kErrLongMatch);
  // a lane may emit more than 65 535 matches (no synchronising byte for a long stretch); the 16-bit rank in a
  // buffered record is only read when the whole tile emitted <= the record capacity, so that is not an error

  // ---- ranks: block exclusive scan of per-lane counts, then the tile's base by decoupled look-back
  uint32_t total;
  const uint32_t excl = block_exclusive_scan(sink.n, s_wsum, total);
  s_cnt[tid] = excl;
  tile_lookback(a.status, a.total, a.err, tile, a.ntiles, total, &s_base);
  __syncthreads();
  const uint64_t base = s_base;
  const int64_t origin = a.base + static_cast<int64_t>(tile_lo);

  if (a.out == nullptr) return;
  if (total <= static_cast<uint32_t>(kRecCap)) {
    // scatter buffered records at base + rank, 16-byte stores
    for (uint32_t i = tid; i < total; i += kThreads) {
      const uint32_t key = s_recs[i * 3 + 2];
      const uint64_t row = base + s_cnt[key >> 16] + (key & 0xFFFFu);
      if (row < a.cap) {
        longlong2 v;
        v.x = origin + static_cast<int32_t>(s_recs[i * 3 + 0]);
        v.y = origin + static_cast<int32_t>(s_recs[i * 3 + 1]);
        store_pair_nt(a.out + row * a.row_width, v.x, v.y);   // row_width is even: 16-byte aligned
      }
    }
  } else {
    // LDS record buffer overflowed (dense matches): walk again, writing at the now-known ranks
    DirectSink ds{a.out, a.cap, a.row_width, base + excl, origin, 0u};
    run_lane<KIND>(m, fv, rv, s_info, skip_safe, c0, c1, rend, at_origin, ds);
  }
}

size_t scan_dfa_dynamic_lds(uint32_t fwd_states, uint32_t rev_states) {
  return static_cast<size_t>(fwd_states + rev_states) * kRowStride + 256;
}

hipError_t launch_scan_dfa(uint32_t kind, const ScanArgs& a, uint32_t fwd_states, uint32_t rev_states, hipStream_t stream) {
  const size_t dyn = scan_dfa_dynamic_lds(fwd_states, kind == kKindBidir ? rev_states : 0);
  const dim3 grid(static_cast<unsigned>(a.ntiles)), block(kThreads);
  if (kind == kKindDigit) hipLaunchKernelGGL(k_scan_dfa<kKindDigit>, grid, block, dyn, stream, a);
  else hipLaunchKernelGGL(k_scan_dfa<kKindBidir>, grid, block, dyn, stream, a);
  return hipGetLastError();
}

}  // namespace cxgdev
