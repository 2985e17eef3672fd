// scan_teddy.hip — FindAll for UseTeddy (exact literal alternation, Slim Teddy: <= 32 literals).
//
// Phase 1 is the reference's nibble-shuffle candidate scan (prefilter/teddy_ssse3_amd64.s:273) made
// bit-parallel: every thread looks up AB[byte] for the 16-byte vectors it loaded (coalesced) and ANDs
// the first-byte mask of position i with the second-byte mask of position i+1, leaving one candidate
// bit per haystack byte in LDS.  Phase 2 is the owner walk of walk.hpp lane_teddy: next candidate by
// bit scan, exact literal compare against the bytes (L2-resident, rare), FindAll advance.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_common.hpp"
#include "scan_dfa.h"
#include "walk.hpp"

namespace cxgdev {

namespace {
constexpr int kWords = kThreads + kHaloChunks;
constexpr int kRecCapT = 1024;
constexpr int kAuxMax = 4096;

struct TeddyMem {
  const uint64_t* bits;
  const uint8_t* g;
  int32_t lim;
  int32_t flag_at = 0x7FFFFFFF;   // serial-walk cut (scan_dfa.h walk_limit)
  mutable uint32_t over = 0;
  __device__ __forceinline__ uint32_t byte(int32_t r) const { over |= static_cast<uint32_t>(r >= flag_at); return g[r]; }
  __device__ __forceinline__ uint64_t cands(int32_t w) const { return bits[w]; }
  __device__ __forceinline__ int32_t bitmap_limit() const { return lim; }
};

struct RecSinkT {
  uint32_t* recs; uint32_t* rec_count; uint32_t lane; uint32_t n;
  __device__ __forceinline__ void emit(int32_t s, int32_t e) {
    const uint32_t j = n++;
    const uint32_t slot = atomicAdd(rec_count, 1u);
    if (slot < static_cast<uint32_t>(kRecCapT)) {
      recs[slot * 3 + 0] = static_cast<uint32_t>(s);
      recs[slot * 3 + 1] = static_cast<uint32_t>(e);
      recs[slot * 3 + 2] = (lane << 16) | (j & 0xFFFFu);
    }
  }
};
struct DirectSinkT {
  int64_t* out; uint64_t cap; uint64_t first; int64_t origin; uint32_t n;
  __device__ __forceinline__ void emit(int32_t s, int32_t e) {
    const uint64_t row = first + n++;
    if (row < cap) { longlong2 v; v.x = origin + s; v.y = origin + e; store_pair_nt(out + row * 2, v.x, v.y); }
  }
};
}  // namespace

__global__ __launch_bounds__(kThreads) void k_scan_teddy(ScanArgs a) {
  __shared__ __attribute__((aligned(16))) uint64_t s_bits[kWords];
  __shared__ __attribute__((aligned(16))) uint8_t s_info[256];
  __shared__ __attribute__((aligned(16))) uint8_t s_aux[kAuxMax];
  __shared__ uint32_t s_recs[kRecCapT * 3];
  __shared__ uint32_t s_cnt[kThreads];
  __shared__ uint32_t s_wsum[4];
  __shared__ uint32_t s_rec_count;
  __shared__ uint32_t s_tile_id;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x;
  if (tid == 0) { s_tile_id = static_cast<uint32_t>(claim_tile(a.ticket, a.ntiles)); s_rec_count = 0; }
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(a.blob);
  if (tid < 64) reinterpret_cast<uint32_t*>(s_info)[tid] = reinterpret_cast<const uint32_t*>(a.blob + h->info_off)[tid];
  for (uint32_t i = tid; i < h->aux_len / 4; i += kThreads)
    reinterpret_cast<uint32_t*>(s_aux)[i] = reinterpret_cast<const uint32_t*>(a.blob + h->aux_off)[i];
  __syncthreads();
  const TeddyAux* ax = reinterpret_cast<const TeddyAux*>(s_aux);
  TeddyView tv{reinterpret_cast<const uint16_t*>(s_aux + ax->ab_off), s_aux + ax->order_off, s_aux + ax->lens_off,
               s_aux + ax->bucket_off, reinterpret_cast<const uint16_t*>(s_aux + ax->off_off), s_aux + ax->bytes_off, ax->nlits};
  const uint64_t tile = s_tile_id;
  if (tile >= a.ntiles) return;
  const uint64_t tile_lo = tile * static_cast<uint64_t>(kTile);
  const uint64_t remaining = a.len - tile_lo;
  const WalkLimit wl = walk_limit(remaining, kTile + kHalo);   // serial-walk budget, scan_dfa.h
  const int32_t rend = wl.rend;
  const int32_t stage = rend < kTile + kHalo ? rend : kTile + kHalo;
  const uint8_t* g = a.hay + tile_lo;

  {
    uint16_t* pieces = reinterpret_cast<uint16_t*>(s_bits);
    const int nfull = stage >> 4;
    for (int v = tid; v < kWords * 4; v += kThreads) {
      uint32_t mask = 0;
      const int base = v << 4;
      if (v < nfull) {
        const uint4 x = *reinterpret_cast<const uint4*>(g + static_cast<size_t>(base));
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
        const uint32_t nb = (base + 16 < rend) ? g[base + 16] : 0u;   // first byte of the next vector (L1 hit)
        uint32_t prevA = tv.ab[w[0] & 0xFFu] & 0xFFu;
#pragma unroll
        for (int k = 1; k <= 16; k++) {
          const uint32_t b = (k < 16) ? ((w[k >> 2] >> ((k & 3) * 8)) & 0xFFu) : nb;
          const uint32_t e = tv.ab[b];
          const bool valid = (k < 16) || (base + 16 < rend);
          if (valid && (prevA & (e >> 8))) mask |= 1u << (k - 1);
          prevA = e & 0xFFu;
        }
      } else if (v == nfull) {
        for (int k = 0; base + k + 1 < stage || (base + k + 1 < rend && base + k < stage); k++) {
          if (base + k >= stage) break;
          const uint32_t e0 = tv.ab[g[base + k]], e1 = tv.ab[g[base + k + 1]];
          if ((e0 & 0xFFu) & (e1 >> 8)) mask |= 1u << k;
          if (k == 15) break;
        }
      }
      pieces[v] = static_cast<uint16_t>(mask);
    }
  }
  __syncthreads();

  TeddyMem m{s_bits, g, stage};
  m.flag_at = wl.flag_at;
  const int32_t c0 = tid * kChunk, c1 = c0 + kChunk;
  const bool at_origin = (tile_lo == 0 && tid == 0);
  RecSinkT sink{s_recs, &s_rec_count, static_cast<uint32_t>(tid), 0u};
  lane_teddy(m, tv, s_info, c0, c1, rend, at_origin, sink);
  if (m.over) raise_err(a.err, kErrSerialLimit);
  // a lane may emit more than 65 535 matches (no synchronising byte for a long stretch); the 16-bit rank in a
  // buffered record is only read when the whole tile emitted <= the record capacity, so that is not an error

  uint32_t total;
  const uint32_t excl = block_exclusive_scan(sink.n, s_wsum, total);
  s_cnt[tid] = excl;
  tile_lookback(a.status, a.total, a.err, tile, a.ntiles, total, &s_base);
  const uint64_t base = s_base;
  const int64_t origin = a.base + static_cast<int64_t>(tile_lo);
  if (a.out == nullptr) return;
  if (total <= static_cast<uint32_t>(kRecCapT)) {
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
    DirectSinkT ds{a.out, a.cap, base + excl, origin, 0u};
    lane_teddy(m, tv, s_info, c0, c1, rend, at_origin, ds);
  }
}

hipError_t launch_scan_teddy(const ScanArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(k_scan_teddy, dim3(static_cast<unsigned>(a.ntiles)), dim3(kThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace cxgdev
