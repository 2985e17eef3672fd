// scan_charclass.hip — FindAll for UseCharClassSearcher (`[class]+`), the GPU form of
// nfa.CharClassSearcher.FindAllIndices (nfa/charclass_searcher.go:158-211): maximal runs of member
// bytes, emitted in order, a trailing run closed at end of input.
//
// The reference is a 2-state scalar machine, one byte per iteration.  Runs are a pure bit problem, so
// here the work is data-parallel: each thread classifies the 16-byte vectors it loaded (coalesced,
// 1 KiB per wave instruction) through the 256-entry table in LDS and deposits a 16-bit membership
// mask; after a barrier lane t owns the 64-bit word for bytes [64t, 64t+64) and
//     starts = M & ~((M << 1) | carry_in)
// gives its run starts.  A run belongs to the lane where it starts; its end is the next zero bit,
// found in the lane's own word, the following words, or (past the staged halo) in HBM.  Counts go
// through the block scan + look-back of block_common.hpp; rows are assembled in rank order in LDS and
// leave as fully coalesced 16-byte stores.
//
// Roofline: HBM-bound and write-heavy: 16 KiB read + 16 B x (one run per ~5.5 B of log text) written.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_common.hpp"
#include "scan_dfa.h"
#include "walk.hpp"

namespace cxgdev {

namespace {
constexpr int kWords = kThreads + kHaloChunks;   // 64-bit membership words staged per tile
constexpr int kCcRows = 3072;                     // LDS row buffer (u32 start, u32 end)
}  // namespace

__global__ __launch_bounds__(kThreads) void k_scan_charclass(ScanArgs a) {
  __shared__ __attribute__((aligned(16))) uint64_t s_bits[kWords];
  __shared__ __attribute__((aligned(16))) uint8_t s_info[256];
  __shared__ uint2 s_rows[kCcRows];
  __shared__ uint32_t s_wsum[4];
  __shared__ uint32_t s_tile_id;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x;
  if (tid == 0) s_tile_id = static_cast<uint32_t>(claim_tile(a.ticket, a.ntiles));
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(a.blob);
  if (tid < 64) reinterpret_cast<uint32_t*>(s_info)[tid] = reinterpret_cast<const uint32_t*>(a.blob + h->info_off)[tid];
  __syncthreads();
  const uint64_t tile = s_tile_id;
  if (tile >= a.ntiles) return;
  const uint64_t tile_lo = tile * static_cast<uint64_t>(kTile);
  const uint64_t remaining = a.len - tile_lo;
  const WalkLimit wl = walk_limit(remaining, kTile + kHalo);   // serial-walk budget, scan_dfa.h
  const int32_t rend = wl.rend;
  const int32_t stage = rend < kTile + kHalo ? rend : kTile + kHalo;
  const uint8_t* g = a.hay + tile_lo;

  // ---- phase 1: classify, 16 bytes per thread per step, one 16-bit mask each
  {
    uint16_t* pieces = reinterpret_cast<uint16_t*>(s_bits);
    const int nfull = stage >> 4;
    for (int v = tid; v < kWords * 4; v += kThreads) {
      uint32_t mask = 0;
      if (v < nfull) {
        const uint4 x = *reinterpret_cast<const uint4*>(g + (static_cast<size_t>(v) << 4));
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const uint32_t b = (w[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
          mask |= ((s_info[b] >> 1) & 1u) << k;
        }
      } else if (v == nfull) {
        const int base = v << 4;
        for (int k = 0; base + k < stage; k++) mask |= ((s_info[g[base + k]] >> 1) & 1u) << k;
      }
      pieces[v] = static_cast<uint16_t>(mask);
    }
  }
  __syncthreads();

  // ---- phase 2: run starts of this lane's 64 bytes
  const uint64_t M = s_bits[tid];
  uint64_t carry;
  if (tid > 0) carry = s_bits[tid - 1] >> 63;
  else carry = (tile_lo > 0) ? ((s_info[g[-1]] >> 1) & 1u) : 0u;
  uint64_t starts = M & ~((M << 1) | carry);
  const uint32_t mine = static_cast<uint32_t>(__popcll(starts));
  uint32_t total;
  const uint32_t excl = block_exclusive_scan(mine, s_wsum, total);
  tile_lookback(a.status, a.total, a.err, tile, a.ntiles, total, &s_base);
  if (a.out == nullptr) return;
  const uint64_t base = s_base;
  const int64_t origin = a.base + static_cast<int64_t>(tile_lo);
  const bool buffered = total <= static_cast<uint32_t>(kCcRows);

  // ---- phase 3: pair every start with its end
  uint32_t j = 0;
  while (starts) {
    const int bit = __builtin_ctzll(starts);
    starts &= starts - 1;
    const int32_t s = tid * 64 + bit;
    int32_t e;
    const uint64_t z = ~M & (~0ull << bit);
    if (z) {
      e = tid * 64 + __builtin_ctzll(z);
    } else {
      e = -1;
      for (int w = tid + 1; w < kWords; w++) {
        const uint64_t nz = ~s_bits[w];
        if (nz) { e = w * 64 + __builtin_ctzll(nz); break; }
      }
      if (e < 0) e = kWords * 64;
    }
    if (e >= stage) {                       // ran off the staged bitmap: finish in HBM
      e = stage;
      while (e < rend && ((s_info[g[e]] >> 1) & 1u)) e++;
      if (e >= wl.flag_at) raise_err(a.err, kErrSerialLimit);   // the run outlasts the serial-walk budget
    }
    const uint32_t row = excl + j++;
    if (buffered) {
      s_rows[row] = make_uint2(static_cast<uint32_t>(s), static_cast<uint32_t>(e));
    } else if (base + row < a.cap) {
      longlong2 v; v.x = origin + s; v.y = origin + e;
      store_pair_nt(a.out + (base + row) * 2, v.x, v.y);
    }
  }
  if (!buffered) return;
  __syncthreads();
  for (uint32_t i = tid; i < total; i += kThreads) {
    if (base + i < a.cap) {
      const uint2 r = s_rows[i];
      longlong2 v; v.x = origin + static_cast<int32_t>(r.x); v.y = origin + static_cast<int32_t>(r.y);
      store_pair_nt(a.out + (base + i) * 2, v.x, v.y);
    }
  }
}

hipError_t launch_scan_charclass(const ScanArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(k_scan_charclass, dim3(static_cast<unsigned>(a.ntiles)), dim3(kThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace cxgdev
