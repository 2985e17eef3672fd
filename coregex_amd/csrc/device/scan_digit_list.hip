// scan_digit_list.hip — UseDigitPrefilter FindAll, third kernel generation ("candidates first").
//
// Precondition kFlagFastDigit (walk.hpp): the only positions the reference can ever try are digit-run
// starts, and whether one succeeds depends on the bytes alone.  So per 16 KiB tile:
//   A  stage tile + halo (coalesced 16-byte loads) and pack a digit bit per byte           [all lanes]
//   B  run starts = D & ~((D << 1) | carry): compact their positions into an LDS list      [block scan]
//   C  verify candidate k on lane k mod 256: anchored DFA walk that jumps over digit runs
//      in self-loop states via the bitmap; store the match length                           [balanced]
//   D  owners (lane = 64-byte chunk) pick, in order, the successful candidates of the segments they
//      own with start >= previous end — the reference's FindAll order — count, look-back, write.
// Compared with the flat walk no lane ever scans or idles: phase C is ~2 table steps per failing
// candidate, ~7 per IPv4 match, spread evenly over the wave.
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
constexpr int kCandCap = 4096;

__device__ __forceinline__ int lds_pad3(int r) { return r + (r >> 6) * 4; }

struct ListMem {
  const uint8_t* lds;
  const uint64_t* bits;
  const uint8_t* g;
  int32_t lim;
  int32_t flag_at = 0x7FFFFFFF;   // serial-walk cut (scan_dfa.h walk_limit)
  mutable uint32_t over = 0;
  __device__ __forceinline__ uint32_t byte(int32_t r) const {
    if (static_cast<uint32_t>(r) < static_cast<uint32_t>(lim)) return lds[lds_pad3(r)];
    over |= static_cast<uint32_t>(r >= flag_at);
    return g[r];
  }
  __device__ __forceinline__ uint64_t digits(int32_t w) const { return bits[w]; }
  __device__ __forceinline__ int32_t bitmap_limit() const { return lim; }
};

struct CountSink { uint32_t n; __device__ __forceinline__ void emit(int32_t, int32_t) { n++; } };
struct WriteSink {
  int64_t* out; uint64_t cap; uint64_t first; int64_t origin; uint32_t n;
  __device__ __forceinline__ void emit(int32_t s, int32_t e) {
    const uint64_t row = first + n++;
    if (row < cap) { longlong2 v; v.x = origin + s; v.y = origin + e; *reinterpret_cast<longlong2*>(out + row * 2) = v; }
  }
};

__device__ __forceinline__ uint32_t digit_bits4l(uint32_t x) {
  const uint32_t t = digit_mask4(x) >> 7;
  return ((t * 0x00204081u) >> 21) & 0xFu;
}

}  // namespace

__global__ __launch_bounds__(kThreads) void k_scan_digit_list(ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];   // DFA table, info, sflags
  __shared__ __attribute__((aligned(16))) uint8_t s_tile[kLdsTileBytes];
  __shared__ __attribute__((aligned(16))) uint64_t s_bits[kWords];
  __shared__ uint16_t s_cpos[kCandCap];
  __shared__ uint8_t s_clen[kCandCap];
  __shared__ uint32_t s_cexcl[kThreads];
  __shared__ uint32_t s_wsum[4];
  __shared__ uint32_t s_halo[kHaloChunks];
  __shared__ uint64_t s_tile_id;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x;
  if (tid == 0) s_tile_id = claim_tile(a.ticket, a.ntiles);
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(a.blob);
  const uint32_t fwd_states = h->fwd_states;
  uint8_t* s_fwd = s_dyn;
  uint8_t* s_info = s_fwd + fwd_states * kRowStride;
  uint8_t* s_sfl = s_info + 256;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.blob + h->fwd_off);
    for (uint32_t i = tid; i < fwd_states * 64u; i += kThreads)
      *reinterpret_cast<uint32_t*>(s_fwd + (i >> 6) * kRowStride + (i & 63u) * 4u) = src[i];
    if (tid < 64) reinterpret_cast<uint32_t*>(s_info)[tid] = reinterpret_cast<const uint32_t*>(a.blob + h->info_off)[tid];
    else if (tid < 128) reinterpret_cast<uint32_t*>(s_sfl)[tid - 64] = reinterpret_cast<const uint32_t*>(a.blob + h->aux_off)[tid - 64];
  }
  __syncthreads();
  const uint64_t tile = s_tile_id;
  if (tile >= a.ntiles) return;
  const uint64_t tile_lo = tile * static_cast<uint64_t>(kTile);
  const uint64_t remaining = a.len - tile_lo;
  const WalkLimit wl = walk_limit(remaining, kTile + kHalo);   // serial-walk budget, scan_dfa.h
  const int32_t rend = wl.rend;
  const int32_t stage = rend < kTile + kHalo ? rend : kTile + kHalo;
  const uint8_t* g = a.hay + tile_lo;

  // ---- A: stage bytes + digit bitmap (all loads issued before use)
  {
    uint16_t* pieces = reinterpret_cast<uint16_t*>(s_bits);
    const int nfull = stage >> 4;
    constexpr int kIter = (kWords * 4 + kThreads - 1) / kThreads;
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
        uint32_t* d = reinterpret_cast<uint32_t*>(s_tile + lds_pad3(v << 4));
        d[0] = x[k].x; d[1] = x[k].y; d[2] = x[k].z; d[3] = x[k].w;
        mask = digit_bits4l(x[k].x) | (digit_bits4l(x[k].y) << 4) | (digit_bits4l(x[k].z) << 8) | (digit_bits4l(x[k].w) << 12);
      } else if (v == nfull) {
        const int base = v << 4;
        for (int j = 0; base + j < stage; j++) {
          const uint32_t b = g[base + j];
          s_tile[lds_pad3(base + j)] = static_cast<uint8_t>(b);
          mask |= (is_digit(b) ? 1u : 0u) << j;
        }
      }
      pieces[v] = static_cast<uint16_t>(mask);
    }
  }
  __syncthreads();

  if (a.dbg & 1u) {   // timing experiment: staging only
    if (tid == 0 && tile == a.ntiles - 1) *a.total = 0;
    return;
  }
  // ---- B: run starts -> candidate list (positions ascending)
  const uint64_t M = s_bits[tid];
  uint64_t carry;
  if (tid > 0) carry = s_bits[tid - 1] >> 63;
  else carry = (tile_lo > 0) ? (is_digit(g[-1]) ? 1u : 0u) : 0u;
  uint64_t rs = M & ~((M << 1) | carry);
  uint64_t hrs = 0;                                    // halo words: lanes 0..3 also own word 256+tid
  if (tid < kHaloChunks) {
    const uint64_t HM = s_bits[kThreads + tid];
    hrs = HM & ~((HM << 1) | (s_bits[kThreads + tid - 1] >> 63));
    s_halo[tid] = static_cast<uint32_t>(__popcll(hrs));
  }
  uint32_t main_total;
  const uint32_t cexcl = block_exclusive_scan(static_cast<uint32_t>(__popcll(rs)), s_wsum, main_total);
  s_cexcl[tid] = cexcl;
  uint32_t ncand = main_total;
  uint32_t hexcl = main_total;
#pragma unroll
  for (int k = 0; k < kHaloChunks; k++) { if (k < tid) hexcl += s_halo[k]; ncand += s_halo[k]; }
  if (ncand > static_cast<uint32_t>(kCandCap)) {       // digit-dense tile: the host reruns with the flat kernel
    if (tid == 0) atomicOr(a.err, 8u);
    ncand = kCandCap;
  }
  {
    uint32_t k = cexcl;
    while (rs) { const int bit = __builtin_ctzll(rs); rs &= rs - 1; if (k < kCandCap) s_cpos[k] = static_cast<uint16_t>(tid * 64 + bit); k++; }
    if (tid < kHaloChunks) {
      k = hexcl;
      while (hrs) { const int bit = __builtin_ctzll(hrs); hrs &= hrs - 1; if (k < kCandCap) s_cpos[k] = static_cast<uint16_t>((kThreads + tid) * 64 + bit); k++; }
    }
  }
  __syncthreads();

  // ---- C: verify, one candidate per lane per round
  ListMem m{s_tile, s_bits, g, stage};
  m.flag_at = wl.flag_at;
  DfaView fv{s_fwd, kRowStride, h->fwd_start, h->fwd_first_accept};
  for (uint32_t k = tid; k < ncand; k += kThreads) {
    const int32_t c = s_cpos[k];
    const int32_t e = (a.dbg & 4u) ? -1 : verify_jump(m, fv, s_sfl, c, rend);
    const int32_t len = e < 0 ? 0 : (e - c > 255 ? 255 : e - c);
    s_clen[k] = static_cast<uint8_t>(len);
  }
  __syncthreads();

  // ---- D: owners select, count, look back, write
  const int32_t c0 = tid * kChunk, c1 = c0 + kChunk;
  const bool at_origin = (tile_lo == 0 && tid == 0);
  CountSink cs{0u};
  if (!(a.dbg & 8u)) lane_select(m, fv, s_info, s_sfl, s_cpos, s_clen, ncand, cexcl, stage, c0, c1, rend, at_origin, cs);
  if (m.over) raise_err(a.err, kErrSerialLimit);
  uint32_t total;
  const uint32_t excl = block_exclusive_scan(cs.n, s_wsum, total);
  tile_lookback(a.status, a.total, a.err, tile, a.ntiles, total, &s_base);
  if (a.out == nullptr) return;
  if (cs.n) {
    WriteSink ws{a.out, a.cap, s_base + excl, a.base + static_cast<int64_t>(tile_lo), 0u};
    lane_select(m, fv, s_info, s_sfl, s_cpos, s_clen, ncand, cexcl, stage, c0, c1, rend, at_origin, ws);
  }
}

hipError_t launch_scan_digit_list(const ScanArgs& a, uint32_t fwd_states, hipStream_t stream) {
  const size_t dyn = static_cast<size_t>(fwd_states) * kRowStride + 512;
  hipLaunchKernelGGL(k_scan_digit_list, dim3(static_cast<unsigned>(a.ntiles)), dim3(kThreads), dyn, stream, a);
  return hipGetLastError();
}

}  // namespace cxgdev
