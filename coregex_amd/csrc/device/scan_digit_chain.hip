// scan_digit_chain.hip — UseDigitPrefilter FindAll, fourth kernel generation: bit-parallel chain
// prefilter + DFA verification of the survivors.
//
// Preconditions kFlagFastDigit | kFlagChain (walk.hpp): candidates are digit-run starts, and the
// anchored DFA begins with a chain of run(class+) / byte(class) steps (for `\d+\.\d+\.\d+\.\d+` the
// whole pattern).  Per 16 KiB tile (+256 B halo):
//   A  coalesced 16-byte loads; each thread packs, per chain class, a 16-bit membership mask of its 16
//      bytes (SWAR range / equality tests, no table lookups) and stores it BIT-REVERSED, so that LDS holds
//      one reversed bitmap per class.  The haystack bytes themselves are not staged at all.
//   B  the chain is evaluated right to left on those bitmaps, one 64-bit word per lane: byte steps are a
//      shift+AND with the neighbour word, run steps one multi-word addition whose carries are resolved
//      through LDS (usually one extra round).  AND with the digit-run starts = surviving candidates
//      (for the IPv4 pattern: exactly the matching positions, ~1 per 100 B instead of ~13).
//   C  survivors are compacted in position order; lane k verifies survivor k with the anchored DFA walk
//      (bytes from L2) — this is what makes the result exact — and finds its segment start for ownership.
//   D  one lane applies the reference's FindAll order (start >= previous end) to the owned successes;
//      block scan, look-back, 16-byte row stores.
// A workgroup processes a GROUP of 8 consecutive tiles: one ticket atomic, one table staging, one
// look-back and one row write-out per 128 KiB.  (A ticket per 16 KiB tile costs ~0.6 ms per GiB on its
// own — scripts/microbench/stream.hip: 6.6 TB/s plain tile reads vs 1.4 TB/s with a per-tile ticket.)
// Rows of the group are collected in LDS in FindAll order and written with coalesced 16-byte stores.
// If the halo holds no synchronising byte, or a tile has more survivors than the LDS list, the tile
// raises a flag and the host reruns the scan with the flat kernel (exact, slower).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_common.hpp"
#include "scan_dfa.h"
#include "walk.hpp"

namespace cxgdev {

namespace {

constexpr int kRowStride = 260;
constexpr int kWords = kThreads + kHaloChunks;   // 260 data words
constexpr int kNW = kWords + 1;                  // + one all-zero word beyond the staged window (reversed word 0)
constexpr int kNP = kNW * 4;                     // 16-bit pieces per bitmap
constexpr int kSurvCap = 512;
constexpr int kRowCap = 2048;                    // rows buffered per group

struct GlobalMem {       // verification reads the haystack straight from L2/HBM (rare)
  const uint8_t* g;
  int32_t flag_at = 0x7FFFFFFF;   // serial-walk cut (scan_dfa.h walk_limit)
  mutable uint32_t over = 0;
  __device__ __forceinline__ uint32_t byte(int32_t r) const { over |= static_cast<uint32_t>(r >= flag_at); return g[r]; }
  __device__ __forceinline__ uint64_t digits(int32_t) const { return 0; }
  __device__ __forceinline__ int32_t bitmap_limit() const { return 0; }
};

__device__ __forceinline__ uint32_t gather4(uint32_t m80) {           // 0x80 flags of 4 bytes -> 4 bits
  return (((m80 >> 7) * 0x00204081u) >> 21) & 0xFu;
}
__device__ __forceinline__ uint32_t cls_bits4(uint32_t x, uint32_t kind, uint32_t lo, uint32_t hi) {
  if (kind == kClsDigit) return gather4(digit_mask4(x));
  if (kind == kClsByte) {
    const uint32_t v = x ^ (lo * 0x01010101u);
    const uint32_t nz = (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;   // 0x80 where the byte is non-zero
    return gather4(nz ^ 0x80808080u);
  }
  const uint32_t ge = ((x | 0x80808080u) - lo * 0x01010101u) & 0x80808080u;       // low7 >= lo
  const uint32_t gt = ((x & 0x7F7F7F7Fu) + (0x7Fu - hi) * 0x01010101u) & 0x80808080u;  // low7 > hi
  return gather4(ge & ~gt & ~x & 0x80808080u);
}
__device__ __forceinline__ uint32_t brev16(uint32_t m) { return __brev(m) >> 16; }

}  // namespace

__global__ __launch_bounds__(kThreads) void k_scan_digit_chain(ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];   // DFA table, info, sflags, ChainAux
  __shared__ __attribute__((aligned(16))) uint64_t s_cls[kChainMaxCls][kNW];
  __shared__ __attribute__((aligned(16))) uint64_t s_g[2][kNW];
  __shared__ uint32_t s_carry[kNW];
  __shared__ uint16_t s_spos[kSurvCap];
  __shared__ uint16_t s_slen[kSurvCap];
  __shared__ uint8_t s_sown[kSurvCap];
  __shared__ uint32_t s_rowpos[kRowCap];
  __shared__ uint16_t s_rowlen[kRowCap];
  __shared__ uint32_t s_nrows;
  __shared__ uint32_t s_wsum[4];
  __shared__ uint32_t s_halo[8];
  __shared__ uint64_t s_tile_id;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x;
  if (tid == 0) { s_tile_id = claim_tile(a.ticket, a.ngroups); s_nrows = 0; }
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(a.blob);
  const uint32_t fwd_states = h->fwd_states;
  uint8_t* s_fwd = s_dyn;
  uint8_t* s_info = s_fwd + fwd_states * kRowStride;
  uint8_t* s_sfl = s_info + 256;
  ChainAux* s_chain = reinterpret_cast<ChainAux*>(s_sfl + 256);
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.blob + h->fwd_off);
    for (uint32_t i = tid; i < fwd_states * 64u; i += kThreads)
      *reinterpret_cast<uint32_t*>(s_fwd + (i >> 6) * kRowStride + (i & 63u) * 4u) = src[i];
    if (tid < 64) reinterpret_cast<uint32_t*>(s_info)[tid] = reinterpret_cast<const uint32_t*>(a.blob + h->info_off)[tid];
    else if (tid < 128) reinterpret_cast<uint32_t*>(s_sfl)[tid - 64] = reinterpret_cast<const uint32_t*>(a.blob + h->aux_off)[tid - 64];
    else if (tid < 128 + sizeof(ChainAux) / 4)
      reinterpret_cast<uint32_t*>(s_chain)[tid - 128] = reinterpret_cast<const uint32_t*>(a.blob + h->aux_off + 256)[tid - 128];
  }
  __syncthreads();
  const uint64_t group = s_tile_id;
  if (group >= a.ngroups) return;
  const uint32_t ncls = s_chain->ncls, nops = s_chain->nops;
  DfaView fv{s_fwd, kRowStride, h->fwd_start, h->fwd_first_accept};
  const int wa = kNW - 1 - tid;
  const bool has_b = tid < 5;
  const int wb = tid;

  for (int gj = 0; gj < kGroupTiles; gj++) {
  const uint64_t tile = group * kGroupTiles + gj;
  if (tile >= a.ntiles) break;
  const uint64_t tile_lo = tile * static_cast<uint64_t>(kTile);
  const uint64_t remaining = a.len - tile_lo;
  const WalkLimit wl = walk_limit(remaining, kTile + kHalo);   // serial-walk budget, scan_dfa.h
  const int32_t rend = wl.rend;
  const int32_t stage = rend < kTile + kHalo ? rend : kTile + kHalo;
  const uint8_t* g = a.hay + tile_lo;

  // ---- A: class bitmaps, reversed
  {
    const int nfull = stage >> 4;
    constexpr int kIter = (kWords * 4 + kThreads - 1) / kThreads;   // 5
    uint4 x[kIter];
#pragma unroll
    for (int k = 0; k < kIter; k++) {
      const int v = tid + k * kThreads;
      x[k] = (v < nfull) ? *reinterpret_cast<const uint4*>(g + (static_cast<size_t>(v) << 4)) : make_uint4(0, 0, 0, 0);
    }
    for (uint32_t c = 0; c < ncls; c++) {
      const uint32_t kind = s_chain->cls_kind[c], lo = s_chain->cls_lo[c], hi = s_chain->cls_hi[c];
      uint16_t* pieces = reinterpret_cast<uint16_t*>(s_cls[c]);
#pragma unroll
      for (int k = 0; k < kIter; k++) {
        const int v = tid + k * kThreads;
        if (v >= kWords * 4) break;
        uint32_t mask = 0;
        if (v < nfull) {
          mask = cls_bits4(x[k].x, kind, lo, hi) | (cls_bits4(x[k].y, kind, lo, hi) << 4) |
                 (cls_bits4(x[k].z, kind, lo, hi) << 8) | (cls_bits4(x[k].w, kind, lo, hi) << 12);
        } else if (v == nfull) {
          const int base = v << 4;
          for (int j = 0; base + j < stage; j++) mask |= (chain_class_has(*s_chain, static_cast<int>(c), g[base + j]) ? 1u : 0u) << j;
        }
        pieces[kNP - 1 - v] = static_cast<uint16_t>(brev16(mask));
      }
      if (tid < 4) pieces[tid] = 0;                                 // the word beyond the staged window
    }
  }
  // lanes and words: lane t owns reversed word wa = kNW-1-t (its own 64-byte chunk); lanes 0..4 also own word t
  s_g[0][wa] = ~0ull;
  if (has_b) s_g[0][wb] = ~0ull;
  __syncthreads();

  // ---- B: chain, right to left
  int cur = 0;
  for (int k = static_cast<int>(nops) - 1; k >= 0; k--) {
    const uint64_t* C = s_cls[s_chain->op_cls[k]];
    const uint64_t* src = s_g[cur];
    uint64_t* dst = s_g[cur ^ 1];
    if (s_chain->op_kind[k] == kChainByte) {
      {
        const uint64_t lowin = src[wa - 1] >> 63;                     // wa >= 5
        dst[wa] = C[wa] & ((src[wa] << 1) | lowin);
      }
      if (has_b) {
        const uint64_t lowin = wb > 0 ? (src[wb - 1] >> 63) : 1ull;
        dst[wb] = C[wb] & ((src[wb] << 1) | lowin);
      }
      __syncthreads();
    } else {
      auto kword = [&](int w) -> uint64_t {                          // markers just below a reversed run, in G_{k+1}
        const uint64_t cup = (C[w] >> 1) | ((w + 1 < kNW ? C[w + 1] : 0ull) << 63);
        return src[w] & ~C[w] & cup;
      };
      uint64_t s1a, s1b = 0; uint32_t ga, gb = 0, pa, pb = 0;
      {
        const uint64_t M = (kword(wa) << 1) | (kword(wa - 1) >> 63);
        s1a = C[wa] + M; ga = s1a < M ? 1u : 0u; pa = (s1a == ~0ull) ? 1u : 0u;
      }
      if (has_b) {
        const uint64_t M = (kword(wb) << 1) | (wb > 0 ? (kword(wb - 1) >> 63) : 0ull);
        s1b = C[wb] + M; gb = s1b < M ? 1u : 0u; pb = (s1b == ~0ull) ? 1u : 0u;
      }
      uint32_t cina = 0, cinb = 0;
      for (int round = 0; round < kNW + 1; round++) {
        s_carry[wa] = ga | (pa & cina);
        if (has_b) s_carry[wb] = gb | (pb & cinb);
        __syncthreads();
        const uint32_t na = s_carry[wa - 1];
        const uint32_t nb = (has_b && wb > 0) ? s_carry[wb - 1] : 0u;
        const int changed = (na != cina) | (nb != cinb);
        cina = na; cinb = nb;
        if (!__syncthreads_or(changed)) break;
      }
      dst[wa] = C[wa] & ~(s1a + cina);
      if (has_b) dst[wb] = C[wb] & ~(s1b + cinb);
      __syncthreads();
    }
    cur ^= 1;
  }

  // ---- survivors = digit-run starts that pass the chain
  const uint64_t* G1 = s_g[cur];
  const uint64_t* D = s_cls[0];
  auto survivors = [&](int w) -> uint64_t {
    uint64_t up;                                                      // digit flag of the byte before, per position
    if (w + 1 < kNW) up = (D[w] >> 1) | (D[w + 1] << 63);
    else up = (D[w] >> 1) | ((tile_lo > 0 && is_digit(g[-1])) ? (1ull << 63) : 0ull);
    return D[w] & ~up & G1[w];
  };
  uint64_t sva = survivors(wa);
  uint64_t svb = (has_b && wb > 0) ? survivors(wb) : 0ull;            // wb: 4,3,2,1 <-> chunks 256..259
  if (has_b) s_halo[tid] = static_cast<uint32_t>(__popcll(svb));
  uint32_t main_total;
  const uint32_t sexcl = block_exclusive_scan(static_cast<uint32_t>(__popcll(sva)), s_wsum, main_total);
  uint32_t nsurv = main_total + s_halo[4] + s_halo[3] + s_halo[2] + s_halo[1];
  if (nsurv > static_cast<uint32_t>(kSurvCap)) {
    if (tid == 0) atomicOr(a.err, 8u);
    nsurv = kSurvCap;
  }
  {
    uint32_t k = sexcl;                                               // ascending position = descending reversed bit
    while (sva) { const int bit = 63 - __builtin_clzll(sva); sva &= ~(1ull << bit); if (k < kSurvCap) s_spos[k] = static_cast<uint16_t>(tid * 64 + (63 - bit)); k++; }
    if (has_b && wb > 0) {
      k = main_total;
      for (int j = 4; j > wb; j--) k += s_halo[j];
      const int chunk = kNW - 1 - wb;
      while (svb) { const int bit = 63 - __builtin_clzll(svb); svb &= ~(1ull << bit); if (k < kSurvCap) s_spos[k] = static_cast<uint16_t>(chunk * 64 + (63 - bit)); k++; }
    }
  }
  // the halo must contain a synchronising byte, otherwise an owned segment may extend past what was seen
  int halo_sync = (stage == rend) ? 1 : 0;
  for (int p = kTile - 1 + tid; p < stage; p += kThreads) halo_sync |= (s_info[g[p]] & kInfoSync) ? 1 : 0;
  if (!__syncthreads_or(halo_sync)) { if (tid == 0) atomicOr(a.err, 8u); }

  // ---- C: verify survivors with the DFA; ownership = where their segment starts
  GlobalMem m{g};
  m.flag_at = wl.flag_at;
  for (uint32_t k = tid; k < nsurv; k += kThreads) {
    const int32_t c = s_spos[k];
    const int32_t e = verify_jump(m, fv, s_sfl, c, rend);
    int32_t len = e < 0 ? 0 : e - c;
    if (len > 0xFFFF) { atomicOr(a.err, 8u); len = 0; }
    if (m.over) { raise_err(a.err, kErrSerialLimit); m.over = 0; }
    s_slen[k] = static_cast<uint16_t>(len);
    uint8_t owned = 0;
    if (len) {
      int32_t p = c - 1;
      while (p >= 0 && !(s_info[g[p]] & kInfoSync)) p--;
      int32_t seg;                                                    // first position of the segment holding c
      if (p >= 0) seg = p + 1;
      else seg = (tile_lo == 0 || (s_info[g[-1]] & kInfoSync)) ? 0 : -1;
      owned = (seg >= 0 && seg < kTile) ? 1 : 0;
    }
    s_sown[k] = owned;
  }
  __syncthreads();

  // ---- D: FindAll order among the owned successes (sequential by nature, a few dozen items): the lane
  // appends the emitted rows to the group's row list, which therefore is in FindAll order.
  if (tid == 0) {
    int32_t cur_end = -1;
    uint32_t n = s_nrows;
    for (uint32_t k = 0; k < nsurv; k++) {
      if (!s_sown[k]) continue;
      const int32_t c = s_spos[k];
      if (c < cur_end) continue;
      cur_end = c + s_slen[k];
      if (n < static_cast<uint32_t>(kRowCap)) { s_rowpos[n] = static_cast<uint32_t>(gj) * kTile + static_cast<uint32_t>(c); s_rowlen[n] = s_slen[k]; }
      n++;
    }
    s_nrows = n;
  }
  __syncthreads();                                                    // LDS is reused by the next tile
  }  // tiles of the group

  uint32_t total = s_nrows;
  if (total > static_cast<uint32_t>(kRowCap)) { if (tid == 0) atomicOr(a.err, 8u); total = kRowCap; }
  tile_lookback(a.status, a.total, a.err, group, a.ngroups, total, &s_base);
  if (a.out == nullptr) return;
  const uint64_t base = s_base;
  const int64_t origin = a.base + static_cast<int64_t>(group * kGroupTiles * static_cast<uint64_t>(kTile));
  for (uint32_t i = tid; i < total; i += kThreads) {
    const uint64_t row = base + i;
    if (row < a.cap) {
      longlong2 v; v.x = origin + s_rowpos[i]; v.y = v.x + s_rowlen[i];
      *reinterpret_cast<longlong2*>(a.out + row * 2) = v;
    }
  }
}

hipError_t launch_scan_digit_chain(const ScanArgs& a, uint32_t fwd_states, hipStream_t stream) {
  const size_t dyn = static_cast<size_t>(fwd_states) * kRowStride + 512 + sizeof(ChainAux) + 16;
  hipLaunchKernelGGL(k_scan_digit_chain, dim3(static_cast<unsigned>(a.ngroups)), dim3(kThreads), dyn, stream, a);
  return hipGetLastError();
}

}  // namespace cxgdev
