// scan_charclass_wave.hip — FindAll for UseCharClassSearcher (`[class]+`) with the wave as the unit of work.
// GPU form of nfa.CharClassSearcher.FindAllIndices (nfa/charclass_searcher.go:158-211): maximal runs of member
// bytes, in order, a trailing run closed at the end of input.
//
// Output dominates this path (one 16-byte span per ~5.5 bytes of log text: 3 bytes written per byte read), so a
// group's rows are never buffered as a whole: bitmaps wait in LDS for the group's base, rows are staged one
// wave-tile at a time.
//   pass 1  per wave-tile (60 x 64 B = 3840 B + 256 B halo = 64 bitmap words, one per lane): window by four
//           buffer_load_dwordx4 per lane, membership by SWAR range tests (the class is a union of <= 4 ASCII
//           ranges) + v_dot4 gather, 16-bit pieces through LDS -> word M per lane;
//             starts S = M & ~(M << 1 | carry),  ends E = ~M & (M << 1 | carry)   (exclusive ends)
//           starts and ends are owned SEPARATELY: the tile owns the starts at its bytes [0, 3840) and the exclusive
//           ends at (0, 3840] — a run may be as long as the haystack, nobody has to see both of its ends.  Runs
//           alternate start, end, start, end, so the k-th start and the k-th end of the haystack are row k: with
//           B = number of starts in front of the tile (look-back), the tile's i-th start is row B + i and its j-th
//           end is row B - open + j, open = 1 when a run crosses the tile's first byte.  S and E words stay in LDS,
//           the start counts are summed per tile;
//   group   (4 waves x 4 wave-tiles = 60 KiB) one barrier, one look-back -> global base of every tile;
//   pass 2  per wave-tile every lane drops the starts and ends it holds into the wave's LDS staging at their
//           ranks, then the wave writes the rows that have both halves here as fully coalesced 16-byte stores
//           (1 KiB per instruction); the end of a run begun in an earlier tile and the start of a run that ends in a
//           later one go out as lone 8-byte stores.
// Fallback flag (err bit 8: the host reruns the scan with scan_charclass.hip): more than 1024 starts or ends in one
// wave-tile.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_common.hpp"
#include "scan_dfa.h"
#include "walk.hpp"
#include "wave_common.hpp"

#ifndef CXG_CC_PLANS
#define CXG_CC_PLANS 1
#endif
// Staging index of rank r (pass 2).  In one ds_write_b16 lane l writes rank r_l + i with r_l ~ 11 l on log text: indices r / 2 mod 32
// put ~11 lanes on every LDS bank.  r * 13 mod 1024 (a bijection) spreads them — the coalesced read-out of consecutive ranks stays
// two lanes per bank.
#ifndef CXG_CC_SWIZZLE
#define CXG_CC_SWIZZLE 1
#endif
#ifndef CXG_CC_MERGED
#define CXG_CC_MERGED 0                                      // one extraction loop over starts and ends (0: two loops, round 4)
#endif
#ifndef CXG_CC_LOAD_AUX
#define CXG_CC_LOAD_AUX 0                                    // cache policy of the haystack loads (2 = nt; A/B)
#endif
// Windows in flight per wave in pass 1 (1, 2 or 4 = all of the wave's tiles issued before the first is classified).  Pass 1 alone
// (a count-only launch) ran at 2.5 TB/s with one window of prefetch: 16 waves per CU x 4 KiB are too few bytes in flight.
#ifndef CXG_CC_DEPTH
#define CXG_CC_DEPTH 1
#endif
#ifndef CXG_CC_HALVES
#define CXG_CC_HALVES 0                                      // extraction: both 32-bit halves of a word per loop iteration
#endif

namespace cxgdev {

namespace {
constexpr int kWin = kWaveTile + kWaveHalo;       // 4096
constexpr int kCcStage = 1024;                    // rows staged per wave-tile (typical log text: ~700)
__device__ __forceinline__ uint32_t cc_slot(uint32_t r) { return CXG_CC_SWIZZLE ? ((r * 13u) & static_cast<uint32_t>(kCcStage - 1)) : r; }
}

__global__ __launch_bounds__(kThreads, 4) void k_scan_charclass_wave(ScanArgs a) {
  __shared__ __attribute__((aligned(16))) uint64_t s_m[kWavesPerBlock][64];                    // membership pieces -> words
  __shared__ __attribute__((aligned(16))) uint64_t s_S[kWavesPerBlock][kCcTilesPerWave][64];   // owned starts
  __shared__ __attribute__((aligned(16))) uint64_t s_E[kWavesPerBlock][kCcTilesPerWave][64];   // their ends
  __shared__ uint16_t s_rs[kWavesPerBlock][kCcStage];              // pass 2 staging: start / end inside the window
  __shared__ uint16_t s_re[kWavesPerBlock][kCcStage];
  __shared__ uint32_t s_cnt[kWavesPerBlock][kCcTilesPerWave];
  __shared__ uint32_t s_qbase[kWavesPerBlock * kCcTilesPerWave + 1];
  __shared__ uint64_t s_group;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = lane0;
  if (tid == 0) s_group = claim_group(a.static_groups != 0, a.ticket, a.ngroups);
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(a.blob);
  const CharClassAux* ax = reinterpret_cast<const CharClassAux*>(a.blob + h->aux_off);   // uniform address: scalar loads
  SetRanges rg;
  rg.n = ax->nr;
  const uint32_t flip = ax->neg ? 0u : 0xFFFFu;                     // notset4 flags the bytes OUTSIDE the ranges: the members of a complemented class
  // `Q[^Q]*Q` programs (walk.hpp CharClassAux::pairs): the EVENTS are the occurrences of Q; the k-th of the haystack opens row
  // k / 2 when k is even and closes it (exclusive end: the byte behind it) when k is odd.  Pass 1 keeps the occurrence bitmap and
  // counts events; which of them are starts is known behind the look-back, from the parity of the events in front of the tile.
  const bool pairs = ax->pairs != 0u;
#pragma unroll
  for (int q = 0; q < 4; q++) { rg.lo4[q] = ax->lo[q] * 0x01010101u; rg.hi4[q] = (0x7Fu - ax->hi[q]) * 0x01010101u; }
  // round 5: the ranges as a class plan (wave_common.hpp: `\w` in 9 instructions per dword instead of 15); shape 0 keeps notset4
  static_assert(CXG_CC_PLANS == 0 || CXG_CC_PLANS == 1, "");
  const ClassPlan& plan = a.plan;                                     // kernel arguments: scalar loads
  const int shape = CXG_CC_PLANS ? static_cast<int>(a.plan_shape) : 0;
  __syncthreads();
  const uint64_t group = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_group >> 32))) << 32) |
                         static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_group)));
  if (group >= a.ngroups) return;
  if (limit_reached_skip(a, group, &s_base)) return;                 // FindAll with n > 0 (block_common.hpp)
  uint32_t fallback = 0;

  static_assert(CXG_CC_DEPTH == 1 || CXG_CC_DEPTH == 2 || CXG_CC_DEPTH == 4, "");
  u32x4 xs[CXG_CC_DEPTH][4];
  uint32_t xprevs[CXG_CC_DEPTH];
  auto issue_loads = [&](int jj, u32x4 (&x)[4], uint32_t& xprev) {
    const uint64_t wtn = group * (kWavesPerBlock * kCcTilesPerWave) + static_cast<uint64_t>(jj) * kWavesPerBlock + wave;
    const uint64_t lo = wtn * static_cast<uint64_t>(kWaveTile);
    int nrec = 0;
    if (jj < kCcTilesPerWave && lo < a.len) {
      const uint64_t rem = a.len - lo;
      nrec = rem >= static_cast<uint64_t>(kWin) ? kWin : static_cast<int>((rem + 3) & ~3ull);
    }
    const int pre = (nrec && lo) ? 16 : 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.hay) + (nrec ? lo - pre : 0), 0, nrec + pre, 0x00020000);
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (lane + 64 * k) << 4, pre, CXG_CC_LOAD_AUX);
    xprev = __builtin_amdgcn_raw_buffer_load_b32(rsrc, 0, pre ? 12 : nrec + pre, 0);
  };
#pragma unroll
  for (int d = 0; d < CXG_CC_DEPTH; d++) issue_loads(d, xs[d], xprevs[d]);

  // ---- pass 1: bitmaps and counts
#if CXG_CC_DEPTH > 1
#pragma unroll
#endif
  for (int j = 0; j < kCcTilesPerWave; j++) {
    u32x4 (&x)[4] = xs[j % CXG_CC_DEPTH];
    uint32_t& xprev = xprevs[j % CXG_CC_DEPTH];
    lane = lane0;
    asm volatile("" : "+v"(lane));
    const uint64_t wt = group * (kWavesPerBlock * kCcTilesPerWave) + static_cast<uint64_t>(j) * kWavesPerBlock + wave;
    const uint64_t tile_lo = wt * static_cast<uint64_t>(kWaveTile);
    uint64_t S = 0, E = 0;
    uint32_t n = 0;
    if (tile_lo < a.len) {
      const uint64_t remaining = a.len - tile_lo;
      const int32_t rend = remaining > 0x7FFF0000ull ? 0x7FFF0000 : static_cast<int32_t>(remaining);
      const int32_t stage = rend < kWin ? rend : kWin;
      uint16_t* pieces = reinterpret_cast<uint16_t*>(s_m[wave]);
      with_shape(shape, [&]<int SHAPE>() {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          pieces[lane + 64 * k] = static_cast<uint16_t>(notshape16<SHAPE>(x[k], plan, rg) ^ flip);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      const uint32_t xprev_cur = xprev;
      if (CXG_CC_DEPTH == 1 || j + CXG_CC_DEPTH < kCcTilesPerWave) issue_loads(j + CXG_CC_DEPTH, x, xprev);
      wave_lds_sync();
      uint64_t M = s_m[wave][lane];
      if (stage != kWin) {                                          // short last window: nothing past the data is a member
        const int32_t nv = stage - 64 * lane;
        M &= nv <= 0 ? 0ull : (nv >= 64 ? ~0ull : ((1ull << nv) - 1ull));
      }
      // the byte in front of the tile
      uint32_t prev_member = 0;
      if (tile_lo > 0) {
        const uint32_t pb = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(xprev_cur))) >> 24;
#pragma unroll
        for (int q = 0; q < 4; q++) if (static_cast<uint32_t>(q) < ax->nr && pb >= ax->lo[q] && pb <= ax->hi[q]) prev_member = 1;
        prev_member ^= ax->neg ? 1u : 0u;
      }
      uint64_t carry = from_lower64(M) >> 63;                      // DPP outside lane-dependent branches
      if (lane == 0) carry = prev_member;
      const uint64_t P = (M << 1) | carry;                          // "the previous byte is a member"
      S = M & ~P;
      E = ~M & P;                                                   // exclusive end: first non-member after a run
      if (pairs) { S = M; E = 0; prev_member = 0; }                 // every occurrence is an event of the tile that holds it
      S &= word_range(lane, 0, kWaveTile - 1);                      // starts at [0, 3840), exclusive ends at (0, 3840]
      E &= word_range(lane, 1, kWaveTile);
      const uint32_t ns = static_cast<uint32_t>(__popcll(S)), ne = static_cast<uint32_t>(__popcll(E));
      const uint32_t incl = wave_inclusive_sum(ns | (ne << 16));
      const uint32_t tot = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
      n = tot & 0xFFFFu;
      const uint32_t n_ends = tot >> 16;
      // a run crosses the tile's first byte: the byte in front and the first byte are both members
      const uint32_t open = prev_member & static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(M) & 1u)));
      if (n > static_cast<uint32_t>(kCcStage) || n_ends > static_cast<uint32_t>(kCcStage)) fallback |= 8;
      s_S[wave][j][lane] = S;
      s_E[wave][j][lane] = E;
      if (lane == 0) s_cnt[wave][j] = n | (n_ends << 12) | (open << 31);   // n, n_ends <= 3840 < 4096
    } else {
      s_S[wave][j][lane] = 0; s_E[wave][j][lane] = 0;
      if (lane == 0) s_cnt[wave][j] = 0;
    }
  }
  if (fallback != 0 && lane0 == 0) raise_err(a.err, 8u | (fallback << 8));
  __syncthreads();

  // ---- group: exclusive prefix over the wave-tiles q = j*4 + wave, look-back
  if (tid < 64) {
    const int q = tid;
    const uint32_t v = (q < kWavesPerBlock * kCcTilesPerWave) ? (s_cnt[q % kWavesPerBlock][q / kWavesPerBlock] & 0xFFFu) : 0u;
    const uint32_t incl = wave_inclusive_sum(v);
    if (q < kWavesPerBlock * kCcTilesPerWave) s_qbase[q] = incl - v;
    if (q == kWavesPerBlock * kCcTilesPerWave - 1) s_qbase[kWavesPerBlock * kCcTilesPerWave] = incl;
  }
  __syncthreads();
  const uint32_t total = s_qbase[kWavesPerBlock * kCcTilesPerWave];
  tile_lookback(a.status, a.total, a.err, group, a.ngroups, total, &s_base, a.epoch, a.limit, a.stop);
  if (pairs && group == a.ngroups - 1 && tid == 0) *a.total = (s_base + total) >> 1;   // rows = events / 2 (an unpaired last Q opens nothing)
  if (a.out == nullptr) return;

  // ---- pass 2: starts and ends straight to their rows
  const uint64_t base = s_base;
  const bool u32 = a.u32_rows != 0u;                                 // compact rows: (start, end) as uint32 relative to the haystack
  uint32_t* const out32 = reinterpret_cast<uint32_t*>(a.out);
  const int64_t origin = (u32 ? 0 : a.base) + static_cast<int64_t>(group * static_cast<uint64_t>(kWaveTile) * kWavesPerBlock * kCcTilesPerWave);
  for (int j = 0; j < kCcTilesPerWave; j++) {
    const uint32_t cn = s_cnt[wave][j];
    uint32_t n = cn & 0xFFFu, n_ends = (cn >> 12) & 0xFFFu, open = cn >> 31;
    if (n == 0 && n_ends == 0) continue;
    const uint64_t S = s_S[wave][j][lane0], E = s_E[wave][j][lane0];
    const uint32_t ns = static_cast<uint32_t>(__popcll(S)), ne = static_cast<uint32_t>(__popcll(E));
    const uint32_t incl = wave_inclusive_sum(ns | (ne << 16));
    uint64_t row0 = base + s_qbase[j * kWavesPerBlock + wave];
    if (pairs) {
      // events in front of the tile: row0 of them; event k of the haystack is the start of row k / 2 (k even) or its end (k odd)
      const uint32_t par = static_cast<uint32_t>(row0) & 1u;
      uint32_t k = par + (incl & 0xFFFFu) - ns;                    // parity-true index of this lane's first event, counted from an even event
      uint64_t sb = S;
      while (sb) {
        const int bit = __builtin_ctzll(sb);
        sb &= sb - 1;
        // k = par + (rank in the tile): even k opens, odd k closes.  Ranks among the tile's starts / ends: (k >> 1) - par / k >> 1
        // (par == 1: the tile's first event, k = 1, is end 0 and closes a row opened in front of the tile)
        const uint32_t slot = (k & 1u) ? (k >> 1) : (k >> 1) - par;
        if (slot < static_cast<uint32_t>(kCcStage)) {
          if (k & 1u) s_re[wave][cc_slot(slot)] = static_cast<uint16_t>(64 * lane0 + bit + 1);   // closes: exclusive end behind the Q
          else s_rs[wave][cc_slot(slot)] = static_cast<uint16_t>(64 * lane0 + bit);
        }
        k++;
      }
      // starts: events with even k; ends: odd k.  par == 1: the tile's first event closes a row opened earlier (open)
      n_ends = par ? (n + 1u) >> 1 : n >> 1;
      n = par ? n >> 1 : (n + 1u) >> 1;
      open = par;
      row0 = (row0 + 1u) >> 1;                                       // rows opened in front of the tile
    } else {
#if CXG_CC_MERGED
    uint32_t r = (incl & 0xFFFFu) - ns, q = (incl >> 16) - ne;
    uint64_t sb = S, eb = E;
    while (sb | eb) {                                               // one loop for both bitmaps: a lane has as many ends as starts, give or take one
      if (sb) {
        const int bit = __builtin_ctzll(sb);
        sb &= sb - 1;
        if (r < static_cast<uint32_t>(kCcStage)) s_rs[wave][cc_slot(r)] = static_cast<uint16_t>(64 * lane0 + bit);
        r++;
      }
      if (eb) {
        const int bit = __builtin_ctzll(eb);
        eb &= eb - 1;
        if (q < static_cast<uint32_t>(kCcStage)) s_re[wave][cc_slot(q)] = static_cast<uint16_t>(64 * lane0 + bit);
        q++;
      }
    }
#elif CXG_CC_HALVES
    // both halves of the word per iteration: the loop runs max(popcount of a half) times instead of popcount of the word.  No bound
    // check on the rank: cc_slot masks it, and a tile with more than kCcStage starts or ends has raised the fallback flag.
    static_assert(CXG_CC_SWIZZLE, "cc_slot must mask");
    {
      uint32_t b0 = static_cast<uint32_t>(S), b1 = static_cast<uint32_t>(S >> 32);
      uint32_t r0 = (incl & 0xFFFFu) - ns, r1 = r0 + static_cast<uint32_t>(__popc(b0));
      while (b0 | b1) {
        if (b0) { const int bit = __builtin_ctz(b0); b0 &= b0 - 1; s_rs[wave][cc_slot(r0)] = static_cast<uint16_t>(64 * lane0 + bit); r0++; }
        if (b1) { const int bit = __builtin_ctz(b1); b1 &= b1 - 1; s_rs[wave][cc_slot(r1)] = static_cast<uint16_t>(64 * lane0 + 32 + bit); r1++; }
      }
      b0 = static_cast<uint32_t>(E); b1 = static_cast<uint32_t>(E >> 32);
      r0 = (incl >> 16) - ne; r1 = r0 + static_cast<uint32_t>(__popc(b0));
      while (b0 | b1) {
        if (b0) { const int bit = __builtin_ctz(b0); b0 &= b0 - 1; s_re[wave][cc_slot(r0)] = static_cast<uint16_t>(64 * lane0 + bit); r0++; }
        if (b1) { const int bit = __builtin_ctz(b1); b1 &= b1 - 1; s_re[wave][cc_slot(r1)] = static_cast<uint16_t>(64 * lane0 + 32 + bit); r1++; }
      }
    }
#else
    uint32_t r = (incl & 0xFFFFu) - ns;
    uint64_t sb = S;
    while (sb) {
      const int bit = __builtin_ctzll(sb);
      sb &= sb - 1;
      if (r < static_cast<uint32_t>(kCcStage)) s_rs[wave][cc_slot(r)] = static_cast<uint16_t>(64 * lane0 + bit);
      r++;
    }
    r = (incl >> 16) - ne;
    uint64_t eb = E;
    while (eb) {
      const int bit = __builtin_ctzll(eb);
      eb &= eb - 1;
      if (r < static_cast<uint32_t>(kCcStage)) s_re[wave][cc_slot(r)] = static_cast<uint16_t>(64 * lane0 + bit);
      r++;
    }
#endif
    }
    wave_lds_sync();
    const int64_t tb = origin + static_cast<int64_t>(j * kWavesPerBlock + wave) * kWaveTile;
    const uint32_t nst = n < static_cast<uint32_t>(kCcStage) ? n : static_cast<uint32_t>(kCcStage);
    const uint32_t nen = n_ends < static_cast<uint32_t>(kCcStage) ? n_ends : static_cast<uint32_t>(kCcStage);
    for (uint32_t i = lane0; i < nst; i += 64) {
      if (row0 + i < a.cap) {
        if (i + open < nen) {                                       // both halves of the row are this tile's
          longlong2 v; v.x = tb + s_rs[wave][cc_slot(i)]; v.y = tb + s_re[wave][cc_slot(i + open)];
          if (u32) store_pair32_nt(out32 + (row0 + i) * 2, static_cast<uint32_t>(v.x), static_cast<uint32_t>(v.y));
          else store_pair_nt(a.out + (row0 + i) * 2, v.x, v.y);
        } else if (u32) out32[(row0 + i) * 2] = static_cast<uint32_t>(tb + s_rs[wave][cc_slot(i)]);
        else a.out[(row0 + i) * 2] = tb + s_rs[wave][cc_slot(i)];  // the run ends in a later tile
      }
    }
    if (open && nen != 0 && lane0 == 0 && row0 - 1 < a.cap) {       // a run begun in an earlier tile ends here
      if (u32) out32[(row0 - 1) * 2 + 1] = static_cast<uint32_t>(tb + s_re[wave][cc_slot(0)]);
      else a.out[(row0 - 1) * 2 + 1] = tb + s_re[wave][cc_slot(0)];
    }
    wave_lds_sync();                                                // staging is reused by the next tile
  }
}

hipError_t launch_scan_charclass_wave(const ScanArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(k_scan_charclass_wave, dim3(static_cast<unsigned>(a.ngroups)), dim3(kThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace cxgdev
