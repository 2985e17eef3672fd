// scan_charclass_wave.hip — FindAll for UseCharClassSearcher (`[class]+`) with the wave as the unit of work.
// GPU form of nfa.CharClassSearcher.FindAllIndices (nfa/charclass_searcher.go:158-211): maximal runs of member
// bytes, in order, a trailing run closed at the end of input.
//
// Output dominates this path (one 16-byte span per ~5.5 bytes of log text: 3 bytes written per byte read), so a
// group's rows are never buffered as a whole: membership words wait in LDS for the group's base, rows are staged one
// wave-tile at a time.
//   pass 1  per wave-tile (60 x 64 B = 3840 B + 256 B halo = 64 bitmap words, one per lane): window by four
//           buffer_load_dwordx4 per lane (two windows in flight per wave), membership by SWAR range tests (class plans,
//           wave_common.hpp) + v_dot4 gather, 16-bit pieces through LDS -> word M per lane, which stays there;
//             starts S = M & ~(M << 1 | carry),  ends E = ~M & (M << 1 | carry)   (exclusive ends)
//           starts and ends are owned SEPARATELY: the tile owns the starts at its bytes [0, 3840) and the exclusive
//           ends at (0, 3840] — a run may be as long as the haystack, nobody has to see both of its ends.  Runs
//           alternate start, end, start, end, so the k-th start and the k-th end of the haystack are row k: with
//           B = number of starts in front of the tile (look-back), the tile's i-th start is row B + i and its j-th
//           end is row B - open + j, open = 1 when a run crosses the tile's first byte.  The start counts are summed per tile;
//   group   (4 waves x 4 wave-tiles = 60 KiB) one barrier, look-back -> base of every tile (CXG_CC_EARLY=1 stages the first tile of
//           every wave in front of the look-back: measured, no gain);
//   pass 2  per wave-tile S and E again from M (a DPP shift and six bit operations), every lane drops the starts and ends it
//           holds into the wave's LDS staging at their ranks — four streams per lane, no exec-masked branch (see `drop`) — then the
//           wave writes the rows that have both halves here as fully coalesced 16-byte stores (1 KiB per instruction); the end
//           of a run begun in an earlier tile and the start of a run that ends in a later one go out as lone 8-byte stores.
// LDS: 25.2 KB per workgroup (M 8 KB, staging 16 KB, dump slots) — six workgroups per CU.  Round 5 measurements behind this form
// (profiles/README.md, calls 8-18): parking S and E (34.7 KB, four workgroups), one window in flight, the class read from the
// program image per tile, one bit per iteration of exec-masked 64-bit loops: 1.08 ms per GiB of config 4; this form 0.86-0.93.
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
#ifndef CXG_CC_LOAD_AUX
#define CXG_CC_LOAD_AUX 0                                    // cache policy of the haystack loads (2 = nt; A/B: no difference)
#endif
// Windows in flight per wave in pass 1 (1, 2 or 4 = all of the wave's tiles issued before the first is classified; 4 needs 84
// VGPRs: five waves per SIMD).
#ifndef CXG_CC_DEPTH
#define CXG_CC_DEPTH 2
#endif
#ifndef CXG_CC_OCC
#define CXG_CC_OCC 6
#endif
// Stage the wave's first tile between publishing the group's total and the look-back (1) or behind it like the others (0).
#ifndef CXG_CC_EARLY
#define CXG_CC_EARLY 0                                       // (measured: 1.016 against 1.003 ms — no gain)
#endif
// Start-up stagger in shader-clock cycles per resident slot (0 = off): the workgroups that fill the device at the start of a launch all
// read, then all wait for the look-back, then all extract and write — phase by phase in step, so that the time of a launch is the SUM of
// a memory-bound and an issue-bound phase.  Workgroup b of the first 256 x OCC waits (b / 256) x this before it begins.
#ifndef CXG_CC_STAGGER
#define CXG_CC_STAGGER 0
#endif

namespace cxgdev {

namespace {
constexpr int kWin = kWaveTile + kWaveHalo;       // 4096
constexpr int kCcStage = 1024;                    // rows staged per wave-tile (typical log text: ~700)
// Staging slot of rank r (pass 2).  In one ds_write_b16 lane l writes rank r_l + i with r_l ~ 11 l on log text: slots r / 2 mod 32
// would put ~11 lanes on every LDS bank.  r * 13 mod 1024 (a bijection) spreads them — the coalesced read-out of consecutive ranks
// stays two lanes per bank.  Byte offset = (r * 26) & 2046, OR-ed onto the wave's 2 KB-aligned staging array.
__device__ __forceinline__ uint32_t cc_slot(uint32_t r) { return (r * 13u) & static_cast<uint32_t>(kCcStage - 1); }
typedef __attribute__((address_space(3))) uint16_t lds_u16;
__device__ __forceinline__ uint32_t lds_off(const void* p) { return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const void*)p)); }
__device__ __forceinline__ void lds_st16(uint32_t addr, uint32_t v) { *reinterpret_cast<lds_u16*>(addr) = static_cast<uint16_t>(v); }
__device__ __forceinline__ uint32_t lds_ld16(uint32_t addr) { return *reinterpret_cast<lds_u16*>(addr); }
}

__global__ __launch_bounds__(kThreads, CXG_CC_OCC) void k_scan_charclass_wave(ScanArgs a) {
  __shared__ __attribute__((aligned(16))) uint64_t s_M[kWavesPerBlock][kCcTilesPerWave][64];   // membership words (the 16-bit pieces land here)
  __shared__ __attribute__((aligned(2048))) uint16_t s_rs[kWavesPerBlock][kCcStage];   // pass 2 staging: start / end inside the window
  __shared__ __attribute__((aligned(2048))) uint16_t s_re[kWavesPerBlock][kCcStage];   // (a wave's 2 KB: aligned, so that offsets are OR-ed on)
  __shared__ uint16_t s_dump[kWavesPerBlock][64];                 // where an exhausted stream of pass 2 writes (one slot per lane)
  __shared__ uint32_t s_cnt[kWavesPerBlock][kCcTilesPerWave];
  __shared__ uint32_t s_qbase[kWavesPerBlock * kCcTilesPerWave + 1];
  __shared__ uint64_t s_group;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = lane0;
#if CXG_CC_STAGGER
  if (a.ngroups > 4u * 256u * CXG_CC_OCC && blockIdx.x < 256u * CXG_CC_OCC && blockIdx.x >= 256u) {
    const uint64_t wait = static_cast<uint64_t>(blockIdx.x / 256u) * CXG_CC_STAGGER, t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
#endif
  const uint64_t pt0 = a.prof ? __builtin_readcyclecounter() : 0ull;   // CXG_PROF=1: wave 0's timestamps per phase, one record per workgroup
  // the class (walk.hpp CharClassAux) travels as kernel arguments: read from the program image it cost two dependent scalar loads
  // before the first window could be requested, and three scalar loads per tile — which were the bound of pass 1 (0.36 -> 0.26 ms
  // for a count-only launch over 1 GiB, profiles/r05_c15_cfg4.txt)
  SetRanges rg;
  rg.n = a.cc_nr;
  const uint32_t cls_nr = a.cc_nr, cls_neg = a.cc_neg ? 1u : 0u;
  const uint32_t flip = cls_neg ? 0u : 0xFFFFu;                     // notset4 flags the bytes OUTSIDE the ranges: the members of a complemented class
  // `Q[^Q]*Q` programs (walk.hpp CharClassAux::pairs): the EVENTS are the occurrences of Q; the k-th of the haystack opens row
  // k / 2 when k is even and closes it (exclusive end: the byte behind it) when k is odd.  Pass 1 keeps the occurrence bitmap and
  // counts events; which of them are starts is known behind the look-back, from the parity of the events in front of the tile.
  const bool pairs = a.cc_pairs != 0u;
  uint32_t cls_lo[4], cls_hi[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { cls_lo[q] = a.cc_lo[q]; cls_hi[q] = a.cc_hi[q]; rg.lo4[q] = cls_lo[q] * 0x01010101u; rg.hi4[q] = (0x7Fu - cls_hi[q]) * 0x01010101u; }
  // the ranges as a class plan (wave_common.hpp: `\w` in 9 instructions per dword instead of 15); shape 0 keeps notset4
  static_assert(CXG_CC_PLANS == 0 || CXG_CC_PLANS == 1, "");
  const ClassPlan& plan = a.plan;                                     // kernel arguments: scalar loads
  const int shape = CXG_CC_PLANS ? static_cast<int>(a.plan_shape) : 0;
  uint64_t group = blockIdx.x;                                       // static groups: no claim, no barrier in front of the first loads
  if (!a.static_groups) {
    if (tid == 0) s_group = claim_group(false, a.ticket, a.ngroups);
    __syncthreads();
    group = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_group >> 32))) << 32) |
            static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_group)));
  }
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
  const uint64_t pt1 = a.prof ? __builtin_readcyclecounter() : 0ull;

  // ---- pass 1: membership words and counts
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
    uint32_t n = 0;
    if (tile_lo < a.len) {
      const uint64_t remaining = a.len - tile_lo;
      const int32_t rend = remaining > 0x7FFF0000ull ? 0x7FFF0000 : static_cast<int32_t>(remaining);
      const int32_t stage = rend < kWin ? rend : kWin;
      uint16_t* pieces = reinterpret_cast<uint16_t*>(s_M[wave][j]);
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
      uint64_t M = s_M[wave][j][lane];
      if (stage != kWin) {                                          // short last window: nothing past the data is a member
        const int32_t nv = stage - 64 * lane;
        M &= nv <= 0 ? 0ull : (nv >= 64 ? ~0ull : ((1ull << nv) - 1ull));
        s_M[wave][j][lane] = M;                                     // pass 2 reads the masked word
      }
      // the byte in front of the tile
      uint32_t prev_member = 0;
      if (tile_lo > 0) {
        const uint32_t pb = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(xprev_cur))) >> 24;
#pragma unroll
        for (int q = 0; q < 4; q++) prev_member |= (static_cast<uint32_t>(q) < cls_nr && pb >= cls_lo[q] && pb <= cls_hi[q]) ? 1u : 0u;
        prev_member ^= cls_neg;
      }
      uint64_t carry = from_lower64(M) >> 63;                      // DPP outside lane-dependent branches
      if (lane == 0) carry = prev_member;
      const uint64_t P = (M << 1) | carry;                          // "the previous byte is a member"
      uint64_t S = M & ~P;
      uint64_t E = ~M & P;                                          // exclusive end: first non-member after a run
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
      if (lane == 0) s_cnt[wave][j] = n | (n_ends << 12) | (prev_member << 30) | (open << 31);   // n, n_ends <= 3840 < 4096
    } else {
      if (lane == 0) s_cnt[wave][j] = 0;
    }
  }
  if (fallback != 0 && lane0 == 0) raise_err(a.err, 8u | (fallback << 8));
  const uint64_t pt2 = a.prof ? __builtin_readcyclecounter() : 0ull;
  __syncthreads();
  const uint64_t pt3 = a.prof ? __builtin_readcyclecounter() : 0ull;

  // ---- group: exclusive prefix over the wave-tiles q = j*4 + wave
  if (tid < 64) {
    const int q = tid;
    const uint32_t v = (q < kWavesPerBlock * kCcTilesPerWave) ? (s_cnt[q % kWavesPerBlock][q / kWavesPerBlock] & 0xFFFu) : 0u;
    const uint32_t incl = wave_inclusive_sum(v);
    if (q < kWavesPerBlock * kCcTilesPerWave) s_qbase[q] = incl - v;
    if (q == kWavesPerBlock * kCcTilesPerWave - 1) s_qbase[kWavesPerBlock * kCcTilesPerWave] = incl;
  }
  __syncthreads();
  const uint32_t total = s_qbase[kWavesPerBlock * kCcTilesPerWave];

  // S and E of tile j again from its parked M, the tile's counts word cn
  const uint32_t p0 = 64u * static_cast<uint32_t>(lane0), p1 = p0 + 32u;
  const uint32_t rsb = lds_off(s_rs[wave]), reb = lds_off(s_re[wave]);
  auto owned = [&](int j, uint32_t cn, uint64_t& S, uint64_t& E) {
    const uint64_t M = s_M[wave][j][lane0];
    uint64_t carry = from_lower64(M) >> 63;
    if (lane0 == 0) carry = (cn >> 30) & 1u;                        // the byte in front of the tile (pass 1)
    const uint64_t P = (M << 1) | carry;
    S = M & ~P;
    E = ~M & P;
    if (pairs) { S = M; E = 0; }
    S &= word_range(lane0, 0, kWaveTile - 1);
    E &= word_range(lane0, 1, kWaveTile);
  };
  // Every lane drops the bits of its S and E words at their ranks into the staging arrays: four streams per lane (the 32-bit halves
  // of both words), one bit of each per iteration, NO exec-masked branch inside — an exhausted stream writes to the lane's dump slot
  // — and one uniform branch per iteration (any stream of any lane left?).  The wave kernels are bound by instruction issue (a SIMD
  // issues about one instruction of whatever kind per 4 cycles: measured), so the scalar mask juggling and the
  // three branches per iteration of the masked form cost as much as its arithmetic.  No bound check on the rank: the slot is masked,
  // and a tile with more than kCcStage starts or ends has raised the fallback flag.
  const uint32_t dumpa = lds_off(&s_dump[wave][lane0]);
  auto step = [&](uint32_t& b, uint32_t& ofs, uint32_t base, uint32_t pos) {
    uint32_t bit;
    asm("v_ffbl_b32 %0, %1" : "=v"(bit) : "v"(b));                  // -1 for an empty stream: the value goes to the dump slot
    lds_st16(b != 0u ? ((ofs & 2046u) | base) : dumpa, pos | bit);
    b &= b - 1u;
    ofs += 26u;
  };
  auto drop = [&](uint64_t S, uint32_t srank, uint64_t E, uint32_t erank) {
    uint32_t s0 = static_cast<uint32_t>(S), s1 = static_cast<uint32_t>(S >> 32), e0 = static_cast<uint32_t>(E), e1 = static_cast<uint32_t>(E >> 32);
    uint32_t as0 = srank * 26u, as1 = as0 + static_cast<uint32_t>(__popc(s0)) * 26u;
    uint32_t ae0 = erank * 26u, ae1 = ae0 + static_cast<uint32_t>(__popc(e0)) * 26u;
    while (__builtin_amdgcn_ballot_w64((s0 | s1 | e0 | e1) != 0u) != 0ull) {
      step(s0, as0, rsb, p0);
      step(s1, as1, rsb, p1);
      step(e0, ae0, reb, p0);
      step(e1, ae1, reb, p1);
    }
  };
  auto stage_runs = [&](int j, uint32_t cn) {                        // (not for pairs: their roles depend on the base)
    uint64_t S, E;
    owned(j, cn, S, E);
    const uint32_t ns = static_cast<uint32_t>(__popcll(S)), ne = static_cast<uint32_t>(__popcll(E));
    const uint32_t incl = wave_inclusive_sum(ns | (ne << 16));
    drop(S, (incl & 0xFFFFu) - ns, E, (incl >> 16) - ne);
  };

  // the group's total goes out first (tile_lookback stores the same word again), then the wave's first tile is staged — it
  // needs no base, and the groups in front are given that much more time before this one polls them
  const bool early = CXG_CC_EARLY && a.out != nullptr && !pairs;
  if (early) {
    if (tid == 0)
      __hip_atomic_store(a.status + group, (group == 0 ? kFlagInclusive : kFlagAggregate) | (static_cast<uint64_t>(a.epoch) << kEpochShift) | static_cast<uint64_t>(total),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t cn = s_cnt[wave][0];
    if ((cn & 0xFFFFFFu) != 0u) stage_runs(0, cn);
  }
  tile_lookback(a.status, a.total, a.err, group, a.ngroups, total, &s_base, a.epoch, a.limit, a.stop);
  if (pairs && group == a.ngroups - 1 && tid == 0) *a.total = (s_base + total) >> 1;   // rows = events / 2 (an unpaired last Q opens nothing)
  const uint64_t pt4 = a.prof ? __builtin_readcyclecounter() : 0ull;
  auto prof_out = [&]() {
    if (a.prof && tid == 0 && group < (1u << 18)) {
      uint64_t* r = a.prof + 16 + group * 8;
      r[0] = pt0; r[1] = pt1; r[2] = pt2; r[3] = pt3; r[4] = pt4; r[5] = __builtin_readcyclecounter();
    }
  };
  if (a.out == nullptr) { prof_out(); return; }

  // ---- pass 2: starts and ends straight to their rows
  const uint64_t base = s_base;
  const bool u32 = a.u32_rows != 0u;                                 // compact rows: (start, end) as uint32 relative to the haystack
  uint32_t* const out32 = reinterpret_cast<uint32_t*>(a.out);
  const int64_t origin = (u32 ? 0 : a.base) + static_cast<int64_t>(group * static_cast<uint64_t>(kWaveTile) * kWavesPerBlock * kCcTilesPerWave);
  for (int j = 0; j < kCcTilesPerWave; j++) {
    const uint32_t cn = s_cnt[wave][j];
    uint32_t n = cn & 0xFFFu, n_ends = (cn >> 12) & 0xFFFu, open = cn >> 31;
    if (n == 0 && n_ends == 0) continue;
    uint64_t row0 = base + s_qbase[j * kWavesPerBlock + wave];
    if (pairs) {
      uint64_t S, E;
      owned(j, cn, S, E);
      const uint32_t ns = static_cast<uint32_t>(__popcll(S));
      const uint32_t incl = wave_inclusive_sum(ns);
      // events in front of the tile: row0 of them; event k of the haystack is the start of row k / 2 (k even) or its end (k odd)
      const uint32_t par = static_cast<uint32_t>(row0) & 1u;
      uint32_t k = par + incl - ns;                                  // parity-true index of this lane's first event, counted from an even event
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
    } else if (!(early && j == 0)) {
      stage_runs(j, cn);
    }
    wave_lds_sync();
    const int64_t tb = origin + static_cast<int64_t>(j * kWavesPerBlock + wave) * kWaveTile;
    const uint32_t nst = n < static_cast<uint32_t>(kCcStage) ? n : static_cast<uint32_t>(kCcStage);
    const uint32_t nen = n_ends < static_cast<uint32_t>(kCcStage) ? n_ends : static_cast<uint32_t>(kCcStage);
    // rows of this tile that fit the output array (uniform): no per-lane capacity test in the loops.  Rows [0, both) have their end
    // in this tile; at most one more start follows (its run ends in a later tile: starts and ends alternate).
    const uint32_t fit = row0 >= a.cap ? 0u : (a.cap - row0 < nst ? static_cast<uint32_t>(a.cap - row0) : nst);
    const uint32_t both = nen > open ? nen - open : 0u;
    const uint32_t npair = both < fit ? both : fit;
    uint32_t ao = static_cast<uint32_t>(lane0) * 26u, ae = ao + open * 26u;
    if (!u32) {
      int64_t* po = a.out + (row0 + lane0) * 2;
      for (uint32_t i = lane0; i < npair; i += 64, po += 128, ao += 64u * 26u, ae += 64u * 26u)
        store_pair_nt(po, tb + static_cast<int64_t>(lds_ld16((ao & 2046u) | rsb)), tb + static_cast<int64_t>(lds_ld16((ae & 2046u) | reb)));
      if (both < fit && lane0 == 0) a.out[(row0 + both) * 2] = tb + s_rs[wave][cc_slot(both)];
    } else {
      uint32_t* po = out32 + (row0 + lane0) * 2;
      for (uint32_t i = lane0; i < npair; i += 64, po += 128, ao += 64u * 26u, ae += 64u * 26u)
        store_pair32_nt(po, static_cast<uint32_t>(tb) + lds_ld16((ao & 2046u) | rsb), static_cast<uint32_t>(tb) + lds_ld16((ae & 2046u) | reb));
      if (both < fit && lane0 == 0) out32[(row0 + both) * 2] = static_cast<uint32_t>(tb) + s_rs[wave][cc_slot(both)];
    }
    if (open && nen != 0 && lane0 == 0 && row0 - 1 < a.cap) {       // a run begun in an earlier tile ends here
      if (u32) out32[(row0 - 1) * 2 + 1] = static_cast<uint32_t>(tb + s_re[wave][cc_slot(0)]);
      else a.out[(row0 - 1) * 2 + 1] = tb + s_re[wave][cc_slot(0)];
    }
    wave_lds_sync();                                                // staging is reused by the next tile
  }
  prof_out();
}

hipError_t launch_scan_charclass_wave(const ScanArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(k_scan_charclass_wave, dim3(static_cast<unsigned>(a.ngroups)), dim3(kThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace cxgdev
