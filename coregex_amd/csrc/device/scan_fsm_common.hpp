// scan_fsm_common.hpp — what the two translation units of the transducer kernels share (scan_fsm.hip: k_scan_fsm with the machinery for
// entry states that do not collapse; scan_fsml.hip: the lean kernel k_scan_fsml): the LDS window, the buffer geometry by match density,
// the rows of a tile from its event bits.  Everything has internal linkage.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "block_common.hpp"
#include "fsm.hpp"
#include "scan_dfa.h"
#include "wave_common.hpp"

// -DCXG_FSM_PROF=1 (experiments only): s_memtime at the phase boundaries, cycles summed into ScanArgs::prof[8..15]
#ifndef CXG_FSM_PROF
#define CXG_FSM_PROF 0
#endif
// -DCXG_FSM_ABL=n (experiments only, results WRONG): bit 0 = no entry-state walks, bit 1 = no lockstep walk of the
// chunk, bit 2 = no row gathering / starts.  Attributes instruction counts to the phases.
#ifndef CXG_FSM_ABL
#define CXG_FSM_ABL 0
#endif
// -DCXG_FSM_FAST_STARTS=0 (A/B): match starts by the loop alone, without the 16 branch-free steps in front of it (fsm.hpp fsm_match_start16)
#ifndef CXG_FSM_FAST_STARTS
#define CXG_FSM_FAST_STARTS 1
#endif
#if CXG_FSM_PROF
#define FSM_MARK(i) do { const uint64_t t_ = __builtin_readcyclecounter(); pacc[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define FSM_MARK(i) do { } while (0)
#endif

#ifndef CXG_FSML_OCC2
#define CXG_FSML_OCC2 4         // workgroups per CU the lean kernel's register allocation aims at in mode 2 (38 KB of LDS per workgroup)
#endif

namespace cxgdev {

namespace {

constexpr int kFsmStride = 68;                       // LDS bytes per 64-byte chunk
constexpr int kFsmWinBytes = 64 * kFsmStride + 16;   // 4352 per wave + the dword BEHIND the window (look-around: the kind of the byte behind a step)
constexpr int kFsmLeft = 64;                         // bytes staged in front of the tile
constexpr int32_t kFsmWinEnd = 4096 - kFsmLeft;      // tile-relative end of the window (192 bytes past the tile)

// The tile loop reads haystack bytes from the LDS window ONLY.  A load from HBM anywhere in the loop body — even on a
// path that is never taken — makes the compiler wait for vmcnt(0) at the join, i.e. for the window of the NEXT tile
// that is in flight: the prefetch would be worth nothing.  Walks that leave the window are finished elsewhere: a match
// start in front of the window in the epilogue (rows marked unresolved), a walk past the window's end by the fallback.
typedef __attribute__((address_space(3))) const uint8_t* lds_bytes_t;
template <int LOOK>
struct FsmMem : FsmClassify<FsmMem<LOOK>, LOOK> {
  lds_bytes_t win;         // this wave's LDS window: tile-relative bytes [-kFsmLeft, 4096 - kFsmLeft)
  int32_t last_;           // LOOK == 2: tile-relative position of the haystack's last byte (fsm.hpp "End of text")
  __device__ __forceinline__ int32_t last() const { return last_; }
  __device__ __forceinline__ uint32_t byte(int32_t r) const {
    const uint32_t w = static_cast<uint32_t>(r + kFsmLeft);
    return win[w + (w >> 6) * 4u];
  }
  __device__ __forceinline__ uint32_t dword(int32_t r) const {
    const uint32_t w = static_cast<uint32_t>(r + kFsmLeft);
    return *reinterpret_cast<__attribute__((address_space(3))) const uint32_t*>(win + w + (w >> 6) * 4u);
  }
  // the N + 1 bytes below e (fsm.hpp fsm_match_start16; N = 16 or 8): N / 4 + 1 aligned dwords, shifted into place
  template <int N>
  __device__ __forceinline__ void below(int32_t e, uint32_t (&W)[N / 4 + 1]) const {
    constexpr int D = N / 4 + 1;
    const uint32_t w0 = static_cast<uint32_t>(e - (N + 1) + kFsmLeft), wb = w0 & ~3u, sh = w0 & 3u;
    uint32_t d[D];
#pragma unroll
    for (int j = 0; j < D; j++) {
      const uint32_t w = wb + 4u * j;
      d[j] = *reinterpret_cast<__attribute__((address_space(3))) const uint32_t*>(win + w + (w >> 6) * 4u);
    }
#pragma unroll
    for (int j = 0; j < D; j++) W[j] = __builtin_amdgcn_alignbyte(d[j < D - 1 ? j + 1 : D - 1], d[j], sh);
  }
};
struct LdsRows {
  uint16_t* slot;          // this lane's kFsmLaneRows ends
  __device__ __forceinline__ void set_end(uint32_t r, int32_t e) { slot[r] = static_cast<uint16_t>(e); }
};
struct LdsEvents {
  uint16_t* slot;          // kFsmLaneEvents alias rows of one sub-chunk
  __device__ __forceinline__ void push(uint32_t k, uint32_t row) { slot[k] = static_cast<uint16_t>(row); }
  __device__ __forceinline__ uint32_t row_at(uint32_t k) const { return slot[k]; }
};

__device__ __forceinline__ FsmView view_of(const uint8_t* body, const FsmHeader* h) {   // body = image without its header, in LDS
  FsmView v;
  const uint32_t hs = static_cast<uint32_t>(sizeof(FsmHeader));
  v.tab = body;             // fixed layout (host/fsm.cc): the transition table first — at LDS address 0, see FsmLds
  v.cls2 = body + (h->cls_off - hs);
  v.rev = body + (h->rev_off - hs);
  v.ncls2 = 2u * h->ncls;
  v.alias_lo = h->alias_lo; v.u_lo = h->u_lo; v.top_off = h->top_off;
  v.rev_start_off = h->rev_start_off; v.rev_accept_off = h->rev_accept_off; v.rev_text_col = h->rev_text_col; v.end_col = h->end_col; v.rev_dead = h->rev_off - hs;
  v.create_lo = h->create_lo; v.rematch_lo = h->rematch_lo;
  v.mem = body + (h->mem_off - hs); v.row_shift = h->row_shift;
  v.knd = body + (h->knd_off - hs);
  v.lk16 = nullptr;          // (set by the kernel once it has filled its table)
  v.nk = h->nk;
  return v;
}

// MODE: 0 = 8 wave-tiles per wave and group, 512 rows buffered per wave; 1 = 2 wave-tiles (four times the row room per tile, match-dense
// input); 2 = 1 wave-tile, 1 024 rows per tile; 3 = 1 wave-tile, 2 048 rows (one match per 2 bytes); 2 and 3: 16 rows / 32 events per
// 32-byte sub-chunk.  The host escalates after an overflow and remembers the mode for the program (capi_ladder.hip).  Round 6 put
// mode 2 in front of the old one (now 3): with 2 048 rows the row buffers alone are 33 KB and two workgroups fit a CU — `\\b\\d+\\b`
// (535 rows per tile) ran 2.74 ms per GiB there and runs 1.77 with 1 024 rows and four workgroups (profiles/r06_c22_mode2_*).
template <int MODE> struct FsmMode {
  static constexpr int kTpw = MODE == 0 ? kTilesPerWave : (MODE == 1 ? kDenseTilesPerWave : 1);
  static constexpr int kRows = MODE >= 2 ? kFsmLaneRowsMax : kFsmLaneRows;
  static constexpr int kEvents = MODE >= 2 ? kFsmLaneEventsMax : kFsmLaneEvents;
  static constexpr int kRowsPerWave = MODE == 3 ? 2048 : (MODE == 2 ? 1024 : 512);
};

// Rows of a tile from the event bits of its lanes (fsm.hpp "Round 6"): an event is a row's end unless the event behind it is a rematch.
// KK: the lane's two sub-chunks (two bits per byte), active: the lane walked, owned: its rows are this tile's (lanes 1..60; the
// three lanes behind them only contribute their bits).  The ends land in s_re[nrows_w ...] in ascending order; returns their
// number.  pend_at_end(): pending levels of the row behind the lane's second sub-chunk (asked of lane 63 only, rarely).  tag: or-ed
// into every end (the lean kernel keeps the tile's number j in bits 12..14 — an end is at most 4032 — so that its epilogue can walk
// the wave's rows as ONE list).
template <int kRowsPerWave, class Pend>
__device__ __forceinline__ uint32_t fsm_rows_from_events(const uint64_t (&KK)[2], bool active, bool owned, int32_t rend, int32_t c0, int lane,
                                                         uint16_t* s_re_wave, uint32_t nrows_w, uint32_t& fallback, Pend pend_at_end, uint32_t tag = 0u) {
  uint64_t k0 = active ? KK[0] : 0ull, k1 = active ? KK[1] : 0ull;
  if (rend < kFsmWinEnd) { k0 &= fsm_valid_bits(rend - c0); k1 &= fsm_valid_bits(rend - c0 - kFsmSub); }   // (uniform: the input ends inside this window)
  const uint32_t T[4] = {static_cast<uint32_t>(k0), static_cast<uint32_t>(k0 >> 32), static_cast<uint32_t>(k1), static_cast<uint32_t>(k1 >> 32)};
  const bool ne = (T[0] | T[1] | T[2] | T[3]) != 0u;
  const unsigned long long NE = __ballot(ne), FR = __ballot(ne && fsm_first_is_r(T));
  const unsigned long long LN = fsm_lanes_succ_r(NE, FR);           // (scalar unit)
  // The window's last event with the input going on behind the window: whether a rematch follows is not known here.  A row
  // of this tile only when that event lies in an owned lane AND a match is still pending at the window's end — a match
  // that reaches 190 bytes past its tile: the host's next rung.
  if (rend > kFsmWinEnd && NE != 0ull && 63 - __builtin_clzll(NE) <= kWaveTile / kFsmChunk) {
    const uint32_t pend = pend_at_end();
    if (__builtin_amdgcn_readlane(static_cast<int>(pend), 63) != 0) fallback |= 8u;
  }
  uint32_t Er[4];
  fsm_lane_ends(T, static_cast<uint32_t>(LN >> lane) & 1u, Er);
  if (!owned || (CXG_FSM_ABL & 4)) Er[0] = Er[1] = Er[2] = Er[3] = 0u;
  const uint32_t nl = static_cast<uint32_t>(__builtin_popcount(Er[0]) + __builtin_popcount(Er[1]) + __builtin_popcount(Er[2]) + __builtin_popcount(Er[3]));
  const uint32_t incl = wave_inclusive_sum(nl);
  const uint32_t tot = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
  uint32_t idx = nrows_w + incl - nl;
  // the ends in ascending order: T's dword i is Er[3 - i] reversed, so its lowest position is the highest bit there.  No masked
  // branch in the loop (a lane without a bit writes the dump slot), one uniform branch per round.
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t xw = Er[3 - i];
    const int32_t e0 = (c0 + 16 * i + 1) | static_cast<int32_t>(tag);     // (c0 is a multiple of 64 and the sum stays below 4096: the tag's bits are free)
    while (__builtin_amdgcn_ballot_w64(xw != 0u) != 0ull) {
      const bool has = xw != 0u;
      const uint32_t q = static_cast<uint32_t>(__builtin_clz(xw | 1u));
      xw &= ~(0x80000000u >> q);
      const uint32_t slot = (has && idx < static_cast<uint32_t>(kRowsPerWave)) ? idx : static_cast<uint32_t>(kRowsPerWave);
      s_re_wave[slot] = static_cast<uint16_t>(e0 + static_cast<int32_t>(q >> 1));
      idx += has ? 1u : 0u;
    }
  }
  return tot;
}


}  // namespace

}  // namespace cxgdev
