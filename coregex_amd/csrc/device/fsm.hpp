// fsm.hpp — FindAll as ONE left-to-right finite-state transducer, and the per-lane replay that turns its
// events into match rows.  Shared by scan_fsm.hip (device) and tests/emu (host twin); plain C++.
//
// Why.  The table-walking kernels of round 1 (walk.hpp lane_bidir / lane_digit) follow the reference's loop
// literally — search from `pos`, walk until the DFA dies, jump BACK to the last accepting position, restart
// (meta/findall.go:216-239 over dfa/lazy/lazy.go:1102-1315).  A lane may only start such a loop where the
// reference's state is known, i.e. after a synchronising byte, and input without such bytes was refused.  Here the
// whole loop — including the jump back — is folded into a finite-state machine built on the host (host/fsm.cc):
//
//   state  = stack of searches  [C_0 | C_1 | ... | C_{k-1} | I]
//            C_j : ordered NFA thread list of a search whose match is PENDING — it has seen an accepting state, the
//                  threads of higher priority than the match ("conts") are still alive and may extend it;
//            I   : the innermost search, started where the newest pending match ended (or at the last commit).
//   step b : every level moves on b separately (ordered closure, dfa/lazy/builder.go:183-293).  The outermost level
//            whose new list holds Match REMATCHES: its pending end moves here and every deeper level is discarded
//            (those searches started inside a match that just grew).  A pending level whose conts died is COMMITTED
//            relative to its parents and leaves the stack.  When the innermost list holds Match a new pending level
//            is CREATED and a fresh search starts at this position.  Break-at-match (builder.go:210-213) is the
//            truncation of a list at its Match state.
//
// The machine never moves backwards, so "the state at byte p" is a well-defined function of hay[0, p): a lane can
// replay any 64-byte chunk once it knows its entry state, and entry states come from a second, tiny automaton over
// SETS of states (the "uncertainty" rows of the table): start from "any state at all" 64 bytes early and walk; on
// real text the set collapses to one state within a few bytes (any byte outside the pattern's alphabet does it,
// and so do most bytes inside).  No synchronising byte is required for correctness any more.
//
// Rows.  A match is owned by the lane in whose chunk its pending level was created.  The lane keeps walking past
// its chunk until every level that can still change its rows (its own pending levels, and levels that were already
// alive when it entered) is resolved; what happens to levels created beyond the chunk is the next lanes' business.
// Match STARTS come from the reverse DFA exactly as in the reference (SearchReverse, lazy.go:1769-1920), bounded
// below by the previous row's end.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CXG_FSM_HD __host__ __device__ __forceinline__
#else
#define CXG_FSM_HD inline
#endif

namespace cxgdev {

constexpr uint32_t kFsmMagic = 0x43584732u;   // "CXG2"
constexpr int kFsmMaxLevels = 7;              // pending searches alive at once (4-bit refs in one dword)
constexpr int kFsmLaneRows = 8;               // rows a lane buffers for its 64-byte chunk (denser input: fallback flag)
constexpr int kFsmChunk = 64;
constexpr uint32_t kFsmMaxRows = 255;         // table rows: states + uncertainty sets (+ the "wide" row), u8 ids
constexpr uint32_t kFsmMaxTableBytes = 20u * 1024u;
constexpr int kFsmMembers = 8;                // members listed per uncertainty row (more: the row counts as wide)

// Event descriptor (u16), indexed by the high byte of a table entry (0 = no event):
//   bits 0-1 kind: 0 levels died only, 1 create (innermost matched), 2 rematch (pending level j matched again)
//   bits 2-4 j (rematch), bit 5 conts: the matched level keeps live threads (stays pending)
//   bits 8-14 died: pending levels (numbered before the step) whose threads died without a match
constexpr uint32_t kFsmEvDied = 0, kFsmEvCreate = 1, kFsmEvRematch = 2;

struct FsmHeader {              // device image; offsets in bytes from the header
  uint32_t magic, n_t, n_rows, ncls;        // n_t: transducer states (rows [0, n_t)); n_rows: all table rows
  uint32_t top_row, wide_row, n_events, depth;   // top_row: "any state"; wide_row: set not tabulated (absorbing)
  uint32_t stride, cls_off, tab_off, ev_off;     // cls: u8[256]; tab: u16[n_rows][stride] = next row | event << 8
  uint32_t lev_off, mem_off, rev_off, rev_states;  // lev: u8[n_t] pending levels of a state; mem: u8[n_rows][8] members, 0xFF pad
  uint32_t rev_start, rev_first_accept, flags, total_bytes;   // rev: u8[rev_states][ncls], state 0 dead
  uint32_t lds_bytes, max_len, pad0, pad1;
};

struct FsmView {
  const uint8_t* cls;
  const uint16_t* tab;
  const uint16_t* ev;
  const uint8_t* lev;
  const uint8_t* rev;
  uint32_t stride, n_t, top_row, ncls, rev_start, rev_first_accept;
};

// Mem concept: uint32_t byte(int32_t r) for any r with 0 <= origin + r < len (r relative to the tile origin);
//              uint32_t dword(int32_t r): little-endian bytes r..r+3, r % 4 == 0, all four inside one staged 64-byte chunk.

// Pure state walk over [from, to): the warm-up that finds a chunk's entry state.  aligned: the range is whole dwords
// of staged chunks.
template <class Mem>
CXG_FSM_HD uint32_t fsm_walk(const FsmView& v, const Mem& m, uint32_t row, int32_t from, int32_t to, bool aligned) {
  int32_t i = from;
  if (aligned) {
    for (; i + 4 <= to; i += 4) {
      const uint32_t d = m.dword(i);
      row = v.tab[row * v.stride + v.cls[d & 0xFFu]] & 0xFFu;
      row = v.tab[row * v.stride + v.cls[(d >> 8) & 0xFFu]] & 0xFFu;
      row = v.tab[row * v.stride + v.cls[(d >> 16) & 0xFFu]] & 0xFFu;
      row = v.tab[row * v.stride + v.cls[d >> 24]] & 0xFFu;
    }
  }
  for (; i < to; i++) row = v.tab[row * v.stride + v.cls[m.byte(i)]] & 0xFFu;
  return row;
}

struct FsmLane {
  uint32_t x = 0;        // transducer state
  uint32_t nlev = 0;     // live pending levels
  uint32_t lev = 0;      // 4 bits per level, outermost first: 0 created beyond the chunk, 1 alive at entry (foreign), 2 + r own row r
  uint32_t nrows = 0;    // own rows so far
  uint32_t flags = 0;    // 1: more than kFsmLaneRows rows, 2: level stack overflow, 4: walk budget exhausted
};

// Rows concept: void set_end(uint32_t r, int32_t e).
template <class Rows>
CXG_FSM_HD void fsm_apply(FsmLane& L, uint32_t ev, int32_t e, bool in_chunk, Rows& rows) {
  const uint32_t kind = ev & 3u, j = (ev >> 2) & 7u, conts = (ev >> 5) & 1u, died = ev >> 8;
  const uint32_t keep_n = kind == kFsmEvRematch ? j : L.nlev;   // levels that survive unless they died
  uint32_t nl = 0, nn = 0;
  for (uint32_t q = 0; q < keep_n; q++) {
    if ((died >> q) & 1u) continue;
    nl |= ((L.lev >> (4u * q)) & 15u) << (4u * nn);
    nn++;
  }
  uint32_t ref = 0xFFu;                                         // level to push, if any
  if (kind == kFsmEvRematch) {
    ref = (L.lev >> (4u * j)) & 15u;
    if (ref >= 2u) { L.nrows = ref - 1u; rows.set_end(ref - 2u, e); }   // its row keeps its place, later rows are gone
    else if (ref == 1u) L.nrows = 0;                            // a level older than the chunk grew: every own row was inside it
    if (!conts) ref = 0xFFu;
  } else if (kind == kFsmEvCreate) {
    if (in_chunk) {
      if (L.nrows >= static_cast<uint32_t>(kFsmLaneRows)) { L.flags |= 1u; ref = 0u; }
      else { rows.set_end(L.nrows, e); ref = 2u + L.nrows; L.nrows++; }
    } else ref = 0u;
    if (!conts) ref = 0xFFu;
  }
  if (ref != 0xFFu) {
    if (nn >= static_cast<uint32_t>(kFsmMaxLevels)) L.flags |= 2u;
    else { nl |= ref << (4u * nn); nn++; }
  }
  L.lev = nl;
  L.nlev = nn;
}

// Replays the chunk [c0, c1) from entry state `entry` (pending levels of the entry state are foreign), then keeps
// walking until no level that can still change an own row is alive.  rend: end of input, budget: last position the
// lane may read (serial-walk budget, scan_dfa.h).  Returns the number of own rows (ends in `rows`).
template <class Mem, class Rows>
CXG_FSM_HD void fsm_replay(const FsmView& v, const Mem& m, uint32_t entry, int32_t c0, int32_t c1, int32_t rend, int32_t budget,
                           FsmLane& L, Rows& rows) {
  L.x = entry;
  L.nlev = v.lev[entry];
  L.lev = 0x1111111u & ((1u << (4u * L.nlev)) - 1u);
  L.nrows = 0;
  L.flags = 0;
  int32_t i = c0;
  if (c1 <= rend && c1 <= budget) {                  // the whole chunk is staged data: dword reads, no end tests
    for (; i < c1; i += 4) {
      const uint32_t d = m.dword(i);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int k = 0; k < 4; k++) {
        const uint32_t t = v.tab[L.x * v.stride + v.cls[(d >> (8 * k)) & 0xFFu]];
        L.x = t & 0xFFu;
        if (t >> 8) fsm_apply(L, v.ev[t >> 8], i + k + 1, true, rows);
      }
    }
  }
  for (;; i++) {
    if (i >= c1 && L.lev == 0u) break;               // only levels created beyond the chunk are left
    if (i >= rend) break;                            // end of input: every pending match is committed as it stands
    if (i >= budget) { L.flags |= 4u; break; }
    const uint32_t t = v.tab[L.x * v.stride + v.cls[m.byte(i)]];
    L.x = t & 0xFFu;
    if (t >> 8) fsm_apply(L, v.ev[t >> 8], i + 1, i < c1, rows);
  }
}

// Start of the match that ends at e: the smallest p >= bound with hay[p, e) in the language — the anchored reverse
// DFA without break-at-match (meta/compile.go:193-194), walked from e - 1 downwards (lazy.go:1769-1920).  lowest:
// first position that exists.  Returns kFsmNoStart when the reverse DFA never accepts (cannot happen for a real match);
// positions are relative to the tile origin and may be negative.
constexpr int32_t kFsmNoStart = -0x7FFFFFFF - 1;
template <class Mem>
CXG_FSM_HD int32_t fsm_match_start(const FsmView& v, const Mem& m, int32_t e, int32_t bound, int32_t budget_lo, uint32_t& over) {
  uint32_t s = v.rev_start;
  int32_t st = kFsmNoStart;
  for (int32_t at = e - 1; at >= bound; at--) {
    if (at < budget_lo) { over = 1u; break; }
    s = v.rev[s * v.ncls + v.cls[m.byte(at)]];
    if (s == 0u) break;
    if (s >= v.rev_first_accept) st = at;
  }
  return st;
}

}  // namespace cxgdev
