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

// The lockstep walks interleave N independent chains of dependent LDS reads per lane.  In the look-around instantiations
// (two lookups per class) the machine scheduler clusters each chain's steps and serialises the waits; a scheduling barrier
// after every round of steps keeps round k of all chains together there (+5..7 % on `\\berror\\b`, `(?m)^\\d+`).  Without
// look-around the scheduler's own order is 3-4 % faster than the forced one (measured both ways, scripts/gpu_r2n.sh).
#if defined(__HIP_DEVICE_COMPILE__)
#define CXG_FSM_ROUND_END(look) do { if (look) __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CXG_FSM_ROUND_END(look) do { } while (0)
#endif

namespace cxgdev {

constexpr uint32_t kFsmMagic = 0x43584736u;   // "CXG6"
constexpr int kFsmMaxLevels = 7;              // pending searches alive at once (4-bit refs in one dword)
constexpr int kFsmLaneRows = 4;               // rows buffered per 32-byte chunk (denser input: fallback flag) ...
constexpr int kFsmLaneEvents = 8;             // events recorded inside a 32-byte chunk before they are applied (power of two) ...
constexpr int kFsmLaneRowsMax = 16, kFsmLaneEventsMax = 32;   // ... in the kernel's mode for very dense matches (one per 2 bytes)
constexpr int kFsmChunk = 64;                 // bytes per lane (two sub-chunks of kFsmSub bytes, walked in lockstep)
constexpr uint32_t kFsmMaxTableBytes = 24u * 1024u;
constexpr int kFsmMembers = 8;                // members listed per uncertainty row (more: the row counts as wide)

// Event descriptor (u16):
//   bits 0-1 kind: 0 levels died only, 1 create (innermost matched), 2 rematch (pending level j matched again)
//   bits 2-4 j (rematch), bit 5 conts: the matched level keeps live threads (stays pending)
//   bits 8-14 died: pending levels (numbered before the step) whose threads died without a match
constexpr uint32_t kFsmEvDied = 0, kFsmEvCreate = 1, kFsmEvRematch = 2;

// Table.  A row is `stride` u16 (a power of two): [0, ncls) the transitions, [ncls] the event descriptor of an ALIAS
// row, [ncls + 1] the number of pending levels of the row's state, [ncls + 2] the row of the state itself (an alias
// row names the state it copies).  A transition holds the BYTE OFFSET of the target
// row (a multiple of the row size >= 16) plus two flag bits: bit 0 the step CREATES a match, bit 1 it REMATCHES one.
// The dependent chain of a walk is one LDS read and one v_and_or:  x = tab[(x & ~3) | 2 * class].  Row order:
//   [0, n_t)           the transducer states;
//   [n_t, n_t + n_a)   alias rows: a transition that carries an event targets a copy of its target state's row with
//                      the event in column ncls — "an event happened" is `x >= alias_lo` (ordered by event kind);
//   then n_u uncertainty rows (sets of states; the first is "any state"), entered only by warm-up walks; their
//   transitions lead to other sets or, once the set has collapsed, to the state's own row;
//   last the wide row (a set that was not tabulated; absorbing).
struct FsmHeader {              // device image; offsets in bytes from the header
  uint32_t magic, n_t, n_a, n_u;
  uint32_t ncls, stride, row_bytes, depth;
  uint32_t alias_lo, u_lo, top_off, wide_off;      // byte offsets of the first alias row, the first set row, "any state", wide
  uint32_t cls_off, tab_off, mem_off, rev_off;     // cls: u8[256] = 2 * class; tab: u16[rows][stride]; mem: u16[n_u + 1][8] rows of a set's members, 0xFFFF pad / not listed
  uint32_t rev_states, rev_start_off, rev_accept_off, rev_row_bytes;   // rev: u16[rev_states][rev_row_bytes / 2] (a power of two); an entry = the target row's offset FROM THE END OF THE HEADER
                                                   // (its LDS address in the kernel; rev_start_off / rev_accept_off count the same way) | 1 when the target accepts; row 0 = dead, leads to itself
  uint32_t total_bytes, lds_bytes, max_len, nk;    // nk: kinds of the byte BEHIND a step that the step depends on (1: none; 2 or 3, see "Look-around")
  uint32_t create_lo, rematch_lo, row_shift, knd_off; // row_bytes == 1 << row_shift; alias rows are ordered by event kind: [alias_lo, create_lo) levels died only,
                                                   // [create_lo, rematch_lo) create, [rematch_lo, u_lo) rematch.  nk > 1: knd = u8[256] 2 * kind of a byte, then
                                                   // u16[nk] start rows by the kind of the haystack's first byte, then u16[nk * nk] reverse start rows by
                                                   // nk * kind(hay[e-1]) + kind(hay[e]) for a match that ends at e
  uint32_t outside_byte;                           // what the positions in front of and behind the haystack read as ('\n' with line anchors, else 0)
  uint32_t rev_text_col;                           // != 0: the pattern holds a text-start anchor (\A, ^): byte offset of the reverse rows' extra column,
                                                   // columns by the kind of the text's first byte, 1 = "accepting if this position is the start of the text"
  // Round 6 — byte-indexed tables of the kernel's DIRECT mode (shallow machines without look-around whose rows fit): see "Direct mode"
  uint32_t direct_off, direct_bytes;               // section offset from the header (0: none) and its size: d_slots rows of 256 bytes, then u8[256] per slot: 0x80 a set, else pending levels
  uint32_t d_slots;                                // rows ("slots"); an entry is the slot of the next row
  uint32_t d_racc_lo, d_rstart;                    // reverse automaton in the same slot space: accepting slots >= d_racc_lo, start slot; dead: d_rdead
  uint32_t d_top;                                  // slot of the set "any state"
  uint32_t end_col;                                // != 0: the pattern holds an end-of-text anchor (\z, $ without (?m)): 2 * the kind "end of the text" — a kind NO byte
                                                   // has; the step over the haystack's last byte takes column class + end_col (fsm.hpp "End of text")
  uint32_t d_rdead;                                // slot of the dead reverse state (its row leads to itself)
  uint32_t d_pad[2];
};
constexpr uint32_t kFsmdMaxBytes = 12288u;         // largest direct section the kernel has an instantiation for: beyond it the rows cost a resident workgroup per CU and their
                                                   // reads collide in the LDS banks (README IPv4 pattern, 20 KiB: 0.71 ms against 0.62 class-indexed — profiles/r06_c2_*)

// Look-around (word boundaries `\b` `\B`, multi-line anchors `(?m)^` `(?m)$`; nfa.Look, nfa/nfa.go:92-117).  An assertion
// at position p reads the bytes on both sides of p.  The machine stays a plain left-to-right transducer with IMMEDIATE
// acceptance when the step over byte i also sees the KIND of byte i + 1 — word / other, newline / other, or all three,
// whatever the pattern's assertions distinguish; the positions outside the haystack read as "not a word byte" and, for
// the line anchors, as a newline (nfa/pikevm.go:1646-1674: pos == 0 / pos == len) — : the input symbol of a step is the
// pair (class of hay[i], kind of hay[i+1]), column  nk * class + kind.  After the step both sides of position i + 1 are known, so every
// assertion there is decided on the host when the table is built (host/fsm.cc) and a match that ends at i + 1 is
// reported by the step over byte i as always.  The reverse DFA mirrors it: the step over byte i (walking down) sees
// the kind of hay[i-1].  Only the class lookup changes — it is part of the Mem concept below, so programs without
// assertions (nk == 1) compile to the same instructions as before.
//
// End of text (`\z`, `$` without (?m); nfa.LookEndText, nfa/pikevm.go:1651: pos == len) inside an unanchored pattern (`a$|z`).  The
// position behind the haystack is no byte: it gets a kind of its own (the LAST of the nk kinds, FsmHeader::end_col = 2 * kind), which
// is a line edge and not a word byte like the other outside position, and the only kind at which the anchor holds.  Exactly one
// forward step sees it — the step over the haystack's last byte — and one reverse start row (a match that ends at len); a thread
// that passed the anchor can consume nothing, so everything else runs as if the anchor never held.  LOOK == 2 instantiations
// compare a step's position with the last byte's (Mem::last()) and swap the kind in: two VALU per byte, for these programs only.
CXG_FSM_HD uint32_t fsm_u16(const uint8_t* p, uint32_t byte_off) {
  uint32_t r = *reinterpret_cast<const uint16_t*>(p + byte_off);
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(r));   // opaque: knowing that the upper half is zero the compiler narrows the next step's  (entry & ~3) | class  to 16-bit operations
#endif                 // and widens the result again — two or three instructions where v_and_or_b32 is one (round 6, ISA of the walks)
  return r;
}
struct FsmView {
  const uint8_t* cls2;      // 2 * nk * class of a byte: byte offset of the class's first column
  const uint8_t* knd;       // 2 * kind of a byte (nk > 1); behind it the start-row tables (FsmHeader::knd_off)
  const uint16_t* lk16;     // look-around (round 6): cls2[b] | knd[b] << 8 — ONE lookup per byte serves the step over it (its class) and the step in front
                            // of it (its kind); built where the view is (the kernels: 256 threads from the staged image; the twin: a host array)
  uint32_t nk;
  const uint8_t* tab;       // rows, addressed by byte offset
  const uint8_t* rev;
  uint32_t ncls2;           // 2 * ncls: byte offset of the event column inside a row
  uint32_t alias_lo, u_lo, top_off, rev_start_off, rev_accept_off;
  uint32_t rev_dead;        // the dead reverse state's entry (= rev_off - header size)
  uint32_t rev_text_col;    // FsmHeader::rev_text_col
  uint32_t end_col;         // FsmHeader::end_col
  uint32_t create_lo, rematch_lo;
  const uint8_t* mem;       // members of the set rows
  uint32_t row_shift;
};
// Class lookups of a Mem type, from its byte() / dword(): CRTP base shared by the kernel's LDS window and the twin's
// host memory.  LOOK: 0 = no assertions (nk == 1), 1 = the image has nk > 1, 2 = ... and an end-of-text kind (the Mem type then
// has int32_t last(): position of the haystack's last byte, relative like every position; out of reach: any value no step takes).
template <class Derived, int LOOK>
struct FsmClassify {
  CXG_FSM_HD const Derived& self() const { return *static_cast<const Derived*>(this); }
  // column offset of the forward step over byte i
  CXG_FSM_HD uint32_t cls(const FsmView& v, int32_t i) const {
    if (!LOOK) return v.cls2[self().byte(i)];
    uint32_t kn = static_cast<uint32_t>(v.lk16[self().byte(i + 1)]) >> 8;
    if (LOOK == 2) kn = i == self().last() ? v.end_col : kn;
    return (v.lk16[self().byte(i)] & 0xFFu) + kn;
  }
  // ... of the four steps over the aligned dword at r
  CXG_FSM_HD void classes4(const FsmView& v, int32_t r, uint32_t (&k)[4]) const {
    const uint32_t d = self().dword(r);
    if (!LOOK) { k[0] = v.cls2[d & 0xFFu]; k[1] = v.cls2[(d >> 8) & 0xFFu]; k[2] = v.cls2[(d >> 16) & 0xFFu]; k[3] = v.cls2[d >> 24]; return; }
    // five lookups for four steps (class in the low byte, kind in the high one) instead of eight into two tables: 3.5 instructions
    // per byte where there were five (round 6)
    const uint32_t t[5] = {v.lk16[d & 0xFFu], v.lk16[(d >> 8) & 0xFFu], v.lk16[(d >> 16) & 0xFFu], v.lk16[d >> 24], v.lk16[self().byte(r + 4)]};
    uint32_t kn[4] = {t[1] >> 8, t[2] >> 8, t[3] >> 8, t[4] >> 8};
    if (LOOK == 2) {
      const int32_t dl = self().last() - r;
      kn[0] = dl == 0 ? v.end_col : kn[0]; kn[1] = dl == 1 ? v.end_col : kn[1]; kn[2] = dl == 2 ? v.end_col : kn[2]; kn[3] = dl == 3 ? v.end_col : kn[3];
    }
    k[0] = (t[0] & 0xFFu) + kn[0]; k[1] = (t[1] & 0xFFu) + kn[1]; k[2] = (t[2] & 0xFFu) + kn[2]; k[3] = (t[3] & 0xFFu) + kn[3];
  }
  // column offset of the reverse step over byte i (walking down)
  CXG_FSM_HD uint32_t rcls(const FsmView& v, int32_t i) const {
    if (!LOOK) return v.cls2[self().byte(i)];
    return (v.lk16[self().byte(i)] & 0xFFu) + (static_cast<uint32_t>(v.lk16[self().byte(i - 1)]) >> 8);
  }
  // reverse start row for a match that ends at e
  CXG_FSM_HD uint32_t rstart(const FsmView& v, int32_t e) const {
    if (!LOOK) return v.rev_start_off;
    uint32_t right = v.knd[self().byte(e)];
    if (LOOK == 2) right = e == self().last() + 1 ? v.end_col : right;
    const uint32_t idx = (v.knd[self().byte(e - 1)] >> 1) * v.nk + (right >> 1);
    return fsm_u16(v.knd, 256u + 2u * v.nk + 2u * idx);
  }
  // start row of the search at the haystack's first byte (tile-relative position 0 of the first tile)
  CXG_FSM_HD uint32_t origin(const FsmView& v) const {
    if (!LOOK) return 0u;
    return fsm_u16(v.knd, 256u + v.knd[self().byte(0)]);
  }
  static constexpr bool kLook = LOOK != 0;
};
// the state's own row (an alias row is a copy of it)
CXG_FSM_HD uint32_t fsm_canon(const FsmView& v, uint32_t x) { return fsm_u16(v.tab, x + v.ncls2 + 4u); }
// member j (0..7) of the set row u, as a row offset; 0xFFFF: none / the set is not listed
CXG_FSM_HD uint32_t fsm_member(const FsmView& v, uint32_t u, uint32_t j) { return fsm_u16(v.mem, (((u - v.u_lo) >> v.row_shift) * 8u + j) * 2u); }
// one step: entry t (row offset | flags) and 2 * class -> next entry
CXG_FSM_HD uint32_t fsm_next(const FsmView& v, uint32_t t, uint32_t cls2) { return fsm_u16(v.tab, (t & 0xFFFCu) | cls2); }   // (an entry is 16 bits wide: with ~3 the compiler zero-extends it by a second instruction in some walks)
CXG_FSM_HD uint32_t fsm_shift_in2(uint32_t mask, uint32_t t) {   // (mask >> 2) | (t << 30): v_alignbit_b32
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(t, mask, 2u);
#else
  return (mask >> 2) | (t << 30);
#endif
}

// Mem concept: uint32_t byte(int32_t r) for any r with 0 <= origin + r < len (r relative to the tile origin) — with
//              look-around also one position outside on either side, which reads as 0;
//              uint32_t dword(int32_t r): little-endian bytes r..r+3, r % 4 == 0, all four inside one staged 64-byte chunk;
//              the class lookups of FsmClassify.

// Pure state walk over [from, to): the warm-up that finds a chunk's entry state.  aligned: the range is whole dwords
// of staged chunks.  Rows are byte offsets.
template <class Mem>
CXG_FSM_HD uint32_t fsm_walk(const FsmView& v, const Mem& m, uint32_t x, int32_t from, int32_t to, bool aligned) {
  int32_t i = from;
  if (aligned) {
    for (; i + 4 <= to; i += 4) {
      uint32_t k[4];
      m.classes4(v, i, k);
      x = fsm_next(v, x, k[0]);
      x = fsm_next(v, x, k[1]);
      x = fsm_next(v, x, k[2]);
      x = fsm_next(v, x, k[3]);
    }
  }
  for (; i < to; i++) x = fsm_next(v, x, m.cls(v, i));
  return x & ~3u;
}

// N warm-up walks of `nbytes` (multiple of 4) staged bytes each, in lockstep, all from row x0.
template <int N, class Mem>
CXG_FSM_HD void fsm_walk_n(const FsmView& v, const Mem& m, uint32_t x0, const int32_t (&from)[N], int32_t nbytes, uint32_t (&x)[N]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int a = 0; a < N; a++) x[a] = x0;
  for (int32_t i = 0; i < nbytes; i += 4) {
    uint32_t k[N][4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int a = 0; a < N; a++) m.classes4(v, from[a] + i, k[a]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < 4; q++) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int a = 0; a < N; a++) x[a] = fsm_next(v, x[a], k[a][q]);
      CXG_FSM_ROUND_END(Mem::kLook);
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int a = 0; a < N; a++) x[a] &= ~3u;
}

struct FsmLane {
  uint32_t x = 0;        // current row (byte offset)
  uint32_t nlev = 0;     // live pending levels
  uint64_t lev = 0;      // 8 bits per level, outermost first: 0 created beyond the chunk, 1 alive at entry (foreign), 2 + r own row r
  uint32_t nrows = 0;    // own rows so far
  uint32_t flags = 0;    // 1: more than kFsmLaneRows rows, 2: level stack overflow, 4: walk budget exhausted, 8: more than kFsmLaneEvents events
  uint32_t xc1 = 0;      // row at the chunk's end (or at the end of input inside it): the next chunk's true entry state
  uint32_t max_rows = kFsmLaneRows, max_events = kFsmLaneEvents;   // buffer sizes of the caller (set before a replay)
};

// Rows concept: void set_end(uint32_t r, int32_t e).
template <class Rows>
CXG_FSM_HD void fsm_apply(FsmLane& L, uint32_t ev, int32_t e, bool in_chunk, Rows& rows) {
  const uint32_t kind = ev & 3u, j = (ev >> 2) & 7u, conts = (ev >> 5) & 1u, died = ev >> 8;
  const uint32_t keep_n = kind == kFsmEvRematch ? j : L.nlev;   // levels that survive unless they died
  uint64_t nl = 0;
  uint32_t nn = 0;
  for (uint32_t q = 0; q < keep_n; q++) {
    if ((died >> q) & 1u) continue;
    nl |= ((L.lev >> (8u * q)) & 255ull) << (8u * nn);
    nn++;
  }
  uint32_t ref = 0xFFFFu;                                       // level to push, if any
  if (kind == kFsmEvRematch) {
    ref = static_cast<uint32_t>((L.lev >> (8u * j)) & 255ull);
    if (ref >= 2u) { L.nrows = ref - 1u; rows.set_end(ref - 2u, e); }   // its row keeps its place, later rows are gone
    else if (ref == 1u) L.nrows = 0;                            // a level older than the chunk grew: every own row was inside it
    if (!conts) ref = 0xFFFFu;
  } else if (kind == kFsmEvCreate) {
    if (in_chunk) {
      if (L.nrows >= L.max_rows) { L.flags |= 1u; ref = 0u; }
      else { rows.set_end(L.nrows, e); ref = 2u + L.nrows; L.nrows++; }
    } else ref = 0u;
    if (!conts) ref = 0xFFFFu;
  }
  if (ref != 0xFFFFu) {
    if (nn >= static_cast<uint32_t>(kFsmMaxLevels)) L.flags |= 2u;
    else { nl |= static_cast<uint64_t>(ref) << (8u * nn); nn++; }
  }
  L.lev = nl;
  L.nlev = nn;
}

// ---- replay of one chunk = fast part (whole staged chunk: walk + record events) + finish (apply the recorded events,
// then walk on, applying events as they come, until no level that can still change an own row is alive).
// Events concept: void push(uint32_t k, uint32_t row); uint32_t row_at(uint32_t k).
struct FsmTrace { uint32_t x, evbits, nev, cap; };   // state after the chunk, event positions (bit per byte), events seen, event slots (power of two)

template <class Events>
CXG_FSM_HD void fsm_step_rec(const FsmView& v, uint32_t cls2, uint32_t bit, FsmTrace& t, Events& evs) {
  t.x = fsm_next(v, t.x, cls2) & ~3u;
  if (t.x >= v.alias_lo) {                           // rare per lane: recorded now, applied after the walk
    evs.push(t.nev & (t.cap - 1u), t.x);
    t.nev++;
    t.evbits |= bit;
  }
}
// N chunks of kFsmSub bytes walked in lockstep (N independent dependency chains per lane: the table reads of one
// hide behind those of the others).  Class lookups do not depend on the state: those of the NEXT dword are issued
// before this dword's chain, so a chain step is one LDS read + one add.
constexpr int kFsmSub = 32;
constexpr int32_t kFsmNoStart = -0x7FFFFFFF - 1;   // fsm_match_start: the reverse automaton never accepted
template <int N, class Mem, class Events>
CXG_FSM_HD void fsm_fast(const FsmView& v, const Mem& m, const int32_t (&c0)[N], FsmTrace (&t)[N], Events* evs) {
  uint32_t kk[N][4], nn[N][4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int a = 0; a < N; a++) m.classes4(v, c0[a], kk[a]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int q = 0; q < kFsmSub / 4; q++) {
    if (q + 1 < kFsmSub / 4) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int a = 0; a < N; a++) m.classes4(v, c0[a] + 4 * (q + 1), nn[a]);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; k++) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int a = 0; a < N; a++) fsm_step_rec(v, kk[a][k], 1u << (4 * q + k), t[a], evs[a]);
      CXG_FSM_ROUND_END(Mem::kLook);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int a = 0; a < N; a++) for (int k = 0; k < 4; k++) kk[a][k] = nn[a][k];
  }
}

// SHALLOW machines (FsmHeader::depth <= 1: never more than one pending match — `\d+\.\d+`, alternations of literals,
// most log patterns): a chunk's rows follow from two bitmaps, no per-event bookkeeping.  Every create event opens a row;
// the row's end is the last create / rematch event before the next create (a rematch before the first create belongs
// to a match that was pending when the chunk was entered: another lane's row).
struct FsmTraceS { uint32_t x, k0, k1; };             // entry after the chunk; the flag bits of its 32 steps, two per byte
template <int N, class Mem>
CXG_FSM_HD void fsm_fast_shallow(const FsmView& v, const Mem& m, const int32_t (&c0)[N], FsmTraceS (&t)[N]) {
  uint32_t kk[N][4], nn[N][4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int a = 0; a < N; a++) m.classes4(v, c0[a], kk[a]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int q = 0; q < kFsmSub / 4; q++) {
    if (q + 1 < kFsmSub / 4) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int a = 0; a < N; a++) m.classes4(v, c0[a] + 4 * (q + 1), nn[a]);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; k++) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int a = 0; a < N; a++) {                    // the new entry's two flag bits shift into the mask from the top:
        t[a].x = fsm_next(v, t[a].x, kk[a][k]);        // three VALU per byte, no compare, no branch
        if (q < 4) t[a].k0 = fsm_shift_in2(t[a].k0, t[a].x); else t[a].k1 = fsm_shift_in2(t[a].k1, t[a].x);
      }
      CXG_FSM_ROUND_END(Mem::kLook);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int a = 0; a < N; a++) for (int k = 0; k < 4; k++) kk[a][k] = nn[a][k];
  }
}
// Rows of a whole chunk [c0, c0 + kFsmSub) from its trace, then the walk past the chunk while its last match is pending.
template <class Mem, class Rows>
CXG_FSM_HD void fsm_finish_shallow(const FsmView& v, const Mem& m, const FsmTraceS& t, int32_t c0, int32_t rend, int32_t budget,
                                   FsmLane& L, Rows& rows) {
  const int32_t c1 = c0 + kFsmSub;
  L.flags = 0;
  // two bits per byte: bit 2p = the step over byte p created a match, bit 2p + 1 = it rematched one
  const uint64_t K = (static_cast<uint64_t>(t.k1) << 32) | t.k0;
  uint64_t C = K & 0x5555555555555555ull;
  const uint64_t CR = C | ((K >> 1) & 0x5555555555555555ull);
  uint32_t n = 0;
  while (C) {
    const uint32_t c = static_cast<uint32_t>(__builtin_ctzll(C));
    C &= C - 1;
    const uint64_t below_next = C ? ((1ull << static_cast<uint32_t>(__builtin_ctzll(C))) - 1ull) : ~0ull;
    const uint64_t win = CR & below_next & ~((1ull << c) - 1ull);            // events of this row (bit c is set)
    const int32_t top = (63 - static_cast<int32_t>(__builtin_clzll(win))) >> 1;
    if (n < L.max_rows) rows.set_end(n, c0 + top + 1); else L.flags |= 1u;
    n++;
  }
  L.nrows = n < L.max_rows ? n : L.max_rows;
  L.x = t.x & ~3u;
  L.xc1 = L.x;
  L.nlev = fsm_u16(v.tab, L.x + v.ncls2 + 2u);                               // 0 or 1
  // a match pending at the chunk's end is the chunk's last row when the chunk created one, else it is older than the chunk
  L.lev = L.nlev ? (L.nrows ? 1u + L.nrows : 1u) : 0u;
  for (int32_t i = c1;; i++) {
    if (L.lev == 0u) break;
    if (i >= rend) break;
    if (i >= budget) { L.flags |= 4u; break; }
    L.x = fsm_next(v, L.x, m.cls(v, i)) & ~3u;
    if (L.x >= v.alias_lo) fsm_apply(L, fsm_u16(v.tab, L.x + v.ncls2), i + 1, false, rows);
  }
}

// ---- Direct mode (round 6).  The class-indexed walk costs five instructions per byte: extract the byte, look its class up, v_and_or,
// table read, flag shift.  With rows indexed by the BYTE (256 one-byte entries, an entry = the slot of the next row) a step is
//   v_perm_b32  addr = slot << 8 | byte k of the data dword      ds_read_u8  slot = table[addr]      v_alignbit  flags
// — no class lookup, one LDS read instead of two.  The two event flags live in the low bits of the slot number:
//   slot 4k      state k                         slot 4k + 1   state k entered by a step that CREATED a match   (a copy of row 4k)
//   slot 4k + 2  ... that REMATCHED one          any free slot a set of possible states (warm-up rows; d_top = "any state"), the wide row
// so the warm-up walk and the walk of the chunk are one walk over one table; a 256-byte property table behind the rows says which
// slots are sets (a chunk entered through one is unresolved) and how many levels a state has pending.  Rows of the reverse
// automaton sit in free slots too (accepting ones above the others).  Only for machines that fit: <= 64 states, <= 256 slots,
// kFsmdMaxBytes.  Built by host/fsm.cc behind the minimisation.
CXG_FSM_HD uint32_t fsmd_addr(uint32_t slot, uint32_t dword, int k) {       // slot << 8 | byte k of dword
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(slot, dword, 0x0C0C0400u | static_cast<uint32_t>(k));
#else
  return ((slot & 0xFFu) << 8) | ((dword >> (8 * k)) & 0xFFu);
#endif
}
// Mem concept as below (dword()); Tab: uint32_t at(uint32_t addr) — the kernel's LDS image at address 0, the twin's byte array.
// N walks of nbytes (a multiple of 4) staged bytes each in lockstep, x in / out: the warm-up in front of a chunk.
template <int N, class Mem, class Tab>
CXG_FSM_HD void fsmd_walk_n(const Mem& m, const Tab& tab, const int32_t (&from)[N], int32_t nbytes, uint32_t (&x)[N]) {
  for (int32_t i = 0; i < nbytes; i += 4) {
    uint32_t d[N];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int a = 0; a < N; a++) d[a] = m.dword(from[a] + i);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; k++) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int a = 0; a < N; a++) x[a] = tab.at(fsmd_addr(x[a], d[a], k));
    }
  }
}
// N chunks of kFsmSub bytes in lockstep from the slots t[a].x; the two flag bits of every step are shifted into k0 / k1 as in
// fsm_fast_shallow (a slot's low bits ARE its flags).
template <int N, class Mem, class Tab>
CXG_FSM_HD void fsmd_chunk(const Mem& m, const Tab& tab, const int32_t (&c0)[N], FsmTraceS (&t)[N]) {
  uint32_t d[N], dn[N];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int a = 0; a < N; a++) { d[a] = m.dword(c0[a]); dn[a] = 0u; t[a].k0 = t[a].k1 = 0u; }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int q = 0; q < kFsmSub / 4; q++) {
    if (q + 1 < kFsmSub / 4) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int a = 0; a < N; a++) dn[a] = m.dword(c0[a] + 4 * (q + 1));
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; k++) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int a = 0; a < N; a++) {
        t[a].x = tab.at(fsmd_addr(t[a].x, d[a], k));
        if (q < 4) t[a].k0 = fsm_shift_in2(t[a].k0, t[a].x); else t[a].k1 = fsm_shift_in2(t[a].k1, t[a].x);
      }
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int a = 0; a < N; a++) d[a] = dn[a];
  }
}
// fsm_match_start over the byte-indexed reverse rows: smallest p >= bound with hay[p, e) in the language, kFsmNoStart when the
// automaton never accepts; over: still alive at budget_lo with the haystack going on in front of it.  R: the reverse automaton's
// start, first accepting and dead slots (FsmHeader::d_rstart, d_racc_lo, d_rdead).
struct FsmdRev { uint32_t start, acc_lo, dead; };
template <class Mem, class Tab>
CXG_FSM_HD int32_t fsmd_match_start_from(const Mem& m, const Tab& tab, const FsmdRev& R, uint32_t s, int32_t st, int32_t at, int32_t bound, int32_t budget_lo, uint32_t& over) {
  const int32_t low = bound > budget_lo ? bound : budget_lo;
  uint32_t b[4], bn[4];                                  // four bytes fetched together (clamped to the window), the next four before this group's chain
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < 4; k++) { const int32_t p = at - k; b[k] = m.byte(p > budget_lo ? p : budget_lo); }
  while (at >= low) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; k++) { const int32_t p = at - 4 - k; bn[k] = m.byte(p > budget_lo ? p : budget_lo); }
    bool dead = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; k++) {
      if (dead || at - k < low) break;
      s = tab.at(fsmd_addr(s, b[k], 0));
      if (s == R.dead) { dead = true; break; }
      if (s >= R.acc_lo) st = at - k;
    }
    if (dead) return st;
    at = at - 4 >= low - 1 ? at - 4 : low - 1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; k++) b[k] = bn[k];
  }
  if (at >= bound) over = 1u;
  return st;
}
template <class Mem, class Tab>
CXG_FSM_HD int32_t fsmd_match_start(const Mem& m, const Tab& tab, const FsmdRev& R, int32_t e, int32_t bound, int32_t budget_lo, uint32_t& over) {
  return fsmd_match_start_from(m, tab, R, R.start, kFsmNoStart, e - 1, bound, budget_lo, over);
}
// ... and its first 16 steps without a branch (fsm_match_start16 below, where the reasons are): v_perm_b32 + ds_read_u8 per step, the
// accept test as a compare whose carry is added into the flag word.
template <int N, class Mem, class Tab>
CXG_FSM_HD int32_t fsmd_match_startN(const Mem& m, const Tab& tab, const FsmdRev& R, int32_t e, int32_t bound, int32_t budget_lo, uint32_t& over) {
  uint32_t W[N / 4 + 1];
  m.template below<N>(e, W);
  uint32_t s = R.start, acc = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < N; k++) {                           // step k is over byte e - 1 - k = byte N - k of W
    const int i = N - k;
    s = tab.at(fsmd_addr(s, W[i >> 2], i & 3));
    acc = acc + acc + (s >= R.acc_lo ? 1u : 0u);          // bit N - 1 - k
  }
  const int32_t low = bound > budget_lo ? bound : budget_lo;
  const uint32_t room = static_cast<uint32_t>(e - low);
  if (room <= static_cast<uint32_t>(N) && low != bound) return fsmd_match_start(m, tab, R, e, bound, budget_lo, over);
  uint32_t f = acc;
  if (room < static_cast<uint32_t>(N)) f &= ~((1u << (static_cast<uint32_t>(N) - room)) - 1u);
  int32_t st = f ? e - N + static_cast<int32_t>(__builtin_ctz(f)) : kFsmNoStart;   // lowest bit = largest k: k = N - 1 - ctz, start e - 1 - k
  if (room > static_cast<uint32_t>(N) && s != R.dead) st = fsmd_match_start_from(m, tab, R, s, st, e - (N + 1), bound, budget_lo, over);
  return st;
}
template <class Mem, class Tab>
CXG_FSM_HD int32_t fsmd_match_start16(const Mem& m, const Tab& tab, const FsmdRev& R, int32_t e, int32_t bound, int32_t budget_lo, uint32_t& over) { return fsmd_match_startN<16>(m, tab, R, e, bound, budget_lo, over); }

// ---- Round 6: rows of a SHALLOW machine from the event bits alone (no walk past the chunk, no per-lane row buffers).
// The steps of a tile are one stream of events, two bits per byte (bit 2p the step over byte p created a match, bit
// 2p + 1 it rematched one).  With at most one pending match a rematch always extends the match of the event in front of
// it, and a create always finds that match committed.  So:   an event is a ROW END  <=>  the event behind it is not a
// rematch   (no event behind it: the match stands as it is — end of input, or its threads die).  A row belongs to the
// chunk its end lies in, whoever created it; the events behind a tile come from lanes that walk the window's tail for
// nothing but their bits.  "The event behind me is a rematch" is a predecessor search: in the bit-reversed stream it is
// one subtraction  D = T' - ((R' << 1) | cin)  whose borrow runs from a rematch through the gap in front of it and stops at
// the first event it meets; ends = T' & D.  cin: the first event behind this word is a rematch (fsm_lanes_succ_r, the same
// trick one level up, lanes as bits).
// T: a lane's 128 event bits (two sub-chunks, ascending); Er: its row ends, bit-reversed — Er[0] bit 0 is T[3] bit 31.
CXG_FSM_HD uint32_t fsm_brev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(x);
#else
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
  x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
  return (x >> 16) | (x << 16);
#endif
}
CXG_FSM_HD void fsm_lane_ends(const uint32_t (&T)[4], uint32_t cin, uint32_t (&Er)[4]) {
  uint32_t tr[4], a[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 0; i < 4; i++) {
    tr[i] = fsm_brev32(T[3 - i]);
    a[i] = (tr[i] & 0x55555555u) << 1;                 // rematch bits (odd before the reversal, even after), moved onto the next slot
  }
  a[0] |= cin & 1u;
  const uint64_t tl = (static_cast<uint64_t>(tr[1]) << 32) | tr[0], th = (static_cast<uint64_t>(tr[3]) << 32) | tr[2];
  const uint64_t al = (static_cast<uint64_t>(a[1]) << 32) | a[0], ah = (static_cast<uint64_t>(a[3]) << 32) | a[2];
  const uint64_t dl = tl - al, dh = th - ah - (tl < al ? 1ull : 0ull);
  Er[0] = tr[0] & static_cast<uint32_t>(dl); Er[1] = tr[1] & static_cast<uint32_t>(dl >> 32);
  Er[2] = tr[2] & static_cast<uint32_t>(dh); Er[3] = tr[3] & static_cast<uint32_t>(dh >> 32);
}
// the first (lowest) event of the 128 bits is a rematch
CXG_FSM_HD bool fsm_first_is_r(const uint32_t (&T)[4]) {
  const uint32_t w = T[0] ? T[0] : (T[1] ? T[1] : (T[2] ? T[2] : T[3]));
  return ((w & (0u - w)) & 0xAAAAAAAAu) != 0u;
}
// ne: lanes with events, fr: lanes whose first event is a rematch (a subset of ne).  Returns the lanes whose LAST event
// is followed by a rematch somewhere in the lanes above.
CXG_FSM_HD uint64_t fsm_brev64(uint64_t v) { return (static_cast<uint64_t>(fsm_brev32(static_cast<uint32_t>(v))) << 32) | fsm_brev32(static_cast<uint32_t>(v >> 32)); }
CXG_FSM_HD uint64_t fsm_lanes_succ_r(uint64_t ne, uint64_t fr) {
  const uint64_t rne = fsm_brev64(ne), rfr = fsm_brev64(fr);
  return fsm_brev64(((rfr << 1) + ~rne) & rne);
}
// low 2 * n bits (n bytes of a 32-byte sub-chunk are input)
CXG_FSM_HD uint64_t fsm_valid_bits(int32_t n) { return n >= 32 ? ~0ull : (n <= 0 ? 0ull : ((1ull << (2 * n)) - 1ull)); }

// fast: trace of the chunk's fast part, or nullptr when the chunk was not walked yet (edge chunks: end of input inside).
template <class Mem, class Rows, class Events>
CXG_FSM_HD void fsm_finish(const FsmView& v, const Mem& m, uint32_t entry, const FsmTrace* fast, int32_t c0, int32_t c1, int32_t rend,
                           int32_t budget, FsmLane& L, Rows& rows, Events& evs) {
  L.x = entry;
  L.nlev = fsm_u16(v.tab, entry + v.ncls2 + 2u);
  L.lev = 0x01010101010101ull & ((1ull << (8u * L.nlev)) - 1ull);
  L.nrows = 0;
  L.flags = 0;
  int32_t i = c0;
  L.xc1 = entry;
  if (fast) {
    i = c1;
    L.x = fast->x;
    L.xc1 = fast->x;
    uint32_t evbits = fast->evbits;
    if (fast->nev > fast->cap) { L.flags |= 8u; evbits = 0; }
    uint32_t k = 0;
    while (evbits) {                                 // apply in order
      const uint32_t pos = static_cast<uint32_t>(__builtin_ctz(evbits));
      evbits &= evbits - 1;
      fsm_apply(L, fsm_u16(v.tab, evs.row_at(k) + v.ncls2), c0 + static_cast<int32_t>(pos) + 1, true, rows);
      k++;
    }
  }
  for (;; i++) {
    if (i >= c1 && L.lev == 0u) break;               // only levels created beyond the chunk are left
    if (i >= rend) break;                            // end of input: every pending match is committed as it stands
    if (i >= budget) { L.flags |= 4u; break; }
    L.x = fsm_next(v, L.x, m.cls(v, i)) & ~3u;
    if (i < c1) L.xc1 = L.x;
    if (L.x >= v.alias_lo) fsm_apply(L, fsm_u16(v.tab, L.x + v.ncls2), i + 1, i < c1, rows);
  }
}

// One chunk [c0, c1), c1 - c0 <= kFsmSub, from entry row `entry` (pending levels of the entry state are foreign).
// rend: end of input, budget: last position the lane may read (serial-walk budget).
template <class Mem, class Rows, class Events>
CXG_FSM_HD void fsm_replay(const FsmView& v, const Mem& m, uint32_t entry, int32_t c0, int32_t c1, int32_t rend, int32_t budget,
                           FsmLane& L, Rows& rows, Events& evs) {
  if (c1 <= rend && c1 <= budget && c1 - c0 == kFsmSub) {
    const int32_t cc[1] = {c0};
    FsmTrace t[1] = {{entry, 0u, 0u, L.max_events}};
    fsm_fast<1>(v, m, cc, t, &evs);
    fsm_finish(v, m, entry, &t[0], c0, c1, rend, budget, L, rows, evs);
  } else {
    fsm_finish(v, m, entry, static_cast<const FsmTrace*>(nullptr), c0, c1, rend, budget, L, rows, evs);
  }
}

// Start of the match that ends at e: the smallest p >= bound with hay[p, e) in the language — the anchored reverse
// DFA without break-at-match (meta/compile.go:193-194), walked from e - 1 downwards (lazy.go:1769-1920).  lowest:
// first position that exists.  Returns kFsmNoStart when the reverse DFA never accepts (cannot happen for a real match);
// positions are relative to the tile origin and may be negative.  A reverse entry is the target row's byte offset | 1 when
// the target accepts (round 6); rows are a power of two long.
// fsm_match_start_from: the walk goes on from state s (an entry) with byte `at` the next to step over and st the smallest start so far.
template <class Mem>
CXG_FSM_HD int32_t fsm_match_start_from(const FsmView& v, const Mem& m, uint32_t s, int32_t st, int32_t at, int32_t bound, int32_t budget_lo, uint32_t& over, int32_t text_pos = kFsmNoStart) {
  // Four steps at a time: the byte reads and the class lookups of a group do not depend on the automaton's state and are
  // issued together — and those of the NEXT group before this group's chain (round 6) — so the dependent chain per step is
  // ONE table read instead of three (byte -> class -> row).  Positions below the walk's lower end are read clamped (they lie
  // in the window, their classes are not used): short walks between dense rows (`\b\d+\b`: the previous row ends a few
  // bytes below) take the same path instead of a byte-by-byte loop of three dependent reads per step.
  const int32_t low = bound > budget_lo ? bound : budget_lo;
  uint32_t c[4], cn[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < 4; k++) { const int32_t p = at - k; c[k] = m.rcls(v, p > budget_lo ? p : budget_lo); }
  while (at >= low) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; k++) { const int32_t p = at - 4 - k; cn[k] = m.rcls(v, p > budget_lo ? p : budget_lo); }
    bool dead = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; k++) {
      if (dead || at - k < low) break;
      s = fsm_u16(v.tab, (s & ~1u) | c[k]);
      if (s == v.rev_dead) { dead = true; break; }
      if (s & 1u) st = at - k;
    }
    if (dead) return st;
    at = at - 4 >= low - 1 ? at - 4 : low - 1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; k++) c[k] = cn[k];
  }
  if (at >= bound) over = 1u;                           // stopped at the window's first byte with the haystack going on in front of it
  // text_pos: where the text starts, relative to `m` (kFsmNoStart: not within reach).  A walk that stepped over the text's first
  // byte alive stands there: a text-start anchor of the pattern (\A, ^) holds now and nowhere else — the state says whether that
  // makes the position a match start (host/fsm.cc).
  if (v.rev_text_col != 0u && text_pos != kFsmNoStart && at == text_pos - 1 && s != v.rev_dead && !over && fsm_u16(v.tab, (s & ~1u) + v.rev_text_col + v.knd[m.byte(text_pos)]) != 0u) st = text_pos;
  return st;
}
template <class Mem>
CXG_FSM_HD int32_t fsm_match_start(const FsmView& v, const Mem& m, int32_t e, int32_t bound, int32_t budget_lo, uint32_t& over, int32_t text_pos = kFsmNoStart) {
  return fsm_match_start_from(v, m, m.rstart(v, e), kFsmNoStart, e - 1, bound, budget_lo, over, text_pos);
}

// Round 6: the first 16 steps of that walk without a branch.  Most matches of log patterns are shorter (an IPv4 address, a number, a
// word), and the loop above pays for its exits: every step tests "dead" and "below the bound" with the exec mask of the lanes that are
// still walking.  Here the 16 bytes below e come as four dwords (Mem::below: five aligned reads and v_alignbyte on the device), their
// classes are looked up at once, and a step is  s = rev[(s & ~1) | column]  + one v_alignbit that shifts the entry's accept bit into a
// word: the dead state (row 0) absorbs, steps below the bound are masked out of the word afterwards, the smallest start is its highest
// bit.  Only a walk that is still alive after 16 steps with room below goes on in the loop above.  Needs e - 17 inside the window
// (rows end behind the tile origin and the window begins 64 bytes in front of it) and a pattern without a text-start anchor
// (the caller's business).  Mem::below<N>(e, W): the N + 1 bytes e - N - 1 .. e - 1 in ascending order, byte i = (W[i >> 2] >> 8 (i & 3)) & 255.
CXG_FSM_HD uint32_t fsm_shift_in1(uint32_t mask, uint32_t t) {   // (mask >> 1) | (t << 31)
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(t, mask, 1u);
#else
  return (mask >> 1) | (t << 31);
#endif
}
// N: 16, or 8 where matches are short (the kernels take 8 for tiles with 128 rows and more: `\\b\\d+\\b`, 9 rounds of rows per tile).
template <int N, class Mem>
CXG_FSM_HD int32_t fsm_match_startN(const FsmView& v, const Mem& m, int32_t e, int32_t bound, int32_t budget_lo, uint32_t& over) {
  uint32_t W[N / 4 + 1];
  m.template below<N>(e, W);
  uint32_t c[N];
  if (Mem::kLook) {                                       // step k is over byte e - 1 - k = byte N - k of W and sees the kind of the byte in front: N + 1 lookups
    uint32_t t[N + 1];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i <= N; i++) t[i] = v.lk16[(W[i >> 2] >> (8 * (i & 3))) & 0xFFu];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < N; k++) c[k] = (t[N - k] & 0xFFu) + (t[N - k - 1] >> 8);
  } else {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < N; k++) { const int i = N - k; c[k] = v.cls2[(W[i >> 2] >> (8 * (i & 3))) & 0xFFu]; }
  }
  uint32_t s = m.rstart(v, e), acc = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < N; k++) {
    s = fsm_u16(v.tab, (s & ~1u) | c[k]);               // (an entry is the row's offset in the image = its LDS address: no base to add)
    acc = fsm_shift_in1(acc, s);
  }
  const int32_t low = bound > budget_lo ? bound : budget_lo;
  const uint32_t room = static_cast<uint32_t>(e - low);   // steps the walk may take (>= 1)
  if (room <= static_cast<uint32_t>(N) && low != bound) return fsm_match_start(v, m, e, bound, budget_lo, over);   // (the window's first byte within N bytes of a row's end: not in the kernel's geometry)
  uint32_t f = acc >> (32 - N);                           // bit k: hay[e - 1 - k, e) is in the language
  if (room < static_cast<uint32_t>(N)) f &= (1u << room) - 1u;
  int32_t st = f ? e - 1 - (31 - static_cast<int32_t>(__builtin_clz(f))) : kFsmNoStart;
  // room <= N ends at the bound (low == bound then: the window begins 64 bytes in front of the tile and rows end behind its origin)
  if (room > static_cast<uint32_t>(N) && (s & ~1u) != v.rev_dead) st = fsm_match_start_from(v, m, s, st, e - (N + 1), bound, budget_lo, over);
  return st;
}
template <class Mem>
CXG_FSM_HD int32_t fsm_match_start16(const FsmView& v, const Mem& m, int32_t e, int32_t bound, int32_t budget_lo, uint32_t& over) { return fsm_match_startN<16>(v, m, e, bound, budget_lo, over); }

}  // namespace cxgdev
