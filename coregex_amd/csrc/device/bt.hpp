// bt.hpp — capture slots of ONE match by bounded backtracking over the pattern NFA: the capture pass for patterns
// that are not one-pass (`(a|ab)(c|bcd)`, `(a+)(a*)`), where the table walk of walk.hpp capture_walk has no forced path.
//
// Reference semantics: FindAllSubmatch reports the slots of the PikeVM's leftmost-first thread
// (nfa/pikevm.go:2186-2328; slot stamping in addSearchThread :1986).  The span [s, e) of the match is already known
// (span kernel, same leftmost-first semantics).  Leftmost-first IS "the first path a depth-first search in priority order
// completes" (the reference's own BoundedBacktracker, nfa/backtrack.go, relies on the same equivalence): explore from the
// anchored start at s, Split left before right, stamp capture slots on the way and restore them on the way back; the
// first path that reaches Match is the winner, and it ends at e because e is that winner's end.  The haystack is cut at
// e, as the reference's SearchWithCapturesInSpan does (pikevm.go:1210).  A (state, position) pair that failed once fails
// again: a visited bitmap bounds the work by states x (e - s + 1).
// Assertions (LOOK states, lo = nfa.Look 0 and 2..5: \A (?m)^ (?m)$ \b \B; pikevm.go:1646-1674) read the haystack bytes on both sides of
// the position — also in front of s and behind e — inside [hay_lo, hay_hi); outside of it counts as a line break, not a
// word byte (what the transducer kernel assumes around a shard, fsm.hpp "Look-around").
// Shared by capi_captures.hip (device) and tests/emu (host twin); plain C++.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CXG_BT_HD __host__ __device__ __forceinline__
#else
#define CXG_BT_HD inline
#endif

namespace cxgdev {

constexpr uint32_t kBtMagic = 0x43584254u;     // "CXBT"
struct BtHeader {            // device image: header, states, sparse transitions
  uint32_t magic, n_states, n_trans, start;    // start: anchored start state
  uint32_t nslots, states_off, trans_off, total_bytes;
};
struct BtState {             // 16 bytes; kinds = cxg_nfa_kind (include/coregex_hip.h)
  uint8_t kind, lo, hi, cap_slot;              // cap_slot: 2 * group + (closing ? 1 : 0), CAPTURE states
  uint32_t next, alt;                          // next: BYTE_RANGE / EPSILON / CAPTURE / SPLIT.left; alt: SPLIT.right
  uint32_t trans_off_len;                      // SPARSE: offset << 12 | length
};
struct BtTrans { uint8_t lo, hi; uint16_t pad; uint32_t next; };

constexpr uint32_t kBtInvalid = 0xFFFFFFFFu;
constexpr uint32_t kBtStackEntries = 1024;     // per thread: 8 KiB
constexpr uint32_t kBtVisitedWords = 2048;     // per thread: 8 KiB = 65 536 (state, position) pairs

// Small tier of the device kernel (capi_captures.hip k_captures_bt_lds): scratch of one thread in LDS.  Rows that do not fit are
// left to the large tier (k_captures_bt, scratch in HBM): log matches are short, so the small tier takes almost all rows
// with sixteen times the threads in flight.
constexpr uint32_t kBtSmallStack = 24, kBtSmallVisited = 16;   // 192 + 64 = 256 bytes per thread
constexpr int64_t kBtRowPending = -0x7FFFFFFFFFFFFFFFll - 1;   // row[2] of a row the small tier left to the large tier

// hay: absolute haystack; row: 2 * ngroups int64 with row[0], row[1] = s, e.  visited: `visited_words` zeroed words,
// stack: `stack_entries` entries (defaults: the large tier's kBtVisitedWords / kBtStackEntries).  Returns 0 ok, 1 the span
// is too long for the visited bitmap / the stack overflowed (the row's slots are then undefined), 2 no path reaches Match
// at e (cannot happen for a real match).  Visited / Stack: plain pointers or LDS pointers.  hay_lo / hay_hi: the absolute
// positions the haystack buffer covers (only read by LOOK states).
CXG_BT_HD bool bt_word_byte(uint32_t b) { return (b - '0' < 10u) || ((b | 0x20u) - 'a' < 26u) || b == '_'; }

// LOOK = false compiles the walk without the assertion branch: the instantiation for patterns that hold none (a LOOK state then
// ends the path, as every other kind the walk does not know).
template <bool LOOK, class Visited, class Stack>
CXG_BT_HD uint32_t bt_captures(const BtHeader* h, const uint8_t* hay, int64_t* row, uint32_t nslots, Visited visited, Stack stack,
                               uint32_t visited_words = kBtVisitedWords, uint32_t stack_entries = kBtStackEntries,
                               int64_t hay_lo = 0, int64_t hay_hi = 0) {
  const BtState* st = reinterpret_cast<const BtState*>(reinterpret_cast<const uint8_t*>(h) + h->states_off);
  const BtTrans* tr = reinterpret_cast<const BtTrans*>(reinterpret_cast<const uint8_t*>(h) + h->trans_off);
  const int64_t s = row[0], e = row[1];
  const uint64_t span = static_cast<uint64_t>(e - s) + 1;
  if (span * h->n_states > static_cast<uint64_t>(visited_words) * 32u) return 1u;
  for (uint32_t k = 2; k < nslots; k++) row[k] = -1;
  // stack entry: kind (2 bits) | payload.  0: explore (state << 32 | position offset << 2), 1: restore slot
  // (slot << 34 | (old offset + 1) << 2 | 1), old offset + 1 == 0 means "was unset"
  // Is the branch that starts at state q dead at offset `off` — would the walk, entering it, end without consuming a byte or
  // reaching Match at e?  Then it need not be kept for later: a greedy loop's exit (`(\S+)`: Epsilon, Capture, Match — not at
  // e yet) and the other lead-byte branches of a UTF-8 class are not pushed, and the stack of a long repetition stays flat
  // instead of growing by an entry per byte.  A leaf is followed through EPSILON / CAPTURE (not stamped); a right-leaning
  // chain of SPLITs (buildSplitChain) whose left leaves are all dead is dead; anything else counts as alive.
  auto dead_leaf = [&](uint32_t q, uint32_t off) -> bool {
    for (int hops = 0; hops < 6; hops++) {
      if (q == kBtInvalid || q >= h->n_states) return true;
      const BtState x = st[q];
      if (x.kind == 0) return s + static_cast<int64_t>(off) != e;
      if (x.kind == 1) return s + static_cast<int64_t>(off) >= e || hay[s + off] < x.lo || hay[s + off] > x.hi;
      if (x.kind == 2) {
        if (s + static_cast<int64_t>(off) >= e) return true;
        const uint32_t b = hay[s + off], t0 = x.trans_off_len >> 12, tn = x.trans_off_len & 0xFFFu;
        for (uint32_t k = 0; k < tn; k++) if (b >= tr[t0 + k].lo && b <= tr[t0 + k].hi) return false;
        return true;
      }
      if (x.kind == 4 || x.kind == 5) { q = x.next; continue; }
      return false;
    }
    return false;
  };
  auto dead_branch = [&](uint32_t q, uint32_t off) -> bool {
    for (int hops = 0; hops < 12; hops++) {
      if (q == kBtInvalid || q >= h->n_states) return true;
      const BtState x = st[q];
      if (x.kind == 3) { if (!dead_leaf(x.next, off)) return false; q = x.alt; continue; }
      if (x.kind == 4 || x.kind == 5) { q = x.next; continue; }
      return dead_leaf(q, off);
    }
    return false;
  };
  uint32_t sp = 0;
  stack[sp++] = (static_cast<uint64_t>(h->start) << 32);
  while (sp) {
    const uint64_t en = stack[--sp];
    if (en & 1ull) {                                                  // undo a capture stamp
      const uint32_t slot = static_cast<uint32_t>(en >> 34);
      const uint32_t old1 = static_cast<uint32_t>(en >> 2);
      row[slot] = old1 ? s + static_cast<int64_t>(old1 - 1u) : -1;
      continue;
    }
    uint32_t q = static_cast<uint32_t>(en >> 32);
    uint32_t off = static_cast<uint32_t>(en) >> 2;
    for (;;) {                                                        // follow one path until it branches or dies
      if (q == kBtInvalid || q >= h->n_states) break;
      const uint64_t bit = static_cast<uint64_t>(off) * h->n_states + q;
      if ((visited[bit >> 5] >> (bit & 31u)) & 1u) break;
      visited[bit >> 5] |= 1u << (bit & 31u);
      const BtState x = st[q];
      if (x.kind == 0 /*MATCH*/) {
        if (s + static_cast<int64_t>(off) == e) return 0u;           // the winner: slots are in place
        break;                                                        // (a shorter path: not the leftmost-first thread of this span)
      } else if (x.kind == 1 /*BYTE_RANGE*/) {
        if (s + static_cast<int64_t>(off) >= e) break;
        const uint32_t b = hay[s + off];
        if (b < x.lo || b > x.hi) break;
        q = x.next; off++;
      } else if (x.kind == 2 /*SPARSE*/) {
        if (s + static_cast<int64_t>(off) >= e) break;
        const uint32_t b = hay[s + off];
        const uint32_t t0 = x.trans_off_len >> 12, tn = x.trans_off_len & 0xFFFu;
        uint32_t nx = kBtInvalid;
        for (uint32_t k = 0; k < tn; k++) if (b >= tr[t0 + k].lo && b <= tr[t0 + k].hi) { nx = tr[t0 + k].next; break; }
        if (nx == kBtInvalid) break;
        q = nx; off++;
      } else if (x.kind == 3 /*SPLIT*/) {
        if (!dead_branch(x.alt, off)) {
          if (sp >= stack_entries) return 1u;
          stack[sp++] = (static_cast<uint64_t>(x.alt) << 32) | (static_cast<uint64_t>(off) << 2);   // right: after everything the left branch tries
        }
        q = x.next;
      } else if (x.kind == 4 /*EPSILON*/) {
        q = x.next;
      } else if (x.kind == 5 /*CAPTURE*/) {
        const uint32_t slot = x.cap_slot;
        if (slot >= 2 && slot < nslots) {
          if (sp >= stack_entries) return 1u;
          const int64_t old = row[slot];
          const uint32_t old1 = old < 0 ? 0u : static_cast<uint32_t>(old - s) + 1u;
          stack[sp++] = (static_cast<uint64_t>(slot) << 34) | (static_cast<uint64_t>(old1) << 2) | 1ull;
          row[slot] = s + static_cast<int64_t>(off);
        }
        q = x.next;
      } else if (LOOK && x.kind == 7 /*LOOK*/) {
        const int64_t pos = s + static_cast<int64_t>(off);
        const bool has_left = pos > hay_lo, has_right = pos < hay_hi;
        const uint32_t left = has_left ? hay[pos - 1] : '\n', right = has_right ? hay[pos] : '\n';
        bool ok;
        if (x.lo == 2) ok = left == '\n';                                   // StartLine
        else if (x.lo == 3) ok = right == '\n';                             // EndLine
        else if (x.lo == 4) ok = bt_word_byte(left) != bt_word_byte(right);  // WordBoundary
        else if (x.lo == 5) ok = bt_word_byte(left) == bt_word_byte(right);  // NoWordBoundary
        else if (x.lo == 0) ok = !has_left;                                  // StartText (round 4): the first position of the haystack
        else ok = !has_right;                                                // EndText (round 6): behind the haystack's last byte
        if (!ok) break;
        q = x.next;
      } else break;                                                    // FAIL
    }
  }
  return 2u;
}

}  // namespace cxgdev
