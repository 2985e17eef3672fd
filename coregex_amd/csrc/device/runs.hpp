// runs.hpp — image of the alphabet-run kernel (scan_runs_wave.hip), shared with the host builder (host/lookdfa.cc
// buildRunsImage) and the sequential twin (tests/emu/emu.cc).
//
// A pattern that consumes only bytes of a small alphabet A (`(?:25[0-5]|…)\.…` : digits and '.') matches inside the maximal
// runs of A-bytes of the haystack and nowhere else, and an assertion at a position reads the two bytes around it: FindAll of
// the haystack is the concatenation of FindAll over its A-runs, each with one byte of context on either side.  Runs shorter
// than the shortest match hold nothing.  Inside a run the kernel does what the reference's digit-prefilter loop does
// (meta/find_indices.go:1050-1088) and what leftmost-first means for a non-nullable pattern: at each position in turn the
// ANCHORED leftmost-first automaton T; the first position with a match reports it and the search goes on at its end.
//
// T (lookdfa.cc buildLeftmostFirst from the anchored start, minimised): a state is the ordered thread list + the kind of the
// byte behind; the entry for (state, symbol) holds "a match ends in front of this symbol" (bit 0) and the successor's row
// offset (0 = dead; rows are an even number of bytes).  Symbols: the reference's byte classes refined by what the
// pattern's assertions tell apart (word / other, newline / other); the last column is the end of the haystack.
#pragma once
#include <stdint.h>

namespace cxgdev {

constexpr uint32_t kRunsMagic = 0x534E5552u;      // "RUNS"
constexpr uint32_t kRunsMaxRun = 255;             // longer runs: the kernel gives up (fallback flag), the host takes the transducer
constexpr uint32_t kRunsMaxImage = 6144;          // header + cls + tab; cls and tab are staged in LDS
constexpr uint32_t kRunsMaxSym = 32;              // symbols incl. the end column (cls keeps 5 bits)

struct RunsHeader {
  uint32_t magic, total_bytes;
  uint32_t nstates, nsym;                          // nsym counts the end column (index nsym - 1)
  uint32_t start[4];                               // row byte offsets by the kind of the byte behind: other, word byte, '\n', none (text start)
  uint32_t min_len;                                // the shortest match: shorter runs are skipped
  uint32_t nr;                                     // alphabet = union of nr <= 4 ASCII ranges
  uint8_t lo[4], hi[4];
  uint32_t cls_off, tab_off;                       // cls: u8[256] = symbol | kind << 5 | member << 7; tab: u16[nstates][nsym], row 0 dead
  uint32_t pad[2];
};
static_assert(sizeof(RunsHeader) == 64, "RunsHeader layout");

// One anchored attempt at position p of a byte string seen through `byte(i)` (0 <= i < n): exclusive end of the leftmost-first
// match that starts at p, or -1.  `behind` = kind of the byte in front of p (3: p is the start of the text).
template <class Bytes>
inline int64_t runs_attempt(const RunsHeader* h, const uint8_t* cls, const uint16_t* tab, const Bytes& byte, int64_t p, int64_t n, uint32_t behind) {
  uint32_t st = h->start[behind];
  int64_t last = -1;
  for (int64_t q = p; st != 0u; q++) {
    const uint32_t c = q < n ? (cls[byte(q)] & 31u) : h->nsym - 1u;
    const uint32_t e = tab[(st >> 1) + c];
    if (e & 1u) last = q;
    st = e & 0xFFFEu;
    if (q >= n) break;
  }
  return last;
}

}  // namespace cxgdev
