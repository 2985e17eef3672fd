// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into the product path.
//
// engines.hpp: CPU restatements of the reference's search engines on the FindAll path.
//   LazyDFA           dfa/lazy/lazy.go:219-324 (SearchAtAnchored), :1102-1315 (searchAt),
//                     :1336-1446 (determinize), :1569-1613 (getStartState), :1769-1920 (SearchReverse);
//                     dfa/lazy/builder.go:183-293 (move + incremental closure), state.go:329-373 (key)
//   PikeVM            nfa/pikevm.go:147-173,1646-1674,1895-2174,2186-2432; nfa/slot_table.go:45-98
//   Teddy             prefilter/teddy.go:189-311 (masks), :391-550 (FindMatch/verifyBucket)
//   CharClassSearcher nfa/charclass_searcher.go:32-43,89-125,158-211
//   memchrDigitAt     simd/memchr_generic_impl.go:231 (scalar twin of memchr_digit_amd64.s:26)
//                     look-around: start.go:64-172 (five start kinds), look.go:88-112, builder.go:295-425
//                     (resolveWordBoundaries), :437-472 (CheckEOIMatch), lazy.go:1350-1354 ($ re-closure on '\n'),
//                     :1413-1421 (matchAtWordBoundary flags), :1533-1560 (checkWordBoundaryMatch)
// Not restated because unobservable in results: DFA cache capacity / clear / PikeVM
// fallback ladder (lazy.go:1472,1623), the 4x loop unrolling.  The prefilter skip at start-tagged states
// (lazy.go:1210-1227) is restated (`prefilterFind`): without assertions it only saves time, with them the restart at the
// candidate picks the start state by the byte in front of it.  State acceleration (lazy.go:1253-1259, builder.go:610-690) is not
// restated either: a state is examined once, on its first slow-path visit (AccelChecked latch,
// lazy.go:1650-1668), and qualifies only if nearly all of its transitions are cached by then.
//
// HISTORY DEPENDENCE, restated as it is.  (1) A state is filed under its sorted NFA set (state.go:329-373) but stepped
// in the insertion order of the variant determinized first.  (2) Transitions are cached per byte CLASS
// (nfa/alphabet.go), yet with assertions the successor depends on the byte itself — is it a word byte, is it '\n' —
// and the classes are not refined by that (nfa/builder.go:45-78 marks byte ranges only): within a class that mixes
// kinds the byte seen first decides for all.  (3) searchAt skips the word-boundary check at a start-tagged state whose
// transition is already cached (lazy.go:1229-1243).  One LazyDFA object here = one DFACache there.
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "nfa.hpp"

namespace orc {

using Bytes = const uint8_t*;

inline bool isWordByte(uint8_t b) {
  return (b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z') || (b >= 'a' && b <= 'z') || b == '_';
}

// ---------------------------------------------------------------- lazy DFA
struct LazyDFA {
  struct DState {
    std::vector<StateID> nfaStates;   // insertion order (identity for break-at-match)
    bool isMatch = false, isFromWord = false;
    bool matchAtWordBoundary = false, matchAtNonWordBoundary = false;   // lazy.go:1413-1421 (set by determinize only)
    bool startTagged = false;         // StateID.WithStartTag (lazy.go:1607)
    std::vector<int32_t> trans;       // per byte class: kUnknown / kDead / state index
  };
  static constexpr int32_t kUnknown = -2, kDead = -1;
  enum StartKind { StartNonWord = 0, StartWord, StartText, StartLineLF, StartLineCR, kStartKinds };   // start.go:19-37
  enum : uint32_t { HaveStartText = 1, HaveEndText = 2, HaveStartLine = 4, HaveEndLine = 8 };        // look.go:17-40

  const NFA* nfa = nullptr;
  bool breakAtMatch = true;           // lazy.Config.BreakAtMatch (false for reverse DFAs, meta/compile.go:193)
  bool hasWordBoundary = false, hasEndLine = false;   // builder.go:714-750
  std::vector<DState> states;
  std::map<std::vector<uint32_t>, int32_t> cache;   // exact key instead of FNV-1a (state.go:329)
  int32_t starts[2][kStartKinds];                    // [anchored][kind], -2 = not computed (start.go:64-77)
  // DFA.prefilter (lazy.CompileWithPrefilter, meta/compile.go:161): position of the first candidate at or after `pos`,
  // -1 = none.  Unset = no prefilter.
  std::function<int64_t(Bytes h, int64_t n, int64_t pos)> prefilterFind;

  void init(const NFA* n, bool brk);
  bool supported() const { return nfa != nullptr; }

  int32_t startState(Bytes h, int64_t pos, bool anchored);
  int32_t startStateOfKind(StartKind kind, bool anchored);
  int32_t next(int32_t sid, uint8_t b);               // flatTrans lookup + determinize on miss
  bool matchesEmpty();
  bool eoiMatch(int32_t sid) const;                   // checkEOIMatch, lazy.go:1512-1522
  bool wordBoundaryMatch(int32_t sid, uint8_t b) const;   // checkWordBoundaryMatch, lazy.go:1533-1560

  int64_t searchAtAnchored(Bytes h, int64_t n, int64_t at);   // lazy.go:219
  int64_t searchAt(Bytes h, int64_t n, int64_t at);           // lazy.go:190 -> :1102
  bool isMatchAt(Bytes h, int64_t n, int64_t at);             // lazy.go:546 -> :561 searchEarliestMatch
  int64_t searchReverse(Bytes h, int64_t n, int64_t start, int64_t end);  // lazy.go:1769
  size_t numStates() const { return states.size(); }

 private:
  void closureInto(std::vector<StateID>& out, std::vector<uint8_t>& in, StateID seed, uint32_t lookHave = 0) const;
  std::vector<StateID> closureOf(const std::vector<StateID>& seeds, uint32_t lookHave) const;       // builder.go:134-146
  std::vector<StateID> resolveWordBoundaries(const std::vector<StateID>& set, bool satisfied) const;   // builder.go:295-425
  bool holdsMatch(const std::vector<StateID>& set) const;
};

// Reverse automaton for SearchReverse.  The reference builds it in nfa/reverse.go:8-300;
// because the reverse DFA runs with BreakAtMatch=false its result is the set-theoretic
// minimum start, independent of state order, so only the language is restated here.
NFA reverseNFA(const NFA& fwd);

// ---------------------------------------------------------------- PikeVM
struct PikeVM {
  const NFA* nfa = nullptr;
  void init(const NFA* n) { nfa = n; }
  // SearchWithSlotTableCapturesAt (pikevm.go:2186). slots: 2*captureCount values, -1 unset.
  bool searchCaptures(Bytes h, int64_t n, int64_t at, std::vector<int64_t>& slots);
  // SearchAt (pikevm.go:747): same leftmost-first semantics, group 0 only.
  bool searchAt(Bytes h, int64_t n, int64_t at, int64_t& s, int64_t& e);
};

// ---------------------------------------------------------------- Teddy (slim: 2..32 patterns, fat: 33..64)
struct Teddy {
  std::vector<std::vector<uint8_t>> patterns;
  std::vector<std::vector<int>> buckets;
  uint16_t lo[2][16] = {}, hi[2][16] = {};   // bit = bucket (8 buckets slim, 16 fat)
  int fpLen = 2, minLen = 0;
  bool build(const std::vector<std::vector<uint8_t>>& pats);   // NewTeddy, teddy.go:189
  void findCandidate(Bytes h, int64_t n, int64_t& pos, uint16_t& mask) const;  // teddy.go:491
  bool findMatch(Bytes h, int64_t n, int64_t start, int64_t& s, int64_t& e) const;  // teddy.go:391
};

// ---------------------------------------------------------------- CharClassSearcher
struct CharClassSearcher {
  std::array<bool, 256> membership{};
  int minMatch = 1;
  void findAll(Bytes h, int64_t n, std::vector<int64_t>& out) const;   // charclass_searcher.go:158
  bool searchAt(Bytes h, int64_t n, int64_t at, int64_t& s, int64_t& e) const;  // :89
};

int64_t memchrDigitAt(Bytes h, int64_t n, int64_t at);   // simd/memchr_digit_amd64.go:34

}  // namespace orc
