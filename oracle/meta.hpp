// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into the product path.
//
// meta.hpp: restatement of the reference's meta engine for the FindAll hot path.
//   Strategy / SelectStrategy   meta/strategy.go:19-230,1377-1546 (+ helpers :416-560,:974-1100,:1143-1308)
//   literal prefixes/suffixes   literal/extractor.go:128-365,376-485,575-700; literal/seq.go:206-216,343-364,433-508
//   CompileRegexp               meta/compile.go:440-654 (engine wiring :115-205,:305-379)
//   FindAllIndicesStreaming     meta/findall.go:155-283
//   Count                       meta/findall.go:297-376
//   FindAllSubmatch             meta/findall.go:63-128,390-447
//   per-strategy find           meta/find_indices.go:1050-1088 (digit), :925-951 (Teddy),
//                               :841-848 (char class), :408-441 (adaptive), :686-705 (bidirectional)
// Strategies outside SURVEY §8 (reverse searchers, backtracker, composite, branch
// dispatch, Aho-Corasick, anchored literal) are *classified* as far as the restated
// helpers allow but searched with the PikeVM: every strategy returns the same
// leftmost-first spans, so results stay authoritative while `strategyRestated` says
// whether the reference's own inner loop for that strategy is what ran here.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "engines.hpp"

namespace orc {

enum Strategy : int {  // meta/strategy.go:19-230 (same order)
  UseNFA = 0, UseDFA, UseBoth, UseReverseAnchored, UseReverseSuffix, UseOnePass, UseReverseInner,
  UseBoundedBacktracker, UseTeddy, UseReverseSuffixSet, UseCharClassSearcher, UseCompositeSearcher,
  UseBranchDispatch, UseDigitPrefilter, UseAhoCorasick, UseAnchoredLiteral, UseMultilineReverseSuffix
};
const char* strategyName(int s);

struct Lit { std::vector<uint8_t> bytes; bool complete; };
struct Seq {
  std::vector<Lit> lits;
  bool empty() const { return lits.empty(); }
  bool allComplete() const;
  std::vector<uint8_t> lcp() const;
  std::vector<uint8_t> lcs() const;
};
Seq extractPrefixes(const ReP& re);
Seq extractSuffixes(const ReP& re);
Seq extractInner(const ReP& re);      // ExtractInner extractor.go:744-810

struct Engine {
  std::string pattern;
  ReP re;
  NFA nfa, revNfa;
  Strategy strategy = UseNFA;
  bool strategyRestated = true;
  bool hasPrefilter = false;     // Engine.prefilter != nil (prefilter/prefilter.go:261-297)
  bool dfaGatesPikeVM = false;   // UseDFA over assertions, no reverse DFA, no prefilter: DFA.IsMatchAt decides whether the PikeVM runs
  Seq prefixes;
  bool digitRunSkipSafe = false;
  bool teddyLineAnchor = false;   // UseTeddy behind (?m)^: prefilter.WrapLineAnchor (compile.go:670-677, prefilter/wrap.go:45-66)
  bool hasReverseDFA = false;
  LazyDFA dfa, revDfa;
  PikeVM pikevm;
  Teddy teddy;
  CharClassSearcher ccs;

  // FindAllIndicesStreaming(haystack, n, nil): flat [s0,e0,s1,e1,...]
  void findAll(Bytes h, int64_t len, int64_t n, std::vector<int64_t>& out);
  int64_t count(Bytes h, int64_t len, int64_t n);
  // FindAllSubmatch -> FindAllSubmatchIndex rows of 2*groups, -1 unset (regex.go:1423-1450)
  void findAllSubmatch(Bytes h, int64_t len, int64_t n, std::vector<int64_t>& out);
  bool findAt(Bytes h, int64_t len, int64_t at, int64_t& s, int64_t& e);  // findIndicesAtWithState
  int numGroups() const { return nfa.captureCount; }
};

std::unique_ptr<Engine> compileEngine(const std::string& pattern);

}  // namespace orc
