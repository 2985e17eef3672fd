// Host front-end of the MI355X FindAll path: pattern -> AST -> Thompson NFA -> strategy.
//
// This is the C++ stand-in for the part of the reference that `north_star` keeps in Go
// (coregex.Compile -> meta.CompileRegexp, meta/compile.go:440-654): there is no Go toolchain
// in this image, so the host side above the C ABI is written here, mirroring the reference's
// observable outputs for the accelerated subset:
//   * AST shape of regexp/syntax.Parse(pattern, syntax.Perl) (meta/compile.go:58) — literal-run
//     merging, alternation factoring, single-rune classes;
//   * NFA state numbering of nfa.Compiler (nfa/compile.go:99-233,1225-1682) — creation order;
//   * prefix literal extraction (literal/extractor.go:128-365) and SelectStrategy
//     (meta/strategy.go:1377-1546) for the strategies the device path serves.
// Everything outside the subset yields Unsupported so the caller keeps its CPU loop
// (CXG_E_UNSUPPORTED).  Independent of oracle/ by construction: nothing here includes it.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/coregex_hip.h"

namespace cxg {

enum class Node : uint8_t {
  NoMatch, Empty, Lit, Class, AnyNotNL, Any, BeginLine, EndLine, BeginText, EndText, WordB, NoWordB,
  Capture, Star, Plus, Quest, Repeat, Concat, Alt
};

struct Ast {
  struct N {
    Node kind = Node::Empty;
    bool fold = false;        // FoldCase literal
    bool lazy = false;        // non-greedy quantifier
    std::vector<int32_t> r;   // runes, or class range pairs
    std::vector<int> kids;
    int min = 0, max = 0, cap = 0;
  };
  std::vector<N> nodes;
  int root = -1;
  int ncap = 0;
  int add(Node k) { nodes.emplace_back(); nodes.back().kind = k; return static_cast<int>(nodes.size()) - 1; }
  N& at(int i) { return nodes[i]; }
  const N& at(int i) const { return nodes[i]; }
  std::string repr(int i) const;
};

struct FrontendError {
  int code;  // CXG_E_SYNTAX or CXG_E_UNSUPPORTED
  std::string msg;
};

Ast parsePattern(const std::string& pattern);  // throws FrontendError

struct HostNfa {
  std::vector<cxg_nfa_state> states;
  std::vector<cxg_nfa_trans> trans;
  uint32_t startAnchored = 0, startUnanchored = 0;
  uint32_t captureCount = 1;
  bool alwaysAnchored = false;
  cxg_nfa view() const {
    return cxg_nfa{states.data(), static_cast<uint32_t>(states.size()), trans.data(),
                   static_cast<uint32_t>(trans.size()), startAnchored, startUnanchored, captureCount};
  }
};

HostNfa buildNfa(const Ast& ast);  // throws FrontendError(CXG_E_UNSUPPORTED) for text anchors, \p classes

struct PrefixLit { std::vector<uint8_t> bytes; bool exact; };

struct Plan {
  int strategy = CXG_USE_NFA;
  uint32_t flags = 0;
  std::vector<PrefixLit> prefixes;
  uint8_t membership[256] = {0};   // UseCharClassSearcher
  bool confident = true;           // false: the reference's answer for this pattern is not reproduced with certainty (`why` says what)
  std::string why;
  bool lineStartAll = false;       // ... and every match of the pattern begins at a line start
  bool lineStart = false;          // the pattern holds (?m)^: a UseTeddy program checks its candidates for a line start (prefilter.WrapLineAnchor)
};

Plan selectStrategy(const Ast& ast, const HostNfa& nfa);

// The first two rules of meta.SelectStrategy (strategy.go:1377-1440), which need no NFA: a pattern anchored at the end of the text
// only is UseReverseAnchored; one anchored at its start is UseAnchoredLiteral / UseBranchDispatch / UseBoundedBacktracker (the
// last is returned, `exact` = false: the two detectors are not restated).  -1: neither rule applies.  None of these strategies has a
// device kernel (a start-anchored FindAll has one match at most); the answer only makes the refusal name the reference's engine.
int textAnchorStrategy(const Ast& ast, bool& exact);

// Bounded repetition on the chain kernel (program.cc attachBoundedChain): when the pattern is a concatenation of
// single-byte items (a literal byte, a class) and runs of one (`x+`, `x{m,n}` with m >= 1, greedy) — `(?:...){k}` groups
// unrolled — returns the surrogate pattern with every run unbounded (`\d{1,3}\.\d{1,3}` -> `\d+\.\d+`) and the
// (min, max) of each run in order (max 0 = unbounded).  False when the pattern has another shape or no bounded run.
bool boundedSurrogate(const Ast& ast, Ast& out, std::vector<std::pair<int, int>>& bounds);

}  // namespace cxg
