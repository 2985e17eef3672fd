// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into the product path.
//
// syntax.hpp: restatement of the Go stdlib `regexp/syntax` parser (flags = syntax.Perl)
// for the ASCII subset this build accelerates.  The reference calls it at
// meta/compile.go:58 and nfa/compile.go:87; the package itself is NOT under
// /root/reference (Go stdlib, go.mod:3 pins Go 1.25.4), so this file restates its
// *published* behaviour: literal-run merging (parser.maybeConcat), alternation
// factoring rounds 1-4 (parser.factor), single-rune classes -> literals (parser.push),
// repetition squashing (parser.repeat / parser.op) and class canonicalisation
// (cleanClass).  Its output is pinned only indirectly: through the reference's
// pattern->strategy table (meta/strategy_selection_test.go:16-63) and the
// differential FindAll vectors (meta/stdlib_compat_test.go:18-199).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace orc {

enum Op : int {
  OpNoMatch = 1, OpEmptyMatch, OpLiteral, OpCharClass, OpAnyCharNotNL, OpAnyChar,
  OpBeginLine, OpEndLine, OpBeginText, OpEndText, OpWordBoundary, OpNoWordBoundary,
  OpCapture, OpStar, OpPlus, OpQuest, OpRepeat, OpConcat, OpAlternate
};

enum Flags : int {
  FoldCase = 1, Literal = 2, ClassNL = 4, DotNL = 8, OneLine = 16, NonGreedy = 32,
  PerlX = 64, UnicodeGroups = 128, WasDollar = 256, Simple = 512,
  Perl = ClassNL | OneLine | PerlX | UnicodeGroups
};

struct Regexp;
using ReP = std::shared_ptr<Regexp>;

struct Regexp {
  Op op = OpNoMatch;
  int flags = 0;
  std::vector<int> rune;   // literal runes, or class range pairs lo,hi,...
  std::vector<ReP> sub;
  int min = 0, max = 0;    // OpRepeat (max == -1: unbounded)
  int cap = 0;             // OpCapture index
  std::string name;
  bool equal(const Regexp& o) const;
};

struct ParseError { std::string msg; };

// Parse(pattern, syntax.Perl).  Throws ParseError for syntax errors and for
// constructs outside the restated subset (Unicode classes, back-references).
ReP parse(const std::string& pattern);

std::string dump(const ReP& re);  // debug s-expression

}  // namespace orc
