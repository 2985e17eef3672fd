// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into the product path.
//
// nfa.hpp: restatement of the reference's Thompson NFA and its compiler.
//   state kinds / State      nfa/nfa.go:23-154
//   Builder (Add*, Patch)    nfa/builder.go:34-339
//   ByteClassSet/ByteClasses nfa/alphabet.go:21-166
//   Compiler                 nfa/compile.go:99-233,237-437,1225-1682
// State ids are creation order, exactly as in the reference, because DFA
// break-at-match and PikeVM priority both depend on closure insertion order.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "syntax.hpp"

namespace orc {

using StateID = uint32_t;
constexpr StateID kInvalidState = 0xFFFFFFFFu;

enum StateKind : uint8_t {  // nfa/nfa.go:23-60
  StateMatch = 0, StateByteRange, StateSparse, StateSplit, StateEpsilon, StateCapture,
  StateFail, StateLook, StateRuneAny, StateRuneAnyNotNL
};

enum Look : uint8_t {  // nfa/nfa.go:92-117
  LookStartText = 0, LookEndText, LookStartLine, LookEndLine, LookWordBoundary, LookNoWordBoundary
};

struct Transition { uint8_t lo, hi; StateID next; };

struct NState {
  StateKind kind = StateFail;
  uint8_t lo = 0, hi = 0;
  StateID next = kInvalidState;
  std::vector<Transition> trans;
  StateID left = kInvalidState, right = kInvalidState;
  bool quantSplit = false;
  uint32_t capIndex = 0;
  bool capStart = false;
  Look look = LookStartText;
};

struct NFA {
  std::vector<NState> states;
  StateID startAnchored = kInvalidState, startUnanchored = kInvalidState;
  bool anchored = false;          // IsAlwaysAnchored()
  int captureCount = 1;           // incl. group 0
  std::array<uint8_t, 256> byteClasses{};
  int alphabetLen = 1;
  bool hasLook = false, hasWordBoundary = false;
  bool isMatch(StateID s) const { return s < states.size() && states[s].kind == StateMatch; }
};

struct CompileError { std::string msg; };

// nfa.NewCompiler{UTF8:true, Anchored:false}.CompileRegexp(re)  (meta/compile.go:442-449)
NFA compileNFA(const ReP& re);

// Helpers used by strategy selection (nfa/compile.go:1755-1850, charclass_extract.go:19-77)
bool isPatternStartAnchored(const ReP& re);
bool canMatchEmpty(const ReP& re);  // nfa/compile.go:1364-1410 (same rule set as meta/strategy.go)

std::string dumpNFA(const NFA& n);

}  // namespace orc
