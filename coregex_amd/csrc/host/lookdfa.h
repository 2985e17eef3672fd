// Build-time proof obligation for look-around programs of the reference's lazy-DFA strategies (lookdfa.cc).
#pragma once
#include "../../../include/coregex_hip.h"
#include "program.h"

namespace cxg {

// Returns when the reference's look-aware lazy DFA (and, with `reverse` = the reversed NFA of program.cc reverseOf, its
// assertion-blind reverse DFA) provably gives the leftmost-first answer on every haystack and independent of cache history;
// throws BuildError(CXG_E_UNSUPPORTED) with the reason otherwise.  Non-nullable patterns only (also checked here).
void refuseLookDfaQuirks(const cxg_nfa& nfa, const cxg_nfa* reverse);

}  // namespace cxg
