// Build-time proof obligation for look-around programs of the reference's lazy-DFA strategies (lookdfa.cc).
#pragma once
#include "../../../include/coregex_hip.h"
#include "program.h"

namespace cxg {

// `reverse` = the reversed NFA of program.cc reverseOf (UseDFA: forward DFA for the end, reverse DFA for the start): returns when
// the reference's look-aware lazy DFA and its assertion-blind reverse DFA provably give the leftmost-first answer on every
// haystack, independent of cache history.  `reverse` == nullptr (UseBoth: the DFA's end only places the PikeVM's start):
// returns when that end is provably never behind the leftmost-first end.  Throws BuildError(CXG_E_UNSUPPORTED) with the reason
// otherwise.  Non-nullable patterns only (also checked here).  existenceOnly (UseDFA without reverse DFA: the DFA's IsMatchAt only
// decides whether the PikeVM runs): returns when that DFA provably never says no while a match exists.
void refuseLookDfaQuirks(const cxg_nfa& nfa, const cxg_nfa* reverse, bool existenceOnly = false);

// UseDigitPrefilter over an NFA with assertions: returns when SearchAtAnchored of the look-aware lazy DFA is provably the
// leftmost-first anchored search, history-free, and (runSkip = CXG_FLAG_DIGIT_RUN_SKIP_SAFE) the digit-run skip is sound.
void refuseLookDigitQuirks(const cxg_nfa& nfa, bool runSkip);

}  // namespace cxg
