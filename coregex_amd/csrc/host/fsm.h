// FindAll transducer image for scan_fsm.hip (device/fsm.hpp explains the machine).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/coregex_hip.h"
#include "../device/fsm.hpp"
#include "program.h"

namespace cxg {

constexpr uint32_t kFsmStateCap = 224;   // transducer states explored before the pattern is left to the table-walking kernels

// nfa: the pattern NFA (unanchored start required); rev: its anchored reverse DFA without break-at-match (program.cc
// reverseOf + determinize); max_len: kBothRestartSpan for UseBoth programs (a longer match raises CXG_E_INPUT), else 0.
// False + why when the pattern is outside the budget (the program then keeps its table-walking image only).
// An NFA with word-boundary assertions (CXG_NFA_LOOK states, cxg_nfa_state.lo = nfa.Look) needs revNfa, its reversed NFA
// (program.cc reverseOf): the reverse automaton is then built here, look-aware, and `rev` is ignored.
bool buildFsmImage(const cxg_nfa& nfa, const Dfa& rev, uint32_t max_len, std::vector<uint8_t>& image, std::string& why,
                   const cxg_nfa* revNfa = nullptr);

}  // namespace cxg
