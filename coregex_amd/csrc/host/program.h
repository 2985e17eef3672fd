// cxg_program: strategy + eager tables built on the host, shipped to the device as one blob.
//
// The reference determinizes lazily (dfa/lazy/lazy.go:1336-1446) because state sets can blow up;
// for the accelerated subset the whole table is built up front: lazy vs eager is
// unobservable, the table must fit LDS anyway, and a pattern whose DFA does not fit is refused
// (CXG_E_UNSUPPORTED) instead of degraded.  Semantics kept from the reference:
//   * closure insertion order  dfa/lazy/builder.go:245-293 (stack, push right then left)
//   * break-at-match           dfa/lazy/builder.go:210-213 (forward DFAs only, meta/compile.go:193)
//   * byte-range / sparse move dfa/lazy/builder.go:215-230
// Deliberate difference: a DFA state is identified by its *ordered* NFA list (the reference keys on
// the sorted set, dfa/lazy/state.go:342-346, and keeps whichever order it met first).
#pragma once
#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/coregex_hip.h"
#include "../device/walk.hpp"
#include "frontend.h"

namespace cxg {

struct Dfa {
  uint32_t nstates = 0;          // including the dead state 0
  uint32_t start = 0;
  uint32_t firstAccept = 0;      // states >= firstAccept contain Match
  std::vector<uint8_t> table;    // [nstates][256]
};

struct BuildError { int code; std::string msg; };

// Throws BuildError(CXG_E_UNSUPPORTED) when the NFA has look-around or the DFA exceeds maxStates.
Dfa determinize(const cxg_nfa& nfa, uint32_t startState, bool breakAtMatch, uint32_t maxStates);
HostNfa reverseOf(const cxg_nfa& fwd);             // language-reversed automaton (nfa/reverse.go:8-300)
void alphabetOf(const cxg_nfa& nfa, bool inAlphabet[256]);  // bytes some pattern transition accepts

}  // namespace cxg

struct cxg_program {
  int strategy = CXG_USE_NFA;
  uint32_t flags = 0;
  int ngroups = 1;
  int nfaStates = -1;
  bool supported = false;
  std::string whyNot;
  // Nullable pattern (`a*`, `x?y*`; always UseNFA, meta/strategy.go:1503): the device program is the pattern's NON-EMPTY variant;
  // capi_nullable.hip scanNullable merges its rows with the empty matches of meta/findall.go:251-275.  nullableOnlyEmpty: no path of
  // higher priority than the empty one consumes a byte (`a*?`, `(|a)`): every match is empty, no device program at all.
  bool nullable = false, nullableOnlyEmpty = false;
  cxg::HostNfa nfa;              // kept for cxg_program_nfa (cxg_compile only)
  cxg::Dfa fwd, rev;
  std::vector<uint8_t> blob;     // cxgdev::BlobHeader + tables
  // FindAll transducer (device/fsm.hpp, host/fsm.cc): the general-DFA kernel scan_fsm.hip.  Empty when the program is
  // served by a bit-parallel / literal / char-class kernel alone or the transducer exceeds its budget (fsmWhyNot).
  std::vector<uint8_t> fsmBlob, subFsmBlob;   // FindAllIndex / Count; spans of FindAllSubmatchIndex
  std::string fsmWhyNot;
  // FindAllSubmatchIndex: spans from a bidirectional DFA image + one-pass capture table (any strategy:
  // the reference sends FindAllSubmatch of DFA/Both/NFA/DigitPrefilter engines to the PikeVM, whose
  // result is plain leftmost-first, meta/findall.go:89-98)
  bool subSupported = false;
  bool subNullable = false;      // FindAllSubmatch of a nullable pattern: spans by the FindAllIndex program, slots by the backtracking pass (no span image)
  std::string subWhyNot;
  std::vector<uint8_t> subBlob;  // kKindBidir image
  std::vector<uint8_t> capBlob;  // cxgdev::CapHeader + arrays
  bool capHasLook = false;       // the backtracking image holds assertion states: capi_captures.hip launches the LOOK instantiation of its kernels
  mutable std::atomic<uint8_t> noPair[2] = {{0}, {0}};       // scan_teddy_pair.hip raised its fallback flag on this program's input once: scan_teddy_wave.hip from then on
  mutable std::atomic<uint8_t> denseChain[2] = {{0}, {0}};   // [spans, submatch]: a wave kernel overflowed its row buffers on this program's
                                                              // input once: later calls start with two tiles per wave (capi_ladder.hip)
  mutable std::atomic<uint8_t> fsmMode[2] = {{0}, {0}};      // ... the transducer kernel's density mode seen necessary: 0, 1 (2 tiles per wave), 2 (1 tile)
  mutable std::atomic<uint8_t> fsmNoDirect[2] = {{0}, {0}};  // ... its lean kernel (k_scan_fsml) met an entry state that did not collapse: k_scan_fsm from now on
  // Offset captures (round 4): every capture boundary lies a fixed number of bytes behind the match's start or in front of its end
  // (`user=(\S+)`, `"([^"]*)"`, `\[([^\]]+)\]`).  FindAllSubmatch is then FindAll + one expansion kernel (capi_nullable.hip scanOffsetCaps)
  // instead of a backtracking pass per row.  offCaps[0] != 0: on; slot k >= 2: offSrc[k] 0 = start, 1 = end; offDelta[k] added.
  uint8_t offCapsOn = 0, offSrc[32] = {0};
  int32_t offDelta[32] = {0};
  uint32_t delim[4] = {0, 0, 0, 0};   // cxgdev::DelimAux: `O [^E]+ E` / `O [^E]* E` programs ([3] != 0), the delimiter kernel in front of the transducer
  uint8_t chainBounds[40] = {0}; // cxgdev::ChainCaps with on == 2: field bounds of a bounded-repetition program (kFlagChainBounded)
  uint8_t chainCaps[40] = {0};   // cxgdev::ChainCaps: captures straight from the chain kernel ([0] == 0: not available)
  // device copies, one per device, created on first use (capi_state.hip)
  void* dev[16] = {nullptr};
  void* devSub[16] = {nullptr};
  void* devCap[16] = {nullptr};
  void* devFsm[16] = {nullptr};
  void* devSubFsm[16] = {nullptr};
};

namespace cxg {
// False + why when a caller-supplied NFA has an index out of range / an unknown kind (cxg_program_from_nfa -> CXG_E_INVALID).
bool validateNfa(const cxg_nfa& nfa, std::string& why);
// Fills p->fwd/rev/blob/supported from (nfa, strategy, flags).  Never throws: unsupported programs
// get supported=false + whyNot.
void buildProgramFromNfa(cxg_program* p, const cxg_nfa& nfa, int strategy, uint32_t flags);
// `strategy` (meta.Strategy of the engine, -1 unknown) only matters for an NFA with assertions: their captures are served for the
// strategies whose FindAllSubmatch is the PikeVM (meta/findall.go:89-98) and for UseTeddy behind (?m)^ when the program itself is served.
void buildSubmatchProgram(cxg_program* p, const cxg_nfa& nfa, int strategy = -1);
// Bounded repetition: `surrogate` is the NFA of the pattern with every bounded run made unbounded (frontend.h
// boundedSurrogate), bounds the (min, max) of its runs in order.  Adds the surrogate's chain to an already built,
// supported digit / DFA-pair program when that chain has a shape the BND kernels take; otherwise leaves p alone.
void attachBoundedChain(cxg_program* p, const cxg_nfa& surrogate, const std::vector<std::pair<int, int>>& bounds);   // fills subBlob/capBlob/subSupported
// Fills p->offCapsOn / offSrc / offDelta from the NFA when every capture slot sits on the split-free chain behind the pattern's
// start or in front of its Match (each of those states entered from exactly one place, each slot written by one CAPTURE state).
void deriveOffsetCaps(cxg_program* p, const cxg_nfa& nfa);
void buildProgramFromCharClass(cxg_program* p, const uint8_t membership[256], uint32_t minMatch, bool pairs = false);
void buildProgramFromLiterals(cxg_program* p, const std::vector<std::vector<uint8_t>>& lits);
}  // namespace cxg
