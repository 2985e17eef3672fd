// host/fsm.cc — builds the FindAll transducer image (device/fsm.hpp) from an NFA.
//
// Semantics restated from the reference, as program.cc determinize does: ordered epsilon closure
// (dfa/lazy/builder.go:245-293: stack, push right then left), byte-range / sparse move in list order
// (builder.go:215-230), break at the first Match of a list (builder.go:210-213, forward searches only,
// meta/compile.go:193).  What is new is that the FindAll loop around the search (meta/findall.go:216-239: take the
// last accepting position when the DFA dies, restart there) is determinized too — a state is a STACK of thread lists,
// see fsm.hpp — so the machine runs left to right without ever moving back.
#include "fsm.h"

#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <map>
#include <set>

namespace cxg {

namespace {

constexpr uint32_t kSep = 0xFFFFFFFEu;     // separates the levels of a stack in its key vector
constexpr uint8_t kLookStartText = 0, kLookEndText = 1, kLookStartLine = 2, kLookEndLine = 3, kLookWordBoundary = 4, kLookNoWordBoundary = 5;   // nfa.Look (nfa/nfa.go:92-117), carried in cxg_nfa_state.lo
inline int wordKind(int b) { return (b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z') || b == '_' || (b >= 'a' && b <= 'z'); }

struct Stepper {
  const cxg_nfa& n;
  std::vector<uint32_t> mark;
  uint32_t gen = 0;
  std::vector<uint32_t> stack;
  // look-around context of the position the closure is taken at: kinds of the byte in front of it and of the byte
  // behind it (word[k] / newline[k] say what kind k is; the positions outside the haystack have a kind too: not a word
  // byte, and a line edge); checkLook, nfa/pikevm.go:1646-1674
  int left = 0, right = 0;
  bool word[4] = {false, false, false, false}, newline[4] = {false, false, false, false};
  bool endText[4] = {false, false, false, false};                  // the kind "behind the haystack's last byte" (round 6): no byte has it
  bool textStart = false;                                          // the position is the start of the text (\A, ^ without (?m)): only the
                                                                   // search at the haystack's first byte and the reverse walk that reaches it
  explicit Stepper(const cxg_nfa& nfa) : n(nfa), mark(nfa.n_states, 0) {}
  bool lookHolds(uint8_t look) const {
    if (look == kLookWordBoundary) return word[left] != word[right];
    if (look == kLookNoWordBoundary) return word[left] == word[right];
    if (look == kLookStartLine) return newline[left];            // pos == 0 || hay[pos-1] == '\n'
    if (look == kLookEndLine) return newline[right];             // pos == len || hay[pos] == '\n'
    if (look == kLookStartText) return textStart;
    if (look == kLookEndText) return endText[right];             // pos == len
    return false;
  }
  void closure(std::vector<uint32_t>& out, uint32_t seed) {      // epsilonClosureInto, builder.go:245-293
    stack.clear();
    stack.push_back(seed);
    while (!stack.empty()) {
      const uint32_t cur = stack.back();
      stack.pop_back();
      if (cur == CXG_NFA_INVALID || cur >= n.n_states || mark[cur] == gen) continue;
      mark[cur] = gen;
      out.push_back(cur);
      const cxg_nfa_state& s = n.states[cur];
      if (s.kind == CXG_NFA_EPSILON || s.kind == CXG_NFA_CAPTURE) { if (s.next != CXG_NFA_INVALID) stack.push_back(s.next); }
      else if (s.kind == CXG_NFA_LOOK) { if (lookHolds(s.lo) && s.next != CXG_NFA_INVALID) stack.push_back(s.next); }
      else if (s.kind == CXG_NFA_SPLIT) {
        if (s.right != CXG_NFA_INVALID) stack.push_back(s.right);
        if (s.left != CXG_NFA_INVALID) stack.push_back(s.left);
      }
    }
  }
  // one level moves on byte `b`; the list never holds a Match state on entry (lists are cut there)
  std::vector<uint32_t> step(const uint32_t* lst, size_t len, int b) {
    std::vector<uint32_t> out;
    gen++;
    for (size_t i = 0; i < len; i++) {
      const cxg_nfa_state& s = n.states[lst[i]];
      if (s.kind == CXG_NFA_BYTE_RANGE) { if (b >= s.lo && b <= s.hi) closure(out, s.next); }
      else if (s.kind == CXG_NFA_SPARSE)
        for (uint32_t k = 0; k < s.trans_len; k++) {
          const cxg_nfa_trans& t = n.trans[s.trans_off + k];
          if (b >= t.lo && b <= t.hi) closure(out, t.next);
        }
    }
    return out;
  }
  int matchIndex(const std::vector<uint32_t>& l) const {
    for (size_t i = 0; i < l.size(); i++) if (n.states[l[i]].kind == CXG_NFA_MATCH) return static_cast<int>(i);
    return -1;
  }
};

}  // namespace

namespace {
bool buildFsmImageCapped(const cxg_nfa& nfa, const Dfa& rev, uint32_t max_len, std::vector<uint8_t>& image, std::string& why, const cxg_nfa* revNfa, uint32_t rowBudget);
}

// The uncertainty rows (sets of possible entry states) are an optimisation of the table, not part of the machine: a set that is
// not tabulated maps to the absorbing wide row and the chunk's entry is resolved by the maps / the fallback instead.  When the
// image does not fit the LDS budget with 512 of them, fewer are tabulated before the program is given up (the
// in-loop estimate undercounts a row — (classes + 3) entries rounded up to a power of two — and knows nothing of the alias rows,
// the member lists and the reverse table, so the hard checks used to reject images that fit with fewer set rows).
bool buildFsmImage(const cxg_nfa& nfa, const Dfa& rev, uint32_t max_len, std::vector<uint8_t>& image, std::string& why, const cxg_nfa* revNfa) {
  static const bool noRetry = getenv("CXG_FSM_NO_SET_ROW_RETRY") != nullptr;   // A/B: round 2's behaviour (512 rows or nothing)
  for (uint32_t budget = 512; budget >= 1; budget /= 4) {
    if (noRetry && budget != 512) break;
    why.clear();
    if (buildFsmImageCapped(nfa, rev, max_len, image, why, revNfa, budget)) return true;
    if (why.find("exceeds the LDS budget") == std::string::npos) return false;
  }
  return false;
}

namespace {
bool buildFsmImageCapped(const cxg_nfa& nfa, const Dfa& rev, uint32_t max_len, std::vector<uint8_t>& image, std::string& why, const cxg_nfa* revNfa, const uint32_t rowBudget) {
  image.clear();
  bool hasWord = false, hasLine = false, hasText = false, hasEnd = false;
  for (uint32_t i = 0; i < nfa.n_states; i++)
    if (nfa.states[i].kind == CXG_NFA_LOOK) {
      const uint8_t lk = nfa.states[i].lo;
      if (lk == kLookWordBoundary || lk == kLookNoWordBoundary) hasWord = true;
      else if (lk == kLookStartLine || lk == kLookEndLine) hasLine = true;
      else if (lk == kLookStartText) hasText = true;
      else if (lk == kLookEndText) hasEnd = true;
      else { why = "unknown assertion in NFA"; return false; }
    }
  // Text start (\A, ^ without (?m); nfa.LookStartText, dfa/lazy/start.go:64-172 StartText): holds at position 0 and nowhere else.
  // Forward it only changes the state the scan STARTS in (the start rows below are closed with it); positions > 0 are searched
  // from states closed without it.  Backward it can only make position 0 a match start: every reverse state carries a flag
  // "accepting if this is the start of the text", read by the walk that arrives at position 0 alive (fsm.hpp fsm_match_start).
  // Text end (\z, $ without (?m); nfa.LookEndText, nfa/pikevm.go:1651; round 6): holds at position len and nowhere else.  The position
  // behind the haystack gets a kind of its own, the last one (kEnd): not a word byte, a line edge, and the only right-hand kind at
  // which the anchor holds.  The step over the haystack's last byte takes its column (fsm.hpp "End of text"); nothing else ever does.
  const bool hasLook = hasWord || hasLine || hasText || hasEnd;
  if (hasLook && !revNfa) { why = "internal: look-around program without its reversed NFA"; return false; }
  // kinds of the byte behind a step (fsm.hpp "Look-around"): what the pattern's assertions tell apart
  const uint32_t nkBytes = (hasWord && hasLine) ? 3u : ((hasWord || hasLine || hasText) ? 2u : 1u);   // kinds a byte can have
  const uint32_t nk = nkBytes + (hasEnd ? 1u : 0u);
  const int kEnd = hasEnd ? static_cast<int>(nkBytes) : -1;
  const int kWord = hasWord ? 1 : -1, kNl = hasLine ? (hasWord ? 2 : 1) : -1;
  auto kindOfByte = [&](int b) { return (hasWord && wordKind(b)) ? kWord : ((hasLine && b == '\n') ? kNl : 0); };
  const int outsideKind = hasLine ? kNl : 0;     // in front of / behind the haystack: a line edge, not a word byte
  const uint32_t outsideByte = hasLine ? '\n' : 0u;
  if (nfa.start_unanchored == nfa.start_anchored) { why = "start-anchored pattern"; return false; }
  // byte classes (nfa/alphabet.go:100-166)
  bool boundary[256] = {false};
  auto markb = [&](int lo, int hi) { if (lo > 0) boundary[lo - 1] = true; boundary[hi] = true; };
  for (uint32_t i = 0; i < nfa.n_states; i++) {
    const cxg_nfa_state& s = nfa.states[i];
    if (s.kind == CXG_NFA_BYTE_RANGE) markb(s.lo, s.hi);
    else if (s.kind == CXG_NFA_SPARSE) for (uint32_t k = 0; k < s.trans_len; k++) markb(nfa.trans[s.trans_off + k].lo, nfa.trans[s.trans_off + k].hi);
  }
  if (hasWord) { markb('0', '9'); markb('A', 'Z'); markb('_', '_'); markb('a', 'z'); }   // a class is of one kind
  if (hasLine) markb('\n', '\n');
  std::vector<int> reps;
  uint8_t cls[256];
  for (int b = 0; b < 256; b++) { if (b == 0 || boundary[b - 1]) reps.push_back(b); cls[b] = static_cast<uint8_t>(reps.size() - 1); }
  const uint32_t nbc = static_cast<uint32_t>(reps.size());      // byte classes
  const uint32_t ncls = nbc * nk;                               // input symbols = table columns: nk * class + kind of the next byte
  if (ncls > 64) { why = "more than 64 input symbols (byte classes x kinds)"; return false; }
  auto kindOf = [&](uint32_t bc) { return kindOfByte(reps[bc]); };
  auto setKinds = [&](Stepper& x) { if (kWord >= 0) x.word[kWord] = true; if (kNl >= 0) x.newline[kNl] = true; if (kEnd >= 0) { x.newline[kEnd] = true; x.endText[kEnd] = true; } };

  Stepper st(nfa);
  setKinds(st);
  // the search that starts at a position, by the kinds of the bytes on its two sides
  std::vector<uint32_t> freshLR[4][4];
  for (int l = 0; l < static_cast<int>(nkBytes); l++)              // (the kind in front of a position is a byte's, or the outside kind: never kEnd)
    for (int r = 0; r < static_cast<int>(nk); r++) {
      st.gen++;
      st.left = l; st.right = r;
      st.closure(freshLR[l][r], nfa.start_unanchored);
      if (st.matchIndex(freshLR[l][r]) >= 0) { why = "nullable pattern (empty matches)"; return false; }
    }

  // ---- transducer states: stacks of thread lists
  std::map<std::vector<uint32_t>, uint32_t> ids;
  std::vector<std::vector<uint32_t>> keys;
  std::vector<uint8_t> levels;                    // pending levels of a state
  std::vector<std::vector<uint32_t>> trans;       // [state][class] next | event descriptor << 16
  uint32_t depth = 0;
  bool createUnderPending = false;                // some step creates a match while an older pending level stays alive
  bool tooBig = false;
  auto intern = [&](const std::vector<std::vector<uint32_t>>& stack) -> uint32_t {
    std::vector<uint32_t> key;
    for (size_t l = 0; l < stack.size(); l++) { if (l) key.push_back(kSep); key.insert(key.end(), stack[l].begin(), stack[l].end()); }
    auto it = ids.find(key);
    if (it != ids.end()) return it->second;
    if (keys.size() >= kFsmStateCap || stack.size() - 1 > static_cast<size_t>(cxgdev::kFsmMaxLevels)) { tooBig = true; return 0; }
    const uint32_t id = static_cast<uint32_t>(keys.size());
    ids.emplace(key, id);
    keys.push_back(key);
    levels.push_back(static_cast<uint8_t>(stack.size() - 1));
    depth = std::max<uint32_t>(depth, static_cast<uint32_t>(stack.size() - 1));
    trans.emplace_back(ncls, 0);
    return id;
  };
  auto eventOf = [&](uint32_t kind, uint32_t j, bool conts, uint32_t died) -> uint32_t {   // descriptor, 0 = nothing happened
    return kind | (j << 2) | (conts ? 32u : 0u) | (died << 8);
  };
  uint32_t startOf[4] = {0, 0, 0, 0};                // the search at the haystack's first byte, by the kind of that byte; row 0 = kind 0
  for (uint32_t k = 0; k < nk; k++) {
    std::vector<uint32_t> first;
    st.gen++;
    st.left = outsideKind; st.right = static_cast<int>(k);
    st.textStart = hasText;
    st.closure(first, nfa.start_unanchored);
    st.textStart = false;
    if (st.matchIndex(first) >= 0) { why = static_cast<int>(k) == kEnd ? "nullable pattern (matches the empty text)" : "nullable pattern (empty match at the start of the text)"; return false; }
    if (static_cast<int>(k) != kEnd) startOf[k] = intern({first});   // (an empty haystack is never scanned: no start row for kEnd)
  }
  for (uint32_t cur = 0; cur < keys.size() && !tooBig; cur++) {
    // split the key into its levels
    std::vector<std::pair<size_t, size_t>> lv;    // [begin, end) in keys[cur]
    {
      const std::vector<uint32_t>& k = keys[cur];
      size_t b = 0;
      for (size_t i = 0; i <= k.size(); i++) if (i == k.size() || k[i] == kSep) { lv.emplace_back(b, i); b = i + 1; }
    }
    const size_t nl = lv.size() - 1;              // pending levels; lv[nl] is the innermost search
    for (uint32_t c = 0; c < ncls && !tooBig; c++) {
      const int b = reps[c / nk];
      st.left = kindOf(c / nk); st.right = static_cast<int>(c % nk);   // both sides of the position behind this byte
      const std::vector<uint32_t>& fresh = freshLR[st.left][st.right];
      const std::vector<uint32_t> key = keys[cur];   // copy: keys grows
      std::vector<std::vector<uint32_t>> next;
      uint32_t died = 0, ev = 0;
      bool done = false;
      for (size_t j = 0; j < nl && !done; j++) {
        std::vector<uint32_t> moved = st.step(key.data() + lv[j].first, lv[j].second - lv[j].first, b);
        const int m = st.matchIndex(moved);
        if (m >= 0) {                               // pending level j matches again: its end moves, deeper levels vanish
          moved.resize(static_cast<size_t>(m));
          const bool conts = !moved.empty();
          if (conts) next.push_back(std::move(moved));
          next.push_back(fresh);
          ev = eventOf(cxgdev::kFsmEvRematch, static_cast<uint32_t>(j), conts, died);
          done = true;
        } else if (moved.empty()) died |= 1u << j;  // committed relative to its parents
        else next.push_back(std::move(moved));
      }
      if (!done) {
        std::vector<uint32_t> moved = st.step(key.data() + lv[nl].first, lv[nl].second - lv[nl].first, b);
        const int m = st.matchIndex(moved);
        if (m >= 0) {
          moved.resize(static_cast<size_t>(m));
          const bool conts = !moved.empty();
          if (conts) next.push_back(std::move(moved));
          next.push_back(fresh);
          ev = eventOf(cxgdev::kFsmEvCreate, 0, conts, died);
          if (died != (nl ? (1u << nl) - 1u : 0u)) createUnderPending = true;
        } else {
          if (moved.empty()) { why = "internal: innermost search died (unanchored prefix missing)"; return false; }
          next.push_back(std::move(moved));
          ev = eventOf(cxgdev::kFsmEvDied, 0, false, died);
        }
      }
      const uint32_t to = intern(next);
      trans[cur][c] = to | (ev << 16);
    }
  }
  if (tooBig) { why = "FindAll transducer exceeds the table budget (states, pending levels or events)"; return false; }
  // FsmHeader::depth <= 1 selects the two-bitmap row derivation of the kernel (fsm.hpp fsm_finish_shallow), which
  // attributes a rematch to the LATEST created row.  That holds only when no match is created while an older pending level
  // survives the step: `(?:ab)*[ab]` on "abb" creates [1,2) under the pending [0,1) — no threads of its own, so the stack
  // stays one deep — and the parent then grows to [0,3) and must drop it.  Such machines take the event-list path.
  if (createUnderPending && depth < 2) depth = 2;
  // ---- Round 6: minimise.  Stacks of ordered thread lists tell apart much that no input can: the README IPv4 pattern
  // (README.md:64) explores 95 stacks that behave as 20 states, `https?://[^ ]+` 28 as 10.  What the kernels observe of a state
  // is its number of pending levels and, per input symbol, the event descriptor and the next state — Moore's refinement over
  // exactly that.  Fewer states means a smaller image (more workgroups per CU), fewer and smaller sets of possible entry states
  // (they collapse sooner), and it is what lets the byte-indexed tables of the kernel's direct mode fit into LDS.
  static const bool noMinimise = getenv("CXG_FSM_NO_MINIMISE") != nullptr;   // A/B
  if (!noMinimise) {
    const uint32_t n0 = static_cast<uint32_t>(keys.size());
    std::vector<uint32_t> part(n0);
    for (uint32_t s = 0; s < n0; s++) part[s] = levels[s];
    size_t nblocks = 0;
    for (;;) {
      std::map<std::vector<uint32_t>, uint32_t> sig;
      std::vector<uint32_t> next(n0);
      for (uint32_t s = 0; s < n0; s++) {
        std::vector<uint32_t> k;
        k.reserve(ncls + 1);
        k.push_back(part[s]);
        for (uint32_t c = 0; c < ncls; c++) k.push_back((trans[s][c] & 0xFFFF0000u) | part[trans[s][c] & 0xFFFFu]);
        next[s] = sig.emplace(std::move(k), static_cast<uint32_t>(sig.size())).first->second;   // blocks numbered by first occurrence: state 0 stays 0
      }
      part.swap(next);
      if (sig.size() == nblocks) break;
      nblocks = sig.size();
    }
    if (nblocks < n0) {
      std::vector<std::vector<uint32_t>> ntrans(nblocks);
      std::vector<uint8_t> nlevels(nblocks, 0);
      std::vector<std::vector<uint32_t>> nkeys(nblocks);
      std::vector<bool> have(nblocks, false);
      for (uint32_t s = 0; s < n0; s++) {
        const uint32_t b = part[s];
        if (have[b]) continue;
        have[b] = true;
        nlevels[b] = levels[s];
        nkeys[b] = keys[s];
        ntrans[b].resize(ncls);
        for (uint32_t c = 0; c < ncls; c++) ntrans[b][c] = (trans[s][c] & 0xFFFF0000u) | part[trans[s][c] & 0xFFFFu];
      }
      for (uint32_t k = 0; k < nk; k++) startOf[k] = part[startOf[k]];
      trans.swap(ntrans); levels.swap(nlevels); keys.swap(nkeys);
    }
  }
  const uint32_t nT = static_cast<uint32_t>(keys.size());

  // ---- uncertainty rows: sets of states, from "any state" (top) until they collapse to one state
  std::map<std::vector<uint8_t>, uint32_t> setId;
  std::vector<std::vector<uint8_t>> sets;
  std::vector<std::vector<uint16_t>> utrans;
  bool wideUsed = false;
  // (rowBudget: sets tabulated at most; the rest maps to the wide row)
  {
    std::vector<uint8_t> top(nT);
    for (uint32_t i = 0; i < nT; i++) top[i] = static_cast<uint8_t>(i);
    setId.emplace(top, 0);
    sets.push_back(top);
  }
  constexpr uint16_t kWideMark = 0xFFFF, kSetBase = 0x8000;
  for (size_t cur = 0; cur < sets.size(); cur++) {
    utrans.emplace_back(ncls, 0);
    for (uint32_t c = 0; c < ncls; c++) {
      std::set<uint8_t> img;
      for (uint8_t s : sets[cur]) img.insert(static_cast<uint8_t>(trans[s][c] & 0xFFFFu));
      if (img.size() == 1) { utrans[cur][c] = *img.begin(); continue; }
      std::vector<uint8_t> v(img.begin(), img.end());
      auto it = setId.find(v);
      if (it == setId.end()) {
        if (sets.size() >= rowBudget || (nT + sets.size() + 64) * (ncls + 2) * 2 > cxgdev::kFsmMaxTableBytes) { utrans[cur][c] = kWideMark; wideUsed = true; continue; }
        it = setId.emplace(v, static_cast<uint32_t>(sets.size())).first;
        sets.push_back(v);
      }
      utrans[cur][c] = static_cast<uint16_t>(kSetBase + it->second);
    }
  }
  const uint32_t nU = static_cast<uint32_t>(sets.size());
  (void)wideUsed;
  // alias rows: one per distinct (target state, event) pair
  std::map<uint32_t, uint32_t> aliasOf;                        // to | ev << 16 -> alias index
  std::vector<uint32_t> aliases;
  for (uint32_t s = 0; s < nT; s++)
    for (uint32_t c = 0; c < ncls; c++)
      if (trans[s][c] >> 16) { if (aliasOf.emplace(trans[s][c], static_cast<uint32_t>(aliases.size())).second) aliases.push_back(trans[s][c]); }
  // ordered by event kind (died only, create, rematch): shallow machines read the kind off the row's position
  std::stable_sort(aliases.begin(), aliases.end(), [](uint32_t a, uint32_t b) { return ((a >> 16) & 3u) < ((b >> 16) & 3u); });
  for (size_t i = 0; i < aliases.size(); i++) aliasOf[aliases[i]] = static_cast<uint32_t>(i);
  const uint32_t nA = static_cast<uint32_t>(aliases.size());
  uint32_t nDied = 0, nCreate = 0;
  for (uint32_t a : aliases) { if (((a >> 16) & 3u) == cxgdev::kFsmEvDied) nDied++; else if (((a >> 16) & 3u) == cxgdev::kFsmEvCreate) nCreate++; }
  // rows are a power of two long: a walk step is then  x = tab[(entry & ~3) | 2 * class]  — one v_and_or on the chain —
  // and the two low bits of an entry are free for the event flags of shallow machines (fsm.hpp)
  uint32_t rowBytes = 16;
  uint32_t rowShift = 4;
  while (rowBytes < (ncls + 3) * 2) { rowBytes *= 2; rowShift++; }
  const uint32_t stride = rowBytes / 2;
  const uint32_t nRows = nT + nA + nU + 1;
  if (static_cast<size_t>(nRows) * rowBytes > cxgdev::kFsmMaxTableBytes) { why = "FindAll transducer table exceeds the LDS budget"; return false; }
  // Reverse automaton of a look-around program: subset construction over the same symbols, mirrored — the step over byte i
  // (walking down) sees the kind of hay[i-1], so both sides of position i are known when its closure is taken.  Sets, not
  // lists: the reverse search runs without break-at-match (meta/compile.go:193-194).  Row 0 dead, accepting rows last.
  std::vector<std::vector<uint32_t>> rtab;          // [state][ncls] (renumbered)
  uint32_t rStates = rev.nstates, rFirstAccept = rev.firstAccept, rStart = rev.start;
  uint32_t rStart9[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<uint8_t> rTextAcc;                   // hasText: per reverse state (renumbered)
  if (hasLook) {
    Stepper rs(*revNfa);
    setKinds(rs);
    std::map<std::vector<uint32_t>, uint32_t> rid;
    std::vector<std::vector<uint32_t>> rsets;
    std::vector<std::vector<uint32_t>> rnext;
    auto rintern = [&](std::vector<uint32_t> set) -> uint32_t {
      std::sort(set.begin(), set.end());
      set.erase(std::unique(set.begin(), set.end()), set.end());
      auto it = rid.find(set);
      if (it != rid.end()) return it->second;
      const uint32_t id = static_cast<uint32_t>(rsets.size());
      rid.emplace(set, id);
      rsets.push_back(set);
      rnext.emplace_back(ncls, 0u);
      return id;
    };
    rintern({});                                   // 0: dead
    uint32_t s9[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int l = 0; l < static_cast<int>(nkBytes); l++)
      for (int r = 0; r < static_cast<int>(nk); r++) {
        std::vector<uint32_t> set;
        rs.gen++;
        rs.left = l; rs.right = r;
        rs.closure(set, revNfa->start_anchored);
        s9[static_cast<int>(nk) * l + r] = rintern(set);
      }
    for (uint32_t cur = 1; cur < rsets.size(); cur++) {
      if (rsets.size() > 4096) { why = "reverse automaton too large"; return false; }
      for (uint32_t c = 0; c < ncls; c++) {
        if (static_cast<int>(c % nk) == kEnd) continue;            // (no reverse step has the end of the text in front of its byte: column unused, dead)
        rs.left = static_cast<int>(c % nk); rs.right = kindOf(c / nk);
        const std::vector<uint32_t> lst = rsets[cur];
        rnext[cur][c] = rintern(rs.step(lst.data(), lst.size(), reps[c / nk]));
      }
    }
    std::vector<uint32_t> order, renum(rsets.size(), 0);
    auto accepting = [&](uint32_t i) { return rs.matchIndex(rsets[i]) >= 0; };
    for (uint32_t i = 0; i < rsets.size(); i++) if (!accepting(i)) { renum[i] = static_cast<uint32_t>(order.size()); order.push_back(i); }
    rFirstAccept = static_cast<uint32_t>(order.size());
    for (uint32_t i = 0; i < rsets.size(); i++) if (accepting(i)) { renum[i] = static_cast<uint32_t>(order.size()); order.push_back(i); }
    rStates = static_cast<uint32_t>(order.size());
    rtab.assign(rStates, std::vector<uint32_t>(ncls, 0u));
    for (uint32_t i = 0; i < rStates; i++) for (uint32_t c = 0; c < ncls; c++) rtab[i][c] = renum[rnext[order[i]][c]];
    for (uint32_t q = 0; q < nk * nk; q++) { rStart9[q] = renum[s9[q]]; if (rStart9[q] >= rFirstAccept) { why = "nullable pattern (empty matches)"; return false; } }
    rStart = rStart9[0];
    if (hasText) {
      // "accepting at the start of the text": the state's set closed once more with the anchor holding, in front of the haystack's
      // first byte (left = outside), once per kind of that byte (the walk has just stepped over it and looks its kind up).
      rTextAcc.assign(static_cast<size_t>(rStates) * nk, 0);
      for (uint32_t i = 0; i < rStates; i++) {
        const std::vector<uint32_t>& set = rsets[order[i]];
        for (int r = 0; r < static_cast<int>(nk); r++) {
          std::vector<uint32_t> closed;
          rs.gen++;
          rs.left = outsideKind; rs.right = r;
          rs.textStart = true;
          for (uint32_t q : set) rs.closure(closed, q);
          rs.textStart = false;
          rTextAcc[static_cast<size_t>(i) * nk + static_cast<size_t>(r)] = rs.matchIndex(closed) >= 0 ? 1 : 0;
        }
      }
    }
  }
  const uint32_t rcols = ncls + (hasText ? nk : 0u);   // hasText: nk more columns, the text-start flags of the state
  // Round 6: reverse rows are a power of two long and an entry's bit 0 says "the target accepts" (rows stay ordered, accepting ones last):
  // a reverse step is  s = rev[(s & ~1) | column]  and the kernel shifts the accept bits of 16 steps into one word (fsm.hpp fsm_match_start16)
  uint32_t revRowP2 = 4;
  while (revRowP2 < rcols * 2) revRowP2 *= 2;
  if (rStates == 0 || static_cast<size_t>(rStates) * revRowP2 > 65535) { why = "reverse DFA missing or too large"; return false; }
  auto offT = [&](uint32_t s) { return s * rowBytes; };
  auto offA = [&](uint32_t a) { return (nT + a) * rowBytes; };
  auto offU = [&](uint32_t u) { return (nT + nA + u) * rowBytes; };
  const uint32_t wideOff = (nT + nA + nU) * rowBytes;
  auto target = [&](uint32_t t) -> uint16_t {      // row offset | create flag | rematch flag << 1
    if (!(t >> 16)) return static_cast<uint16_t>(offT(t & 0xFFFFu));
    const uint32_t kind = (t >> 16) & 3u;
    return static_cast<uint16_t>(offA(aliasOf[t]) | (kind == cxgdev::kFsmEvCreate ? 1u : 0u) | (kind == cxgdev::kFsmEvRematch ? 2u : 0u));
  };

  cxgdev::FsmHeader h;
  std::memset(&h, 0, sizeof h);
  h.magic = cxgdev::kFsmMagic; h.n_t = nT; h.n_a = nA; h.n_u = nU; h.ncls = ncls; h.stride = stride; h.row_bytes = rowBytes; h.depth = depth;
  h.nk = nk; h.outside_byte = outsideByte; h.end_col = hasEnd ? 2u * static_cast<uint32_t>(kEnd) : 0u;
  h.alias_lo = offA(0); h.u_lo = offU(0); h.top_off = offU(0); h.wide_off = wideOff; h.max_len = max_len;
  h.create_lo = offA(nDied); h.rematch_lo = offA(nDied + nCreate); h.row_shift = rowShift;
  std::vector<uint8_t> img(sizeof h, 0);
  auto put = [&](const void* d, size_t n, uint32_t& off) {
    while (img.size() % 16) img.push_back(0);
    off = static_cast<uint32_t>(img.size());
    const uint8_t* q = static_cast<const uint8_t*>(d);
    img.insert(img.end(), q, q + n);
  };
  uint8_t cls2[256], knd[256];
  for (int b = 0; b < 256; b++) { cls2[b] = static_cast<uint8_t>(2 * nk * cls[b]); knd[b] = static_cast<uint8_t>(2 * kindOfByte(b)); }
  std::vector<uint16_t> tab(static_cast<size_t>(nRows) * stride, 0);
  for (uint32_t s = 0; s < nT; s++) {
    for (uint32_t c = 0; c < ncls; c++) tab[static_cast<size_t>(s) * stride + c] = target(trans[s][c]);
    tab[static_cast<size_t>(s) * stride + ncls + 1] = levels[s];
    tab[static_cast<size_t>(s) * stride + ncls + 2] = static_cast<uint16_t>(offT(s));
  }
  for (uint32_t a = 0; a < nA; a++) {
    const uint32_t to = aliases[a] & 0xFFFFu;
    for (uint32_t c = 0; c < ncls; c++) tab[static_cast<size_t>(nT + a) * stride + c] = target(trans[to][c]);
    tab[static_cast<size_t>(nT + a) * stride + ncls] = static_cast<uint16_t>(aliases[a] >> 16);
    tab[static_cast<size_t>(nT + a) * stride + ncls + 1] = levels[to];
    tab[static_cast<size_t>(nT + a) * stride + ncls + 2] = static_cast<uint16_t>(offT(to));
  }
  for (uint32_t u = 0; u < nU; u++)
    for (uint32_t c = 0; c < ncls; c++) {
      const uint16_t t = utrans[u][c];
      tab[static_cast<size_t>(nT + nA + u) * stride + c] = static_cast<uint16_t>(t == kWideMark ? wideOff : (t >= kSetBase ? offU(t - kSetBase) : offT(t)));
    }
  for (uint32_t c = 0; c < ncls; c++) tab[static_cast<size_t>(nT + nA + nU) * stride + c] = static_cast<uint16_t>(wideOff);
  put(tab.data(), tab.size() * 2, h.tab_off);
  if (h.tab_off != sizeof h) { why = "internal: image layout (scan_fsm.hip expects the transition table first)"; return false; }
  put(cls2, 256, h.cls_off);
  std::vector<uint8_t> kndBlock(knd, knd + 256);   // kinds, then the start rows by kind, then the reverse start rows by pair of kinds
  std::vector<uint16_t> mem(static_cast<size_t>(nU + 1) * cxgdev::kFsmMembers, 0xFFFF);
  for (uint32_t u = 0; u < nU; u++)
    if (sets[u].size() <= static_cast<size_t>(cxgdev::kFsmMembers))
      for (size_t k = 0; k < sets[u].size(); k++) mem[static_cast<size_t>(u) * cxgdev::kFsmMembers + k] = static_cast<uint16_t>(offT(sets[u][k]));
  put(mem.data(), mem.size() * 2, h.mem_off);
  // reverse DFA, class-compressed, entries = byte offset of the target row; the classes come from the same NFA ranges,
  // so a class never straddles a reverse transition
  // ... and the offset is counted from the image's first byte behind the header, i.e. it is the row's LDS address in the kernel: a step
  // needs no base added.  The table is aligned to its row size for the `|`; row 0 is the dead state and leads to itself.
  const uint32_t revRow = revRowP2, rstride = revRowP2 / 2;
  while ((img.size() - sizeof h) % revRow) img.push_back(0);
  const uint32_t revBase = static_cast<uint32_t>(img.size() - sizeof h);
  if (revBase + static_cast<size_t>(rStates) * revRow > 65535) { why = "reverse DFA missing or too large"; return false; }
  std::vector<uint16_t> rv(static_cast<size_t>(rStates) * rstride, 0);
  auto rentry = [&](uint32_t to) { return static_cast<uint16_t>((revBase + to * revRow) | (to >= rFirstAccept ? 1u : 0u)); };
  if (hasLook) {
    for (uint32_t s = 0; s < rStates; s++) {
      for (uint32_t c = 0; c < ncls; c++) rv[static_cast<size_t>(s) * rstride + c] = rentry(rtab[s][c]);
      if (hasText) for (uint32_t r = 0; r < nk; r++) rv[static_cast<size_t>(s) * rstride + ncls + r] = rTextAcc[static_cast<size_t>(s) * nk + r];
    }
  } else {
    for (uint32_t s = 0; s < rev.nstates; s++)
      for (uint32_t c = 0; c < ncls; c++) rv[static_cast<size_t>(s) * rstride + c] = rentry(rev.table[static_cast<size_t>(s) * 256 + reps[c]]);
    for (uint32_t s = 0; s < rev.nstates; s++)
      for (int b = 0; b < 256; b++)
        if (rentry(rev.table[static_cast<size_t>(s) * 256 + b]) != rv[static_cast<size_t>(s) * rstride + cls[b]]) { why = "internal: reverse DFA splits a byte class"; return false; }
  }
  put(rv.data(), rv.size() * 2, h.rev_off);
  if (h.rev_off - sizeof h != revBase) { why = "internal: image layout (reverse table)"; return false; }
  h.rev_states = rStates; h.rev_start_off = revBase + rStart * revRow; h.rev_accept_off = revBase + rFirstAccept * revRow; h.rev_row_bytes = revRow;
  h.rev_text_col = hasText ? ncls * 2u : 0u;
  if (hasLook) {
    auto put16 = [&](uint32_t v16) { kndBlock.push_back(static_cast<uint8_t>(v16 & 0xFF)); kndBlock.push_back(static_cast<uint8_t>(v16 >> 8)); };
    for (uint32_t k = 0; k < nk; k++) put16(offT(startOf[k]));
    for (uint32_t q = 0; q < nk * nk; q++) put16(revBase + rStart9[q] * revRow);
    put(kndBlock.data(), kndBlock.size(), h.knd_off);
  } else h.knd_off = h.cls_off;
  while (img.size() % 16) img.push_back(0);
  h.total_bytes = static_cast<uint32_t>(img.size());
  h.lds_bytes = h.total_bytes - static_cast<uint32_t>(sizeof h);
  if (h.lds_bytes > 28672) { why = "FindAll transducer image exceeds the LDS budget"; return false; }
  // ---- Round 6: the byte-indexed tables of the kernel's direct mode (fsm.hpp "Direct mode"), behind the image it replaces in LDS
  static const bool noDirect = getenv("CXG_FSM_NO_DIRECT") != nullptr;   // A/B
  if (!noDirect && depth <= 1 && !hasLook && nT <= 64 && rev.nstates >= 2) {
    // the reverse automaton, minimised (Moore over dead / accepting / other; the README IPv4 pattern: 37 -> 25 states)
    const uint32_t rn = rev.nstates;
    std::vector<uint32_t> rpart(rn);
    for (uint32_t s = 0; s < rn; s++) rpart[s] = s == 0 ? 0u : (s >= rev.firstAccept ? 2u : 1u);
    for (size_t nb = 0;;) {
      std::map<std::vector<uint32_t>, uint32_t> sig;
      std::vector<uint32_t> next(rn);
      for (uint32_t s = 0; s < rn; s++) {
        std::vector<uint32_t> k(1 + nbc);
        k[0] = rpart[s];
        for (uint32_t c = 0; c < nbc; c++) k[1 + c] = rpart[rev.table[static_cast<size_t>(s) * 256 + reps[c]]];
        next[s] = sig.emplace(std::move(k), static_cast<uint32_t>(sig.size())).first->second;   // state 0 (dead) keeps block 0
      }
      rpart.swap(next);
      if (sig.size() == nb) break;
      nb = sig.size();
    }
    uint32_t rblocks = 0;
    for (uint32_t s = 0; s < rn; s++) rblocks = std::max(rblocks, rpart[s] + 1);
    std::vector<uint32_t> rrep(rblocks, 0xFFFFFFFFu);
    for (uint32_t s = 0; s < rn; s++) if (rrep[rpart[s]] == 0xFFFFFFFFu) rrep[rpart[s]] = s;
    // slots: state k at 4k, its create / rematch copies at 4k + 1 / 4k + 2 where some step enters them; everything else is free for the
    // reverse rows (accepting ones above the others) and the sets of possible entry states
    std::vector<int> used(256, 0);
    auto slotOfTrans = [&](uint32_t t) -> uint32_t {
      const uint32_t kind = (t >> 16) & 3u, to = t & 0xFFFFu;
      return 4u * to + (kind == cxgdev::kFsmEvCreate ? 1u : (kind == cxgdev::kFsmEvRematch ? 2u : 0u));
    };
    for (uint32_t s = 0; s < nT; s++) { used[4 * s] = 1; for (uint32_t c = 0; c < ncls; c++) used[slotOfTrans(trans[s][c])] = 1; }
    const std::vector<int> fwdUsed = used;
    std::vector<uint32_t> rslot(rblocks, 0u), uslot(nU + 1, 0u);
    uint32_t nslots = 4u * nT, cursor = 1;
    bool fits = true;
    auto takeFree = [&](uint32_t from) -> uint32_t { uint32_t q = from; while (q < 256 && used[q]) q++; return q; };
    for (int pass = 0; pass < 2 && fits; pass++)                   // (block 0 = dead has a row of its own that leads to itself: the branch-free walk needs it to absorb)
      for (uint32_t b = 0; b < rblocks && fits; b++) {
        const bool acc = rrep[b] >= rev.firstAccept;
        if (acc != (pass == 1)) continue;
        const uint32_t q = takeFree(cursor);
        if (q >= 256) { fits = false; break; }
        used[q] = 1; rslot[b] = q; cursor = q + 1;               // ascending: every accepting slot lies above every other reverse slot
        nslots = std::max(nslots, q + 1);
      }
    for (uint32_t u = 0; u <= nU && fits; u++) {                  // (u == nU: the wide row)
      const uint32_t q = takeFree(1);
      if (q >= 256) { fits = false; break; }
      used[q] = 1; uslot[u] = q;
      nslots = std::max(nslots, q + 1);
    }
    uint32_t raccLo = 256;
    for (uint32_t b = 1; b < rblocks; b++) if (rrep[b] >= rev.firstAccept) raccLo = std::min(raccLo, rslot[b]);
    const uint32_t dbytes = nslots * 256u + 256u;
    if (fits && raccLo < 256 && rpart[rev.start] != 0 && dbytes <= cxgdev::kFsmdMaxBytes) {
      std::vector<uint8_t> d(dbytes, 0);
      uint8_t* prop = d.data() + static_cast<size_t>(nslots) * 256;   // per slot: 0x80 a set (or the wide row), else the pending levels of the state
      for (uint32_t s = 0; s < nT; s++)
        for (uint32_t f = 0; f < 3; f++) {
          if (!fwdUsed[4 * s + f]) continue;                       // (a create / rematch copy nobody enters is a free slot)
          for (int b = 0; b < 256; b++) d[static_cast<size_t>(4 * s + f) * 256 + b] = static_cast<uint8_t>(slotOfTrans(trans[s][cls[b]]));
          prop[4 * s + f] = levels[s];
        }
      for (uint32_t u = 0; u < nU; u++) {
        for (int b = 0; b < 256; b++) {
          const uint16_t t = utrans[u][cls[b]];
          d[static_cast<size_t>(uslot[u]) * 256 + b] = static_cast<uint8_t>(t == kWideMark ? uslot[nU] : (t >= kSetBase ? uslot[t - kSetBase] : 4u * t));
        }
        prop[uslot[u]] = 0x80;
      }
      for (int b = 0; b < 256; b++) d[static_cast<size_t>(uslot[nU]) * 256 + b] = static_cast<uint8_t>(uslot[nU]);
      prop[uslot[nU]] = 0x80;
      for (uint32_t b = 0; b < rblocks; b++)
        for (int x = 0; x < 256; x++) d[static_cast<size_t>(rslot[b]) * 256 + x] = static_cast<uint8_t>(rslot[rpart[rev.table[static_cast<size_t>(rrep[b]) * 256 + x]]]);
      put(d.data(), d.size(), h.direct_off);
      h.direct_bytes = dbytes; h.d_slots = nslots; h.d_racc_lo = raccLo; h.d_rstart = rslot[rpart[rev.start]]; h.d_top = uslot[0]; h.d_rdead = rslot[0];
      while (img.size() % 16) img.push_back(0);
      h.total_bytes = static_cast<uint32_t>(img.size());           // (lds_bytes stays the class-indexed image's: the two are staged alternatively)
    }
  }
  std::memcpy(img.data(), &h, sizeof h);
  image.swap(img);
  return true;
}
}  // namespace

}  // namespace cxg
