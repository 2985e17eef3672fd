// host/fsm.cc — builds the FindAll transducer image (device/fsm.hpp) from an NFA.
//
// Semantics restated from the reference, as program.cc determinize does: ordered epsilon closure
// (dfa/lazy/builder.go:245-293: stack, push right then left), byte-range / sparse move in list order
// (builder.go:215-230), break at the first Match of a list (builder.go:210-213, forward searches only,
// meta/compile.go:193).  What is new is that the FindAll loop around the search (meta/findall.go:216-239: take the
// last accepting position when the DFA dies, restart there) is determinized too — a state is a STACK of thread lists,
// see fsm.hpp — so the machine runs left to right without ever moving back.
#include "fsm.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <set>

namespace cxg {

namespace {

constexpr uint32_t kSep = 0xFFFFFFFEu;     // separates the levels of a stack in its key vector

struct Stepper {
  const cxg_nfa& n;
  std::vector<uint32_t> mark;
  uint32_t gen = 0;
  std::vector<uint32_t> stack;
  explicit Stepper(const cxg_nfa& nfa) : n(nfa), mark(nfa.n_states, 0) {}
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

bool buildFsmImage(const cxg_nfa& nfa, const Dfa& rev, uint32_t max_len, std::vector<uint8_t>& image, std::string& why) {
  image.clear();
  for (uint32_t i = 0; i < nfa.n_states; i++)
    if (nfa.states[i].kind == CXG_NFA_LOOK) { why = "look-around assertion in NFA"; return false; }
  if (nfa.start_unanchored == nfa.start_anchored) { why = "start-anchored pattern"; return false; }
  // byte classes (nfa/alphabet.go:100-166)
  bool boundary[256] = {false};
  auto markb = [&](int lo, int hi) { if (lo > 0) boundary[lo - 1] = true; boundary[hi] = true; };
  for (uint32_t i = 0; i < nfa.n_states; i++) {
    const cxg_nfa_state& s = nfa.states[i];
    if (s.kind == CXG_NFA_BYTE_RANGE) markb(s.lo, s.hi);
    else if (s.kind == CXG_NFA_SPARSE) for (uint32_t k = 0; k < s.trans_len; k++) markb(nfa.trans[s.trans_off + k].lo, nfa.trans[s.trans_off + k].hi);
  }
  std::vector<int> reps;
  uint8_t cls[256];
  for (int b = 0; b < 256; b++) { if (b == 0 || boundary[b - 1]) reps.push_back(b); cls[b] = static_cast<uint8_t>(reps.size() - 1); }
  const uint32_t ncls = static_cast<uint32_t>(reps.size());
  if (ncls > 64) { why = "more than 64 byte classes"; return false; }

  Stepper st(nfa);
  std::vector<uint32_t> fresh;
  st.gen++;
  st.closure(fresh, nfa.start_unanchored);
  if (st.matchIndex(fresh) >= 0) { why = "nullable pattern (empty matches)"; return false; }

  // ---- transducer states: stacks of thread lists
  std::map<std::vector<uint32_t>, uint32_t> ids;
  std::vector<std::vector<uint32_t>> keys;
  std::vector<uint8_t> levels;                    // pending levels of a state
  std::vector<std::vector<uint16_t>> trans;       // [state][class] next | event << 8
  std::vector<uint16_t> events{0};                // descriptor table; id 0 = none
  std::map<uint16_t, uint32_t> eventId;
  uint32_t depth = 0;
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
  auto eventOf = [&](uint32_t kind, uint32_t j, bool conts, uint32_t died) -> uint32_t {
    const uint16_t d = static_cast<uint16_t>(kind | (j << 2) | (conts ? 32u : 0u) | (died << 8));
    if (d == 0) return 0;
    auto it = eventId.find(d);
    if (it != eventId.end()) return it->second;
    if (events.size() >= 255) { tooBig = true; return 0; }
    events.push_back(d);
    eventId.emplace(d, static_cast<uint32_t>(events.size() - 1));
    return static_cast<uint32_t>(events.size() - 1);
  };
  intern({fresh});
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
      const int b = reps[c];
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
        } else {
          if (moved.empty()) { why = "internal: innermost search died (unanchored prefix missing)"; return false; }
          next.push_back(std::move(moved));
          ev = eventOf(cxgdev::kFsmEvDied, 0, false, died);
        }
      }
      const uint32_t to = intern(next);
      trans[cur][c] = static_cast<uint16_t>(to | (ev << 8));
    }
  }
  if (tooBig) { why = "FindAll transducer exceeds the table budget (states, pending levels or events)"; return false; }
  const uint32_t nT = static_cast<uint32_t>(keys.size());

  // ---- uncertainty rows: sets of states, from "any state" (top) until they collapse to one state
  std::map<std::vector<uint8_t>, uint32_t> setId;
  std::vector<std::vector<uint8_t>> sets;
  std::vector<std::vector<uint16_t>> utrans;
  bool wideUsed = false;
  const uint32_t rowBudget = cxgdev::kFsmMaxRows - 1;          // one id kept for the wide row
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
      for (uint8_t s : sets[cur]) img.insert(static_cast<uint8_t>(trans[s][c] & 0xFFu));
      if (img.size() == 1) { utrans[cur][c] = *img.begin(); continue; }
      std::vector<uint8_t> v(img.begin(), img.end());
      auto it = setId.find(v);
      if (it == setId.end()) {
        if (nT + sets.size() >= rowBudget) { utrans[cur][c] = kWideMark; wideUsed = true; continue; }
        it = setId.emplace(v, static_cast<uint32_t>(sets.size())).first;
        sets.push_back(v);
      }
      utrans[cur][c] = static_cast<uint16_t>(kSetBase + it->second);
    }
  }
  const uint32_t nU = static_cast<uint32_t>(sets.size());
  const uint32_t wideRow = nT + nU;
  const uint32_t nRows = nT + nU + 1;                          // the wide row always exists (simplifies the kernel)
  (void)wideUsed;
  if (nRows > cxgdev::kFsmMaxRows + 1u) { why = "FindAll transducer exceeds 255 table rows"; return false; }
  const uint32_t stride = ncls | 1u;
  if (static_cast<size_t>(nRows) * stride * 2 > cxgdev::kFsmMaxTableBytes) { why = "FindAll transducer table exceeds the LDS budget"; return false; }
  if (rev.nstates == 0 || rev.nstates > 255) { why = "reverse DFA missing"; return false; }

  // ---- image
  cxgdev::FsmHeader h;
  std::memset(&h, 0, sizeof h);
  h.magic = cxgdev::kFsmMagic; h.n_t = nT; h.n_rows = nRows; h.ncls = ncls; h.top_row = nT; h.wide_row = wideRow;
  h.n_events = static_cast<uint32_t>(events.size()); h.depth = depth; h.stride = stride; h.max_len = max_len;
  std::vector<uint8_t> img(sizeof h, 0);
  auto put = [&](const void* d, size_t n, uint32_t& off) {
    while (img.size() % 16) img.push_back(0);
    off = static_cast<uint32_t>(img.size());
    const uint8_t* q = static_cast<const uint8_t*>(d);
    img.insert(img.end(), q, q + n);
  };
  put(cls, 256, h.cls_off);
  std::vector<uint16_t> tab(static_cast<size_t>(nRows) * stride, 0);
  for (uint32_t s = 0; s < nT; s++) for (uint32_t c = 0; c < ncls; c++) tab[static_cast<size_t>(s) * stride + c] = trans[s][c];
  for (uint32_t u = 0; u < nU; u++)
    for (uint32_t c = 0; c < ncls; c++) {
      const uint16_t t = utrans[u][c];
      tab[static_cast<size_t>(nT + u) * stride + c] = static_cast<uint16_t>(t == kWideMark ? wideRow : (t >= kSetBase ? nT + (t - kSetBase) : t));
    }
  for (uint32_t c = 0; c < stride; c++) tab[static_cast<size_t>(wideRow) * stride + c] = static_cast<uint16_t>(wideRow);
  put(tab.data(), tab.size() * 2, h.tab_off);
  put(events.data(), events.size() * 2, h.ev_off);
  put(levels.data(), levels.size(), h.lev_off);
  std::vector<uint8_t> mem(static_cast<size_t>(nRows) * cxgdev::kFsmMembers, 0xFF);
  for (uint32_t s = 0; s < nT; s++) mem[static_cast<size_t>(s) * cxgdev::kFsmMembers] = static_cast<uint8_t>(s);
  for (uint32_t u = 0; u < nU; u++)
    if (sets[u].size() <= static_cast<size_t>(cxgdev::kFsmMembers))
      for (size_t k = 0; k < sets[u].size(); k++) mem[static_cast<size_t>(nT + u) * cxgdev::kFsmMembers + k] = sets[u][k];
  put(mem.data(), mem.size(), h.mem_off);
  // reverse DFA, class-compressed: the classes come from the same NFA ranges, so a class never straddles a reverse transition
  std::vector<uint8_t> rv(static_cast<size_t>(rev.nstates) * ncls, 0);
  for (uint32_t s = 0; s < rev.nstates; s++)
    for (uint32_t c = 0; c < ncls; c++) rv[static_cast<size_t>(s) * ncls + c] = rev.table[static_cast<size_t>(s) * 256 + reps[c]];
  for (uint32_t s = 0; s < rev.nstates; s++)
    for (int b = 0; b < 256; b++)
      if (rev.table[static_cast<size_t>(s) * 256 + b] != rv[static_cast<size_t>(s) * ncls + cls[b]]) { why = "internal: reverse DFA splits a byte class"; return false; }
  put(rv.data(), rv.size(), h.rev_off);
  h.rev_states = rev.nstates; h.rev_start = rev.start; h.rev_first_accept = rev.firstAccept;
  while (img.size() % 16) img.push_back(0);
  h.total_bytes = static_cast<uint32_t>(img.size());
  h.lds_bytes = h.total_bytes - static_cast<uint32_t>(sizeof h);
  std::memcpy(img.data(), &h, sizeof h);
  image.swap(img);
  return true;
}

}  // namespace cxg
