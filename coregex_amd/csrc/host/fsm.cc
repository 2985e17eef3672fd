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
  std::vector<std::vector<uint32_t>> trans;       // [state][class] next | event descriptor << 16
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
  auto eventOf = [&](uint32_t kind, uint32_t j, bool conts, uint32_t died) -> uint32_t {   // descriptor, 0 = nothing happened
    return kind | (j << 2) | (conts ? 32u : 0u) | (died << 8);
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
      trans[cur][c] = to | (ev << 16);
    }
  }
  if (tooBig) { why = "FindAll transducer exceeds the table budget (states, pending levels or events)"; return false; }
  const uint32_t nT = static_cast<uint32_t>(keys.size());

  // ---- uncertainty rows: sets of states, from "any state" (top) until they collapse to one state
  std::map<std::vector<uint8_t>, uint32_t> setId;
  std::vector<std::vector<uint8_t>> sets;
  std::vector<std::vector<uint16_t>> utrans;
  bool wideUsed = false;
  const uint32_t rowBudget = 512;                              // sets tabulated at most; the rest maps to the wide row
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
  if (rev.nstates == 0 || static_cast<size_t>(rev.nstates) * ncls * 2 > 65535) { why = "reverse DFA missing or too large"; return false; }
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
  h.alias_lo = offA(0); h.u_lo = offU(0); h.top_off = offU(0); h.wide_off = wideOff; h.max_len = max_len;
  h.create_lo = offA(nDied); h.rematch_lo = offA(nDied + nCreate); h.row_shift = rowShift;
  std::vector<uint8_t> img(sizeof h, 0);
  auto put = [&](const void* d, size_t n, uint32_t& off) {
    while (img.size() % 16) img.push_back(0);
    off = static_cast<uint32_t>(img.size());
    const uint8_t* q = static_cast<const uint8_t*>(d);
    img.insert(img.end(), q, q + n);
  };
  uint8_t cls2[256];
  for (int b = 0; b < 256; b++) cls2[b] = static_cast<uint8_t>(2 * cls[b]);
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
  std::vector<uint16_t> mem(static_cast<size_t>(nU + 1) * cxgdev::kFsmMembers, 0xFFFF);
  for (uint32_t u = 0; u < nU; u++)
    if (sets[u].size() <= static_cast<size_t>(cxgdev::kFsmMembers))
      for (size_t k = 0; k < sets[u].size(); k++) mem[static_cast<size_t>(u) * cxgdev::kFsmMembers + k] = static_cast<uint16_t>(offT(sets[u][k]));
  put(mem.data(), mem.size() * 2, h.mem_off);
  // reverse DFA, class-compressed, entries = byte offset of the target row; the classes come from the same NFA ranges,
  // so a class never straddles a reverse transition
  const uint32_t revRow = ncls * 2;
  std::vector<uint16_t> rv(static_cast<size_t>(rev.nstates) * ncls, 0);
  for (uint32_t s = 0; s < rev.nstates; s++)
    for (uint32_t c = 0; c < ncls; c++) rv[static_cast<size_t>(s) * ncls + c] = static_cast<uint16_t>(rev.table[static_cast<size_t>(s) * 256 + reps[c]] * revRow);
  for (uint32_t s = 0; s < rev.nstates; s++)
    for (int b = 0; b < 256; b++)
      if (rev.table[static_cast<size_t>(s) * 256 + b] * revRow != rv[static_cast<size_t>(s) * ncls + cls[b]]) { why = "internal: reverse DFA splits a byte class"; return false; }
  put(rv.data(), rv.size() * 2, h.rev_off);
  h.rev_states = rev.nstates; h.rev_start_off = rev.start * revRow; h.rev_accept_off = rev.firstAccept * revRow; h.rev_row_bytes = revRow;
  while (img.size() % 16) img.push_back(0);
  h.total_bytes = static_cast<uint32_t>(img.size());
  h.lds_bytes = h.total_bytes - static_cast<uint32_t>(sizeof h);
  if (h.lds_bytes > 28672) { why = "FindAll transducer image exceeds the LDS budget"; return false; }
  std::memcpy(img.data(), &h, sizeof h);
  image.swap(img);
  return true;
}

}  // namespace cxg
