// Look-around inside the reference's lazy-DFA strategies (UseDFA / UseBoth over an NFA with \b \B (?m)^ (?m)$).
//
// The reference answers such programs with its look-aware lazy DFA (dfa/lazy/lazy.go:1102-1315 searchAt, :1336-1446
// determinize, builder.go:183-242 moveWithWordContextBreak, :295-425 resolveWordBoundaries, :437-472 CheckEOIMatch,
// start.go:64-254) and, for UseDFA, a reverse DFA over the reverse NFA in which every assertion became an epsilon edge
// (nfa/reverse.go:124-129, meta/compile.go:184-205).  That machine is NOT leftmost-first in general:
//   * a transition is cached per byte class (lazy.go:1341,1397) although its target depends on the byte being a word byte
//     or '\n' — unless every class is pure in that respect the answer depends on what the cache saw before;
//   * resolveWordBoundaries returns the set and the states behind a crossed \b / \B in SORTED order (state.go:487-497);
//   * searchAt returns at the first byte at which checkWordBoundaryMatch holds (lazy.go:1262-1264, :1533-1560) — true
//     whenever the state's set holds a match state and the state is not match-tagged;
//   * the reverse DFA ignores the assertions.
// The device serves leftmost-first (the transducer of fsm.cc).  This file decides, at build time, whether the two agree on
// EVERY haystack: it builds the reference's forward machine R state by state (un-conflated: ordered NFA list, from-word flag,
// delayed-match flag) and the leftmost-first machine T (ordered list, kind of the byte behind; every assertion resolved when
// the next byte is known), and walks their product; then the same for the two reverse machines (UseDFA), or only "R's end is
// never behind T's" (UseBoth, whose answer comes from the PikeVM), or both machines from the anchored start (UseDigitPrefilter).
// Any difference that can show in an answer, any state in which bytes of one class lead to different behaviour, any pair of
// priority orders filed under one cache key that behave differently: CXG_E_UNSUPPORTED, the caller keeps its CPU loop.
// Sound by construction (only "equal on all inputs" passes); scripts/cpu_fuzz_lookdfa.py checks it against the restated
// reference DFA of the oracle (test infrastructure) through the transducer's sequential twin.
#include "lookdfa.h"

#include <algorithm>
#include <array>
#include <map>
#include <set>

namespace cxg {
namespace {

enum : uint8_t { kLkStartText = 0, kLkEndText = 1, kLkStartLine = 2, kLkEndLine = 3, kLkWordB = 4, kLkNoWordB = 5 };
enum : int { kSymEnd = -1 };   // end of input (forward) / begin of input (reverse)
constexpr size_t kMaxStates = 4096;

bool isWord(int b) { return (b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z') || b == '_' || (b >= 'a' && b <= 'z'); }

struct Machine {
  const cxg_nfa& n;
  std::vector<uint32_t> mark;
  uint32_t gen = 0;
  std::vector<uint32_t> stack;
  explicit Machine(const cxg_nfa& nfa) : n(nfa), mark(nfa.n_states, 0) {}
  void begin() { gen++; }
  // epsilonClosureInto (builder.go:245-293): add-on-pop, right pushed before left; `pass(look)` says which assertions hold
  template <class Pass>
  void into(std::vector<uint32_t>& out, uint32_t seed, Pass pass) {
    stack.clear();
    stack.push_back(seed);
    while (!stack.empty()) {
      const uint32_t cur = stack.back();
      stack.pop_back();
      if (cur == CXG_NFA_INVALID || cur >= n.n_states || mark[cur] == gen) continue;
      mark[cur] = gen;
      out.push_back(cur);
      const cxg_nfa_state& s = n.states[cur];
      switch (s.kind) {
        case CXG_NFA_EPSILON: case CXG_NFA_CAPTURE: if (s.next != CXG_NFA_INVALID) stack.push_back(s.next); break;
        case CXG_NFA_SPLIT:
          if (s.right != CXG_NFA_INVALID) stack.push_back(s.right);
          if (s.left != CXG_NFA_INVALID) stack.push_back(s.left);
          break;
        case CXG_NFA_LOOK: if (s.next != CXG_NFA_INVALID && pass(s.lo)) stack.push_back(s.next); break;
        default: break;
      }
    }
  }
  bool holdsMatch(const std::vector<uint32_t>& v, size_t cnt) const {
    for (size_t i = 0; i < cnt; i++) if (n.states[v[i]].kind == CXG_NFA_MATCH) return true;
    return false;
  }
  // byte transitions of `v[0..cnt)` on b, targets closed in order; `brk`: stop at the first match state
  template <class Pass>
  void move(const std::vector<uint32_t>& v, size_t cnt, int b, bool brk, std::vector<uint32_t>& out, Pass pass) {
    begin();
    for (size_t i = 0; i < cnt; i++) {
      const cxg_nfa_state& s = n.states[v[i]];
      if (brk && s.kind == CXG_NFA_MATCH) break;
      if (s.kind == CXG_NFA_BYTE_RANGE) {
        if (b >= s.lo && b <= s.hi) into(out, s.next, pass);
      } else if (s.kind == CXG_NFA_SPARSE) {
        for (uint32_t k = 0; k < s.trans_len; k++) {
          const cxg_nfa_trans& t = n.trans[s.trans_off + k];
          if (b >= t.lo && b <= t.hi) into(out, t.next, pass);
        }
      }
    }
  }
};

// resolveWordBoundaries, builder.go:295-425
std::vector<uint32_t> resolveWordBoundaries(const cxg_nfa& n, const std::vector<uint32_t>& set, size_t cnt, bool satisfied) {
  std::vector<uint8_t> crossed(n.n_states, 0);
  std::vector<uint32_t> stack;
  auto cross = [&](const cxg_nfa_state& st) {
    if (st.next == CXG_NFA_INVALID || st.next >= n.n_states) return;
    const bool ok = (st.lo == kLkWordB && satisfied) || (st.lo == kLkNoWordB && !satisfied);
    if (ok && !crossed[st.next]) { crossed[st.next] = 1; stack.push_back(st.next); }
  };
  auto follow = [&](uint32_t t) { if (t != CXG_NFA_INVALID && t < n.n_states && !crossed[t]) { crossed[t] = 1; stack.push_back(t); } };
  for (size_t i = 0; i < cnt; i++) if (n.states[set[i]].kind == CXG_NFA_LOOK) cross(n.states[set[i]]);
  if (stack.empty()) return std::vector<uint32_t>(set.begin(), set.begin() + static_cast<long>(cnt));
  while (!stack.empty()) {
    const uint32_t cur = stack.back();
    stack.pop_back();
    const cxg_nfa_state& st = n.states[cur];
    switch (st.kind) {
      case CXG_NFA_LOOK: cross(st); break;
      case CXG_NFA_EPSILON: case CXG_NFA_CAPTURE: follow(st.next); break;
      case CXG_NFA_SPLIT: follow(st.left); follow(st.right); break;
      default: break;
    }
  }
  for (size_t i = 0; i < cnt; i++) crossed[set[i]] = 1;
  std::vector<uint32_t> out;
  for (uint32_t i = 0; i < n.n_states; i++) if (crossed[i]) out.push_back(i);
  return out;
}

struct Refuse { std::string why; };

// ---------------------------------------------------------------------------------------------------------- forward
// Common shape of both forward machines: per state and symbol (class representative, or kSymEnd) a flag "a match ends
// in front of this symbol" and a successor (-1: the search is over).
struct Automaton {
  std::vector<std::vector<int32_t>> next;   // [state][symbol index]; the kSymEnd column is not stored
  std::vector<std::vector<uint8_t>> flag;   // [state][symbol index], last column = kSymEnd
  std::vector<std::vector<uint8_t>> early;  // R only: the flag is searchAt's early return (decided on the byte itself, never cached)
  std::vector<uint8_t> live;                // some symbol sequence from here raises a flag
  int32_t start[4] = {-1, -1, -1, -1};      // by the byte behind the search start: non-word, word, '\n', none (text start)
  void computeLive() {
    const size_t ns = next.size();
    live.assign(ns, 0);
    for (size_t s = 0; s < ns; s++) for (uint8_t f : flag[s]) if (f) live[s] = 1;
    for (bool changed = true; changed;) {
      changed = false;
      for (size_t s = 0; s < ns; s++)
        if (!live[s]) for (int32_t t : next[s]) if (t >= 0 && live[static_cast<size_t>(t)]) { live[s] = 1; changed = true; break; }
    }
  }
};

enum StartKind { kAfterNonWord = 0, kAfterWord = 1, kAfterNewline = 2, kAtTextStart = 3 };

// R: the reference's lazy DFA + searchAt, as a finite machine (classes are kind-pure here, so a representative decides).
Automaton buildReference(const cxg_nfa& n, uint32_t startState, const std::vector<int>& reps, const std::vector<int>& classOf, bool hasWordB, bool hasEndLine) {
  Machine m(n);
  Automaton a;
  std::map<std::vector<uint32_t>, int32_t> ids;       // ordered list + flags word
  std::map<std::vector<uint32_t>, std::vector<int32_t>> filedAs;   // sorted set + flags word (state.go:329-373) -> orders filed there
  std::vector<std::vector<uint32_t>> tuples;
  auto intern = [&](std::vector<uint32_t>&& list, uint32_t flagsWord) -> int32_t {
    std::vector<uint32_t> key(list);
    std::sort(key.begin(), key.end());
    key.push_back(0x80000000u | flagsWord);
    list.push_back(0x80000000u | flagsWord);
    auto it = ids.find(list);
    if (it != ids.end()) return it->second;
    const int32_t id = static_cast<int32_t>(tuples.size());
    if (tuples.size() >= kMaxStates) throw Refuse{"look-aware reference DFA exceeds the build-time exploration budget"};
    std::vector<int32_t>& filed = filedAs[key];
    filed.push_back(id);
    ids.emplace(list, id);
    tuples.push_back(std::move(list));
    return id;
  };
  const uint32_t have[4] = {0u, 0u, 1u << kLkStartLine, (1u << kLkStartLine) | (1u << kLkStartText)};   // look.go:88-107
  for (int k = 0; k < 4; k++) {
    std::vector<uint32_t> set;
    m.begin();
    m.into(set, startState, [&](uint8_t l) { return (have[k] >> l) & 1u; });
    // the four kinds may share a state (same set, same from-word flag): GetOrInsert, lazy.go:1592-1601
    std::vector<uint32_t> probe(set);
    probe.push_back(0x80000000u | (k == kAfterWord ? 1u : 0u));
    auto it = ids.find(probe);
    a.start[k] = it != ids.end() ? it->second : intern(std::move(set), k == kAfterWord ? 1u : 0u);
  }
  for (size_t cur = 0; cur < tuples.size(); cur++) {
    const std::vector<uint32_t> src = tuples[cur];   // copy: tuples grows
    const size_t cnt = src.size() - 1;
    const bool fromWord = src[cnt] & 1u, tagged = src[cnt] & 2u;
    std::vector<int32_t> nx(reps.size(), -1);
    std::vector<uint8_t> fl(reps.size() + 1, 0), er(reps.size(), 0);
    for (size_t ri = 0; ri < reps.size(); ri++) {
      const int b = reps[ri];
      // checkWordBoundaryMatch (lazy.go:1533-1560): the search returns here
      if (hasWordB && !tagged) {
        const std::vector<uint32_t> atB = resolveWordBoundaries(n, src, cnt, fromWord != isWord(b));
        if (m.holdsMatch(atB, atB.size())) { fl[ri] = 1; er[ri] = 1; continue; }
      }
      // determinize (lazy.go:1336-1446)
      std::vector<uint32_t> curSet(src.begin(), src.begin() + static_cast<long>(cnt));
      if (hasEndLine && b == '\n') {
        std::vector<uint32_t> re;
        m.begin();
        for (uint32_t s : curSet) m.into(re, s, [](uint8_t l) { return l == kLkEndLine; });
        curSet.swap(re);
      }
      const bool srcMatch = m.holdsMatch(curSet, curSet.size());
      const std::vector<uint32_t> resolved = hasWordB ? resolveWordBoundaries(n, curSet, curSet.size(), fromWord != isWord(b)) : curSet;
      std::vector<uint32_t> out;
      const bool nl = b == '\n';
      m.move(resolved, resolved.size(), b, srcMatch, out, [nl](uint8_t l) { return nl && l == kLkStartLine; });
      if (out.empty() && !srcMatch) continue;   // dead
      nx[ri] = intern(std::move(out), (isWord(b) ? 1u : 0u) | (srcMatch ? 2u : 0u));
      fl[ri] = srcMatch ? 1 : 0;                  // the successor is match-tagged: lastMatch = index of b
    }
    {  // CheckEOIMatch (builder.go:437-472)
      const std::vector<uint32_t> resolved = resolveWordBoundaries(n, src, cnt, fromWord);
      std::vector<uint32_t> fin;
      m.begin();
      for (uint32_t s : resolved) m.into(fin, s, [](uint8_t l) { return l == kLkEndText || l == kLkEndLine; });
      fl[reps.size()] = m.holdsMatch(fin, fin.size()) ? 1 : 0;
    }
    a.next.push_back(std::move(nx));
    a.flag.push_back(std::move(fl));
    a.early.push_back(std::move(er));
  }
  a.computeLive();
  {
    // What the cache can do to this machine.  (1) It keeps whichever ORDER of a set it determinized first (state.go:329-373).
    // (2) It keeps one successor per byte CLASS (lazy.go:1341,1397) although the successor depends on the byte — is it a word
    // byte, is it '\n' — and the classes are not refined by that: the byte that reached determinize first decides for its
    // class.  (The early return is decided on the byte itself before the lookup, so bytes that return early never fill or read
    // the entry.)  Both are harmless exactly when the alternatives behave alike — same flags on every symbol sequence: Moore
    // partition of the exact, un-conflated machine built above (states that can never flag again count as dead), as
    // program.cc priorityOrderConflict does for programs without assertions.
    const size_t ns = a.next.size();
    std::vector<uint32_t> cls(ns);
    {
      std::map<std::vector<uint8_t>, uint32_t> byFlags;
      for (size_t i = 0; i < ns; i++) cls[i] = byFlags.emplace(a.flag[i], static_cast<uint32_t>(byFlags.size())).first->second;
    }
    for (size_t nclasses = 0;;) {
      std::map<std::vector<uint32_t>, uint32_t> sig;
      std::vector<uint32_t> ncls(ns);
      for (size_t i = 0; i < ns; i++) {
        std::vector<uint32_t> k{cls[i]};
        for (int32_t t : a.next[i]) k.push_back(t < 0 || !a.live[static_cast<size_t>(t)] ? 0xFFFFFFFFu : cls[static_cast<size_t>(t)]);
        ncls[i] = sig.emplace(std::move(k), static_cast<uint32_t>(sig.size())).first->second;
      }
      cls.swap(ncls);
      if (sig.size() == nclasses) break;
      nclasses = sig.size();
    }
    for (const auto& kv : filedAs)
      for (int32_t id : kv.second)
        if (cls[static_cast<size_t>(id)] != cls[static_cast<size_t>(kv.second[0])])
          throw Refuse{"reference DFA cache conflates priority orders of one NFA set (result depends on cache history)"};
    auto target = [&](size_t st, size_t sym) -> uint64_t {
      const int32_t t = a.next[st][sym];
      return (static_cast<uint64_t>(a.flag[st][sym]) << 32) | (t < 0 || !a.live[static_cast<size_t>(t)] ? 0xFFFFFFFFu : cls[static_cast<size_t>(t)]);
    };
    for (size_t st = 0; st < ns; st++) {
      if (!a.live[st]) continue;
      for (size_t i = 0; i < reps.size(); i++) {
        if (a.early[st][i]) continue;
        for (size_t j = i + 1; j < reps.size() && classOf[j] == classOf[i]; j++)
          if (!a.early[st][j] && target(st, i) != target(st, j))
            throw Refuse{"the reference caches lazy-DFA transitions per byte class, and bytes of one class (word / non-word / newline) lead to different states in this program: its answer depends on cache history"};
      }
    }
  }
  return a;
}

// T: leftmost-first.  A state is the ordered thread list at a position (assertions not yet passed) plus the kind of the byte
// behind the position; the symbol ahead completes the context, then every assertion is decided, a match state in the
// expanded list ends a match in front of the symbol and cuts the lower-priority threads, and the rest moves on.
Automaton buildLeftmostFirst(const cxg_nfa& n, uint32_t startState, const std::vector<int>& reps) {
  Machine m(n);
  Automaton a;
  std::map<std::vector<uint32_t>, int32_t> ids;
  std::vector<std::vector<uint32_t>> tuples;   // list, then the kind behind
  auto intern = [&](std::vector<uint32_t>&& list, uint32_t behind) -> int32_t {
    list.push_back(0x80000000u | behind);
    auto it = ids.find(list);
    if (it != ids.end()) return it->second;
    if (tuples.size() >= kMaxStates) throw Refuse{"leftmost-first automaton exceeds the build-time exploration budget"};
    const int32_t id = static_cast<int32_t>(tuples.size());
    ids.emplace(list, id);
    tuples.push_back(std::move(list));
    return id;
  };
  auto none = [](uint8_t) { return false; };
  for (int k = 0; k < 4; k++) {
    std::vector<uint32_t> set;
    m.begin();
    m.into(set, startState, none);
    a.start[k] = intern(std::move(set), static_cast<uint32_t>(k));
  }
  for (size_t cur = 0; cur < tuples.size(); cur++) {
    const std::vector<uint32_t> src = tuples[cur];
    const size_t cnt = src.size() - 1;
    const uint32_t behind = src[cnt] & 3u;
    std::vector<int32_t> nx(reps.size(), -1);
    std::vector<uint8_t> fl(reps.size() + 1, 0);
    for (size_t ri = 0; ri <= reps.size(); ri++) {
      const int b = ri < reps.size() ? reps[ri] : kSymEnd;
      const bool wordAhead = b >= 0 && isWord(b), wordBehind = behind == kAfterWord;
      const bool endLine = b < 0 || b == '\n', startLine = behind == kAfterNewline || behind == kAtTextStart;
      auto pass = [&](uint8_t l) {
        switch (l) {
          case kLkStartText: return behind == kAtTextStart;
          case kLkEndText: return b < 0;
          case kLkStartLine: return startLine;
          case kLkEndLine: return endLine;
          case kLkWordB: return wordAhead != wordBehind;
          case kLkNoWordB: return wordAhead == wordBehind;
          default: return false;
        }
      };
      std::vector<uint32_t> full;
      m.begin();
      for (size_t i = 0; i < cnt; i++) m.into(full, src[i], pass);
      size_t keep = full.size();
      for (size_t i = 0; i < full.size(); i++) if (n.states[full[i]].kind == CXG_NFA_MATCH) { keep = i; fl[ri] = 1; break; }
      if (b < 0) continue;
      std::vector<uint32_t> out;
      m.move(full, keep, b, false, out, none);
      if (out.empty()) continue;
      nx[ri] = intern(std::move(out), b == '\n' ? kAfterNewline : isWord(b) ? kAfterWord : kAfterNonWord);
    }
    a.next.push_back(std::move(nx));
    a.flag.push_back(std::move(fl));
  }
  a.computeLive();
  return a;
}

void compareForward(const Automaton& r, const Automaton& t, size_t nsym) {
  std::set<std::pair<int32_t, int32_t>> seen;
  std::vector<std::pair<int32_t, int32_t>> todo;
  for (int k = 0; k < 4; k++) {
    for (size_t s = 0; s <= nsym; s++)
      if (t.flag[static_cast<size_t>(t.start[k])][s] || r.flag[static_cast<size_t>(r.start[k])][s]) throw Refuse{"pattern matches the empty string at some position (nullable)"};
    if (seen.emplace(r.start[k], t.start[k]).second) todo.emplace_back(r.start[k], t.start[k]);
  }
  while (!todo.empty()) {
    const auto [rs, ts] = todo.back();
    todo.pop_back();
    const auto& rf = r.flag[static_cast<size_t>(rs)];
    const auto& tf = t.flag[static_cast<size_t>(ts)];
    for (size_t s = 0; s <= nsym; s++)
      if ((rf[s] != 0) != (tf[s] != 0)) throw Refuse{"the reference's look-aware lazy DFA does not answer leftmost-first for this program (early return at a word boundary / sorted boundary resolution)"};
    for (size_t s = 0; s < nsym; s++) {
      int32_t rn = r.next[static_cast<size_t>(rs)][s], tn = t.next[static_cast<size_t>(ts)][s];
      if (rn >= 0 && !r.live[static_cast<size_t>(rn)]) rn = -1;
      if (tn >= 0 && !t.live[static_cast<size_t>(tn)]) tn = -1;
      if ((rn < 0) != (tn < 0)) throw Refuse{"the reference's look-aware lazy DFA does not answer leftmost-first for this program (it stops or goes on where leftmost-first does not)"};
      if (rn >= 0 && seen.emplace(rn, tn).second) todo.emplace_back(rn, tn);
    }
  }
}

// States from which some continuation, possibly empty, raises no further flag (the end of input included).
std::vector<uint8_t> statesThatCanAvoidFlags(const Automaton& t, size_t nsym) {
  const size_t nt = t.next.size();
  std::vector<uint8_t> canAvoid(nt, 0);
  for (size_t i = 0; i < nt; i++) canAvoid[i] = !t.flag[i][nsym];
  for (bool changed = true; changed;) {
    changed = false;
    for (size_t i = 0; i < nt; i++) {
      if (canAvoid[i]) continue;
      for (size_t s = 0; s < nsym && !canAvoid[i]; s++) {
        if (t.flag[i][s]) continue;
        const int32_t tn = t.next[i][s];
        if (tn < 0 || canAvoid[static_cast<size_t>(tn)]) { canAvoid[i] = 1; changed = true; }
      }
    }
  }
  return canAvoid;
}

// UseBoth takes the DFA's match end only to choose where its PikeVM starts — at the search start, or 100 bytes in front of that
// end when it lies further away (find_indices.go:425-431) — and the PikeVM is leftmost-first.  With no match longer than the
// span (checked per haystack by the kernel) the answer is the leftmost-first one whenever the DFA's end is NOT BEHIND the
// leftmost-first end; too early, or none at all, only moves the PikeVM's start further left.  So only this can hurt: R raises a
// flag at a position where T does not, T has raised one before, and the input can go on (or end) without T raising another.
void compareForwardNoLaterEnd(const Automaton& r, const Automaton& t, size_t nsym) {
  const std::vector<uint8_t> canAvoid = statesThatCanAvoidFlags(t, nsym);
  std::set<std::array<int32_t, 3>> seen;
  std::vector<std::array<int32_t, 3>> todo;
  for (int k = 0; k < 4; k++) {
    for (size_t s = 0; s <= nsym; s++)
      if (t.flag[static_cast<size_t>(t.start[k])][s] || r.flag[static_cast<size_t>(r.start[k])][s]) throw Refuse{"pattern matches the empty string at some position (nullable)"};
    const std::array<int32_t, 3> st{r.start[k], t.start[k], 0};
    if (seen.insert(st).second) todo.push_back(st);
  }
  const char* why = "the reference's look-aware lazy DFA can report a match end behind the leftmost-first one for this program: its PikeVM restart (UseBoth) would skip the leftmost match";
  while (!todo.empty()) {
    const auto [rs, ts, flagged] = todo.back();
    todo.pop_back();
    for (size_t s = 0; s <= nsym; s++) {
      const bool rf = r.flag[static_cast<size_t>(rs)][s] != 0;
      const bool tf = ts >= 0 && t.flag[static_cast<size_t>(ts)][s] != 0;
      const int32_t tn = (s < nsym && ts >= 0) ? t.next[static_cast<size_t>(ts)][s] : -1;
      if (rf && !tf && flagged && (s == nsym || tn < 0 || canAvoid[static_cast<size_t>(tn)])) throw Refuse{why};
      if (s == nsym) continue;
      const int32_t rn = r.next[static_cast<size_t>(rs)][s];
      if (rn < 0 || !r.live[static_cast<size_t>(rn)]) continue;                // R raises nothing further
      const int32_t nowFlagged = (flagged || tf) ? 1 : 0;
      if (tn < 0 && !nowFlagged) continue;                                        // T never matches on this input: any R end is harmless
      const std::array<int32_t, 3> nx{rn, tn, nowFlagged};
      if (seen.insert(nx).second) todo.push_back(nx);
    }
  }
}

// UseDFA without a reverse DFA (non-greedy quantifiers, meta/compile.go:184-205) and without a prefilter: the reference asks
// DFA.IsMatchAt first (find_indices.go:396-403 -> lazy.go:561-828 searchEarliestMatch: true at the first match-tagged state, at
// a boundary flag, or at the end of input; false when the walk dies) and runs its PikeVM — leftmost-first — only on a yes.  A
// wrong yes costs nothing (the PikeVM then finds nothing); a wrong no loses the match.  So: wherever T raises a flag, R must have
// raised one or be unable to avoid raising one, and R must not die while T can still raise one.
void compareForwardNoMiss(const Automaton& r, const Automaton& t, size_t nsym) {
  const std::vector<uint8_t> canAvoid = statesThatCanAvoidFlags(r, nsym);
  std::set<std::pair<int32_t, int32_t>> seen;
  std::vector<std::pair<int32_t, int32_t>> todo;
  for (int k = 0; k < 4; k++) {
    for (size_t s = 0; s <= nsym; s++)
      if (t.flag[static_cast<size_t>(t.start[k])][s] || r.flag[static_cast<size_t>(r.start[k])][s]) throw Refuse{"pattern matches the empty string at some position (nullable)"};
    if (seen.emplace(r.start[k], t.start[k]).second) todo.emplace_back(r.start[k], t.start[k]);
  }
  const char* why = "the reference's DFA.IsMatchAt can miss a match of this program (look-aware lazy DFA), and the PikeVM is then never asked";
  while (!todo.empty()) {
    const auto [rs, ts] = todo.back();
    todo.pop_back();
    for (size_t s = 0; s <= nsym; s++) {
      const bool rf = r.flag[static_cast<size_t>(rs)][s] != 0, tf = t.flag[static_cast<size_t>(ts)][s] != 0;
      if (rf) continue;                                                       // IsMatchAt says yes here
      const int32_t rn = s < nsym ? r.next[static_cast<size_t>(rs)][s] : -1;
      const bool rCanStayQuiet = s == nsym || rn < 0 || canAvoid[static_cast<size_t>(rn)];
      if (tf) { if (rCanStayQuiet) throw Refuse{why}; continue; }             // a real match: R must be bound to say yes later
      if (s == nsym) continue;
      const int32_t tn = t.next[static_cast<size_t>(ts)][s];
      if (tn < 0 || !t.live[static_cast<size_t>(tn)]) continue;               // no match on this input any more
      if (rn < 0 || !r.live[static_cast<size_t>(rn)]) throw Refuse{why};      // R is done, T is not
      if (seen.emplace(rn, tn).second) todo.emplace_back(rn, tn);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------- reverse
// Backwards from a match end: the reference's reverse DFA passes every assertion (they became epsilon edges) and reports the
// smallest start it accepts; leftmost-first needs the smallest start of a real match.  States are sets here (no priorities:
// the reverse DFA runs without break-at-match, meta/compile.go:193-194).  Walk both in lockstep; once the real machine has
// accepted, an acceptance of the erased one alone would move the reported start: refuse.
void compareReverse(const cxg_nfa& rv, const std::vector<int>& reps) {
  Machine m(rv);
  auto all = [](uint8_t) { return true; };
  auto none = [](uint8_t) { return false; };
  auto acceptIn = [&](const std::vector<uint32_t>& v) { return m.holdsMatch(v, v.size()); };
  enum { kRightNonWord = 0, kRightWord = 1, kRightNewline = 2, kRightEnd = 3 };
  struct Node { std::vector<uint32_t> erased, real; uint32_t right; bool accepted; };
  auto keyOf = [](const Node& nd) {
    std::vector<uint32_t> k(nd.erased);
    k.push_back(0xFFFFFFFFu);
    k.insert(k.end(), nd.real.begin(), nd.real.end());
    k.push_back(0x80000000u | nd.right | (nd.accepted ? 4u : 0u));
    return k;
  };
  std::set<std::vector<uint32_t>> seen;
  std::vector<Node> todo;
  for (uint32_t right = 0; right < 4; right++) {
    Node nd;
    m.begin(); m.into(nd.erased, rv.start_anchored, all);
    m.begin(); m.into(nd.real, rv.start_anchored, none);
    std::sort(nd.erased.begin(), nd.erased.end());
    std::sort(nd.real.begin(), nd.real.end());
    nd.right = right; nd.accepted = false;
    if (seen.insert(keyOf(nd)).second) todo.push_back(std::move(nd));
  }
  while (!todo.empty()) {
    if (seen.size() > kMaxStates * 4) throw Refuse{"reverse automata exceed the build-time exploration budget"};
    const Node nd = std::move(todo.back());
    todo.pop_back();
    const bool accErased = acceptIn(nd.erased);
    for (size_t ri = 0; ri <= reps.size(); ri++) {
      const int c = ri < reps.size() ? reps[ri] : kSymEnd;   // the byte to the left, or the begin of input
      const bool wordLeft = c >= 0 && isWord(c), wordRight = nd.right == kRightWord;
      auto pass = [&](uint8_t l) {
        switch (l) {
          case kLkStartText: return c < 0;
          case kLkEndText: return nd.right == kRightEnd;
          case kLkStartLine: return c < 0 || c == '\n';
          case kLkEndLine: return nd.right == kRightEnd || nd.right == kRightNewline;
          case kLkWordB: return wordLeft != wordRight;
          case kLkNoWordB: return wordLeft == wordRight;
          default: return false;
        }
      };
      std::vector<uint32_t> full;
      m.begin();
      for (uint32_t s : nd.real) m.into(full, s, pass);
      const bool accReal = acceptIn(full);
      if (accReal && !accErased) throw Refuse{"internal: erased reverse automaton misses a real start"};
      if (accErased && !accReal && nd.accepted) throw Refuse{"the reference's reverse DFA ignores the assertions and would report an earlier match start"};
      if (c < 0) continue;
      Node nx;
      m.move(nd.erased, nd.erased.size(), c, false, nx.erased, all);
      if (nx.erased.empty()) continue;   // the reference's reverse search is over; the real machine (a subset) too
      m.move(full, full.size(), c, false, nx.real, none);
      std::sort(nx.erased.begin(), nx.erased.end());
      std::sort(nx.real.begin(), nx.real.end());
      nx.right = c == '\n' ? kRightNewline : isWord(c) ? kRightWord : kRightNonWord;
      nx.accepted = nd.accepted || accReal;
      if (seen.insert(keyOf(nx)).second) todo.push_back(std::move(nx));
    }
  }
}

}  // namespace

namespace {

struct Symbols { std::vector<int> reps, classOf; bool hasWordB = false, hasLine = false, hasEndLine = false; };

// One representative byte per (byte class of the reference, kind the pattern's assertions can tell apart).
Symbols symbolsOf(const cxg_nfa& nfa) {
  Symbols sy;
  for (uint32_t i = 0; i < nfa.n_states; i++) {
    const cxg_nfa_state& s = nfa.states[i];
    if (s.kind != CXG_NFA_LOOK) continue;
    // (\A / ^: the machines below know the Text start kind — start.go:64-172, look.go:88-107; the transducer serves it, fsm.cc)
    // (\z / $: both machines below know the end of the input — kSymEnd, CheckEOIMatch — and the transducer has its kind since round 6)
    if (s.lo == kLkWordB || s.lo == kLkNoWordB) sy.hasWordB = true;
    if (s.lo == kLkStartLine || s.lo == kLkEndLine) sy.hasLine = true;
    if (s.lo == kLkEndLine) sy.hasEndLine = true;
  }
  // byte classes of the reference (nfa/alphabet.go:100-166): boundaries at the ends of the pattern's byte ranges only
  bool boundary[256] = {false};
  auto markRange = [&](int lo, int hi) { if (lo > 0) boundary[lo - 1] = true; boundary[hi] = true; };
  for (uint32_t i = 0; i < nfa.n_states; i++) {
    const cxg_nfa_state& s = nfa.states[i];
    if (s.kind == CXG_NFA_BYTE_RANGE) markRange(s.lo, s.hi);
    else if (s.kind == CXG_NFA_SPARSE) for (uint32_t k = 0; k < s.trans_len; k++) markRange(nfa.trans[s.trans_off + k].lo, nfa.trans[s.trans_off + k].hi);
  }
  int nclass = 0;
  for (int b = 0, lo = 0; b < 256; b++) {
    if (b == 255 || boundary[b]) {
      bool seen[2][2] = {{false, false}, {false, false}};
      for (int x = lo; x <= b; x++) {
        const int w = sy.hasWordB && isWord(x), nl = sy.hasLine && x == '\n';
        if (seen[w][nl]) continue;
        seen[w][nl] = true;
        sy.reps.push_back(x);
        sy.classOf.push_back(nclass);
      }
      nclass++;
      lo = b + 1;
    }
  }
  return sy;
}

}  // namespace

void refuseLookDfaQuirks(const cxg_nfa& nfa, const cxg_nfa* reverse, bool existenceOnly) {
  const Symbols sy = symbolsOf(nfa);
  try {
    const Automaton r = buildReference(nfa, nfa.start_unanchored, sy.reps, sy.classOf, sy.hasWordB, sy.hasEndLine);
    const Automaton t = buildLeftmostFirst(nfa, nfa.start_unanchored, sy.reps);
    if (existenceOnly) {
      compareForwardNoMiss(r, t, sy.reps.size());
    } else if (reverse) {
      compareForward(r, t, sy.reps.size());
      compareReverse(*reverse, sy.reps);
    } else {
      compareForwardNoLaterEnd(r, t, sy.reps.size());
    }
  } catch (const Refuse& e) {
    throw BuildError{CXG_E_UNSUPPORTED, e.why};
  }
}

namespace {

// The digit-run skip (find_indices.go:1079-1084): after a failure at the first digit of a run the reference goes on behind the
// run.  Sound iff no later digit of the run would have matched: walk the search from the head of the run (X, any start kind,
// at least one digit in) and a fresh search from a later digit (Y, start kind "after a word byte") over the same bytes; wherever
// Y reports a match, X must have reported one or be unable to avoid reporting one.  (The criterion of the plain digit programs —
// the first digit leads to one state that every digit keeps — is the special case in which X and Y coincide at once.)
void runSkipIsSound(const Automaton& t, const std::vector<int>& reps) {
  const size_t nsym = reps.size();
  const std::vector<uint8_t> canAvoid = statesThatCanAvoidFlags(t, nsym);
  const char* why = "digit-scan order differs from leftmost-first (run-skip quirk)";
  std::vector<size_t> digits;
  for (size_t i = 0; i < nsym; i++) if (reps[i] >= '0' && reps[i] <= '9') digits.push_back(i);
  std::set<std::array<int32_t, 3>> seen;            // (X or -1, Y, X has reported)
  std::vector<std::array<int32_t, 3>> todo;
  {                                                 // heads: every state X reaches inside a run, paired with a fresh Y
    std::set<std::pair<int32_t, int32_t>> heads;
    std::vector<std::pair<int32_t, int32_t>> hq;
    for (int k = 0; k < 4; k++) if (heads.emplace(t.start[k], 0).second) hq.emplace_back(t.start[k], 0);
    while (!hq.empty()) {
      const auto [x, xf] = hq.back();
      hq.pop_back();
      for (size_t d : digits) {
        const int32_t xf2 = (xf || t.flag[static_cast<size_t>(x)][d]) ? 1 : 0;
        const int32_t xn = t.next[static_cast<size_t>(x)][d];
        const std::array<int32_t, 3> st{xn, t.start[kAfterWord], xf2};
        if (!xf2 && seen.insert(st).second) todo.push_back(st);      // (a head that has reported is a success: nothing is skipped)
        if (xn >= 0 && heads.emplace(xn, xf2).second) hq.emplace_back(xn, xf2);
      }
    }
  }
  while (!todo.empty()) {
    const auto [x, y, unused] = todo.back();
    (void)unused;
    todo.pop_back();
    for (size_t s = 0; s <= nsym; s++) {
      const bool yf = t.flag[static_cast<size_t>(y)][s] != 0;
      const bool xf = x >= 0 && t.flag[static_cast<size_t>(x)][s] != 0;
      const int32_t xn = (s < nsym && x >= 0) ? t.next[static_cast<size_t>(x)][s] : -1;
      if (yf && !xf && (s == nsym || xn < 0 || canAvoid[static_cast<size_t>(xn)])) throw Refuse{why};
      if (s == nsym || xf || yf) continue;           // X has reported, or is bound to: this input is no failure of the head
      const int32_t yn = t.next[static_cast<size_t>(y)][s];
      if (yn < 0 || !t.live[static_cast<size_t>(yn)]) continue;
      const std::array<int32_t, 3> nx{xn, yn, 0};
      if (seen.insert(nx).second) todo.push_back(nx);
    }
  }
}

}  // namespace

// UseDigitPrefilter (find_indices.go:1050-1088): at each digit position in turn, SearchAtAnchored — the same lazy DFA from its
// ANCHORED start state of the kind of the byte in front (lazy.go:219-324; its boundary check reads the flags determinize
// stored, state.go:238-247, which equal checkWordBoundaryMatch for every state determinize made, and a start state of a
// pattern that begins with a digit holds no match behind a boundary).  First position that succeeds wins: leftmost-first over
// matches that begin with a digit, i.e. over all matches of such a pattern — provided the anchored machine is the leftmost-
// first one, and provided the digit-run skip (digitRunSkipSafe, compile.go:176, find_indices.go:1079-1084) is sound: after a
// failure at the first digit of a run every later digit of the run must fail too (runSkipIsSound).
void refuseLookDigitQuirks(const cxg_nfa& nfa, bool runSkip) {
  const Symbols sy = symbolsOf(nfa);
  try {
    const Automaton r = buildReference(nfa, nfa.start_anchored, sy.reps, sy.classOf, sy.hasWordB, sy.hasEndLine);
    const Automaton t = buildLeftmostFirst(nfa, nfa.start_anchored, sy.reps);
    compareForward(r, t, sy.reps.size());
    if (runSkip) runSkipIsSound(t, sy.reps);
  } catch (const Refuse& e) {
    throw BuildError{CXG_E_UNSUPPORTED, e.why};
  }
}

}  // namespace cxg
