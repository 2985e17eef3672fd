// ORACLE — TEST INFRASTRUCTURE ONLY (see engines.hpp header for the reference map).
#include "engines.hpp"

#include <algorithm>
#include <cstring>

namespace orc {

// =============================================================== lazy DFA
void LazyDFA::init(const NFA* n, bool brk) {
  nfa = n; breakAtMatch = brk;
  hasWordBoundary = hasEndLine = false;   // builder.go:714-750
  for (const NState& s : n->states)
    if (s.kind == StateLook) {
      if (s.look == LookWordBoundary || s.look == LookNoWordBoundary) hasWordBoundary = true;
      if (s.look == LookEndLine) hasEndLine = true;
    }
  for (auto& row : starts) for (int32_t& x : row) x = -2;
  states.clear();
  cache.clear();
}

static bool lookSatisfied(uint32_t have, Look l) {  // LookSet.Contains, look.go:43-60: \b and \B are never in a LookSet
  switch (l) {
    case LookStartText: return have & LazyDFA::HaveStartText;
    case LookEndText: return have & LazyDFA::HaveEndText;
    case LookStartLine: return have & LazyDFA::HaveStartLine;
    case LookEndLine: return have & LazyDFA::HaveEndLine;
    default: return false;
  }
}

void LazyDFA::closureInto(std::vector<StateID>& out, std::vector<uint8_t>& in, StateID seed, uint32_t lookHave) const {
  // builder.go:245-293: explicit stack, add-on-pop, push right then left; a Look state is added, and passed only when
  // its assertion is in lookHave.
  std::vector<StateID> stack{seed};
  while (!stack.empty()) {
    StateID cur = stack.back();
    stack.pop_back();
    if (cur == kInvalidState || cur >= nfa->states.size()) continue;
    if (in[cur]) continue;
    in[cur] = 1;
    out.push_back(cur);
    const NState& s = nfa->states[cur];
    switch (s.kind) {
      case StateEpsilon: if (s.next != kInvalidState) stack.push_back(s.next); break;
      case StateSplit:
        if (s.right != kInvalidState) stack.push_back(s.right);
        if (s.left != kInvalidState) stack.push_back(s.left);
        break;
      case StateCapture: if (s.next != kInvalidState) stack.push_back(s.next); break;
      case StateLook: if (lookSatisfied(lookHave, s.look) && s.next != kInvalidState) stack.push_back(s.next); break;
      default: break;
    }
  }
}

std::vector<StateID> LazyDFA::closureOf(const std::vector<StateID>& seeds, uint32_t lookHave) const {
  std::vector<StateID> out;
  std::vector<uint8_t> in(nfa->states.size(), 0);
  for (StateID s : seeds) closureInto(out, in, s, lookHave);
  return out;
}

bool LazyDFA::holdsMatch(const std::vector<StateID>& set) const {
  for (StateID s : set) if (nfa->isMatch(s)) return true;
  return false;
}

std::vector<StateID> LazyDFA::resolveWordBoundaries(const std::vector<StateID>& set, bool satisfied) const {
  // builder.go:295-425.  Only states BEHIND a crossed \b / \B are added; when any was crossed the result is the union
  // in SORTED order (StateSet.ToSlice, state.go:487-497) — the priority order of the input is gone.
  std::vector<uint8_t> crossed(nfa->states.size(), 0);
  std::vector<StateID> stack;
  auto cross = [&](const NState& st) {
    if (st.next == kInvalidState) return;
    const bool ok = (st.look == LookWordBoundary && satisfied) || (st.look == LookNoWordBoundary && !satisfied);
    if (ok && st.next < crossed.size() && !crossed[st.next]) { crossed[st.next] = 1; stack.push_back(st.next); }
  };
  auto follow = [&](StateID t) { if (t != kInvalidState && t < crossed.size() && !crossed[t]) { crossed[t] = 1; stack.push_back(t); } };
  for (StateID sid : set) {
    if (sid >= nfa->states.size()) continue;
    const NState& st = nfa->states[sid];
    if (st.kind == StateLook) cross(st);
  }
  if (stack.empty()) return set;
  while (!stack.empty()) {
    const StateID cur = stack.back();
    stack.pop_back();
    const NState& st = nfa->states[cur];
    switch (st.kind) {
      case StateLook: cross(st); break;
      case StateEpsilon: follow(st.next); break;
      case StateSplit: follow(st.left); follow(st.right); break;
      case StateCapture: follow(st.next); break;
      default: break;
    }
  }
  std::vector<uint8_t> in(crossed);
  for (StateID sid : set) if (sid < in.size()) in[sid] = 1;
  std::vector<StateID> out;
  for (StateID i = 0; i < in.size(); i++) if (in[i]) out.push_back(i);
  return out;
}

static std::vector<uint32_t> makeKey(const std::vector<StateID>& ids, bool fromWord, bool isMatch) {
  // state.go:329-373: sorted ids + flags (the reference hashes this tuple with FNV-1a).
  std::vector<uint32_t> k(ids.begin(), ids.end());
  std::sort(k.begin(), k.end());
  k.push_back(0x80000000u | (fromWord ? 1u : 0u) | (isMatch ? 2u : 0u));
  return k;
}

int32_t LazyDFA::startStateOfKind(StartKind kind, bool anchored) {
  // lazy.go:1569-1613 + start.go:205-254.  The assertions a start position satisfies (look.go:88-107): StartText \A and ^,
  // StartLineLF ^ only; \b and \B wait for the first byte (isFromWord = kind == StartWord).  Never a match state.
  int32_t& slot = starts[anchored ? 1 : 0][kind];
  if (slot != -2) return slot;
  const uint32_t have = kind == StartText ? (HaveStartText | HaveStartLine) : kind == StartLineLF ? HaveStartLine : 0u;
  std::vector<StateID> set = closureOf({anchored ? nfa->startAnchored : nfa->startUnanchored}, have);
  const bool fromWord = kind == StartWord;
  auto key = makeKey(set, fromWord, false);
  auto it = cache.find(key);
  int32_t id;
  if (it != cache.end()) id = it->second;   // GetOrInsert: an existing state (its order, its flags) is kept
  else {
    DState d;
    d.nfaStates = std::move(set); d.isMatch = false; d.isFromWord = fromWord;   // no matchAt* flags: ComputeStartState sets none
    d.trans.assign(nfa->alphabetLen, kUnknown);
    states.push_back(std::move(d));
    id = static_cast<int32_t>(states.size() - 1);
    cache[key] = id;
  }
  // WithStartTag marks the State object (lazy.go:1607).  (Transitions cached BEFORE the tag keep the untagged id in the
  // reference; here the tag shows on every way in.  Observable only through the skipped word-boundary check in searchAt,
  // i.e. for patterns that match empty at a start position.)
  states[id].startTagged = true;
  return slot = id;
}

int32_t LazyDFA::startState(Bytes h, int64_t pos, bool anchored) {
  StartKind kind = StartText;   // start.go:96-109, :128-133
  if (pos > 0) {
    const uint8_t p = h[pos - 1];
    kind = p == '\n' ? StartLineLF : p == '\r' ? StartLineCR : isWordByte(p) ? StartWord : StartNonWord;
  }
  return startStateOfKind(kind, anchored);
}

int32_t LazyDFA::next(int32_t sid, uint8_t b) {
  int cls = nfa->byteClasses[b];
  int32_t t = states[sid].trans[cls];
  if (t != kUnknown) return t;
  // determinize, lazy.go:1336-1446
  std::vector<StateID> cur = states[sid].nfaStates;  // copy: states may reallocate
  if (hasEndLine && b == '\n') cur = closureOf(cur, HaveEndLine);   // :1350-1354: $ holds in front of this byte
  const bool sourceHasMatch = holdsMatch(cur);
  const bool brk = sourceHasMatch && breakAtMatch;
  // moveWithWordContextBreak, builder.go:183-242
  const std::vector<StateID> resolved = hasWordBoundary ? resolveWordBoundaries(cur, states[sid].isFromWord != isWordByte(b)) : cur;
  const uint32_t lookAfter = b == '\n' ? HaveStartLine : 0u;
  std::vector<StateID> nextSet;
  std::vector<uint8_t> in(nfa->states.size(), 0);
  for (StateID s : resolved) {
    const NState& st = nfa->states[s];
    if (brk && st.kind == StateMatch) break;
    if (st.kind == StateByteRange) {
      if (b >= st.lo && b <= st.hi) closureInto(nextSet, in, st.next, lookAfter);
    } else if (st.kind == StateSparse) {
      for (auto& tr : st.trans)
        if (b >= tr.lo && b <= tr.hi) closureInto(nextSet, in, tr.next, lookAfter);
    }
  }
  bool isMatch = sourceHasMatch;
  if (nextSet.empty() && !isMatch) { states[sid].trans[cls] = kDead; return kDead; }
  bool nextFromWord = isWordByte(b);
  auto key = makeKey(nextSet, nextFromWord, isMatch);
  auto it = cache.find(key);
  if (it != cache.end()) { states[sid].trans[cls] = it->second; return it->second; }
  DState d;
  d.isMatch = isMatch; d.isFromWord = nextFromWord;
  if (hasWordBoundary && !isMatch) {   // :1413-1421
    d.matchAtWordBoundary = holdsMatch(resolveWordBoundaries(nextSet, true));
    d.matchAtNonWordBoundary = holdsMatch(resolveWordBoundaries(nextSet, false));
  }
  d.nfaStates = std::move(nextSet);
  d.trans.assign(nfa->alphabetLen, kUnknown);
  states.push_back(std::move(d));
  int32_t id = static_cast<int32_t>(states.size() - 1);
  cache[key] = id;
  states[sid].trans[cls] = id;
  return id;
}

bool LazyDFA::eoiMatch(int32_t sid) const {
  // checkEOIMatch lazy.go:1512-1522 -> CheckEOIMatch builder.go:437-472: outside the haystack counts as a non-word byte,
  // \z and $ hold.
  const DState& st = states[sid];
  return holdsMatch(closureOf(resolveWordBoundaries(st.nfaStates, st.isFromWord), HaveEndText | HaveEndLine));
}

bool LazyDFA::wordBoundaryMatch(int32_t sid, uint8_t b) const {
  // checkWordBoundaryMatch lazy.go:1533-1560.  True as soon as the state's own NFA set holds a match state, crossed
  // boundary or not: a match-holding state that is not match-TAGGED (1-byte delay) ends the search at this byte.
  const DState& st = states[sid];
  if (st.isMatch) return false;
  return holdsMatch(resolveWordBoundaries(st.nfaStates, st.isFromWord != isWordByte(b)));
}

bool LazyDFA::matchesEmpty() {
  // lazy.go:1636-1648.  cache.getState(StartState) is nil until a cache clear (ids start at `stride`, cache.go:99-107,
  // :317-321), so this is the PikeVM asked about the EMPTY haystack — not about the position the caller stands at.
  PikeVM pv;
  pv.init(nfa);
  int64_t s = -1, e = -1;
  const uint8_t none = 0;
  return pv.searchAt(&none, 0, 0, s, e) && s == 0 && e == 0;
}

int64_t LazyDFA::searchAtAnchored(Bytes h, int64_t n, int64_t at) {
  if (at > n) return -1;
  if (at == n || n == 0) return matchesEmpty() ? at : -1;
  int32_t sid = startState(h, at, true);
  int64_t lastMatch = -1;
  for (int64_t pos = at; pos < n; pos++) {
    if (hasWordBoundary) {   // checkWordBoundaryFast state.go:238-247: the flags determinize stored
      const DState& st = states[sid];
      if (!st.isMatch && ((st.isFromWord != isWordByte(h[pos])) ? st.matchAtWordBoundary : st.matchAtNonWordBoundary)) return pos;
    }
    int32_t nx = next(sid, h[pos]);
    if (nx == kDead) return lastMatch;
    sid = nx;
    if (states[sid].isMatch) lastMatch = pos;  // 1-byte match delay, lazy.go:310-312
  }
  if (eoiMatch(sid)) return n;
  return lastMatch;
}

int64_t LazyDFA::searchAt(Bytes h, int64_t n, int64_t at) {
  if (at > n) return -1;
  if (at == n || n == 0) return matchesEmpty() ? at : -1;
  if (nfa->anchored && at > 0) return -1;  // lazy.go:1108-1110
  int32_t sid = startState(h, at, false);
  int64_t lastMatch = -1;
  for (int64_t pos = at; pos < n; pos++) {
    if (states[sid].startTagged) {
      if (prefilterFind && lastMatch < 0 && pos > at) {   // lazy.go:1210-1227: skip ahead, restart at the candidate
        const int64_t cand = prefilterFind(h, n, pos);
        if (cand == -1) return lastMatch;
        if (cand > pos) { pos = cand; sid = startState(h, pos, false); pos--; continue; }
      }
      // lazy.go:1229-1243: cached transition out of a start state, no boundary check
      const int32_t t = states[sid].trans[nfa->byteClasses[h[pos]]];
      if (t != kUnknown && t != kDead) {
        sid = t;
        if (states[sid].isMatch) lastMatch = pos;
        continue;
      }
    }
    if (hasWordBoundary && wordBoundaryMatch(sid, h[pos])) return pos;   // :1262-1264
    int32_t nx = next(sid, h[pos]);
    if (nx == kDead) return lastMatch;
    sid = nx;
    if (states[sid].isMatch) lastMatch = pos;  // lazy.go:1301-1303
  }
  if (eoiMatch(sid)) return n;
  return lastMatch;
}

bool LazyDFA::isMatchAt(Bytes h, int64_t n, int64_t at) {
  // IsMatchAt lazy.go:546-557 -> searchEarliestMatch :561-828: true at the first match-tagged state, at a byte at which the
  // boundary flags of the state hold (checkWordBoundaryFast, the flags determinize stored: a start state has none), or at
  // the end of input through checkEOIMatch; false when the walk dies.  With a prefilter a dead walk restarts at the next
  // candidate (:805-825) and a start-tagged state skips ahead (:682-700).
  if (at > n) return false;
  if (at == n) return matchesEmpty();
  if (nfa->anchored && at > 0) return false;
  int32_t sid = startState(h, at, false);
  int64_t pos = at;
  while (pos < n) {
    if (states[sid].startTagged) {
      if (prefilterFind && pos > at) {
        const int64_t cand = prefilterFind(h, n, pos);
        if (cand == -1) return false;
        if (cand > pos) { pos = cand; sid = startState(h, pos, false); continue; }
      }
      const int32_t t = states[sid].trans[nfa->byteClasses[h[pos]]];
      if (t != kUnknown && t != kDead) { sid = t; pos++; if (states[sid].isMatch) return true; continue; }
    }
    const uint8_t b = h[pos];
    if (hasWordBoundary) {
      const DState& st = states[sid];
      if (!st.isMatch && ((st.isFromWord != isWordByte(b)) ? st.matchAtWordBoundary : st.matchAtNonWordBoundary)) return true;
    }
    const int32_t nx = next(sid, b);
    if (nx == kDead) {
      if (!prefilterFind) return false;
      pos++;
      if (pos >= n) return false;
      const int64_t cand = prefilterFind(h, n, pos);
      if (cand == -1) return false;
      pos = cand; sid = startState(h, pos, false);
      continue;
    }
    sid = nx; pos++;
    if (states[sid].isMatch) return true;
  }
  return eoiMatch(sid);
}

int64_t LazyDFA::searchReverse(Bytes h, int64_t n, int64_t start, int64_t end) {
  if (end <= start || end > n) return -1;
  // getStartStateForReverse lazy.go:2123-2158: the kind of the byte at `end` (StartText at the end of the haystack).
  // The reverse NFA holds no assertions (nfa/reverse.go:124-129 turns them into epsilon edges), so only the key differs.
  StartKind kind = StartText;
  if (end < n) { const uint8_t p = h[end]; kind = p == '\n' ? StartLineLF : p == '\r' ? StartLineCR : isWordByte(p) ? StartWord : StartNonWord; }
  int32_t sid = startStateOfKind(kind, true);
  int64_t lastMatch = -1;
  for (int64_t at = end - 1; at >= start; at--) {
    int32_t nx = next(sid, h[at]);
    if (nx == kDead) return lastMatch;
    sid = nx;
    if (states[sid].isMatch) lastMatch = at + 1;  // lazy.go:1905-1907
  }
  if (holdsMatch(states[sid].nfaStates)) lastMatch = start;   // lazy.go:1914-1917
  return lastMatch;
}

NFA reverseNFA(const NFA& fwd) {
  // R(t) for every forward state t means "the forward run sits at t here".  Reading a byte b
  // backwards moves R(t) -> R(s) when s consumes b and s.next == t; epsilon edges are reversed.
  // Start = R(Match); accepting = R(startAnchored).  The unanchored prefix states are excluded.
  const size_t N = fwd.states.size();
  std::vector<std::vector<Transition>> byteIn(N);   // incoming byte edges: (lo,hi,from)
  std::vector<std::vector<StateID>> epsIn(N);
  std::vector<uint8_t> skip(N, 0);
  if (fwd.startUnanchored != fwd.startAnchored) {
    // prefix = Split(pattern, anyByte) + the anyByte state (compile.go:1633-1650)
    skip[fwd.startUnanchored] = 1;
    const NState& sp = fwd.states[fwd.startUnanchored];
    if (sp.kind == StateSplit && sp.right != kInvalidState) skip[sp.right] = 1;
  }
  for (StateID s = 0; s < N; s++) {
    if (skip[s]) continue;
    const NState& st = fwd.states[s];
    switch (st.kind) {
      case StateByteRange:
        if (st.next != kInvalidState) byteIn[st.next].push_back({st.lo, st.hi, s});
        break;
      case StateSparse:
        for (auto& t : st.trans) if (t.next != kInvalidState) byteIn[t.next].push_back({t.lo, t.hi, s});
        break;
      case StateSplit:
        if (st.left != kInvalidState) epsIn[st.left].push_back(s);
        if (st.right != kInvalidState) epsIn[st.right].push_back(s);
        break;
      case StateEpsilon: case StateCapture: case StateLook:
        if (st.next != kInvalidState) epsIn[st.next].push_back(s);
        break;
      default: break;
    }
  }
  NFA r;
  // layout: R(t) entry epsilon at id t (0..N-1), then helper states appended.
  r.states.resize(N);
  for (StateID t = 0; t < N; t++) { r.states[t].kind = StateEpsilon; r.states[t].next = kInvalidState; }
  auto add = [&](NState s) { r.states.push_back(std::move(s)); return static_cast<StateID>(r.states.size() - 1); };
  NState m; m.kind = StateMatch;
  StateID revMatch = add(m);
  uint64_t bits[4] = {0, 0, 0, 0};
  auto setBit = [&](uint8_t b) { bits[b / 64] |= uint64_t{1} << (b % 64); };
  for (StateID t = 0; t < N; t++) {
    if (skip[t]) continue;
    std::vector<StateID> alts;
    if (t == fwd.startAnchored) alts.push_back(revMatch);
    if (!byteIn[t].empty()) {
      NState sp; sp.kind = StateSparse; sp.trans = byteIn[t];
      for (auto& tr : sp.trans) { if (tr.lo > 0) setBit(tr.lo - 1); setBit(tr.hi); }
      alts.push_back(add(sp));
    }
    for (StateID s : epsIn[t]) alts.push_back(s);
    if (alts.empty()) { r.states[t].kind = StateFail; continue; }
    StateID chain = alts.back();
    for (size_t i = alts.size() - 1; i-- > 0;) {
      NState sp; sp.kind = StateSplit; sp.left = alts[i]; sp.right = chain;
      chain = add(sp);
    }
    r.states[t].next = chain;
  }
  StateID fwdMatch = kInvalidState;
  for (StateID s = 0; s < N; s++) if (fwd.states[s].kind == StateMatch) fwdMatch = s;
  r.startAnchored = r.startUnanchored = fwdMatch;
  r.anchored = true;
  r.captureCount = 1;
  uint8_t cls = 0;
  for (int b = 0; b < 256; b++) {
    r.byteClasses[b] = cls;
    if (bits[b / 64] & (uint64_t{1} << (b % 64))) cls++;
  }
  int mx = 0;
  for (int b = 0; b < 256; b++) mx = std::max<int>(mx, r.byteClasses[b]);
  r.alphabetLen = mx + 1;
  r.hasLook = fwd.hasLook;
  r.hasWordBoundary = fwd.hasWordBoundary;
  return r;
}

// =============================================================== PikeVM
namespace {

bool checkLook(Look look, Bytes h, int64_t n, int64_t pos) {  // pikevm.go:1646-1674
  switch (look) {
    case LookStartText: return pos == 0;
    case LookEndText: return pos == n;
    case LookStartLine: return pos == 0 || h[pos - 1] == '\n';
    case LookEndLine: return pos == n || h[pos] == '\n';
    case LookWordBoundary: case LookNoWordBoundary: {
      bool before = pos > 0 && isWordByte(h[pos - 1]);
      bool after = pos < n && isWordByte(h[pos]);
      bool wb = before != after;
      return look == LookWordBoundary ? wb : !wb;
    }
  }
  return false;
}

struct SparseSet {  // internal/sparse/sparse.go:42-114
  std::vector<uint32_t> dense, sparse;
  uint32_t len = 0;
  void init(size_t cap) { dense.assign(cap, 0); sparse.assign(cap, 0); len = 0; }
  bool contains(uint32_t v) const { uint32_t i = sparse[v]; return i < len && dense[i] == v; }
  bool insert(uint32_t v) {
    if (contains(v)) return false;
    dense[len] = v; sparse[v] = len; len++;
    return true;
  }
  void clear() { len = 0; }
};

struct Thread { StateID state; int64_t startPos; };
struct Frame { StateID state; int64_t startPos; int slot; int64_t value; };

struct Pike {
  const NFA& nfa;
  Bytes h; int64_t n;
  int totalSlots; bool track;
  std::vector<Thread> queue, nextQueue;
  SparseSet visited;
  std::vector<int64_t> table, nextTable, currSlots;
  std::vector<Frame> stack;

  Pike(const NFA& a, Bytes hh, int64_t nn, int slots)
      : nfa(a), h(hh), n(nn), totalSlots(slots), track(slots > 2) {
    visited.init(nfa.states.size());
    table.assign(nfa.states.size() * totalSlots, -1);
    nextTable.assign(nfa.states.size() * totalSlots, -1);
    currSlots.assign(totalSlots, -1);
  }

  // addSearchThread / addSearchThreadToNext (pikevm.go:1895-2006 / :2066-2174)
  void addThread(Thread t, int64_t pos, bool toNext, StateID src) {
    std::vector<Thread>& q = toNext ? nextQueue : queue;
    std::vector<int64_t>& tbl = toNext ? nextTable : table;
    if (toNext && track)
      std::copy(table.begin() + src * totalSlots, table.begin() + (src + 1) * totalSlots, currSlots.begin());
    stack.clear();
    stack.push_back({t.state, t.startPos, 0, 0});
    while (!stack.empty()) {
      Frame f = stack.back();
      stack.pop_back();
      if (f.state == kInvalidState) {  // RestoreCapture frame
        if (track && f.slot < totalSlots) currSlots[f.slot] = f.value;
        continue;
      }
      StateID sid = f.state;
      if (sid >= nfa.states.size()) continue;
      if (!visited.insert(sid)) continue;
      const NState& st = nfa.states[sid];
      switch (st.kind) {
        case StateMatch: case StateByteRange: case StateSparse:
          if (track) std::copy(currSlots.begin(), currSlots.end(), tbl.begin() + sid * totalSlots);
          q.push_back({sid, f.startPos});
          break;
        case StateEpsilon:
          if (st.next != kInvalidState) stack.push_back({st.next, f.startPos, 0, 0});
          break;
        case StateSplit:
          if (st.right != kInvalidState) stack.push_back({st.right, f.startPos, 0, 0});
          if (st.left != kInvalidState) stack.push_back({st.left, f.startPos, 0, 0});
          break;
        case StateCapture:
          if (st.next != kInvalidState) {
            if (track) {
              int slot = static_cast<int>(st.capIndex) * 2 + (st.capStart ? 0 : 1);
              if (slot < totalSlots) {
                stack.push_back({kInvalidState, 0, slot, currSlots[slot]});
                currSlots[slot] = pos;
              }
            }
            stack.push_back({st.next, f.startPos, 0, 0});
          }
          break;
        case StateLook:
          if (checkLook(st.look, h, n, pos) && st.next != kInvalidState)
            stack.push_back({st.next, f.startPos, 0, 0});
          break;
        default: break;
      }
    }
  }

  void step(const Thread& t, uint8_t b, int64_t nextPos) {  // pikevm.go:2009-2063
    const NState& st = nfa.states[t.state];
    if (st.kind == StateByteRange) {
      if (b >= st.lo && b <= st.hi) addThread({st.next, t.startPos}, nextPos, true, t.state);
    } else if (st.kind == StateSparse) {
      for (auto& tr : st.trans)
        if (b >= tr.lo && b <= tr.hi) addThread({tr.next, t.startPos}, nextPos, true, t.state);
    }
  }

  static bool better(int64_t bs, int64_t be, int64_t cs, int64_t ce) {  // pikevm.go:155-173
    if (bs == -1) return true;
    if (cs < bs) return true;
    if (cs > bs) return false;
    return ce > be;
  }

  bool runUnanchored(int64_t startAt, std::vector<int64_t>& out) {  // pikevm.go:2225-2328
    int64_t bestStart = -1, bestEnd = -1;
    std::vector<int64_t> bestSlots;
    for (int64_t pos = startAt; pos <= n; pos++) {
      if (bestStart == -1) {
        std::fill(currSlots.begin(), currSlots.end(), -1);
        addThread({nfa.startAnchored, pos}, pos, false, 0);  // Visited NOT cleared (pikevm.go:2248-2255)
      }
      if (pos < n) {
        uint8_t b = h[pos];
        visited.clear();
        for (size_t i = 0; i < queue.size(); i++) {
          Thread t = queue[i];
          if (nfa.isMatch(t.state)) {
            if (better(bestStart, bestEnd, t.startPos, pos)) {
              bestStart = t.startPos; bestEnd = pos;
              if (totalSlots > 0)
                bestSlots.assign(table.begin() + t.state * totalSlots, table.begin() + (t.state + 1) * totalSlots);
            }
            break;
          }
          step(t, b, pos + 1);
        }
      } else {
        for (size_t i = 0; i < queue.size(); i++) {
          Thread t = queue[i];
          if (nfa.isMatch(t.state)) {
            if (better(bestStart, bestEnd, t.startPos, pos)) {
              bestStart = t.startPos; bestEnd = pos;
              if (totalSlots > 0)
                bestSlots.assign(table.begin() + t.state * totalSlots, table.begin() + (t.state + 1) * totalSlots);
            }
            break;
          }
        }
      }
      if (pos >= n) break;
      if (bestStart != -1) {
        bool has = false;
        for (auto& t : nextQueue) if (t.startPos <= bestStart) { has = true; break; }
        if (!has) break;
      }
      queue.swap(nextQueue);
      nextQueue.clear();
      table.swap(nextTable);
    }
    if (bestStart == -1) return false;
    finish(bestSlots, bestStart, bestEnd, out);
    return true;
  }

  bool runAnchored(int64_t startPos, std::vector<int64_t>& out) {  // pikevm.go:2332-2406
    std::fill(currSlots.begin(), currSlots.end(), -1);
    addThread({nfa.startAnchored, startPos}, startPos, false, 0);
    int64_t lastMatch = -1;
    std::vector<int64_t> bestSlots;
    for (int64_t pos = startPos; pos <= n; pos++) {
      if (pos < n) {
        uint8_t b = h[pos];
        visited.clear();
        for (size_t i = 0; i < queue.size(); i++) {
          Thread t = queue[i];
          if (nfa.isMatch(t.state)) {
            if (pos > lastMatch || lastMatch == -1) {
              lastMatch = pos;
              if (totalSlots > 0)
                bestSlots.assign(table.begin() + t.state * totalSlots, table.begin() + (t.state + 1) * totalSlots);
            }
            break;
          }
          step(t, b, pos + 1);
        }
      } else {
        for (size_t i = 0; i < queue.size(); i++) {
          Thread t = queue[i];
          if (nfa.isMatch(t.state)) {
            if (pos > lastMatch || lastMatch == -1) {
              lastMatch = pos;
              if (totalSlots > 0)
                bestSlots.assign(table.begin() + t.state * totalSlots, table.begin() + (t.state + 1) * totalSlots);
            }
            break;
          }
        }
      }
      if (nextQueue.empty() && (pos >= n || lastMatch != -1)) break;
      if (pos >= n) break;
      queue.swap(nextQueue);
      nextQueue.clear();
      table.swap(nextTable);
    }
    if (lastMatch == -1) return false;
    finish(bestSlots, startPos, lastMatch, out);
    return true;
  }

  void finish(const std::vector<int64_t>& slots, int64_t s, int64_t e, std::vector<int64_t>& out) {
    // buildCapturesFromSlots, pikevm.go:2409-2432
    int groups = nfa.captureCount;
    out.assign(groups * 2, -1);
    out[0] = s; out[1] = e;
    if (track && !slots.empty())
      for (int i = 1; i < groups && i * 2 + 1 < static_cast<int>(slots.size()); i++)
        if (slots[i * 2] >= 0 && slots[i * 2 + 1] >= 0) { out[i * 2] = slots[i * 2]; out[i * 2 + 1] = slots[i * 2 + 1]; }
  }

  bool matchesEmptyAt(int64_t at) {
    std::fill(currSlots.begin(), currSlots.end(), -1);
    visited.clear(); queue.clear();
    addThread({nfa.startAnchored, at}, at, false, 0);
    for (auto& t : queue) if (nfa.isMatch(t.state)) return true;
    return false;
  }
};

}  // namespace

bool PikeVM::searchCaptures(Bytes h, int64_t n, int64_t at, std::vector<int64_t>& slots) {
  if (at > n) return false;
  Pike p(*nfa, h, n, nfa->captureCount * 2);
  if (at == n || n == 0) {  // pikevm.go:2201-2212
    if (p.matchesEmptyAt(at)) { slots.assign(nfa->captureCount * 2, -1); slots[0] = at; slots[1] = at; return true; }
    return false;
  }
  if (nfa->anchored) return p.runAnchored(at, slots);
  return p.runUnanchored(at, slots);
}

bool PikeVM::searchAt(Bytes h, int64_t n, int64_t at, int64_t& s, int64_t& e) {
  if (at > n) return false;
  Pike p(*nfa, h, n, 2);
  std::vector<int64_t> out;
  bool ok;
  if (at == n || n == 0) {
    if (!p.matchesEmptyAt(at)) return false;
    s = e = at; return true;
  }
  if (nfa->anchored) {
    if (at > 0) return false;
    ok = p.runAnchored(at, out);
  } else ok = p.runUnanchored(at, out);
  if (!ok) return false;
  s = out[0]; e = out[1];
  return true;
}

// =============================================================== Teddy
bool Teddy::build(const std::vector<std::vector<uint8_t>>& pats) {
  // NewTeddy teddy.go:189-253 (2..32 patterns, each >= 3 bytes), buildMasks :271-311.
  // 33..64 patterns: NewFatTeddy teddy_fat.go:127-185, buildFatMasks :198-236 — the same tables with 16 buckets
  // (bucket = id mod 16; the reference keeps buckets 8-15 in a second 16-byte lane, here bits 8-15 of a u16), chosen
  // by newTeddyFromSeq teddy.go:629-660.
  if (pats.size() < 2 || pats.size() > 64) return false;
  minLen = static_cast<int>(pats[0].size());
  for (auto& p : pats) {
    if (p.size() < 3) return false;
    minLen = std::min<int>(minLen, static_cast<int>(p.size()));
  }
  fpLen = std::min(2, minLen);
  patterns = pats;
  int nb = pats.size() > 32 ? 16 : std::min<int>(8, static_cast<int>(pats.size()));
  buckets.assign(nb, {});
  std::memset(lo, 0, sizeof lo);
  std::memset(hi, 0, sizeof hi);
  for (size_t id = 0; id < pats.size(); id++) {
    int bucket = static_cast<int>(id % nb);
    buckets[bucket].push_back(static_cast<int>(id));
    uint16_t bit = static_cast<uint16_t>(1u << bucket);
    for (int p = 0; p < fpLen; p++) {
      uint8_t b = pats[id][p];
      lo[p][b & 15] |= bit;
      hi[p][b >> 4] |= bit;
    }
  }
  return true;
}

void Teddy::findCandidate(Bytes h, int64_t n, int64_t& pos, uint16_t& mask) const {
  // findScalarCandidate teddy.go:491-519 (== asm incl. its tail, teddy_ssse3_amd64.s:369-420); Fat: teddy_fat.go:432-468
  for (int64_t i = 0; i + fpLen <= n; i++) {
    uint16_t m = 0xFFFF;
    for (int p = 0; p < fpLen; p++) {
      uint8_t b = h[i + p];
      m &= lo[p][b & 15] & hi[p][b >> 4];
    }
    if (m) { pos = i; mask = m; return; }
  }
  pos = -1; mask = 0;
}

bool Teddy::findMatch(Bytes hay, int64_t len, int64_t start, int64_t& s, int64_t& e) const {
  if (start < 0 || start >= len) return false;
  Bytes h = hay + start;
  int64_t n = len - start;
  if (n < 16) {  // findMatchScalar teddy.go:447-458 (Fat: teddy_fat.go:418-429): position-major, pattern-id-minor
    for (int64_t i = 0; i < n - minLen + 1; i++)
      for (auto& p : patterns) {
        int64_t pl = static_cast<int64_t>(p.size());
        if (i + pl <= n && std::memcmp(h + i, p.data(), p.size()) == 0) { s = start + i; e = s + pl; return true; }
      }
    return false;
  }
  int64_t acc = 0, pos; uint16_t mask;   // FindMatch teddy.go:391-445, Fat: teddy_fat.go:346-392 (bits.TrailingZeros16)
  findCandidate(h, n, pos, mask);
  while (pos != -1) {
    while (mask) {
      int bucket = __builtin_ctz(mask);
      mask &= static_cast<uint16_t>(~(1u << bucket));
      // verifyBucket teddy.go:532-550 on haystack[acc:]
      int64_t rem = n - acc;
      if (pos >= 0 && pos < rem && bucket < static_cast<int>(buckets.size())) {
        for (int id : buckets[bucket]) {
          const auto& p = patterns[id];
          int64_t end = pos + static_cast<int64_t>(p.size());
          if (end <= rem && std::memcmp(h + acc + pos, p.data(), p.size()) == 0) {
            s = start + acc + pos; e = s + static_cast<int64_t>(p.size()); return true;
          }
        }
      }
    }
    int64_t nextStart = acc + pos + 1;
    if (nextStart >= n) break;
    acc = nextStart;
    findCandidate(h + acc, n - acc, pos, mask);
  }
  return false;
}

// =============================================================== CharClassSearcher
void CharClassSearcher::findAll(Bytes h, int64_t n, std::vector<int64_t>& out) const {
  bool matching = false;
  int64_t matchStart = 0;
  for (int64_t i = 0; i < n; i++) {
    bool m = membership[h[i]];
    if (!matching) {
      if (m) { matchStart = i; matching = true; }
    } else if (!m) {
      if (i - matchStart >= minMatch) { out.push_back(matchStart); out.push_back(i); }
      matching = false;
    }
  }
  if (matching && n - matchStart >= minMatch) { out.push_back(matchStart); out.push_back(n); }
}

bool CharClassSearcher::searchAt(Bytes h, int64_t n, int64_t at, int64_t& s, int64_t& e) const {
  for (;;) {
    if (at >= n) return false;
    int64_t start = -1;
    for (int64_t i = at; i < n; i++) if (membership[h[i]]) { start = i; break; }
    if (start == -1) return false;
    int64_t end = start + 1;
    while (end < n && membership[h[end]]) end++;
    if (end - start < minMatch) { at = start + 1; continue; }
    s = start; e = end;
    return true;
  }
}

int64_t memchrDigitAt(Bytes h, int64_t n, int64_t at) {
  if (at < 0 || at >= n) return -1;
  for (int64_t i = at; i < n; i++) if (h[i] >= '0' && h[i] <= '9') return i;
  return -1;
}

}  // namespace orc
