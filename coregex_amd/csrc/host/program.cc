// See program.h.
#include "program.h"
#include "fsm.h"
#include "lookdfa.h"
#include "../device/bt.hpp"

#include <algorithm>
#include <array>
#include <cstring>
#include <functional>
#include <map>

namespace cxg {

void buildLiteralImage(cxg_program* p, const std::vector<std::vector<uint8_t>>& lits, size_t min_count, bool fold = false);
bool makeLiteralAux(const std::vector<std::vector<uint8_t>>& lits, size_t min_count, std::vector<uint8_t>& aux, std::string& why, bool fold = false);

namespace {

struct Closure {
  const cxg_nfa& n;
  std::vector<uint32_t> mark;   // generation stamps
  uint32_t gen = 0;
  std::vector<uint32_t> stack;
  explicit Closure(const cxg_nfa& nfa) : n(nfa), mark(nfa.n_states, 0) {}
  void begin() { gen++; }
  // epsilonClosureInto, builder.go:245-293
  void into(std::vector<uint32_t>& out, uint32_t seed) {
    stack.clear();
    stack.push_back(seed);
    while (!stack.empty()) {
      uint32_t cur = stack.back();
      stack.pop_back();
      if (cur == CXG_NFA_INVALID || cur >= n.n_states) continue;
      if (mark[cur] == gen) continue;
      mark[cur] = gen;
      out.push_back(cur);
      const cxg_nfa_state& s = n.states[cur];
      switch (s.kind) {
        case CXG_NFA_EPSILON: case CXG_NFA_CAPTURE:
          if (s.next != CXG_NFA_INVALID) stack.push_back(s.next);
          break;
        case CXG_NFA_SPLIT:
          if (s.right != CXG_NFA_INVALID) stack.push_back(s.right);
          if (s.left != CXG_NFA_INVALID) stack.push_back(s.left);
          break;
        default: break;
      }
    }
  }
};

}  // namespace

void alphabetOf(const cxg_nfa& nfa, bool in[256]) {
  std::memset(in, 0, 256);
  // the unanchored prefix (compile.go:1633-1650) is Split(pattern, [00-FF] -> self): not pattern alphabet
  uint32_t prefixAny = CXG_NFA_INVALID;
  if (nfa.start_unanchored != nfa.start_anchored && nfa.start_unanchored < nfa.n_states) {
    const cxg_nfa_state& sp = nfa.states[nfa.start_unanchored];
    if (sp.kind == CXG_NFA_SPLIT && sp.right < nfa.n_states) {
      const cxg_nfa_state& a = nfa.states[sp.right];
      if (a.kind == CXG_NFA_BYTE_RANGE && a.lo == 0 && a.hi == 255 && a.next == nfa.start_unanchored) prefixAny = sp.right;
    }
  }
  for (uint32_t i = 0; i < nfa.n_states; i++) {
    if (i == prefixAny) continue;
    const cxg_nfa_state& s = nfa.states[i];
    if (s.kind == CXG_NFA_BYTE_RANGE) for (int b = s.lo; b <= s.hi; b++) in[b] = true;
    else if (s.kind == CXG_NFA_SPARSE)
      for (uint32_t k = 0; k < s.trans_len; k++) {
        const cxg_nfa_trans& t = nfa.trans[s.trans_off + k];
        for (int b = t.lo; b <= t.hi; b++) in[b] = true;
      }
  }
}

Dfa determinize(const cxg_nfa& nfa, uint32_t startState, bool breakAtMatch, uint32_t maxStates) {
  for (uint32_t i = 0; i < nfa.n_states; i++)
    if (nfa.states[i].kind == CXG_NFA_LOOK) throw BuildError{CXG_E_UNSUPPORTED, "look-around assertion in NFA"};
  // byte classes (nfa/alphabet.go:100-166): determinize once per class representative
  bool boundary[256] = {false};
  auto mark = [&](int lo, int hi) { if (lo > 0) boundary[lo - 1] = true; boundary[hi] = true; };
  for (uint32_t i = 0; i < nfa.n_states; i++) {
    const cxg_nfa_state& s = nfa.states[i];
    if (s.kind == CXG_NFA_BYTE_RANGE) mark(s.lo, s.hi);
    else if (s.kind == CXG_NFA_SPARSE) for (uint32_t k = 0; k < s.trans_len; k++) mark(nfa.trans[s.trans_off + k].lo, nfa.trans[s.trans_off + k].hi);
  }
  std::vector<int> repOf(256);
  std::vector<int> reps;
  { int rep = 0; for (int b = 0; b < 256; b++) { if (b == 0 || boundary[b - 1]) { rep = b; reps.push_back(b); } repOf[b] = rep; } }

  Closure cl(nfa);
  std::map<std::vector<uint32_t>, uint32_t> ids;   // ordered NFA list -> provisional id
  std::vector<std::vector<uint32_t>> sets;
  std::vector<std::vector<uint32_t>> next;          // [id][256] provisional
  auto intern = [&](std::vector<uint32_t>&& set) -> uint32_t {
    auto it = ids.find(set);
    if (it != ids.end()) return it->second;
    uint32_t id = static_cast<uint32_t>(sets.size());
    if (id >= maxStates) throw BuildError{CXG_E_UNSUPPORTED, "DFA exceeds the LDS state budget"};
    ids.emplace(set, id);
    sets.push_back(std::move(set));
    next.emplace_back(256, 0u);
    return id;
  };
  intern({});  // id 0 = dead (empty set)
  std::vector<uint32_t> st;
  cl.begin();
  cl.into(st, startState);
  uint32_t start = intern(std::move(st));
  auto holdsMatch = [&](const std::vector<uint32_t>& s) { for (uint32_t x : s) if (nfa.states[x].kind == CXG_NFA_MATCH) return true; return false; };
  for (uint32_t cur = 1; cur < sets.size(); cur++) {
    const bool brk = breakAtMatch && holdsMatch(sets[cur]);
    for (int rep : reps) {
      std::vector<uint32_t> out;
      cl.begin();
      const std::vector<uint32_t> src = sets[cur];  // copy: sets may grow
      for (uint32_t sid : src) {  // moveWithWordContextBreak, builder.go:183-242
        const cxg_nfa_state& s = nfa.states[sid];
        if (brk && s.kind == CXG_NFA_MATCH) break;
        if (s.kind == CXG_NFA_BYTE_RANGE) {
          if (rep >= s.lo && rep <= s.hi) cl.into(out, s.next);
        } else if (s.kind == CXG_NFA_SPARSE) {
          for (uint32_t k = 0; k < s.trans_len; k++) {
            const cxg_nfa_trans& t = nfa.trans[s.trans_off + k];
            if (rep >= t.lo && rep <= t.hi) cl.into(out, t.next);
          }
        }
      }
      uint32_t to = out.empty() ? 0u : intern(std::move(out));
      next[cur][rep] = to;
    }
  }
  // renumber: dead 0, non-accepting, then accepting
  const uint32_t n = static_cast<uint32_t>(sets.size());
  std::vector<uint32_t> perm(n, 0);
  uint32_t k = 1;
  for (uint32_t i = 1; i < n; i++) if (!holdsMatch(sets[i])) perm[i] = k++;
  const uint32_t firstAccept = k;
  for (uint32_t i = 1; i < n; i++) if (holdsMatch(sets[i])) perm[i] = k++;
  Dfa d;
  d.nstates = n;
  d.start = perm[start];
  d.firstAccept = firstAccept;
  d.table.assign(static_cast<size_t>(n) * 256, 0);
  for (uint32_t i = 1; i < n; i++)
    for (int b = 0; b < 256; b++) d.table[static_cast<size_t>(perm[i]) * 256 + b] = static_cast<uint8_t>(perm[next[i][repOf[b]]]);
  return d;
}

// The reference's lazy DFA files a state under its SORTED NFA set plus the from-word and match-delay flags
// (dfa/lazy/state.go:329-373) but steps through the set in the insertion order of whichever variant was determinized
// first, breaking at the first match state (builder.go:183-242).  When two different priority orders of one set are
// reachable AND behave differently (`a?(a|b)`: after "b" and after "a"), what the reference returns depends on what
// its cache saw earlier — in this call or a previous one on the same Regex.  There is no single answer to reproduce,
// so such programs are refused.  Explores the reference's states without conflating them — (ordered list, from-word,
// source-held-a-match) — and minimises that automaton (Moore); orders filed under one key that land in one
// equivalence class are harmless (`a?c`: the cache may keep either, every continuation reports the same matches),
// and then the eager DFA above (keyed by the ordered list) is the reference's automaton whatever its history.
bool priorityOrderConflict(const cxg_nfa& nfa, std::initializer_list<uint32_t> starts) {
  bool boundary[256] = {false};
  auto mark = [&](int lo, int hi) { if (lo > 0) boundary[lo - 1] = true; boundary[hi] = true; };
  for (uint32_t i = 0; i < nfa.n_states; i++) {
    const cxg_nfa_state& s = nfa.states[i];
    if (s.kind == CXG_NFA_BYTE_RANGE) mark(s.lo, s.hi);
    else if (s.kind == CXG_NFA_SPARSE) for (uint32_t k = 0; k < s.trans_len; k++) mark(nfa.trans[s.trans_off + k].lo, nfa.trans[s.trans_off + k].hi);
  }
  mark('0', '9'); mark('A', 'Z'); mark('_', '_'); mark('a', 'z');   // the from-word flag splits classes too
  std::vector<int> reps;
  for (int b = 0; b < 256; b++) if (b == 0 || boundary[b - 1]) reps.push_back(b);
  auto isWord = [](int b) { return (b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z') || b == '_' || (b >= 'a' && b <= 'z'); };
  auto holdsMatch = [&](const std::vector<uint32_t>& v, size_t n) { for (size_t i = 0; i < n; i++) if (nfa.states[v[i]].kind == CXG_NFA_MATCH) return true; return false; };

  constexpr size_t kMaxTuples = 4096;
  Closure cl(nfa);
  std::map<std::vector<uint32_t>, uint32_t> ids;                    // ordered list + flags word -> tuple id
  std::map<std::vector<uint32_t>, std::vector<uint32_t>> filedAs;   // sorted set + flags word -> tuple ids filed there
  std::vector<std::vector<uint32_t>> tuples;                        // ordered list, then the flags word
  std::vector<std::vector<int32_t>> next;                           // [tuple][rep index], -1 = dead
  bool anyConflict = false;
  auto intern = [&](std::vector<uint32_t>&& list, uint32_t flagsWord) -> int32_t {
    std::vector<uint32_t> key(list);
    std::sort(key.begin(), key.end());
    key.push_back(0x80000000u | flagsWord);
    list.push_back(0x80000000u | flagsWord);
    auto it = ids.find(list);
    if (it != ids.end()) return static_cast<int32_t>(it->second);
    const uint32_t id = static_cast<uint32_t>(tuples.size());
    if (id >= kMaxTuples) throw BuildError{CXG_E_UNSUPPORTED, "DFA exceeds the LDS state budget"};
    std::vector<uint32_t>& filed = filedAs[key];
    if (!filed.empty()) anyConflict = true;
    filed.push_back(id);
    ids.emplace(list, id);
    tuples.push_back(std::move(list));
    next.emplace_back(reps.size(), -1);
    return static_cast<int32_t>(id);
  };
  for (uint32_t st : starts)
    for (uint32_t w = 0; w < 2; w++) {
      std::vector<uint32_t> set;
      cl.begin();
      cl.into(set, st);
      intern(std::move(set), w);
    }
  for (size_t cur = 0; cur < tuples.size(); cur++) {
    const std::vector<uint32_t> src = tuples[cur];   // copy: tuples grows
    const size_t n = src.size() - 1;
    const bool srcMatch = holdsMatch(src, n);
    for (size_t ri = 0; ri < reps.size(); ri++) {
      const int rep = reps[ri];
      std::vector<uint32_t> out;
      cl.begin();
      for (size_t i = 0; i < n; i++) {
        const cxg_nfa_state& s = nfa.states[src[i]];
        if (srcMatch && s.kind == CXG_NFA_MATCH) break;
        if (s.kind == CXG_NFA_BYTE_RANGE) {
          if (rep >= s.lo && rep <= s.hi) cl.into(out, s.next);
        } else if (s.kind == CXG_NFA_SPARSE) {
          for (uint32_t k = 0; k < s.trans_len; k++) {
            const cxg_nfa_trans& t = nfa.trans[s.trans_off + k];
            if (rep >= t.lo && rep <= t.hi) cl.into(out, t.next);
          }
        }
      }
      if (out.empty() && !srcMatch) continue;   // dead
      next[cur][ri] = intern(std::move(out), (isWord(rep) ? 1u : 0u) | (srcMatch ? 2u : 0u));
    }
  }
  if (!anyConflict) return false;
  // Moore minimisation.  Observable per state: the delayed match flag and "holds a match" (end of input) — both
  // functions of the filing key, so states filed together start in one class.
  const size_t nt = tuples.size();
  std::vector<uint32_t> cls(nt);
  for (size_t i = 0; i < nt; i++) {
    const size_t n = tuples[i].size() - 1;
    cls[i] = ((tuples[i][n] & 2u) ? 1u : 0u) | (holdsMatch(tuples[i], n) ? 2u : 0u);
  }
  size_t nclasses = 0;
  for (;;) {
    std::map<std::vector<uint32_t>, uint32_t> sig;
    std::vector<uint32_t> ncls(nt);
    for (size_t i = 0; i < nt; i++) {
      std::vector<uint32_t> k;
      k.reserve(reps.size() + 1);
      k.push_back(cls[i]);
      for (size_t ri = 0; ri < reps.size(); ri++) k.push_back(next[i][ri] < 0 ? 0xFFFFFFFFu : cls[static_cast<size_t>(next[i][ri])]);
      ncls[i] = sig.emplace(std::move(k), static_cast<uint32_t>(sig.size())).first->second;
    }
    cls.swap(ncls);
    if (sig.size() == nclasses) break;
    nclasses = sig.size();
  }
  for (const auto& kv : filedAs)
    for (uint32_t id : kv.second)
      if (cls[id] != cls[kv.second[0]]) return true;
  return false;
}

void refuseOrderConflict(const cxg_nfa& nfa, std::initializer_list<uint32_t> starts) {
  if (priorityOrderConflict(nfa, starts))
    throw BuildError{CXG_E_UNSUPPORTED, "reference DFA cache conflates priority orders of one NFA set (result depends on cache history)"};
}

HostNfa reverseOf(const cxg_nfa& fwd) {
  // R(t) == "the forward run is at state t here".  Reading byte b backwards moves R(t) -> R(s) for
  // every byte state s with s.next == t that accepts b; epsilon edges are reversed.  The reverse
  // automaton starts at R(Match) and accepts at R(start_anchored).  The reference builds the same
  // language in nfa/reverse.go; state order is irrelevant because the reverse DFA runs without
  // break-at-match (meta/compile.go:193-194) and therefore reports the set-theoretic minimum start.
  const uint32_t N = fwd.n_states;
  std::vector<std::vector<cxg_nfa_trans>> byteIn(N);
  std::vector<std::vector<uint32_t>> epsIn(N);
  std::vector<std::vector<std::pair<uint32_t, uint8_t>>> lookIn(N);   // (source state, nfa.Look): an assertion holds at a POSITION, so it reads the same backwards
  std::vector<uint8_t> skip(N, 0);
  if (fwd.start_unanchored != fwd.start_anchored && fwd.start_unanchored < N) {
    skip[fwd.start_unanchored] = 1;
    const cxg_nfa_state& sp = fwd.states[fwd.start_unanchored];
    if (sp.kind == CXG_NFA_SPLIT && sp.right < N) skip[sp.right] = 1;
  }
  uint32_t fwdMatch = CXG_NFA_INVALID;
  for (uint32_t s = 0; s < N; s++) {
    const cxg_nfa_state& st = fwd.states[s];
    if (st.kind == CXG_NFA_MATCH) fwdMatch = s;
    if (skip[s]) continue;
    switch (st.kind) {
      case CXG_NFA_BYTE_RANGE: if (st.next < N) byteIn[st.next].push_back({st.lo, st.hi, 0, s}); break;
      case CXG_NFA_SPARSE:
        for (uint32_t k = 0; k < st.trans_len; k++) { const cxg_nfa_trans& t = fwd.trans[st.trans_off + k]; if (t.next < N) byteIn[t.next].push_back({t.lo, t.hi, 0, s}); }
        break;
      case CXG_NFA_SPLIT:
        if (st.left < N) epsIn[st.left].push_back(s);
        if (st.right < N) epsIn[st.right].push_back(s);
        break;
      case CXG_NFA_EPSILON: case CXG_NFA_CAPTURE: if (st.next < N) epsIn[st.next].push_back(s); break;
      case CXG_NFA_LOOK: if (st.next < N) lookIn[st.next].push_back({s, st.lo}); break;
      default: break;
    }
  }
  HostNfa r;
  auto blank = [](uint8_t kind) { cxg_nfa_state s; std::memset(&s, 0, sizeof s); s.kind = kind; s.next = s.left = s.right = CXG_NFA_INVALID; return s; };
  r.states.assign(N, blank(CXG_NFA_FAIL));
  auto add = [&](cxg_nfa_state s) { r.states.push_back(s); return static_cast<uint32_t>(r.states.size() - 1); };
  const uint32_t revMatch = add(blank(CXG_NFA_MATCH));
  for (uint32_t t = 0; t < N; t++) {
    if (skip[t]) continue;
    std::vector<uint32_t> alts;
    if (t == fwd.start_anchored) alts.push_back(revMatch);
    if (!byteIn[t].empty()) {
      cxg_nfa_state sp = blank(CXG_NFA_SPARSE);
      sp.trans_off = static_cast<uint32_t>(r.trans.size());
      sp.trans_len = static_cast<uint32_t>(byteIn[t].size());
      for (auto& tr : byteIn[t]) r.trans.push_back(tr);
      alts.push_back(add(sp));
    }
    for (uint32_t s : epsIn[t]) alts.push_back(s);
    for (auto& lk : lookIn[t]) { cxg_nfa_state ls = blank(CXG_NFA_LOOK); ls.lo = lk.second; ls.next = lk.first; alts.push_back(add(ls)); }
    if (alts.empty()) continue;
    uint32_t chain = alts.back();
    for (size_t i = alts.size() - 1; i-- > 0;) { cxg_nfa_state sp = blank(CXG_NFA_SPLIT); sp.left = alts[i]; sp.right = chain; chain = add(sp); }
    r.states[t] = blank(CXG_NFA_EPSILON);
    r.states[t].next = chain;
  }
  r.startAnchored = r.startUnanchored = fwdMatch;
  r.captureCount = 1;
  r.alwaysAnchored = true;
  return r;
}

namespace {

constexpr uint32_t kMaxDfaStates = 224;  // u8 ids; 224*260 B = 58 KB of LDS leaves room for the tile

void appendTable(std::vector<uint8_t>& blob, const Dfa& d, uint32_t& off) {
  off = static_cast<uint32_t>(blob.size());
  blob.insert(blob.end(), d.table.begin(), d.table.end());
  while (blob.size() % 16) blob.push_back(0);
}

// Follows the anchored DFA `d` from its start while it is a chain of run(class+) / byte(class) steps whose
// classes have a SWAR-friendly form (contiguous ASCII range).  `complete`: the chain IS the DFA — it ended in
// the only accepting state, that state either loops on the last run's class or is terminal, nothing else
// leaves it, and the pattern alphabet (non-sync bytes of `info`) equals the union of the chain classes; then
// a position satisfying the chain is a match and its end follows from the same steps.  `ordered`: no run's
// class meets the class of the step before it, so two matches starting at different candidates never reach
// the same step at the same byte: the k-th start pairs with the k-th end (scan_chain_wave.hip).
void extractChain(const Dfa& d, const uint8_t info[256], cxgdev::ChainAux& chain, bool& complete, bool& ordered) {
  std::memset(&chain, 0, sizeof chain);
  complete = ordered = false;
  struct Cls { uint8_t kind, lo, hi, nr, rlo[cxgdev::kChainMaxRanges], rhi[cxgdev::kChainMaxRanges]; };
  auto classOf = [&](const bool in[256], Cls& c) {       // a class is a union of at most 4 ASCII ranges
    std::memset(&c, 0, sizeof c);
    int nr = 0;
    for (int b = 0; b < 256; b++) {
      if (!in[b] || (b > 0 && in[b - 1])) continue;
      int e = b;
      while (e + 1 < 256 && in[e + 1]) e++;
      if (e > 127 || nr >= cxgdev::kChainMaxRanges) return false;
      c.rlo[nr] = static_cast<uint8_t>(b); c.rhi[nr] = static_cast<uint8_t>(e); nr++;
    }
    if (nr == 0) return false;
    c.nr = static_cast<uint8_t>(nr); c.lo = c.rlo[0]; c.hi = c.rhi[nr - 1];
    if (nr > 1) c.kind = cxgdev::kClsSet;
    else c.kind = (c.lo == '0' && c.hi == '9') ? cxgdev::kClsDigit : (c.lo == c.hi ? cxgdev::kClsByte : cxgdev::kClsRange);
    return true;
  };
  uint32_t q = d.start;
  for (int b = 0; b < 256; b++)                          // `x*...`: an optional leading run is not a chain step
    if (d.table[static_cast<size_t>(q) * 256 + b] == q) return;
  for (int step = 0; step < cxgdev::kChainMaxOps - 1; step++) {
    if (q >= d.firstAccept) break;                       // a match may end here: later steps are not necessary
    int target = -1; bool branching = false;
    bool F[256] = {false};
    for (int b = 0; b < 256; b++) {
      const uint32_t t = d.table[static_cast<size_t>(q) * 256 + b];
      if (t == 0 || t == q) continue;
      if (target < 0) target = static_cast<int>(t);
      if (static_cast<int>(t) != target) { branching = true; break; }
      F[b] = true;
    }
    if (branching || target < 0) break;
    bool loopT[256] = {false}; bool anyLoop = false, same = true;
    for (int b = 0; b < 256; b++) { loopT[b] = d.table[static_cast<size_t>(target) * 256 + b] == static_cast<uint32_t>(target); anyLoop = anyLoop || loopT[b]; }
    for (int b = 0; b < 256; b++) if (loopT[b] != F[b]) same = false;
    Cls cl;
    if (!classOf(F, cl)) break;
    if (anyLoop && !same) break;                         // loops on a different class: not a plain run
    int ci = -1;
    for (uint32_t k = 0; k < chain.ncls; k++)
      if (chain.cls_kind[k] == cl.kind && chain.cls_lo[k] == cl.lo && chain.cls_hi[k] == cl.hi && chain.cls_nr[k] == cl.nr &&
          std::memcmp(chain.cls_rlo[k], cl.rlo, sizeof cl.rlo) == 0 && std::memcmp(chain.cls_rhi[k], cl.rhi, sizeof cl.rhi) == 0) ci = static_cast<int>(k);
    if (ci < 0) {
      if (chain.ncls >= cxgdev::kChainMaxCls) break;
      ci = static_cast<int>(chain.ncls++);
      chain.cls_kind[ci] = cl.kind; chain.cls_lo[ci] = cl.lo; chain.cls_hi[ci] = cl.hi; chain.cls_nr[ci] = cl.nr;
      std::memcpy(chain.cls_rlo[ci], cl.rlo, sizeof cl.rlo); std::memcpy(chain.cls_rhi[ci], cl.rhi, sizeof cl.rhi);
    }
    chain.op_kind[chain.nops] = anyLoop ? cxgdev::kChainRun : cxgdev::kChainByte;
    chain.op_cls[chain.nops] = static_cast<uint8_t>(ci);
    chain.nops++;
    q = static_cast<uint32_t>(target);
  }
  if (chain.nops == 0) return;
  for (uint32_t k = 0; k < chain.nops; k++) {              // packed copy of the steps for the kernel's scalar registers
    if (chain.op_kind[k] == cxgdev::kChainRun) chain.run_bits |= 1ull << k;
    const uint64_t c = chain.op_cls[k] & 3u;
    if (k < 32) chain.cls2_lo |= c << (2 * k); else chain.cls2_hi |= c << (2 * (k - 32));
  }
  complete = q >= d.firstAccept && d.firstAccept == d.nstates - 1;
  if (complete) {
    const bool lastRun = chain.op_kind[chain.nops - 1] == cxgdev::kChainRun;
    for (int b = 0; b < 256 && complete; b++) {
      const uint32_t t = d.table[static_cast<size_t>(q) * 256 + b];
      if (t != 0 && t != q) complete = false;
      if (t == q && !(lastRun && cxgdev::chain_class_has(chain, chain.op_cls[chain.nops - 1], static_cast<uint32_t>(b)))) complete = false;
    }
    for (int b = 0; b < 256 && complete; b++) {
      bool inChain = false;
      for (uint32_t k = 0; k < chain.ncls; k++) inChain = inChain || cxgdev::chain_class_has(chain, static_cast<int>(k), static_cast<uint32_t>(b));
      const bool inAlphabet = !(info[b] & cxgdev::kInfoSync);
      if (inChain != inAlphabet) complete = false;
    }
  }
  ordered = true;
  for (uint32_t k = 1; k < chain.nops; k++) {
    if (chain.op_kind[k] != cxgdev::kChainRun) continue;
    for (int b = 0; b < 256; b++)
      if (cxgdev::chain_class_has(chain, chain.op_cls[k], static_cast<uint32_t>(b)) && cxgdev::chain_class_has(chain, chain.op_cls[k - 1], static_cast<uint32_t>(b))) ordered = false;
  }
  // A chain that begins with a run takes its starts at run starts.  FindAll resumes at the end of the previous
  // match, so a match may also start INSIDE a run of the first class when the previous match ended there
  // (`z+\.\w\w` on "z.azz.bc": [0,4] then [4,8]).  That needs the last byte of a match to be in the first class
  // with another first-class byte behind it: excluded when the last class misses the first class, or when the chain
  // ends with a run whose class covers the first class (the match then stops at a byte outside both).  Otherwise the
  // chain kernel looks for such an end in every tile (rare in real text: `key=12next=3`) and hands the scan to the
  // table-walking kernel when it finds one.
  if (chain.nops >= 1 && chain.op_kind[0] == cxgdev::kChainRun) {
    const int first = chain.op_cls[0], last = chain.op_cls[chain.nops - 1];
    bool meet = false, firstOutsideLast = false;
    for (int b = 0; b < 256; b++) {
      const bool inF = cxgdev::chain_class_has(chain, first, static_cast<uint32_t>(b)), inL = cxgdev::chain_class_has(chain, last, static_cast<uint32_t>(b));
      meet = meet || (inF && inL);
      firstOutsideLast = firstOutsideLast || (inF && !inL);
    }
    const bool lastRun = chain.op_kind[chain.nops - 1] == cxgdev::kChainRun;
    if (meet && (!lastRun || firstOutsideLast)) chain.restart_check = 1;   // exact as long as no match ends inside a first-class run: checked per tile
  }
}

}  // namespace

// Every match begins with one of at most 32 byte strings of one length D, 3 <= D <= 8: all paths of length D from the
// anchored start, no match shorter.  The literal kernels find the occurrences, the anchored DFA gives each its end
// (the reference runs the same split as prefilter + DFA, a18); leftmost-first, non-overlapping, as the DFA pair would
// answer.  The largest D that keeps the set within 32 wins (fewest false candidates).
bool requiredPrefixes(const Dfa& anch, std::vector<std::vector<uint8_t>>& lits) {
  lits.clear();
  if (anch.nstates > 64) return false;               // the walk table lives in LDS: 16 KiB at most
  struct Item { uint32_t q; std::vector<uint8_t> bytes; };
  std::vector<Item> cur{{anch.start, {}}};
  for (int depth = 0; depth < 8; depth++) {
    std::vector<Item> nxt;
    bool ok = true;
    for (const Item& it : cur) {
      if (it.q >= anch.firstAccept) { ok = false; break; }          // a match of `depth` bytes: no longer prefix
      for (int b = 0; b < 256 && ok; b++) {
        const uint32_t t = anch.table[static_cast<size_t>(it.q) * 256 + b];
        if (!t) continue;
        Item n{t, it.bytes};
        n.bytes.push_back(static_cast<uint8_t>(b));
        nxt.push_back(std::move(n));
        if (nxt.size() > 32) ok = false;
      }
      if (!ok) break;
    }
    if (!ok || nxt.empty()) break;
    cur.swap(nxt);
    if (depth + 1 >= 3) { lits.clear(); for (const Item& it : cur) lits.push_back(it.bytes); }
  }
  return !lits.empty();
}

// Appends the literal tables + the anchored DFA as the aux section of a DFA-pair image (kFlagPrefixLiteral).
void appendPrefixAux(std::vector<uint8_t>& blob, cxgdev::BlobHeader& h, const std::vector<std::vector<uint8_t>>& lits, const Dfa& anch) {
  std::vector<uint8_t> aux;
  std::string why;
  if (!makeLiteralAux(lits, 1, aux, why) || aux.size() > 2048) return;
  cxgdev::TeddyAux ax;
  std::memcpy(&ax, aux.data(), sizeof ax);
  ax.dfa_off = static_cast<uint32_t>(aux.size());
  ax.dfa_states = anch.nstates; ax.dfa_start = anch.start; ax.dfa_first_accept = anch.firstAccept;
  aux.insert(aux.end(), anch.table.begin(), anch.table.end());
  while (aux.size() % 16) aux.push_back(0);
  std::memcpy(aux.data(), &ax, sizeof ax);
  while (blob.size() % 16) blob.push_back(0);
  h.aux_off = static_cast<uint32_t>(blob.size());
  h.aux_len = static_cast<uint32_t>(aux.size());
  blob.insert(blob.end(), aux.begin(), aux.end());
  h.flags |= cxgdev::kFlagPrefixLiteral;
}

// A caller-supplied NFA (cgo shim: flattenNFA) is foreign data: every index is checked before anything walks it.
// nfa.InvalidState (0xFFFFFFFF) is a legal "no target" in every next/left/right field (nfa/nfa.go:62-64).
bool validateNfa(const cxg_nfa& nfa, std::string& why) {
  auto bad = [&](uint32_t i, const char* what) { why = "malformed NFA: state " + std::to_string(i) + ": " + what; return false; };
  if (!nfa.states || nfa.n_states == 0) { why = "malformed NFA: no states"; return false; }
  if (nfa.n_states > (1u << 20)) { why = "malformed NFA: more than 2^20 states"; return false; }
  if (nfa.n_trans && !nfa.trans) { why = "malformed NFA: n_trans > 0 with a null transition array"; return false; }
  if (nfa.start_anchored >= nfa.n_states || nfa.start_unanchored >= nfa.n_states) { why = "malformed NFA: start state out of range"; return false; }
  if (nfa.capture_count == 0 || nfa.capture_count > 1024) { why = "malformed NFA: capture_count must be 1..1024 (group 0 included)"; return false; }
  auto target = [&](uint32_t t) { return t == CXG_NFA_INVALID || t < nfa.n_states; };
  for (uint32_t i = 0; i < nfa.n_states; i++) {
    const cxg_nfa_state& s = nfa.states[i];
    switch (s.kind) {
      case CXG_NFA_MATCH: case CXG_NFA_FAIL: break;
      case CXG_NFA_BYTE_RANGE:
        if (s.lo > s.hi) return bad(i, "byte range with lo > hi");
        if (!target(s.next)) return bad(i, "next out of range");
        break;
      case CXG_NFA_SPARSE:
        if (s.trans_off > nfa.n_trans || s.trans_len > nfa.n_trans - s.trans_off) return bad(i, "sparse transitions outside the transition array");
        for (uint32_t k = 0; k < s.trans_len; k++) {
          const cxg_nfa_trans& t = nfa.trans[s.trans_off + k];
          if (t.lo > t.hi) return bad(i, "sparse transition with lo > hi");
          if (!target(t.next)) return bad(i, "sparse transition target out of range");
        }
        break;
      case CXG_NFA_SPLIT:
        if (!target(s.left) || !target(s.right)) return bad(i, "split target out of range");
        break;
      case CXG_NFA_EPSILON:
        if (!target(s.next)) return bad(i, "next out of range");
        break;
      case CXG_NFA_LOOK:
        if (!target(s.next)) return bad(i, "next out of range");
        if (s.lo > 5) return bad(i, "unknown look-around kind (nfa.Look is 0..5)");
        break;
      case CXG_NFA_CAPTURE:
        if (!target(s.next)) return bad(i, "next out of range");
        // capture_count == 1: the caller wants spans only (the FindAllIndex / Count program of an NFA WITH groups — what the cgo shim
        // passes, integration/go/meta/findall_hip.go nfaProgram): every capture state is then an epsilon to the span program
        if (nfa.capture_count > 1 && s.cap_index >= nfa.capture_count) return bad(i, "capture index >= capture_count");
        break;
      default: return bad(i, "unknown state kind");
    }
  }
  return true;
}

// Literals between assertions: `\berror\b`, `(?m)^(GET|POST|PUT)`, `\b(warn|fatal)\b` — UseNFA programs in the reference (a small
// pattern with a word boundary or a multi-line anchor, meta/strategy.go:1503), i.e. its PikeVM: leftmost-first over an assertion,
// an alternation of plain literals, an assertion.  For a prefix-free literal set at most one alternative matches at a position, so
// the answer is: the occurrences of the literals around which both assertions hold, leftmost first, non-overlapping — the
// literal kernel's candidates with one more test in their verification (scan_teddy_wave.hip), instead of the look-around
// transducer over every byte.  Reads the shape off the NFA (so that cxg_program_from_nfa gets it as well): start, captures /
// epsilons, an optional LOOK, a tree of splits over chains of single-byte states, an optional LOOK, Match.
bool wrappedLiterals(const cxg_nfa& nfa, std::vector<std::vector<uint8_t>>& lits, uint32_t& looks) {
  lits.clear();
  looks = 0;
  const uint32_t N = nfa.n_states;
  auto skip = [&](uint32_t q) {                                      // over epsilons and captures
    for (uint32_t g = 0; g < N && q < N; g++) {
      const cxg_nfa_state& x = nfa.states[q];
      if (x.kind == CXG_NFA_EPSILON || x.kind == CXG_NFA_CAPTURE) q = x.next; else break;
    }
    return q;
  };
  uint32_t cur = skip(nfa.start_anchored);
  if (cur >= N) return false;
  uint32_t pre = 0, post = 0;
  if (nfa.states[cur].kind == CXG_NFA_LOOK) { if (nfa.states[cur].lo < 2) return false; pre = nfa.states[cur].lo + 1u; cur = skip(nfa.states[cur].next); }
  if (cur >= N) return false;
  uint32_t terminal = CXG_NFA_INVALID;
  std::vector<uint8_t> path;
  size_t visited = 0, nFolded = 0, nPlainLetters = 0;
  bool ok = true;
  std::function<void(uint32_t)> dfs = [&](uint32_t q) {
    if (!ok) return;
    q = skip(q);
    if (q >= N || ++visited > 8192 || path.size() > 255) { ok = false; return; }
    const cxg_nfa_state& x = nfa.states[q];
    switch (x.kind) {
      case CXG_NFA_SPLIT: {
        // `(?i)e`: a split over the two cases of one letter that join again is ONE folded character, not two alternatives
        // (`(?i)s`: [Ss] and the two bytes of U+017F beside them — the pair folds, the rest stays an alternative).
        std::vector<uint32_t> branches, todo{q};
        size_t steps = 0;
        while (!todo.empty() && branches.size() <= 128) {             // the split tree under q, in priority order
          if (++steps > 1024) { ok = false; return; }                // (a cycle of splits — `(?:a*)*` — is no literal tree: found by the sanitizer run)
          const uint32_t b = skip(todo.back());
          todo.pop_back();
          if (b >= N) { ok = false; return; }
          if (nfa.states[b].kind == CXG_NFA_SPLIT) { todo.push_back(nfa.states[b].right); todo.push_back(nfa.states[b].left); }
          else branches.push_back(b);
        }
        if (branches.size() > 128) { ok = false; return; }
        std::vector<uint8_t> role(branches.size(), 0);               // 1: upper-case half of a pair (walked with its lower-case half), 2: the lower-case half
        for (size_t i = 0; i < branches.size(); i++)
          for (size_t k = 0; k < branches.size() && !role[i]; k++) {
            const cxg_nfa_state& a = nfa.states[branches[i]];
            const cxg_nfa_state& b = nfa.states[branches[k]];
            if (i != k && !role[k] && a.kind == CXG_NFA_BYTE_RANGE && b.kind == CXG_NFA_BYTE_RANGE && a.lo == a.hi && b.lo == b.hi && a.lo >= 'A' && a.lo <= 'Z' &&
                b.lo == (a.lo | 0x20u) && skip(a.next) == skip(b.next)) { role[i] = 1; role[k] = 2; }
          }
        for (size_t i = 0; i < branches.size(); i++) {
          if (role[i] == 1) continue;
          if (role[i] == 2) { nFolded++; path.push_back(nfa.states[branches[i]].lo); dfs(nfa.states[branches[i]].next); path.pop_back(); continue; }
          dfs(branches[i]);
        }
        return;
      }
      case CXG_NFA_BYTE_RANGE:
        if (x.lo != x.hi) { ok = false; return; }
        if ((x.lo >= 'a' && x.lo <= 'z') || (x.lo >= 'A' && x.lo <= 'Z')) nPlainLetters++;
        path.push_back(x.lo); dfs(x.next); path.pop_back(); return;
      case CXG_NFA_SPARSE:
        {                                                             // a small set of bytes (`ju[ln]` under (?i): [LNln]): one alternative per byte, case pairs folded
          std::vector<std::pair<uint8_t, uint32_t>> alts;
          for (uint32_t k = 0; k < x.trans_len; k++) {
            const cxg_nfa_trans& t = nfa.trans[x.trans_off + k];
            if (t.hi - t.lo > 3 || alts.size() > 16) { ok = false; return; }
            for (uint32_t b = t.lo; b <= t.hi; b++) alts.emplace_back(static_cast<uint8_t>(b), t.next);
          }
          std::vector<uint8_t> role(alts.size(), 0);
          for (size_t i = 0; i < alts.size(); i++)
            for (size_t k = 0; k < alts.size() && !role[i]; k++)
              if (i != k && !role[k] && alts[i].first >= 'A' && alts[i].first <= 'Z' && alts[k].first == (alts[i].first | 0x20u) && skip(alts[i].second) == skip(alts[k].second)) { role[i] = 1; role[k] = 2; }
          for (size_t i = 0; i < alts.size(); i++) {
            if (role[i] == 1) continue;
            const uint8_t c = alts[i].first;
            if (role[i] == 2) nFolded++;
            else if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z')) nPlainLetters++;
            path.push_back(c); dfs(alts[i].second); path.pop_back();
          }
          return;
        }
      case CXG_NFA_LOOK: case CXG_NFA_MATCH:
        if (terminal == CXG_NFA_INVALID) terminal = q;
        if (terminal != q || path.empty() || lits.size() >= 64) { ok = false; return; }
        lits.push_back(path);
        return;
      default: ok = false; return;
    }
  };
  dfs(cur);
  if (!ok || lits.empty() || terminal == CXG_NFA_INVALID) return false;
  uint32_t t = terminal;
  if (nfa.states[t].kind == CXG_NFA_LOOK) { if (nfa.states[t].lo < 2) return false; post = nfa.states[t].lo + 1u; t = skip(nfa.states[t].next); }
  if (t >= N || nfa.states[t].kind != CXG_NFA_MATCH) return false;
  // case-insensitive literals (`(?i)(error|fail|panic)`: 600 case variants for the reference, its PikeVM — UseNFA): the literal
  // kernels compare letters ignoring bit 5 (walk.hpp kTeddyFold).  All letters folded or none — a mixed set stays where it was.
  const bool fold = nFolded != 0;
  if (fold && nPlainLetters != 0) return false;
  if (pre == 0 && post == 0 && !fold) return false;
  looks = pre | (post << 8) | (fold ? cxgdev::kTeddyFold : 0u);
  return true;
}

// Nullable patterns.  At a search position `at` the leftmost-first match of a pattern that can match the empty string starts
// AT `at` — the empty path is always there — and is the first accepting path in priority order: a path that consumes bytes wins
// only if it ranks ABOVE the empty path, and paths below the empty one never win.  So FindAll (meta/findall.go:216-283: report,
// skip an empty match at the end of the previous non-empty one, advance by one byte behind an empty match) is
//   * the leftmost-first FindAll of the NON-EMPTY VARIANT — the pattern restricted to the consuming threads that precede Match
//     in the closure of its start (order of dfa/lazy/builder.go:245-293 = order of pikevm.go's thread list) —, and
//   * an empty match at every position 0..len that no non-empty match [s, e) covers with its closed interval [s, e].
// The variant is the same NFA entered through a chain of splits over those threads (their continuations are untouched); the
// device serves it like any UseNFA program, capi_nullable.hip scanNullable adds the empty matches.  Returns false when the start's
// closure holds no Match (not nullable); sawLook: an assertion stands in front of it (refused: nullable at SOME positions only).
bool nonEmptyVariant(const cxg_nfa& nfa, HostNfa& out, bool& sawLook, size_t& nthreads) {
  sawLook = false;
  nthreads = 0;
  std::vector<uint8_t> seen(nfa.n_states, 0);
  std::vector<uint32_t> stack{nfa.start_anchored}, threads;
  bool nullable = false;
  while (!stack.empty() && !nullable) {
    const uint32_t cur = stack.back();
    stack.pop_back();
    if (cur == CXG_NFA_INVALID || cur >= nfa.n_states || seen[cur]) continue;
    seen[cur] = 1;
    const cxg_nfa_state& st = nfa.states[cur];
    switch (st.kind) {
      case CXG_NFA_MATCH: nullable = true; break;
      case CXG_NFA_BYTE_RANGE: case CXG_NFA_SPARSE: threads.push_back(cur); break;
      case CXG_NFA_EPSILON: case CXG_NFA_CAPTURE: stack.push_back(st.next); break;
      case CXG_NFA_SPLIT: stack.push_back(st.right); stack.push_back(st.left); break;
      case CXG_NFA_LOOK: sawLook = true; break;                     // (whether it holds depends on the position)
      default: break;
    }
  }
  if (!nullable) return false;
  nthreads = threads.size();
  out = HostNfa();
  out.states.assign(nfa.states, nfa.states + nfa.n_states);
  out.trans.assign(nfa.trans, nfa.trans + nfa.n_trans);
  out.captureCount = nfa.capture_count;
  if (threads.empty()) return true;
  auto blank = [](uint8_t kind) { cxg_nfa_state x; std::memset(&x, 0, sizeof x); x.kind = kind; x.next = x.left = x.right = CXG_NFA_INVALID; return x; };
  uint32_t entry = threads.back();
  for (size_t i = threads.size() - 1; i-- > 0;) {
    cxg_nfa_state sp = blank(CXG_NFA_SPLIT);
    sp.left = threads[i]; sp.right = entry;
    out.states.push_back(sp);
    entry = static_cast<uint32_t>(out.states.size() - 1);
  }
  cxg_nfa_state any = blank(CXG_NFA_BYTE_RANGE);                   // unanchored prefix: Split(pattern, [00-FF] -> self), compile.go:1633-1650
  any.lo = 0x00; any.hi = 0xFF;
  out.states.push_back(any);
  const uint32_t anyIdx = static_cast<uint32_t>(out.states.size() - 1);
  cxg_nfa_state sp = blank(CXG_NFA_SPLIT);
  sp.left = entry; sp.right = anyIdx;
  out.states.push_back(sp);
  const uint32_t spIdx = static_cast<uint32_t>(out.states.size() - 1);
  out.states[anyIdx].next = spIdx;
  out.startAnchored = entry;
  out.startUnanchored = spIdx;
  // What the new starts do not reach — the pattern's own unanchored prefix above all — must not stay behind: reverseOf() reverses
  // every state it finds, and the old prefix loop, entered backwards from the old start, would keep the reverse DFA alive for ever
  // (a start walk over the whole haystack; found on the device: 300 000-byte haystacks refused with the serial-walk error).
  {
    std::vector<uint8_t> reach(out.states.size(), 0);
    std::vector<uint32_t> st{spIdx};
    while (!st.empty()) {
      const uint32_t q = st.back(); st.pop_back();
      if (q == CXG_NFA_INVALID || q >= out.states.size() || reach[q]) continue;
      reach[q] = 1;
      const cxg_nfa_state& x = out.states[q];
      switch (x.kind) {
        case CXG_NFA_BYTE_RANGE: case CXG_NFA_EPSILON: case CXG_NFA_CAPTURE: case CXG_NFA_LOOK: st.push_back(x.next); break;
        case CXG_NFA_SPLIT: st.push_back(x.left); st.push_back(x.right); break;
        case CXG_NFA_SPARSE: for (uint32_t k = 0; k < x.trans_len; k++) st.push_back(out.trans[x.trans_off + k].next); break;
        default: break;
      }
    }
    for (size_t q = 0; q < out.states.size(); q++) if (!reach[q]) out.states[q] = blank(CXG_NFA_FAIL);
  }
  return true;
}

// Is the language of the anchored break-at-match DFA `C+` for one byte set C — every live state leaves on exactly the bytes of
// C, every state but the start accepts?  (The DFA need not be minimal: `[^,]+` over UTF-8 has a state per pending sequence
// shape, all of them accepting with the same exits, because the reference's automaton of a class that covers everything past
// U+007F also takes any byte >= 0x80 alone, nfa/compile.go:440-590.)  Then leftmost-first FindAll = the maximal runs of C.
bool isClassPlus(const Dfa& d, uint8_t member[256]) {
  if (d.start == 0 || d.start >= d.firstAccept) return false;
  for (int b = 0; b < 256; b++) member[b] = d.table[static_cast<size_t>(d.start) * 256 + b] != 0;
  std::vector<uint8_t> seen(d.nstates, 0);
  std::vector<uint32_t> st{d.start};
  seen[d.start] = 1;
  bool any = false;
  while (!st.empty()) {
    const uint32_t q = st.back(); st.pop_back();
    if (q != d.start && q < d.firstAccept) return false;
    for (int b = 0; b < 256; b++) {
      const uint32_t t = d.table[static_cast<size_t>(q) * 256 + b];
      if ((t != 0) != (member[b] != 0)) return false;
      if (t == d.start) return false;                              // (cannot happen for a break-at-match DFA of a non-nullable pattern)
      if (t != 0) { any = true; if (!seen[t]) { seen[t] = 1; st.push_back(t); } }
    }
  }
  return any;
}

// Is the language of the anchored break-at-match DFA `Q[^Q]*Q` for one byte Q (`"[^"]*"`, `'[^']*'`, `\|[^|]*\|`)?  The start
// leaves on Q alone; every state reached from there without a Q is inside the pair — it is not accepting, stays inside on every
// byte but Q (all 255 of them: the reference's automaton of a class that covers everything past U+007F takes any byte >= 0x80
// alone) — and Q leads to an accepting state nothing leaves (the match ends with the closing Q: break-at-match).
bool isQuotePairs(const Dfa& d, int& quote) {
  if (d.start == 0 || d.start >= d.firstAccept) return false;
  quote = -1;
  for (int b = 0; b < 256; b++)
    if (d.table[static_cast<size_t>(d.start) * 256 + b] != 0) { if (quote >= 0) return false; quote = b; }
  if (quote < 0) return false;
  const uint32_t first = d.table[static_cast<size_t>(d.start) * 256 + quote];
  if (first >= d.firstAccept || first == d.start) return false;
  std::vector<uint8_t> seen(d.nstates, 0);
  std::vector<uint32_t> st{first};
  seen[first] = 1;
  while (!st.empty()) {
    const uint32_t q = st.back(); st.pop_back();
    if (q >= d.firstAccept || q == d.start) return false;
    for (int b = 0; b < 256; b++) {
      const uint32_t t = d.table[static_cast<size_t>(q) * 256 + b];
      if (t == 0) return false;
      if (b == quote) {
        if (t < d.firstAccept) return false;
        for (int c = 0; c < 256; c++) if (d.table[static_cast<size_t>(t) * 256 + c] != 0) return false;
      } else if (!seen[t]) { seen[t] = 1; st.push_back(t); }
    }
  }
  return true;
}

// Is the language of the anchored break-at-match DFA `O [^E]+ E` or `O [^E]* E` for two different bytes O, E?  The start leaves on
// O alone; behind it every state stays inside on every byte but E (all 255: see isQuotePairs) and is not accepting; E leads to
// an accepting state nothing leaves — from every inside state (`*`), or from every inside state but the one behind O, where E
// is dead (`+`).
bool isDelimited(const Dfa& d, int& open, int& close, bool& plus) {
  if (d.start == 0 || d.start >= d.firstAccept) return false;
  open = close = -1;
  for (int b = 0; b < 256; b++)
    if (d.table[static_cast<size_t>(d.start) * 256 + b] != 0) { if (open >= 0) return false; open = b; }
  if (open < 0) return false;
  const uint32_t first = d.table[static_cast<size_t>(d.start) * 256 + open];
  if (first == 0 || first >= d.firstAccept || first == d.start) return false;
  // E: the one byte that does not keep an inside state inside — found on the state behind the first non-E byte
  auto terminal = [&](uint32_t t) { if (t < d.firstAccept) return false; for (int c = 0; c < 256; c++) if (d.table[static_cast<size_t>(t) * 256 + c] != 0) return false; return true; };
  int dead = -1, term = -1;
  for (int b = 0; b < 256; b++) {
    const uint32_t t = d.table[static_cast<size_t>(first) * 256 + b];
    if (t == 0) { if (dead >= 0) return false; dead = b; }
    else if (t >= d.firstAccept) { if (term >= 0 || !terminal(t)) return false; term = b; }
  }
  if ((dead >= 0) == (term >= 0)) return false;
  plus = dead >= 0;
  close = plus ? dead : term;
  if (close == open) return false;
  std::vector<uint8_t> seen(d.nstates, 0);
  std::vector<uint32_t> st;
  seen[first] = 1;
  for (int b = 0; b < 256; b++) {
    if (b == close) continue;
    const uint32_t t = d.table[static_cast<size_t>(first) * 256 + b];
    if (t == 0 || t >= d.firstAccept) return false;
    if (!seen[t]) { seen[t] = 1; st.push_back(t); }
  }
  if (!plus) st.push_back(first);                                  // `*`: the state behind O is an inside state like the others
  bool firstChecked = plus;
  while (!st.empty()) {
    const uint32_t q = st.back(); st.pop_back();
    if (q == first && firstChecked) return false;                  // `+`: nothing leads back behind O
    if (q == first) firstChecked = true;
    if (q >= d.firstAccept || q == d.start) return false;
    for (int b = 0; b < 256; b++) {
      const uint32_t t = d.table[static_cast<size_t>(q) * 256 + b];
      if (b == close) { if (!terminal(t)) return false; continue; }
      if (t == 0 || t >= d.firstAccept) return false;
      if (!seen[t]) { seen[t] = 1; st.push_back(t); }
    }
  }
  return true;
}

void buildProgramFromNfa(cxg_program* p, const cxg_nfa& nfa, int strategy, uint32_t flags) {
  p->strategy = strategy;
  p->flags = flags;
  p->ngroups = static_cast<int>(nfa.capture_count);
  p->nfaStates = static_cast<int>(nfa.n_states);
  p->supported = false;
  try {
    std::string why;
    if (!validateNfa(nfa, why)) throw BuildError{CXG_E_INVALID, why};
    cxgdev::BlobHeader h;
    std::memset(&h, 0, sizeof h);
    h.magic = cxgdev::kBlobMagic;
    h.ngroups = nfa.capture_count;
    bool inAlpha[256];
    alphabetOf(nfa, inAlpha);
    uint8_t info[256];
    for (int b = 0; b < 256; b++) info[b] = inAlpha[b] ? 0 : cxgdev::kInfoSync;
    std::vector<uint8_t> blob(sizeof h, 0);
    std::vector<uint8_t> sflags;
    std::vector<uint8_t> pureLiteral;              // UseDFA program that is one plain literal the chain kernel does not take
    std::vector<std::vector<uint8_t>> prefixLiterals;   // ... or that begins with one of a few required literals of >= 3 bytes:
    Dfa prefixDfa;                                 //     the anchored DFA that extends an occurrence to the match end
    cxgdev::ChainAux chain;
    std::memset(&chain, 0, sizeof chain);
    bool hasLook = false;
    for (uint32_t i = 0; i < nfa.n_states; i++) hasLook = hasLook || nfa.states[i].kind == CXG_NFA_LOOK;
    if (strategy == CXG_USE_BOUNDED_BACKTRACKER) {
      // Unanchored programs of this strategy are concatenations / repetitions of character classes (isSimpleCharClass,
      // strategy.go:1095-1128; anchored ones never get here).  The reference answers them with its bounded backtracker —
      // priority-ordered depth-first search, first match wins: leftmost-first (nfa/backtrack.go:264-300, Longest is off) — while
      // states x (remaining + 1) fits 32 M visited entries, and with the forward + reverse lazy DFA before that, i.e. for all
      // but the last few hundred KiB of a large haystack (find_indices.go:1279-1283 -> :711-732; both DFAs are built for this
      // strategy whatever the quantifiers, compile.go:207-217).  Two leftmost-first engines: the program is the DFA pair's,
      // under the same proof that the lazy DFA's answer does not depend on its cache history.
      if (hasLook) throw BuildError{CXG_E_UNSUPPORTED, "UseBoundedBacktracker program with assertions (the backtracker searches a slice of the haystack: no context in front of it)"};
      strategy = CXG_USE_DFA;
      flags |= CXG_FLAG_HAS_REVERSE_DFA;
      {  // `\d*`, `[a-z]*`: nullable, and still this strategy (isSimpleCharClass is asked before canMatchEmpty, strategy.go:1095-1128).
         // Both of its engines are leftmost-first, so FindAll is the loop of meta/findall.go:216-283 over plain leftmost-first
         // matches: the UseNFA branch's non-empty variant + scanNullable.
        HostNfa probe;
        bool sawLook = false;
        size_t nthreads = 0;
        if (nonEmptyVariant(nfa, probe, sawLook, nthreads)) strategy = CXG_USE_NFA;
      }
    }
    // A look-around program that passed its proof (lookdfa.cc) is the pattern's transducer and nothing else: no table-walking image.
    auto finishFsmOnly = [&](uint32_t maxLen) {
      HostNfa rn = reverseOf(nfa);
      cxg_nfa rvw = rn.view();
      Dfa none;
      if (!buildFsmImage(nfa, none, maxLen, p->fsmBlob, p->fsmWhyNot, &rvw)) { p->fsmBlob.clear(); throw BuildError{CXG_E_UNSUPPORTED, p->fsmWhyNot}; }
      h.kind = cxgdev::kKindFsmOnly;
      h.info_off = static_cast<uint32_t>(blob.size());
      blob.insert(blob.end(), info, info + 256);
      h.total_bytes = static_cast<uint32_t>(blob.size());
      std::memcpy(blob.data(), &h, sizeof h);
      p->blob.swap(blob);
      p->supported = true;
    };
    if (strategy == CXG_USE_DIGIT_PREFILTER && hasLook) {
      // Assertions behind the leading digits (`\d+\.\d+\.\d+\.\d+\b`): the reference runs SearchAtAnchored of its look-aware lazy
      // DFA at every digit (find_indices.go:1050-1088).  Served when that is provably the leftmost-first anchored search and the
      // digit-run skip is sound (lookdfa.cc refuseLookDigitQuirks); then FindAll is plain leftmost-first: the transducer.
      if (nfa.start_unanchored == nfa.start_anchored) throw BuildError{CXG_E_UNSUPPORTED, "start-anchored pattern"};
      refuseLookDigitQuirks(nfa, (flags & CXG_FLAG_DIGIT_RUN_SKIP_SAFE) != 0);
      finishFsmOnly(0u);
      return;
    }
    if (strategy == CXG_USE_DIGIT_PREFILTER) {
      // findIndicesDigitPrefilterAtWithState: anchored DFA at each digit candidate
      p->fwd = determinize(nfa, nfa.start_anchored, true, kMaxDfaStates);
      if (p->fwd.start >= p->fwd.firstAccept) throw BuildError{CXG_E_UNSUPPORTED, "nullable pattern (empty matches)"};
      refuseOrderConflict(nfa, {nfa.start_anchored});
      h.kind = cxgdev::kKindDigit;
      if (flags & CXG_FLAG_DIGIT_RUN_SKIP_SAFE) h.flags |= cxgdev::kFlagRunSkip;
      for (int b = '0'; b <= '9'; b++) info[b] &= ~cxgdev::kInfoSync;  // digits are candidates, never sync
      // per-state flags + the "tail closed" property for the candidate-list kernel (walk.hpp lane_select)
      sflags.assign(256, 0);
      bool tailClosed = true;
      for (uint32_t q = 1; q < p->fwd.nstates; q++) {
        bool loop = true;
        for (int b = '0'; b <= '9'; b++) {
          const uint8_t t = p->fwd.table[static_cast<size_t>(q) * 256 + b];
          if (t != q) loop = false;
          if (q >= p->fwd.firstAccept && t < p->fwd.firstAccept) tailClosed = false;
        }
        if (loop) sflags[q] |= cxgdev::kStateDigitLoop;
      }
      if ((flags & CXG_FLAG_DIGIT_RUN_SKIP_SAFE) && tailClosed) h.flags |= cxgdev::kFlagFastDigit;
      // chain prefilter (walk.hpp "chain prefilter"); the candidates are digit-run starts, so the chain is
      // only usable if it begins with run(digit)
      if (h.flags & cxgdev::kFlagFastDigit) {
        bool complete = false, ordered = false;
        extractChain(p->fwd, info, chain, complete, ordered);
        if (chain.nops >= 2 && chain.op_kind[0] == cxgdev::kChainRun && chain.cls_kind[chain.op_cls[0]] == cxgdev::kClsDigit && chain.op_cls[0] == 0) {
          bool sets = false;
          for (uint32_t k = 0; k < chain.ncls; k++) sets = sets || chain.cls_kind[k] == cxgdev::kClsSet;
          if (!sets || (complete && ordered)) {
            h.flags |= cxgdev::kFlagChain;
            if (complete) h.flags |= cxgdev::kFlagChainComplete;
            if (complete && ordered) h.flags |= cxgdev::kFlagChainOrdered;
            if (sets) h.flags |= cxgdev::kFlagChainSets;
          } else {
            std::memset(&chain, 0, sizeof chain);
          }
        } else {
          std::memset(&chain, 0, sizeof chain);
        }
      } else if (!(flags & CXG_FLAG_DIGIT_RUN_SKIP_SAFE)) {
        // No run skip (the pattern does not start with an unbounded digit run, meta/strategy.go:530-560): the
        // reference tries EVERY digit position in order (find_indices.go:1059-1088), i.e. plain leftmost-first over
        // matches that start with a digit.  A complete ordered chain (`\d{4}-\d{2}-\d{2}`) can then run on the
        // bit-parallel kernel; kFlagChain stays clear, generations 4/5 assume digit-run-start candidates.
        bool complete = false, ordered = false;
        extractChain(p->fwd, info, chain, complete, ordered);
        if (chain.nops >= 1 && complete && ordered) {
          h.flags |= cxgdev::kFlagChainComplete | cxgdev::kFlagChainOrdered;
          for (uint32_t k = 0; k < chain.ncls; k++) if (chain.cls_kind[k] == cxgdev::kClsSet) h.flags |= cxgdev::kFlagChainSets;
        } else {
          std::memset(&chain, 0, sizeof chain);
        }
      }
    } else if (strategy == CXG_USE_DFA || strategy == CXG_USE_BOTH) {
      // (UseDFA without the reference's reverse DFA — non-greedy quantifiers, meta/compile.go:184-205 — answers through its
      // PikeVM, find_indices.go:647: plain leftmost-first, which this forward + reverse pair computes as well: the end of
      // the leftmost-first match, then the smallest start that reaches it — no match starts earlier, or it would be the
      // leftmost one.  No lazy DFA is involved there, so there is no cache history to depend on.)
      // useDFADirect (meta/findall.go:216-239): unanchored forward DFA + anchored reverse DFA.
      // UseBoth (findIndicesAdaptiveAtWithState, find_indices.go:408-441): the DFA's match end only picks where the
      // PikeVM starts — `at`, or end-100 when end > at+100.  The PikeVM is leftmost-first, and nothing starts between
      // `at` and the leftmost match, so the answer is the plain leftmost-first match unless that match is longer than
      // 100 bytes (then the PikeVM starts inside it).  The kernels compute leftmost-first and raise error bit 64 on a
      // longer match (kFlagBothRestart): CXG_E_INPUT, the caller keeps its CPU loop for that haystack.
      // (With a prefilter — CXG_FLAG_HAS_PREFILTER — the reference's PikeVM starts at the prefilter's position, in front of the
      // leftmost match: plain leftmost-first whatever the length, find_indices.go:411-429.  Programs with assertions keep the flag:
      // adjustForAnchors may have dropped the prefilter, compile.go:660-680.)
      bool lookAny = false;
      for (uint32_t i = 0; i < nfa.n_states; i++) lookAny = lookAny || nfa.states[i].kind == CXG_NFA_LOOK;
      if (strategy == CXG_USE_BOTH && (lookAny || !(flags & CXG_FLAG_HAS_PREFILTER))) h.flags |= cxgdev::kFlagBothRestart;
      if (nfa.start_unanchored == nfa.start_anchored) throw BuildError{CXG_E_UNSUPPORTED, "start-anchored pattern"};
      bool look = false;
      for (uint32_t i = 0; i < nfa.n_states; i++) look = look || nfa.states[i].kind == CXG_NFA_LOOK;
      if (look) {
        // Assertions inside a lazy-DFA strategy: the reference answers with its look-aware lazy DFA, which is leftmost-first
        // only for some programs and history-free only when its byte classes are pure in word-ness / newline-ness
        // (lookdfa.cc).  Those that pass the proof are the pattern's transducer, like a UseNFA program; UseBoth keeps its
        // 100-byte restart span.  UseDFA without the reverse DFA (non-greedy quantifiers) first asks DFA.IsMatchAt
        // (find_indices.go:396-403) and then runs its PikeVM: served when that question provably never gets a wrong "no".
        // (With a prefilter the reference skips the question, :381-393; the ABI does not say, so the proof is asked for anyway.)
        if (strategy == CXG_USE_DFA && !(flags & CXG_FLAG_HAS_REVERSE_DFA)) {
          refuseLookDfaQuirks(nfa, nullptr, true);
          finishFsmOnly(0u);
          return;
        }
        HostNfa rn = reverseOf(nfa);
        cxg_nfa rvw = rn.view();
        refuseLookDfaQuirks(nfa, strategy == CXG_USE_DFA ? &rvw : nullptr);
        finishFsmOnly(strategy == CXG_USE_BOTH ? cxgdev::kBothRestartSpan : 0u);
        return;
      }
      if (strategy == CXG_USE_DFA) {
        // `\S+`, `[^,]+`, `[^"]+`, `[a-z0-9_.-]+` ...: one class, one or more times.  The DFA pair's leftmost-first FindAll (forward
        // DFA to the end of the run, reverse DFA back to its start, dfa/lazy/lazy.go:1102-1315, 1769-1920) is the list of the
        // maximal runs of the class — the char-class kernels' job (scan_charclass_wave.hip: runs as long as the haystack,
        // rows straight from the bitmaps), not the transducer's: 7.9 ms -> per GiB for `\S+`, 1.4 s for `[^,]+` on the table kernel.
        static const bool noClassRuns = getenv("CXG_NO_CLASS_RUNS") != nullptr;
        uint8_t member[256];
        bool plus = false;
        int quote = -1;
        if (!noClassRuns) {
          try {
            const Dfa anch = determinize(nfa, nfa.start_anchored, true, kMaxDfaStates);
            plus = isClassPlus(anch, member);
            // `"[^"]*"`: the occurrences of the quote two at a time (the same kernel: starts and ends are owned separately there
            // already, here the parity of the occurrences in front says which is which).  4.1 ms per GiB on the transducer — no
            // byte synchronises it, every tile went through the maps.
            if (!plus && isQuotePairs(anch, quote) && quote < 128) {
              std::memset(member, 0, sizeof member);
              member[quote] = 1;
              plus = true;
            } else quote = -1;
            // `\[[^\]]+\]`, `<[^>]+>`: the delimiter kernel in front of the images built below (which stay: its fallback)
            int dOpen = -1, dClose = -1;
            bool dPlus = false;
            if (!plus && isDelimited(anch, dOpen, dClose, dPlus)) { p->delim[0] = static_cast<uint32_t>(dOpen); p->delim[1] = static_cast<uint32_t>(dClose); p->delim[2] = dPlus ? 1u : 0u; p->delim[3] = 1u; }
          } catch (const BuildError&) { plus = false; }
        }
        if (plus) {
          const int keepStrategy = p->strategy;
          const int keepGroups = p->ngroups;
          buildProgramFromCharClass(p, member, 1, quote >= 0);
          p->strategy = keepStrategy; p->ngroups = keepGroups;
          if (p->supported) return;
        }
      }
      p->fwd = determinize(nfa, nfa.start_unanchored, true, kMaxDfaStates);
      if (p->fwd.start >= p->fwd.firstAccept) throw BuildError{CXG_E_UNSUPPORTED, "nullable pattern (empty matches)"};
      if (strategy == CXG_USE_BOTH || (flags & CXG_FLAG_HAS_REVERSE_DFA)) refuseOrderConflict(nfa, {nfa.start_unanchored});
      HostNfa rn = reverseOf(nfa);
      cxg_nfa rv = rn.view();
      p->rev = determinize(rv, rv.start_anchored, false, kMaxDfaStates);
      if (p->fwd.nstates + p->rev.nstates > kMaxDfaStates) throw BuildError{CXG_E_UNSUPPORTED, "DFA pair exceeds the LDS state budget"};
      h.kind = cxgdev::kKindBidir;
      for (int b = 0; b < 256; b++)
        if (p->fwd.table[static_cast<size_t>(p->fwd.start) * 256 + b] == p->fwd.start) info[b] |= cxgdev::kInfoStartIdle;
      // When the anchored DFA is one complete, ordered chain (a literal, `[a-z]+=\d+`, ...) the leftmost-first
      // FindAll result is a pure bit-parallel function of the class bitmaps: scan_chain_wave.hip.
      try {
        const Dfa anch = determinize(nfa, nfa.start_anchored, true, kMaxDfaStates);
        bool complete = false, ordered = false;
        extractChain(anch, info, chain, complete, ordered);
        if (chain.nops >= 1 && complete && ordered) {
          h.flags |= cxgdev::kFlagChain | cxgdev::kFlagChainComplete | cxgdev::kFlagChainOrdered;
          for (uint32_t k = 0; k < chain.ncls; k++) if (chain.cls_kind[k] == cxgdev::kClsSet) h.flags |= cxgdev::kFlagChainSets;
          sflags.assign(256, 0);                                    // keeps the aux layout of the digit image
        } else {
          std::memset(&chain, 0, sizeof chain);
          // Not a chain the bit-parallel kernel takes (more than four distinct bytes, typically): a plain literal —
          // every state of the anchored DFA leaves on exactly one byte — is searched by the literal kernels instead
          // (three-byte fingerprint + exact compare; leftmost, non-overlapping: the same answer as the DFA pair).
          std::vector<uint8_t> lit;
          uint32_t q = anch.start;
          bool plain = true;
          while (plain && q < anch.firstAccept && lit.size() < 256) {
            int only = -1;
            for (int b = 0; b < 256; b++)
              if (anch.table[static_cast<size_t>(q) * 256 + b] != 0) { if (only >= 0) plain = false; only = b; }
            if (only < 0) plain = false;
            if (plain) { lit.push_back(static_cast<uint8_t>(only)); q = anch.table[static_cast<size_t>(q) * 256 + only]; }
          }
          if (plain && q >= anch.firstAccept && anch.firstAccept == anch.nstates - 1) {
            for (int b = 0; b < 256; b++) if (anch.table[static_cast<size_t>(q) * 256 + b] != 0) plain = false;   // nothing follows the literal
            // (a UseBoth literal longer than the 100-byte restart span keeps the DFA-pair image, whose kernels check the span)
            if (plain && lit.size() >= 3 && lit.size() <= 255 && (strategy != CXG_USE_BOTH || lit.size() <= cxgdev::kBothRestartSpan)) pureLiteral = lit;
          } else if (requiredPrefixes(anch, prefixLiterals)) {
            prefixDfa = anch;
          }
        }
      } catch (const BuildError&) { std::memset(&chain, 0, sizeof chain); }
      if (!prefixLiterals.empty()) appendPrefixAux(blob, h, prefixLiterals, prefixDfa);
    } else if (strategy == CXG_USE_NFA || strategy == CXG_USE_TEDDY) {
      // UseNFA: the reference answers through its PikeVM (find_indices.go:520-560 -> nfa/pikevm.go SearchAt): plain
      // leftmost-first with the assertions of nfa.Look checked at each position (pikevm.go:1646-1674).  Reached by small
      // patterns with word boundaries or multi-line anchors (meta/strategy.go:1377-1546: `\berror\b`, `(?m)^line`), which no
      // DFA strategy takes.  Served by the transducer kernel alone (fsm.hpp "Look-around"): no table-walking image exists.
      // UseTeddy reaches here with the NFA of a literal alternation behind (?m)^ (`(?m)^(GET|POST|PUT)`): the reference
      // filters Teddy's candidates by a line-start check (prefilter.WrapLineAnchor, prefilter/wrap.go:45-66) and — literals
      // of different lengths — runs the PikeVM from the first candidate that passes (find_indices.go:941-950): the
      // leftmost-first match of the whole pattern, which is what the transducer computes.
      if (nfa.start_unanchored == nfa.start_anchored) throw BuildError{CXG_E_UNSUPPORTED, "start-anchored pattern"};
      if (strategy == CXG_USE_TEDDY) {
        // The reference's line-start check sits in front of EVERY literal candidate as soon as the pattern holds a (?m)^
        // anywhere (compile.go:670-677): that equals the pattern's meaning only when no byte can be consumed before a
        // StartLine assertion has been crossed.  (cxg_compile checks the same on the AST; a caller's NFA is checked here.)
        bool anyLine = false;
        for (uint32_t i = 0; i < nfa.n_states; i++) anyLine = anyLine || (nfa.states[i].kind == CXG_NFA_LOOK && nfa.states[i].lo == 2);
        if (!anyLine) throw BuildError{CXG_E_UNSUPPORTED, "UseTeddy program without (?m)^ handed over as an NFA: use cxg_program_from_literals"};
        std::vector<uint8_t> seen(nfa.n_states, 0);
        std::vector<uint32_t> st{nfa.start_anchored};
        bool sawLine = false;
        while (!st.empty()) {
          const uint32_t q = st.back(); st.pop_back();
          if (q == CXG_NFA_INVALID || q >= nfa.n_states || seen[q]) continue;
          seen[q] = 1;
          const cxg_nfa_state& x = nfa.states[q];
          switch (x.kind) {
            case CXG_NFA_EPSILON: case CXG_NFA_CAPTURE: st.push_back(x.next); break;
            case CXG_NFA_SPLIT: st.push_back(x.left); st.push_back(x.right); break;
            case CXG_NFA_LOOK: if (x.lo == 2 /* StartLine */) sawLine = true; else st.push_back(x.next); break;
            case CXG_NFA_BYTE_RANGE: case CXG_NFA_SPARSE: case CXG_NFA_MATCH:
              throw BuildError{CXG_E_UNSUPPORTED, "(?m)^ on some alternatives only: the reference applies its line-start check to every literal candidate (prefilter.WrapLineAnchor)"};
            default: break;
          }
        }
        (void)sawLine;
      }
      bool look = false;
      for (uint32_t i = 0; i < nfa.n_states; i++) look = look || nfa.states[i].kind == CXG_NFA_LOOK;
      // Nullable pattern (the reference sends every one of them here: canMatchEmpty, meta/strategy.go:1503): the program is the
      // non-empty variant's transducer, the empty matches are added behind the scan (capi_nullable.hip scanNullable).
      HostNfa variant;
      const cxg_nfa* prog = &nfa;
      cxg_nfa variantView;
      if (strategy == CXG_USE_NFA) {
        bool sawLook = false;
        size_t nthreads = 0;
        if (nonEmptyVariant(nfa, variant, sawLook, nthreads)) {
          if (look) throw BuildError{CXG_E_UNSUPPORTED, "nullable pattern with assertions (empty matches at some positions only)"};
          p->nullable = true;
          if (nthreads == 0) {                                      // `a*?`, `(|a)`: only empty matches
            p->nullableOnlyEmpty = true;
            h.kind = cxgdev::kKindFsmOnly;
            h.info_off = static_cast<uint32_t>(blob.size());
            blob.insert(blob.end(), info, info + 256);
            h.total_bytes = static_cast<uint32_t>(blob.size());
            std::memcpy(blob.data(), &h, sizeof h);
            p->blob.swap(blob);
            p->supported = true;
            return;
          }
          variantView = variant.view();
          prog = &variantView;
        } else if (sawLook && !look) {
          throw BuildError{CXG_E_INTERNAL, "assertion met in a program without LOOK states"};
        }
      }
      HostNfa rn = reverseOf(*prog);
      cxg_nfa rvw = rn.view();
      Dfa rv;
      if (!look) rv = determinize(rvw, rvw.start_anchored, false, kMaxDfaStates);
      // (the determinisation above throws for the few programs past its state budget — with them the literal shape below is not tried)
      const bool fsmBuilt = buildFsmImage(*prog, rv, 0u, p->fsmBlob, p->fsmWhyNot, look ? &rvw : nullptr);
      if (!fsmBuilt) p->fsmBlob.clear();
      // literals between assertions / case-insensitive literals: the literal kernel in front of the transducer (which stays: its
      // fallback — where the pattern has one: `(?i)\b(error|fail|exception|panic|fatal)\b` has more than 64 symbols x kinds and
      // runs on the literal kernel alone, CXG_E_INPUT for a haystack that kernel gives up on)
      const bool noWrapped = getenv("CXG_NO_WRAPPED_LITERALS") != nullptr;   // (read per build: tests/test_gpu_fsm.py keeps these programs on the transducer)
      std::vector<std::vector<uint8_t>> wl;
      uint32_t wlooks = 0;
      if (!p->nullable && !noWrapped && wrappedLiterals(nfa, wl, wlooks)) {
        cxg_program tmp;
        buildLiteralImage(&tmp, wl, 1, (wlooks & cxgdev::kTeddyFold) != 0u);
        if (tmp.supported) {
          const cxgdev::BlobHeader* th = reinterpret_cast<const cxgdev::BlobHeader*>(tmp.blob.data());
          reinterpret_cast<cxgdev::TeddyAux*>(tmp.blob.data() + th->aux_off)->looks = wlooks;
          p->blob.swap(tmp.blob);
          p->supported = true;
          return;
        }
      }
      if (!fsmBuilt) { p->nullable = false; throw BuildError{CXG_E_UNSUPPORTED, p->fsmWhyNot}; }
      h.kind = cxgdev::kKindFsmOnly;
      h.info_off = static_cast<uint32_t>(blob.size());
      blob.insert(blob.end(), info, info + 256);
      h.total_bytes = static_cast<uint32_t>(blob.size());
      std::memcpy(blob.data(), &h, sizeof h);
      p->blob.swap(blob);
      p->supported = true;
      return;
    } else {
      throw BuildError{CXG_E_UNSUPPORTED, std::string("strategy ") + cxg_strategy_name(strategy) + " has no device kernel"};
    }
    // General-DFA kernel (scan_fsm.hip): the FindAll transducer of the pattern.  For UseDFA / UseBoth it IS the
    // reference's loop (forward DFA with break-at-match, restart at the match end, reverse DFA for the start).  A
    // UseDigitPrefilter program qualifies when its digit-scan order equals plain leftmost-first: run skip safe and tail
    // closed (kFlagFastDigit) or no run skip at all (see above); the others keep the reference's quirk on the
    // table-walking kernel.  Programs that already are complete ordered chains / plain literals get the image too: it is
    // their fallback for match-dense or synchronisation-free input.
    {
      // Run skip (find_indices.go:1079-1084) leaves plain leftmost-first order untouched exactly when it is SOUND: the
      // anchored DFA goes from its start to one state on every digit and stays there on every digit, so a candidate
      // anywhere inside a digit run fails or succeeds with the run's first position.  The reference also sets the flag
      // for leads like `[0-5]+` (a quirk that loses matches: "61x" for `[0-5]+x`); those keep the table-walking kernel.
      bool skipSound = false;
      if (h.kind == cxgdev::kKindDigit && (flags & CXG_FLAG_DIGIT_RUN_SKIP_SAFE)) {
        const uint32_t s1 = p->fwd.table[static_cast<size_t>(p->fwd.start) * 256 + '0'];
        skipSound = s1 != 0;
        for (int b = '0'; b <= '9' && skipSound; b++)
          skipSound = p->fwd.table[static_cast<size_t>(p->fwd.start) * 256 + b] == s1 && p->fwd.table[static_cast<size_t>(s1) * 256 + b] == s1;
      }
      const bool plainOrder = h.kind == cxgdev::kKindBidir || skipSound || !(flags & CXG_FLAG_DIGIT_RUN_SKIP_SAFE);
      if (plainOrder && nfa.start_unanchored != nfa.start_anchored) {
        try {
          Dfa rv = p->rev;
          if (rv.nstates == 0) { HostNfa rn = reverseOf(nfa); cxg_nfa rvw = rn.view(); rv = determinize(rvw, rvw.start_anchored, false, kMaxDfaStates); }
          if (!buildFsmImage(nfa, rv, (h.flags & cxgdev::kFlagBothRestart) ? cxgdev::kBothRestartSpan : 0u, p->fsmBlob, p->fsmWhyNot)) p->fsmBlob.clear();
        } catch (const BuildError& e) { p->fsmBlob.clear(); p->fsmWhyNot = e.msg; }
      } else p->fsmWhyNot = "digit-scan order differs from leftmost-first (run-skip quirk)";
    }
    h.fwd_states = p->fwd.nstates; h.fwd_start = p->fwd.start; h.fwd_first_accept = p->fwd.firstAccept;
    appendTable(blob, p->fwd, h.fwd_off);
    if (h.kind == cxgdev::kKindBidir) {
      h.rev_states = p->rev.nstates; h.rev_start = p->rev.start; h.rev_first_accept = p->rev.firstAccept;
      appendTable(blob, p->rev, h.rev_off);
    }
    h.info_off = static_cast<uint32_t>(blob.size());
    blob.insert(blob.end(), info, info + 256);
    if (!sflags.empty()) {
      h.aux_off = static_cast<uint32_t>(blob.size());
      blob.insert(blob.end(), sflags.begin(), sflags.end());
      const uint8_t* cb = reinterpret_cast<const uint8_t*>(&chain);
      blob.insert(blob.end(), cb, cb + sizeof chain);           // ChainAux follows sflags[256]
      while (blob.size() % 16) blob.push_back(0);
      h.aux_len = static_cast<uint32_t>(blob.size()) - h.aux_off;
    }
    h.total_bytes = static_cast<uint32_t>(blob.size());
    std::memcpy(blob.data(), &h, sizeof h);
    p->blob.swap(blob);
    p->supported = true;
    if (!pureLiteral.empty()) {                   // replace the DFA-pair image by the literal image (same answers, ~10x the speed)
      std::vector<uint8_t> keep;
      keep.swap(p->blob);
      buildLiteralImage(p, {pureLiteral}, 1);
      if (!p->supported) { p->blob.swap(keep); p->supported = true; p->whyNot.clear(); }
      p->ngroups = static_cast<int>(nfa.capture_count);
    }
    // `(?i)(select|insert|update|delete)`: a UseDFA / UseBoth program that is a case-insensitive alternation of literals — the folded
    // literal set (wrappedLiterals, walk.hpp kTeddyFold) instead of literal prefixes + anchored DFA walks; the transducer stays its
    // fallback.  (UseBoth: no literal is longer than the restart span, so the long-match condition cannot arise.)
    if ((strategy == CXG_USE_DFA || strategy == CXG_USE_BOTH) && !hasLook && !p->nullable && getenv("CXG_NO_WRAPPED_LITERALS") == nullptr &&
        reinterpret_cast<const cxgdev::BlobHeader*>(p->blob.data())->kind == cxgdev::kKindBidir) {
      std::vector<std::vector<uint8_t>> wl;
      uint32_t wlooks = 0;
      if (wrappedLiterals(nfa, wl, wlooks) && wlooks == cxgdev::kTeddyFold) {
        size_t longest = 0;
        for (auto& l : wl) longest = std::max(longest, l.size());
        cxg_program tmp;
        if (longest <= cxgdev::kBothRestartSpan) buildLiteralImage(&tmp, wl, 1, true);
        if (tmp.supported) {
          const cxgdev::BlobHeader* th = reinterpret_cast<const cxgdev::BlobHeader*>(tmp.blob.data());
          reinterpret_cast<cxgdev::TeddyAux*>(tmp.blob.data() + th->aux_off)->looks = wlooks;
          p->blob.swap(tmp.blob);
        }
      }
    }
  } catch (const BuildError& e) {
    p->whyNot = e.msg;
  }
}

namespace {

// Ordered epsilon closure with the capture slots met on the way (DFS, left first; first visit wins —
// the PikeVM's Visited set, nfa/pikevm.go:1925-1928).
struct CapClosure {
  const cxg_nfa& n;
  struct Item { uint32_t state; uint32_t mask; };
  explicit CapClosure(const cxg_nfa& nfa) : n(nfa) {}
  void run(uint32_t seed, std::vector<Item>& out) const {
    std::vector<uint8_t> seen(n.n_states, 0);
    struct Fr { uint32_t s, m; };
    std::vector<Fr> st{{seed, 0}};
    while (!st.empty()) {
      Fr f = st.back(); st.pop_back();
      if (f.s == CXG_NFA_INVALID || f.s >= n.n_states || seen[f.s]) continue;
      seen[f.s] = 1;
      const cxg_nfa_state& x = n.states[f.s];
      switch (x.kind) {
        case CXG_NFA_MATCH: case CXG_NFA_BYTE_RANGE: case CXG_NFA_SPARSE: out.push_back({f.s, f.m}); break;
        case CXG_NFA_EPSILON: st.push_back({x.next, f.m}); break;
        case CXG_NFA_SPLIT: st.push_back({x.right, f.m}); st.push_back({x.left, f.m}); break;
        case CXG_NFA_CAPTURE: {
          const uint32_t slot = x.cap_index * 2 + (x.cap_start ? 0u : 1u);
          st.push_back({x.next, slot < 32 ? (f.m | (1u << slot)) : f.m});
          break;
        }
        default: break;
      }
    }
  }
};

}  // namespace

namespace {

// Captures from chain boundaries.  In a complete ordered chain every step boundary of a match is forced, and the
// one-pass table stamps a slot on a fixed kind of transition (first byte of a step, a further byte of a run, the
// end), so a slot is "boundary + constant".  The relation is read off witness matches — every run once with length
// 1, 2 and several distinct longer lengths — walked through the capture table itself (walk.hpp capture_walk); a slot
// that is not the same boundary + constant on all of them, or more than four run ends needed, leaves the program on
// the two-kernel path.
void deriveChainCaps(const cxgdev::ChainAux& chain, const std::vector<uint8_t>& capBlob, uint32_t nslots, uint8_t out[40]) {
  std::memset(out, 0, 40);
  cxgdev::ChainCaps cc;
  std::memset(&cc, 0, sizeof cc);
  static_assert(sizeof(cxgdev::ChainCaps) == 40, "ChainCaps layout");
  if (nslots < 4 || nslots > 16 || chain.nops == 0) return;
  const cxgdev::CapHeader* ch = reinterpret_cast<const cxgdev::CapHeader*>(capBlob.data());
  cxgdev::CapView view{capBlob.data() + ch->next_off, capBlob.data() + ch->maskid_off, capBlob.data() + ch->fin_off,
                       reinterpret_cast<const uint32_t*>(capBlob.data() + ch->masks_off), ch->n_entries, ch->start_entry};
  const uint32_t nops = chain.nops;
  auto repByte = [&](int cls) -> int {
    for (int b = 0; b < 128; b++) if (cxgdev::chain_class_has(chain, cls, static_cast<uint32_t>(b))) return b;
    return -1;
  };
  static const int kLens[6][8] = {{1, 1, 1, 1, 1, 1, 1, 1}, {2, 2, 2, 2, 2, 2, 2, 2}, {2, 3, 4, 5, 6, 7, 8, 9}, {9, 7, 5, 3, 2, 4, 6, 8},
                                  {1, 4, 1, 3, 1, 5, 1, 2}, {3, 1, 5, 1, 2, 1, 4, 1}};
  constexpr int kW = 6;
  std::vector<std::vector<int64_t>> bnd(kW), slots(kW);
  for (int w = 0; w < kW; w++) {
    std::vector<uint8_t> hay;
    bnd[w].push_back(0);
    for (uint32_t k = 0; k < nops; k++) {
      const int b = repByte(chain.op_cls[k]);
      if (b < 0) return;
      const int len = chain.op_kind[k] == cxgdev::kChainRun ? kLens[w][k % 8] : 1;
      hay.insert(hay.end(), static_cast<size_t>(len), static_cast<uint8_t>(b));
      bnd[w].push_back(static_cast<int64_t>(hay.size()));
    }
    std::vector<int64_t> row(nslots, -1);
    row[0] = 0; row[1] = static_cast<int64_t>(hay.size());
    hay.resize(hay.size() + 8, 0);
    if (!cxgdev::capture_walk(view, hay.data(), row.data(), nslots)) return;
    slots[w] = row;
  }
  // sources: boundary 0 (start), boundary nops (end), boundary k+1 of every run step k
  for (int i = 0; i < cxgdev::kCapMaxRuns; i++) cc.run_op[i] = 0xFF;
  auto runSource = [&](uint32_t k) -> int {           // allocate / find the compaction slot of run step k
    for (int i = 0; i < cxgdev::kCapMaxRuns; i++) if (cc.run_op[i] == k) return i;
    for (int i = 0; i < cxgdev::kCapMaxRuns; i++) if (cc.run_op[i] == 0xFF) { cc.run_op[i] = static_cast<uint8_t>(k); cc.nruns = static_cast<uint8_t>(i + 1); return i; }
    return -1;
  };
  for (uint32_t sl = 0; sl < nslots; sl++) {
    bool allUnset = true, anyUnset = false;
    for (int w = 0; w < kW; w++) { if (slots[w][sl] < 0) anyUnset = true; else allUnset = false; }
    if (allUnset) { cc.src[sl] = cxgdev::kCapSrcUnset; cc.off[sl] = 0; continue; }
    if (anyUnset) return;
    bool found = false;
    // candidate boundaries: start, end, then run ends; the consistent one with the smallest constant wins (a literal's
    // witnesses are all the same string, so several boundaries are consistent there)
    std::vector<uint32_t> cand = {0u, nops};
    for (uint32_t k = 0; k + 1 < nops; k++) if (chain.op_kind[k] == cxgdev::kChainRun) cand.push_back(k + 1);
    int64_t bestAbs = 1000; uint32_t bestB = 0; int64_t bestD = 0;
    for (uint32_t bi : cand) {
      const int64_t dlt = slots[0][sl] - bnd[0][bi];
      bool same = dlt >= -8 && dlt <= 8;
      for (int w = 1; w < kW && same; w++) same = (slots[w][sl] - bnd[w][bi]) == dlt;
      if (!same) continue;
      const int64_t ab = dlt < 0 ? -dlt : dlt;
      const int64_t cost = ab + ((bi != 0 && bi != nops) ? 100 : 0);   // a run end costs a compaction in the kernel: last resort
      if (cost < bestAbs) { bestAbs = cost; bestB = bi; bestD = dlt; found = true; }
    }
    if (found) {
      uint8_t src;
      if (bestB == 0) src = cxgdev::kCapSrcStart;
      else if (bestB == nops) src = cxgdev::kCapSrcEnd;
      else {
        const int r = runSource(bestB - 1);
        if (r < 0) return;
        src = static_cast<uint8_t>(cxgdev::kCapSrcRun0 + r);
      }
      cc.src[sl] = src; cc.off[sl] = static_cast<int8_t>(bestD);
    }
    if (!found) return;
  }
  if (cc.src[0] != cxgdev::kCapSrcStart || cc.off[0] != 0 || cc.src[1] != cxgdev::kCapSrcEnd || cc.off[1] != 0) return;
  cc.on = 1;
  cc.nslots = static_cast<uint8_t>(nslots);
  std::memcpy(out, &cc, sizeof cc);
}

}  // namespace

// General capture pass: the NFA itself as a device image, walked depth-first in priority order per match row (device/bt.hpp).
static void buildBtCaptureImage(cxg_program* p, const cxg_nfa& nfa, bool hasLook) {
      if (nfa.n_states > 4096 || nfa.n_trans > (1u << 20)) throw BuildError{CXG_E_UNSUPPORTED, "NFA too large for the backtracking capture image"};
      cxgdev::BtHeader bh;
      std::memset(&bh, 0, sizeof bh);
      bh.magic = cxgdev::kBtMagic; bh.n_states = nfa.n_states; bh.n_trans = nfa.n_trans; bh.start = nfa.start_anchored;
      bh.nslots = nfa.capture_count * 2;
      std::vector<uint8_t> bb(sizeof bh, 0);
      bh.states_off = static_cast<uint32_t>(bb.size());
      for (uint32_t i = 0; i < nfa.n_states; i++) {
        const cxg_nfa_state& x = nfa.states[i];
        cxgdev::BtState t;
        std::memset(&t, 0, sizeof t);
        t.kind = x.kind; t.lo = x.lo; t.hi = x.hi;
        t.next = x.kind == CXG_NFA_SPLIT ? x.left : x.next;
        t.alt = x.kind == CXG_NFA_SPLIT ? x.right : cxgdev::kBtInvalid;
        if (x.kind == CXG_NFA_CAPTURE) { const uint32_t sl = x.cap_index * 2 + (x.cap_start ? 0u : 1u); t.cap_slot = static_cast<uint8_t>(sl < 255 ? sl : 255); }
        if (x.kind == CXG_NFA_SPARSE) {
          if (x.trans_len > 0xFFFu) throw BuildError{CXG_E_UNSUPPORTED, "sparse state with more than 4095 transitions"};
          t.trans_off_len = (x.trans_off << 12) | x.trans_len;
        }
        const uint8_t* q = reinterpret_cast<const uint8_t*>(&t);
        bb.insert(bb.end(), q, q + sizeof t);
      }
      bh.trans_off = static_cast<uint32_t>(bb.size());
      for (uint32_t i = 0; i < nfa.n_trans; i++) {
        cxgdev::BtTrans t{nfa.trans[i].lo, nfa.trans[i].hi, 0, nfa.trans[i].next};
        const uint8_t* q = reinterpret_cast<const uint8_t*>(&t);
        bb.insert(bb.end(), q, q + sizeof t);
      }
      while (bb.size() % 16) bb.push_back(0);
      bh.total_bytes = static_cast<uint32_t>(bb.size());
      std::memcpy(bb.data(), &bh, sizeof bh);
      p->capBlob.swap(bb);
      p->capHasLook = hasLook;
      std::memset(p->chainCaps, 0, sizeof p->chainCaps);
}

void buildSubmatchProgram(cxg_program* p, const cxg_nfa& nfa, int strategy) {
  p->subSupported = false;
  p->subNullable = false;
  try {
    if (nfa.capture_count > 16) throw BuildError{CXG_E_UNSUPPORTED, "more than 15 capture groups"};
    if (nfa.start_unanchored == nfa.start_anchored) throw BuildError{CXG_E_UNSUPPORTED, "start-anchored pattern"};
    cxgdev::ChainAux spanChain;                   // set when the spans come from the chain kernel
    std::memset(&spanChain, 0, sizeof spanChain);
    std::memset(p->chainCaps, 0, sizeof p->chainCaps);
    bool hasLook = false;
    for (uint32_t i = 0; i < nfa.n_states; i++) hasLook = hasLook || nfa.states[i].kind == CXG_NFA_LOOK;
    if (hasLook) {
      // Assertions.  FindAllSubmatch of a UseNFA / UseDFA / UseBoth / UseDigitPrefilter / UseBoundedBacktracker engine is the
      // PikeVM over the whole haystack (meta/findall.go:89-98 -> nfa/pikevm.go:2186-2328): plain leftmost-first with the
      // assertions checked at each position (pikevm.go:1646-1674) — no lazy DFA, so none of lookdfa.cc's conditions apply.
      // UseTeddy behind (?m)^ finds the span first (line-filtered literal candidates, find_indices.go:925-951) and the slots
      // inside it: the same rows when the program itself is served (every alternative behind the assertion).
      // Spans: the look-aware transducer; slots: the backtracking pass, whose LOOK states read the bytes around the position
      // (device/bt.hpp).  No table-walking image, no one-pass table (a table walk cannot test an assertion).
      const bool pike = strategy == CXG_USE_NFA || strategy == CXG_USE_DFA || strategy == CXG_USE_BOTH || strategy == CXG_USE_DIGIT_PREFILTER ||
                        strategy == CXG_USE_BOUNDED_BACKTRACKER;
      if (!pike && !(strategy == CXG_USE_TEDDY && p->supported))
        throw BuildError{CXG_E_UNSUPPORTED, "assertions: FindAllSubmatch of this strategy is not the PikeVM over the whole haystack"};
      HostNfa rn = reverseOf(nfa);
      cxg_nfa rvw = rn.view();
      Dfa none;
      std::string w;
      if (!buildFsmImage(nfa, none, 0u, p->subFsmBlob, w, &rvw)) { p->subFsmBlob.clear(); throw BuildError{CXG_E_UNSUPPORTED, w}; }
      cxgdev::BlobHeader h;
      std::memset(&h, 0, sizeof h);
      h.magic = cxgdev::kBlobMagic; h.kind = cxgdev::kKindFsmOnly; h.ngroups = nfa.capture_count;
      bool inAlpha[256]; alphabetOf(nfa, inAlpha);
      std::vector<uint8_t> blob(sizeof h, 0);
      h.info_off = static_cast<uint32_t>(blob.size());
      for (int b = 0; b < 256; b++) blob.push_back(inAlpha[b] ? 0 : cxgdev::kInfoSync);
      h.total_bytes = static_cast<uint32_t>(blob.size());
      std::memcpy(blob.data(), &h, sizeof h);
      p->subBlob.swap(blob);
    } else {
    // ---- spans: unanchored forward + reverse DFA (the bidirectional image of buildProgramFromNfa)
    Dfa fwd = determinize(nfa, nfa.start_unanchored, true, kMaxDfaStates);
    if (fwd.start >= fwd.firstAccept) {
      // Nullable pattern (round 5; meta/findall.go:390-447, the same empty-match rule as FindAllIndex :247-257): the spans are what the
      // FindAllIndex program of the pattern gives (non-empty variant + merged empty matches, capi_nullable.hip scanNullable); every row — the empty
      // ones too: the top-priority empty path decides which groups take part — gets its slots from the backtracking pass over THIS NFA
      // anchored at the row's start and ending at its end.  A search of a nullable pattern always answers at its own start position, so
      // the reported match is the top-priority path from there, which is what the pass finds first.
      if (!(p->nullable && p->supported)) throw BuildError{CXG_E_UNSUPPORTED, "nullable pattern (empty matches)"};
      buildBtCaptureImage(p, nfa, false);
      p->subBlob.clear(); p->subFsmBlob.clear();
      std::memset(p->chainCaps, 0, sizeof p->chainCaps);
      p->subNullable = true;
      p->subSupported = true;
      return;
    }
    HostNfa rn = reverseOf(nfa);
    cxg_nfa rv = rn.view();
    Dfa rev = determinize(rv, rv.start_anchored, false, kMaxDfaStates);
    if (fwd.nstates + rev.nstates > kMaxDfaStates) throw BuildError{CXG_E_UNSUPPORTED, "DFA pair exceeds the LDS state budget"};
    {
      cxgdev::BlobHeader h;
      std::memset(&h, 0, sizeof h);
      h.magic = cxgdev::kBlobMagic; h.kind = cxgdev::kKindBidir; h.ngroups = nfa.capture_count;
      bool inAlpha[256]; alphabetOf(nfa, inAlpha);
      uint8_t info[256];
      for (int b = 0; b < 256; b++) {
        info[b] = inAlpha[b] ? 0 : cxgdev::kInfoSync;
        if (fwd.table[static_cast<size_t>(fwd.start) * 256 + b] == fwd.start) info[b] |= cxgdev::kInfoStartIdle;
      }
      std::vector<uint8_t> blob(sizeof h, 0);
      h.fwd_states = fwd.nstates; h.fwd_start = fwd.start; h.fwd_first_accept = fwd.firstAccept;
      appendTable(blob, fwd, h.fwd_off);
      h.rev_states = rev.nstates; h.rev_start = rev.start; h.rev_first_accept = rev.firstAccept;
      appendTable(blob, rev, h.rev_off);
      h.info_off = static_cast<uint32_t>(blob.size());
      blob.insert(blob.end(), info, info + 256);
      // spans by the bit-parallel chain kernel when the anchored DFA is one complete, ordered chain
      try {
        const Dfa anch = determinize(nfa, nfa.start_anchored, true, kMaxDfaStates);
        cxgdev::ChainAux chain;
        bool complete = false, ordered = false;
        uint8_t syncInfo[256];
        for (int b = 0; b < 256; b++) syncInfo[b] = info[b] & cxgdev::kInfoSync;
        extractChain(anch, syncInfo, chain, complete, ordered);
        if (chain.nops >= 1 && complete && ordered) {
          h.flags |= cxgdev::kFlagChain | cxgdev::kFlagChainComplete | cxgdev::kFlagChainOrdered;
          for (uint32_t k = 0; k < chain.ncls; k++) if (chain.cls_kind[k] == cxgdev::kClsSet) h.flags |= cxgdev::kFlagChainSets;
          h.aux_off = static_cast<uint32_t>(blob.size());
          blob.insert(blob.end(), 256, 0);
          const uint8_t* cb = reinterpret_cast<const uint8_t*>(&chain);
          blob.insert(blob.end(), cb, cb + sizeof chain);
          while (blob.size() % 16) blob.push_back(0);
          h.aux_len = static_cast<uint32_t>(blob.size()) - h.aux_off;
          spanChain = chain;
        } else {
          std::vector<std::vector<uint8_t>> lits;
          if (requiredPrefixes(anch, lits)) appendPrefixAux(blob, h, lits, anch);
        }
      } catch (const BuildError&) {}
      h.total_bytes = static_cast<uint32_t>(blob.size());
      std::memcpy(blob.data(), &h, sizeof h);
      p->subBlob.swap(blob);
    }
    {  // spans by the general-DFA kernel (plain leftmost-first: what the reference's PikeVM reports as group 0)
      std::string w;
      if (!buildFsmImage(nfa, rev, 0u, p->subFsmBlob, w)) p->subFsmBlob.clear();
    }
    }
    // ---- one-pass capture table; patterns that are not one-pass get the backtracking image instead (device/bt.hpp)
    try {
    if (hasLook) throw BuildError{CXG_E_UNSUPPORTED, "assertions: slots by the backtracking pass"};
    CapClosure cc(nfa);
    std::vector<uint32_t> entryState;             // entry id -> NFA state whose closure is taken
    std::map<uint32_t, uint32_t> entryOf;
    auto entry = [&](uint32_t st) -> uint32_t {
      auto it = entryOf.find(st);
      if (it != entryOf.end()) return it->second;
      if (entryState.size() >= 254) throw BuildError{CXG_E_UNSUPPORTED, "capture table too large"};
      uint32_t id = static_cast<uint32_t>(entryState.size());
      entryOf[st] = id; entryState.push_back(st);
      return id;
    };
    std::vector<uint32_t> masks;                  // distinct slot masks
    auto maskId = [&](uint32_t m) -> uint8_t {
      for (size_t i = 0; i < masks.size(); i++) if (masks[i] == m) return static_cast<uint8_t>(i);
      if (masks.size() >= 254) throw BuildError{CXG_E_UNSUPPORTED, "too many distinct capture actions"};
      masks.push_back(m);
      return static_cast<uint8_t>(masks.size() - 1);
    };
    std::vector<uint8_t> next, mid, fin;
    entry(nfa.start_anchored);
    for (uint32_t e = 0; e < entryState.size(); e++) {
      next.resize((e + 1) * 256, 0xFF); mid.resize((e + 1) * 256, 0); fin.resize(e + 1, 0xFF);
      std::vector<CapClosure::Item> items;
      cc.run(entryState[e], items);
      std::vector<int> owner(256, -1);           // which consuming state claimed the byte
      for (auto& it : items) {
        const cxg_nfa_state& x = nfa.states[it.state];
        if (x.kind == CXG_NFA_MATCH) { if (fin[e] == 0xFF) fin[e] = maskId(it.mask); continue; }
        auto claim = [&](int lo, int hi, uint32_t to) {
          for (int b = lo; b <= hi; b++) {
            if (owner[b] >= 0 && owner[b] != static_cast<int>(it.state))
              throw BuildError{CXG_E_UNSUPPORTED, "pattern is not one-pass (two NFA paths accept the same byte); captures need the general PikeVM pass"};
            if (owner[b] < 0) {
              owner[b] = static_cast<int>(it.state);
              const uint32_t ne = entry(to);       // may grow entryState (vectors resized at loop top)
              next[e * 256 + b] = static_cast<uint8_t>(ne);
              mid[e * 256 + b] = maskId(it.mask);
            }
          }
        };
        if (x.kind == CXG_NFA_BYTE_RANGE) claim(x.lo, x.hi, x.next);
        else for (uint32_t k = 0; k < x.trans_len; k++) { const cxg_nfa_trans& t = nfa.trans[x.trans_off + k]; claim(t.lo, t.hi, t.next); }
      }
    }
    cxgdev::CapHeader ch;
    std::memset(&ch, 0, sizeof ch);
    ch.magic = cxgdev::kBlobMagic; ch.n_entries = static_cast<uint32_t>(entryState.size()); ch.start_entry = 0;
    ch.n_masks = static_cast<uint32_t>(masks.size()); ch.nslots = nfa.capture_count * 2;
    std::vector<uint8_t> cb(sizeof ch, 0);
    auto put = [&](const void* d, size_t n, uint32_t& off) { while (cb.size() % 16) cb.push_back(0); off = static_cast<uint32_t>(cb.size()); const uint8_t* q = static_cast<const uint8_t*>(d); cb.insert(cb.end(), q, q + n); };
    put(next.data(), next.size(), ch.next_off);
    put(mid.data(), mid.size(), ch.maskid_off);
    put(fin.data(), fin.size(), ch.fin_off);
    if (masks.empty()) masks.push_back(0);
    put(masks.data(), masks.size() * 4, ch.masks_off);
    ch.total_bytes = static_cast<uint32_t>(cb.size());
    std::memcpy(cb.data(), &ch, sizeof ch);
    p->capBlob.swap(cb);
    if (spanChain.nops) deriveChainCaps(spanChain, p->capBlob, nfa.capture_count * 2, p->chainCaps);
    } catch (const BuildError& onePassErr) {
      buildBtCaptureImage(p, nfa, hasLook);
      (void)onePassErr;
    }
    p->subSupported = true;
  } catch (const BuildError& e) {
    p->subWhyNot = e.msg;
  }
}

void attachBoundedChain(cxg_program* p, const cxg_nfa& surrogate, const std::vector<std::pair<int, int>>& bounds) {
  if (!p->supported || p->blob.size() < sizeof(cxgdev::BlobHeader)) return;
  cxgdev::BlobHeader h;
  std::memcpy(&h, p->blob.data(), sizeof h);
  if (h.kind != cxgdev::kKindDigit && h.kind != cxgdev::kKindBidir) return;
  if (h.flags & (cxgdev::kFlagChainOrdered | cxgdev::kFlagPrefixLiteral)) return;     // already on a fast path
  try {
    const Dfa anch = determinize(surrogate, surrogate.start_anchored, true, kMaxDfaStates);
    uint8_t syncInfo[256];
    for (int b = 0; b < 256; b++) syncInfo[b] = p->blob[h.info_off + b] & cxgdev::kInfoSync;
    if (h.kind == cxgdev::kKindDigit) for (int b = '0'; b <= '9'; b++) syncInfo[b] = 0;
    cxgdev::ChainAux chain;
    bool complete = false, ordered = false;
    extractChain(anch, syncInfo, chain, complete, ordered);
    if (!complete || !ordered || chain.ncls != 2 || chain.restart_check) return;
    if ((chain.nops & 1u) == 0 || chain.nops < 3 || chain.nops > 7) return;              // run (byte run){1..3}: the unrolled shapes
    for (uint32_t k = 0; k < chain.nops; k++) {
      if (chain.op_kind[k] != ((k & 1u) ? cxgdev::kChainByte : cxgdev::kChainRun) || chain.op_cls[k] != (k & 1u)) return;
      if (chain.cls_kind[chain.op_cls[k]] == cxgdev::kClsSet) return;
    }
    const uint32_t nfields = (chain.nops + 1) / 2;
    if (bounds.size() != nfields) return;
    cxgdev::ChainCaps cc;
    std::memset(&cc, 0, sizeof cc);
    cc.on = 2;
    cc.nruns = static_cast<uint8_t>(nfields - 1);
    for (int i = 0; i < cxgdev::kCapMaxRuns; i++) cc.run_op[i] = static_cast<uint32_t>(i) < nfields - 1 ? static_cast<uint8_t>(2 * i) : 0xFF;
    for (uint32_t f = 0; f < nfields; f++) { cc.src[f] = static_cast<uint8_t>(bounds[f].first); cc.src[8 + f] = static_cast<uint8_t>(bounds[f].second); }
    // new aux section at the end of the image: [256 B per-state flags of the old aux, or zeros][ChainAux of the surrogate]
    std::vector<uint8_t> blob = p->blob;
    while (blob.size() % 16) blob.push_back(0);
    const uint32_t newAux = static_cast<uint32_t>(blob.size());
    if (h.aux_len >= 256) blob.insert(blob.end(), p->blob.begin() + h.aux_off, p->blob.begin() + h.aux_off + 256);
    else blob.insert(blob.end(), 256, 0);
    const uint8_t* cb = reinterpret_cast<const uint8_t*>(&chain);
    blob.insert(blob.end(), cb, cb + sizeof chain);
    while (blob.size() % 16) blob.push_back(0);
    h.aux_off = newAux;
    h.aux_len = static_cast<uint32_t>(blob.size()) - newAux;
    h.flags |= cxgdev::kFlagChainComplete | cxgdev::kFlagChainOrdered | cxgdev::kFlagChainBounded;
    h.total_bytes = static_cast<uint32_t>(blob.size());
    std::memcpy(blob.data(), &h, sizeof h);
    p->blob.swap(blob);
    std::memcpy(p->chainBounds, &cc, sizeof cc);
  } catch (const BuildError&) {
  }
}

void deriveOffsetCaps(cxg_program* p, const cxg_nfa& nfa) {
  p->offCapsOn = 0;
  if (!p->supported || p->nullable || nfa.capture_count < 2 || nfa.capture_count > 16 || nfa.start_unanchored == nfa.start_anchored) return;
  // The spans must be what the reference's FindAllSubmatch reports as group 0: its PikeVM over the whole haystack, plain
  // leftmost-first (meta/findall.go:89-98).  FindAllIndex of a UseBoth program restarts inside matches longer than 100 bytes
  // (find_indices.go:425-431) and the digit prefilter has its run-skip rule: their FindAll spans are not always those.  (Found by
  // the device fuzz: `[\d.]+[x-z]+.*(a|b)`, UseBoth, 230 rows with other spans than the PikeVM's.)
  if (!(p->strategy == CXG_USE_DFA || p->strategy == CXG_USE_NFA || p->strategy == CXG_USE_TEDDY || p->strategy == CXG_USE_CHARCLASS_SEARCHER ||
        p->strategy == CXG_USE_BOUNDED_BACKTRACKER)) return;
  if (p->blob.size() >= sizeof(cxgdev::BlobHeader) && (reinterpret_cast<const cxgdev::BlobHeader*>(p->blob.data())->flags & cxgdev::kFlagBothRestart)) return;
  const uint32_t N = nfa.n_states;
  // the pattern's states: what the anchored start reaches (the unanchored prefix in front of it does not count as a way in)
  std::vector<uint8_t> reach(N, 0);
  std::vector<uint32_t> indeg(N, 0), pred(N, CXG_NFA_INVALID);
  std::vector<uint32_t> st{nfa.start_anchored};
  auto edge = [&](uint32_t from, uint32_t to) { if (to != CXG_NFA_INVALID && to < N) { indeg[to]++; pred[to] = from; if (!reach[to]) { reach[to] = 1; st.push_back(to); } } };
  if (nfa.start_anchored >= N) return;
  reach[nfa.start_anchored] = 1;
  uint32_t match = CXG_NFA_INVALID;
  int perSlot[32] = {0};
  while (!st.empty()) {
    const uint32_t q = st.back(); st.pop_back();
    const cxg_nfa_state& x = nfa.states[q];
    switch (x.kind) {
      case CXG_NFA_MATCH: if (match != CXG_NFA_INVALID && match != q) return; match = q; break;
      case CXG_NFA_BYTE_RANGE: case CXG_NFA_EPSILON: case CXG_NFA_LOOK: edge(q, x.next); break;
      case CXG_NFA_CAPTURE: {
        const uint32_t slot = x.cap_index * 2u + (x.cap_start ? 0u : 1u);
        if (slot >= 32) return;
        perSlot[slot]++;
        edge(q, x.next);
        break;
      }
      case CXG_NFA_SPLIT: edge(q, x.left); edge(q, x.right); break;
      case CXG_NFA_SPARSE: {
        std::vector<uint32_t> seen;
        for (uint32_t k = 0; k < x.trans_len; k++) { const uint32_t t = nfa.trans[x.trans_off + k].next; if (std::find(seen.begin(), seen.end(), t) == seen.end()) { seen.push_back(t); edge(q, t); } }
        break;
      }
      default: break;
    }
  }
  if (match == CXG_NFA_INVALID) return;
  const uint32_t nslots = nfa.capture_count * 2u;
  bool have[32] = {false};
  // one consuming step to exactly one state?
  auto single = [&](const cxg_nfa_state& x, uint32_t& to) -> bool {
    if (x.kind == CXG_NFA_BYTE_RANGE) { to = x.next; return true; }
    if (x.kind != CXG_NFA_SPARSE || x.trans_len == 0) return false;
    to = nfa.trans[x.trans_off].next;
    for (uint32_t k = 1; k < x.trans_len; k++) if (nfa.trans[x.trans_off + k].next != to) return false;
    return true;
  };
  {  // forward: start, then states entered from one place only, until the first branch
    uint32_t cur = nfa.start_anchored;
    int32_t consumed = 0;
    for (uint32_t steps = 0; steps < N + 1 && cur != CXG_NFA_INVALID && cur < N; steps++) {
      if (cur == nfa.start_anchored ? indeg[cur] != 0u : indeg[cur] != 1u) break;
      const cxg_nfa_state& x = nfa.states[cur];
      uint32_t to = CXG_NFA_INVALID;
      if (x.kind == CXG_NFA_EPSILON) cur = x.next;
      else if (x.kind == CXG_NFA_CAPTURE) {
        const uint32_t slot = x.cap_index * 2u + (x.cap_start ? 0u : 1u);
        if (perSlot[slot] == 1) { have[slot] = true; p->offSrc[slot] = 0; p->offDelta[slot] = consumed; }
        cur = x.next;
      } else if (single(x, to)) { consumed++; cur = to; }
      else break;
    }
  }
  {  // backward: Match, then its one predecessor, and so on while every state has one way in and one way out
    uint32_t cur = match;
    int32_t consumed = 0;
    for (uint32_t steps = 0; steps < N + 1; steps++) {
      if (indeg[cur] != 1u) break;
      const uint32_t q = pred[cur];
      if (q == CXG_NFA_INVALID || q >= N) break;
      const cxg_nfa_state& x = nfa.states[q];
      uint32_t to = CXG_NFA_INVALID;
      if (x.kind == CXG_NFA_EPSILON) cur = q;
      else if (x.kind == CXG_NFA_CAPTURE) {
        const uint32_t slot = x.cap_index * 2u + (x.cap_start ? 0u : 1u);
        if (perSlot[slot] == 1 && !have[slot]) { have[slot] = true; p->offSrc[slot] = 1; p->offDelta[slot] = -consumed; }
        cur = q;
      } else if (single(x, to) && to == cur) { consumed++; cur = q; }
      else break;
      if (cur == nfa.start_anchored) break;
    }
  }
  for (uint32_t k = 2; k < nslots; k++) if (!have[k]) return;
  p->offSrc[0] = 0; p->offDelta[0] = 0; p->offSrc[1] = 1; p->offDelta[1] = 0;
  p->offCapsOn = 1;
}

void buildProgramFromCharClass(cxg_program* p, const uint8_t membership[256], uint32_t minMatch, bool pairs) {
  p->strategy = CXG_USE_CHARCLASS_SEARCHER;
  p->ngroups = 1;
  p->supported = false;
  if (minMatch != 1) { p->whyNot = "minMatch != 1"; return; }
  cxgdev::BlobHeader h;
  std::memset(&h, 0, sizeof h);
  h.magic = cxgdev::kBlobMagic;
  h.kind = cxgdev::kKindCharClass;
  h.ngroups = 1;
  std::vector<uint8_t> blob(sizeof h, 0);
  h.info_off = static_cast<uint32_t>(blob.size());
  for (int b = 0; b < 256; b++) blob.push_back(membership[b] ? cxgdev::kInfoMember : (pairs ? 0 : cxgdev::kInfoSync));   // (pairs: no byte synchronises)
  {  // membership as ranges: the wave kernel (scan_charclass_wave.hip) classifies with SWAR range tests
    cxgdev::CharClassAux ax;
    std::memset(&ax, 0, sizeof ax);
    bool ok = true;
    for (int b = 0; b < 256 && ok; b++) {
      if (!membership[b] || (b > 0 && membership[b - 1])) continue;
      int e = b;
      while (e + 1 < 256 && membership[e + 1]) e++;
      if (e > 127 || ax.nr >= 4) { ok = false; break; }
      ax.lo[ax.nr] = static_cast<uint8_t>(b); ax.hi[ax.nr] = static_cast<uint8_t>(e); ax.nr++;
    }
    if (!ok || ax.nr < 1) {
      // ... or the COMPLEMENT of such a union with every byte >= 0x80 a member: `\S`, `[^,]`, `[^"]` as byte sets (round 4)
      std::memset(&ax, 0, sizeof ax);
      ok = true;
      for (int b = 128; b < 256 && ok; b++) ok = membership[b] != 0;
      for (int b = 0; b < 128 && ok; b++) {
        if (membership[b] || (b > 0 && !membership[b - 1])) continue;
        int e = b;
        while (e + 1 < 128 && !membership[e + 1]) e++;
        if (ax.nr >= 4) { ok = false; break; }
        ax.lo[ax.nr] = static_cast<uint8_t>(b); ax.hi[ax.nr] = static_cast<uint8_t>(e); ax.nr++;
      }
      ax.neg = 1;
    }
    ax.pairs = pairs ? 1u : 0u;
    if (pairs && !(ok && ax.nr == 1 && ax.neg == 0 && ax.lo[0] == ax.hi[0])) { p->whyNot = "internal: quote-pair program without its one-byte class"; return; }
    if (ok && ax.nr >= 1) {
      h.flags |= cxgdev::kFlagCcRanges;
      h.aux_off = static_cast<uint32_t>(blob.size());
      const uint8_t* ab = reinterpret_cast<const uint8_t*>(&ax);
      blob.insert(blob.end(), ab, ab + sizeof ax);
      h.aux_len = sizeof ax;
    }
  }
  h.total_bytes = static_cast<uint32_t>(blob.size());
  std::memcpy(blob.data(), &h, sizeof h);
  p->blob.swap(blob);
  p->supported = true;
}

namespace {
// Thompson NFA of `lit0|lit1|...` (alternation in pattern-ID order) behind the unanchored prefix: for a prefix-free
// set at most one literal matches at a position, so leftmost-first over this NFA is what Teddy.FindMatch + the FindAll
// loop report (prefilter/teddy.go:391-444, meta/find_indices.go:925-951).  Only the transducer kernel uses it — the
// fallback of the literal kernels for input they cannot take (match-dense, no synchronising bytes).
HostNfa literalNfa(const std::vector<std::vector<uint8_t>>& lits) {
  HostNfa n;
  auto blank = [](uint8_t kind) { cxg_nfa_state s; std::memset(&s, 0, sizeof s); s.kind = kind; s.next = s.left = s.right = CXG_NFA_INVALID; return s; };
  auto add = [&](cxg_nfa_state s) { n.states.push_back(s); return static_cast<uint32_t>(n.states.size() - 1); };
  const uint32_t match = add(blank(CXG_NFA_MATCH));
  std::vector<uint32_t> heads;
  for (const auto& l : lits) {
    uint32_t next = match;
    for (size_t i = l.size(); i-- > 0;) {
      cxg_nfa_state b = blank(CXG_NFA_BYTE_RANGE);
      b.lo = b.hi = l[i]; b.next = next;
      next = add(b);
    }
    heads.push_back(next);
  }
  uint32_t alt = heads.back();
  for (size_t i = heads.size() - 1; i-- > 0;) { cxg_nfa_state sp = blank(CXG_NFA_SPLIT); sp.left = heads[i]; sp.right = alt; alt = add(sp); }
  n.startAnchored = alt;
  cxg_nfa_state any = blank(CXG_NFA_BYTE_RANGE);
  any.lo = 0; any.hi = 255;
  const uint32_t anyId = add(any);
  cxg_nfa_state pre = blank(CXG_NFA_SPLIT);
  pre.left = alt; pre.right = anyId;
  n.startUnanchored = add(pre);
  n.states[anyId].next = n.startUnanchored;
  n.captureCount = 1;
  return n;
}
}  // namespace

void buildProgramFromLiterals(cxg_program* p, const std::vector<std::vector<uint8_t>>& lits) {
  p->strategy = CXG_USE_TEDDY;
  p->ngroups = 1;
  buildLiteralImage(p, lits, 2);
  if (!p->supported) return;
  try {                                              // transducer image: the literal kernels' fallback
    const HostNfa ln = literalNfa(lits);
    const cxg_nfa lv = ln.view();
    HostNfa rn = reverseOf(lv);
    cxg_nfa rvw = rn.view();
    const Dfa rv = determinize(rvw, rvw.start_anchored, false, kMaxDfaStates);
    if (!buildFsmImage(lv, rv, 0u, p->fsmBlob, p->fsmWhyNot)) p->fsmBlob.clear();
  } catch (const BuildError& e) { p->fsmBlob.clear(); p->fsmWhyNot = e.msg; }
}

// Device image of a literal set (kKindTeddy): fingerprint tables + the literals for exact verification.
// NewTeddy / buildMasks (prefilter/teddy.go:189-311): 2..32 literals of >= 3 bytes, bucket = id mod 8,
// 2-byte fingerprint.  33..64 literals are the reference's Fat Teddy (prefilter/teddy_fat.go:127-253, bucket = id mod 16):
// the device keeps ONE 8-bit mask per fingerprint byte and folds bucket b and b+8 together.  That only widens the
// candidate set; every candidate is verified against the literals, and for a prefix-free set (required below) at most
// one literal matches at a position, so the bucket order of verifyBucket (teddy_fat.go:471-487) cannot be observed.
// > 64 literals (Aho-Corasick) are outside the device subset.
// min_count 1: a single literal of a UseDFA program (buildProgramFromNfa) searched with the same kernels.
// Literal tables of the Teddy kernels (walk.hpp TeddyAux + arrays), appended to `aux`; false + why when the set is
// outside the device subset.
bool makeLiteralAux(const std::vector<std::vector<uint8_t>>& lits, size_t min_count, std::vector<uint8_t>& aux, std::string& why, bool fold) {
  if (lits.size() < min_count || lits.size() > 64) { why = "Teddy takes 2..64 literals"; return false; }
  size_t minlen = SIZE_MAX, maxlen = 0;
  for (auto& l : lits) { minlen = std::min(minlen, l.size()); maxlen = std::max(maxlen, l.size()); }
  if (minlen < 3) { why = "Teddy literal shorter than 3 bytes"; return false; }
  if (maxlen > 255) { why = "Teddy literal longer than 255 bytes"; return false; }
  for (size_t i = 0; i < lits.size(); i++)
    for (size_t j = 0; j < lits.size(); j++)
      if (i != j && lits[i].size() <= lits[j].size() && std::equal(lits[i].begin(), lits[i].end(), lits[j].begin())) {
        why = "literal set is not prefix-free (the reference's verification order becomes observable)";
        return false;
      }
  // Buckets.  The reference deals literals round-robin (bucket = id mod 8, teddy.go:283; mod 16 for Fat Teddy,
  // teddy_fat.go:209) and filters with nibble-mask products.  Neither is observable here: the set is prefix-free, so at
  // most one literal matches at a position whatever order the buckets are verified in, and any superset of the match
  // starts is a valid candidate set.  The device therefore (i) keeps EXACT per-byte masks (its lookup table has 256
  // entries anyway) and (ii) puts literals that share their first three bytes into one bucket and deals the distinct
  // prefixes, in lexicographic order, contiguously over the 8 buckets: a bucket's false candidates are the mixed
  // triples of its prefixes, and neighbours in that order share leading bytes.  48 literals with 8 distinct prefixes
  // then filter exactly on three bytes (round robin: 7 % of all bytes of a word-like text were candidates).
  std::vector<uint8_t> bucketOf(lits.size(), 0);
  uint32_t nb = 0;
  {
    std::map<std::array<uint8_t, 3>, std::vector<size_t>> byPrefix;
    for (size_t id = 0; id < lits.size(); id++) byPrefix[{lits[id][0], lits[id][1], lits[id][2]}].push_back(id);
    const size_t ng = byPrefix.size();
    nb = static_cast<uint32_t>(std::min<size_t>(8, ng));
    size_t g = 0;
    for (auto& kv : byPrefix) { for (size_t id : kv.second) bucketOf[id] = static_cast<uint8_t>(g * nb / ng); g++; }
  }
  uint16_t ab[256] = {0};
  for (size_t id = 0; id < lits.size(); id++) {
    ab[lits[id][0]] |= static_cast<uint16_t>(1u << bucketOf[id]);
    ab[lits[id][1]] |= static_cast<uint16_t>(0x100u << bucketOf[id]);
    if (fold) {                                      // folded set (lower-case letters stand for both cases): the other case is a candidate too
      if (lits[id][0] >= 'a' && lits[id][0] <= 'z') ab[lits[id][0] ^ 0x20u] |= static_cast<uint16_t>(1u << bucketOf[id]);
      if (lits[id][1] >= 'a' && lits[id][1] <= 'z') ab[lits[id][1] ^ 0x20u] |= static_cast<uint16_t>(0x100u << bucketOf[id]);
    }
  }
  cxgdev::TeddyAux ax;
  std::memset(&ax, 0, sizeof ax);
  ax.nlits = static_cast<uint32_t>(lits.size()); ax.nbuckets = nb; ax.minlen = static_cast<uint32_t>(minlen); ax.maxlen = static_cast<uint32_t>(maxlen);
  aux.assign(sizeof ax, 0);
  auto align = [&](size_t a) { while (aux.size() % a) aux.push_back(0); };
  ax.ab_off = static_cast<uint32_t>(aux.size());
  aux.insert(aux.end(), reinterpret_cast<uint8_t*>(ab), reinterpret_cast<uint8_t*>(ab) + sizeof ab);
  ax.order_off = static_cast<uint32_t>(aux.size());
  for (uint32_t b = 0; b < nb; b++) for (size_t id = 0; id < lits.size(); id++) if (bucketOf[id] == b) aux.push_back(static_cast<uint8_t>(id));
  align(4);
  ax.lens_off = static_cast<uint32_t>(aux.size());
  for (auto& l : lits) aux.push_back(static_cast<uint8_t>(l.size()));
  align(4);
  ax.bucket_off = static_cast<uint32_t>(aux.size());
  for (size_t id = 0; id < lits.size(); id++) aux.push_back(bucketOf[id]);
  align(4);
  ax.off_off = static_cast<uint32_t>(aux.size());
  { uint16_t o = 0; for (auto& l : lits) { aux.push_back(o & 0xFF); aux.push_back(o >> 8); o = static_cast<uint16_t>(o + l.size()); } }
  align(4);
  ax.bytes_off = static_cast<uint32_t>(aux.size());
  for (auto& l : lits) aux.insert(aux.end(), l.begin(), l.end());
  ax.bytes_len = static_cast<uint32_t>(aux.size()) - ax.bytes_off;
  align(16);
  if (aux.size() > 4096) { why = "literal table exceeds the kernels' LDS budget (4 KiB)"; return false; }
  std::memcpy(aux.data(), &ax, sizeof ax);
  return true;
}

// The tables of scan_teddy_pair.hip (walk.hpp PairImage): the entry of every byte pair — AB / BC / CD / DE: the pair is bytes 0,1 / 1,2 / 2,3 /
// 3,4 of a literal; A2: its second byte begins one; E1: its first byte is a fifth byte; S1 / S2: the byte lies outside the alphabet; a
// literal shorter than the byte asked for lets every value pass —, the verification slots by first byte, the slot records.  A folded set
// keeps lower-case literals: a letter stands for both cases.  tests/test_teddy_pair_cpu.py restates the table in numpy and compares.
static std::vector<uint8_t> buildPairImage(const std::vector<std::vector<uint8_t>>& lits, bool fold, const bool* inAlpha) {
  std::vector<uint8_t> img(sizeof(cxgdev::PairImage), 0);
  cxgdev::PairImage* I = reinterpret_cast<cxgdev::PairImage*>(img.data());
  auto lower = [&](uint32_t c) { return fold && c >= 'a' && c <= 'z'; };
  auto same = [&](uint32_t b, uint32_t c) { return b == c || (lower(c) && (b | 0x20u) == c); };
  uint8_t F[256], G[256];
  for (uint32_t b = 0; b < 256; b++) {
    uint32_t f = inAlpha[b] ? 0u : 0x40u, g = inAlpha[b] ? 0u : 0x80u;
    for (auto& l : lits) {
      if (same(b, l[0])) g |= 2u;                                     // A2
      if (l.size() == 3) { if (same(b, l[2])) f |= 4u; f |= 0x30u; }  // CD with any second byte; E1 and DE: nothing to ask
      else if (l.size() == 4) { if (same(b, l[3])) f |= 0x20u; f |= 0x10u; }   // DE with any second byte; E1: nothing to ask
      else if (same(b, l[4])) f |= 0x10u;                             // E1
    }
    F[b] = static_cast<uint8_t>(f); G[b] = static_cast<uint8_t>(g);
  }
  for (uint32_t b1 = 0; b1 < 256; b1++)
    for (uint32_t b0 = 0; b0 < 256; b0++) I->tab[cxgdev::pair_addr(b0, b1)] = F[b0] | G[b1];
  for (auto& l : lits)
    for (size_t k = 0; k < 4 && k + 1 < l.size(); k++) {               // the exact pairs: AB (bit 0), BC (3), CD (2), DE (5)
      const uint8_t bit = k == 0 ? 1 : k == 1 ? 8 : k == 2 ? 4 : 0x20;
      for (uint32_t i0 = 0; i0 < (lower(l[k]) ? 2u : 1u); i0++)
        for (uint32_t i1 = 0; i1 < (lower(l[k + 1]) ? 2u : 1u); i1++)
          I->tab[cxgdev::pair_addr(l[k] ^ (i0 ? 0x20u : 0u), l[k + 1] ^ (i1 ? 0x20u : 0u))] |= bit;
    }
  uint32_t maxrun = 0;
  for (uint32_t b = 0; b < 256; b++) {                                 // (an upper-case letter asks as its lower-case twin)
    const uint32_t nb = (fold && b >= 'A' && b <= 'Z') ? (b | 0x20u) : b;
    uint32_t beg = 0, cnt = 0;
    for (auto& l : lits) { beg += l[0] < nb ? 1u : 0u; cnt += l[0] == nb ? 1u : 0u; }
    I->FB[b] = beg | ((beg + cnt) << 8) | (inAlpha[b] ? 0u : 0x1000000u);
    maxrun = std::max(maxrun, cnt);
  }
  I->maxrun = maxrun;
  for (size_t id = 0; id < lits.size() && id < 64; id++) {
    const auto& l = lits[id];
    uint32_t slot = 0;
    for (size_t o = 0; o < lits.size(); o++) slot += (lits[o][0] < l[0] || (lits[o][0] == l[0] && o < id)) ? 1u : 0u;
    uint32_t* x = I->litx[slot];
    for (uint32_t k = 0; k < 3; k++)
      for (uint32_t b = 0; b < 4; b++) if (4 * k + b < l.size()) {
        const uint32_t c = l[4 * k + b];
        x[k] |= c << (8 * b);
        x[3 + k] |= (lower(c) ? 0xDFu : 0xFFu) << (8 * b);
      }
    x[6] = static_cast<uint32_t>(l.size()); x[7] = static_cast<uint32_t>(id);
  }
  return img;
}

void buildLiteralImage(cxg_program* p, const std::vector<std::vector<uint8_t>>& lits, size_t min_count, bool fold) {
  p->supported = false;
  std::vector<uint8_t> aux;
  if (!makeLiteralAux(lits, min_count, aux, p->whyNot, fold)) return;
  cxgdev::BlobHeader h;
  std::memset(&h, 0, sizeof h);
  h.magic = cxgdev::kBlobMagic;
  h.kind = cxgdev::kKindTeddy;
  h.ngroups = 1;
  std::vector<uint8_t> blob(sizeof h, 0);
  h.info_off = static_cast<uint32_t>(blob.size());
  bool inAlpha[256] = {false};
  for (auto& l : lits) for (uint8_t b : l) { inAlpha[b] = true; if (fold && b >= 'a' && b <= 'z') inAlpha[b ^ 0x20u] = true; }
  for (int b = 0; b < 256; b++) blob.push_back(inAlpha[b] ? 0 : cxgdev::kInfoSync);
  h.aux_off = static_cast<uint32_t>(blob.size());
  h.aux_len = static_cast<uint32_t>(aux.size());
  blob.insert(blob.end(), aux.begin(), aux.end());
  while (blob.size() % 16) blob.push_back(0);
  {                                                                  // scan_teddy_pair.hip's tables: the LAST section of the image
    const std::vector<uint8_t> img = buildPairImage(lits, fold, inAlpha);
    reinterpret_cast<cxgdev::TeddyAux*>(blob.data() + h.aux_off)->pair_off = static_cast<uint32_t>(blob.size());
    blob.insert(blob.end(), img.begin(), img.end());
  }
  h.total_bytes = static_cast<uint32_t>(blob.size());
  std::memcpy(blob.data(), &h, sizeof h);
  p->blob.swap(blob);
  p->supported = true;
}

}  // namespace cxg
