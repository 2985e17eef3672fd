// ORACLE — TEST INFRASTRUCTURE ONLY (see nfa.hpp header for the reference map).
#include "nfa.hpp"

#include <algorithm>
#include <sstream>

namespace orc {

namespace {

struct Builder {  // nfa/builder.go:34-339
  std::vector<NState> states;
  uint64_t bits[4] = {0, 0, 0, 0};  // ByteClassSet (nfa/alphabet.go:100-146)

  void setBit(uint8_t b) { bits[b / 64] |= (uint64_t{1} << (b % 64)); }
  void setRange(uint8_t lo, uint8_t hi) {  // alphabet.go:118-123
    if (lo > 0) setBit(lo - 1);
    setBit(hi);
  }
  StateID add(NState s) { states.push_back(std::move(s)); return static_cast<StateID>(states.size() - 1); }

  StateID addMatch() { NState s; s.kind = StateMatch; return add(s); }
  StateID addByteRange(uint8_t lo, uint8_t hi, StateID next) {
    setRange(lo, hi);
    NState s; s.kind = StateByteRange; s.lo = lo; s.hi = hi; s.next = next; return add(s);
  }
  StateID addSparse(const std::vector<Transition>& tr) {
    for (auto& t : tr) setRange(t.lo, t.hi);
    NState s; s.kind = StateSparse; s.trans = tr; return add(s);
  }
  StateID addSplit(StateID l, StateID r, bool quant) {
    NState s; s.kind = StateSplit; s.left = l; s.right = r; s.quantSplit = quant; return add(s);
  }
  StateID addEpsilon(StateID next) { NState s; s.kind = StateEpsilon; s.next = next; return add(s); }
  StateID addCapture(uint32_t idx, bool isStart, StateID next) {
    NState s; s.kind = StateCapture; s.capIndex = idx; s.capStart = isStart; s.next = next; return add(s);
  }
  StateID addLook(Look l, StateID next) { NState s; s.kind = StateLook; s.look = l; s.next = next; return add(s); }

  bool patch(StateID id, StateID target) {  // builder.go:189-212
    if (id >= states.size()) return false;
    NState& s = states[id];
    switch (s.kind) {
      case StateByteRange: case StateEpsilon: case StateCapture: case StateLook:
      case StateRuneAny: case StateRuneAnyNotNL:
        s.next = target; return true;
      default: return false;
    }
  }
};

struct Frag { StateID start, end; };

struct Compiler {  // nfa/compile.go
  Builder b;
  int depth = 0;
  int captureCount = 0;

  [[noreturn]] static void fail(const std::string& m) { throw CompileError{m}; }

  void countCaptures(const ReP& re) {  // compile.go:1703-1721
    switch (re->op) {
      case OpCapture:
        if (re->cap > captureCount) captureCount = re->cap;
        for (auto& s : re->sub) countCaptures(s);
        break;
      case OpConcat: case OpAlternate:
        for (auto& s : re->sub) countCaptures(s);
        break;
      case OpStar: case OpPlus: case OpQuest: case OpRepeat:
        if (!re->sub.empty()) countCaptures(re->sub[0]);
        break;
      default: break;
    }
  }

  static int encodeRune(uint8_t* buf, int r) {  // compile.go:1590-1612
    if (r < 0x80) { buf[0] = static_cast<uint8_t>(r); return 1; }
    if (r < 0x800) { buf[0] = 0xC0 | (r >> 6); buf[1] = 0x80 | (r & 0x3F); return 2; }
    if (r < 0x10000) {
      buf[0] = 0xE0 | (r >> 12); buf[1] = 0x80 | ((r >> 6) & 0x3F); buf[2] = 0x80 | (r & 0x3F);
      return 3;
    }
    buf[0] = 0xF0 | (r >> 18); buf[1] = 0x80 | ((r >> 12) & 0x3F);
    buf[2] = 0x80 | ((r >> 6) & 0x3F); buf[3] = 0x80 | (r & 0x3F);
    return 4;
  }

  Frag compileSingleRune(int r) {  // compile.go:333-355
    uint8_t buf[4];
    int n = encodeRune(buf, r);
    StateID prev = kInvalidState, first = kInvalidState;
    for (int i = 0; i < n; i++) {
      StateID id = b.addByteRange(buf[i], buf[i], kInvalidState);
      if (first == kInvalidState) first = id;
      if (prev != kInvalidState) b.patch(prev, id);
      prev = id;
    }
    return {first, prev};
  }

  static bool isASCIILetter(int r) { return (r >= 'a' && r <= 'z') || (r >= 'A' && r <= 'Z'); }

  Frag compileLiteral(const ReP& re) {  // compile.go:237-331
    if (re->rune.empty()) return compileEmptyMatch();
    bool fold = re->flags & FoldCase;
    StateID prev = kInvalidState, first = kInvalidState;
    for (int r : re->rune) {
      if (fold && isASCIILetter(r)) {
        int upper = (r >= 'a' && r <= 'z') ? r - 32 : r;
        int lower = (r >= 'A' && r <= 'Z') ? r + 32 : r;
        Frag u = compileSingleRune(upper);
        Frag l = compileSingleRune(lower);
        StateID join = b.addEpsilon(kInvalidState);
        b.patch(u.end, join);
        b.patch(l.end, join);
        StateID split = b.addSplit(u.start, l.start, false);
        if (prev == kInvalidState) first = split; else b.patch(prev, split);
        prev = join;
      } else {
        uint8_t buf[4];
        int n = encodeRune(buf, r);
        for (int i = 0; i < n; i++) {
          StateID id = b.addByteRange(buf[i], buf[i], kInvalidState);
          if (first == kInvalidState) first = id;
          if (prev != kInvalidState) b.patch(prev, id);
          prev = id;
        }
      }
    }
    return {first, prev};
  }

  // ---- UTF-8 automata: classes that reach past U+007F, and `.` ---------------------------------------------------------
  struct SuffixCache {  // nfa/utf8_suffix.go:20-123: direct-mapped, 64 entries, FNV-1a over (target, lo, hi); collisions overwrite
    struct Entry { bool used = false; StateID from = 0; uint8_t lo = 0, hi = 0; StateID val = 0; };
    Entry e[64];
    static int slot(StateID from, uint8_t lo, uint8_t hi) {  // utf8_suffix.go:70-77
      uint64_t h = 14695981039346656037ull;
      h = (h ^ static_cast<uint64_t>(from)) * 1099511628211ull;
      h = (h ^ static_cast<uint64_t>(lo)) * 1099511628211ull;
      h = (h ^ static_cast<uint64_t>(hi)) * 1099511628211ull;
      return static_cast<int>(h % 64u);
    }
    StateID getOrCreate(Builder& b, StateID target, uint8_t lo, uint8_t hi) {  // utf8_suffix.go:112-123
      Entry& x = e[slot(target, lo, hi)];
      if (x.used && x.from == target && x.lo == lo && x.hi == hi) return x.val;
      StateID id = b.addByteRange(lo, hi, target);
      x.used = true; x.from = target; x.lo = lo; x.hi = hi; x.val = id;
      return id;
    }
  };

  Frag compileUTF8Any(bool includeNL) {  // compile.go:1142-1222 (the default configuration: no ASCIIOnly, no UseRuneStates)
    StateID endState = b.addEpsilon(kInvalidState);
    SuffixCache cache;
    static const uint8_t seqs[8][4][2] = {
        {{0xC2, 0xDF}, {0x80, 0xBF}, {0, 0}, {0, 0}},
        {{0xE0, 0xE0}, {0xA0, 0xBF}, {0x80, 0xBF}, {0, 0}},
        {{0xE1, 0xEC}, {0x80, 0xBF}, {0x80, 0xBF}, {0, 0}},
        {{0xED, 0xED}, {0x80, 0x9F}, {0x80, 0xBF}, {0, 0}},
        {{0xEE, 0xEF}, {0x80, 0xBF}, {0x80, 0xBF}, {0, 0}},
        {{0xF0, 0xF0}, {0x90, 0xBF}, {0x80, 0xBF}, {0x80, 0xBF}},
        {{0xF1, 0xF3}, {0x80, 0xBF}, {0x80, 0xBF}, {0x80, 0xBF}},
        {{0xF4, 0xF4}, {0x80, 0x8F}, {0x80, 0xBF}, {0x80, 0xBF}}};
    static const int seqLen[8] = {2, 3, 3, 3, 3, 4, 4, 4};
    std::vector<StateID> branches;
    if (includeNL) branches.push_back(b.addByteRange(0x00, 0x7F, endState));
    else branches.push_back(b.addSparse({{0x00, 0x09, endState}, {0x0B, 0x7F, endState}}));
    for (int q = 0; q < 8; q++) {
      StateID target = endState;
      for (int i = seqLen[q] - 1; i >= 0; i--) target = cache.getOrCreate(b, target, seqs[q][i][0], seqs[q][i][1]);
      branches.push_back(target);
    }
    // bytes that begin no sequence match alone (the lead bytes C2..F4 do not: :1205-1209)
    branches.push_back(b.addSparse({{0x80, 0xBF, endState}, {0xC0, 0xC1, endState}, {0xF5, 0xFF, endState}}));
    return {buildSplitChain(branches, 0), endState};
  }

  std::vector<StateID> buildUTF8NonASCIIBranches(StateID endState) {  // compile.go:845-917 (no suffix sharing here)
    std::vector<StateID> br;
    auto cont = [&](StateID next) { return b.addByteRange(0x80, 0xBF, next); };
    { StateID c1 = cont(endState); br.push_back(b.addByteRange(0xC2, 0xDF, c1)); }
    { StateID c2 = cont(endState); StateID c1 = b.addByteRange(0xA0, 0xBF, c2); br.push_back(b.addByteRange(0xE0, 0xE0, c1)); }
    { StateID c2 = cont(endState); StateID c1 = cont(c2); br.push_back(b.addByteRange(0xE1, 0xEC, c1)); }
    { StateID c2 = cont(endState); StateID c1 = b.addByteRange(0x80, 0x9F, c2); br.push_back(b.addByteRange(0xED, 0xED, c1)); }
    { StateID c2 = cont(endState); StateID c1 = cont(c2); br.push_back(b.addByteRange(0xEE, 0xEF, c1)); }
    { StateID c3 = cont(endState); StateID c2 = cont(c3); StateID c1 = b.addByteRange(0x90, 0xBF, c2); br.push_back(b.addByteRange(0xF0, 0xF0, c1)); }
    { StateID c3 = cont(endState); StateID c2 = cont(c3); StateID c1 = cont(c2); br.push_back(b.addByteRange(0xF1, 0xF3, c1)); }
    { StateID c3 = cont(endState); StateID c2 = cont(c3); StateID c1 = b.addByteRange(0x80, 0x8F, c2); br.push_back(b.addByteRange(0xF4, 0xF4, c1)); }
    return br;
  }

  void utf8Range2(int lo, int hi, StateID endState, std::vector<StateID>& starts) {  // compile.go:663-701
    const uint8_t loLead = 0xC0 | (lo >> 6), loCont = 0x80 | (lo & 0x3F), hiLead = 0xC0 | (hi >> 6), hiCont = 0x80 | (hi & 0x3F);
    if (loLead == hiLead) {
      StateID cont = b.addByteRange(loCont, hiCont, endState);
      starts.push_back(b.addByteRange(loLead, loLead, cont));
      return;
    }
    StateID cont1 = b.addByteRange(loCont, 0xBF, endState);
    starts.push_back(b.addByteRange(loLead, loLead, cont1));
    if (hiLead > loLead + 1) {
      StateID contM = b.addByteRange(0x80, 0xBF, endState);
      starts.push_back(b.addByteRange(loLead + 1, hiLead - 1, contM));
    }
    StateID cont2 = b.addByteRange(0x80, hiCont, endState);
    starts.push_back(b.addByteRange(hiLead, hiLead, cont2));
  }

  void utf8Range3Simple(int lo, int hi, StateID endState, std::vector<StateID>& starts) {  // compile.go:740-792
    const int loLead = 0xE0 | (lo >> 12), loC1 = 0x80 | ((lo >> 6) & 0x3F), loC2 = 0x80 | (lo & 0x3F);
    const int hiLead = 0xE0 | (hi >> 12), hiC1 = 0x80 | ((hi >> 6) & 0x3F), hiC2 = 0x80 | (hi & 0x3F);
    auto seq = [&](int lead, int c1, int c2lo, int c2hi) {
      StateID s2 = b.addByteRange(static_cast<uint8_t>(c2lo), static_cast<uint8_t>(c2hi), endState);
      StateID s1 = b.addByteRange(static_cast<uint8_t>(c1), static_cast<uint8_t>(c1), s2);
      starts.push_back(b.addByteRange(static_cast<uint8_t>(lead), static_cast<uint8_t>(lead), s1));
    };
    if (loLead == hiLead && loC1 == hiC1) { seq(loLead, loC1, loC2, hiC2); return; }
    if (loLead == hiLead) {
      for (int c1 = loC1; c1 <= hiC1; c1++) seq(loLead, c1, c1 == loC1 ? loC2 : 0x80, c1 == hiC1 ? hiC2 : 0xBF);   // :922-934
      return;
    }
    for (int lead = loLead; lead <= hiLead; lead++) {
      const int c1Lo = lead == loLead ? loC1 : lead == 0xE0 ? 0xA0 : 0x80;   // :937-946
      const int c1Hi = lead == hiLead ? hiC1 : lead == 0xED ? 0x9F : 0xBF;   // :949-958
      for (int c1 = c1Lo; c1 <= c1Hi; c1++)
        seq(lead, c1, (lead == loLead && c1 == loC1) ? loC2 : 0x80, (lead == hiLead && c1 == hiC1) ? hiC2 : 0xBF);   // :960-972
    }
  }

  void utf8Range3(int lo, int hi, StateID endState, std::vector<StateID>& starts) {  // compile.go:706-737
    if (lo <= 0xD7FF && hi >= 0xE000) { utf8Range3Simple(lo, 0xD7FF, endState, starts); utf8Range3Simple(0xE000, hi, endState, starts); return; }
    if (lo >= 0xD800 && hi <= 0xDFFF) return;
    if (lo >= 0xD800 && lo <= 0xDFFF) lo = 0xE000;
    if (hi >= 0xD800 && hi <= 0xDFFF) hi = 0xD7FF;
    if (lo > hi) return;
    utf8Range3Simple(lo, hi, endState, starts);
  }

  void utf8Range4(int lo, int hi, StateID endState, std::vector<StateID>& starts) {  // compile.go:796-840: whole lead bytes, not the exact range
    if (hi > 0x10FFFF) hi = 0x10FFFF;
    if (lo < 0x10000) lo = 0x10000;
    if (lo > hi) return;
    const int loLead = 0xF0 | (lo >> 18), hiLead = 0xF0 | (hi >> 18);
    for (int lead = loLead; lead <= hiLead; lead++) {
      const uint8_t c1Lo = lead == 0xF0 ? 0x90 : 0x80, c1Hi = lead == 0xF4 ? 0x8F : 0xBF;
      StateID c3 = b.addByteRange(0x80, 0xBF, endState);
      StateID c2 = b.addByteRange(0x80, 0xBF, c3);
      StateID c1 = b.addByteRange(c1Lo, c1Hi, c2);
      starts.push_back(b.addByteRange(static_cast<uint8_t>(lead), static_cast<uint8_t>(lead), c1));
    }
  }

  std::vector<StateID> compileUTF8Range(int lo, int hi, StateID endState) {  // compile.go:600-654
    std::vector<StateID> starts;
    if (lo <= 0x7F) { starts.push_back(b.addByteRange(static_cast<uint8_t>(lo), static_cast<uint8_t>(std::min(hi, 0x7F)), endState)); lo = 0x80; }
    if (lo > hi) return starts;
    if (lo <= 0x7FF) { utf8Range2(lo, std::min(hi, 0x7FF), endState, starts); lo = 0x800; }
    if (lo > hi) return starts;
    if (lo <= 0xFFFF) { utf8Range3(lo, std::min(hi, 0xFFFF), endState, starts); lo = 0x10000; }
    if (lo > hi) return starts;
    utf8Range4(lo, hi, endState, starts);
    return starts;
  }

  Frag compileUnicodeClassLarge(const std::vector<int>& ranges) {  // compile.go:491-590
    std::vector<Transition> ascii;
    std::vector<std::pair<int, int>> rest;
    for (size_t i = 0; i + 1 < ranges.size(); i += 2) {
      const int lo = ranges[i], hi = ranges[i + 1];
      if (hi < 0x80) ascii.push_back({static_cast<uint8_t>(lo), static_cast<uint8_t>(hi), kInvalidState});
      else if (lo >= 0x80) rest.push_back({lo, hi});
      else { ascii.push_back({static_cast<uint8_t>(lo), 0x7F, kInvalidState}); rest.push_back({0x80, hi}); }
    }
    const bool coversAllNonASCII = rest.size() == 1 && rest[0].first <= 0x80 && rest[0].second >= 0x10FFFF;
    StateID target = b.addEpsilon(kInvalidState);
    std::vector<StateID> alts;
    if (!ascii.empty()) {
      for (auto& t : ascii) t.next = target;
      if (ascii.size() == 1) alts.push_back(b.addByteRange(ascii[0].lo, ascii[0].hi, target));
      else alts.push_back(b.addSparse(ascii));
    }
    if (!rest.empty()) {
      if (coversAllNonASCII) {
        for (StateID s : buildUTF8NonASCIIBranches(target)) alts.push_back(s);
        alts.push_back(b.addByteRange(0x80, 0xFF, target));      // any other byte >= 0x80 alone (:557-567)
      } else {
        for (auto& r : rest) for (StateID s : compileUTF8Range(r.first, r.second, target)) alts.push_back(s);
      }
    }
    if (alts.empty()) return compileNoMatch();
    if (alts.size() == 1) return {alts[0], target};
    return {buildSplitChain(alts, 0), target};
  }

  Frag compileUnicodeClass(const std::vector<int>& ranges) {  // compile.go:440-481
    if (ranges.empty()) return compileNoMatch();
    int64_t total = 0;
    for (size_t i = 0; i + 1 < ranges.size(); i += 2) {
      total += static_cast<int64_t>(ranges[i + 1]) - ranges[i] + 1;
      if (total > 256) return compileUnicodeClassLarge(ranges);
    }
    std::vector<ReP> alts;                                        // small classes: an alternation of their characters
    for (size_t i = 0; i + 1 < ranges.size(); i += 2)
      for (int r = ranges[i]; r <= ranges[i + 1]; r++) {
        auto lit = std::make_shared<Regexp>();
        lit->op = OpLiteral; lit->rune = {r};
        alts.push_back(lit);
      }
    if (alts.size() == 1) return compile(alts[0]);
    return compileAlternate(alts);
  }

  Frag compileCharClass(const std::vector<int>& ranges) {  // compile.go:384-437
    if (ranges.empty()) return compileNoMatch();
    for (int r : ranges)
      if (r > 127) return compileUnicodeClass(ranges);
    std::vector<Transition> tr;
    for (size_t i = 0; i + 1 < ranges.size(); i += 2)
      tr.push_back({static_cast<uint8_t>(ranges[i]), static_cast<uint8_t>(ranges[i + 1]), kInvalidState});
    if (tr.size() == 1) {
      StateID id = b.addByteRange(tr[0].lo, tr[0].hi, kInvalidState);
      return {id, id};
    }
    StateID target = b.addEpsilon(kInvalidState);
    for (auto& t : tr) t.next = target;
    StateID id = b.addSparse(tr);
    return {id, target};
  }

  Frag compileEmptyMatch() { StateID id = b.addEpsilon(kInvalidState); return {id, id}; }
  Frag compileNoMatch() {
    StateID s = b.addEpsilon(kInvalidState);
    StateID e = b.addEpsilon(kInvalidState);
    return {s, e};
  }

  void connect(StateID end, StateID target) {  // the "Patch or insert epsilon" idiom
    if (!b.patch(end, target)) {
      StateID eps = b.addEpsilon(target);
      if (!b.patch(end, eps)) fail("cannot patch state");
    }
  }

  Frag compileConcat(const std::vector<ReP>& subs) {  // compile.go:1225-1256
    if (subs.empty()) return compileEmptyMatch();
    if (subs.size() == 1) return compile(subs[0]);
    Frag f = compile(subs[0]);
    for (size_t i = 1; i < subs.size(); i++) {
      Frag n = compile(subs[i]);
      connect(f.end, n.start);
      f.end = n.end;
    }
    return f;
  }

  StateID buildSplitChain(const std::vector<StateID>& t, size_t from) {  // compile.go:1296-1309
    size_t n = t.size() - from;
    if (n == 1) return t[from];
    if (n == 2) return b.addSplit(t[from], t[from + 1], false);
    StateID right = buildSplitChain(t, from + 1);
    return b.addSplit(t[from], right, false);
  }

  Frag compileAlternate(const std::vector<ReP>& subs) {  // compile.go:1259-1293
    if (subs.empty()) return compileEmptyMatch();
    if (subs.size() == 1) return compile(subs[0]);
    std::vector<StateID> starts, ends;
    for (auto& s : subs) {
      Frag f = compile(s);
      starts.push_back(f.start);
      ends.push_back(f.end);
    }
    StateID split = buildSplitChain(starts, 0);
    StateID join = b.addEpsilon(kInvalidState);
    for (StateID e : ends) b.patch(e, join);
    return {split, join};
  }

  Frag compileStar(const ReP& sub, bool nonGreedy) {  // compile.go:1312-1347
    if (canMatchEmpty(sub)) return compileStarViaPlus(sub, nonGreedy);
    Frag f = compile(sub);
    StateID end = b.addEpsilon(kInvalidState);
    StateID split = nonGreedy ? b.addSplit(end, f.start, true) : b.addSplit(f.start, end, true);
    connect(f.end, split);
    return {split, end};
  }

  Frag compileStarViaPlus(const ReP& sub, bool nonGreedy) {  // compile.go:1351-1385
    Frag f = compile(sub);
    StateID end = b.addEpsilon(kInvalidState);
    StateID plus = nonGreedy ? b.addSplit(end, f.start, true) : b.addSplit(f.start, end, true);
    connect(f.end, plus);
    StateID quest = nonGreedy ? b.addSplit(end, f.start, true) : b.addSplit(f.start, end, true);
    return {quest, end};
  }

  Frag compilePlus(const ReP& sub, bool nonGreedy) {  // compile.go:1433-1456
    Frag f = compile(sub);
    StateID end = b.addEpsilon(kInvalidState);
    StateID split = nonGreedy ? b.addSplit(end, f.start, true) : b.addSplit(f.start, end, true);
    connect(f.end, split);
    return {f.start, end};
  }

  Frag compileQuest(const ReP& sub, bool nonGreedy) {  // compile.go:1459-1481
    Frag f = compile(sub);
    StateID end = b.addEpsilon(kInvalidState);
    StateID split = nonGreedy ? b.addSplit(end, f.start, true) : b.addSplit(f.start, end, true);
    connect(f.end, end);
    return {split, end};
  }

  Frag compileRepeat(const ReP& sub, int mn, int mx, bool nonGreedy) {  // compile.go:1484-1566
    std::vector<ReP> subs;
    if (mx == -1) {
      if (mn == 0) return compileStar(sub, nonGreedy);
      for (int i = 0; i < mn; i++) subs.push_back(sub);
      auto star = std::make_shared<Regexp>();
      star->op = OpStar; star->flags = nonGreedy ? NonGreedy : 0; star->sub = {sub};
      subs.push_back(star);
      return compileConcat(subs);
    }
    if (mn == mx) {
      if (mn == 0) return compileEmptyMatch();
      if (mn == 1) return compile(sub);
      for (int i = 0; i < mn; i++) subs.push_back(sub);
      return compileConcat(subs);
    }
    if (mn > mx) fail("invalid repeat range");
    for (int i = 0; i < mn; i++) subs.push_back(sub);
    for (int i = 0; i < mx - mn; i++) {
      auto q = std::make_shared<Regexp>();
      q->op = OpQuest; q->flags = nonGreedy ? NonGreedy : 0; q->sub = {sub};
      subs.push_back(q);
    }
    return compileConcat(subs);
  }

  Frag compileCapture(const ReP& re) {  // compile.go:1654-1682
    if (re->sub.empty()) return compileEmptyMatch();
    Frag f = compile(re->sub[0]);
    StateID close = b.addCapture(static_cast<uint32_t>(re->cap), false, kInvalidState);
    connect(f.end, close);
    StateID open = b.addCapture(static_cast<uint32_t>(re->cap), true, f.start);
    return {open, close};
  }

  Frag compile(const ReP& re) {  // compile.go:155-233
    if (++depth > 100) fail("pattern too complex");
    struct D { int& d; ~D() { d--; } } guard{depth};
    bool ng = re->flags & NonGreedy;
    switch (re->op) {
      case OpLiteral: return compileLiteral(re);
      case OpCharClass: return compileCharClass(re->rune);
      case OpAnyChar: return compileUTF8Any(true);          // compile.go:977-992
      case OpAnyCharNotNL: return compileUTF8Any(false);    // compile.go:995-1010
      case OpConcat: return compileConcat(re->sub);
      case OpAlternate: return compileAlternate(re->sub);
      case OpStar: return compileStar(re->sub[0], ng);
      case OpPlus: return compilePlus(re->sub[0], ng);
      case OpQuest: return compileQuest(re->sub[0], ng);
      case OpRepeat: return compileRepeat(re->sub[0], re->min, re->max, ng);
      case OpCapture: return compileCapture(re);
      case OpBeginText: { StateID id = b.addLook(LookStartText, kInvalidState); return {id, id}; }
      case OpEndText: { StateID id = b.addLook(LookEndText, kInvalidState); return {id, id}; }
      case OpBeginLine: { StateID id = b.addLook(LookStartLine, kInvalidState); return {id, id}; }
      case OpEndLine: { StateID id = b.addLook(LookEndLine, kInvalidState); return {id, id}; }
      case OpWordBoundary: { StateID id = b.addLook(LookWordBoundary, kInvalidState); return {id, id}; }
      case OpNoWordBoundary: { StateID id = b.addLook(LookNoWordBoundary, kInvalidState); return {id, id}; }
      case OpEmptyMatch: return compileEmptyMatch();
      default: fail("unsupported regex operation");
    }
  }
};

}  // namespace

bool isPatternStartAnchored(const ReP& re) {  // compile.go:1755-1769
  switch (re->op) {
    case OpBeginText: return true;
    case OpConcat: case OpCapture:
      if (!re->sub.empty()) return isPatternStartAnchored(re->sub[0]);
      return false;
    default: return false;
  }
}

bool canMatchEmpty(const ReP& re) {  // compile.go:1388-1430
  switch (re->op) {
    case OpEmptyMatch: return true;
    case OpLiteral: return re->rune.empty();
    case OpCharClass: case OpAnyCharNotNL: case OpAnyChar: return false;
    case OpCapture: return re->sub.empty() ? true : canMatchEmpty(re->sub[0]);
    case OpStar: case OpQuest: return true;
    case OpPlus: return !re->sub.empty() && canMatchEmpty(re->sub[0]);
    case OpRepeat: return re->min == 0 || (!re->sub.empty() && canMatchEmpty(re->sub[0]));
    case OpConcat:
      for (auto& s : re->sub) if (!canMatchEmpty(s)) return false;
      return true;
    case OpAlternate:
      for (auto& s : re->sub) if (canMatchEmpty(s)) return true;
      return false;
    case OpNoMatch: return false;
    case OpBeginLine: case OpEndLine: case OpBeginText: case OpEndText:
    case OpWordBoundary: case OpNoWordBoundary: return true;
  }
  return false;
}

NFA compileNFA(const ReP& re) {  // compile.go:99-151
  Compiler c;
  c.countCaptures(re);
  bool allAnchored = isPatternStartAnchored(re);
  Frag f = c.compile(re);
  StateID matchID = c.b.addMatch();
  c.connect(f.end, matchID);
  StateID anchoredStart = f.start;
  StateID unanchoredStart;
  if (allAnchored) {
    unanchoredStart = anchoredStart;
  } else {  // compileUnanchoredPrefix, compile.go:1633-1650
    StateID anyByte = c.b.addByteRange(0x00, 0xFF, kInvalidState);
    StateID split = c.b.addSplit(f.start, anyByte, false);
    c.b.patch(anyByte, split);
    unanchoredStart = split;
  }
  NFA n;
  n.states = std::move(c.b.states);
  n.startAnchored = anchoredStart;
  n.startUnanchored = unanchoredStart;
  n.anchored = allAnchored;
  n.captureCount = c.captureCount + 1;
  // ByteClassSet.ByteClasses(), alphabet.go:152-165
  uint8_t cls = 0;
  for (int bb = 0; bb < 256; bb++) {
    n.byteClasses[bb] = cls;
    if (c.b.bits[bb / 64] & (uint64_t{1} << (bb % 64))) cls++;
  }
  int mx = 0;
  for (int bb = 0; bb < 256; bb++) if (n.byteClasses[bb] > mx) mx = n.byteClasses[bb];
  n.alphabetLen = mx + 1;
  for (auto& s : n.states)
    if (s.kind == StateLook) {
      n.hasLook = true;
      if (s.look == LookWordBoundary || s.look == LookNoWordBoundary) n.hasWordBoundary = true;
    }
  return n;
}

std::string dumpNFA(const NFA& n) {
  std::ostringstream o;
  o << "states=" << n.states.size() << " startA=" << n.startAnchored << " startU=" << n.startUnanchored
    << " caps=" << n.captureCount << " classes=" << n.alphabetLen << "\n";
  for (size_t i = 0; i < n.states.size(); i++) {
    const NState& s = n.states[i];
    o << i << ": ";
    switch (s.kind) {
      case StateMatch: o << "Match"; break;
      case StateByteRange: o << "ByteRange[" << int(s.lo) << "-" << int(s.hi) << "]->" << s.next; break;
      case StateSparse:
        o << "Sparse";
        for (auto& t : s.trans) o << " [" << int(t.lo) << "-" << int(t.hi) << "]->" << t.next;
        break;
      case StateSplit: o << (s.quantSplit ? "QSplit(" : "Split(") << s.left << "," << s.right << ")"; break;
      case StateEpsilon: o << "Eps->" << static_cast<int64_t>(s.next == kInvalidState ? -1 : int64_t(s.next)); break;
      case StateCapture: o << "Cap" << s.capIndex << (s.capStart ? "(" : ")") << "->" << s.next; break;
      case StateLook: o << "Look" << int(s.look) << "->" << s.next; break;
      default: o << "?"; break;
    }
    o << "\n";
  }
  return o.str();
}

}  // namespace orc
