// ORACLE — TEST INFRASTRUCTURE ONLY (see nfa.hpp header for the reference map).
#include "nfa.hpp"

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

  Frag compileCharClass(const std::vector<int>& ranges) {  // compile.go:384-437
    if (ranges.empty()) return compileNoMatch();
    for (int r : ranges)
      if (r > 127) fail("unsupported: non-ASCII character class (UTF-8 automata are out of scope, SURVEY 2.1)");
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
      case OpAnyChar: case OpAnyCharNotNL:
        fail("unsupported: '.' (UTF-8 rune states are out of scope, SURVEY 2.1)");
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
