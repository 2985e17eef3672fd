// ORACLE — TEST INFRASTRUCTURE ONLY (see syntax.hpp header).
// Stack-machine restatement of Go's regexp/syntax.Parse(pattern, syntax.Perl).
#include "syntax.hpp"

#include <algorithm>
#include <cstring>

namespace orc {

namespace {

constexpr int kMaxRune = 0x10FFFF;
constexpr int opLeftParen = 128;
constexpr int opVerticalBar = 129;

ReP mk(int op) {
  auto r = std::make_shared<Regexp>();
  r->op = static_cast<Op>(op);
  return r;
}

bool isCharClassLike(const ReP& re) {
  return (re->op == OpLiteral && re->rune.size() == 1) || re->op == OpCharClass ||
         re->op == OpAnyCharNotNL || re->op == OpAnyChar;
}

// --- class helpers (regexp/syntax: appendRange, cleanClass, negateClass, appendFoldedRange)
void appendRange(std::vector<int>& r, int lo, int hi) {
  // Go tries to merge with the last two ranges; cleanClass canonicalises anyway.
  size_t n = r.size();
  for (size_t i = 2; i <= 4; i += 2) {
    if (n >= i) {
      int rlo = r[n - i], rhi = r[n - i + 1];
      if (lo <= rhi + 1 && rlo <= hi + 1) {
        if (lo < rlo) r[n - i] = lo;
        if (hi > rhi) r[n - i + 1] = hi;
        return;
      }
    }
  }
  r.push_back(lo);
  r.push_back(hi);
}

// unicode.SimpleFold restricted to what the ASCII subset can observe.
int simpleFold(int c) {
  if (c == 'K') return 'k';
  if (c == 'k') return 0x212A;
  if (c == 0x212A) return 'K';
  if (c == 'S') return 's';
  if (c == 's') return 0x17F;
  if (c == 0x17F) return 'S';
  if (c >= 'A' && c <= 'Z') return c + 32;
  if (c >= 'a' && c <= 'z') return c - 32;
  return c;
}

int minFoldRune(int r) {
  int m = r, r0 = r;
  for (r = simpleFold(r); r != r0; r = simpleFold(r)) m = std::min(m, r);
  return m;
}

void appendFoldedRange(std::vector<int>& r, int lo, int hi) {
  // (a rune past U+007F written under (?i) needs the Unicode fold tables, which are not restated: `(?i)[é]` is {É, é})
  if (hi >= 0x80) throw ParseError{"unsupported: Unicode case folding (a rune past U+007F under (?i), outside restated subset)"};
  appendRange(r, lo, hi);
  // Only ASCII letters (and the two non-ASCII members of the K / S orbits) fold
  // inside the restated subset; other runes > 0x7F make the class non-ASCII and
  // the NFA compiler rejects it as out of scope anyway.
  for (int c = std::max(lo, 0); c <= std::min(hi, 0x7F); c++) {
    for (int f = simpleFold(c); f != c; f = simpleFold(f)) appendRange(r, f, f);
  }
  if (lo <= 0x17F && 0x17F <= hi) { appendRange(r, 'S', 'S'); appendRange(r, 's', 's'); }
  if (lo <= 0x212A && 0x212A <= hi) { appendRange(r, 'K', 'K'); appendRange(r, 'k', 'k'); }
}

void cleanClass(std::vector<int>& r) {
  size_t n = r.size() / 2;
  std::vector<std::pair<int, int>> v(n);
  for (size_t i = 0; i < n; i++) v[i] = {r[2 * i], r[2 * i + 1]};
  std::sort(v.begin(), v.end(), [](auto& a, auto& b) {
    return a.first < b.first || (a.first == b.first && a.second > b.second);
  });
  std::vector<int> out;
  for (auto& p : v) {
    if (!out.empty() && p.first <= out.back() + 1) {
      if (p.second > out.back()) out.back() = p.second;
      continue;
    }
    out.push_back(p.first);
    out.push_back(p.second);
  }
  r.swap(out);
}

void negateClass(std::vector<int>& r) {
  std::vector<int> out;
  int next = 0;
  for (size_t i = 0; i + 1 < r.size(); i += 2) {
    int lo = r[i], hi = r[i + 1];
    if (next <= lo - 1) { out.push_back(next); out.push_back(lo - 1); }
    next = hi + 1;
  }
  if (next <= kMaxRune) { out.push_back(next); out.push_back(kMaxRune); }
  r.swap(out);
}

void appendClass(std::vector<int>& r, const std::vector<int>& x) {
  for (size_t i = 0; i + 1 < x.size(); i += 2) appendRange(r, x[i], x[i + 1]);
}
void appendFoldedClass(std::vector<int>& r, const std::vector<int>& x) {
  for (size_t i = 0; i + 1 < x.size(); i += 2) appendFoldedRange(r, x[i], x[i + 1]);
}
void appendNegatedClass(std::vector<int>& r, const std::vector<int>& x) {
  int next = 0;
  for (size_t i = 0; i + 1 < x.size(); i += 2) {
    int lo = x[i], hi = x[i + 1];
    if (next <= lo - 1) appendRange(r, next, lo - 1);
    next = hi + 1;
  }
  if (next <= kMaxRune) appendRange(r, next, kMaxRune);
}

void appendLiteral(std::vector<int>& r, int x, int flags) {
  if (flags & FoldCase) appendFoldedRange(r, x, x); else appendRange(r, x, x);
}

bool matchRune(const ReP& re, int r) {
  switch (re->op) {
    case OpLiteral: return re->rune.size() == 1 && re->rune[0] == r;
    case OpCharClass:
      for (size_t i = 0; i + 1 < re->rune.size(); i += 2)
        if (re->rune[i] <= r && r <= re->rune[i + 1]) return true;
      return false;
    case OpAnyCharNotNL: return r != '\n';
    case OpAnyChar: return true;
    default: return false;
  }
}

void mergeCharClass(const ReP& dst, const ReP& src) {
  switch (dst->op) {
    case OpAnyChar: break;
    case OpAnyCharNotNL:
      if (matchRune(src, '\n')) dst->op = OpAnyChar;
      break;
    case OpCharClass:
      if (src->op == OpLiteral) appendLiteral(dst->rune, src->rune[0], src->flags);
      else appendClass(dst->rune, src->rune);
      break;
    case OpLiteral: {
      if (src->rune[0] == dst->rune[0] && src->flags == dst->flags) break;
      dst->op = OpCharClass;
      int d = dst->rune[0];
      dst->rune.clear();
      appendLiteral(dst->rune, d, dst->flags);
      appendLiteral(dst->rune, src->rune[0], src->flags);
      break;
    }
    default: break;
  }
}

void cleanAlt(const ReP& re) {
  if (re->op != OpCharClass) return;
  cleanClass(re->rune);
  if (re->rune.size() == 2 && re->rune[0] == 0 && re->rune[1] == kMaxRune) {
    re->rune.clear(); re->op = OpAnyChar; return;
  }
  if (re->rune.size() == 4 && re->rune[0] == 0 && re->rune[1] == '\n' - 1 &&
      re->rune[2] == '\n' + 1 && re->rune[3] == kMaxRune) {
    re->rune.clear(); re->op = OpAnyCharNotNL; return;
  }
}

struct PerlGroup { int sign; std::vector<int> cls; };

const PerlGroup* perlGroup(char c) {
  static const PerlGroup d{+1, {'0', '9'}}, D{-1, {'0', '9'}};
  static const PerlGroup s{+1, {0x9, 0xa, 0xc, 0xd, 0x20, 0x20}}, S{-1, {0x9, 0xa, 0xc, 0xd, 0x20, 0x20}};
  static const PerlGroup w{+1, {'0', '9', 'A', 'Z', '_', '_', 'a', 'z'}}, W{-1, {'0', '9', 'A', 'Z', '_', '_', 'a', 'z'}};
  switch (c) {
    case 'd': return &d; case 'D': return &D; case 's': return &s;
    case 'S': return &S; case 'w': return &w; case 'W': return &W;
  }
  return nullptr;
}

const std::vector<int>* posixGroup(const std::string& name) {
  static const std::pair<const char*, std::vector<int>> tbl[] = {
      {"alnum", {'0', '9', 'A', 'Z', 'a', 'z'}}, {"alpha", {'A', 'Z', 'a', 'z'}},
      {"ascii", {0, 0x7F}}, {"blank", {'\t', '\t', ' ', ' '}},
      {"cntrl", {0, 0x1F, 0x7F, 0x7F}}, {"digit", {'0', '9'}}, {"graph", {'!', '~'}},
      {"lower", {'a', 'z'}}, {"print", {' ', '~'}},
      {"punct", {'!', '/', ':', '@', '[', '`', '{', '~'}},
      {"space", {'\t', '\r', ' ', ' '}}, {"upper", {'A', 'Z'}},
      {"word", {'0', '9', 'A', 'Z', '_', '_', 'a', 'z'}},
      {"xdigit", {'0', '9', 'A', 'F', 'a', 'f'}}};
  for (auto& e : tbl) if (name == e.first) return &e.second;
  return nullptr;
}

struct Parser {
  int flags = Perl;
  std::vector<ReP> stack;
  int numCap = 0;
  std::string whole;

  [[noreturn]] void fail(const std::string& m) { throw ParseError{m + ": `" + whole + "`"}; }

  // nextRune: decode one UTF-8 rune.
  int nextRune(const std::string& s, size_t& i) {
    unsigned char c = s[i];
    if (c < 0x80) { i++; return c; }
    int n = (c >= 0xF0) ? 4 : (c >= 0xE0) ? 3 : (c >= 0xC0) ? 2 : 0;
    if (n == 0 || i + n > s.size()) fail("invalid UTF-8");
    int r = c & (0xFF >> (n + 1));
    for (int k = 1; k < n; k++) {
      unsigned char cc = s[i + k];
      if ((cc & 0xC0) != 0x80) fail("invalid UTF-8");
      r = (r << 6) | (cc & 0x3F);
    }
    i += n;
    return r;
  }

  bool maybeConcat(int r, int fl) {
    size_t n = stack.size();
    if (n < 2) return false;
    ReP re1 = stack[n - 1], re2 = stack[n - 2];
    if (re1->op != OpLiteral || re2->op != OpLiteral ||
        (re1->flags & FoldCase) != (re2->flags & FoldCase))
      return false;
    re2->rune.insert(re2->rune.end(), re1->rune.begin(), re1->rune.end());
    if (r >= 0) {
      re1->rune.assign(1, r);
      re1->flags = fl;
      return true;
    }
    stack.pop_back();
    return false;
  }

  ReP push(ReP re) {
    if (re->op == OpCharClass && re->rune.size() == 2 && re->rune[0] == re->rune[1]) {
      if (maybeConcat(re->rune[0], flags & ~FoldCase)) return nullptr;
      re->op = OpLiteral;
      re->rune.resize(1);
      re->flags = flags & ~FoldCase;
    } else if (re->op == OpCharClass && re->rune.size() == 4 && re->rune[0] == re->rune[1] &&
               re->rune[2] == re->rune[3] && simpleFold(re->rune[0]) == re->rune[2] &&
               simpleFold(re->rune[2]) == re->rune[0]) {
      if (maybeConcat(re->rune[0], flags | FoldCase)) return nullptr;
      re->op = OpLiteral;
      re->rune.resize(1);
      re->flags = flags | FoldCase;
    } else {
      maybeConcat(-1, 0);
    }
    stack.push_back(re);
    return re;
  }

  ReP op(int o) {
    ReP re = mk(o);
    re->flags = flags;
    return push(re);
  }

  void literal(int r) {
    ReP re = mk(OpLiteral);
    re->flags = flags;
    if ((flags & FoldCase) && r >= 0x80) throw ParseError{"unsupported: Unicode case folding (a rune past U+007F under (?i), outside restated subset)"};
    if (flags & FoldCase) r = minFoldRune(r);
    re->rune.assign(1, r);
    push(re);
  }

  static bool repeatIsValid(const ReP& re, int n) {
    if (re->op == OpRepeat) {
      int m = re->max;
      if (m == 0) return true;
      if (m < 0) m = re->min;
      if (m > n) return false;
      if (m > 0) n /= m;
    }
    for (auto& s : re->sub) if (!repeatIsValid(s, n)) return false;
    return true;
  }

  void repeat(int o, int mn, int mx, const std::string& t, size_t& i, bool lastRepeat) {
    int fl = flags;
    if (i < t.size() && t[i] == '?') { i++; fl ^= NonGreedy; }
    if (lastRepeat) fail("invalid nested repetition operator");
    if (stack.empty() || stack.back()->op >= opLeftParen) fail("missing argument to repetition operator");
    ReP sub = stack.back();
    ReP re = mk(o);
    re->min = mn; re->max = mx; re->flags = fl;
    re->sub.push_back(sub);
    stack.back() = re;
    if (o == OpRepeat && (mn >= 2 || mx >= 2) && !repeatIsValid(re, 1000)) fail("invalid repeat count");
  }

  // --- alternation factoring (regexp/syntax parser.factor) --------------------
  static bool leadingString(const ReP& re0, std::vector<int>*& str, int& fl) {
    ReP re = re0;
    if (re->op == OpConcat && !re->sub.empty()) re = re->sub[0];
    if (re->op != OpLiteral) { str = nullptr; fl = 0; return false; }
    str = &re->rune; fl = re->flags & FoldCase;
    return true;
  }

  ReP removeLeadingString(ReP re, size_t n) {
    if (re->op == OpConcat && !re->sub.empty()) {
      ReP sub = removeLeadingString(re->sub[0], n);
      re->sub[0] = sub;
      if (sub->op == OpEmptyMatch) {
        switch (re->sub.size()) {
          case 0: case 1: re->op = OpEmptyMatch; re->sub.clear(); break;
          case 2: re = re->sub[1]; break;
          default: re->sub.erase(re->sub.begin()); break;
        }
      }
      return re;
    }
    if (re->op == OpLiteral) {
      re->rune.erase(re->rune.begin(), re->rune.begin() + n);
      if (re->rune.empty()) re->op = OpEmptyMatch;
    }
    return re;
  }

  static ReP leadingRegexp(const ReP& re) {
    if (re->op == OpEmptyMatch) return nullptr;
    if (re->op == OpConcat && !re->sub.empty()) {
      if (re->sub[0]->op == OpEmptyMatch) return nullptr;
      return re->sub[0];
    }
    return re;
  }

  ReP removeLeadingRegexp(ReP re) {
    if (re->op == OpConcat && !re->sub.empty()) {
      re->sub.erase(re->sub.begin());
      if (re->sub.empty()) { re->op = OpEmptyMatch; }
      else if (re->sub.size() == 1) re = re->sub[0];
      return re;
    }
    return mk(OpEmptyMatch);
  }

  std::vector<ReP> factor(std::vector<ReP> sub) {
    if (sub.size() < 2) return sub;
    // Round 1: common literal prefixes.
    {
      std::vector<ReP> out;
      std::vector<int> str; int strflags = 0; size_t start = 0;
      for (size_t i = 0; i <= sub.size(); i++) {
        std::vector<int>* istr = nullptr; int iflags = 0;
        std::vector<int> istrCopy;
        if (i < sub.size()) {
          leadingString(sub[i], istr, iflags);
          if (istr) istrCopy = *istr;
          if (iflags == strflags) {
            size_t same = 0;
            while (same < str.size() && same < istrCopy.size() && str[same] == istrCopy[same]) same++;
            if (same > 0) { str.resize(same); continue; }
          }
        }
        if (i == start) {
        } else if (i == start + 1) {
          out.push_back(sub[start]);
        } else {
          ReP prefix = mk(OpLiteral);
          prefix->flags = strflags;
          prefix->rune = str;
          std::vector<ReP> rest;
          for (size_t j = start; j < i; j++) rest.push_back(removeLeadingString(sub[j], str.size()));
          ReP suffix = collapse(rest, OpAlternate);
          ReP re = mk(OpConcat);
          re->sub = {prefix, suffix};
          out.push_back(re);
        }
        start = i;
        str = istrCopy;
        strflags = iflags;
      }
      sub.swap(out);
    }
    // Round 2: common leading char-class-like piece.
    {
      std::vector<ReP> out;
      ReP first; size_t start = 0;
      for (size_t i = 0; i <= sub.size(); i++) {
        ReP ifirst;
        if (i < sub.size()) {
          ifirst = leadingRegexp(sub[i]);
          if (first && ifirst && first->equal(*ifirst) &&
              (isCharClassLike(first) ||
               (first->op == OpRepeat && first->min == first->max && isCharClassLike(first->sub[0]))))
            continue;
        }
        if (i == start) {
        } else if (i == start + 1) {
          out.push_back(sub[start]);
        } else {
          ReP prefix = first;
          std::vector<ReP> rest;
          for (size_t j = start; j < i; j++) rest.push_back(removeLeadingRegexp(sub[j]));
          ReP suffix = collapse(rest, OpAlternate);
          ReP re = mk(OpConcat);
          re->sub = {prefix, suffix};
          out.push_back(re);
        }
        start = i;
        first = ifirst;
      }
      sub.swap(out);
    }
    // Round 3: runs of single literals / classes -> one class.
    {
      std::vector<ReP> out;
      size_t start = 0;
      for (size_t i = 0; i <= sub.size(); i++) {
        if (i < sub.size() && isCharClassLike(sub[i])) continue;
        if (i == start) {
        } else if (i == start + 1) {
          out.push_back(sub[start]);
        } else {
          size_t mx = start;
          for (size_t j = start + 1; j < i; j++)
            if (sub[mx]->op < sub[j]->op ||
                (sub[mx]->op == sub[j]->op && sub[mx]->rune.size() < sub[j]->rune.size()))
              mx = j;
          std::swap(sub[start], sub[mx]);
          for (size_t j = start + 1; j < i; j++) mergeCharClass(sub[start], sub[j]);
          cleanAlt(sub[start]);
          out.push_back(sub[start]);
        }
        if (i < sub.size()) out.push_back(sub[i]);
        start = i + 1;
      }
      sub.swap(out);
    }
    // Round 4: runs of empty matches -> one.
    {
      std::vector<ReP> out;
      for (size_t i = 0; i < sub.size(); i++) {
        if (i + 1 < sub.size() && sub[i]->op == OpEmptyMatch && sub[i + 1]->op == OpEmptyMatch) continue;
        out.push_back(sub[i]);
      }
      sub.swap(out);
    }
    return sub;
  }

  ReP collapse(const std::vector<ReP>& subs, int o) {
    if (subs.size() == 1) return subs[0];
    ReP re = mk(o);
    for (auto& s : subs) {
      if (s->op == o) re->sub.insert(re->sub.end(), s->sub.begin(), s->sub.end());
      else re->sub.push_back(s);
    }
    if (o == OpAlternate) {
      re->sub = factor(re->sub);
      if (re->sub.size() == 1) re = re->sub[0];
    }
    return re;
  }

  ReP concat() {
    maybeConcat(-1, 0);
    size_t i = stack.size();
    while (i > 0 && stack[i - 1]->op < opLeftParen) i--;
    std::vector<ReP> subs(stack.begin() + i, stack.end());
    stack.resize(i);
    if (subs.empty()) return push(mk(OpEmptyMatch));
    return push(collapse(subs, OpConcat));
  }

  ReP alternate() {
    size_t i = stack.size();
    while (i > 0 && stack[i - 1]->op < opLeftParen) i--;
    std::vector<ReP> subs(stack.begin() + i, stack.end());
    stack.resize(i);
    if (!subs.empty()) cleanAlt(subs.back());
    if (subs.empty()) return push(mk(OpNoMatch));
    return push(collapse(subs, OpAlternate));
  }

  bool swapVerticalBar() {
    size_t n = stack.size();
    if (n >= 3 && stack[n - 2]->op == opVerticalBar && isCharClassLike(stack[n - 1]) &&
        isCharClassLike(stack[n - 3])) {
      ReP re1 = stack[n - 1], re3 = stack[n - 3];
      if (re1->op > re3->op) { std::swap(re1, re3); stack[n - 3] = re3; }
      mergeCharClass(re3, re1);
      stack.pop_back();
      return true;
    }
    if (n >= 2) {
      ReP re1 = stack[n - 1], re2 = stack[n - 2];
      if (re2->op == opVerticalBar) {
        if (n >= 3) cleanAlt(stack[n - 3]);
        stack[n - 2] = re1;
        stack[n - 1] = re2;
        return true;
      }
    }
    return false;
  }

  void parseVerticalBar() {
    concat();
    if (!swapVerticalBar()) op(opVerticalBar);
  }

  void parseRightParen() {
    concat();
    if (swapVerticalBar()) stack.pop_back();
    alternate();
    size_t n = stack.size();
    if (n < 2) fail("unexpected )");
    ReP re1 = stack[n - 1], re2 = stack[n - 2];
    stack.resize(n - 2);
    if (re2->op != opLeftParen) fail("unexpected )");
    flags = re2->flags;
    if (re2->cap == 0) {
      push(re1);
    } else {
      re2->op = OpCapture;
      re2->sub = {re1};
      push(re2);
    }
  }

  static bool isValidCaptureName(const std::string& n) {
    if (n.empty()) return false;
    for (char c : n)
      if (c != '_' && !(c >= '0' && c <= '9') && !(c >= 'a' && c <= 'z') && !(c >= 'A' && c <= 'Z'))
        return false;
    return true;
  }

  void parsePerlFlags(const std::string& t, size_t& i) {
    // t[i..] starts with "(?"
    bool startsP = t.compare(i, 4, "(?P<") == 0;
    bool startsLt = t.compare(i, 3, "(?<") == 0;
    if (startsP || startsLt) {
      size_t b = i + (startsP ? 4 : 3);
      size_t e = t.find('>', b);
      if (e == std::string::npos) fail("invalid named capture");
      std::string name = t.substr(b, e - b);
      if (!isValidCaptureName(name)) fail("invalid named capture");
      numCap++;
      ReP re = op(opLeftParen);
      re->cap = numCap;
      re->name = name;
      i = e + 1;
      return;
    }
    size_t j = i + 2;
    int fl = flags;
    int sign = +1;
    bool sawFlag = false;
    while (j < t.size()) {
      int c = nextRune(t, j);
      switch (c) {
        case 'i': fl |= FoldCase; sawFlag = true; break;
        case 'm': fl &= ~OneLine; sawFlag = true; break;
        case 's': fl |= DotNL; sawFlag = true; break;
        case 'U': fl |= NonGreedy; sawFlag = true; break;
        case '-':
          if (sign < 0) fail("invalid or unsupported Perl syntax");
          sign = -1; fl = ~fl; sawFlag = false;
          break;
        case ':': case ')':
          if (sign < 0) {
            if (!sawFlag) fail("invalid or unsupported Perl syntax");
            fl = ~fl;
          }
          if (c == ':') op(opLeftParen);
          flags = fl;
          i = j;
          return;
        default: fail("invalid or unsupported Perl syntax");
      }
    }
    fail("missing closing )");
  }

  static int unhex(int c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
  }

  int parseEscape(const std::string& t, size_t& i) {
    size_t j = i + 1;
    if (j >= t.size()) fail("trailing backslash at end of expression");
    int c = nextRune(t, j);
    switch (c) {
      case '1': case '2': case '3': case '4': case '5': case '6': case '7':
        if (j >= t.size() || t[j] < '0' || t[j] > '7') break;
        [[fallthrough]];
      case '0': {
        int r = c - '0';
        for (int k = 1; k < 3; k++) {
          if (j >= t.size() || t[j] < '0' || t[j] > '7') break;
          r = r * 8 + (t[j] - '0');
          j++;
        }
        i = j; return r;
      }
      case 'x': {
        if (j >= t.size()) break;
        int c2 = nextRune(t, j);
        if (c2 == '{') {
          int nhex = 0, r = 0; bool ok = false;
          while (j < t.size()) {
            int c3 = nextRune(t, j);
            if (c3 == '}') { ok = true; break; }
            int v = unhex(c3);
            if (v < 0) break;
            r = r * 16 + v;
            if (r > kMaxRune) break;
            nhex++;
          }
          if (!ok || nhex == 0) break;
          i = j; return r;
        }
        int x = unhex(c2);
        if (j >= t.size()) break;
        int c3 = nextRune(t, j);
        int y = unhex(c3);
        if (x < 0 || y < 0) break;
        i = j; return x * 16 + y;
      }
      case 'a': i = j; return 7;
      case 'f': i = j; return '\f';
      case 'n': i = j; return '\n';
      case 'r': i = j; return '\r';
      case 't': i = j; return '\t';
      case 'v': i = j; return '\v';
      default:
        if (c < 0x80 && !((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'))) {
          i = j; return c;
        }
    }
    fail("invalid escape sequence");
  }

  bool parsePerlClassEscape(const std::string& t, size_t& i, std::vector<int>& cls) {
    if (i + 1 >= t.size() || t[i] != '\\') return false;
    const PerlGroup* g = perlGroup(t[i + 1]);
    if (!g) return false;
    i += 2;
    appendGroup(cls, g->sign, g->cls);
    return true;
  }

  void appendGroup(std::vector<int>& cls, int sign, const std::vector<int>& g) {
    if (!(flags & FoldCase)) {
      if (sign > 0) appendClass(cls, g); else appendNegatedClass(cls, g);
    } else {
      std::vector<int> tmp;
      appendFoldedClass(tmp, g);
      cleanClass(tmp);
      if (sign > 0) appendClass(cls, tmp); else appendNegatedClass(cls, tmp);
    }
  }

  bool parseNamedClass(const std::string& t, size_t& i, std::vector<int>& cls) {
    if (t.size() < i + 2 || t[i] != '[' || t[i + 1] != ':') return false;
    size_t e = t.find(":]", i + 2);
    if (e == std::string::npos) return false;
    std::string name = t.substr(i + 2, e - (i + 2));
    int sign = +1;
    if (!name.empty() && name[0] == '^') { sign = -1; name = name.substr(1); }
    const std::vector<int>* g = posixGroup(name);
    if (!g) fail("invalid character class range");
    appendGroup(cls, sign, *g);
    i = e + 2;
    return true;
  }

  void parseClass(const std::string& t, size_t& i) {
    size_t j = i + 1;
    ReP re = mk(OpCharClass);
    re->flags = flags;
    int sign = +1;
    if (j < t.size() && t[j] == '^') {
      sign = -1; j++;
      if (!(flags & ClassNL)) { re->rune.push_back('\n'); re->rune.push_back('\n'); }
    }
    std::vector<int>& cls = re->rune;
    bool first = true;
    while (j >= t.size() || t[j] != ']' || first) {
      if (j >= t.size()) fail("missing closing ]");
      first = false;
      if (t.size() - j > 2 && t[j] == '[' && t[j + 1] == ':') {
        if (parseNamedClass(t, j, cls)) continue;
      }
      if (j + 1 < t.size() && t[j] == '\\' && (t[j + 1] == 'p' || t[j + 1] == 'P'))
        fail("unsupported: Unicode class \\p (outside restated subset)");
      if (parsePerlClassEscape(t, j, cls)) continue;
      int lo = (t[j] == '\\') ? parseEscape(t, j) : nextRune(t, j);
      int hi = lo;
      if (j + 1 < t.size() && t[j] == '-' && t[j + 1] != ']') {
        j++;
        if (j >= t.size()) fail("missing closing ]");
        hi = (t[j] == '\\') ? parseEscape(t, j) : nextRune(t, j);
        if (hi < lo) fail("invalid character class range");
      }
      if (!(flags & FoldCase)) appendRange(cls, lo, hi); else appendFoldedRange(cls, lo, hi);
    }
    j++;  // ]
    cleanClass(cls);
    if (sign < 0) negateClass(cls);
    push(re);
    i = j;
  }

  static int parseInt(const std::string& t, size_t& i) {
    if (i >= t.size() || t[i] < '0' || t[i] > '9') return -2;
    if (i + 1 < t.size() && t[i] == '0' && t[i + 1] >= '0' && t[i + 1] <= '9') return -2;
    size_t b = i;
    while (i < t.size() && t[i] >= '0' && t[i] <= '9') i++;
    if (i - b >= 8) return -1;
    int n = 0;
    for (size_t k = b; k < i; k++) n = n * 10 + (t[k] - '0');
    return n;
  }

  bool parseRepeat(const std::string& t, size_t i, int& mn, int& mx, size_t& after) {
    if (i >= t.size() || t[i] != '{') return false;
    i++;
    mn = parseInt(t, i);
    if (mn == -2) return false;
    if (i >= t.size()) return false;
    if (t[i] != ',') {
      mx = mn;
    } else {
      i++;
      if (i >= t.size()) return false;
      if (t[i] == '}') mx = -1;
      else {
        mx = parseInt(t, i);
        if (mx == -2) return false;
        if (mx == -1) mn = -1;  // too big
      }
    }
    if (i >= t.size() || t[i] != '}') return false;
    after = i + 1;
    return true;
  }

  ReP run(const std::string& s) {
    whole = s;
    const std::string& t = s;
    size_t i = 0;
    bool lastRepeat = false;
    while (i < t.size()) {
      bool isRepeat = false;
      switch (t[i]) {
        default: { int c = nextRune(t, i); literal(c); break; }
        case '(':
          if (i + 1 < t.size() && t[i + 1] == '?') { parsePerlFlags(t, i); break; }
          numCap++;
          op(opLeftParen)->cap = numCap;
          i++;
          break;
        case '|': parseVerticalBar(); i++; break;
        case ')': parseRightParen(); i++; break;
        case '^': op((flags & OneLine) ? OpBeginText : OpBeginLine); i++; break;
        case '$':
          if (flags & OneLine) op(OpEndText)->flags |= WasDollar; else op(OpEndLine);
          i++;
          break;
        case '.': op((flags & DotNL) ? OpAnyChar : OpAnyCharNotNL); i++; break;
        case '[': parseClass(t, i); break;
        case '*': case '+': case '?': {
          int o = t[i] == '*' ? OpStar : t[i] == '+' ? OpPlus : OpQuest;
          i++;
          repeat(o, 0, 0, t, i, lastRepeat);
          isRepeat = true;
          break;
        }
        case '{': {
          int mn, mx; size_t after;
          if (!parseRepeat(t, i, mn, mx, after)) { literal('{'); i++; break; }
          if (mn < 0 || mn > 1000 || mx > 1000 || (mx >= 0 && mn > mx)) fail("invalid repeat count");
          i = after;
          repeat(OpRepeat, mn, mx, t, i, lastRepeat);
          isRepeat = true;
          break;
        }
        case '\\': {
          if (i + 1 < t.size()) {
            char c = t[i + 1];
            if (c == 'A') { op(OpBeginText); i += 2; break; }
            if (c == 'b') { op(OpWordBoundary); i += 2; break; }
            if (c == 'B') { op(OpNoWordBoundary); i += 2; break; }
            if (c == 'C') fail("invalid escape sequence");
            if (c == 'z') { op(OpEndText); i += 2; break; }
            if (c == 'Q') {
              size_t e = t.find("\\E", i + 2);
              std::string lit = t.substr(i + 2, e == std::string::npos ? std::string::npos : e - (i + 2));
              size_t k = 0;
              while (k < lit.size()) { int r = nextRune(lit, k); literal(r); }
              i = (e == std::string::npos) ? t.size() : e + 2;
              break;
            }
            if (c == 'p' || c == 'P') fail("unsupported: Unicode class \\p (outside restated subset)");
          }
          ReP re = mk(OpCharClass);
          re->flags = flags;
          if (parsePerlClassEscape(t, i, re->rune)) { push(re); break; }
          int r = parseEscape(t, i);
          literal(r);
          break;
        }
      }
      lastRepeat = isRepeat;
    }
    concat();
    if (swapVerticalBar()) stack.pop_back();
    alternate();
    if (stack.size() != 1) fail("missing closing )");
    return stack[0];
  }
};

}  // namespace

bool Regexp::equal(const Regexp& y) const {
  const Regexp& x = *this;
  if (x.op != y.op) return false;
  switch (x.op) {
    case OpEndText:
      if ((x.flags & WasDollar) != (y.flags & WasDollar)) return false;
      break;
    case OpLiteral: case OpCharClass:
      if (x.rune != y.rune) return false;
      if (x.op == OpLiteral && (x.flags & FoldCase) != (y.flags & FoldCase)) return false;
      break;
    case OpAlternate: case OpConcat:
      if (x.sub.size() != y.sub.size()) return false;
      for (size_t i = 0; i < x.sub.size(); i++) if (!x.sub[i]->equal(*y.sub[i])) return false;
      break;
    case OpStar: case OpPlus: case OpQuest:
      if ((x.flags & NonGreedy) != (y.flags & NonGreedy) || !x.sub[0]->equal(*y.sub[0])) return false;
      break;
    case OpRepeat:
      if ((x.flags & NonGreedy) != (y.flags & NonGreedy) || x.min != y.min || x.max != y.max ||
          !x.sub[0]->equal(*y.sub[0]))
        return false;
      break;
    case OpCapture:
      if (x.cap != y.cap || x.name != y.name || !x.sub[0]->equal(*y.sub[0])) return false;
      break;
    default: break;
  }
  return true;
}

ReP parse(const std::string& pattern) {
  Parser p;
  return p.run(pattern);
}

std::string dump(const ReP& re) {
  static const char* names[] = {"?", "nomatch", "empty", "lit", "cc", "dnl", "dot", "bol", "eol",
                                "bot", "eot", "wb", "nwb", "cap", "star", "plus", "quest", "rep",
                                "cat", "alt"};
  std::string s = names[re->op];
  if (re->flags & NonGreedy && (re->op >= OpStar && re->op <= OpRepeat)) s += "?";
  if (re->op == OpLiteral) {
    s += (re->flags & FoldCase) ? "i{" : "{";
    for (int r : re->rune) {
      if (r >= 0x20 && r < 0x7F) s += static_cast<char>(r);
      else s += "\\x{" + std::to_string(r) + "}";
    }
    s += "}";
  } else if (re->op == OpCharClass) {
    s += "[";
    for (size_t i = 0; i + 1 < re->rune.size(); i += 2) {
      if (i) s += " ";
      s += std::to_string(re->rune[i]) + "-" + std::to_string(re->rune[i + 1]);
    }
    s += "]";
  } else if (re->op == OpRepeat) {
    s += "{" + std::to_string(re->min) + "," + std::to_string(re->max) + "}";
  } else if (re->op == OpCapture) {
    s += std::to_string(re->cap);
  }
  if (!re->sub.empty()) {
    s += "(";
    for (size_t i = 0; i < re->sub.size(); i++) { if (i) s += ","; s += dump(re->sub[i]); }
    s += ")";
  }
  return s;
}

}  // namespace orc
