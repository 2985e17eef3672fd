// See frontend.h.  Recursive-descent parser + normaliser reproducing the AST shape of Go's
// regexp/syntax (Perl flags), Thompson construction in the reference's creation order, and the
// strategy gate for the accelerated subset.
#include "frontend.h"

#include <algorithm>
#include <cstring>
#include <functional>
#include <set>

namespace cxg {

namespace {

constexpr int32_t kMaxRune = 0x10FFFF;

[[noreturn]] void syntaxErr(const std::string& m) { throw FrontendError{CXG_E_SYNTAX, m}; }
[[noreturn]] void unsupported(const std::string& m) { throw FrontendError{CXG_E_UNSUPPORTED, m}; }

// ------------------------------------------------------------------ rune ranges
struct Ranges {
  std::vector<int32_t> v;  // lo,hi pairs
  void add(int32_t lo, int32_t hi) { v.push_back(lo); v.push_back(hi); }
  void canon() {
    std::vector<std::pair<int32_t, int32_t>> p;
    for (size_t i = 0; i + 1 < v.size(); i += 2) p.emplace_back(v[i], v[i + 1]);
    std::sort(p.begin(), p.end());
    v.clear();
    for (auto& q : p) {
      if (!v.empty() && q.first <= v.back() + 1) { v.back() = std::max(v.back(), q.second); continue; }
      v.push_back(q.first); v.push_back(q.second);
    }
  }
  void negate() {
    std::vector<int32_t> o;
    int32_t nx = 0;
    for (size_t i = 0; i + 1 < v.size(); i += 2) {
      if (nx <= v[i] - 1) { o.push_back(nx); o.push_back(v[i] - 1); }
      nx = v[i + 1] + 1;
    }
    if (nx <= kMaxRune) { o.push_back(nx); o.push_back(kMaxRune); }
    v.swap(o);
  }
};

int32_t foldNext(int32_t c) {  // unicode.SimpleFold orbit, ASCII letters + the K/S special members
  switch (c) {
    case 'K': return 'k'; case 'k': return 0x212A; case 0x212A: return 'K';
    case 'S': return 's'; case 's': return 0x17F; case 0x17F: return 'S';
  }
  if (c >= 'A' && c <= 'Z') return c + 32;
  if (c >= 'a' && c <= 'z') return c - 32;
  return c;
}

void addFolded(Ranges& r, int32_t lo, int32_t hi) {
  // unicode.SimpleFold is restated for the ASCII letters (with the two runes their orbits reach, U+017F and U+212A) only: a rune
  // past U+007F written under (?i) would need the Unicode fold tables (`(?i)[é]` is {É, é} in the reference)
  if (hi >= 0x80) unsupported("Unicode case folding (a rune past U+007F under (?i))");
  r.add(lo, hi);
  for (int32_t c = std::max<int32_t>(lo, 'A'); c <= std::min<int32_t>(hi, 'z'); c++)
    for (int32_t f = foldNext(c); f != c; f = foldNext(f)) r.add(f, f);
  for (int32_t special : {0x17F, 0x212A})
    if (lo <= special && special <= hi)
      for (int32_t f = foldNext(special); f != special; f = foldNext(f)) r.add(f, f);
}

struct NamedClass { const char* name; std::vector<int32_t> r; };
const std::vector<int32_t>& perlClass(char c) {
  static const std::vector<int32_t> d{'0', '9'}, s{9, 10, 12, 13, 32, 32}, w{'0', '9', 'A', 'Z', '_', '_', 'a', 'z'}, none;
  switch (c | 0x20) { case 'd': return d; case 's': return s; case 'w': return w; }
  return none;
}
const std::vector<int32_t>* posixClass(const std::string& n) {
  static const NamedClass t[] = {
      {"alnum", {'0', '9', 'A', 'Z', 'a', 'z'}}, {"alpha", {'A', 'Z', 'a', 'z'}}, {"ascii", {0, 127}},
      {"blank", {9, 9, 32, 32}}, {"cntrl", {0, 31, 127, 127}}, {"digit", {'0', '9'}}, {"graph", {'!', '~'}},
      {"lower", {'a', 'z'}}, {"print", {' ', '~'}}, {"punct", {'!', '/', ':', '@', '[', '`', '{', '~'}},
      {"space", {9, 13, 32, 32}}, {"upper", {'A', 'Z'}}, {"word", {'0', '9', 'A', 'Z', '_', '_', 'a', 'z'}},
      {"xdigit", {'0', '9', 'A', 'F', 'a', 'f'}}};
  for (auto& e : t) if (n == e.name) return &e.r;
  return nullptr;
}

// ------------------------------------------------------------------ parser
struct P {
  const std::string& s;
  size_t i = 0;
  Ast ast;
  struct Fl { bool fold = false, dotnl = false, multiline = false, swapGreed = false; } fl;

  explicit P(const std::string& src) : s(src) {}
  bool eof() const { return i >= s.size(); }
  char peek(size_t k = 0) const { return i + k < s.size() ? s[i + k] : '\0'; }

  int32_t rune() {
    unsigned char c = s[i];
    if (c < 0x80) { i++; return c; }
    int n = c >= 0xF0 ? 4 : c >= 0xE0 ? 3 : c >= 0xC0 ? 2 : 0;
    if (!n || i + n > s.size()) syntaxErr("invalid UTF-8");
    int32_t r = c & (0xFF >> (n + 1));
    for (int k = 1; k < n; k++) r = (r << 6) | (static_cast<unsigned char>(s[i + k]) & 0x3F);
    i += n;
    return r;
  }

  bool classLike(int n) const {
    const auto& x = ast.at(n);
    return (x.kind == Node::Lit && x.r.size() == 1) || x.kind == Node::Class || x.kind == Node::AnyNotNL || x.kind == Node::Any;
  }

  int lit(int32_t r) {
    int n = ast.add(Node::Lit);
    if (fl.fold && r >= 0x80) unsupported("Unicode case folding (a rune past U+007F under (?i))");
    if (fl.fold) {  // minFoldRune
      int32_t m = r;
      for (int32_t f = foldNext(r); f != r; f = foldNext(f)) m = std::min(m, f);
      r = m;
      ast.at(n).fold = true;
    }
    ast.at(n).r = {r};
    return n;
  }

  // class node -> literal when it holds one rune or one simple case pair (parser.push rules)
  int finishClass(Ranges& rg) {
    auto& v = rg.v;
    int n;
    if (v.size() == 2 && v[0] == v[1]) {
      n = ast.add(Node::Lit); ast.at(n).r = {v[0]}; ast.at(n).fold = false; return n;
    }
    if (v.size() == 4 && v[0] == v[1] && v[2] == v[3] && foldNext(v[0]) == v[2] && foldNext(v[2]) == v[0]) {
      n = ast.add(Node::Lit); ast.at(n).r = {v[0]}; ast.at(n).fold = true; return n;
    }
    n = ast.add(Node::Class);
    ast.at(n).r = v;
    return n;
  }

  int32_t escapeRune() {  // after '\\'; parser.parseEscape
    if (eof()) syntaxErr("trailing backslash");
    size_t save = i;
    int32_t c = rune();
    auto hex = [](int32_t x) { return x >= '0' && x <= '9' ? x - '0' : x >= 'a' && x <= 'f' ? x - 'a' + 10 : x >= 'A' && x <= 'F' ? x - 'A' + 10 : -1; };
    switch (c) {
      case '1': case '2': case '3': case '4': case '5': case '6': case '7':
        if (eof() || peek() < '0' || peek() > '7') break;
        [[fallthrough]];
      case '0': {
        int32_t r = c - '0';
        for (int k = 1; k < 3 && !eof() && peek() >= '0' && peek() <= '7'; k++) r = r * 8 + (s[i++] - '0');
        return r;
      }
      case 'x': {
        if (eof()) break;
        int32_t d = rune();
        if (d == '{') {
          int32_t r = 0; int nh = 0; bool closed = false;
          while (!eof()) {
            int32_t e = rune();
            if (e == '}') { closed = true; break; }
            if (hex(e) < 0) { nh = 0; break; }
            r = r * 16 + hex(e); nh++;
            if (r > kMaxRune) { nh = 0; break; }
          }
          if (closed && nh) return r;
          break;
        }
        if (eof()) break;
        int32_t e = rune();
        if (hex(d) < 0 || hex(e) < 0) break;
        return hex(d) * 16 + hex(e);
      }
      case 'a': return 7; case 'f': return 12; case 'n': return 10; case 'r': return 13; case 't': return 9; case 'v': return 11;
      default:
        if (c < 0x80 && !isalnum(c)) return c;
    }
    i = save;
    syntaxErr("invalid escape sequence");
  }

  void addGroup(Ranges& dst, const std::vector<int32_t>& g, bool neg) {
    Ranges t;
    for (size_t k = 0; k + 1 < g.size(); k += 2) { if (fl.fold) addFolded(t, g[k], g[k + 1]); else t.add(g[k], g[k + 1]); }
    t.canon();
    if (neg) t.negate();
    dst.v.insert(dst.v.end(), t.v.begin(), t.v.end());
  }

  int bracket() {  // at '['
    i++;
    Ranges rg;
    bool neg = false;
    if (peek() == '^') { neg = true; i++; }
    bool first = true;
    while (eof() || peek() != ']' || first) {
      if (eof()) syntaxErr("missing closing ]");
      first = false;
      if (peek() == '[' && peek(1) == ':') {
        size_t e = s.find(":]", i + 2);
        if (e != std::string::npos) {
          std::string name = s.substr(i + 2, e - i - 2);
          bool n2 = !name.empty() && name[0] == '^';
          if (n2) name.erase(0, 1);
          const auto* g = posixClass(name);
          if (!g) syntaxErr("invalid character class range");
          addGroup(rg, *g, n2);
          i = e + 2;
          continue;
        }
      }
      if (peek() == '\\' && (peek(1) == 'p' || peek(1) == 'P')) unsupported("Unicode class \\p");
      if (peek() == '\\' && !perlClass(peek(1)).empty()) {
        char c = peek(1);
        addGroup(rg, perlClass(c), c >= 'A' && c <= 'Z');
        i += 2;
        continue;
      }
      auto one = [&]() -> int32_t { if (peek() == '\\') { i++; return escapeRune(); } return rune(); };
      int32_t lo = one(), hi = lo;
      if (peek() == '-' && i + 1 < s.size() && peek(1) != ']') {
        i++;
        if (eof()) syntaxErr("missing closing ]");
        hi = one();
        if (hi < lo) syntaxErr("invalid character class range");
      }
      if (fl.fold) addFolded(rg, lo, hi); else rg.add(lo, hi);
    }
    i++;
    rg.canon();
    if (neg) rg.negate();
    return finishClass(rg);
  }

  // One atom without its quantifier.  Returns -1 at '|' / ')' / end.
  int atom() {
    if (eof()) return -1;
    char c = peek();
    switch (c) {
      case '|': case ')': return -1;
      case '(': return group();
      case '[': return bracket();
      case '.': i++; return ast.add(fl.dotnl ? Node::Any : Node::AnyNotNL);
      case '^': i++; return ast.add(fl.multiline ? Node::BeginLine : Node::BeginText);
      case '$': i++; return ast.add(fl.multiline ? Node::EndLine : Node::EndText);
      case '*': case '+': case '?': syntaxErr("missing argument to repetition operator");
      case '\\': {
        char d = peek(1);
        if (d == 'A') { i += 2; return ast.add(Node::BeginText); }
        if (d == 'z') { i += 2; return ast.add(Node::EndText); }
        if (d == 'b') { i += 2; return ast.add(Node::WordB); }
        if (d == 'B') { i += 2; return ast.add(Node::NoWordB); }
        if (d == 'C') syntaxErr("invalid escape sequence");
        if (d == 'p' || d == 'P') unsupported("Unicode class \\p");
        if (d == 'Q') return -2;  // handled by caller (expands to several literals)
        if (!perlClass(d).empty() && d != '\0') {
          Ranges rg;
          addGroup(rg, perlClass(d), d >= 'A' && d <= 'Z');
          i += 2;
          rg.canon();
          return finishClass(rg);
        }
        i++;
        return lit(escapeRune());
      }
      default: return lit(rune());
    }
  }

  int group() {  // at '('
    Fl saved = fl;
    int cap = 0;
    if (peek(1) == '?') {
      if (s.compare(i, 4, "(?P<") == 0 || (s.compare(i, 3, "(?<") == 0 && peek(3) != '=' && peek(3) != '!')) {
        size_t b = i + (peek(2) == 'P' ? 4 : 3);
        size_t e = s.find('>', b);
        if (e == std::string::npos || e == b) syntaxErr("invalid named capture");
        for (size_t k = b; k < e; k++) if (!isalnum(static_cast<unsigned char>(s[k])) && s[k] != '_') syntaxErr("invalid named capture");
        i = e + 1;
        cap = ++ast.ncap;
      } else {
        size_t j = i + 2;
        Fl nf = fl;
        bool negv = false, saw = false;
        for (;; j++) {
          if (j >= s.size()) syntaxErr("missing closing )");
          char c = s[j];
          if (c == 'i') { nf.fold = !negv; saw = true; }
          else if (c == 'm') { nf.multiline = !negv; saw = true; }
          else if (c == 's') { nf.dotnl = !negv; saw = true; }
          else if (c == 'U') { nf.swapGreed = !negv; saw = true; }
          else if (c == '-') { if (negv) syntaxErr("invalid or unsupported Perl syntax"); negv = true; saw = false; }
          else if (c == ':' || c == ')') {
            if (negv && !saw) syntaxErr("invalid or unsupported Perl syntax");
            if (c == ')') { fl = nf; i = j + 1; return -3; }  // flag-only group: affects the rest of the enclosing group
            i = j + 1;
            fl = nf;
            int inner = alternation();
            if (peek() != ')') syntaxErr("missing closing )");
            i++;
            fl = saved;
            return inner;
          } else syntaxErr("invalid or unsupported Perl syntax");
        }
      }
    } else {
      i++;
      cap = ++ast.ncap;
    }
    int inner = alternation();
    if (peek() != ')') syntaxErr("missing closing )");
    i++;
    fl = saved;
    int n = ast.add(Node::Capture);
    ast.at(n).cap = cap;
    ast.at(n).kids = {inner};
    return n;
  }

  static bool repeatOK(const Ast& a, int n, int budget) {
    const auto& x = a.at(n);
    if (x.kind == Node::Repeat) {
      int m = x.max;
      if (m == 0) return true;
      if (m < 0) m = x.min;
      if (m > budget) return false;
      if (m > 0) budget /= m;
    }
    for (int k : x.kids) if (!repeatOK(a, k, budget)) return false;
    return true;
  }

  bool braces(int& mn, int& mx) {  // at '{'; leaves i after '}' on success
    size_t j = i + 1;
    auto num = [&](int& out) -> bool {
      if (j >= s.size() || !isdigit(static_cast<unsigned char>(s[j]))) return false;
      if (s[j] == '0' && j + 1 < s.size() && isdigit(static_cast<unsigned char>(s[j + 1]))) return false;
      size_t b = j;
      while (j < s.size() && isdigit(static_cast<unsigned char>(s[j]))) j++;
      if (j - b >= 8) { out = -1; return true; }
      out = std::stoi(s.substr(b, j - b));
      return true;
    };
    if (!num(mn)) return false;
    if (j >= s.size()) return false;
    if (s[j] == ',') {
      j++;
      if (j >= s.size()) return false;
      if (s[j] == '}') mx = -1;
      else { if (!num(mx)) return false; if (mx == -1) mn = -1; }
    } else mx = mn;
    if (j >= s.size() || s[j] != '}') return false;
    i = j + 1;
    return true;
  }

  // concat: pieces until '|' or ')' or end
  int concat() {
    std::vector<int> pieces;
    bool lastWasRepeat = false;
    while (!eof()) {
      char c = peek();
      if (c == '|' || c == ')') break;
      if (c == '*' || c == '+' || c == '?' || c == '{') {
        int mn = 0, mx = 0;
        Node k = c == '*' ? Node::Star : c == '+' ? Node::Plus : c == '?' ? Node::Quest : Node::Repeat;
        if (c == '{') {
          size_t save = i;
          if (!braces(mn, mx)) { i = save; pieces.push_back(lit(rune())); lastWasRepeat = false; continue; }
          if (mn < 0 || mn > 1000 || mx > 1000 || (mx >= 0 && mn > mx)) syntaxErr("invalid repeat count");
        } else i++;
        bool lazy = fl.swapGreed;
        if (peek() == '?') { i++; lazy = !lazy; }
        if (lastWasRepeat) syntaxErr("invalid nested repetition operator");
        if (pieces.empty()) syntaxErr("missing argument to repetition operator");
        int n = ast.add(k);
        ast.at(n).kids = {pieces.back()};
        ast.at(n).lazy = lazy; ast.at(n).min = mn; ast.at(n).max = mx;
        pieces.back() = n;
        if (k == Node::Repeat && (mn >= 2 || mx >= 2) && !repeatOK(ast, n, 1000)) syntaxErr("invalid repeat count");
        lastWasRepeat = true;
        continue;
      }
      lastWasRepeat = false;
      int a = atom();
      if (a == -1) break;
      if (a == -3) continue;  // (?flags)
      if (a == -2) {          // \Q...\E
        size_t e = s.find("\\E", i + 2);
        size_t stop = e == std::string::npos ? s.size() : e;
        i += 2;
        while (i < stop) pieces.push_back(lit(rune()));
        i = e == std::string::npos ? s.size() : e + 2;
        continue;
      }
      pieces.push_back(a);
    }
    return makeConcat(pieces);
  }

  // Go merges adjacent literal *stack entries* with equal fold flag (parser.maybeConcat); a quantifier
  // only ever binds the last rune because the top literal is merged lazily.
  int makeConcat(std::vector<int>& pieces) {
    std::vector<int> merged;
    for (int p : pieces) {
      if (!merged.empty() && ast.at(p).kind == Node::Lit && ast.at(merged.back()).kind == Node::Lit &&
          ast.at(p).fold == ast.at(merged.back()).fold) {
        auto& dst = ast.at(merged.back()).r;
        dst.insert(dst.end(), ast.at(p).r.begin(), ast.at(p).r.end());
      } else merged.push_back(p);
    }
    if (merged.empty()) return ast.add(Node::Empty);
    if (merged.size() == 1) return merged[0];
    int n = ast.add(Node::Concat);
    for (int p : merged) {  // parser.collapse: splice nested concats, no further literal merging
      if (ast.at(p).kind == Node::Concat) for (int k : ast.at(p).kids) ast.at(n).kids.push_back(k);
      else ast.at(n).kids.push_back(p);
    }
    return n;
  }

  // ---- alternation: eager class merging (swapVerticalBar) then factor() rounds 1-4
  void mergeClass(int dst, int src) {
    auto& d = ast.at(dst);
    const auto& x = ast.at(src);
    auto litRanges = [&](const Ast::N& l, Ranges& out) { if (l.fold) addFolded(out, l.r[0], l.r[0]); else out.add(l.r[0], l.r[0]); };
    switch (d.kind) {
      case Node::Any: break;
      case Node::AnyNotNL: {
        bool nl = x.kind == Node::Any || (x.kind == Node::Lit && x.r[0] == '\n');
        if (x.kind == Node::Class) for (size_t k = 0; k + 1 < x.r.size(); k += 2) if (x.r[k] <= '\n' && '\n' <= x.r[k + 1]) nl = true;
        if (nl) d.kind = Node::Any;
        break;
      }
      case Node::Class: {
        Ranges rg; rg.v = d.r;
        if (x.kind == Node::Lit) litRanges(x, rg); else rg.v.insert(rg.v.end(), x.r.begin(), x.r.end());
        d.r = rg.v;
        break;
      }
      case Node::Lit: {
        if (x.r[0] == d.r[0] && x.fold == d.fold) break;
        Ranges rg;
        litRanges(d, rg);
        litRanges(x, rg);
        d.kind = Node::Class; d.fold = false; d.r = rg.v;
        break;
      }
      default: break;
    }
  }
  void cleanAlt(int n) {
    auto& x = ast.at(n);
    if (x.kind != Node::Class) return;
    Ranges rg; rg.v = x.r; rg.canon(); x.r = rg.v;
    if (x.r.size() == 2 && x.r[0] == 0 && x.r[1] == kMaxRune) { x.kind = Node::Any; x.r.clear(); }
    else if (x.r.size() == 4 && x.r[0] == 0 && x.r[1] == 9 && x.r[2] == 11 && x.r[3] == kMaxRune) { x.kind = Node::AnyNotNL; x.r.clear(); }
  }
  static int rankOf(Node k) { return static_cast<int>(k); }

  int alternation() {
    std::vector<int> alts;
    for (;;) {
      int c = concat();
      if (alts.size() >= 1 && classLike(c) && classLike(alts.back())) {
        int a = alts.back();
        if (rankOf(ast.at(c).kind) > rankOf(ast.at(a).kind)) { std::swap(a, c); alts.back() = a; }
        mergeClass(a, c);
      } else {
        if (!alts.empty()) cleanAlt(alts.back());
        alts.push_back(c);
      }
      if (peek() == '|') { i++; continue; }
      break;
    }
    if (!alts.empty()) cleanAlt(alts.back());
    return makeAlt(alts);
  }

  bool sameTree(int a, int b) const {
    const auto& x = ast.at(a); const auto& y = ast.at(b);
    if (x.kind != y.kind) return false;
    switch (x.kind) {
      case Node::Lit: if (x.r != y.r || x.fold != y.fold) return false; break;
      case Node::Class: if (x.r != y.r) return false; break;
      case Node::Star: case Node::Plus: case Node::Quest: if (x.lazy != y.lazy) return false; break;
      case Node::Repeat: if (x.lazy != y.lazy || x.min != y.min || x.max != y.max) return false; break;
      case Node::Capture: if (x.cap != y.cap) return false; break;
      default: break;
    }
    if (x.kids.size() != y.kids.size()) return false;
    for (size_t k = 0; k < x.kids.size(); k++) if (!sameTree(x.kids[k], y.kids[k])) return false;
    return true;
  }

  int leadLitNode(int n) const {
    if (ast.at(n).kind == Node::Concat && !ast.at(n).kids.empty()) n = ast.at(n).kids[0];
    return ast.at(n).kind == Node::Lit ? n : -1;
  }
  int stripLeadString(int n, size_t cnt) {
    auto& x = ast.at(n);
    if (x.kind == Node::Concat && !x.kids.empty()) {
      int sub = stripLeadString(x.kids[0], cnt);
      ast.at(n).kids[0] = sub;
      if (ast.at(sub).kind == Node::Empty) {
        auto& y = ast.at(n);
        if (y.kids.size() <= 1) { y.kind = Node::Empty; y.kids.clear(); }
        else if (y.kids.size() == 2) return y.kids[1];
        else y.kids.erase(y.kids.begin());
      }
      return n;
    }
    if (x.kind == Node::Lit) {
      x.r.erase(x.r.begin(), x.r.begin() + cnt);
      if (x.r.empty()) x.kind = Node::Empty;
    }
    return n;
  }
  int leadNode(int n) const {
    const auto& x = ast.at(n);
    if (x.kind == Node::Empty) return -1;
    if (x.kind == Node::Concat && !x.kids.empty()) return ast.at(x.kids[0]).kind == Node::Empty ? -1 : x.kids[0];
    return n;
  }
  int stripLeadNode(int n) {
    auto& x = ast.at(n);
    if (x.kind == Node::Concat && !x.kids.empty()) {
      x.kids.erase(x.kids.begin());
      if (x.kids.empty()) { x.kind = Node::Empty; return n; }
      if (x.kids.size() == 1) return x.kids[0];
      return n;
    }
    return ast.add(Node::Empty);
  }

  std::vector<int> factor(std::vector<int> sub) {
    if (sub.size() < 2) return sub;
    {  // round 1: common literal prefix of adjacent alternatives
      std::vector<int> out;
      std::vector<int32_t> str; bool strFold = false; size_t start = 0;
      for (size_t k = 0; k <= sub.size(); k++) {
        std::vector<int32_t> istr; bool ifold = false;
        if (k < sub.size()) {
          int ln = leadLitNode(sub[k]);
          if (ln >= 0) { istr = ast.at(ln).r; ifold = ast.at(ln).fold; }
          if (ifold == strFold) {
            size_t same = 0;
            while (same < str.size() && same < istr.size() && str[same] == istr[same]) same++;
            if (same > 0) { str.resize(same); continue; }
          }
        }
        if (k == start) {
        } else if (k == start + 1) out.push_back(sub[start]);
        else {
          int pre = ast.add(Node::Lit);
          ast.at(pre).r = str; ast.at(pre).fold = strFold;
          std::vector<int> rest;
          for (size_t j = start; j < k; j++) rest.push_back(stripLeadString(sub[j], str.size()));
          int suf = makeAlt(rest);
          int cat = ast.add(Node::Concat);
          ast.at(cat).kids = {pre, suf};
          out.push_back(cat);
        }
        start = k; str = istr; strFold = ifold;
      }
      sub.swap(out);
    }
    {  // round 2: common leading class-like piece
      std::vector<int> out;
      int first = -1; size_t start = 0;
      for (size_t k = 0; k <= sub.size(); k++) {
        int ifirst = -1;
        if (k < sub.size()) {
          ifirst = leadNode(sub[k]);
          if (first >= 0 && ifirst >= 0 && sameTree(first, ifirst)) {
            const auto& f = ast.at(first);
            if (classLike(first) || (f.kind == Node::Repeat && f.min == f.max && classLike(f.kids[0]))) continue;
          }
        }
        if (k == start) {
        } else if (k == start + 1) out.push_back(sub[start]);
        else {
          std::vector<int> rest;
          for (size_t j = start; j < k; j++) rest.push_back(stripLeadNode(sub[j]));
          int suf = makeAlt(rest);
          int cat = ast.add(Node::Concat);
          ast.at(cat).kids = {first, suf};
          out.push_back(cat);
        }
        start = k; first = ifirst;
      }
      sub.swap(out);
    }
    {  // round 3: runs of class-like alternatives
      std::vector<int> out;
      size_t start = 0;
      for (size_t k = 0; k <= sub.size(); k++) {
        if (k < sub.size() && classLike(sub[k])) continue;
        if (k == start) {
        } else if (k == start + 1) out.push_back(sub[start]);
        else {
          size_t mx = start;
          for (size_t j = start + 1; j < k; j++) {
            const auto& a = ast.at(sub[mx]); const auto& b = ast.at(sub[j]);
            if (rankOf(a.kind) < rankOf(b.kind) || (a.kind == b.kind && a.r.size() < b.r.size())) mx = j;
          }
          std::swap(sub[start], sub[mx]);
          for (size_t j = start + 1; j < k; j++) mergeClass(sub[start], sub[j]);
          cleanAlt(sub[start]);
          out.push_back(sub[start]);
        }
        if (k < sub.size()) out.push_back(sub[k]);
        start = k + 1;
      }
      sub.swap(out);
    }
    {  // round 4: runs of empty matches
      std::vector<int> out;
      for (size_t k = 0; k < sub.size(); k++) {
        if (k + 1 < sub.size() && ast.at(sub[k]).kind == Node::Empty && ast.at(sub[k + 1]).kind == Node::Empty) continue;
        out.push_back(sub[k]);
      }
      sub.swap(out);
    }
    return sub;
  }

  int makeAlt(const std::vector<int>& alts) {
    if (alts.size() == 1) return alts[0];
    std::vector<int> flat;
    for (int a : alts) {
      if (ast.at(a).kind == Node::Alt) for (int k : ast.at(a).kids) flat.push_back(k);
      else flat.push_back(a);
    }
    flat = factor(flat);
    if (flat.size() == 1) return flat[0];
    int n = ast.add(Node::Alt);
    ast.at(n).kids = flat;
    return n;
  }
};

}  // namespace

Ast parsePattern(const std::string& pattern) {
  P p(pattern);
  int root = p.alternation();
  if (!p.eof()) syntaxErr(p.peek() == ')' ? "unexpected )" : "trailing garbage");
  p.ast.root = root;
  return std::move(p.ast);
}

std::string Ast::repr(int n) const {
  static const char* nm[] = {"nomatch", "empty", "lit", "cc", "dnl", "dot", "bol", "eol", "bot", "eot", "wb", "nwb",
                             "cap", "star", "plus", "quest", "rep", "cat", "alt"};
  const auto& x = at(n);
  std::string s = nm[static_cast<int>(x.kind)];
  if (x.lazy) s += "?";
  if (x.kind == Node::Lit) {
    s += x.fold ? "i{" : "{";
    for (int32_t r : x.r) { if (r >= 0x20 && r < 0x7F) s += static_cast<char>(r); else s += "\\x{" + std::to_string(r) + "}"; }
    s += "}";
  } else if (x.kind == Node::Class) {
    s += "[";
    for (size_t k = 0; k + 1 < x.r.size(); k += 2) { if (k) s += " "; s += std::to_string(x.r[k]) + "-" + std::to_string(x.r[k + 1]); }
    s += "]";
  } else if (x.kind == Node::Repeat) s += "{" + std::to_string(x.min) + "," + std::to_string(x.max) + "}";
  else if (x.kind == Node::Capture) s += std::to_string(x.cap);
  if (!x.kids.empty()) {
    s += "(";
    for (size_t k = 0; k < x.kids.size(); k++) { if (k) s += ","; s += repr(x.kids[k]); }
    s += ")";
  }
  return s;
}

// ====================================================================== Thompson NFA
namespace {

struct NfaB {
  HostNfa out;
  const Ast& ast;
  int depth = 0;
  explicit NfaB(const Ast& a) : ast(a) {}

  uint32_t push(cxg_nfa_state s) { out.states.push_back(s); return static_cast<uint32_t>(out.states.size() - 1); }
  static cxg_nfa_state blank(uint8_t kind) {
    cxg_nfa_state s;
    std::memset(&s, 0, sizeof s);
    s.kind = kind; s.next = s.left = s.right = CXG_NFA_INVALID;
    return s;
  }
  uint32_t byteRange(uint8_t lo, uint8_t hi) { auto s = blank(CXG_NFA_BYTE_RANGE); s.lo = lo; s.hi = hi; return push(s); }
  uint32_t eps(uint32_t next = CXG_NFA_INVALID) { auto s = blank(CXG_NFA_EPSILON); s.next = next; return push(s); }
  uint32_t split(uint32_t l, uint32_t r) { auto s = blank(CXG_NFA_SPLIT); s.left = l; s.right = r; return push(s); }
  void link(uint32_t from, uint32_t to) {
    auto& s = out.states[from];
    if (s.kind == CXG_NFA_BYTE_RANGE || s.kind == CXG_NFA_EPSILON || s.kind == CXG_NFA_CAPTURE || s.kind == CXG_NFA_LOOK) { s.next = to; return; }
    // The reference inserts an epsilon when Patch fails (compile.go:1246-1252); no fragment built
    // here ends in an unpatchable state, so reaching this is a front-end bug, not a pattern property.
    unsupported("internal: unpatchable fragment end");
  }

  struct F { uint32_t in, out; };

  F runes(const Ast::N& x) {
    if (x.r.empty()) { uint32_t e = eps(); return {e, e}; }
    uint32_t first = CXG_NFA_INVALID, prev = CXG_NFA_INVALID;
    auto chain = [&](int32_t r, uint32_t& f, uint32_t& p) {
      uint8_t b[4]; int n;
      if (r < 0x80) { b[0] = r; n = 1; }
      else if (r < 0x800) { b[0] = 0xC0 | (r >> 6); b[1] = 0x80 | (r & 63); n = 2; }
      else if (r < 0x10000) { b[0] = 0xE0 | (r >> 12); b[1] = 0x80 | ((r >> 6) & 63); b[2] = 0x80 | (r & 63); n = 3; }
      else { b[0] = 0xF0 | (r >> 18); b[1] = 0x80 | ((r >> 12) & 63); b[2] = 0x80 | ((r >> 6) & 63); b[3] = 0x80 | (r & 63); n = 4; }
      for (int k = 0; k < n; k++) {
        uint32_t id = byteRange(b[k], b[k]);
        if (f == CXG_NFA_INVALID) f = id;
        if (p != CXG_NFA_INVALID) out.states[p].next = id;
        p = id;
      }
    };
    for (int32_t r : x.r) {
      bool letter = (r >= 'a' && r <= 'z') || (r >= 'A' && r <= 'Z');
      if (x.fold && letter) {  // compileFoldCaseRune compile.go:273-304
        uint32_t uf = CXG_NFA_INVALID, up = CXG_NFA_INVALID, lf = CXG_NFA_INVALID, lp = CXG_NFA_INVALID;
        chain(r & ~0x20, uf, up);
        chain(r | 0x20, lf, lp);
        uint32_t join = eps();
        out.states[up].next = join;
        out.states[lp].next = join;
        uint32_t sp = split(uf, lf);
        if (prev == CXG_NFA_INVALID) first = sp; else out.states[prev].next = sp;
        prev = join;
      } else chain(r, first, prev);
    }
    return {first, prev};
  }

  // ---- classes that reach past U+007F and `.`: byte automata over UTF-8, state for state the reference's -------------------
  uint32_t rangeTo(int lo, int hi, uint32_t next) { auto s = blank(CXG_NFA_BYTE_RANGE); s.lo = static_cast<uint8_t>(lo); s.hi = static_cast<uint8_t>(hi); s.next = next; return push(s); }
  uint32_t sparseTo(std::initializer_list<std::pair<int, int>> rs, uint32_t target) {
    auto s = blank(CXG_NFA_SPARSE);
    s.trans_off = static_cast<uint32_t>(out.trans.size());
    for (auto& r : rs) out.trans.push_back({static_cast<uint8_t>(r.first), static_cast<uint8_t>(r.second), 0, target});
    s.trans_len = static_cast<uint32_t>(rs.size());
    return push(s);
  }
  uint32_t splitChain(const std::vector<uint32_t>& t, size_t from = 0) {  // buildSplitChain compile.go:1298-1311: the innermost split first
    if (t.size() - from == 1) return t[from];
    if (t.size() - from == 2) return split(t[from], t[from + 1]);
    const uint32_t right = splitChain(t, from + 1);
    return split(t[from], right);
  }
  // the eight shapes of a well-formed multi-byte sequence: lead range, then the continuation ranges
  struct Utf8Shape { int n; uint8_t r[4][2]; };
  static const Utf8Shape* utf8Shapes() {
    static const Utf8Shape t[8] = {{2, {{0xC2, 0xDF}, {0x80, 0xBF}}},
                                   {3, {{0xE0, 0xE0}, {0xA0, 0xBF}, {0x80, 0xBF}}}, {3, {{0xE1, 0xEC}, {0x80, 0xBF}, {0x80, 0xBF}}},
                                   {3, {{0xED, 0xED}, {0x80, 0x9F}, {0x80, 0xBF}}}, {3, {{0xEE, 0xEF}, {0x80, 0xBF}, {0x80, 0xBF}}},
                                   {4, {{0xF0, 0xF0}, {0x90, 0xBF}, {0x80, 0xBF}, {0x80, 0xBF}}}, {4, {{0xF1, 0xF3}, {0x80, 0xBF}, {0x80, 0xBF}, {0x80, 0xBF}}},
                                   {4, {{0xF4, 0xF4}, {0x80, 0x8F}, {0x80, 0xBF}, {0x80, 0xBF}}}};
    return t;
  }

  F dot(bool withNL) {  // compileUTF8Any compile.go:1142-1222: suffixes shared through a 64-entry direct-mapped cache (nfa/utf8_suffix.go)
    const uint32_t end = eps();
    struct Slot { bool used; uint32_t to; uint8_t lo, hi; uint32_t id; } cache[64] = {};
    auto shared = [&](uint32_t to, uint8_t lo, uint8_t hi) {
      uint64_t h = 14695981039346656037ull;                         // FNV-1a over (target, lo, hi), utf8_suffix.go:70-77
      for (uint64_t v : {static_cast<uint64_t>(to), static_cast<uint64_t>(lo), static_cast<uint64_t>(hi)}) h = (h ^ v) * 1099511628211ull;
      Slot& c = cache[h % 64u];
      if (c.used && c.to == to && c.lo == lo && c.hi == hi) return c.id;
      c = Slot{true, to, lo, hi, rangeTo(lo, hi, to)};              // a collision overwrites
      return c.id;
    };
    std::vector<uint32_t> alts;
    alts.push_back(withNL ? rangeTo(0x00, 0x7F, end) : sparseTo({{0x00, 0x09}, {0x0B, 0x7F}}, end));
    const Utf8Shape* shp = utf8Shapes();
    for (int q = 0; q < 8; q++) {
      uint32_t to = end;
      for (int k = shp[q].n - 1; k >= 0; k--) to = shared(to, shp[q].r[k][0], shp[q].r[k][1]);
      alts.push_back(to);
    }
    alts.push_back(sparseTo({{0x80, 0xBF}, {0xC0, 0xC1}, {0xF5, 0xFF}}, end));   // bytes that begin no sequence match alone; C2..F4 alone do not
    return {splitChain(alts), end};
  }

  // byte sequences of the code points lo..hi (>= U+0080), compileUTF8Range compile.go:600-840, appended to `alts`
  void utf8Range(int32_t lo, int32_t hi, uint32_t end, std::vector<uint32_t>& alts) {
    auto three = [&](int32_t a, int32_t z) {                        // compileUTF83ByteRangeSimple :740-792: one chain per (lead, first continuation)
      const int la = 0xE0 | (a >> 12), ca = 0x80 | ((a >> 6) & 63), da = 0x80 | (a & 63);
      const int lz = 0xE0 | (z >> 12), cz = 0x80 | ((z >> 6) & 63), dz = 0x80 | (z & 63);
      for (int lead = la; lead <= lz; lead++) {
        int c1a = lead == la ? ca : lead == 0xE0 ? 0xA0 : 0x80, c1z = lead == lz ? cz : lead == 0xED ? 0x9F : 0xBF;
        if (la == lz) { c1a = ca; c1z = cz; }
        for (int c1 = c1a; c1 <= c1z; c1++) {
          const uint32_t s2 = rangeTo((lead == la && c1 == ca) ? da : 0x80, (lead == lz && c1 == cz) ? dz : 0xBF, end);
          alts.push_back(rangeTo(lead, lead, rangeTo(c1, c1, s2)));
        }
      }
    };
    if (lo <= 0x7FF) {                                              // two bytes :663-701
      const int32_t z = std::min<int32_t>(hi, 0x7FF);
      const int la = 0xC0 | (lo >> 6), ca = 0x80 | (lo & 63), lz = 0xC0 | (z >> 6), cz = 0x80 | (z & 63);
      if (la == lz) alts.push_back(rangeTo(la, la, rangeTo(ca, cz, end)));
      else {
        alts.push_back(rangeTo(la, la, rangeTo(ca, 0xBF, end)));
        if (lz > la + 1) alts.push_back(rangeTo(la + 1, lz - 1, rangeTo(0x80, 0xBF, end)));
        alts.push_back(rangeTo(lz, lz, rangeTo(0x80, cz, end)));
      }
      lo = 0x800;
    }
    if (lo > hi) return;
    if (lo <= 0xFFFF) {                                             // three bytes, never a surrogate :706-737
      int32_t a = lo, z = std::min<int32_t>(hi, 0xFFFF);
      if (a <= 0xD7FF && z >= 0xE000) { three(a, 0xD7FF); three(0xE000, z); }
      else if (!(a >= 0xD800 && z <= 0xDFFF)) {
        if (a >= 0xD800 && a <= 0xDFFF) a = 0xE000;
        if (z >= 0xD800 && z <= 0xDFFF) z = 0xD7FF;
        if (a <= z) three(a, z);
      }
      lo = 0x10000;
    }
    if (lo > hi) return;
    for (int lead = 0xF0 | (lo >> 18); lead <= (0xF0 | (std::min<int32_t>(hi, kMaxRune) >> 18)); lead++) {   // four bytes: whole lead bytes :796-840
      const uint32_t c3 = rangeTo(0x80, 0xBF, end), c2 = rangeTo(0x80, 0xBF, c3);
      alts.push_back(rangeTo(lead, lead, rangeTo(lead == 0xF0 ? 0x90 : 0x80, lead == 0xF4 ? 0x8F : 0xBF, c2)));
    }
  }

  F wideClass(const Ast::N& x) {  // compileUnicodeClass / compileUnicodeClassLarge compile.go:440-590
    int64_t total = 0;
    for (size_t k = 0; k + 1 < x.r.size() && total <= 256; k += 2) total += static_cast<int64_t>(x.r[k + 1]) - x.r[k] + 1;
    if (total <= 256) {                                             // few characters: an alternation of one-rune literals
      std::vector<F> fs;
      Ast::N one; one.kind = Node::Lit; one.fold = false;
      for (size_t k = 0; k + 1 < x.r.size(); k += 2)
        for (int32_t r = x.r[k]; r <= x.r[k + 1]; r++) { one.r = {r}; fs.push_back(runes(one)); }
      if (fs.size() == 1) return fs[0];
      std::vector<uint32_t> ins;
      for (auto& f : fs) ins.push_back(f.in);
      const uint32_t sp = splitChain(ins), join = eps();
      for (auto& f : fs) out.states[f.out].next = join;
      return {sp, join};
    }
    std::vector<std::pair<int, int>> low;                           // the part below U+0080, and the rest
    std::vector<std::pair<int32_t, int32_t>> rest;
    for (size_t k = 0; k + 1 < x.r.size(); k += 2) {
      const int32_t lo = x.r[k], hi = x.r[k + 1];
      if (hi < 0x80) low.emplace_back(lo, hi);
      else if (lo >= 0x80) rest.emplace_back(lo, hi);
      else { low.emplace_back(lo, 0x7F); rest.emplace_back(0x80, hi); }
    }
    const uint32_t target = eps();
    std::vector<uint32_t> alts;
    if (low.size() == 1) alts.push_back(rangeTo(low[0].first, low[0].second, target));
    else if (!low.empty()) {
      auto s = blank(CXG_NFA_SPARSE);
      s.trans_off = static_cast<uint32_t>(out.trans.size());
      for (auto& r : low) out.trans.push_back({static_cast<uint8_t>(r.first), static_cast<uint8_t>(r.second), 0, target});
      s.trans_len = static_cast<uint32_t>(low.size());
      alts.push_back(push(s));
    }
    if (rest.size() == 1 && rest[0].first <= 0x80 && rest[0].second >= kMaxRune) {
      // everything past U+007F (`[^"]`, `\S`, `\D`): every well-formed sequence, unshared (buildUTF8NonASCIIBranches :845-917),
      // and any byte >= 0x80 on its own behind them (:557-567)
      const Utf8Shape* shp = utf8Shapes();
      for (int q = 0; q < 8; q++) {
        uint32_t to = target;
        for (int k = shp[q].n - 1; k >= 0; k--) to = rangeTo(shp[q].r[k][0], shp[q].r[k][1], to);
        alts.push_back(to);
      }
      alts.push_back(rangeTo(0x80, 0xFF, target));
    } else {
      for (auto& r : rest) utf8Range(r.first, r.second, target, alts);
    }
    if (alts.empty()) { uint32_t a = eps(), z = eps(); return {a, z}; }   // compileNoMatch (a class of surrogates only)
    if (alts.size() == 1) return {alts[0], target};
    return {splitChain(alts), target};
  }

  F cls(const Ast::N& x) {  // compileCharClass compile.go:384-432
    if (x.r.empty()) { uint32_t a = eps(), b = eps(); return {a, b}; }
    for (int32_t r : x.r) if (r > 127) return wideClass(x);
    if (x.r.size() == 2) { uint32_t id = byteRange(x.r[0], x.r[1]); return {id, id}; }
    uint32_t target = eps();
    auto s = blank(CXG_NFA_SPARSE);
    s.trans_off = static_cast<uint32_t>(out.trans.size());
    for (size_t k = 0; k + 1 < x.r.size(); k += 2) out.trans.push_back({static_cast<uint8_t>(x.r[k]), static_cast<uint8_t>(x.r[k + 1]), 0, target});
    s.trans_len = static_cast<uint32_t>(x.r.size() / 2);
    return {push(s), target};
  }

  bool nullable(int n) const {  // canMatchEmpty compile.go:1388-1430
    const auto& x = ast.at(n);
    switch (x.kind) {
      case Node::Empty: return true;
      case Node::Lit: return x.r.empty();
      case Node::Class: case Node::Any: case Node::AnyNotNL: case Node::NoMatch: return false;
      case Node::Capture: return x.kids.empty() || nullable(x.kids[0]);
      case Node::Star: case Node::Quest: return true;
      case Node::Plus: return nullable(x.kids[0]);
      case Node::Repeat: return x.min == 0 || nullable(x.kids[0]);
      case Node::Concat: for (int k : x.kids) if (!nullable(k)) return false; return true;
      case Node::Alt: for (int k : x.kids) if (nullable(k)) return true; return false;
      default: return true;
    }
  }

  F loop(int sub, bool lazy, bool plus) {  // compileStar/compilePlus compile.go:1312-1347,1433-1456
    if (!plus && nullable(sub)) {         // compileStarViaPlus compile.go:1351-1385
      F f = frag(sub);
      uint32_t end = eps();
      uint32_t pl = lazy ? split(end, f.in) : split(f.in, end);
      link(f.out, pl);
      uint32_t q = lazy ? split(end, f.in) : split(f.in, end);
      return {q, end};
    }
    F f = frag(sub);
    uint32_t end = eps();
    uint32_t sp = lazy ? split(end, f.in) : split(f.in, end);
    link(f.out, sp);
    return {plus ? f.in : sp, end};
  }
  F quest(int sub, bool lazy) {  // compileQuest compile.go:1459-1481
    F f = frag(sub);
    uint32_t end = eps();
    uint32_t sp = lazy ? split(end, f.in) : split(f.in, end);
    link(f.out, end);
    return {sp, end};
  }
  F seq(const std::vector<std::function<F()>>& parts) {
    if (parts.empty()) { uint32_t e = eps(); return {e, e}; }
    F f = parts[0]();
    for (size_t k = 1; k < parts.size(); k++) { F g = parts[k](); link(f.out, g.in); f.out = g.out; }
    return f;
  }

  F frag(int n) {
    if (++depth > 100) unsupported("pattern too deeply nested");
    struct G { int& d; ~G() { d--; } } g{depth};
    const auto& x = ast.at(n);
    switch (x.kind) {
      case Node::Lit: return runes(x);
      case Node::Class: return cls(x);
      case Node::Empty: { uint32_t e = eps(); return {e, e}; }
      case Node::Any: return dot(true);            // compileAnyChar compile.go:977-992 (default configuration)
      case Node::AnyNotNL: return dot(false);      // compileAnyCharNotNL compile.go:995-1010
      case Node::BeginText: case Node::EndText:      // (round 4: \A / ^ inside an unanchored pattern is served, program.cc; \z / $ is refused there)
      case Node::BeginLine: case Node::EndLine:
      case Node::WordB: case Node::NoWordB: {       // compile.go:286-289 addLook; cxg_nfa_state.lo = nfa.Look (nfa/nfa.go:92-117)
        auto s = blank(CXG_NFA_LOOK);
        s.lo = x.kind == Node::BeginText ? 0 : x.kind == Node::EndText ? 1 : x.kind == Node::BeginLine ? 2 : x.kind == Node::EndLine ? 3 : x.kind == Node::WordB ? 4 : 5;
        const uint32_t id = push(s);
        return {id, id};
      }
      case Node::NoMatch: unsupported("empty character class");
      case Node::Concat: {
        std::vector<std::function<F()>> parts;
        for (int k : x.kids) parts.push_back([this, k] { return frag(k); });
        return seq(parts);
      }
      case Node::Alt: {  // compileAlternate compile.go:1259-1309
        std::vector<F> fs;
        for (int k : x.kids) fs.push_back(frag(k));
        std::function<uint32_t(size_t)> chain = [&](size_t from) -> uint32_t {
          if (fs.size() - from == 1) return fs[from].in;
          if (fs.size() - from == 2) return split(fs[from].in, fs[from + 1].in);
          uint32_t right = chain(from + 1);
          return split(fs[from].in, right);
        };
        uint32_t sp = chain(0);
        uint32_t join = eps();
        for (auto& f : fs) {
          auto& s = out.states[f.out];
          if (s.kind == CXG_NFA_BYTE_RANGE || s.kind == CXG_NFA_EPSILON || s.kind == CXG_NFA_CAPTURE || s.kind == CXG_NFA_LOOK) s.next = join;
        }
        return {sp, join};
      }
      case Node::Star: return loop(x.kids[0], x.lazy, false);
      case Node::Plus: return loop(x.kids[0], x.lazy, true);
      case Node::Quest: return quest(x.kids[0], x.lazy);
      case Node::Repeat: {  // compileRepeat compile.go:1484-1566
        int sub = x.kids[0];
        bool lazy = x.lazy;
        std::vector<std::function<F()>> parts;
        if (x.max == -1) {
          if (x.min == 0) return loop(sub, lazy, false);
          for (int k = 0; k < x.min; k++) parts.push_back([this, sub] { return frag(sub); });
          parts.push_back([this, sub, lazy] { return loop(sub, lazy, false); });
          return seq(parts);
        }
        if (x.min == x.max) {
          if (x.min == 0) { uint32_t e = eps(); return {e, e}; }
          for (int k = 0; k < x.min; k++) parts.push_back([this, sub] { return frag(sub); });
          return seq(parts);
        }
        for (int k = 0; k < x.min; k++) parts.push_back([this, sub] { return frag(sub); });
        for (int k = 0; k < x.max - x.min; k++) parts.push_back([this, sub, lazy] { return quest(sub, lazy); });
        return seq(parts);
      }
      case Node::Capture: {  // compileCapture compile.go:1654-1682: close is created before open
        if (x.kids.empty()) { uint32_t e = eps(); return {e, e}; }
        F f = frag(x.kids[0]);
        auto c = blank(CXG_NFA_CAPTURE); c.cap_index = static_cast<uint32_t>(x.cap); c.cap_start = 0;
        uint32_t close = push(c);
        link(f.out, close);
        auto o = blank(CXG_NFA_CAPTURE); o.cap_index = static_cast<uint32_t>(x.cap); o.cap_start = 1; o.next = f.in;
        uint32_t open = push(o);
        return {open, close};
      }
    }
    unsupported("unknown node");
  }
};

}  // namespace

HostNfa buildNfa(const Ast& ast) {
  NfaB b(ast);
  NfaB::F f = b.frag(ast.root);
  uint32_t m = b.push(NfaB::blank(CXG_NFA_MATCH));
  b.link(f.out, m);
  // unanchored prefix: Split(pattern, [00-FF] -> self), compile.go:1633-1650
  uint32_t any = b.byteRange(0x00, 0xFF);
  uint32_t sp = b.split(f.in, any);
  b.out.states[any].next = sp;
  b.out.startAnchored = f.in;
  b.out.startUnanchored = sp;
  b.out.captureCount = static_cast<uint32_t>(ast.ncap) + 1;
  return std::move(b.out);
}

// ====================================================================== strategy gate
namespace {

struct Lits { std::vector<PrefixLit> v; };

constexpr size_t kMaxLits = 256, kMaxLitLen = 64, kMaxClass = 10, kCross = 250;

std::vector<uint8_t> utf8(const std::vector<int32_t>& rs) {
  std::vector<uint8_t> o;
  for (int32_t r : rs) {
    if (r < 0x80) o.push_back(r);
    else if (r < 0x800) { o.push_back(0xC0 | (r >> 6)); o.push_back(0x80 | (r & 63)); }
    else if (r < 0x10000) { o.push_back(0xE0 | (r >> 12)); o.push_back(0x80 | ((r >> 6) & 63)); o.push_back(0x80 | (r & 63)); }
    else { o.push_back(0xF0 | (r >> 18)); o.push_back(0x80 | ((r >> 12) & 63)); o.push_back(0x80 | ((r >> 6) & 63)); o.push_back(0x80 | (r & 63)); }
  }
  return o;
}

struct LitX {  // literal/extractor.go prefix side, restricted to what the gate needs
  const Ast& a;
  explicit LitX(const Ast& ast) : a(ast) {}

  static void trim(Lits& l, size_t n) { for (auto& x : l.v) if (x.bytes.size() > n) { x.bytes.resize(n); x.exact = false; } }
  static void inexact(Lits& l) { for (auto& x : l.v) x.exact = false; }
  static void dedup(Lits& l) {
    std::set<std::vector<uint8_t>> seen;
    std::vector<PrefixLit> k;
    for (auto& x : l.v) if (seen.insert(x.bytes).second) k.push_back(x);
    l.v.swap(k);
  }
  // Case-insensitive literal -> all its case variants (expandCaseFoldLiteral extractor.go:838-941; the prefilters compare bytes).
  // Per rune the orbit of unicode.SimpleFold, starting at the rune itself (the parser stores the smallest of the orbit): `K` is
  // K, k, U+212A.  Orbits are collected while their product stays within the cross-product limit (250); with every rune reached
  // and the product within MaxLiterals (256) the result is every variant, complete; otherwise the longest prefix whose product
  // fits, incomplete.  The reference computes that prefix over an array whose tail the early exit left nil — length 0, which
  // resets the running product — so that a word of nine or more letters yields NO literal where 256 variants of its first eight
  // were meant; inside a concatenation the empty contribution is then skipped as if the word were not there, and literals of what
  // stands in front of it come out complete (`(?i)(get)errorfatal` would be searched as "get").  `quirk` is set when that
  // happens: such a program is refused rather than reproduced (selectStrategy below).
  bool quirk = false;
  Lits foldVariants(const std::vector<int32_t>& rs) {
    if (rs.empty()) return {};
    std::vector<std::vector<int32_t>> orbit(rs.size());
    uint64_t product = 1;
    size_t reached = 0;
    for (size_t k = 0; k < rs.size(); k++) {
      orbit[k].push_back(rs[k]);
      for (int32_t f = foldNext(rs[k]); f != rs[k]; f = foldNext(f)) orbit[k].push_back(f);
      reached = k + 1;
      product *= orbit[k].size();
      if (product > kCross) break;
    }
    size_t take = rs.size();
    bool all = product <= kMaxLits && reached == rs.size();
    if (!all) {
      uint64_t q = 1;
      for (size_t k = 0; k < orbit.size(); k++) { q *= orbit[k].size(); if (q > kMaxLits) { take = k; break; } }   // findMaxCaseFoldPrefix :920-929
      if (take == 0) return {};
      if (take > reached) { quirk = true; return {}; }                 // an orbit the early exit never filled: nothing is generated
    }
    std::vector<std::vector<int32_t>> vs{{}};
    for (size_t k = 0; k < take; k++) {
      std::vector<std::vector<int32_t>> nx;
      for (auto& pre : vs) for (int32_t r : orbit[k]) { nx.push_back(pre); nx.back().push_back(r); }
      vs.swap(nx);
    }
    Lits o;
    for (auto& v : vs) { auto b = utf8(v); if (b.size() > kMaxLitLen) b.resize(kMaxLitLen); o.v.push_back({b, all}); }
    if (!all) { dedup(o); if (o.v.size() > kMaxLits) o.v.resize(kMaxLits); }
    return o;
  }
  Lits expandClass(const Ast::N& x) {  // extractor.go:963-1000
    Lits o;
    size_t cnt = 0;
    for (size_t k = 0; k + 1 < x.r.size(); k += 2) { cnt += x.r[k + 1] - x.r[k] + 1; if (cnt > kMaxClass) return {}; }
    for (size_t k = 0; k + 1 < x.r.size(); k += 2)
      for (int32_t r = x.r[k]; r <= x.r[k + 1]; r++) { o.v.push_back({utf8({r}), true}); if (o.v.size() >= kMaxLits) return o; }
    return o;
  }
  Lits prefixes(int n, int depth) {  // extractor.go:158-215
    if (depth > 100) return {};
    const auto& x = a.at(n);
    switch (x.kind) {
      case Node::Lit: {
        if (x.fold) return foldVariants(x.r);
        auto b = utf8(x.r);
        if (b.size() > kMaxLitLen) b.resize(kMaxLitLen);
        Lits o; o.v.push_back({b, true}); return o;
      }
      case Node::Concat: return concat(x, depth);
      case Node::Alt: {  // extractor.go:217-262
        Lits o; bool over = false;
        for (int k : x.kids) {
          Lits s = prefixes(k, depth + 1);
          if (s.v.empty()) return {};
          for (auto& l : s.v) { o.v.push_back(l); if (o.v.size() > kCross) { over = true; break; } }
          if (over) break;
        }
        if (over || o.v.size() > kMaxLits) { trim(o, 3); inexact(o); dedup(o); if (o.v.size() > kMaxLits) o.v.resize(kMaxLits); }
        return o;
      }
      case Node::Class: return expandClass(x);
      case Node::Capture: return x.kids.empty() ? Lits{} : prefixes(x.kids[0], depth + 1);
      default: return {};
    }
  }
  bool contrib(int n, int depth, Lits& out) {  // concatSubContribution extractor.go:365-416
    const auto& x = a.at(n);
    switch (x.kind) {
      case Node::Lit: if (x.fold) { out = foldVariants(x.r); return true; } out = {}; out.v.push_back({utf8(x.r), true}); return true;   // (an empty contribution is not nil: the cross product skips it)
      case Node::Class: out = expandClass(x); return !out.v.empty();
      case Node::Alt: {  // expandAlternateContribution extractor.go:418-470
        Lits all; bool over = false;
        for (int k : x.kids) {
          Lits s = prefixes(k, depth + 1);
          if (s.v.empty()) return false;
          if (over) {
            for (auto& l : s.v) { auto b = l.bytes; if (b.size() > 3) b.resize(3); all.v.push_back({b, false}); }
            if (all.v.size() > kCross) dedup(all);
            continue;
          }
          for (auto& l : s.v) all.v.push_back(l);
          if (all.v.size() > kCross) { over = true; trim(all, 3); inexact(all); dedup(all); }
        }
        if (over || all.v.size() > kMaxLits) { trim(all, 3); inexact(all); dedup(all); if (all.v.size() > kMaxLits) all.v.resize(kMaxLits); }
        out = all;
        return true;
      }
      case Node::Capture: return !x.kids.empty() && contrib(x.kids[0], depth, out);
      case Node::Repeat:
        if (x.min >= 1 && contrib(x.kids[0], depth, out)) { inexact(out); return true; }
        return false;
      case Node::WordB: case Node::NoWordB: out = {}; out.v.push_back({{}, true}); return true;
      default: return false;
    }
  }
  Lits suffixes(int n, int depth) {  // extractSuffixes extractor.go:575-700
    if (depth > 100) return {};
    const auto& x = a.at(n);
    switch (x.kind) {
      case Node::Lit: {
        if (x.fold) return foldVariants(x.r);
        auto b = utf8(x.r);
        if (b.size() > kMaxLitLen) b.erase(b.begin(), b.end() - kMaxLitLen);
        Lits o; o.v.push_back({b, true}); return o;
      }
      case Node::Concat: {
        int last = static_cast<int>(x.kids.size()) - 1;
        while (last >= 0) {
          Node k = a.at(x.kids[last]).kind;
          if (k != Node::EndLine && k != Node::EndText && k != Node::WordB && k != Node::NoWordB) break;
          last--;
        }
        if (last < 0) return {};
        Lits suf = suffixes(x.kids[last], depth + 1);
        if (suf.v.empty()) return {};
        for (int i = last - 1; i >= 0; i--) {
          const auto& y = a.at(x.kids[i]);
          if (y.kind == Node::WordB || y.kind == Node::NoWordB) continue;
          if (y.kind != Node::Lit) { inexact(suf); return suf; }
          auto pre = utf8(y.r);
          for (auto& l : suf.v) {
            std::vector<uint8_t> nb = pre;
            nb.insert(nb.end(), l.bytes.begin(), l.bytes.end());
            if (nb.size() > kMaxLitLen) nb.erase(nb.begin(), nb.end() - kMaxLitLen);
            l.bytes.swap(nb);
          }
          if (suf.v.size() > kMaxLits) return suf;
        }
        return suf;
      }
      case Node::Alt: {
        Lits all;
        for (int k : x.kids) {
          Lits s2 = suffixes(k, depth + 1);
          if (s2.v.empty()) return {};
          for (auto& l : s2.v) { all.v.push_back(l); if (all.v.size() >= kMaxLits) return all; }
        }
        return all;
      }
      case Node::Class: return expandClass(x);
      case Node::Capture: return x.kids.empty() ? Lits{} : suffixes(x.kids[0], depth + 1);
      default: return {};
    }
  }
  Lits inner(int n, int depth) {  // extractInner extractor.go:744-810
    if (depth > 100) return {};
    const auto& x = a.at(n);
    switch (x.kind) {
      case Node::Lit: {
        if (x.fold) { Lits o = foldVariants(x.r); inexact(o); return o; }
        auto b = utf8(x.r);
        if (b.size() > kMaxLitLen) b.resize(kMaxLitLen);
        Lits o; o.v.push_back({b, false}); return o;
      }
      case Node::Concat: for (int k : x.kids) { Lits s2 = inner(k, depth + 1); if (!s2.v.empty()) return s2; } return {};
      case Node::Alt: {
        Lits all;
        for (int k : x.kids) {
          Lits s2 = inner(k, depth + 1);
          if (s2.v.empty()) return {};
          for (auto& l : s2.v) { all.v.push_back(l); if (all.v.size() >= kMaxLits) return all; }
        }
        return all;
      }
      case Node::Class: return expandClass(x);
      case Node::Capture: return x.kids.empty() ? Lits{} : inner(x.kids[0], depth + 1);
      default: return {};
    }
  }
  Lits concat(const Ast::N& x, int depth) {  // extractPrefixesConcat extractor.go:302-363
    size_t st = 0;
    while (st < x.kids.size() && (a.at(x.kids[st]).kind == Node::BeginLine || a.at(x.kids[st]).kind == Node::BeginText)) st++;
    if (st >= x.kids.size()) return {};
    Lits acc; acc.v.push_back({{}, true});
    for (size_t k = st; k < x.kids.size(); k++) {
      bool any = false;
      for (auto& l : acc.v) if (l.exact) { any = true; break; }
      if (!any) break;
      Lits c;
      if (!contrib(x.kids[k], depth, c)) { inexact(acc); break; }
      if (!acc.v.empty() && !c.v.empty()) {  // Seq.CrossForward seq.go:433-468
        std::vector<PrefixLit> r;
        for (auto& l : acc.v) {
          if (!l.exact) { r.push_back(l); continue; }
          for (auto& m : c.v) { PrefixLit n{l.bytes, m.exact}; n.bytes.insert(n.bytes.end(), m.bytes.begin(), m.bytes.end()); r.push_back(std::move(n)); }
        }
        acc.v.swap(r);
      }
      if (acc.v.size() > kCross || acc.v.size() > kMaxLits) { trim(acc, 4); inexact(acc); dedup(acc); if (acc.v.size() > kMaxLits) acc.v.resize(kMaxLits); break; }
      for (auto& l : acc.v) if (l.bytes.size() > kMaxLitLen) { l.bytes.resize(kMaxLitLen); l.exact = false; }
    }
    if (acc.v.size() == 1 && acc.v[0].bytes.empty()) return {};
    return acc;
  }
};

struct Shape {  // AST predicates of meta/strategy.go
  const Ast& a;
  bool has(int n, std::initializer_list<Node> ks) const {
    for (Node k : ks) if (a.at(n).kind == k) return true;
    for (int c : a.at(n).kids) if (has(c, ks)) return true;
    return false;
  }
  bool lazyAny(int n) const {
    const auto& x = a.at(n);
    if ((x.kind == Node::Star || x.kind == Node::Plus || x.kind == Node::Quest || x.kind == Node::Repeat) && x.lazy) return true;
    for (int c : x.kids) if (lazyAny(c)) return true;
    return false;
  }
  bool foldAny(int n) const {
    if (a.at(n).kind == Node::Lit && a.at(n).fold) return true;
    for (int c : a.at(n).kids) if (foldAny(c)) return true;
    return false;
  }
  static bool digitOnly(const std::vector<int32_t>& r) {  // strategy.go:311-324
    if (r.empty()) return false;
    for (size_t k = 0; k + 1 < r.size(); k += 2) if (r[k] < '0' || r[k + 1] > '9') return false;
    return true;
  }
  bool digitLead(int n) const {  // isDigitLeadPattern strategy.go:416-495, isDigitLeadConcat :331-385
    const auto& x = a.at(n);
    switch (x.kind) {
      case Node::Class: return digitOnly(x.r);
      case Node::Lit: return !x.r.empty() && x.r[0] >= '0' && x.r[0] <= '9';
      case Node::Alt: if (x.kids.empty()) return false; for (int c : x.kids) if (!digitLead(c)) return false; return true;
      case Node::Concat:
        for (int c : x.kids) {
          const auto& y = a.at(c);
          bool optional = y.kind == Node::Quest || y.kind == Node::Star || (y.kind == Node::Repeat && y.min == 0);
          if (!optional) return digitLead(c);
          const auto& in = a.at(y.kids[0]);
          bool ok;
          if (in.kind == Node::Class) ok = digitOnly(in.r);
          else if (in.kind == Node::Lit) { ok = !in.r.empty(); for (int32_t r : in.r) if (r < '0' || r > '9') ok = false; }
          else ok = digitLead(y.kids[0]);
          if (!ok) return false;
        }
        return false;
      case Node::Capture: case Node::Plus: return !x.kids.empty() && digitLead(x.kids[0]);
      case Node::Repeat: return x.min >= 1 && digitLead(x.kids[0]);
      default: return false;
    }
  }
  bool runSkipSafe(int n) const {  // isDigitRunSkipSafe strategy.go:530-560
    const auto& x = a.at(n);
    switch (x.kind) {
      case Node::Concat: case Node::Capture: return !x.kids.empty() && runSkipSafe(x.kids[0]);
      case Node::Plus: case Node::Star: return a.at(x.kids[0]).kind == Node::Class && digitOnly(a.at(x.kids[0]).r);
      case Node::Repeat: return x.max == -1 && a.at(x.kids[0]).kind == Node::Class && digitOnly(a.at(x.kids[0]).r);
      default: return false;
    }
  }
  bool classPlus(int n, uint8_t member[256]) const {  // ExtractCharClassRanges charclass_extract.go:19-71
    const auto& x = a.at(n);
    if (x.kind != Node::Plus) return false;
    const auto& c = a.at(x.kids[0]);
    if (c.kind != Node::Class || c.r.empty()) return false;
    for (int32_t r : c.r) if (r > 127) return false;
    std::memset(member, 0, 256);
    for (size_t k = 0; k + 1 < c.r.size(); k += 2) for (int32_t b = c.r[k]; b <= c.r[k + 1]; b++) member[b] = 1;
    return true;
  }
  bool compositePart(int n) const {  // nfa/composite.go:275-302
    const auto& x = a.at(n);
    if (x.kind == Node::Class) return true;
    if (x.kind == Node::Plus || x.kind == Node::Star || x.kind == Node::Quest || x.kind == Node::Repeat) return a.at(x.kids[0]).kind == Node::Class;
    return false;
  }
  bool simpleClass(int n) const {  // isSimpleCharClass strategy.go:1095-1128
    const auto& x = a.at(n);
    switch (x.kind) {
      case Node::Class: return true;
      case Node::Plus: case Node::Star: case Node::Quest: case Node::Repeat: case Node::Capture: return x.kids.size() == 1 && simpleClass(x.kids[0]);
      case Node::Concat: for (int c : x.kids) if (!simpleClass(c)) return false; return true;
      default: return false;
    }
  }
  bool wildcardOrRep(int n) const {  // extractor.go:1208-1240
    const auto& x = a.at(n);
    switch (x.kind) {
      case Node::Star: case Node::Plus: case Node::Quest: case Node::Repeat: case Node::Any: case Node::AnyNotNL: return true;
      case Node::Concat: case Node::Alt: for (int c : x.kids) if (wildcardOrRep(c)) return true; return false;
      case Node::Capture: return !x.kids.empty() && wildcardOrRep(x.kids[0]);
      default: return false;
    }
  }
  // true when extractInner / extractSuffixes could yield a literal for this element (extractor.go:575-810)
  bool yieldsLiteral(int n) const {
    const auto& x = a.at(n);
    switch (x.kind) {
      case Node::Lit: return !x.r.empty();
      case Node::Class: { size_t cnt = 0; for (size_t k = 0; k + 1 < x.r.size(); k += 2) cnt += x.r[k + 1] - x.r[k] + 1; return cnt <= kMaxClass; }
      case Node::Capture: return !x.kids.empty() && yieldsLiteral(x.kids[0]);
      case Node::Concat: for (int c : x.kids) if (yieldsLiteral(c)) return true; return false;
      case Node::Alt: for (int c : x.kids) if (!yieldsLiteral(c)) return false; return true;
      default: return false;
    }
  }
};

}  // namespace

namespace { Plan selectStrategyOf(const Ast& ast, const HostNfa& nfa); }

Plan selectStrategy(const Ast& ast, const HostNfa& nfa) {
  Plan p = selectStrategyOf(ast, nfa);
  // (see LitX::foldVariants) an empty expansion is harmless where it means "no literal" — the pattern itself, a branch of an
  // alternation — and harmful where the cross product of a concatenation skips it: directly in a concatenation, through groups
  // and through {n,m} with n >= 1 (concatSubContribution extractor.go:378-418)
  LitX lx(ast);
  bool skipped = false;
  std::function<void(int, bool)> visit = [&](int n, bool inConcat) {
    const auto& x = ast.at(n);
    if (x.kind == Node::Lit && x.fold) { lx.quirk = false; lx.foldVariants(x.r); skipped = skipped || (lx.quirk && inConcat); return; }
    const bool pass = x.kind == Node::Concat || ((x.kind == Node::Capture || (x.kind == Node::Repeat && x.min >= 1)) && inConcat);
    for (int c : x.kids) visit(c, pass);
  };
  if (ast.root >= 0) visit(ast.root, false);
  if (skipped) {
    p.confident = false;
    p.why = "a case-insensitive literal of nine or more letters: the reference's literal extraction yields nothing for it and goes on as if it were not there (literal/extractor.go:866-887), which is not reproduced";
  }
  return p;
}

namespace {
Plan selectStrategyOf(const Ast& ast, const HostNfa& nfa) {
  Plan p;
  Shape sh{ast};
  int root = ast.root;
  // The NFA builder already rejected line / text anchors, '.', non-ASCII classes; (?i) literals survive it.
  // Word boundaries (\b \B) are the one kind of look-around that reaches here: hasWordBoundary strategy.go, and they
  // count as anchor assertions / non-line anchors in the rules below.
  const bool wordB = sh.has(root, {Node::WordB, Node::NoWordB});
  // Multi-line anchors (?m)^ (?m)$: hasMultilineLineAnchor strategy.go:1197-1210.  (?m)$ is a "non-line anchor" for the
  // literal engines (hasNonLineAnchors compile.go:686-703), (?m)^ is not: Teddy keeps its complete literals behind a
  // line-start check (prefilter.WrapLineAnchor, compile.go:670-677).  (The multi-line reverse-suffix strategy: below.)
  const bool lineAnchor = sh.has(root, {Node::BeginLine, Node::EndLine});
  const bool nonLineAnchors = wordB || sh.has(root, {Node::EndLine, Node::BeginText, Node::EndText});   // hasNonLineAnchors compile.go:686-703
  p.lineStart = sh.has(root, {Node::BeginLine});
  {
    // Does every match begin at a line start?  (Used for UseTeddy below: the reference wraps the WHOLE literal prefilter
    // in the line-start check as soon as the pattern holds a (?m)^ anywhere — `(?m)^foo|barr` then loses "barr" in mid-line,
    // compile.go:670-677 — which equals the pattern's meaning only when all alternatives are anchored.)
    std::function<bool(int)> needs = [&](int n) -> bool {
      const auto& x = ast.at(n);
      switch (x.kind) {
        case Node::BeginLine: return true;
        case Node::Capture: case Node::Plus: return !x.kids.empty() && needs(x.kids[0]);
        case Node::Repeat: return x.min >= 1 && !x.kids.empty() && needs(x.kids[0]);
        case Node::Concat: return !x.kids.empty() && needs(x.kids[0]);
        case Node::Alt: if (x.kids.empty()) return false; for (int c : x.kids) if (!needs(c)) return false; return true;
        default: return false;
      }
    };
    p.lineStartAll = needs(root);
  }
  LitX lx(ast);
  Lits pre = lx.prefixes(root, 0);  // ExtractPrefixes extractor.go:128-156 (trim cascade for > 64 literals)
  if (pre.v.size() > 64) {
    Lits orig = pre;
    for (size_t keep : {4u, 3u, 2u}) { if (pre.v.size() <= 64) break; LitX::trim(pre, keep); LitX::dedup(pre); }
    if (pre.v.size() > 64) pre = orig;
  }
  p.prefixes = pre.v;
  auto lcpLen = [&]() {
    if (pre.v.empty()) return size_t{0};
    size_t n = pre.v[0].bytes.size();
    for (auto& l : pre.v) { size_t k = 0; while (k < n && k < l.bytes.size() && l.bytes[k] == pre.v[0].bytes[k]) k++; n = k; }
    return n;
  };
  size_t minLen = SIZE_MAX;
  bool allExact = !pre.v.empty();
  for (auto& l : pre.v) { minLen = std::min(minLen, l.bytes.size()); allExact = allExact && l.exact; }
  bool good = lcpLen() >= 1;
  bool teddyLits = pre.v.size() >= 2 && pre.v.size() <= 64 && minLen >= 3;
  bool acLits = pre.v.size() > 64 && minLen >= 1;

  // selectReverseStrategy (strategy.go:974-1093).  The reverse searchers themselves are outside the device subset: when
  // one would be chosen the plan carries that strategy and the program is refused.
  bool fastPrefix = !pre.v.empty() && (good || pre.v.size() == 1 || minLen >= 3);   // hasFastPrefixPrefilter :948-967
  const auto& r = ast.at(root);
  auto lcSuffixLen = [](const Lits& l) {
    if (l.v.empty()) return size_t{0};
    size_t n = l.v[0].bytes.size();
    for (auto& x : l.v) { size_t k = 0; const auto& b0 = l.v[0].bytes; const auto& b = x.bytes; while (k < n && k < b.size() && b0[b0.size() - 1 - k] == b[b.size() - 1 - k]) k++; n = k; }
    return n;
  };
  auto lcPrefixLen = [](const Lits& l) {
    if (l.v.empty()) return size_t{0};
    size_t n = l.v[0].bytes.size();
    for (auto& x : l.v) { size_t k = 0; while (k < n && k < x.bytes.size() && x.bytes[k] == l.v[0].bytes[k]) k++; n = k; }
    return n;
  };
  auto dotLoop = [&](int n) {       // `.*` / `.+`
    const auto& y = ast.at(n);
    return (y.kind == Node::Star || y.kind == Node::Plus) && !y.kids.empty() && (ast.at(y.kids[0]).kind == Node::Any || ast.at(y.kids[0]).kind == Node::AnyNotNL);
  };
  auto safeSuffix = [&](int n0) {   // isSafeForReverseSuffix :605-634
    int n = n0;
    while (ast.at(n).kind == Node::Capture && !ast.at(n).kids.empty()) n = ast.at(n).kids[0];
    const auto& x = ast.at(n);
    if (x.kind != Node::Concat || x.kids.size() < 2) return false;
    int wc = 0;
    for (size_t k = 0; k + 1 < x.kids.size(); k++) {
      int c = x.kids[k];
      while (ast.at(c).kind == Node::Capture && !ast.at(c).kids.empty()) c = ast.at(c).kids[0];
      const auto& y = ast.at(c);
      if (dotLoop(c) || (y.kind == Node::Plus && ast.at(y.kids[0]).kind == Node::Class) || (y.kind == Node::Repeat && y.min >= 1)) wc++;   // isWildcardSubexpression :586-603
    }
    if (wc == 0) return false;
    for (size_t k = 1; k + 1 < x.kids.size(); k++)            // containsAnchor in a middle element (:628-632)
      if (sh.has(x.kids[k], {Node::BeginLine, Node::EndLine, Node::BeginText, Node::EndText})) return false;
    return true;
  };
  auto safeInner = [&](int n0) {    // isSafeForReverseInner :874-907
    int n = n0;
    while (ast.at(n).kind == Node::Capture && !ast.at(n).kids.empty()) n = ast.at(n).kids[0];
    const auto& x = ast.at(n);
    if (x.kind != Node::Concat || x.kids.size() < 2) return false;
    const auto& f = ast.at(x.kids[0]);
    return dotLoop(x.kids[0]) || (f.kind == Node::Plus && ast.at(f.kids[0]).kind == Node::Class);
  };
  // UseMultilineReverseSuffix (strategy.go:1004-1012, decided before the fast-prefix test): `(?m)^.*suffix` — a line-start anchor
  // in front (isSafeForMultilineReverseSuffix :805-843), a wildcard somewhere, and suffix literals with a common suffix
  {
    std::function<bool(int)> lineStart = [&](int n) -> bool {      // containsLineStartAnchor :734-768
      const auto& x = ast.at(n);
      switch (x.kind) {
        case Node::BeginLine: return true;
        case Node::Concat: for (int c : x.kids) if (lineStart(c)) return true; return false;
        case Node::Alt: if (x.kids.empty()) return false; for (int c : x.kids) if (!lineStart(c)) return false; return true;
        case Node::Capture: return !x.kids.empty() && lineStart(x.kids[0]);
        default: return false;
      }
    };
    std::function<bool(int)> wild = [&](int n) -> bool {           // containsWildcard :771-793
      const auto& x = ast.at(n);
      switch (x.kind) {
        case Node::Star: case Node::Plus: return dotLoop(n);
        case Node::Concat: case Node::Alt: for (int c : x.kids) if (wild(c)) return true; return false;
        case Node::Capture: case Node::Quest: case Node::Repeat: return !x.kids.empty() && wild(x.kids[0]);
        default: return false;
      }
    };
    std::function<bool(int)> safeMultiline = [&](int n) -> bool {
      if (!(lineStart(n) && wild(n))) return false;
      const auto& x = ast.at(n);
      if (x.kind == Node::Capture) return !x.kids.empty() && safeMultiline(x.kids[0]);
      if (x.kind != Node::Concat || x.kids.size() < 2) return false;
      bool w = false;
      for (size_t k = 0; k < x.kids.size(); k++) {
        if (k == 0 && ast.at(x.kids[k]).kind == Node::BeginLine) continue;
        const auto& y = ast.at(x.kids[k]);
        if (dotLoop(x.kids[k]) || (y.kind == Node::Plus && !y.kids.empty() && ast.at(y.kids[0]).kind == Node::Class)) w = true;   // isWildcardOp :846-861
      }
      return ast.at(x.kids[0]).kind == Node::BeginLine && w;
    };
    if (!wordB && !sh.has(root, {Node::EndLine}) && safeMultiline(root)) {   // (after the word-boundary and end-anchor returns of :980-995)
      Lits suf = lx.suffixes(root, 0);
      if (!suf.v.empty() && lcSuffixLen(suf) >= 1) { p.strategy = CXG_USE_MULTILINE_REVERSE_SUFFIX; return p; }
    }
  }
  // selectReverseStrategy returns at once for word boundaries (:988) and for an end anchor that does not end the pattern
  // in the sense of nfa.isEndAnchored — which (?m)$ never does (nfa.HasImpossibleEndAnchor, nfa/compile.go:1858-1888)
  std::function<bool(int)> endsWithTextAnchor = [&](int n) -> bool {   // nfa.isEndAnchored, nfa/compile.go:1798-1824 (as in textAnchorStrategy below)
    const auto& x = ast.at(n);
    switch (x.kind) {
      case Node::EndText: return true;
      case Node::Concat: return !x.kids.empty() && endsWithTextAnchor(x.kids.back());
      case Node::Capture: return !x.kids.empty() && endsWithTextAnchor(x.kids[0]);
      case Node::Alt: if (x.kids.empty()) return false; for (int c : x.kids) if (!endsWithTextAnchor(c)) return false; return true;
      default: return false;
    }
  };
  // (round 6: `$[a-z0-9]+:-`, `(\s|$)abc[0-4]+ ` — a \z / $ that does not end the pattern is an impossible end anchor too)
  const bool impossibleEnd = sh.has(root, {Node::EndLine}) || (sh.has(root, {Node::EndText}) && !endsWithTextAnchor(root));
  if (!fastPrefix && !wordB && !impossibleEnd) {
    int reverse = 0;
    bool decided = false;
    Lits suf = lx.suffixes(root, 0);
    if (!suf.v.empty() && lcSuffixLen(suf) >= 1) {
      decided = true;                                         // strategy.go:1041-1051: either way the search for a
      if (safeSuffix(root)) reverse = CXG_USE_REVERSE_SUFFIX; // reverse strategy ends here
    }
    if (!decided && safeSuffix(root) && !suf.v.empty()) {     // shouldUseReverseSuffixSet :909-935
      bool exactAlt = !pre.v.empty() && allExact && pre.v.size() == suf.v.size();
      size_t ms = SIZE_MAX;
      for (auto& l : suf.v) ms = std::min(ms, l.bytes.size());
      if (!exactAlt && suf.v.size() >= 2 && suf.v.size() <= 32 && ms >= 2) { reverse = CXG_USE_REVERSE_SUFFIX_SET; decided = true; }
    }
    if (!decided && r.kind == Node::Concat && r.kids.size() >= 3) {   // ExtractInnerForReverseSearch extractor.go:1061-1100
      for (size_t k = 1; k + 1 < r.kids.size(); k++) {
        Lits in = lx.inner(r.kids[k], 0);
        if (in.v.empty()) continue;
        bool before = false, after = false;
        for (size_t j = 0; j < k; j++) before = before || sh.wildcardOrRep(r.kids[j]);
        for (size_t j = k + 1; j < r.kids.size(); j++) after = after || sh.wildcardOrRep(r.kids[j]);
        if (before && after) {
          const size_t l = lcPrefixLen(in);
          if (l == 1 && sh.digitLead(root)) break;            // Issue #75: DigitPrefilter wins
          if (l >= 1 && safeInner(root)) reverse = CXG_USE_REVERSE_INNER;
          break;
        }
      }
    }
    if (reverse) { p.strategy = reverse; return p; }
  }

  int nfaSize = static_cast<int>(nfa.states.size());
  if (!good && !teddyLits && sh.classPlus(root, p.membership)) { p.strategy = CXG_USE_CHARCLASS_SEARCHER; return p; }
  if (!good && !teddyLits && r.kind == Node::Concat && r.kids.size() >= 2) {
    bool comp = true;
    for (int c : r.kids) comp = comp && sh.compositePart(c);
    if (comp) { p.strategy = CXG_USE_COMPOSITE_SEARCHER; return p; }
  }
  if (!good && !teddyLits && sh.simpleClass(root)) { p.strategy = CXG_USE_BOUNDED_BACKTRACKER; return p; }
  if (teddyLits && allExact && !nonLineAnchors) { p.strategy = CXG_USE_TEDDY; return p; }   // selectLiteralStrategy :1143-1170 (no non-line anchors)
  if (acLits && allExact) { p.strategy = CXG_USE_AHO_CORASICK; return p; }
  if (nfaSize <= 100 && sh.digitLead(root)) {                              // shouldUseDigitPrefilter :511-523
    p.strategy = CXG_USE_DIGIT_PREFILTER;
    if (sh.runSkipSafe(root)) p.flags |= CXG_FLAG_DIGIT_RUN_SKIP_SAFE;
    return p;
  }
  NfaB probe(ast);
  bool nullablePattern = probe.nullable(root);
  if (nfaSize < 20) {
    if (nullablePattern || wordB || lineAnchor) { p.strategy = CXG_USE_NFA; return p; }   // hasWordBoundaryAnchorCombo || canMatchEmpty || hasMultilineLineAnchor, :1503
    p.strategy = CXG_USE_DFA;
    if (!sh.lazyAny(root)) p.flags |= CXG_FLAG_HAS_REVERSE_DFA;             // buildReverseDFA compile.go:184-205
    return p;
  }
  if (!good && !teddyLits && nullablePattern) { p.strategy = CXG_USE_NFA; return p; }
  if (good || teddyLits) {
    if (nfaSize > 200 && !allExact) { p.strategy = CXG_USE_NFA; return p; }
    p.strategy = CXG_USE_DFA;
    if (!sh.lazyAny(root)) p.flags |= CXG_FLAG_HAS_REVERSE_DFA;
    return p;
  }
  if (nfaSize > 100) { p.strategy = CXG_USE_NFA; return p; }
  p.strategy = CXG_USE_BOTH;
  {  // selectPrefilter (prefilter/prefilter.go:261-297) over the prefix literals: one literal, or 2+ literals of >= 3 bytes each
    size_t minLen = ~size_t(0);
    for (auto& l : p.prefixes) minLen = std::min(minLen, l.bytes.size());
    if (!p.prefixes.empty() && (p.prefixes.size() == 1 || minLen >= 3)) p.flags |= CXG_FLAG_HAS_PREFILTER;   // one literal of any length (an empty needle is found at once: the PikeVM runs from `at`)
  }
  return p;
}


}  // namespace

int textAnchorStrategy(const Ast& ast, bool& exact) {
  exact = true;
  if (ast.root < 0) return -1;
  std::function<bool(int)> startsWithText = [&](int n) -> bool {     // Compiler.isPatternAnchored nfa/compile.go:1755-1769
    const auto& x = ast.at(n);
    if (x.kind == Node::BeginText) return true;
    if ((x.kind == Node::Concat || x.kind == Node::Capture) && !x.kids.empty()) return startsWithText(x.kids[0]);
    return false;
  };
  std::function<bool(int)> endsWithText = [&](int n) -> bool {       // isEndAnchored :1798-1824
    const auto& x = ast.at(n);
    switch (x.kind) {
      case Node::EndText: return true;
      case Node::Concat: return !x.kids.empty() && endsWithText(x.kids.back());
      case Node::Capture: return !x.kids.empty() && endsWithText(x.kids[0]);
      case Node::Alt: if (x.kids.empty()) return false; for (int c : x.kids) if (!endsWithText(c)) return false; return true;
      default: return false;
    }
  };
  std::function<bool(int)> anyEnd = [&](int n) -> bool {             // containsEndAnchor :1872-1888
    const auto& x = ast.at(n);
    switch (x.kind) {
      case Node::EndText: case Node::EndLine: return true;
      case Node::Concat: case Node::Alt: for (int c : x.kids) if (anyEnd(c)) return true; return false;
      case Node::Capture: case Node::Star: case Node::Plus: case Node::Quest: case Node::Repeat: return !x.kids.empty() && anyEnd(x.kids[0]);
      default: return false;
    }
  };
  std::function<bool(int)> innerEnd = [&](int n) -> bool {           // hasInternalEndAnchor :1828-1856
    const auto& x = ast.at(n);
    switch (x.kind) {
      case Node::Concat:
        for (size_t k = 0; k + 1 < x.kids.size(); k++) if (anyEnd(x.kids[k])) return true;
        return !x.kids.empty() && innerEnd(x.kids.back());
      case Node::Capture: return !x.kids.empty() && innerEnd(x.kids[0]);
      case Node::Alt: for (int c : x.kids) if (innerEnd(c)) return true; return false;
      default: return false;
    }
  };
  std::function<bool(int)> anyStart = [&](int n) -> bool {           // containsStartAnchor :1902-1926
    const auto& x = ast.at(n);
    switch (x.kind) {
      case Node::BeginText: case Node::BeginLine: return true;
      case Node::Concat: case Node::Alt: for (int c : x.kids) if (anyStart(c)) return true; return false;
      case Node::Capture: case Node::Star: case Node::Plus: case Node::Quest: case Node::Repeat: return !x.kids.empty() && anyStart(x.kids[0]);
      default: return false;
    }
  };
  const bool startAnchored = startsWithText(ast.root);
  const bool endAnchored = endsWithText(ast.root) && !innerEnd(ast.root);
  if (endAnchored && !startAnchored && !anyStart(ast.root)) return CXG_USE_REVERSE_ANCHORED;
  if (startAnchored) { exact = false; return CXG_USE_BOUNDED_BACKTRACKER; }
  return -1;
}

bool boundedSurrogate(const Ast& ast, Ast& out, std::vector<std::pair<int, int>>& bounds) {
  bounds.clear();
  out = Ast();
  if (ast.root < 0 || ast.ncap != 0) return false;
  auto singleByte = [&](int n) {                       // one byte of input, no case folding, ASCII
    const Ast::N& x = ast.at(n);
    if (x.kind == Node::Lit) return x.r.size() == 1 && !x.fold && x.r[0] >= 0 && x.r[0] < 128;
    if (x.kind != Node::Class || x.r.empty()) return false;
    for (int32_t r : x.r) if (r < 0 || r > 127) return false;
    return true;
  };
  // items of the top-level concatenation, `(?:...){k}` and `x{k}` unrolled
  std::vector<int> items;
  std::function<bool(int, int)> flatten = [&](int n, int depth) -> bool {
    const Ast::N& x = ast.at(n);
    if (depth > 4 || items.size() > 64) return false;
    if (x.kind == Node::Concat) { for (int k : x.kids) if (!flatten(k, depth + 1)) return false; return true; }
    if (x.kind == Node::Repeat && x.min == x.max && x.min >= 1 && x.min <= 16 && !x.lazy) {
      for (int c = 0; c < x.min; c++) if (!flatten(x.kids[0], depth + 1)) return false;
      return true;
    }
    items.push_back(n);
    return true;
  };
  if (!flatten(ast.root, 0) || items.empty()) return false;
  bool anyBounded = false;
  std::vector<int> outItems;
  auto copyLeaf = [&](int n) { const int id = out.add(ast.at(n).kind); out.at(id) = ast.at(n); out.at(id).kids.clear(); return id; };
  for (int n : items) {
    const Ast::N& x = ast.at(n);
    if (singleByte(n)) { outItems.push_back(copyLeaf(n)); continue; }
    const bool plus = x.kind == Node::Plus, rep = x.kind == Node::Repeat && x.min >= 1 && (x.max == -1 || x.max > x.min);
    if (!(plus || rep) || x.lazy || x.kids.size() != 1 || !singleByte(x.kids[0])) return false;
    const int kid = copyLeaf(x.kids[0]);
    const int run = out.add(Node::Plus);
    out.at(run).kids.push_back(kid);
    outItems.push_back(run);
    const int mx = plus || x.max == -1 ? 0 : x.max;
    if (mx > 255 || x.min > 255) return false;
    bounds.emplace_back(plus ? 1 : x.min, mx);
    anyBounded = anyBounded || mx != 0 || (rep && x.min > 1);
  }
  if (!anyBounded) return false;
  if (outItems.size() == 1) out.root = outItems[0];
  else { const int c = out.add(Node::Concat); out.at(c).kids = outItems; out.root = c; }
  out.ncap = 0;
  return true;
}

}  // namespace cxg
