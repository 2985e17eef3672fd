// ORACLE — TEST INFRASTRUCTURE ONLY (see meta.hpp header for the reference map).
#include "meta.hpp"

#include <algorithm>
#include <cstring>
#include <functional>
#include <set>

namespace orc {

const char* strategyName(int s) {
  static const char* n[] = {"UseNFA", "UseDFA", "UseBoth", "UseReverseAnchored", "UseReverseSuffix",
                            "UseOnePass", "UseReverseInner", "UseBoundedBacktracker", "UseTeddy",
                            "UseReverseSuffixSet", "UseCharClassSearcher", "UseCompositeSearcher",
                            "UseBranchDispatch", "UseDigitPrefilter", "UseAhoCorasick",
                            "UseAnchoredLiteral", "UseMultilineReverseSuffix"};
  return (s >= 0 && s < 17) ? n[s] : "?";
}

// ----------------------------------------------------------------- literal.Seq
bool Seq::allComplete() const {  // seq.go:206-216
  if (lits.empty()) return false;
  for (auto& l : lits) if (!l.complete) return false;
  return true;
}
std::vector<uint8_t> Seq::lcp() const {  // seq.go:343-364
  if (lits.empty()) return {};
  std::vector<uint8_t> p = lits[0].bytes;
  for (size_t i = 1; i < lits.size(); i++) {
    size_t k = 0;
    while (k < p.size() && k < lits[i].bytes.size() && p[k] == lits[i].bytes[k]) k++;
    p.resize(k);
    if (p.empty()) break;
  }
  return p;
}
std::vector<uint8_t> Seq::lcs() const {  // seq.go:394-430
  if (lits.empty()) return {};
  std::vector<uint8_t> s = lits[0].bytes;
  for (size_t i = 1; i < lits.size(); i++) {
    const auto& b = lits[i].bytes;
    size_t k = 0;
    while (k < s.size() && k < b.size() && s[s.size() - 1 - k] == b[b.size() - 1 - k]) k++;
    s.erase(s.begin(), s.end() - k);
    if (s.empty()) break;
  }
  return s;
}

namespace {

constexpr int kMaxLiterals = 256;   // meta.Config.MaxLiterals (meta/config.go:31-111)
constexpr int kMaxLiteralLen = 64, kMaxClassSize = 10, kCrossLimit = 250;

std::vector<uint8_t> runesToBytes(const std::vector<int>& runes) {
  std::vector<uint8_t> out;
  for (int r : runes) {
    if (r < 0x80) out.push_back(static_cast<uint8_t>(r));
    else if (r < 0x800) { out.push_back(0xC0 | (r >> 6)); out.push_back(0x80 | (r & 0x3F)); }
    else if (r < 0x10000) { out.push_back(0xE0 | (r >> 12)); out.push_back(0x80 | ((r >> 6) & 0x3F)); out.push_back(0x80 | (r & 0x3F)); }
    else { out.push_back(0xF0 | (r >> 18)); out.push_back(0x80 | ((r >> 12) & 0x3F)); out.push_back(0x80 | ((r >> 6) & 0x3F)); out.push_back(0x80 | (r & 0x3F)); }
  }
  return out;
}

void keepFirstBytes(Seq& s, size_t n) {  // seq.go:470-480
  for (auto& l : s.lits) if (l.bytes.size() > n) { l.bytes.resize(n); l.complete = false; }
}
void dedup(Seq& s) {  // seq.go:491-507
  std::set<std::vector<uint8_t>> seen;
  std::vector<Lit> kept;
  for (auto& l : s.lits) if (seen.insert(l.bytes).second) kept.push_back(l);
  s.lits.swap(kept);
}
void markInexact(Seq& s) { for (auto& l : s.lits) l.complete = false; }

int foldNextRune(int c) {  // unicode.SimpleFold for what the parser lets through under (?i): ASCII letters, with the K and S orbits
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

struct Extractor {
  // A case-insensitive literal contributes every case variant (the prefilters compare bytes), extractor.go:838-917.  The fold sets
  // are filled until their product passes the cross-product limit; all variants when the product fits MaxLiterals and every rune
  // was reached, else the longest prefix whose product fits, incomplete.  (As written in the reference: the fold sets behind the
  // early exit stay nil, their length 0 resets the running product, and a long word — nine letters with MaxLiterals 256 — ends
  // with NO literal at all: restated as is.)
  Seq expandCaseFoldLiteral(const std::vector<int>& runes) {
    if (runes.empty()) return {};
    std::vector<std::vector<int>> foldSets(runes.size());
    long long total = 1;
    size_t filled = 0;
    for (size_t i = 0; i < runes.size(); i++) {
      std::vector<int> folds{runes[i]};                             // caseFolds :933-941
      for (int f = foldNextRune(runes[i]); f != runes[i]; f = foldNextRune(f)) folds.push_back(f);
      foldSets[i] = folds;
      filled = i + 1;
      total *= static_cast<long long>(folds.size());
      if (total > kCrossLimit) break;
    }
    auto generate = [&](size_t n) {                                 // generateCaseFoldVariants :891-917
      std::vector<std::vector<int>> variants{{}};
      for (size_t i = 0; i < n; i++) {
        std::vector<std::vector<int>> next;
        for (auto& pre : variants)
          for (int r : foldSets[i]) { auto e = pre; e.push_back(r); next.push_back(std::move(e)); }
        variants.swap(next);
      }
      Seq out;
      for (auto& v : variants) {
        auto b = runesToBytes(v);
        if (b.size() > kMaxLiteralLen) b.resize(kMaxLiteralLen);
        out.lits.push_back({b, true});
      }
      return out;
    };
    if (total <= kMaxLiterals && filled == runes.size()) return generate(filled);
    size_t trim = foldSets.size();                                  // findMaxCaseFoldPrefix :920-929
    {
      long long product = 1;
      for (size_t i = 0; i < foldSets.size(); i++) {
        product *= static_cast<long long>(foldSets[i].size());
        if (product > kMaxLiterals) { trim = i; break; }
      }
    }
    if (trim == 0) return {};
    Seq out = generate(trim);
    markInexact(out);
    dedup(out);
    if (static_cast<int>(out.lits.size()) > kMaxLiterals) out.lits.resize(kMaxLiterals);
    return out;
  }

  Seq expandCharClass(const ReP& re) {  // extractor.go:963-1000
    Seq out;
    int count = 0;
    for (size_t i = 0; i + 1 < re->rune.size(); i += 2) {
      count += re->rune[i + 1] - re->rune[i] + 1;
      if (count > kMaxClassSize) return Seq{};
    }
    for (size_t i = 0; i + 1 < re->rune.size(); i += 2)
      for (int r = re->rune[i]; r <= re->rune[i + 1]; r++) {
        out.lits.push_back({runesToBytes({r}), true});
        if (static_cast<int>(out.lits.size()) >= kMaxLiterals) return out;
      }
    return out;
  }

  Seq prefixes(const ReP& re, int depth) {  // extractor.go:158-215
    if (depth > 100) return {};
    switch (re->op) {
      case OpLiteral: {
        if (re->flags & FoldCase) return expandCaseFoldLiteral(re->rune);   // :167-169
        auto b = runesToBytes(re->rune);
        if (b.size() > kMaxLiteralLen) b.resize(kMaxLiteralLen);
        Seq s; s.lits.push_back({b, true}); return s;
      }
      case OpConcat: return prefixesConcat(re, depth);
      case OpAlternate: return prefixesAlternate(re, depth);
      case OpCharClass: return expandCharClass(re);
      case OpCapture: return re->sub.empty() ? Seq{} : prefixes(re->sub[0], depth + 1);
      default: return {};
    }
  }

  Seq prefixesAlternate(const ReP& re, int depth) {  // extractor.go:217-262
    Seq result;
    bool overflowed = false;
    for (auto& sub : re->sub) {
      Seq s = prefixes(sub, depth + 1);
      if (s.empty()) return {};
      for (auto& l : s.lits) {
        result.lits.push_back(l);
        if (static_cast<int>(result.lits.size()) > kCrossLimit) { overflowed = true; break; }
      }
      if (overflowed) break;
    }
    if (overflowed || static_cast<int>(result.lits.size()) > kMaxLiterals) {
      keepFirstBytes(result, 3);
      markInexact(result);
      dedup(result);
      if (static_cast<int>(result.lits.size()) > kMaxLiterals) result.lits.resize(kMaxLiterals);
    }
    return result;
  }

  bool contribution(const ReP& sub, int depth, Seq& out) {  // extractor.go:365-416; false == nil
    switch (sub->op) {
      case OpLiteral:
        if (sub->flags & FoldCase) { out = expandCaseFoldLiteral(sub->rune); return true; }   // :380-382 (an empty Seq, not nil)
        out = Seq{}; out.lits.push_back({runesToBytes(sub->rune), true}); return true;
      case OpCharClass: {
        out = expandCharClass(sub);
        return !out.empty();
      }
      case OpAlternate: {  // expandAlternateContribution extractor.go:418-470
        Seq all; bool overflowed = false;
        for (auto& s : sub->sub) {
          Seq seq = prefixes(s, depth + 1);
          if (seq.empty()) return false;
          if (overflowed) {
            for (auto& l : seq.lits) {
              auto b = l.bytes; if (b.size() > 3) b.resize(3);
              all.lits.push_back({b, false});
            }
            if (static_cast<int>(all.lits.size()) > kCrossLimit) dedup(all);
            continue;
          }
          for (auto& l : seq.lits) all.lits.push_back(l);
          if (static_cast<int>(all.lits.size()) > kCrossLimit) {
            overflowed = true;
            keepFirstBytes(all, 3); markInexact(all); dedup(all);
          }
        }
        if (overflowed || static_cast<int>(all.lits.size()) > kMaxLiterals) {
          keepFirstBytes(all, 3); markInexact(all); dedup(all);
          if (static_cast<int>(all.lits.size()) > kMaxLiterals) all.lits.resize(kMaxLiterals);
        }
        out = all;
        return true;
      }
      case OpCapture:
        if (sub->sub.empty()) return false;
        return contribution(sub->sub[0], depth, out);
      case OpRepeat:
        if (sub->min >= 1 && !sub->sub.empty()) {
          if (!contribution(sub->sub[0], depth, out)) return false;
          markInexact(out);
          return true;
        }
        return false;
      case OpWordBoundary: case OpNoWordBoundary:
        out = Seq{}; out.lits.push_back({{}, true}); return true;
      default: return false;
    }
  }

  Seq prefixesConcat(const ReP& re, int depth) {  // extractor.go:302-363
    if (re->sub.empty()) return {};
    size_t start = 0;
    while (start < re->sub.size() && (re->sub[start]->op == OpBeginLine || re->sub[start]->op == OpBeginText)) start++;
    if (start >= re->sub.size()) return {};
    Seq acc; acc.lits.push_back({{}, true});
    for (size_t i = start; i < re->sub.size(); i++) {
      bool anyExact = false;
      for (auto& l : acc.lits) if (l.complete) { anyExact = true; break; }
      if (!anyExact) break;
      Seq c;
      if (!contribution(re->sub[i], depth, c)) { markInexact(acc); break; }
      // CrossForward seq.go:433-468
      if (!acc.empty() && !c.empty()) {
        std::vector<Lit> res;
        for (auto& left : acc.lits) {
          if (!left.complete) { res.push_back(left); continue; }
          for (auto& right : c.lits) {
            Lit n{left.bytes, right.complete};
            n.bytes.insert(n.bytes.end(), right.bytes.begin(), right.bytes.end());
            res.push_back(std::move(n));
          }
        }
        acc.lits.swap(res);
      }
      if (static_cast<int>(acc.lits.size()) > kCrossLimit || static_cast<int>(acc.lits.size()) > kMaxLiterals) {
        keepFirstBytes(acc, 4); markInexact(acc); dedup(acc);   // handleCrossProductOverflow :541-551
        if (static_cast<int>(acc.lits.size()) > kMaxLiterals) acc.lits.resize(kMaxLiterals);
        break;
      }
      for (auto& l : acc.lits)
        if (l.bytes.size() > kMaxLiteralLen) { l.bytes.resize(kMaxLiteralLen); l.complete = false; }
    }
    if (acc.lits.size() == 1 && acc.lits[0].bytes.empty()) return {};
    return acc;
  }

  Seq suffixes(const ReP& re, int depth) {  // extractor.go:575-700
    if (depth > 100) return {};
    switch (re->op) {
      case OpLiteral: {
        if (re->flags & FoldCase) return expandCaseFoldLiteral(re->rune);   // :587-589
        auto b = runesToBytes(re->rune);
        if (b.size() > kMaxLiteralLen) b.erase(b.begin(), b.end() - kMaxLiteralLen);
        Seq s; s.lits.push_back({b, true}); return s;
      }
      case OpConcat: {
        if (re->sub.empty()) return {};
        int last = static_cast<int>(re->sub.size()) - 1;
        while (last >= 0) {
          Op o = re->sub[last]->op;
          if (o != OpEndLine && o != OpEndText && o != OpWordBoundary && o != OpNoWordBoundary) break;
          last--;
        }
        if (last < 0) return {};
        Seq suf = suffixes(re->sub[last], depth + 1);
        if (suf.empty()) return {};
        for (int i = last - 1; i >= 0; i--) {
          const ReP& sub = re->sub[i];
          if (sub->op == OpWordBoundary || sub->op == OpNoWordBoundary) continue;
          if (sub->op != OpLiteral) { markInexact(suf); return suf; }
          auto prefix = runesToBytes(sub->rune);
          for (auto& l : suf.lits) {
            std::vector<uint8_t> nb = prefix;
            nb.insert(nb.end(), l.bytes.begin(), l.bytes.end());
            if (nb.size() > kMaxLiteralLen) nb.erase(nb.begin(), nb.end() - kMaxLiteralLen);
            l.bytes.swap(nb);
          }
          if (static_cast<int>(suf.lits.size()) > kMaxLiterals) return suf;
        }
        return suf;
      }
      case OpAlternate: {
        Seq all;
        for (auto& sub : re->sub) {
          Seq s = suffixes(sub, depth + 1);
          if (s.empty()) return {};
          for (auto& l : s.lits) {
            all.lits.push_back(l);
            if (static_cast<int>(all.lits.size()) >= kMaxLiterals) return all;
          }
        }
        return all;
      }
      case OpCharClass: return expandCharClass(re);
      case OpCapture: return re->sub.empty() ? Seq{} : suffixes(re->sub[0], depth + 1);
      default: return {};
    }
  }

  Seq inner(const ReP& re, int depth) {  // extractor.go:744-810
    if (depth > 100) return {};
    switch (re->op) {
      case OpLiteral: {
        if (re->flags & FoldCase) { Seq f = expandCaseFoldLiteral(re->rune); markInexact(f); return f; }   // :753-760
        auto b = runesToBytes(re->rune);
        if (b.size() > kMaxLiteralLen) b.resize(kMaxLiteralLen);
        Seq s; s.lits.push_back({b, false}); return s;
      }
      case OpConcat:
        for (auto& sub : re->sub) { Seq s = inner(sub, depth + 1); if (!s.empty()) return s; }
        return {};
      case OpAlternate: {
        Seq all;
        for (auto& sub : re->sub) {
          Seq s = inner(sub, depth + 1);
          if (s.empty()) return {};
          for (auto& l : s.lits) {
            all.lits.push_back(l);
            if (static_cast<int>(all.lits.size()) >= kMaxLiterals) return all;
          }
        }
        return all;
      }
      case OpCharClass: return expandCharClass(re);
      case OpCapture: return re->sub.empty() ? Seq{} : inner(re->sub[0], depth + 1);
      default: return {};
    }
  }
};

bool isWildcardOrRepetition(const ReP& re) {  // extractor.go:1208-1240
  switch (re->op) {
    case OpStar: case OpPlus: case OpQuest: case OpRepeat: case OpAnyChar: case OpAnyCharNotNL: return true;
    case OpConcat: case OpAlternate:
      for (auto& s : re->sub) if (isWildcardOrRepetition(s)) return true;
      return false;
    case OpCapture: return !re->sub.empty() && isWildcardOrRepetition(re->sub[0]);
    default: return false;
  }
}

// ----------------------------------------------------------------- AST predicates (meta/strategy.go)
bool anyOp(const ReP& re, std::initializer_list<Op> ops) {
  for (Op o : ops) if (re->op == o) return true;
  for (auto& s : re->sub) if (anyOp(s, ops)) return true;
  return false;
}
bool hasWordBoundary(const ReP& re) { return anyOp(re, {OpWordBoundary, OpNoWordBoundary}); }
bool hasAnchorAssertions(const ReP& re) {
  return anyOp(re, {OpBeginLine, OpEndLine, OpBeginText, OpEndText, OpWordBoundary, OpNoWordBoundary});
}
bool hasNonLineAnchors(const ReP& re) {
  return anyOp(re, {OpEndLine, OpBeginText, OpEndText, OpWordBoundary, OpNoWordBoundary});
}
bool hasFold(const ReP& re) {
  if (re->op == OpLiteral && (re->flags & FoldCase)) return true;
  for (auto& s : re->sub) if (hasFold(s)) return true;
  return false;
}
bool hasNonGreedy(const ReP& re) {  // strategy.go:651-668
  if ((re->op == OpStar || re->op == OpPlus || re->op == OpQuest || re->op == OpRepeat) && (re->flags & NonGreedy)) return true;
  for (auto& s : re->sub) if (hasNonGreedy(s)) return true;
  return false;
}
bool isEndAnchoredTail(const ReP& re) {  // nfa/compile.go:1798-1830
  switch (re->op) {
    case OpEndText: return true;
    case OpConcat: return !re->sub.empty() && isEndAnchoredTail(re->sub.back());
    case OpCapture: return !re->sub.empty() && isEndAnchoredTail(re->sub[0]);
    case OpAlternate:
      if (re->sub.empty()) return false;
      for (auto& s : re->sub) if (!isEndAnchoredTail(s)) return false;
      return true;
    default: return false;
  }
}

bool isDigitOnlyClass(const std::vector<int>& r) {
  if (r.empty()) return false;
  for (size_t i = 0; i + 1 < r.size(); i += 2) if (r[i] < '0' || r[i + 1] > '9') return false;
  return true;
}
bool isDigitLead(const ReP& re);
bool isDigitLeadConcat(const std::vector<ReP>& subs) {  // strategy.go:331-385
  for (auto& sub : subs) {
    bool optional = sub->op == OpQuest || sub->op == OpStar || (sub->op == OpRepeat && sub->min == 0);
    if (optional) {
      if (sub->sub.empty()) return false;
      const ReP& in = sub->sub[0];
      bool digitOnly;
      if (in->op == OpCharClass) digitOnly = isDigitOnlyClass(in->rune);
      else if (in->op == OpLiteral) {
        digitOnly = !in->rune.empty();
        for (int r : in->rune) if (r < '0' || r > '9') digitOnly = false;
      } else digitOnly = isDigitLead(in);
      if (!digitOnly) return false;
      continue;
    }
    return isDigitLead(sub);
  }
  return false;
}
bool isDigitLead(const ReP& re) {  // strategy.go:416-495
  switch (re->op) {
    case OpCharClass: return isDigitOnlyClass(re->rune);
    case OpLiteral: return !re->rune.empty() && re->rune[0] >= '0' && re->rune[0] <= '9';
    case OpAlternate:
      if (re->sub.empty()) return false;
      for (auto& s : re->sub) if (!isDigitLead(s)) return false;
      return true;
    case OpConcat: return !re->sub.empty() && isDigitLeadConcat(re->sub);
    case OpCapture: case OpPlus: return !re->sub.empty() && isDigitLead(re->sub[0]);
    case OpRepeat: return !re->sub.empty() && re->min >= 1 && isDigitLead(re->sub[0]);
    default: return false;
  }
}
bool isDigitRunSkipSafe(const ReP& re) {  // strategy.go:530-560
  switch (re->op) {
    case OpConcat: case OpCapture: return !re->sub.empty() && isDigitRunSkipSafe(re->sub[0]);
    case OpPlus: case OpStar:
      return re->sub.size() == 1 && re->sub[0]->op == OpCharClass && isDigitOnlyClass(re->sub[0]->rune);
    case OpRepeat:
      return re->max == -1 && re->sub.size() == 1 && re->sub[0]->op == OpCharClass && isDigitOnlyClass(re->sub[0]->rune);
    default: return false;
  }
}
bool extractCharClassRanges(const ReP& re, std::vector<std::pair<uint8_t, uint8_t>>& out) {  // charclass_extract.go:19-71
  if (re->op != OpPlus || re->sub.size() != 1) return false;
  const ReP& sub = re->sub[0];
  if (sub->op != OpCharClass || sub->rune.size() % 2) return false;
  for (size_t i = 0; i + 1 < sub->rune.size(); i += 2) {
    if (sub->rune[i] > 127 || sub->rune[i + 1] > 127) return false;
    out.push_back({static_cast<uint8_t>(sub->rune[i]), static_cast<uint8_t>(sub->rune[i + 1])});
  }
  return !out.empty();
}
bool isValidCompositePart(const ReP& re) {  // nfa/composite.go:275-302
  switch (re->op) {
    case OpPlus: case OpStar: case OpQuest: case OpRepeat: return re->sub.size() == 1 && re->sub[0]->op == OpCharClass;
    case OpCharClass: return true;
    default: return false;
  }
}
bool isCompositeCharClassPattern(const ReP& re) {
  if (re->op != OpConcat || re->sub.size() < 2) return false;
  for (auto& s : re->sub) if (!isValidCompositePart(s)) return false;
  return true;
}
bool isSimpleCharClass(const ReP& re) {  // strategy.go:1095-1128
  switch (re->op) {
    case OpCharClass: return true;
    case OpPlus: case OpStar: case OpQuest: case OpRepeat: case OpCapture:
      return re->sub.size() == 1 && isSimpleCharClass(re->sub[0]);
    case OpConcat:
      for (auto& s : re->sub) if (!isSimpleCharClass(s)) return false;
      return true;
    default: return false;
  }
}
bool isWildcardSubexpression(ReP re) {  // strategy.go:586-603
  while (re->op == OpCapture && !re->sub.empty()) re = re->sub[0];
  if ((re->op == OpStar || re->op == OpPlus) && !re->sub.empty() &&
      (re->sub[0]->op == OpAnyChar || re->sub[0]->op == OpAnyCharNotNL)) return true;
  if (re->op == OpPlus && !re->sub.empty() && re->sub[0]->op == OpCharClass) return true;
  if (re->op == OpRepeat && re->min >= 1) return true;
  return false;
}
bool containsAnchor(const ReP& re) { return anyOp(re, {OpBeginLine, OpEndLine, OpBeginText, OpEndText}); }
// (?m)^ + wildcard + suffix literal -> UseMultilineReverseSuffix (strategy.go:718-861)
bool containsLineStartAnchor(const ReP& re) {  // strategy.go:734-768
  switch (re->op) {
    case OpBeginLine: return true;
    case OpBeginText: return false;
    case OpConcat:
      if (!re->sub.empty() && re->sub[0]->op == OpBeginLine) return true;
      for (auto& s : re->sub) if (containsLineStartAnchor(s)) return true;
      return false;
    case OpAlternate:
      if (re->sub.empty()) return false;
      for (auto& s : re->sub) if (!containsLineStartAnchor(s)) return false;
      return true;
    case OpCapture: return !re->sub.empty() && containsLineStartAnchor(re->sub[0]);
    default: return false;
  }
}
bool containsWildcard(const ReP& re) {  // strategy.go:771-793
  switch (re->op) {
    case OpStar: case OpPlus:
      return !re->sub.empty() && (re->sub[0]->op == OpAnyChar || re->sub[0]->op == OpAnyCharNotNL);
    case OpConcat: case OpAlternate:
      for (auto& s : re->sub) if (containsWildcard(s)) return true;
      return false;
    case OpCapture: case OpQuest: case OpRepeat: return !re->sub.empty() && containsWildcard(re->sub[0]);
    default: return false;
  }
}
bool isWildcardOp(const ReP& re) {  // strategy.go:846-861
  if ((re->op == OpStar || re->op == OpPlus) && !re->sub.empty()) {
    const ReP& sub = re->sub[0];
    if (sub->op == OpAnyChar || sub->op == OpAnyCharNotNL) return true;
    if (re->op == OpPlus && sub->op == OpCharClass) return true;
  }
  return false;
}
bool isSafeForMultilineReverseSuffix(const ReP& re) {  // strategy.go:805-843
  if (!(containsLineStartAnchor(re) && containsWildcard(re))) return false;   // isMultilineLineAnchored :728-730
  if (re->op == OpCapture) return !re->sub.empty() && isSafeForMultilineReverseSuffix(re->sub[0]);
  if (re->op != OpConcat || re->sub.size() < 2) return false;
  bool line = false, wild = false;
  for (size_t i = 0; i < re->sub.size(); i++) {
    if (i == 0 && re->sub[i]->op == OpBeginLine) { line = true; continue; }
    if (isWildcardOp(re->sub[i])) wild = true;
  }
  return line && wild;
}
bool isSafeForReverseSuffix(const ReP& re) {  // strategy.go:605-634
  if (re->op == OpCapture) return !re->sub.empty() && isSafeForReverseSuffix(re->sub[0]);
  if (re->op != OpConcat || re->sub.size() < 2) return false;
  int wc = 0;
  for (size_t i = 0; i + 1 < re->sub.size(); i++) if (isWildcardSubexpression(re->sub[i])) wc++;
  if (wc == 0) return false;
  for (size_t i = 1; i + 1 < re->sub.size(); i++) if (containsAnchor(re->sub[i])) return false;
  return true;
}
bool isSafeForReverseInner(const ReP& re) {  // strategy.go:874-907
  if (re->op == OpCapture) return !re->sub.empty() && isSafeForReverseInner(re->sub[0]);
  if (re->op != OpConcat || re->sub.size() < 2) return false;
  const ReP& f = re->sub[0];
  if ((f->op == OpStar || f->op == OpPlus) && !f->sub.empty() &&
      (f->sub[0]->op == OpAnyChar || f->sub[0]->op == OpAnyCharNotNL)) return true;
  if (f->op == OpPlus && !f->sub.empty() && f->sub[0]->op == OpCharClass) return true;
  return false;
}

size_t minLitLen(const Seq& s) {
  size_t m = SIZE_MAX;
  for (auto& l : s.lits) m = std::min(m, l.bytes.size());
  return m;
}

}  // namespace

Seq extractPrefixes(const ReP& re) {  // ExtractPrefixes extractor.go:128-156
  Extractor e;
  Seq seq = e.prefixes(re, 0);
  if (seq.lits.size() > 64) {
    Seq original = seq;
    for (int keep : {4, 3, 2}) {
      if (seq.lits.size() <= 64) break;
      keepFirstBytes(seq, keep);
      dedup(seq);
    }
    if (seq.lits.size() > 64) seq = original;
  }
  return seq;
}
Seq extractSuffixes(const ReP& re) { Extractor e; return e.suffixes(re, 0); }
Seq extractInner(const ReP& re) { Extractor e; return e.inner(re, 0); }

// ----------------------------------------------------------------- SelectStrategy
static Strategy selectStrategy(const NFA& nfa, const ReP& re, const Seq& lits, bool& restated) {
  restated = true;
  bool startAnchored = nfa.anchored;
  // nfa.IsPatternEndAnchored (nfa/compile.go:1785-1795): ends with \z / $ and holds no end anchor anywhere else (`(a$)b$`)
  std::function<bool(const ReP&)> anyEndAnchor = [&](const ReP& r) -> bool {          // containsEndAnchor :1872-1888
    switch (r->op) {
      case OpEndText: case OpEndLine: return true;
      case OpConcat: case OpAlternate: for (auto& x : r->sub) if (anyEndAnchor(x)) return true; return false;
      case OpCapture: case OpStar: case OpPlus: case OpQuest: case OpRepeat: return !r->sub.empty() && anyEndAnchor(r->sub[0]);
      default: return false;
    }
  };
  std::function<bool(const ReP&)> internalEndAnchor = [&](const ReP& r) -> bool {     // hasInternalEndAnchor :1828-1856
    switch (r->op) {
      case OpConcat:
        for (size_t i = 0; i + 1 < r->sub.size(); i++) if (anyEndAnchor(r->sub[i])) return true;
        return !r->sub.empty() && internalEndAnchor(r->sub.back());
      case OpCapture: return !r->sub.empty() && internalEndAnchor(r->sub[0]);
      case OpAlternate: for (auto& x : r->sub) if (internalEndAnchor(x)) return true; return false;
      default: return false;
    }
  };
  bool endAnchored = isEndAnchoredTail(re) && !internalEndAnchor(re);
  bool hasStartAnchor = anyOp(re, {OpBeginText, OpBeginLine});   // IsPatternStartAnchored :1897-1926: ^ of either kind, in any branch
  if (endAnchored && !startAnchored && !hasStartAnchor) { restated = false; return UseReverseAnchored; }
  if (startAnchored) { restated = false; return UseBoundedBacktracker; }  // or AnchoredLiteral / BranchDispatch

  // selectReverseStrategy strategy.go:974-1093
  auto reverse = [&]() -> int {
    // nfa.HasImpossibleEndAnchor (nfa/compile.go:1858-1888): an end anchor — (?m)$ counts — that does not END the pattern
    // in the sense of isEndAnchored (\z / non-multiline $ only): no reverse strategy (strategy.go:980-985)
    std::function<bool(const ReP&)> containsEndAnchor = [&](const ReP& r) -> bool {
      switch (r->op) {
        case OpEndText: case OpEndLine: return true;
        case OpConcat: case OpAlternate:
          for (auto& x : r->sub) if (containsEndAnchor(x)) return true;
          return false;
        case OpCapture: case OpStar: case OpPlus: case OpQuest: case OpRepeat: return !r->sub.empty() && containsEndAnchor(r->sub[0]);
        default: return false;
      }
    };
    if (containsEndAnchor(re) && !isEndAnchoredTail(re)) return 0;
    if (hasWordBoundary(re)) return 0;
    if (isSafeForMultilineReverseSuffix(re)) {          // strategy.go:1004-1012, before the fast-prefix check
      Seq suf = extractSuffixes(re);
      if (!suf.empty() && suf.lcs().size() >= 1) return UseMultilineReverseSuffix;
    }
    bool fast = false;  // hasFastPrefixPrefilter strategy.go:948-967
    if (!lits.empty()) {
      if (lits.lcp().size() >= 1) fast = true;
      else if (lits.lits.size() == 1) fast = true;
      else fast = minLitLen(lits) >= 3;   // prefilter.WouldBeFast prefilter.go:314-337
    }
    if (fast) return 0;
    Seq suf = extractSuffixes(re);
    if (!suf.empty()) {
      if (suf.lcs().size() >= 1) {
        if (!isSafeForReverseSuffix(re)) return 0;
        return UseReverseSuffix;
      }
    }
    if (isSafeForReverseSuffix(re) && !suf.empty()) {  // shouldUseReverseSuffixSet :909-935
      bool exactAlt = !lits.empty() && lits.allComplete() && lits.lits.size() == suf.lits.size();
      size_t n = suf.lits.size();
      if (!exactAlt && n >= 2 && n <= 32 && minLitLen(suf) >= 2) return UseReverseSuffixSet;
    }
    if (re->op == OpConcat && re->sub.size() >= 3) {  // ExtractInnerForReverseSearch extractor.go:1061-1100
      Extractor e;
      for (size_t i = 1; i + 1 < re->sub.size(); i++) {
        Seq in = e.inner(re->sub[i], 0);
        if (in.empty()) continue;
        bool before = false, after = false;
        for (size_t j = 0; j < i; j++) if (isWildcardOrRepetition(re->sub[j])) { before = true; break; }
        for (size_t j = i + 1; j < re->sub.size(); j++) if (isWildcardOrRepetition(re->sub[j])) { after = true; break; }
        if (before && after) {
          size_t l = in.lcp().size();
          if (l == 1 && isDigitLead(re)) return 0;
          if (l >= 1) {
            if (!isSafeForReverseInner(re)) return 0;
            return UseReverseInner;
          }
          return 0;  // innerInfo found but no usable lcp
        }
      }
    }
    return 0;
  };
  if (int s = reverse()) { restated = false; return static_cast<Strategy>(s); }

  int nfaSize = static_cast<int>(nfa.states.size());
  bool good = !lits.empty() && lits.lcp().size() >= 1;               // analyzeLiterals :1256-1308
  bool teddyLits = false, acLits = false;
  size_t lc = lits.lits.size();
  if (lc >= 2 && lc <= 64) teddyLits = minLitLen(lits) >= 3;
  if (lc > 64) acLits = minLitLen(lits) >= 1;
  bool anchors = hasAnchorAssertions(re);
  bool nonLineAnchors = anchors && hasNonLineAnchors(re);

  std::vector<std::pair<uint8_t, uint8_t>> ranges;
  if (!good && !teddyLits && extractCharClassRanges(re, ranges)) return UseCharClassSearcher;
  if (!good && !teddyLits && isCompositeCharClassPattern(re)) { restated = false; return UseCompositeSearcher; }
  if (!good && !teddyLits && isSimpleCharClass(re)) return UseBoundedBacktracker;   // (unanchored: findAt below restates its two engines)
  if (teddyLits && lits.allComplete() && !nonLineAnchors) return UseTeddy;     // selectLiteralStrategy :1143-1170
  if (acLits && lits.allComplete()) { restated = false; return UseAhoCorasick; }
  if (nfaSize <= 100 && isDigitLead(re)) return UseDigitPrefilter;             // shouldUseDigitPrefilter :511-523
  if (nfaSize < 20) {
    if ((hasWordBoundary(re) && anchors) || canMatchEmpty(re) || anyOp(re, {OpBeginLine, OpEndLine})) return UseNFA;
    return UseDFA;
  }
  if (!good && !teddyLits && canMatchEmpty(re)) return UseNFA;
  if (good || teddyLits) {
    if (nfaSize > 200 && !lits.allComplete()) return UseNFA;
    return UseDFA;
  }
  if (nfaSize > 100) return UseNFA;
  return UseBoth;
}

// ----------------------------------------------------------------- CompileRegexp
std::unique_ptr<Engine> compileEngine(const std::string& pattern) {
  auto e = std::make_unique<Engine>();
  e->pattern = pattern;
  e->re = parse(pattern);
  e->nfa = compileNFA(e->re);
  if (!e->nfa.anchored) e->prefixes = extractPrefixes(e->re);   // compile.go:466-478
  e->strategy = selectStrategy(e->nfa, e->re, e->prefixes, e->strategyRestated);
  e->pikevm.init(&e->nfa);
  // The lazy DFA carries the look-around state itself (engines.hpp "look-around"): five start kinds, \b / \B resolved when
  // the next byte is known, $ re-closed in front of '\n', \z and $ at the end of input.  Restated with its quirks — the
  // answers of a DFA strategy over an NFA with assertions are the reference's, not always stdlib's.
  const bool dfaOK = true;
  // prefilter.NewBuilder(prefixes, nil).Build() (compile.go:472-476 -> prefilter/prefilter.go:261-297): one literal ->
  // memchr / memmem; 2..64 literals of >= 3 bytes -> Teddy, more -> Aho-Corasick; otherwise none.  Whatever the kind, Find
  // returns the first position at which one of the literals occurs (prefilter.go:77).
  bool& hasPrefilter = e->hasPrefilter;
  hasPrefilter = false;
  if (!e->prefixes.empty()) {
    size_t minLen = ~size_t(0);
    for (auto& l : e->prefixes.lits) minLen = std::min(minLen, l.bytes.size());
    hasPrefilter = e->prefixes.lits.size() == 1 || minLen >= 3;   // (one literal of ANY length: memchr / memmem — an empty needle is found at once, prefilter.go:270-283)
  }
  auto prefilterFind = [lits = e->prefixes.lits](Bytes h, int64_t n, int64_t pos) -> int64_t {
    for (int64_t i = pos; i < n; i++)
      for (auto& l : lits) {
        const int64_t m = static_cast<int64_t>(l.bytes.size());
        if (i + m <= n && std::memcmp(h + i, l.bytes.data(), static_cast<size_t>(m)) == 0) return i;
      }
    return -1;
  };
  switch (e->strategy) {
    case UseDFA:
      if (dfaOK) {
        e->dfa.init(&e->nfa, true);
        // The skip only shows in results when the NFA holds assertions (engines.hpp); otherwise keep the plain walk.
        if (hasPrefilter && e->nfa.hasLook) e->dfa.prefilterFind = prefilterFind;
        if (!hasNonGreedy(e->re)) {  // buildReverseDFA compile.go:184-205
          e->revNfa = reverseNFA(e->nfa);
          e->revDfa.init(&e->revNfa, false);
          e->hasReverseDFA = true;
        } else if (e->nfa.hasLook && !hasPrefilter) {
          // No reverse DFA and no prefilter: findIndicesDFAAtWithState first asks DFA.IsMatchAt (find_indices.go:396-403 ->
          // lazy.go:561-828 searchEarliestMatch) and gives up when that says no; without assertions it cannot miss a match,
          // with them it can (per-class transition cache, boundary flags).  Restated: LazyDFA::isMatchAt, used in findAt below.
          // (With a prefilter the reference goes prefilter -> PikeVM, :381-393, and no DFA is involved.)
          e->dfaGatesPikeVM = true;
        }
      }
      break;
    case UseBoth:
      if (dfaOK) e->dfa.init(&e->nfa, true);
      break;
    case UseDigitPrefilter:
      if (dfaOK) e->dfa.init(&e->nfa, true);
      e->digitRunSkipSafe = isDigitRunSkipSafe(e->re);   // compile.go:176
      break;
    case UseTeddy: {
      std::vector<std::vector<uint8_t>> pats;
      for (auto& l : e->prefixes.lits) pats.push_back(l.bytes);
      if (!e->teddy.build(pats)) {   // Slim (2..32) or Fat (33..64) Teddy
        e->strategyRestated = false;
      }
      // adjustForAnchors compile.go:660-680: the only anchor a UseTeddy pattern can hold is (?m)^ (selectLiteralStrategy
      // excludes the others); the complete prefilter is wrapped with a line-start check
      e->teddyLineAnchor = anyOp(e->re, {OpBeginLine});
      break;
    }
    case UseBoundedBacktracker:
      if (!e->nfa.anchored) {   // buildReverseDFA compile.go:207-217: both lazy DFAs, for inputs past the backtracker's capacity
        e->dfa.init(&e->nfa, true);
        e->revNfa = reverseNFA(e->nfa);
        e->revDfa.init(&e->revNfa, false);
      }
      break;
    case UseCharClassSearcher: {
      std::vector<std::pair<uint8_t, uint8_t>> ranges;
      extractCharClassRanges(e->re, ranges);
      for (auto& r : ranges) for (int b = r.first; b <= r.second; b++) e->ccs.membership[b] = true;
      e->ccs.minMatch = 1;
      break;
    }
    default: break;
  }
  return e;
}

// ----------------------------------------------------------------- per-strategy find
bool Engine::findAt(Bytes h, int64_t len, int64_t at, int64_t& s, int64_t& e) {
  if (at > 0 && nfa.anchored) return false;   // find_indices.go:1129-1131
  switch (strategy) {
    case UseDigitPrefilter: {  // find_indices.go:1050-1088
      if (!dfa.nfa || at >= len) return pikevm.searchAt(h, len, at, s, e);
      int64_t pos = at;
      while (pos < len) {
        int64_t d = memchrDigitAt(h, len, pos);
        if (d < 0) return false;
        int64_t end = dfa.searchAtAnchored(h, len, d);
        if (end != -1) { s = d; e = end; return true; }
        pos = d + 1;
        if (digitRunSkipSafe) while (pos < len && h[pos] >= '0' && h[pos] <= '9') pos++;
      }
      return false;
    }
    case UseTeddy:  // find_indices.go:925-951
      if (!strategyRestated || at >= len) return pikevm.searchAt(h, len, at, s, e);
      if (teddyLineAnchor) {
        // lineAnchorWrapper has Find but no FindMatch (prefilter/wrap.go:52-82): findIndicesTeddyAt takes its fallback,
        // Find + LiteralLen (:941-950).  Find: Teddy candidates until one sits at a line start (wrap.go:52-66).
        int64_t pos = at, cs, ce;
        for (;;) {
          if (!teddy.findMatch(h, len, pos, cs, ce)) return false;
          if (cs == 0 || h[cs - 1] == '\n') break;
          pos = cs + 1;
        }
        size_t uniform = teddy.patterns[0].size();            // Teddy.LiteralLen teddy.go:566-571 (complete, uniformLen)
        for (auto& p : teddy.patterns) if (p.size() != uniform) uniform = 0;
        if (uniform > 0) { s = cs; e = cs + static_cast<int64_t>(uniform); return true; }
        return pikevm.searchAt(h, len, cs, s, e);              // findIndicesNFAAt(haystack, pos)
      }
      return teddy.findMatch(h, len, at, s, e);
    case UseCharClassSearcher: return ccs.searchAt(h, len, at, s, e);   // :841-848
    case UseDFA:
      if (dfa.nfa && hasReverseDFA) {  // findIndicesBidirectionalDFACore :686-705
        int64_t end = dfa.searchAt(h, len, at);
        if (end == -1) return false;
        if (end == at) { s = e = at; return true; }
        int64_t st = revDfa.searchReverse(h, len, at, end);
        if (st < 0) return false;
        s = st; e = end; return true;
      }
      if (dfaGatesPikeVM && at < len && !dfa.isMatchAt(h, len, at)) return false;   // find_indices.go:396-400
      return pikevm.searchAt(h, len, at, s, e);
    case UseBoundedBacktracker: {  // findIndicesBoundedBacktrackerAtWithState :1225-1300 (unanchored programs: classes only, no '.')
      if (nfa.anchored || !dfa.nfa) return pikevm.searchAt(h, len, at, s, e);
      const int64_t remaining = len - at;
      // BoundedBacktracker.CanHandle (nfa/backtrack.go:139-143): states x (span + 1) visited entries within 32 M (:82)
      if (static_cast<int64_t>(nfa.states.size()) * (remaining + 1) <= 32ll * 1024 * 1024) {
        // Search(haystack[at:]): priority-ordered depth-first search per start position, first match wins (backtrack.go:264-300,
        // :401-490) — the leftmost-first match of the slice, which is what the PikeVM reports
        int64_t ss, ee;
        if (!pikevm.searchAt(h + at, remaining, 0, ss, ee)) return false;
        s = at + ss; e = at + ee;
        return true;
      }
      int64_t end = dfa.searchAt(h, len, at);            // findIndicesBidirectionalDFALongest :711-732
      if (end == -1) return false;
      if (end == at) { s = e = at; return true; }
      int64_t st = revDfa.searchReverse(h, len, at, end);
      if (st < 0) return false;
      s = st; e = end; return true;
    }
    case UseBoth:  // findIndicesAdaptiveAtWithState :408-441
      if (dfa.nfa && !hasPrefilter) {   // e.prefilter == nil: selectPrefilter found nothing usable in the prefixes (NOT: no prefixes at all)
        int64_t end = dfa.searchAt(h, len, at);
        if (end != -1) {
          int64_t est = at;
          if (end > at + 100) est = end - 100;
          return pikevm.searchAt(h, len, est, s, e);
        }
      }
      return pikevm.searchAt(h, len, at, s, e);
    default:
      return pikevm.searchAt(h, len, at, s, e);
  }
}

void Engine::findAll(Bytes h, int64_t len, int64_t n, std::vector<int64_t>& out) {
  out.clear();
  if (strategy == UseCharClassSearcher) {  // findall.go:157-169
    ccs.findAll(h, len, out);
    if (n > 0 && static_cast<int64_t>(out.size() / 2) > n) out.resize(n * 2);
    return;
  }
  int64_t pos = 0, lastMatchEnd = -1;
  if (nfa.anchored) {  // findall.go:201-207
    int64_t s, e;
    if (findAt(h, len, 0, s, e)) { out.push_back(s); out.push_back(e); }
    return;
  }
  bool direct = strategy == UseDFA && dfa.nfa && hasReverseDFA;   // findall.go:216-218
  while (n <= 0 || static_cast<int64_t>(out.size() / 2) < n) {
    int64_t s = 0, e = 0; bool found;
    if (direct) {
      int64_t end = dfa.searchAt(h, len, pos);
      if (end < 0) break;
      if (end == pos) { s = e = pos; found = true; }
      else {
        int64_t st = revDfa.searchReverse(h, len, pos, end);
        if (st < 0) break;
        s = st; e = end; found = true;
      }
    } else {
      found = findAt(h, len, pos, s, e);
    }
    if (!found) break;
    if (s == e && s == lastMatchEnd) {  // findall.go:251-257
      pos++;
      if (pos > len) break;
      continue;
    }
    out.push_back(s); out.push_back(e);
    if (s != e) lastMatchEnd = e;
    if (s == e) pos = e + 1; else if (e > pos) pos = e; else pos++;
    if (pos > len) break;
  }
}

int64_t Engine::count(Bytes h, int64_t len, int64_t n) {  // findall.go:297-376
  if (n == 0) return 0;
  int64_t cnt = 0, pos = 0, lastEnd = -1;
  bool direct = strategy == UseDFA && dfa.nfa && hasReverseDFA;
  while (pos <= len) {
    int64_t s = 0, e = 0; bool found;
    if (direct) {
      int64_t end = dfa.searchAt(h, len, pos);
      if (end < 0) break;
      if (end == pos) { s = e = pos; found = true; }
      else {
        int64_t st = revDfa.searchReverse(h, len, pos, end);
        if (st < 0) break;
        s = st; e = end; found = true;
      }
    } else found = findAt(h, len, pos, s, e);
    if (!found) break;
    if (s == e && s == lastEnd) { pos++; if (pos > len) break; continue; }
    cnt++;
    if (s != e) lastEnd = e;
    if (s == e) pos = e + 1; else if (e > pos) pos = e; else pos++;
    if (n > 0 && cnt >= n) break;
  }
  return cnt;
}

void Engine::findAllSubmatch(Bytes h, int64_t len, int64_t n, std::vector<int64_t>& out) {
  out.clear();
  if (n == 0) return;
  int groups = numGroups();
  int64_t pos = 0, lastMatchEnd = -1, cnt = 0;
  while (pos <= len) {
    std::vector<int64_t> slots;
    bool found;
    switch (strategy) {  // findSubmatchAtWithState findall.go:89-98
      case UseBoundedBacktracker: case UseNFA: case UseDFA: case UseBoth: case UseDigitPrefilter:
        found = pikevm.searchCaptures(h, len, pos, slots);
        break;
      default: {  // two-phase, findall.go:100-127
        int64_t s, e;
        found = findAt(h, len, pos, s, e);
        if (found) {
          if (groups <= 1) slots = {s, e};
          else {
            // SearchWithCapturesInSpan (pikevm.go:1210): anchored at s, haystack cut at e
            NFA anch = nfa; anch.anchored = true;
            PikeVM pv; pv.init(&anch);
            if (!pv.searchCaptures(h, e, s, slots)) found = pikevm.searchCaptures(h, len, pos, slots);
          }
        }
      }
    }
    if (!found) break;
    int64_t ms = slots[0], me = slots[1];
    if (ms == me && ms == lastMatchEnd) { pos++; if (pos > len) break; continue; }
    for (int i = 0; i < groups * 2; i++) out.push_back(i < static_cast<int>(slots.size()) ? slots[i] : -1);
    cnt++;
    if (ms != me) lastMatchEnd = me;
    if (ms == me) pos = me + 1; else if (me > pos) pos = me; else pos++;
    if (n > 0 && cnt >= n) break;
  }
}

}  // namespace orc
