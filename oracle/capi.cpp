// ORACLE — TEST INFRASTRUCTURE ONLY. C entry points for the ctypes checker
// (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Never linked into
// libcoregex_hip.so.
#include <cstring>
#include <string>

#include "meta.hpp"

using namespace orc;

namespace {
thread_local std::string g_err;
}

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

void* orc_compile(const char* pattern, int64_t len) {
  try {
    return compileEngine(std::string(pattern, static_cast<size_t>(len))).release();
  } catch (const ParseError& e) {
    g_err = "parse: " + e.msg;
  } catch (const CompileError& e) {
    g_err = "compile: " + e.msg;
  } catch (const std::exception& e) {
    g_err = e.what();
  }
  return nullptr;
}

void orc_free(void* e) { delete static_cast<Engine*>(e); }
int orc_strategy(void* e) { return static_cast<Engine*>(e)->strategy; }
int orc_strategy_restated(void* e) { return static_cast<Engine*>(e)->strategyRestated ? 1 : 0; }
const char* orc_strategy_name(int s) { return strategyName(s); }
int orc_num_groups(void* e) { return static_cast<Engine*>(e)->numGroups(); }
int orc_nfa_states(void* e) { return static_cast<int>(static_cast<Engine*>(e)->nfa.states.size()); }
int orc_alphabet_len(void* e) { return static_cast<Engine*>(e)->nfa.alphabetLen; }
void orc_byte_classes(void* e, uint8_t* out) {
  std::memcpy(out, static_cast<Engine*>(e)->nfa.byteClasses.data(), 256);
}
int orc_dfa_states(void* e) { return static_cast<int>(static_cast<Engine*>(e)->dfa.numStates()); }
int orc_digit_run_skip_safe(void* e) { return static_cast<Engine*>(e)->digitRunSkipSafe; }
int orc_num_prefix_literals(void* e) { return static_cast<int>(static_cast<Engine*>(e)->prefixes.lits.size()); }
int orc_prefix_literal(void* e, int i, uint8_t* buf, int cap, int* complete) {
  auto& l = static_cast<Engine*>(e)->prefixes.lits[i];
  int n = static_cast<int>(l.bytes.size());
  if (n <= cap) std::memcpy(buf, l.bytes.data(), n);
  *complete = l.complete;
  return n;
}

// Literal extractor alone (literal/extractor_test.go vectors): which = 0 ExtractPrefixes, 1 ExtractSuffixes, 2 ExtractInner.
// Output: one record per literal = length byte, complete byte, the bytes.  Returns the number of literals, -1 on a parse
// error, -2 when the records do not fit.
int orc_extract_literals(const char* pattern, int64_t len, int which, uint8_t* buf, int cap) {
  try {
    ReP re = parse(std::string(pattern, static_cast<size_t>(len)));
    Seq seq = which == 0 ? extractPrefixes(re) : which == 1 ? extractSuffixes(re) : extractInner(re);
    int used = 0;
    for (auto& l : seq.lits) {
      const int n = static_cast<int>(l.bytes.size());
      if (n > 255 || used + 2 + n > cap) return -2;
      buf[used++] = static_cast<uint8_t>(n);
      buf[used++] = l.complete ? 1 : 0;
      std::memcpy(buf + used, l.bytes.data(), static_cast<size_t>(n));
      used += n;
    }
    return static_cast<int>(seq.lits.size());
  } catch (const ParseError& e) {
    g_err = "parse: " + e.msg;
  } catch (const std::exception& e) {
    g_err = e.what();
  }
  return -1;
}

static int64_t copyOut(const std::vector<int64_t>& v, int64_t* out, int64_t capVals) {
  int64_t n = static_cast<int64_t>(v.size());
  if (out && n <= capVals) std::memcpy(out, v.data(), n * sizeof(int64_t));
  return n;
}

// Returns the number of int64 values (2 per match); writes them if they fit in cap.
int64_t orc_find_all(void* e, const uint8_t* h, int64_t len, int64_t limit, int64_t* out, int64_t capVals) {
  std::vector<int64_t> v;
  static_cast<Engine*>(e)->findAll(h, len, limit, v);
  return copyOut(v, out, capVals);
}
int64_t orc_count(void* e, const uint8_t* h, int64_t len, int64_t limit) {
  return static_cast<Engine*>(e)->count(h, len, limit);
}
int64_t orc_find_all_submatch(void* e, const uint8_t* h, int64_t len, int64_t limit, int64_t* out, int64_t capVals) {
  std::vector<int64_t> v;
  static_cast<Engine*>(e)->findAllSubmatch(h, len, limit, v);
  return copyOut(v, out, capVals);
}

// Unit-level probes used by the golden-vector tests.
int64_t orc_dfa_search_at_anchored(void* e, const uint8_t* h, int64_t len, int64_t at) {
  Engine* en = static_cast<Engine*>(e);
  if (!en->dfa.nfa) en->dfa.init(&en->nfa, true);
  return en->dfa.searchAtAnchored(h, len, at);
}
int64_t orc_dfa_search_at(void* e, const uint8_t* h, int64_t len, int64_t at) {
  Engine* en = static_cast<Engine*>(e);
  if (!en->dfa.nfa) en->dfa.init(&en->nfa, true);
  return en->dfa.searchAt(h, len, at);
}
int64_t orc_memchr_digit_at(const uint8_t* h, int64_t len, int64_t at) { return memchrDigitAt(h, len, at); }

int orc_pikevm_captures(void* e, const uint8_t* h, int64_t len, int64_t at, int64_t* slots) {
  Engine* en = static_cast<Engine*>(e);
  std::vector<int64_t> s;
  if (!en->pikevm.searchCaptures(h, len, at, s)) return 0;
  std::memcpy(slots, s.data(), s.size() * sizeof(int64_t));
  return 1;
}

// Standalone Teddy (prefilter/teddy_test.go vectors): patterns packed as len-prefixed bytes.
void* orc_teddy_new(const uint8_t* packed, int npat) {
  auto* t = new Teddy();
  std::vector<std::vector<uint8_t>> pats;
  const uint8_t* p = packed;
  for (int i = 0; i < npat; i++) {
    int n = *p++;
    pats.emplace_back(p, p + n);
    p += n;
  }
  if (!t->build(pats)) { delete t; return nullptr; }
  return t;
}
void orc_teddy_free(void* t) { delete static_cast<Teddy*>(t); }
int orc_teddy_find_match(void* t, const uint8_t* h, int64_t len, int64_t start, int64_t* s, int64_t* e) {
  return static_cast<Teddy*>(t)->findMatch(h, len, start, *s, *e) ? 1 : 0;
}

int orc_dump(void* e, char* buf, int cap) {
  Engine* en = static_cast<Engine*>(e);
  std::string s = dump(en->re) + "\n" + dumpNFA(en->nfa);
  int n = static_cast<int>(s.size());
  if (n < cap) std::memcpy(buf, s.c_str(), n + 1);
  return n;
}

}  // extern "C"
