// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU baseline leg of bench.py ("port" kind).
//
// The reference cannot be built or timed here (Go), so the baseline is a C++ port of
// the algorithm the reference runs for BASELINE config 2 (UseDigitPrefilter), written the way the
// reference runs it on amd64:
//   * digit scan, 32 bytes per iteration with AVX2 compares + movemask + tzcnt
//     (simd/memchr_digit_amd64.s:26; scalar below 32 bytes, memchr_digit_amd64.go:24)
//   * anchored walk over a flat premultiplied transition table indexed by byte class
//     (dfa/lazy/lazy.go:261-268), 1-byte match delay, dead-state exit
//   * digit-run skip on failure (meta/find_indices.go:1079-1084) under the FindAll loop
//     (meta/findall.go:176-283).
// The table is taken from the oracle's lazy DFA after it has been warmed on a prefix of the input;
// a transition not yet determinized falls back to the lazy path.  Results are checked against the
// plain oracle by the caller.  One thread: the reference runs a search on the caller's goroutine.
#include <immintrin.h>

#include <cstring>
#include <vector>

#include "meta.hpp"

using namespace orc;

namespace {

inline int64_t digitScanAVX2(const uint8_t* h, int64_t n, int64_t at) {
  int64_t i = at;
  if (n - i >= 32) {
    const __m256i lo = _mm256_set1_epi8(0x2F), hi = _mm256_set1_epi8(0x39);
    for (; i + 32 <= n; i += 32) {
      const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(h + i));
      const __m256i gt = _mm256_cmpgt_epi8(v, lo);           // > '/'   (signed: bytes >= 0x80 compare false)
      const __m256i le = _mm256_cmpgt_epi8(v, hi);           // > '9'
      const unsigned m = static_cast<unsigned>(_mm256_movemask_epi8(_mm256_andnot_si256(le, gt)));
      if (m) return i + __builtin_ctz(m);
    }
  }
  for (; i < n; i++) if (h[i] >= '0' && h[i] <= '9') return i;
  return -1;
}

}  // namespace

extern "C" int64_t orc_baseline_digit_find_all(void* ev, const uint8_t* h, int64_t len, int64_t* out, int64_t capVals) {
  Engine* e = static_cast<Engine*>(ev);
  if (e->strategy != UseDigitPrefilter || !e->dfa.nfa) return -1;
  {  // warm the lazy DFA so the flat table below is (almost always) complete
    std::vector<int64_t> tmp;
    e->findAll(h, len < (1 << 20) ? len : (1 << 20), -1, tmp);
  }
  LazyDFA& d = e->dfa;
  const int stride = e->nfa.alphabetLen;
  auto snapshot = [&](std::vector<int32_t>& flat, std::vector<uint8_t>& isMatch) {
    flat.assign(d.states.size() * stride, LazyDFA::kUnknown);
    isMatch.assign(d.states.size(), 0);
    for (size_t s = 0; s < d.states.size(); s++) {
      isMatch[s] = d.states[s].isMatch;
      for (int c = 0; c < stride; c++) {
        int32_t t = d.states[s].trans[c];
        flat[s * stride + c] = t >= 0 ? t * stride : t;     // premultiplied offsets (state.go:16-42)
      }
    }
  };
  std::vector<int32_t> flat; std::vector<uint8_t> isMatch;
  snapshot(flat, isMatch);
  const uint8_t* cls = e->nfa.byteClasses.data();
  const bool skip = e->digitRunSkipSafe;
  int64_t n = 0, pos = 0;
  while (pos < len) {
    const int64_t dp = digitScanAVX2(h, len, pos);
    if (dp < 0) break;
    // SearchAtAnchored
    int32_t sid = d.startState(h, dp, true);
    int64_t off = static_cast<int64_t>(sid) * stride;
    int64_t last = -1, i = dp;
    for (; i < len; i++) {
      int32_t nx = flat[off + cls[h[i]]];
      if (nx == LazyDFA::kUnknown) {                         // cold transition: determinize, re-snapshot
        const int32_t cur = static_cast<int32_t>(off / stride);
        const int32_t t = d.next(cur, h[i]);
        snapshot(flat, isMatch);
        nx = t >= 0 ? t * stride : t;
      }
      if (nx == LazyDFA::kDead) break;
      off = nx;
      if (isMatch[off / stride]) last = i;
    }
    if (i >= len && d.eoiMatch(static_cast<int32_t>(off / stride))) last = len;
    if (last >= 0) {
      if (out && n + 2 <= capVals) { out[n] = dp; out[n + 1] = last; }
      n += 2;
      pos = last > pos ? last : pos + 1;
    } else {
      pos = dp + 1;
      if (skip) while (pos < len && h[pos] >= '0' && h[pos] <= '9') pos++;
    }
  }
  return n;
}
