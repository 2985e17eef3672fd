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
  LazyDFA& d = e->dfa;
  if (d.states.size() < 8) {  // warm the lazy DFA ONCE per engine so the flat table below is (almost always) complete — round 6: this ran on every
    std::vector<int64_t> tmp;  // call, i.e. the all-cores leg (1 MiB blocks) timed the oracle's warm-up, not the port (256 threads: 2.1 GB/s)
    e->findAll(h, len < (1 << 16) ? len : (1 << 16), -1, tmp);
  }
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

// ---------------------------------------------------------------------------------------------------------------
// Ports of the reference's CPU paths for the other BASELINE configurations (bench.py --config N, cpu_baseline leg).
// Same rule as above: written the way the reference runs on amd64, results checked against the plain oracle / the GPU
// by the caller.  All single-threaded; bench.py runs them on page-aligned blocks from many threads for the all-cores leg.

// Config 3, UseTeddy: Slim Teddy, 2-byte fingerprint, 16 bytes per iteration with SSSE3 PSHUFB nibble lookups
// (prefilter/teddy_ssse3_amd64.s:273-480: lo/hi nibble masks of fingerprint byte 0 and of byte 1 shifted by one
// position, AND, PMOVMSKB of the non-zero bytes), then verifyBucket (prefilter/teddy.go:532-550) per candidate, buckets
// low to high, ids ascending; FindAll loop of meta/find_indices.go:925-951 / meta/findall.go:176-283 around it.
extern "C" int64_t orc_baseline_teddy_find_all(void* ev, const uint8_t* h, int64_t len, int64_t* out, int64_t capVals) {
  Engine* e = static_cast<Engine*>(ev);
  if (e->strategy != UseTeddy) return -1;
  const Teddy& t = e->teddy;
  if (t.buckets.size() > 8) return -1;                 // Fat Teddy (AVX2, 16 buckets) has no SIMD port here
  alignas(16) uint8_t l8[2][16], h8[2][16];            // the oracle keeps u16 masks (fat); slim uses the low byte
  for (int p = 0; p < 2; p++) for (int k = 0; k < 16; k++) { l8[p][k] = static_cast<uint8_t>(t.lo[p][k]); h8[p][k] = static_cast<uint8_t>(t.hi[p][k]); }
  const __m128i lo0 = _mm_load_si128(reinterpret_cast<const __m128i*>(l8[0])), hi0 = _mm_load_si128(reinterpret_cast<const __m128i*>(h8[0]));
  const __m128i lo1 = _mm_load_si128(reinterpret_cast<const __m128i*>(l8[1])), hi1 = _mm_load_si128(reinterpret_cast<const __m128i*>(h8[1]));
  const __m128i nib = _mm_set1_epi8(0x0F), zero = _mm_setzero_si128();
  auto verify = [&](int64_t pos, uint32_t mask, int64_t& ms, int64_t& me) -> bool {
    while (mask) {
      const int bucket = __builtin_ctz(mask);
      mask &= mask - 1;
      if (bucket >= static_cast<int>(t.buckets.size())) continue;
      for (int id : t.buckets[bucket]) {
        const auto& p = t.patterns[id];
        if (pos + static_cast<int64_t>(p.size()) <= len && std::memcmp(h + pos, p.data(), p.size()) == 0) { ms = pos; me = pos + static_cast<int64_t>(p.size()); return true; }
      }
    }
    return false;
  };
  auto maskAt = [&](int64_t i) -> uint32_t {       // scalar tail, teddy.go:491-530
    if (i + 1 >= len) return 0;
    return (t.lo[0][h[i] & 15] & t.hi[0][h[i] >> 4]) & (t.lo[1][h[i + 1] & 15] & t.hi[1][h[i + 1] >> 4]);
  };
  int64_t n = 0, pos = 0;
  while (pos < len) {
    int64_t ms = -1, me = -1;
    int64_t i = pos;
    bool found = false;
    for (; i + 17 <= len && !found; i += 16) {
      const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(h + i));
      const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(h + i + 1));
      const __m128i m0 = _mm_and_si128(_mm_shuffle_epi8(lo0, _mm_and_si128(a, nib)), _mm_shuffle_epi8(hi0, _mm_and_si128(_mm_srli_epi16(a, 4), nib)));
      const __m128i m1 = _mm_and_si128(_mm_shuffle_epi8(lo1, _mm_and_si128(b, nib)), _mm_shuffle_epi8(hi1, _mm_and_si128(_mm_srli_epi16(b, 4), nib)));
      const __m128i c = _mm_and_si128(m0, m1);
      uint32_t hits = static_cast<uint32_t>(_mm_movemask_epi8(_mm_cmpeq_epi8(c, zero))) ^ 0xFFFFu;
      while (hits) {
        const int k = __builtin_ctz(hits);
        hits &= hits - 1;
        alignas(16) uint8_t cm[16];
        _mm_store_si128(reinterpret_cast<__m128i*>(cm), c);
        if (verify(i + k, cm[k], ms, me)) { found = true; break; }
      }
      if (found) break;
    }
    if (!found) for (; i < len; i++) { const uint32_t m = maskAt(i); if (m && verify(i, m, ms, me)) { found = true; break; } }
    if (!found) break;
    if (out && n + 2 <= capVals) { out[n] = ms; out[n + 1] = me; }
    n += 2;
    pos = me > pos ? me : pos + 1;
  }
  return n;
}

// Config 4, UseCharClassSearcher: the scalar 256-entry membership LUT loop, one byte per iteration
// (nfa/charclass_searcher.go:158-211 FindAllIndices: state machine over `matching`, trailing run closed at the end).
extern "C" int64_t orc_baseline_charclass_find_all(void* ev, const uint8_t* h, int64_t len, int64_t* out, int64_t capVals) {
  Engine* e = static_cast<Engine*>(ev);
  if (e->strategy != UseCharClassSearcher) return -1;
  bool lut[256];
  for (int b = 0; b < 256; b++) lut[b] = e->ccs.membership[b];
  int64_t n = 0, start = 0;
  bool matching = false;
  for (int64_t i = 0; i < len; i++) {
    const bool m = lut[h[i]];
    if (!matching) { if (m) { start = i; matching = true; } }
    else if (!m) { if (out && n + 2 <= capVals) { out[n] = start; out[n + 1] = i; } n += 2; matching = false; }
  }
  if (matching) { if (out && n + 2 <= capVals) { out[n] = start; out[n + 1] = len; } n += 2; }
  return n;
}

// Config 1, UseDFA with a complete literal prefix (`error`): the reference's memmem prefilter finds every occurrence
// (prefilter/prefilter.go:440-506 -> simd/memmem.go:53-152) and, the literal being the whole pattern, each hit is a match.
// Port of the path a 2..6-byte needle takes there:
//   SelectRareBytes (simd/byte_frequencies.go:88-135): the two rarest distinct bytes of the needle by its rank table, the
//     rarest first.  The ranks are the reference's data and are not restated here; the caller passes the pair — for `error`
//     that function yields ('r' at 1, 'o' at 3): ranks r 195 < o 205 < e 245 — or asks for this file's own coarse ranking
//     (letters by English text frequency, everything else rarer), which picks the same pair for the 16 literals of config 3.
//   memmemPaired (memmem.go:103-152): MemchrPair for byte1 at p and byte2 at p + (idx2 - idx1), then bytesEqual on the whole
//     needle at p - idx1; on a miss the scan resumes one byte behind the candidate.
//   memchrPairAVX2 (simd/memchr_amd64.s:355-): 32 bytes at p and 32 at p + offset, VPCMPEQB both, VPAND, VPMOVMSKB; scalar tail
//     (memchr_generic_impl.go:253).
// The FindAll loop around it is findall.go:176-283 without its per-call overhead (state pool, interface dispatch).
namespace {
inline int64_t memchrPair(const uint8_t* h, int64_t n, uint8_t b1, uint8_t b2, int64_t off) {
  if (n <= off) return -1;
  int64_t p = 0;
  const __m256i v1 = _mm256_set1_epi8(static_cast<char>(b1)), v2 = _mm256_set1_epi8(static_cast<char>(b2));
  for (; p + off + 32 <= n; p += 32) {
    const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(h + p));
    const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(h + p + off));
    const uint32_t m = static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_and_si256(_mm256_cmpeq_epi8(a, v1), _mm256_cmpeq_epi8(b, v2))));
    if (m) return p + __builtin_ctz(m);
  }
  for (; p + off < n; p++) if (h[p] == b1 && h[p + off] == b2) return p;
  return -1;
}
inline int coarseRank(uint8_t b) {   // higher = more frequent; not the reference's table (see above)
  static const char order[] = "zqxjkvbpygfwmucldrhsnioate";
  const uint8_t c = (b >= 'A' && b <= 'Z') ? static_cast<uint8_t>(b + 32) : b;
  for (int i = 0; order[i]; i++) if (order[i] == c) return 100 + 5 * i + (b == c ? 2 : 0);
  return b == ' ' ? 255 : 50;
}
}  // namespace
extern "C" void orc_baseline_rare_pair(const uint8_t* lit, int64_t n, int32_t* out4) {   // byte1, idx1, byte2, idx2 — the selection loop of byte_frequencies.go:104-134 over coarseRank
  uint8_t b1 = lit[0], b2 = n > 1 ? lit[1] : lit[0];
  int64_t i1 = 0, i2 = n > 1 ? 1 : 0;
  if (coarseRank(b2) < coarseRank(b1)) { std::swap(b1, b2); std::swap(i1, i2); }
  for (int64_t i = 2; i < n; i++) {
    const uint8_t b = lit[i];
    const int r = coarseRank(b);
    if (r < coarseRank(b1)) { b2 = b1; i2 = i1; b1 = b; i1 = i; }
    else if (b != b1 && r < coarseRank(b2)) { b2 = b; i2 = i; }
  }
  out4[0] = b1; out4[1] = static_cast<int32_t>(i1); out4[2] = b2; out4[3] = static_cast<int32_t>(i2);
}
extern "C" int64_t orc_baseline_literal_find_all(const uint8_t* lit, int64_t litLen, const int32_t* pair, const uint8_t* h, int64_t len, int64_t* out, int64_t capVals) {
  uint8_t b1 = static_cast<uint8_t>(pair[0]), b2 = static_cast<uint8_t>(pair[2]);
  int64_t i1 = pair[1], i2 = pair[3];
  if (b1 == b2 || i1 == i2 || litLen > 6 || litLen < 2) return -1;          // (memmemSingle / memmemLong: not on config 1's path)
  if (i1 > i2) { std::swap(b1, b2); std::swap(i1, i2); }
  const int64_t off = i2 - i1;
  int64_t n = 0, pos = 0;
  while (pos + litLen <= len) {                                            // one Memmem call of the reference per iteration
    int64_t found = -1, from = pos;
    for (;;) {
      const int64_t c = memchrPair(h + from, len - from, b1, b2, off);
      if (c < 0) break;
      const int64_t cand = from + c, s = cand - i1;
      if (s >= pos && s + litLen <= len && std::memcmp(h + s, lit, static_cast<size_t>(litLen)) == 0) { found = s; break; }
      from = cand + 1;
      if (from >= len - off) break;
    }
    if (found < 0) break;
    if (out && n + 2 <= capVals) { out[n] = found; out[n + 1] = found + litLen; }
    n += 2;
    pos = found + litLen;
  }
  return n;
}

// ---- all host cores (round 6; VERDICT round 5, weak #1: the Python thread pool over ctypes calls measured its own harness).
// A std::thread pool over page-aligned blocks of ONE pre-generated host buffer of synthlog pages (every page ends in '\n': blocks
// are independent), one engine and one row scratch per thread allocated once, the threads kept across the passes
// (they meet at a barrier in front of every pass), blocks handed out dynamically (an atomic counter: 1 MiB each).  The caller
// sizes the buffer so that a pass takes about a second.  Nothing here is the product: coregex_amd/ never links this file.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include "../coregex_amd/csrc/device/synth.hpp"   // the corpus generator (host twin of the device fill kernel), not an algorithm of the path

extern "C" int64_t orc_baseline_literal_find_all(const uint8_t* lit, int64_t litLen, const int32_t* pair, const uint8_t* h, int64_t len, int64_t* out, int64_t capVals);
extern "C" int64_t orc_baseline_digit_find_all(void* ev, const uint8_t* h, int64_t len, int64_t* out, int64_t capVals);
extern "C" int64_t orc_baseline_teddy_find_all(void* ev, const uint8_t* h, int64_t len, int64_t* out, int64_t capVals);
extern "C" int64_t orc_baseline_charclass_find_all(void* ev, const uint8_t* h, int64_t len, int64_t* out, int64_t capVals);

// config: the BASELINE configuration (1..5: which port runs); synth_config / seed / first_page / npages: the corpus; check_pages: rows
// of the blocks inside the first check_pages pages are also summed separately (the part the GPU scanned: its row count is compared).
// out: [0] rows of all pages, [1] rows of the first check_pages pages, [2] ns to generate the buffer, [3] threads used,
// [4 .. 4 + npasses) wall ns of each pass.  Returns 0; -1 bad argument / compile error; -2 the buffer could not be allocated.
extern "C" int orc_baseline_all_cores(const char* pattern, int64_t plen, int config, const int32_t* pair, uint32_t synth_config, uint64_t seed,
                                      uint64_t first_page, uint64_t npages, uint64_t check_pages, int nthreads, int npasses, int width, uint64_t* out) {
  if (nthreads < 1 || npasses < 1 || npasses > 32 || config < 1 || config > 5 || npages == 0) return -1;
  constexpr uint64_t kBlockPages = 256;                       // 1 MiB per unit of work
  const uint64_t nblocks = (npages + kBlockPages - 1) / kBlockPages;
  std::unique_ptr<uint8_t[]> buf(new (std::nothrow) uint8_t[npages * cxgsynth::kPage + 64]);
  if (!buf) return -2;
  std::atomic<uint64_t> next{0};
  std::atomic<int> bad{0};
  std::vector<uint64_t> blockRows(nblocks, 0);
  // generation, all threads (also pages the buffer in)
  const auto g0 = std::chrono::steady_clock::now();
  {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back([&]() {
      for (;;) {
        const uint64_t b = next.fetch_add(1, std::memory_order_relaxed);
        if (b >= nblocks) break;
        const uint64_t p0 = b * kBlockPages, p1 = std::min(npages, p0 + kBlockPages);
        for (uint64_t p = p0; p < p1; p++) cxgsynth::page(synth_config, seed, first_page + p, buf.get() + p * cxgsynth::kPage);
      }
    });
    for (auto& x : th) x.join();
  }
  std::memset(buf.get() + npages * cxgsynth::kPage, 0, 64);
  const auto g1 = std::chrono::steady_clock::now();
  // the passes: persistent threads, a barrier in front of each pass
  std::mutex mu;
  std::condition_variable cv;
  int pass_open = -1, arrived = 0;
  bool quit = false;
  const std::string pat(pattern, static_cast<size_t>(plen));
  auto worker = [&](int) {
    std::unique_ptr<Engine> e;
    try { e = compileEngine(pat); } catch (...) { bad = 1; }
    std::vector<int64_t> scratch(static_cast<size_t>(kBlockPages * cxgsynth::kPage / 2 + 64) * static_cast<size_t>(width > 2 ? width / 2 : 1));
    std::vector<int64_t> rows;                                 // (the PikeVM port appends)
    int seen = -1;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        arrived++;
        cv.notify_all();
        cv.wait(lk, [&] { return quit || pass_open > seen; });
        if (quit) return;
        seen = pass_open;
      }
      if (bad) continue;
      for (;;) {
        const uint64_t b = next.fetch_add(1, std::memory_order_relaxed);
        if (b >= nblocks) break;
        const uint64_t p0 = b * kBlockPages, p1 = std::min(npages, p0 + kBlockPages);
        const uint8_t* h = buf.get() + p0 * cxgsynth::kPage;
        const int64_t len = static_cast<int64_t>((p1 - p0) * cxgsynth::kPage);
        int64_t nv = -1;
        switch (config) {
          case 1: nv = orc_baseline_literal_find_all(reinterpret_cast<const uint8_t*>(pat.data()), static_cast<int64_t>(pat.size()), pair, h, len, scratch.data(), static_cast<int64_t>(scratch.size())); break;
          case 2: nv = orc_baseline_digit_find_all(e.get(), h, len, scratch.data(), static_cast<int64_t>(scratch.size())); break;
          case 3: nv = orc_baseline_teddy_find_all(e.get(), h, len, scratch.data(), static_cast<int64_t>(scratch.size())); break;
          case 4: nv = orc_baseline_charclass_find_all(e.get(), h, len, scratch.data(), static_cast<int64_t>(scratch.size())); break;
          default: rows.clear(); e->findAllSubmatch(h, len, -1, rows); nv = static_cast<int64_t>(rows.size()); break;
        }
        if (nv < 0) { bad = 1; break; }
        blockRows[b] = static_cast<uint64_t>(nv) / static_cast<uint64_t>(width);
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) th.emplace_back(worker, t);
  for (int p = 0; p < npasses; p++) {
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return arrived == nthreads * (p + 1); });   // everybody is back (engines compiled, previous pass done)
      next = 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    {
      std::unique_lock<std::mutex> lk(mu);
      pass_open = p;
      cv.notify_all();
      cv.wait(lk, [&] { return arrived == nthreads * (p + 2); });
    }
    const auto t1 = std::chrono::steady_clock::now();
    out[4 + p] = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count());
  }
  { std::unique_lock<std::mutex> lk(mu); quit = true; cv.notify_all(); }
  for (auto& x : th) x.join();
  if (bad) return -1;
  uint64_t all = 0, chk = 0;
  for (uint64_t b = 0; b < nblocks; b++) { all += blockRows[b]; if ((b + 1) * kBlockPages <= check_pages) chk += blockRows[b]; }
  out[0] = all; out[1] = chk;
  out[2] = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(g1 - g0).count());
  out[3] = static_cast<uint64_t>(nthreads);
  return 0;
}

// How many cores does this process REALLY get?  nthreads threads each spin through `iters` dependent multiply-adds; returns the wall
// nanoseconds.  bench.py compares 1 thread with all visible threads: a container with a CPU quota (or an oversubscribed host) shows
// far fewer effective cores than its affinity mask (round 6: 256 visible, ~8 effective on the GPU box) — the all-cores leg then
// runs on what is really there and says so.
extern "C" uint64_t orc_spin_ns(int nthreads, uint64_t iters) {
  std::atomic<uint64_t> sink{0};
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) th.emplace_back([&, t]() {
    uint64_t x = 0x9E3779B97F4A7C15ull + static_cast<uint64_t>(t);
    for (uint64_t i = 0; i < iters; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;
    sink.fetch_add(x, std::memory_order_relaxed);
  });
  for (auto& x : th) x.join();
  const auto t1 = std::chrono::steady_clock::now();
  return static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count()) + (sink.load() == 1 ? 1 : 0);
}
