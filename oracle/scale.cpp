// ORACLE — TEST INFRASTRUCTURE ONLY.  The oracle run multi-threaded over synthlog-v1 pages: the scale check of
// SURVEY §8(d)(ii) (count + order-sensitive checksum of all rows at BASELINE sizes, compared with the device's rows) and
// the all-cores leg of bench.py's cpu_baseline.
//
// Why blocks are independent: every synthlog page ends in '\n', and '\n' lies outside the alphabet of every BASELINE
// pattern, so FindAll over the corpus is the concatenation of FindAll over page-aligned blocks with offsets rebased
// (the same argument shards the corpus across GPUs).  Each thread owns its own Engine (the lazy DFA cache
// is mutable, meta/search_state.go:23-62 keeps one per caller too).
//
// Checksum (same formula as tests/test_gpu_parity.py computes on the device with torch): for row k (0-based, global)
// and column j: sum[j] += value * (k + 1 + 7 * j)  mod 2^64.
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "../coregex_amd/csrc/device/synth.hpp"   // the corpus generator (host twin of the fill kernel), not an algorithm of the path
#include "meta.hpp"

using namespace orc;

namespace {
struct BlockResult {
  uint64_t rows = 0;
  uint64_t s1[16] = {0};   // sum of value * (local_k + 1 + 7 j)
  uint64_t s0[16] = {0};   // sum of value
};
}  // namespace

extern "C" {

// out: [0] rows, [1..16] checksum per column, [17] scan nanoseconds of the slowest thread, [18] generation ns of it.
// width: 2 (FindAll) or 2 * groups (FindAllSubmatch).  Returns 0, or -1 on a compile error / bad argument.
// Offsets are relative to the first byte of page `first_page`.
int orc_scan_synth(const char* pattern, int64_t plen, uint32_t config, uint64_t seed, uint64_t first_page, uint64_t npages,
                   int nthreads, int width, uint64_t* out) {
  if (width < 2 || width > 16 || nthreads < 1) return -1;
  constexpr uint64_t kBlockPages = 256;               // 1 MiB per unit of work
  const uint64_t nblocks = (npages + kBlockPages - 1) / kBlockPages;
  std::vector<BlockResult> res(nblocks);
  std::vector<uint64_t> scanNs(nthreads, 0), genNs(nthreads, 0);
  std::vector<int> bad(nthreads, 0);
  auto work = [&](int t) {
    std::unique_ptr<Engine> e;
    try { e = compileEngine(std::string(pattern, static_cast<size_t>(plen))); } catch (...) { bad[t] = 1; return; }
    std::vector<uint8_t> buf(kBlockPages * cxgsynth::kPage + 64, 0);
    std::vector<int64_t> rows;
    for (uint64_t b = t; b < nblocks; b += nthreads) {
      const uint64_t p0 = b * kBlockPages, p1 = std::min(npages, p0 + kBlockPages);
      auto t0 = std::chrono::steady_clock::now();
      for (uint64_t p = p0; p < p1; p++) cxgsynth::page(config, seed, first_page + p, buf.data() + (p - p0) * cxgsynth::kPage);
      auto t1 = std::chrono::steady_clock::now();
      rows.clear();
      const int64_t len = static_cast<int64_t>((p1 - p0) * cxgsynth::kPage);
      if (width == 2) e->findAll(buf.data(), len, -1, rows); else e->findAllSubmatch(buf.data(), len, -1, rows);
      auto t2 = std::chrono::steady_clock::now();
      genNs[t] += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
      scanNs[t] += std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count();
      BlockResult& r = res[b];
      const uint64_t base = p0 * cxgsynth::kPage;
      r.rows = rows.size() / static_cast<size_t>(width);
      for (uint64_t k = 0; k < r.rows; k++)
        for (int j = 0; j < width; j++) {
          const int64_t raw = rows[k * width + j];
          const uint64_t v = raw < 0 ? static_cast<uint64_t>(raw) : static_cast<uint64_t>(raw) + base;   // -1 (unset group) stays -1
          r.s0[j] += v;
          r.s1[j] += v * (k + 1 + 7ull * static_cast<uint64_t>(j));
        }
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
  for (auto& x : th) x.join();
  for (int t = 0; t < nthreads; t++) if (bad[t]) return -1;
  std::memset(out, 0, 19 * sizeof(uint64_t));
  uint64_t rank = 0;
  for (uint64_t b = 0; b < nblocks; b++) {
    for (int j = 0; j < width; j++) out[1 + j] += res[b].s1[j] + rank * res[b].s0[j];   // local rank -> global rank
    rank += res[b].rows;
  }
  out[0] = rank;
  for (int t = 0; t < nthreads; t++) if (scanNs[t] > out[17]) { out[17] = scanNs[t]; out[18] = genNs[t]; }
  return 0;
}

}  // extern "C"
