// bit_tricks.cc — host restatements of two round-5 device rewrites, each checked against the form it replaced on random inputs
// (tests/test_bit_tricks_cpu.py; test infrastructure only).  The device code uses DPP and LDS and cannot be compiled for the host, so
// the arithmetic is restated here line by line:
//   1. trio_rows (coregex_amd/csrc/device/scan_fields_wave.hip): the K bytes outside F below a match's end as the K highest bits of ONE
//      64-bit word W = (pz >> b) | (z << (64 - b)), against the search through the two words (h : l) with take_top;
//   2. k_scan_charclass_wave pass 2 (scan_charclass_wave.hip): four streams per lane (the 32-bit halves of S and of E), one bit of each
//      per iteration, staged at byte offset (rank * 26) & 2046 with an exhausted stream writing to a dump slot, rows read back as
//      (start of rank i, end of rank i + open) — against the plain extraction of set bits in order.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

static int take_top(uint64_t& l, uint64_t& h) {
  if (h) { int k = 63 - __builtin_clzll(h); h &= ~(1ull << k); return 64 + k; }
  if (l) { int k = 63 - __builtin_clzll(l); l &= ~(1ull << k); return k; }
  return -1;
}
template <int K> static void rows_two_words(uint64_t z, uint64_t pz, int b, int base, uint32_t& w0, uint32_t& w1) {
  uint64_t l = pz, h = b ? (z & ((1ull << b) - 1ull)) : 0ull;
  int pl[K - 1];
  for (int i = K - 2; i >= 0; i--) pl[i] = take_top(l, h);
  const int ps = take_top(l, h);
  w0 = w1 = 0;
  if (ps >= 0) {
    w0 = static_cast<uint32_t>(base + ps + 1) | (static_cast<uint32_t>(base + 64 + b) << 16);
    for (int i = 0; i < K - 1; i++) w1 |= static_cast<uint32_t>(pl[i] - ps - 1) << (8 * i);
  }
}
template <int K> static void rows_one_word(uint64_t z, uint64_t pz, int b, int base, uint32_t& w0, uint32_t& w1) {
  w0 = w1 = 0;
  uint64_t W = (pz >> b) | (b ? (z << (64 - b)) : 0ull);
  if (__builtin_popcountll(W) >= K) {
    int idx[K];
    for (int i = 0; i < K; i++) { const int c = __builtin_clzll(W); idx[i] = b + 63 - c; W &= ~(0x8000000000000000ull >> c); }
    const int ps = idx[K - 1];
    w0 = static_cast<uint32_t>(base + ps + 1) | (static_cast<uint32_t>(base + 64 + b) << 16);
    for (int i = 0; i < K - 1; i++) w1 |= static_cast<uint32_t>(idx[K - 2 - i] - ps - 1) << (8 * i);
  } else {
    rows_two_words<K>(z, pz, b, base, w0, w1);
  }
}

static long check_trio(long cases) {
  std::mt19937_64 g(7);
  long bad = 0;
  for (long it = 0; it < cases; it++) {
    uint64_t z = g(), pz = g();
    for (int d = 0; d < it % 6; d++) { z &= g(); pz &= g(); }          // from dense to very sparse (long fields: the fallback)
    if (it % 7 == 0) pz = 0;
    if (it % 11 == 0) z = 0;
    const int b = static_cast<int>(g() % 64), base = static_cast<int>(g() % 64) * 64 - 64;
    uint32_t a0, a1, b0, b1;
    rows_two_words<2>(z, pz, b, base, a0, a1); rows_one_word<2>(z, pz, b, base, b0, b1); bad += (a0 != b0 || a1 != b1);
    rows_two_words<3>(z, pz, b, base, a0, a1); rows_one_word<3>(z, pz, b, base, b0, b1); bad += (a0 != b0 || a1 != b1);
    rows_two_words<4>(z, pz, b, base, a0, a1); rows_one_word<4>(z, pz, b, base, b0, b1); bad += (a0 != b0 || a1 != b1);
  }
  return bad;
}

// one wave-tile of the char-class kernel's pass 2: 64 lanes, words S[l], E[l]; `open` = a run crosses the tile's first byte
static long check_staging(long tiles) {
  std::mt19937_64 g(11);
  long bad = 0;
  std::vector<uint8_t> rs(2048), re(2048);
  uint16_t dump[64];
  for (long t = 0; t < tiles; t++) {
    // a membership word stream with runs of random length, then S and E as the kernel owns them (starts [0, 3840), ends (0, 3840])
    uint64_t M[64];
    for (int l = 0; l < 64; l++) { M[l] = g(); if (t % 3 == 0) M[l] &= g(); if (t % 5 == 0) M[l] |= g(); }
    const uint32_t prev = static_cast<uint32_t>(g() & 1);
    uint64_t S[64], E[64];
    for (int l = 0; l < 64; l++) {
      const uint64_t carry = l ? (M[l - 1] >> 63) : prev;
      const uint64_t P = (M[l] << 1) | carry;
      S[l] = M[l] & ~P; E[l] = ~M[l] & P;
      if (l >= 60) { S[l] = 0; if (l > 60) E[l] = 0; else E[l] &= 1ull; }  // word_range(0, 3839) / (1, 3840)
      if (l == 0) E[l] &= ~1ull;
    }
    const uint32_t open = prev & static_cast<uint32_t>(M[0] & 1ull);
    // plain extraction
    std::vector<uint16_t> ps, pe;
    for (int l = 0; l < 64; l++) for (int k = 0; k < 64; k++) { if (S[l] >> k & 1) ps.push_back(static_cast<uint16_t>(64 * l + k)); if (E[l] >> k & 1) pe.push_back(static_cast<uint16_t>(64 * l + k)); }
    if (ps.size() > 1024 || pe.size() > 1024) continue;               // the kernel raises its fallback flag
    // four streams per lane
    std::memset(rs.data(), 0xEE, rs.size()); std::memset(re.data(), 0xEE, re.size());
    uint32_t srank = 0, erank = 0;
    for (int l = 0; l < 64; l++) {
      uint32_t s0 = static_cast<uint32_t>(S[l]), s1 = static_cast<uint32_t>(S[l] >> 32), e0 = static_cast<uint32_t>(E[l]), e1 = static_cast<uint32_t>(E[l] >> 32);
      uint32_t as0 = srank * 26u, as1 = as0 + static_cast<uint32_t>(__builtin_popcount(s0)) * 26u;
      uint32_t ae0 = erank * 26u, ae1 = ae0 + static_cast<uint32_t>(__builtin_popcount(e0)) * 26u;
      const uint32_t p0 = 64u * static_cast<uint32_t>(l), p1 = p0 + 32u;
      auto step = [&](uint32_t& b, uint32_t& ofs, std::vector<uint8_t>& arr, uint32_t pos) {
        const uint32_t bit = b ? static_cast<uint32_t>(__builtin_ctz(b)) : 0xFFFFFFFFu;
        const uint16_t v = static_cast<uint16_t>(pos | bit);
        if (b != 0u) std::memcpy(&arr[ofs & 2046u], &v, 2); else dump[l] = v;
        b &= b - 1u;
        ofs += 26u;
      };
      while ((s0 | s1 | e0 | e1) != 0u) { step(s0, as0, rs, p0); step(s1, as1, rs, p1); step(e0, ae0, re, p0); step(e1, ae1, re, p1); }
      srank += static_cast<uint32_t>(__builtin_popcountll(S[l])); erank += static_cast<uint32_t>(__builtin_popcountll(E[l]));
    }
    // rows as the kernel reads them back
    const uint32_t n = static_cast<uint32_t>(ps.size()), nen = static_cast<uint32_t>(pe.size());
    const uint32_t both = nen > open ? nen - open : 0u;
    for (uint32_t i = 0; i < n; i++) {
      uint16_t st; std::memcpy(&st, &rs[(i * 26u) & 2046u], 2);
      bad += st != ps[i];
      if (i < both) { uint16_t en; std::memcpy(&en, &re[((i + open) * 26u) & 2046u], 2); bad += en != pe[i + open]; }
    }
    if (n > both + 1u) bad++;                                          // at most one start without its end in the tile
    if (open && nen) { uint16_t en; std::memcpy(&en, &re[0], 2); bad += en != pe[0]; }
  }
  return bad;
}

int main() {
  const long a = check_trio(2000000), b = check_staging(20000);
  std::printf("trio_rows one word vs two words: %ld differ\nchar-class staging, four streams vs plain: %ld differ\n", a, b);
  return (a || b) ? 1 : 0;
}
