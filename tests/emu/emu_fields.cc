// TEST INFRASTRUCTURE ONLY — sequential twin of coregex_amd/csrc/device/scan_fields_wave.hip.
//
// The same steps as the kernel, word by word over "lanes" 0..63 of a 4096-byte window: class bitmaps, links, super-run
// starts of the owned lanes, the K-field hop as multiword additions with the kernel's generate / propagate carry
// resolution, the rare loop for super-runs with more than K fields, and the start search of every end in the bitmap of
// group starts (this lane's word, else the previous lane's).  Returns -(16 + reason) where the kernel would raise its
// fallback flag.  `own_words` (60 on the device) is a parameter so that the CPU tests put many more tile borders on a
// kilobyte of text.  Nothing in coregex_amd/ links or loads this file.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../coregex_amd/csrc/device/scan_dfa.h"
#include "../../coregex_amd/csrc/device/walk.hpp"

using namespace cxgdev;

namespace {
int fields_shape_host(const ChainAux& c) {      // scan_fields_wave.hip fields_shape
  if (c.ncls != 2 || (c.nops & 1u) == 0 || c.nops < 3 || c.nops > 7 || c.restart_check) return 0;
  for (uint32_t k = 0; k < c.nops; k++) {
    if (c.op_kind[k] != ((k & 1u) ? kChainByte : kChainRun)) return 0;
    if (c.op_cls[k] != (k & 1u)) return 0;
  }
  for (int q = 0; q < 2; q++) if (c.cls_kind[q] == kClsSet || c.cls_hi[q] > 0x7Fu || c.cls_lo[q] > c.cls_hi[q]) return 0;
  if (c.cls_lo[0] <= c.cls_hi[1] && c.cls_lo[1] <= c.cls_hi[0]) return 0;
  return static_cast<int>((c.nops + 1) / 2);
}
inline uint32_t ffbh_raw(uint32_t v) { return v ? static_cast<uint32_t>(__builtin_clz(v)) : 0xFFFFFFFFu; }
inline uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
}  // namespace

extern "C" int emu_fields_shape(const uint8_t* blob) {
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(blob);
  if (h->magic != kBlobMagic || !(h->flags & kFlagChainOrdered) || (h->flags & (kFlagChainBounded | kFlagChainSets))) return 0;
  return fields_shape_host(*reinterpret_cast<const ChainAux*>(blob + h->aux_off + 256));
}

// pre_words: window words in front of the tile — 1 for k_scan_fields_wave (64 bytes, lanes 1..60 own, 192 bytes behind), 2 for
// k_scan_fields_pers since round 5 (128 bytes, lanes 2..61 own, 128 bytes behind: line-aligned windows).
extern "C" int64_t emu_find_all_fields2(const uint8_t* blob, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals, int own_words, int pre_words) {
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(blob);
  if (h->magic != kBlobMagic) return -1;
  if (!(h->flags & kFlagChainOrdered)) return -4;
  const ChainAux& ch = *reinterpret_cast<const ChainAux*>(blob + h->aux_off + 256);
  const int K = fields_shape_host(ch);
  if (!K) return -5;
  if (own_words < 1 || pre_words < 1 || pre_words > 2 || own_words + pre_words > 63) return -2;
  const int NW = 64;
  const int64_t tile_bytes = 64LL * own_words, pre = 64LL * pre_words, N = 64LL * NW;
  const unsigned long long own_mask = ((1ull << own_words) - 1ull) << pre_words;   // lanes pre_words .. pre_words + own_words - 1
  std::vector<int64_t> res;
  const uint64_t ntiles = (len + tile_bytes - 1) / tile_bytes;
  for (uint64_t t = 0; t < ntiles; t++) {
    const int64_t tile_lo = static_cast<int64_t>(t) * tile_bytes;
    const int64_t wlo = tile_lo - pre;                                 // haystack position of window bit 0
    uint64_t D[64] = {0}, P[64] = {0};
    for (int64_t b = 0; b < N; b++) {
      const int64_t p = wlo + b;
      if (p < 0 || p >= static_cast<int64_t>(len)) continue;           // in front of the haystack / past the input: no class
      if (chain_class_has(ch, 0, hay[p])) D[b >> 6] |= 1ull << (b & 63);
      if (chain_class_has(ch, 1, hay[p])) P[b >> 6] |= 1ull << (b & 63);
    }
    unsigned long long PPd = 0;
    for (int l = 0; l < 64; l++) if (D[l] == ~0ull) PPd |= 1ull << l;
    uint64_t L[64], B[64];
    for (int l = 0; l < 64; l++) {
      const uint64_t prev_top = l ? (D[l - 1] >> 63) : (D[0] >> 63);   // lane 0 sees its own word (DPP keeps the old value): it owns nothing
      const uint64_t next_bot = l < 63 ? (D[l + 1] & 1ull) : 1ull;     // behind the window: "a field byte follows"
      const uint64_t Dl = (D[l] << 1) | prev_top, Dr = (D[l] >> 1) | (next_bot << 63);
      L[l] = P[l] & Dl & Dr;
    }
    for (int l = 0; l < 64; l++) {
      const uint64_t prev_top = l ? (D[l - 1] >> 63) : (D[0] >> 63);
      const uint64_t prev_ltop = l ? (L[l - 1] >> 63) : (L[0] >> 63);
      const uint64_t Dl = (D[l] << 1) | prev_top, Ll = (L[l] << 1) | prev_ltop;
      B[l] = ((own_mask >> l) & 1ull) ? (D[l] & ~Dl & ~Ll) : 0ull;
    }
    unsigned long long ovf = 0;
    auto carry_in = [&](unsigned long long GG) {
      const unsigned long long Pe = PPd & ~GG;
      const unsigned long long recv = (Pe + (GG << 1)) ^ Pe;
      ovf |= GG | (Pe & recv);
      return recv;
    };
    auto add_words = [&](const uint64_t* A, const uint64_t* Bv, uint64_t* S) {   // per-lane 64-bit addition + the kernel's carry resolution
      unsigned long long GG = 0;
      for (int l = 0; l < 64; l++) {
        const unsigned __int128 s = static_cast<unsigned __int128>(A[l]) + Bv[l];
        S[l] = static_cast<uint64_t>(s);
        if (s >> 64) GG |= 1ull << l;
      }
      const unsigned long long recv = carry_in(GG);
      for (int l = 0; l < 64; l++) S[l] += (recv >> l) & 1ull;
    };
    auto hop = [&](const uint64_t* M, uint64_t* R) {
      uint64_t S[64], T[64], Q[64];
      add_words(D, M, S);
      for (int i = 1; i < K; i++) {
        for (int l = 0; l < 64; l++) { Q[l] = S[l] & L[l]; T[l] = D[l] | Q[l]; }
        add_words(T, Q, S);
      }
      std::memcpy(R, S, sizeof S);
    };
    uint64_t R[64], E[64], EL[64];
    hop(B, R);
    bool more = false;
    for (int l = 0; l < 64; l++) { E[l] = R[l] & ~D[l]; EL[l] = R[l] & L[l]; more = more || EL[l]; }
    while (more) {
      uint64_t M2[64];
      if (EL[63] >> 63) ovf |= 1ull << 63;
      for (int l = 0; l < 64; l++) M2[l] = (EL[l] << 1) | (l ? (EL[l - 1] >> 63) : 0ull);
      for (int l = 0; l < 64; l++) B[l] |= M2[l];
      hop(M2, R);
      more = false;
      for (int l = 0; l < 64; l++) { E[l] |= R[l] & ~D[l]; EL[l] = R[l] & L[l]; more = more || EL[l]; }
    }
    uint32_t reason = (ovf >> 63) ? 1u : 0u;
    // rows: lane-major, low half first
    for (int l = 0; l < 64 && !reason; l++) {
      const uint32_t b0 = static_cast<uint32_t>(B[l]), b1 = static_cast<uint32_t>(B[l] >> 32);
      const uint32_t pb0 = l ? static_cast<uint32_t>(B[l - 1]) : 0u, pb1 = l ? static_cast<uint32_t>(B[l - 1] >> 32) : 0u;
      const uint32_t lane64 = static_cast<uint32_t>(l) << 6;
      auto emit = [&](uint32_t s, uint32_t e) {
        if (s >= e) { reason |= 2u; return; }
        res.push_back(wlo + s); res.push_back(wlo + e);
      };
      {
        const uint32_t tp = umin(ffbh_raw(pb1) | 32u, ffbh_raw(pb0) | 64u);
        uint32_t xx = static_cast<uint32_t>(E[l]);
        while (xx) {
          const uint32_t b = static_cast<uint32_t>(__builtin_ctz(xx));
          xx &= xx - 1u;
          const uint32_t d = umin(ffbh_raw(b0 & ((1u << b) - 1u)), tp);
          emit((lane64 + 31u - d) & 0xFFFFu, lane64 + b);
        }
      }
      {
        const uint32_t tp = umin(umin(ffbh_raw(b0) | 32u, ffbh_raw(pb1) | 64u), ffbh_raw(pb0) | 96u);
        uint32_t xx = static_cast<uint32_t>(E[l] >> 32);
        while (xx) {
          const uint32_t b = static_cast<uint32_t>(__builtin_ctz(xx));
          xx &= xx - 1u;
          const uint32_t d = umin(ffbh_raw(b1 & ((1u << b) - 1u)), tp);
          emit((lane64 + 63u - d) & 0xFFFFu, lane64 + 32u + b);
        }
      }
    }
    if (reason) return -(16 + static_cast<int64_t>(reason));
  }
  const int64_t n = static_cast<int64_t>(res.size());
  if (out && n <= cap_vals) std::memcpy(out, res.data(), n * sizeof(int64_t));
  return n;
}

extern "C" int64_t emu_find_all_fields(const uint8_t* blob, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals, int own_words) {
  return emu_find_all_fields2(blob, hay, len, out, cap_vals, own_words, 1);
}

// ---- twin of k_scan_trio_wave<K> (scan_fields_wave.hip): run(F) (byte(c_i) run(F)){K-1}, K = 2..4 ------------------------------
// Same steps as the kernel: bitmaps D and one per separator class of a 4096-byte window, links, owned span by one multiword
// addition, K - 1 hops per candidate, the loop for matches that share a run, and per end the K nearest bytes outside F in
// (previous word : this word).  Rows of K + 1 positions: start, the links, end.  Returns -(16 + reason) where the kernel would
// raise its fallback flag.
namespace {
int trio_shape_host(const ChainAux& c) {      // scan_fields_wave.hip trio_shape
  if ((c.nops & 1u) == 0 || c.nops < 3 || c.nops > 7 || c.restart_check) return 0;
  const int K = static_cast<int>((c.nops + 1) / 2);
  uint8_t sep[3] = {0, 0, 0};
  for (uint32_t k = 0; k < c.nops; k++) {
    if (c.op_kind[k] != ((k & 1u) ? kChainByte : kChainRun)) return 0;
    if (!(k & 1u)) { if (c.op_cls[k] != 0) return 0; continue; }
    const uint32_t q = c.op_cls[k];
    if (q == 0 || q >= c.ncls || c.cls_kind[q] == kClsSet || c.cls_kind[q] == kClsDigit || c.cls_lo[q] != c.cls_hi[q]) return 0;
    if (chain_class_has(c, 0, c.cls_lo[q])) return 0;
    sep[k >> 1] = c.cls_lo[q];
  }
  if (K >= 3) {
    int same = 0, pairs = 0;
    for (int i = 0; i < K - 1; i++) for (int j = i + 1; j < K - 1; j++) { pairs++; if (sep[i] == sep[j]) same++; }
    if (same == pairs) return K | 8;
    if (same) return 0;
  }
  return K;
}
int take_top(uint64_t& l, uint64_t& h) {
  if (h) { const int k = 63 - __builtin_clzll(h); h &= ~(1ull << k); return 64 + k; }
  if (l) { const int k = 63 - __builtin_clzll(l); l &= ~(1ull << k); return k; }
  return -1;
}
}  // namespace

extern "C" int emu_trio_shape(const uint8_t* blob) {
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(blob);
  if (h->magic != kBlobMagic || !(h->flags & kFlagChainOrdered) || (h->flags & kFlagChainBounded)) return 0;
  return trio_shape_host(*reinterpret_cast<const ChainAux*>(blob + h->aux_off + 256));   // K, | 8: one separator for all links
}

extern "C" int64_t emu_find_all_trio(const uint8_t* blob, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals, int own_words) {
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(blob);
  if (h->magic != kBlobMagic) return -1;
  if (!(h->flags & kFlagChainOrdered)) return -4;
  const ChainAux& ch = *reinterpret_cast<const ChainAux*>(blob + h->aux_off + 256);
  const int shape = trio_shape_host(ch);
  if (!shape) return -5;
  const int K = shape & 7;
  const bool eq = (shape & 8) != 0;
  if (own_words < 1 || own_words > 62) return -2;
  const int64_t tile_bytes = 64LL * own_words, pre = 64, N = 64LL * 64;
  const unsigned long long own_mask = ((own_words == 63 ? ~0ull : ((1ull << own_words) - 1ull)) << 1);
  std::vector<int64_t> res;
  const uint64_t ntiles = (len + tile_bytes - 1) / tile_bytes;
  for (uint64_t t = 0; t < ntiles; t++) {
    const int64_t tile_lo = static_cast<int64_t>(t) * tile_bytes, wlo = tile_lo - pre;
    uint64_t D[64] = {0}, C[3][64] = {{0}};
    for (int64_t b = 0; b < N; b++) {
      const int64_t p = wlo + b;
      if (p < 0 || p >= static_cast<int64_t>(len)) continue;
      if (chain_class_has(ch, 0, hay[p])) D[b >> 6] |= 1ull << (b & 63);
      for (int i = 0; i < K - 1; i++) if (chain_class_has(ch, ch.op_cls[2 * i + 1], hay[p])) C[i][b >> 6] |= 1ull << (b & 63);
    }
    unsigned long long PPd = 0, PPx = 0, ovf = 0;
    uint64_t LK[3][64], L[64], X[64], WS[64];
    for (int l = 0; l < 64; l++) {
      const uint64_t prev_top = l ? (D[l - 1] >> 63) : (D[0] >> 63);
      const uint64_t next_bot = l < 63 ? (D[l + 1] & 1ull) : 1ull;
      const uint64_t Dl = (D[l] << 1) | prev_top, Dr = (D[l] >> 1) | (next_bot << 63);
      L[l] = 0;
      for (int i = 0; i < K - 1; i++) { LK[i][l] = C[i][l] & Dl & Dr; L[l] |= LK[i][l]; }
      X[l] = D[l] | L[l];
      if (D[l] == ~0ull) PPd |= 1ull << l;
      if (X[l] == ~0ull) PPx |= 1ull << l;
    }
    for (int l = 0; l < 64; l++) {
      const uint64_t prev_top = l ? (D[l - 1] >> 63) : (D[0] >> 63);
      const uint64_t prev_ltop = l ? (L[l - 1] >> 63) : (L[0] >> 63);
      const uint64_t Dl = (D[l] << 1) | prev_top, Ll = (L[l] << 1) | prev_ltop;
      WS[l] = ((own_mask >> l) & 1ull) ? (D[l] & ~Dl & ~Ll) : 0ull;
    }
    auto add_words = [&](const uint64_t* P, const uint64_t* Q, uint64_t* S, unsigned long long PP) {
      unsigned long long GG = 0;
      for (int l = 0; l < 64; l++) {
        const unsigned __int128 s = static_cast<unsigned __int128>(P[l]) + Q[l];
        S[l] = static_cast<uint64_t>(s);
        if (s >> 64) GG |= 1ull << l;
      }
      const unsigned long long Pe = PP & ~GG;
      const unsigned long long recv = (Pe + (GG << 1)) ^ Pe;
      ovf |= GG | (Pe & recv);
      for (int l = 0; l < 64; l++) S[l] += (recv >> l) & 1ull;
    };
    uint64_t S[64], OWN[64] = {0};
    if (!eq) {
      add_words(X, WS, S, PPx);
      for (int l = 0; l < 64; l++) OWN[l] = X[l] & ~S[l];
    }
    auto hop = [&](const uint64_t* Q, uint64_t* R) {
      uint64_t T[64];
      for (int l = 0; l < 64; l++) T[l] = D[l] | Q[l];
      add_words(T, Q, R, PPd);
    };
    auto hops = [&](const uint64_t* Q, uint64_t* E) {
      uint64_t R[64], M[64];
      hop(Q, R);
      for (int i = 1; i < K - 1; i++) {
        for (int l = 0; l < 64; l++) M[l] = R[l] & LK[i][l];
        hop(M, R);
      }
      for (int l = 0; l < 64; l++) E[l] = R[l] & ~D[l];
    };
    const uint64_t* LA = LK[0];
    uint64_t Q0[64], E[64];
    bool chains = false;
    if (eq) {
      // one separator: the matches of a super-run are its fields K at a time from its start (what the fields kernel does)
      uint64_t R[64], Q[64], SEL[64] = {0}, Nx[64];
      hop(WS, R);                                                  // over the first run: its link, if it has one
      for (int l = 0; l < 64; l++) Q[l] = R[l] & L[l];
      for (int guard = 0; guard < 64; guard++) {
        hops(Q, E);
        bool any = false;
        for (int l = 0; l < 64; l++) { SEL[l] |= E[l]; Nx[l] = E[l] & L[l]; any = any || Nx[l]; }
        if (!any) break;
        hop(Nx, R);                                                // an end on a link: over the next match's first run
        for (int l = 0; l < 64; l++) Q[l] = R[l] & L[l];
        if (guard == 63) ovf |= 1ull << 63;
      }
      std::memcpy(E, SEL, sizeof E);
    } else {
      for (int l = 0; l < 64; l++) Q0[l] = LA[l] & OWN[l];
      hops(Q0, E);
      for (int l = 0; l < 64; l++) chains = chains || (E[l] & LA[l]);
    }
    if (chains) {
      uint64_t R[64], SEL[64] = {0}, Kb[64], H[64], Q[64];
      std::memcpy(R, E, sizeof R);
      for (int guard = 0; guard < 64; guard++) {
        for (int l = 0; l < 64; l++) Q[l] = R[l] & LA[l];
        hops(Q, Kb);
        for (int l = 0; l < 64; l++) { H[l] = R[l] & ~Kb[l]; SEL[l] |= H[l]; Q[l] = H[l] & LA[l]; }
        hops(Q, Kb);
        bool any = false;
        for (int l = 0; l < 64; l++) { R[l] &= ~(H[l] | Kb[l]); any = any || R[l]; }
        if (!any) break;
        if (guard == 63) ovf |= 1ull << 63;
      }
      std::memcpy(E, SEL, sizeof E);
    }
    uint32_t reason = (ovf >> 63) ? 1u : 0u;
    for (int l = 0; l < 64 && !reason; l++) {
      const uint64_t z = ~D[l], pz = l ? ~D[l - 1] : 0ull;
      const int64_t base = 64LL * l - 64;
      uint64_t ee = E[l];
      while (ee) {
        const int b = __builtin_ctzll(ee);
        ee &= ee - 1ull;
        uint64_t lo = pz, hi = b ? (z & ((1ull << b) - 1ull)) : 0ull;
        int pl[3] = {0, 0, 0};
        for (int i = K - 2; i >= 0; i--) pl[i] = take_top(lo, hi);
        const int ps = take_top(lo, hi);
        if (ps < 0) { reason |= 2u; break; }
        res.push_back(wlo + base + ps + 1);
        for (int i = 0; i < K - 1; i++) res.push_back(wlo + base + pl[i]);
        res.push_back(wlo + base + 64 + b);
      }
    }
    if (reason) return -(16 + static_cast<int64_t>(reason));
  }
  const int64_t n = static_cast<int64_t>(res.size());
  if (out && n <= cap_vals) std::memcpy(out, res.data(), n * sizeof(int64_t));
  return n;
}

// ---- twin of k_scan_fields_pers's literal mode (scan_fields_wave.hip lit_core) ------------------------------------------------
// One bitmap per distinct byte of the literal, occurrences O = AND_j (B_c(j) >> j) by the kernel's 128-bit funnel shifts (this
// lane's word and the next one's), starts restricted to the owned lanes, ends = O << m with the bits that leave a word arriving in
// the next lane's.  Rows: per end bit the highest start below it in (previous word : this word), as fields_rows does.
namespace {
int literal_shape_host(const ChainAux& c) {      // scan_fields_wave.hip literal_shape
  if (c.nops < 2 || c.nops > 63 || c.ncls < 2 || c.ncls > 4 || c.restart_check) return 0;
  uint8_t lit[64];
  for (uint32_t k = 0; k < c.nops; k++) {
    const uint32_t q = c.op_cls[k];
    if (c.op_kind[k] != kChainByte || q >= c.ncls || c.cls_kind[q] == kClsSet || c.cls_kind[q] == kClsDigit || c.cls_lo[q] != c.cls_hi[q] || c.cls_lo[q] > 0x7Fu) return 0;
    lit[k] = c.cls_lo[q];
  }
  for (uint32_t q = 0; q < c.ncls; q++) for (uint32_t r = q + 1; r < c.ncls; r++) if (c.cls_lo[q] == c.cls_lo[r]) return 0;
  for (uint32_t b = 1; b < c.nops; b++) {
    bool same = true;
    for (uint32_t i = 0; i < b && same; i++) same = lit[i] == lit[c.nops - b + i];
    if (same) return 0;
  }
  return static_cast<int>(c.ncls);
}
inline uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t k) { return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> (k & 31u)); }
}  // namespace

extern "C" int emu_literal_shape(const uint8_t* blob) {
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(blob);
  if (h->magic != kBlobMagic || !(h->flags & kFlagChainOrdered) || (h->flags & (kFlagChainBounded | kFlagChainSets))) return 0;
  return literal_shape_host(*reinterpret_cast<const ChainAux*>(blob + h->aux_off + 256));
}

extern "C" int64_t emu_find_all_literal(const uint8_t* blob, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals, int own_words, int pre_words) {
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(blob);
  if (h->magic != kBlobMagic) return -1;
  if (!(h->flags & kFlagChainOrdered)) return -4;
  const ChainAux& ch = *reinterpret_cast<const ChainAux*>(blob + h->aux_off + 256);
  const int NC = literal_shape_host(ch);
  if (!NC) return -5;
  if (own_words < 1 || pre_words < 1 || pre_words > 2 || own_words + pre_words > 63) return -2;
  const uint32_t m = ch.nops;
  const int64_t tile_bytes = 64LL * own_words, pre = 64LL * pre_words, N = 4096;
  const unsigned long long own_mask = ((1ull << own_words) - 1ull) << pre_words;
  std::vector<int64_t> res;
  const uint64_t ntiles = (len + tile_bytes - 1) / tile_bytes;
  for (uint64_t t = 0; t < ntiles; t++) {
    const int64_t wlo = static_cast<int64_t>(t) * tile_bytes - pre;
    uint32_t w0[4][64], w1[4][64];
    std::memset(w0, 0, sizeof w0); std::memset(w1, 0, sizeof w1);
    for (int64_t b = 0; b < N; b++) {
      const int64_t p = wlo + b;
      if (p < 0 || p >= static_cast<int64_t>(len)) continue;
      for (int c = 0; c < NC; c++) if (hay[p] == ch.cls_lo[c]) { if (b & 32) w1[c][b >> 6] |= 1u << (b & 31); else w0[c][b >> 6] |= 1u << (b & 31); }
    }
    uint32_t o0[64], o1[64];
    for (int l = 0; l < 64; l++) {
      uint32_t a0 = ~0u, a1 = ~0u;
      for (uint32_t j = 0; j < m; j++) {
        const uint32_t c = static_cast<uint32_t>((j < 32u ? ch.cls2_lo >> (2u * j) : ch.cls2_hi >> (2u * (j - 32u))) & 3ull);
        const uint32_t h0 = w0[c][l], h1 = w1[c][l], n0 = l < 63 ? w0[c][l + 1] : 0u, n1 = l < 63 ? w1[c][l + 1] : 0u;
        uint32_t r0, r1;
        if (j < 32u) { r0 = alignbit(h1, h0, j); r1 = alignbit(n0, h1, j); }
        else { r0 = alignbit(n0, h1, j - 32u); r1 = alignbit(n1, n0, j - 32u); }
        a0 &= r0; a1 &= r1;
      }
      const bool own = (own_mask >> l) & 1ull;
      o0[l] = own ? a0 : 0u; o1[l] = own ? a1 : 0u;
    }
    for (int l = 0; l < 64; l++) {
      const uint32_t p0 = l ? o0[l - 1] : 0u, p1 = l ? o1[l - 1] : 0u;
      uint32_t e0, e1;
      if (m < 32u) { e0 = alignbit(o0[l], p1, 32u - m); e1 = alignbit(o1[l], o0[l], 32u - m); }
      else if (m == 32u) { e0 = p1; e1 = o0[l]; }
      else { e0 = alignbit(p1, p0, 64u - m); e1 = alignbit(o0[l], p1, 64u - m); }
      const uint32_t b0 = o0[l], b1 = o1[l], pb0 = p0, pb1 = p1;
      const uint32_t lane64 = static_cast<uint32_t>(l) << 6;
      auto emit = [&](uint32_t s, uint32_t e) { res.push_back(wlo + s); res.push_back(wlo + e); };
      {
        const uint32_t tp = umin(ffbh_raw(pb1) | 32u, ffbh_raw(pb0) | 64u);
        uint32_t xx = e0;
        while (xx) {
          const uint32_t b = static_cast<uint32_t>(__builtin_ctz(xx));
          xx &= xx - 1u;
          const uint32_t d = umin(ffbh_raw(b0 & ((1u << b) - 1u)), tp);
          emit((lane64 + 31u - d) & 0xFFFFu, lane64 + b);
        }
      }
      {
        const uint32_t tp = umin(umin(ffbh_raw(b0) | 32u, ffbh_raw(pb1) | 64u), ffbh_raw(pb0) | 96u);
        uint32_t xx = e1;
        while (xx) {
          const uint32_t b = static_cast<uint32_t>(__builtin_ctz(xx));
          xx &= xx - 1u;
          const uint32_t d = umin(ffbh_raw(b1 & ((1u << b) - 1u)), tp);
          emit((lane64 + 63u - d) & 0xFFFFu, lane64 + 32u + b);
        }
      }
    }
  }
  const int64_t n = static_cast<int64_t>(res.size());
  if (out && n <= cap_vals) std::memcpy(out, res.data(), n * sizeof(int64_t));
  return n;
}
