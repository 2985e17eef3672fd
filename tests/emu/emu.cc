// TEST INFRASTRUCTURE ONLY — host emulation of the device lane walks.
//
// Compiles coregex_amd/csrc/device/walk.hpp (the exact per-lane functions the HIP kernels
// instantiate) for the CPU and runs them tile by tile, lane by lane, in the order the kernel's rank
// computation imposes (tile-major, lane-major, emission order).  It lets the CPU-only test tier check
// chunk ownership, sync-byte logic and the eager tables against the oracle without a GPU.  It is NOT
// a fallback: nothing in coregex_amd/ links or loads it, and it is built from tests/ only.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../coregex_amd/csrc/device/scan_dfa.h"
#include "../../coregex_amd/csrc/device/walk.hpp"
#include "../../coregex_amd/csrc/device/bt.hpp"

using namespace cxgdev;

namespace {
struct HostMem {
  const uint8_t* g;   // hay + tile_lo
  int32_t lim;
  const uint64_t* bits = nullptr;   // digit bitmap of the staged bytes (flat walk)
  uint64_t digits(int32_t w) const { return bits[w]; }
  const uint64_t* cbits = nullptr;  // Teddy candidate bitmap
  uint64_t cands(int32_t w) const { return cbits[w]; }
  int32_t bitmap_limit() const { return lim; }
  uint32_t byte(int32_t r) const { return g[r]; }
  uint32_t dword(int32_t r) const { uint32_t v; std::memcpy(&v, g + r, 4); return v; }
  int32_t wide_limit(int32_t x) const { return x < lim ? x : lim; }
};
struct VecSink {
  std::vector<int64_t>* out;
  int64_t origin;
  void emit(int32_t s, int32_t e) { out->push_back(origin + s); out->push_back(origin + e); }
};
}  // namespace

extern "C" int64_t emu_find_all(const uint8_t* blob, const uint8_t* hay, uint64_t len, int chunk, int64_t* out,
                                int64_t cap_vals, int flat) {
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(blob);
  if (h->magic != kBlobMagic) return -1;
  const uint8_t* info = blob + h->info_off;
  std::vector<int64_t> res;
  const int lanes = kThreads;
  const uint64_t tile_bytes = static_cast<uint64_t>(lanes) * chunk;
  const uint64_t ntiles = (len + tile_bytes - 1) / tile_bytes;
  for (uint64_t t = 0; t < ntiles; t++) {
    const uint64_t tile_lo = t * tile_bytes;
    const uint64_t remaining = len - tile_lo;
    const int32_t rend = remaining > 0x7FFF0000ull ? 0x7FFF0000 : static_cast<int32_t>(remaining);
    const int32_t stage = rend < static_cast<int32_t>(tile_bytes) + kHalo ? rend : static_cast<int32_t>(tile_bytes) + kHalo;
    HostMem m{hay + tile_lo, stage};
    std::vector<uint64_t> bits((static_cast<size_t>(stage) + 63) / 64 + 2, 0);
    for (int32_t k = 0; k < stage; k++)
      if (is_digit(hay[tile_lo + k])) bits[k >> 6] |= 1ull << (k & 63);
    m.bits = bits.data();
    std::vector<uint64_t> cbits;
    std::vector<uint16_t> cpos; std::vector<uint8_t> clen; std::vector<uint32_t> cexcl;
    VecSink sink{&res, static_cast<int64_t>(tile_lo)};
    for (int lane = 0; lane < lanes; lane++) {
      const int32_t c0 = lane * chunk, c1 = c0 + chunk;
      const bool at_origin = tile_lo == 0 && lane == 0;
      if (h->kind == kKindDigit) {
        DfaView f{blob + h->fwd_off, 256, h->fwd_start, h->fwd_first_accept};
        if (flat == 2) {
          // candidate-list kernel (scan_digit_list.hip): phases B and C once per tile, D per lane
          if (!(h->flags & kFlagFastDigit)) return -4;
          const uint8_t* sfl = blob + h->aux_off;
          if (lane == 0) {
            cpos.clear(); clen.clear(); cexcl.assign(lanes + 1, 0);
            for (int32_t pos = 0; pos < stage; pos++) {
              const bool dg = is_digit(m.byte(pos));
              const bool prev = (pos > 0 || tile_lo > 0) ? is_digit(m.byte(pos - 1)) : false;
              if (dg && !prev) cpos.push_back(static_cast<uint16_t>(pos));
            }
            for (int l = 0; l <= lanes; l++) {          // candidates below lane l's chunk start
              uint32_t n = 0;
              while (n < cpos.size() && static_cast<int32_t>(cpos[n]) < l * chunk) n++;
              cexcl[l] = n;
            }
            for (uint16_t c : cpos) {
              const int32_t e = verify_jump(m, f, sfl, c, rend);
              clen.push_back(static_cast<uint8_t>(e < 0 ? 0 : (e - c > 255 ? 255 : e - c)));
            }
          }
          lane_select(m, f, info, sfl, cpos.data(), clen.data(), static_cast<uint32_t>(cpos.size()), cexcl[lane], stage, c0,
                      c1, rend, at_origin, sink);
        } else if (flat) lane_digit_flat(m, f, info, (h->flags & kFlagRunSkip) != 0, c0, c1, rend, at_origin, sink);
        else lane_digit(m, f, info, (h->flags & kFlagRunSkip) != 0, c0, c1, rend, at_origin, sink);
      } else if (h->kind == kKindBidir) {
        DfaView f{blob + h->fwd_off, 256, h->fwd_start, h->fwd_first_accept};
        DfaView r{blob + h->rev_off, 256, h->rev_start, h->rev_first_accept};
        lane_bidir(m, f, r, info, c0, c1, rend, at_origin, sink);
      } else if (h->kind == kKindTeddy) {
        const uint8_t* aux = blob + h->aux_off;
        const TeddyAux* ax = reinterpret_cast<const TeddyAux*>(aux);
        TeddyView tv{reinterpret_cast<const uint16_t*>(aux + ax->ab_off), aux + ax->order_off, aux + ax->lens_off,
                     aux + ax->bucket_off, reinterpret_cast<const uint16_t*>(aux + ax->off_off), aux + ax->bytes_off, ax->nlits};
        if (lane == 0) {
          cbits.assign((static_cast<size_t>(stage) + 63) / 64 + 2, 0);
          for (int32_t k = 0; k < stage; k++)
            if (teddy_mask_at(m, tv, k, rend)) cbits[k >> 6] |= 1ull << (k & 63);
          m.cbits = cbits.data();
        }
        lane_teddy(m, tv, info, c0, c1, rend, at_origin, sink);
      } else {
        return -2;
      }
    }
  }
  const int64_t n = static_cast<int64_t>(res.size());
  if (out && n <= cap_vals) std::memcpy(out, res.data(), n * sizeof(int64_t));
  return n;
}

// FindAllSubmatch: spans from the bidirectional image, then the one-pass capture walk per row.
extern "C" int64_t emu_find_all_submatch(const uint8_t* span_blob, const uint8_t* cap_blob, const uint8_t* hay,
                                         uint64_t len, int chunk, int64_t* out, int64_t cap_vals) {
  std::vector<int64_t> spans(1024);
  int64_t n = emu_find_all(span_blob, hay, len, chunk, spans.data(), static_cast<int64_t>(spans.size()), 0);
  if (n < 0) return n;
  if (n > static_cast<int64_t>(spans.size())) {
    spans.resize(n);
    n = emu_find_all(span_blob, hay, len, chunk, spans.data(), n, 0);
  }
  if (reinterpret_cast<const BtHeader*>(cap_blob)->magic == kBtMagic) {     // not one-pass: backtracking per row (k_captures_bt)
    const BtHeader* bh = reinterpret_cast<const BtHeader*>(cap_blob);
    const uint32_t w = bh->nslots;
    const int64_t rows = n / 2;
    if (!out || rows * w > cap_vals) return rows * w;
    std::vector<uint32_t> visited(kBtVisitedWords);
    std::vector<uint64_t> stack(kBtStackEntries);
    for (int64_t i = 0; i < rows; i++) {
      int64_t* row = out + i * w;
      row[0] = spans[2 * i]; row[1] = spans[2 * i + 1];
      std::fill(visited.begin(), visited.end(), 0u);
      // the kernel's two tiers: small scratch first (k_captures_bt_lds), rows that do not fit again with the large one
      uint32_t rc = bt_captures<false>(bh, hay, row, w, visited.data(), stack.data(), kBtSmallVisited, kBtSmallStack);
      if (rc == 1u) {
        std::fill(visited.begin(), visited.end(), 0u);
        rc = bt_captures<false>(bh, hay, row, w, visited.data(), stack.data());
      }
      if (rc) return -3 - static_cast<int64_t>(rc);
    }
    return rows * w;
  }
  const CapHeader* ch = reinterpret_cast<const CapHeader*>(cap_blob);
  CapView cv{cap_blob + ch->next_off, cap_blob + ch->maskid_off, cap_blob + ch->fin_off,
             reinterpret_cast<const uint32_t*>(cap_blob + ch->masks_off), ch->n_entries, ch->start_entry};
  const uint32_t w = ch->nslots;
  const int64_t rows = n / 2;
  if (!out || rows * w > cap_vals) return rows * w;
  for (int64_t i = 0; i < rows; i++) {
    int64_t* row = out + i * w;
    row[0] = spans[2 * i]; row[1] = spans[2 * i + 1];
    if (!capture_walk(cv, hay, row, w)) return -3;
  }
  return rows * w;
}

// The backtracking capture pass alone (k_captures_bt_lds, then k_captures_bt for the rows it leaves), rows = spans from any span
// twin — the transducer's for programs with assertions, which have no table-walking span image.  [hay_lo, hay_hi): what the
// device buffer covers (assertion states read the bytes around a position; outside counts as a line break).
extern "C" int64_t emu_captures_bt(const uint8_t* cap_blob, const uint8_t* hay, uint64_t len, const int64_t* spans, int64_t nrows, int64_t* out) {
  const BtHeader* bh = reinterpret_cast<const BtHeader*>(cap_blob);
  if (bh->magic != kBtMagic) return -1;
  const uint32_t w = bh->nslots;
  std::vector<uint32_t> visited(kBtVisitedWords);
  std::vector<uint64_t> stack(kBtStackEntries);
  for (int64_t i = 0; i < nrows; i++) {
    int64_t* row = out + i * w;
    row[0] = spans[2 * i]; row[1] = spans[2 * i + 1];
    std::fill(visited.begin(), visited.end(), 0u);
    uint32_t rc = bt_captures<true>(bh, hay, row, w, visited.data(), stack.data(), kBtSmallVisited, kBtSmallStack, static_cast<int64_t>(0), static_cast<int64_t>(len));
    if (rc == 1u) {
      std::fill(visited.begin(), visited.end(), 0u);
      rc = bt_captures<true>(bh, hay, row, w, visited.data(), stack.data(), kBtVisitedWords, kBtStackEntries, static_cast<int64_t>(0), static_cast<int64_t>(len));
    }
    if (rc) return -3 - static_cast<int64_t>(rc);
  }
  return nrows * w;
}

// Fourth-generation digit kernel (scan_digit_chain.hip), tile by tile: reversed class bitmaps, the chain
// evaluated with chain_eval_seq, survivors verified with verify_jump, ownership by segment start, greedy
// FindAll order.  Returns -5 when a tile would raise the "rerun with the flat kernel" flag.
namespace {
struct PlainMem {
  const uint8_t* g;
  uint32_t byte(int32_t r) const { return g[r]; }
  uint64_t digits(int32_t) const { return 0; }
  int32_t bitmap_limit() const { return 0; }
};
}  // namespace

extern "C" int64_t emu_find_all_chain(const uint8_t* blob, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals,
                                      int tile_bytes, int halo_bytes) {
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(blob);
  if (h->magic != kBlobMagic || h->kind != kKindDigit) return -1;
  if ((h->flags & (kFlagFastDigit | kFlagChain)) != (kFlagFastDigit | kFlagChain)) return -4;
  const uint8_t* info = blob + h->info_off;
  const uint8_t* sfl = blob + h->aux_off;
  const ChainAux& ch = *reinterpret_cast<const ChainAux*>(blob + h->aux_off + 256);
  DfaView f{blob + h->fwd_off, 256, h->fwd_start, h->fwd_first_accept};
  const int nwd = (tile_bytes + halo_bytes) / 64, NW = nwd + 1;
  const int64_t N = 64LL * NW;
  std::vector<int64_t> res;
  const uint64_t ntiles = (len + tile_bytes - 1) / tile_bytes;
  for (uint64_t t = 0; t < ntiles; t++) {
    const uint64_t tile_lo = t * static_cast<uint64_t>(tile_bytes);
    const uint64_t remaining = len - tile_lo;
    const int32_t rend = remaining > 0x7FFF0000ull ? 0x7FFF0000 : static_cast<int32_t>(remaining);
    const int32_t stage = rend < tile_bytes + halo_bytes ? rend : tile_bytes + halo_bytes;
    const uint8_t* g = hay + tile_lo;
    std::vector<std::vector<uint64_t>> cls(ch.ncls, std::vector<uint64_t>(NW, 0));
    for (int32_t p = 0; p < stage; p++)
      for (uint32_t c = 0; c < ch.ncls; c++)
        if (chain_class_has(ch, static_cast<int>(c), g[p])) { const int64_t i = N - 1 - p; cls[c][i >> 6] |= 1ull << (i & 63); }
    std::vector<const uint64_t*> cp;
    for (auto& v : cls) cp.push_back(v.data());
    std::vector<uint64_t> G(NW), tmp(NW);
    chain_eval_seq(ch, cp.data(), NW, G.data(), tmp.data());
    const bool complete = (h->flags & kFlagChainComplete) != 0;
    std::vector<uint64_t> U(NW, 0);
    for (auto& v : cls) for (int w = 0; w < NW; w++) U[w] |= v[w];
    bool halo_sync = stage == rend;
    for (int32_t p = tile_bytes - 1; p < stage && !halo_sync; p++) {
      const int64_t i = N - 1 - p;
      halo_sync = complete ? !((U[i >> 6] >> (i & 63)) & 1) : (info[g[p]] & kInfoSync) != 0;
    }
    if (!halo_sync) return -5;
    PlainMem m{g};
    int32_t cur_end = -1;
    for (int32_t c = 0; c < stage; c++) {                          // ascending position
      const int64_t i = N - 1 - c;
      const bool dg = is_digit(g[c]);
      const bool prev = (c > 0 || tile_lo > 0) ? is_digit(g[c - 1]) : false;
      if (!(dg && !prev)) continue;
      if (!((G[i >> 6] >> (i & 63)) & 1)) continue;               // pruned by the chain
      int32_t e = -1;
      bool walked = false;
      if (complete) {
        const int32_t ie = chain_walk_end(ch, cp.data(), static_cast<int32_t>(i));
        if (ie >= 0 && (N - 1 - ie < stage || (N - 1 - ie == stage && stage == rend))) { e = static_cast<int32_t>(N - 1 - ie); walked = true; }
      }
      if (!walked) e = verify_jump(m, f, sfl, c, rend);
      if (complete && e != verify_jump(m, f, sfl, c, rend)) return -6;      // the chain must agree with the DFA
      if (e < 0) continue;
      int32_t seg;
      if (complete) {
        const int32_t jz = rev_scan_up_zero(U.data(), static_cast<int32_t>(N - c), static_cast<int32_t>(N));
        // positions at or beyond `stage` read as zero bits but lie BELOW index N-stage, never above i: safe
        seg = jz < static_cast<int32_t>(N) ? static_cast<int32_t>(N - jz) : ((tile_lo == 0 || (info[g[-1]] & kInfoSync)) ? 0 : -1);
      } else {
        int32_t p = c - 1;
        while (p >= 0 && !(info[g[p]] & kInfoSync)) p--;
        seg = p >= 0 ? p + 1 : ((tile_lo == 0 || (info[g[-1]] & kInfoSync)) ? 0 : -1);
      }
      if (!(seg >= 0 && seg < tile_bytes)) continue;
      if (c >= cur_end) { res.push_back(static_cast<int64_t>(tile_lo) + c); res.push_back(static_cast<int64_t>(tile_lo) + e); cur_end = e; }
    }
  }
  const int64_t n = static_cast<int64_t>(res.size());
  if (out && n <= cap_vals) std::memcpy(out, res.data(), n * sizeof(int64_t));
  return n;
}

// ---------------------------------------------------------------------------------------------------------
// Sequential twin of scan_chain_wave.hip (sixth generation): everything from the class bitmaps — starts by
// the right-to-left chain on the reversed bitmap, ownership from the synchronising bytes (zA, zB], ends by
// the left-to-right chain on the forward bitmap, k-th start paired with k-th end.  Window = tile + halo
// bytes (a multiple of 64).  Returns the number of int64 values, or -(16 + reason) when a tile would raise
// the fallback flag (reason bits as in the kernel).
namespace {
struct MW {                       // little multiword helpers on forward or reversed bitmaps of nw words
  static bool get(const std::vector<uint64_t>& w, int64_t i) { return (w[i >> 6] >> (i & 63)) & 1; }
  static void set(std::vector<uint64_t>& w, int64_t i) { w[i >> 6] |= 1ull << (i & 63); }
};
}  // namespace

static int64_t chain6_impl(const uint8_t* blob, const uint8_t* bounds, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals,
                           int tile_bytes, int halo_bytes) {
  const ChainCaps* bnd = reinterpret_cast<const ChainCaps*>(bounds);   // on == 2: bounded repetition (scan_chain_wave.hip BND)
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(blob);
  if (h->magic != kBlobMagic) return -1;
  if (!(h->flags & kFlagChainOrdered)) return -4;
  const ChainAux& ch = *reinterpret_cast<const ChainAux*>(blob + h->aux_off + 256);
  const int NW = (tile_bytes + halo_bytes) / 64;
  const int64_t N = 64LL * NW;
  if (N != tile_bytes + halo_bytes) return -2;
  std::vector<int64_t> res;
  const uint64_t ntiles = (len + tile_bytes - 1) / tile_bytes;
  auto in_alpha = [&](uint32_t b) { for (uint32_t c = 0; c < ch.ncls; c++) if (chain_class_has(ch, static_cast<int>(c), b)) return true; return false; };
  for (uint64_t t = 0; t < ntiles; t++) {
    const uint64_t tile_lo = t * static_cast<uint64_t>(tile_bytes);
    const uint64_t remaining = len - tile_lo;
    const int64_t rend = static_cast<int64_t>(remaining);
    const int64_t stage = rend < N ? rend : N;
    const uint8_t* g = hay + tile_lo;
    // reversed bitmaps get one spare word below the window (index < 64 <-> bytes past the window: no class, any G),
    // which is what the kernel's `inject` amounts to at the end of input and harmless elsewhere (not owned)
    const int64_t Nr = N + 64;
    std::vector<std::vector<uint64_t>> F(ch.ncls, std::vector<uint64_t>(NW, 0)), R(ch.ncls, std::vector<uint64_t>(NW + 1, 0));
    for (int64_t p = 0; p < stage; p++)
      for (uint32_t c = 0; c < ch.ncls; c++)
        if (chain_class_has(ch, static_cast<int>(c), g[p])) { MW::set(F[c], p); MW::set(R[c], Nr - 1 - p); }
    // ---- ownership bounds
    int64_t zA = -1, zB = 1 << 20;
    uint32_t reason = 0;
    if (tile_lo > 0 && in_alpha(g[-1])) {
      zA = 1 << 20;
      for (int64_t p = 0; p < stage; p++) if (!in_alpha(g[p])) { zA = p; break; }
    }
    {
      bool found = false;
      for (int64_t p = tile_bytes - 1; p < stage; p++) if (!in_alpha(g[p])) { zB = p; found = true; break; }
      if (!found && stage != rend) { zB = -2; reason |= 1; }
    }
    // ---- starts: chain right to left on the reversed bitmap (walk.hpp chain_eval_seq)
    std::vector<const uint64_t*> rp;
    for (auto& v : R) rp.push_back(v.data());
    std::vector<uint64_t> G(NW + 1), tmp(NW + 1);
    // chain_eval_seq assumes "nothing required after the chain" beyond the window; at the end of input exactly
    // at the window edge that is what the kernel injects, and inside the window class bits are zero past `stage`.
    chain_eval_seq(ch, rp.data(), NW + 1, G.data(), tmp.data());
    std::vector<uint64_t> S(NW, 0);
    const bool lead_run = ch.op_kind[0] == kChainRun;
    const int lc = ch.op_cls[0];
    for (int64_t p = 0; p < stage; p++) {
      if (!MW::get(G, Nr - 1 - p)) continue;
      if (lead_run) {
        const bool prev = (p > 0 || tile_lo > 0) ? chain_class_has(ch, lc, g[p - 1]) : false;
        if (!chain_class_has(ch, lc, g[p]) || prev) continue;
      }
      if (p > zA && p <= zB) MW::set(S, p);
    }
    // ---- ends: chain left to right on the forward bitmap
    std::vector<uint64_t> M = S;
    bool cout = false;
    for (uint32_t k = 0; k < ch.nops; k++) {
      const std::vector<uint64_t>& C = F[ch.op_cls[k]];
      bool co = false;
      if (ch.op_kind[k] == kChainByte) {
        uint64_t carry = 0;
        for (int w = 0; w < NW; w++) { const uint64_t nx = M[w] >> 63; M[w] = (M[w] << 1) | carry; carry = nx; }
        co = carry != 0;
      } else {
        uint64_t carry = 0;
        for (int w = 0; w < NW; w++) {
          const uint64_t a = M[w], b = C[w];
          uint64_t s = a + b; uint64_t c1 = s < a;
          const uint64_t s2 = s + carry; c1 |= (s2 < s);
          M[w] = s2 & ~b; carry = c1;
        }
        co = carry != 0;
      }
      if (k + 1 == ch.nops) cout = co; else if (co) reason |= 2;
    }
    std::vector<int64_t> sp, ep;
    for (int64_t p = 0; p < N; p++) { if (MW::get(S, p)) sp.push_back(p); if (MW::get(M, p)) ep.push_back(p); }
    if (cout) ep.push_back(N);
    if (sp.size() != ep.size()) reason |= 4;
    if (ch.restart_check && lead_run)          // a match ending inside a run of the first class: the kernel hands the scan over
      for (int64_t e : ep)
        if (e > 0 && e < stage && chain_class_has(ch, lc, g[e]) && chain_class_has(ch, lc, g[e - 1])) reason |= 64;
    if (reason) return -(16 + static_cast<int64_t>(reason));
    if (bnd && bnd->on == 2) {                   // filter the rows of the unbounded surrogate by field length
      std::vector<int64_t> fs, fe;
      const uint32_t nfields = bnd->nruns + 1u;
      for (size_t i = 0; i < sp.size(); i++) {
        int64_t s = sp[i], pos = sp[i];
        const int64_t e = ep[i];
        bool valid = true;
        for (uint32_t f = 0; f < nfields; f++) {
          int64_t fend = pos;                     // the field: a run of class 0 (every run of these chains uses class 0)
          while (fend < e && chain_class_has(ch, 0, g[fend])) fend++;
          if (f + 1 == nfields) fend = e;
          int64_t flen = fend - pos;
          const int64_t mn = bnd->src[f], mx = bnd->src[8 + f];
          if (f == 0 && mx != 0 && flen > mx) { s = fend - mx; flen = mx; }
          if (f + 1 == nfields && mx != 0 && flen > mx) return -(16 + 64);   // FindAll would resume inside the run
          if (flen < mn || (mx != 0 && flen > mx && f + 1 != nfields)) valid = false;
          pos = fend + 1;
        }
        if (valid) { fs.push_back(s); fe.push_back(e); }
      }
      sp.swap(fs); ep.swap(fe);
    }
    int64_t cur_end = -1;
    for (size_t i = 0; i < sp.size(); i++)
      if (sp[i] >= cur_end) { res.push_back(static_cast<int64_t>(tile_lo) + sp[i]); res.push_back(static_cast<int64_t>(tile_lo) + ep[i]); cur_end = ep[i]; }
  }
  const int64_t n = static_cast<int64_t>(res.size());
  if (out && n <= cap_vals) std::memcpy(out, res.data(), n * sizeof(int64_t));
  return n;
}

extern "C" int64_t emu_find_all_chain6(const uint8_t* blob, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals,
                                       int tile_bytes, int halo_bytes) {
  return chain6_impl(blob, nullptr, hay, len, out, cap_vals, tile_bytes, halo_bytes);
}
extern "C" int64_t emu_find_all_chain6_bounded(const uint8_t* blob, const uint8_t* bounds40, const uint8_t* hay, uint64_t len, int64_t* out,
                                               int64_t cap_vals, int tile_bytes, int halo_bytes) {
  return chain6_impl(blob, bounds40, hay, len, out, cap_vals, tile_bytes, halo_bytes);
}

// ---------------------------------------------------------------------------------------------------------
// Sequential twin of scan_teddy_wave.hip: three-byte fingerprint candidates, ownership (zA, zB] from the
// synchronising bytes, exact verification (buckets low to high, ids ascending), FindAll order inside the tile.
// Returns -(16 + reason) when a tile would raise the fallback flag.
extern "C" int64_t emu_find_all_teddy_wave(const uint8_t* blob, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals,
                                           int tile_bytes, int halo_bytes) {
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(blob);
  const bool verify_dfa = h->magic == kBlobMagic && h->kind == kKindBidir && (h->flags & kFlagPrefixLiteral);   // prefix literal + anchored DFA
  if (h->magic != kBlobMagic || (h->kind != kKindTeddy && !verify_dfa)) return -1;
  const uint8_t* info = blob + h->info_off;
  const uint8_t* aux = blob + h->aux_off;
  const TeddyAux* ax = reinterpret_cast<const TeddyAux*>(aux);
  const uint8_t* dfa = aux + ax->dfa_off;
  TeddyView tv{reinterpret_cast<const uint16_t*>(aux + ax->ab_off), aux + ax->order_off, aux + ax->lens_off,
               aux + ax->bucket_off, reinterpret_cast<const uint16_t*>(aux + ax->off_off), aux + ax->bytes_off, ax->nlits};
  uint32_t T[256];
  for (int b = 0; b < 256; b++) T[b] = tv.ab[b] | ((info[b] & kInfoSync) ? 0x1000000u : 0u);
  const bool fold = (ax->looks & kTeddyFold) != 0u;              // case-insensitive set: lower-case literals, a letter matches both cases
  for (uint32_t id = 0; id < tv.nlits; id++) {
    const uint32_t b3 = tv.bytes[tv.off[id] + 2];
    T[b3] |= 0x10000u << tv.bucket[id];
    if (fold && b3 >= 'a' && b3 <= 'z') T[b3 ^ 0x20u] |= 0x10000u << tv.bucket[id];
  }
  const int64_t N = tile_bytes + halo_bytes;
  std::vector<int64_t> res;
  const uint64_t ntiles = (len + tile_bytes - 1) / tile_bytes;
  for (uint64_t t = 0; t < ntiles; t++) {
    const uint64_t tile_lo = t * static_cast<uint64_t>(tile_bytes);
    const int64_t rend = static_cast<int64_t>(len - tile_lo);
    const int64_t stage = rend < N ? rend : N;
    const uint8_t* g = hay + tile_lo;
    auto is_sync = [&](int64_t p) { return (T[g[p]] & 0x1000000u) != 0; };
    int64_t zA = -1, zB = 1 << 20;
    uint32_t reason = 0;
    if (tile_lo > 0 && !is_sync(-1)) {
      zA = 1 << 20;
      for (int64_t p = 0; p < stage; p++) if (is_sync(p)) { zA = p; break; }
    }
    {
      bool found = false;
      for (int64_t p = tile_bytes - 1; p < stage; p++) if (is_sync(p)) { zB = p; found = true; break; }
      if (!found && stage != rend) { zB = -2; reason |= 1; }
    }
    std::vector<int64_t> cand;
    for (int64_t p = 0; p + 2 < stage || (p + 2 < rend && p < stage); p++) {
      if (p >= stage) break;
      const uint32_t b1 = (p + 1 < rend) ? g[p + 1] : 0u, b2 = (p + 2 < rend) ? g[p + 2] : 0u;
      // the kernel sees zeros past the window/input: entries of byte 0 (no literal contains it in these tests)
      const uint32_t e1 = (p + 1 < N) ? T[b1] : 0u, e2 = (p + 2 < N) ? T[b2] : 0u;
      if ((T[g[p]] & 0xFFu) & ((e1 >> 8) & 0xFFu) & ((e2 >> 16) & 0xFFu)) if (p > zA && p <= zB) cand.push_back(p);
    }
    if (cand.size() > 256) reason |= 8;
    if (reason) return -(16 + static_cast<int64_t>(reason));
    int64_t cur_end = -1;
    for (int64_t c : cand) {
      uint32_t mask = (T[g[c]] & 0xFFu) & ((T[g[c + 1]] >> 8) & 0xFFu) & ((T[g[c + 2]] >> 16) & 0xFFu);
      int64_t mlen = 0;
      while (mask && !mlen) {
        const uint32_t bk = static_cast<uint32_t>(__builtin_ctz(mask));
        mask &= mask - 1;
        for (uint32_t k = 0; k < tv.nlits && !mlen; k++) {
          const uint32_t id = tv.order[k];
          if (tv.bucket[id] != bk) continue;
          const int64_t ln = tv.lens[id];
          if (c + ln > rend) continue;
          const uint8_t* lit = tv.bytes + tv.off[id];
          int64_t q = 0;
          while (q < ln && (g[c + q] == lit[q] || (fold && lit[q] >= 'a' && lit[q] <= 'z' && (g[c + q] | 0x20u) == lit[q]))) q++;
          if (q == ln) mlen = ln;
        }
      }
      if (mlen && !verify_dfa && (ax->looks & 0xFFFFu)) {    // literals between assertions: both must hold around the occurrence
        const int pb = (tile_lo + c) > 0 ? g[c - 1] : -1;
        const int nb = c + mlen < rend ? (c + mlen < N ? g[c + mlen] : -2) : -1;
        if (nb == -2) return -(16 + 32);
        if (!teddy_look_holds(ax->looks & 0xFFu, pb, g[c]) || !teddy_look_holds((ax->looks >> 8) & 0xFFu, g[c + mlen - 1], nb)) mlen = 0;
      }
      if (mlen && verify_dfa) {                  // the occurrence of the prefix is extended by the anchored DFA (window bytes only)
        uint32_t q = ax->dfa_start;
        int64_t last = -1, i = c;
        const int64_t lim = rend < N ? rend : N;
        for (;; i++) {
          if (q >= ax->dfa_first_accept) last = i;
          if (i >= lim) break;
          q = dfa[q * 256 + g[i]];
          if (q == 0) break;
        }
        if (q != 0 && i >= N && rend > N) return -(16 + 32);   // still alive at the window edge: the kernel hands the scan over
        mlen = last > c ? last - c : 0;
      }
      if (mlen && c >= cur_end) { res.push_back(static_cast<int64_t>(tile_lo) + c); res.push_back(static_cast<int64_t>(tile_lo) + c + mlen); cur_end = c + mlen; }
    }
  }
  const int64_t n = static_cast<int64_t>(res.size());
  if (out && n <= cap_vals) std::memcpy(out, res.data(), n * sizeof(int64_t));
  return n;
}

// Sequential twin of scan_charclass_wave.hip: starts/ends of member runs per window, the tile owns the runs that
// start in its first tile_bytes bytes, the end that closes a run begun in front of the tile is skipped.
extern "C" int64_t emu_find_all_charclass_wave(const uint8_t* blob, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals,
                                               int tile_bytes, int halo_bytes) {
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(blob);
  if (h->magic != kBlobMagic || h->kind != kKindCharClass) return -1;
  if (!(h->flags & kFlagCcRanges)) return -4;
  const CharClassAux* ax = reinterpret_cast<const CharClassAux*>(blob + h->aux_off);
  auto member = [&](uint32_t b) { bool in = false; for (uint32_t q = 0; q < ax->nr; q++) if (b >= ax->lo[q] && b <= ax->hi[q]) in = true; return in != (ax->neg != 0); };
  const int64_t N = tile_bytes + halo_bytes;
  std::vector<int64_t> res;
  const uint64_t ntiles = (len + tile_bytes - 1) / tile_bytes;
  if (ax->pairs) {
    // `Q[^Q]*Q`: the tile owns the occurrences of Q in its bytes; with B of them in front of the tile, its event of rank r is event
    // k = B + r of the haystack: even k opens row k / 2, odd k closes it behind the Q (scan_charclass_wave.hip, pairs)
    uint64_t B = 0;
    for (uint64_t t = 0; t < ntiles; t++) {
      const uint64_t tile_lo = t * static_cast<uint64_t>(tile_bytes);
      const uint64_t tile_hi = tile_lo + tile_bytes < len ? tile_lo + tile_bytes : len;
      uint64_t n = 0;
      for (uint64_t p = tile_lo; p < tile_hi; p++) {
        if (!member(hay[p])) continue;
        const uint64_t k = B + n++;
        const size_t row = static_cast<size_t>(k >> 1);
        if (res.size() < 2 * (row + 1)) res.resize(2 * (row + 1), -1);
        res[2 * row + (k & 1)] = static_cast<int64_t>(p) + static_cast<int64_t>(k & 1);
      }
      if (n > 1024) return -(16 + 8);
      B += n;
    }
    res.resize(2 * (B >> 1));                                      // an unpaired last Q opens nothing
    const int64_t n = static_cast<int64_t>(res.size());
    if (out && n <= cap_vals) std::memcpy(out, res.data(), n * sizeof(int64_t));
    return n;
  }
  for (uint64_t t = 0; t < ntiles; t++) {
    const uint64_t tile_lo = t * static_cast<uint64_t>(tile_bytes);
    const int64_t rend = static_cast<int64_t>(len - tile_lo);
    const int64_t stage = rend < N ? rend : N;
    const uint8_t* g = hay + tile_lo;
    const bool prev_member = tile_lo > 0 && member(g[-1]);
    // starts owned at [0, tile), exclusive ends owned at (0, tile]; row of the i-th start: B + i, of the j-th end:
    // B - open + j with B = starts in front of the tile (scan_charclass_wave.hip)
    std::vector<int64_t> S, E;
    for (int64_t p = 0; p <= tile_bytes && p < N; p++) {
      const bool m = p < stage && member(g[p]);
      const bool pm = p == 0 ? prev_member : (p - 1 < stage && member(g[p - 1]));
      if (m && !pm && p < tile_bytes) S.push_back(p);
      if (!m && pm && p >= 1) E.push_back(p);
    }
    if (S.size() > 1024 || E.size() > 1024) return -(16 + 8);
    const size_t open = (prev_member && stage > 0 && member(g[0])) ? 1 : 0;
    const size_t B = res.size() / 2;
    for (size_t i = 0; i < S.size(); i++) { res.push_back(static_cast<int64_t>(tile_lo) + S[i]); res.push_back(-1); }
    for (size_t j = 0; j < E.size(); j++) {
      const size_t row = B - open + j;
      if (row >= res.size() / 2) return -3;
      res[2 * row + 1] = static_cast<int64_t>(tile_lo) + E[j];
    }
  }
  for (size_t i = 1; i < res.size(); i += 2) if (res[i] < 0) return -3;   // every run found its end
  const int64_t n = static_cast<int64_t>(res.size());
  if (out && n <= cap_vals) std::memcpy(out, res.data(), n * sizeof(int64_t));
  return n;
}


// Sequential twin of scan_delim_wave.hip (`O [^E]+ E` / `O [^E]* E`): per tile the carry-less addition R + O' bit by bit, the
// correction bits and the kind; kinds chained over the tiles; starts and matched ends placed as the kernel places them (the
// i-th start of a tile is row B + i, its j-th end row B - open + j).  Returns the rows (2 values each), or -(16 + 8) when a
// tile holds more than 1023 starts or ends (the kernel's fallback flag).
extern "C" int64_t emu_find_all_delim(int open_byte, int close_byte, int plus, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals, int tile_bytes) {
  std::vector<int64_t> res;
  const uint64_t ntiles = (len + tile_bytes - 1) / tile_bytes;
  uint32_t carry = 0;                                               // entry carry of the tile
  for (uint64_t t = 0; t < ntiles; t++) {
    const uint64_t lo = t * static_cast<uint64_t>(tile_bytes);
    const int64_t n = static_cast<int64_t>(len - lo < static_cast<uint64_t>(tile_bytes) ? len - lo : static_cast<uint64_t>(tile_bytes));
    std::vector<uint8_t> O(n), E(n), S(n, 0), M(n, 0);
    for (int64_t i = 0; i < n; i++) {
      E[i] = hay[lo + i] == close_byte;
      O[i] = hay[lo + i] == open_byte && !(plus && lo + i + 1 < len && hay[lo + i + 1] == close_byte);
    }
    uint32_t c = 0;                                                 // carry-less: sum = R + O'
    int64_t fe = -1, fs = -1, fm = -1;
    bool anyO = false;
    for (int64_t i = 0; i < n; i++) {
      const uint32_t sum = (E[i] ? 0u : 1u) + O[i] + c;
      const uint32_t bit = sum & 1u;
      c = sum >> 1;
      S[i] = O[i] && !bit;
      M[i] = E[i] && bit;
      anyO = anyO || O[i];
      if (E[i] && fe < 0) fe = i;
      if (S[i] && fs < 0) fs = i;
      if (M[i] && fm < 0) fm = i;
    }
    const uint32_t g0 = c;                                          // (beyond the tile's bytes the kernel sees ones: the carry leaves through the top)
    const uint32_t kind = fe >= 0 ? g0 : (anyO ? 1u : 2u);
    if (carry) {                                                    // under a carry-in: the start below the first E is none, the first E is matched
      if (fs >= 0 && (fe < 0 || fs < fe)) S[fs] = 0;
      if (fe >= 0) M[fe] = 1;
    }
    size_t ns = 0, ne = 0;
    for (int64_t i = 0; i < n; i++) { ns += S[i]; ne += M[i]; }
    if (ns + 1 > 1024 || ne + 1 > 1024) return -(16 + 8);
    const size_t B = res.size() / 2;
    for (int64_t i = 0; i < n; i++) if (S[i]) { res.push_back(static_cast<int64_t>(lo) + i); res.push_back(-1); }
    size_t j = 0;
    for (int64_t i = 0; i < n; i++) if (M[i]) {
      const size_t row = B - carry + j++;
      if (row >= res.size() / 2) return -3;
      res[2 * row + 1] = static_cast<int64_t>(lo) + i + 1;
    }
    if (kind != 2u) carry = kind;
  }
  if (carry && !res.empty()) { res.pop_back(); res.pop_back(); }   // an opening without its E at the end of the haystack is no row
  for (size_t i = 1; i < res.size(); i += 2) if (res[i] < 0) return -3;
  const int64_t nv = static_cast<int64_t>(res.size());
  if (out && nv <= cap_vals) std::memcpy(out, res.data(), nv * sizeof(int64_t));
  return nv;
}
