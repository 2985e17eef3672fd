// TEST INFRASTRUCTURE ONLY — sequential twin of scan_fsm.hip (the FindAll-transducer kernel).
//
// Runs the very lane functions the kernel instantiates (coregex_amd/csrc/device/fsm.hpp) tile by tile, lane by lane:
// entry state by a warm-up walk over the previous chunk from the "any state" row, replay of the own chunk + walk-ahead,
// rows gathered in lane order, starts by the reverse DFA bounded by the previous row's end.  Geometry (tile, chunk) is
// a parameter so that the CPU tier can stress chunk and tile edges with tiny sizes.  Nothing in coregex_amd/ links it.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../coregex_amd/csrc/device/fsm.hpp"

using namespace cxgdev;

namespace {
template <int LOOK>
struct HostMem : FsmClassify<HostMem<LOOK>, LOOK> {
  int32_t last() const { const int64_t d = len - 1 - origin_abs; return d > 0x7FFF0000 ? 0x7FFF0000 : static_cast<int32_t>(d); }   // LOOK == 2: the haystack's last byte (fsm.hpp "End of text")
  const uint8_t* hay;   // whole haystack
  int64_t origin_abs;   // absolute position of the tile origin
  int64_t len;
  uint32_t outside;     // FsmHeader::outside_byte: what the kernel writes into its window around the haystack
  uint32_t byte(int32_t r) const { const int64_t p = origin_abs + r; return (p >= 0 && p < len) ? hay[p] : outside; }
  uint32_t dword(int32_t r) const { return byte(r) | (byte(r + 1) << 8) | (byte(r + 2) << 16) | (byte(r + 3) << 24); }
  template <int N> void below(int32_t e, uint32_t (&W)[N / 4 + 1]) const { for (int j = 0; j < N / 4 + 1; j++) W[j] = dword(e - (N + 1) + 4 * j); }   // fsm.hpp fsm_match_startN
};
struct LaneRows {
  int32_t end[kFsmLaneRowsMax];
  void set_end(uint32_t r, int32_t e) { end[r] = e; }
};
struct LaneEvents {
  uint16_t row[kFsmLaneEventsMax];
  void push(uint32_t k, uint32_t r) { row[k] = static_cast<uint16_t>(r); }
  uint32_t row_at(uint32_t k) const { return row[k]; }
};
FsmView view_of(const uint8_t* img) {
  const FsmHeader* h = reinterpret_cast<const FsmHeader*>(img);
  FsmView v;
  v.cls2 = img + h->cls_off;
  v.tab = img + h->tab_off;
  v.rev = img + h->rev_off;
  v.ncls2 = 2 * h->ncls;
  v.alias_lo = h->alias_lo; v.u_lo = h->u_lo; v.top_off = h->top_off;
  v.rev_start_off = h->rev_start_off; v.rev_accept_off = h->rev_accept_off; v.rev_text_col = h->rev_text_col; v.end_col = h->end_col; v.rev_dead = h->rev_off - static_cast<uint32_t>(sizeof(FsmHeader));
  v.create_lo = h->create_lo; v.rematch_lo = h->rematch_lo;
  v.mem = img + h->mem_off; v.row_shift = h->row_shift;
  v.knd = img + h->knd_off;
  {                                       // FsmView::lk16 (the kernels fill theirs in LDS): one table per thread is enough here
    static thread_local uint16_t lk[256];
    for (int b = 0; b < 256; b++) lk[b] = static_cast<uint16_t>(v.cls2[b] | (static_cast<uint32_t>(h->nk > 1 ? v.knd[b] : 0) << 8));
    v.lk16 = lk;
  }
  v.nk = h->nk;
  return v;
}
}  // namespace

// Returns the number of int64 values written (2 per match) or needed; -16 - reason when a tile would raise the
// fallback flag (reason 1: a lane's entry state did not collapse, 2: more than kFsmLaneRows rows in a chunk,
// 4: level stack overflow, 8: walk budget), -1 on a bad image.  stats (optional, 4 values): chunks, chunks whose entry
// needed the full warm-up, chunks whose entry set did not collapse, rows fixed against the previous row.
template <int LOOK>
static int64_t emu_fsm(const uint8_t* img, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals,
                       int tile, int chunk, int budget_bytes, uint64_t* stats, int dense) {
  const FsmHeader* h = reinterpret_cast<const FsmHeader*>(img);
  const FsmView v = view_of(img);
  std::vector<int64_t> res;
  const int lanes = tile / chunk;
  const uint64_t ntiles = (len + static_cast<uint64_t>(tile) - 1) / static_cast<uint64_t>(tile);
  int64_t prev_end = 0;                                   // absolute end of the previous row
  uint32_t cur_exit = 0;                                  // state at the end of the previous chunk
  uint64_t st[4] = {0, 0, 0, 0};
  for (uint64_t t = 0; t < ntiles; t++) {
    const uint64_t tile_lo = t * static_cast<uint64_t>(tile);
    const uint64_t remaining = len - tile_lo;
    const int32_t rend = remaining > 0x7FFF0000ull ? 0x7FFF0000 : static_cast<int32_t>(remaining);
    const int32_t budget = rend < tile + budget_bytes ? rend : tile + budget_bytes;
    const int32_t lowest = tile_lo > static_cast<uint64_t>(budget_bytes) ? -budget_bytes : -static_cast<int32_t>(tile_lo);
    HostMem<LOOK> m;
    m.hay = hay; m.origin_abs = static_cast<int64_t>(tile_lo); m.len = static_cast<int64_t>(len); m.outside = LOOK ? h->outside_byte : 0u;
    bool first_in_tile = true;
    for (int lane = 0; lane < lanes; lane++) {
      const int32_t c0 = lane * chunk, c1 = c0 + chunk;
      if (c0 >= rend) break;
      st[0]++;
      uint32_t entry = m.origin(v);                       // (only used at the haystack's first byte)
      if (tile_lo + static_cast<uint64_t>(c0) > 0) {
        // the kernel's policy: 16 bytes first, 64 bytes when the set has not collapsed by then
        const int64_t avail = static_cast<int64_t>(tile_lo) + c0;
        const int32_t w1 = static_cast<int32_t>(avail < 16 ? avail : 16), w2 = static_cast<int32_t>(avail < 64 ? avail : 64);
        entry = fsm_walk(v, m, v.top_off, c0 - w1, c0, (w1 % 4) == 0);
        if (entry >= v.u_lo && w2 > w1) { entry = fsm_walk(v, m, v.top_off, c0 - w2, c0, (w2 % 4) == 0); st[1]++; }
        if (entry >= v.u_lo) {
          // the set of possible states did not collapse.  The kernel then takes the true entry state from the chunk in
          // front (scan_fsm.hip: maps over the set's members + a chain through the unresolved chunks); the twin walks
          // in order and simply knows it.  The set must be listed (<= 8 members) and must hold the true state.
          st[2]++;
          if (fsm_member(v, entry, 0) == 0xFFFFu) return -16 - 1;
          bool found = false;
          for (uint32_t j = 0; j < static_cast<uint32_t>(kFsmMembers); j++) found = found || fsm_member(v, entry, j) == cur_exit;
          if (!found) return -3;
          entry = cur_exit;
        }
      }
      FsmLane L;
      if (dense) { L.max_rows = kFsmLaneRowsMax; L.max_events = kFsmLaneEventsMax; }   // the kernel's mode 2
      LaneRows rows;
      LaneEvents evs;
      if (h->depth <= 1 && chunk == kFsmSub && c1 <= rend && c1 <= budget) {   // the kernel's SHALLOW instantiation
        const int32_t cc[1] = {c0};
        FsmTraceS ts[1] = {{entry, 0u, 0u}};
        fsm_fast_shallow<1>(v, m, cc, ts);
        fsm_finish_shallow(v, m, ts[0], c0, rend, budget, L, rows);
      } else {
        fsm_replay(v, m, entry, c0, c1, rend, budget, L, rows, evs);
      }
      if (L.flags) return -16 - static_cast<int64_t>(L.flags << 1);
      cur_exit = fsm_canon(v, L.xc1);
      for (uint32_t r = 0; r < L.nrows; r++) {
        const int32_t e = rows.end[r];
        uint32_t over = 0;
        // the kernel knows the previous row's end only inside the tile; the tile's first row is walked without a bound —
        // one byte below the window, so that a reverse DFA still alive at the window's first byte reports `over` — and
        // checked afterwards.  With look-around a reverse step also reads the byte in front of its own: the walk stops
        // one byte earlier (scan_fsm.hip rev_lowest).
        const bool window_cut = static_cast<int64_t>(tile_lo) + lowest > 0;      // bytes exist in front of the window
        const int32_t rev_lowest = (LOOK && window_cut) ? lowest + 1 : lowest;
        const int32_t bound = first_in_tile ? (window_cut ? lowest - 1 : lowest) : static_cast<int32_t>(prev_end - static_cast<int64_t>(tile_lo));
        int32_t s = (v.rev_text_col == 0u || tile_lo != 0) ? fsm_match_start16(v, m, e, bound, rev_lowest, over)      // as the kernel chooses
                                                           : fsm_match_start(v, m, e, bound, rev_lowest, over, tile_lo == 0 ? 0 : kFsmNoStart);
        if (over || (first_in_tile && s != kFsmNoStart && static_cast<int64_t>(tile_lo) + s < prev_end)) {
          // the kernel's epilogue: the walk again, bounded by the previous row's end, bytes from HBM / L2
          st[3]++;
          over = 0;
          s = fsm_match_start(v, m, e, static_cast<int32_t>(prev_end - static_cast<int64_t>(tile_lo)), -static_cast<int32_t>(tile_lo), over, -static_cast<int32_t>(tile_lo));
        }
        if (over) return -16 - 8;
        if (s == kFsmNoStart) return -2;
        first_in_tile = false;
        res.push_back(static_cast<int64_t>(tile_lo) + s);
        res.push_back(static_cast<int64_t>(tile_lo) + e);
        prev_end = static_cast<int64_t>(tile_lo) + e;
      }
    }
  }
  if (stats) std::memcpy(stats, st, sizeof st);
  const int64_t n = static_cast<int64_t>(res.size());
  if (out && n > 0 && n <= cap_vals) std::memcpy(out, res.data(), static_cast<size_t>(n) * sizeof(int64_t));
  return n;
}


// Round 6: the SHALLOW instantiation's row derivation (scan_fsm.hip "S", fsm.hpp "Round 6").  Lanes of 64 bytes = two sub-chunks of
// kFsmSub; `tile / 64` owned lanes, then `budget_bytes / 64` tail lanes that only contribute their event bits; rows = the events of
// the owned lanes that no rematch follows, through the very functions the kernel calls (fsm_fast_shallow, fsm_first_is_r,
// fsm_lanes_succ_r, fsm_lane_ends).  Entry states by the kernel's policy (16 bytes of warm-up, then 64, then the true state).
struct HostTab {                                          // the direct section: rows of 256 bytes, then the property table
  const uint8_t* d;
  uint32_t at(uint32_t addr) const { return d[addr]; }
};
template <int LOOK, bool DIRECT>
static int64_t emu_fsm_shallow(const uint8_t* img, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals,
                               int tile, int budget_bytes, uint64_t* stats) {
  const FsmHeader* h = reinterpret_cast<const FsmHeader*>(img);
  const FsmView v = view_of(img);
  HostTab tab;
  tab.d = img + h->direct_off;
  const uint32_t prop = h->d_slots << 8;
  std::vector<int64_t> res;
  const int own = tile / 64, tail = budget_bytes / 64;
  const uint64_t ntiles = (len + static_cast<uint64_t>(tile) - 1) / static_cast<uint64_t>(tile);
  int64_t prev_end = 0;
  uint32_t tile_entry = 0;                                // canonical state at the tile's first byte
  uint64_t st[4] = {0, 0, 0, 0};
  for (uint64_t t = 0; t < ntiles; t++) {
    const uint64_t tile_lo = t * static_cast<uint64_t>(tile);
    const uint64_t remaining = len - tile_lo;
    const int32_t rend = remaining > 0x7FFF0000ull ? 0x7FFF0000 : static_cast<int32_t>(remaining);
    const int32_t win_end = tile + budget_bytes;
    const int32_t lowest = tile_lo > static_cast<uint64_t>(budget_bytes) ? -budget_bytes : -static_cast<int32_t>(tile_lo);
    HostMem<LOOK> m;
    m.hay = hay; m.origin_abs = static_cast<int64_t>(tile_lo); m.len = static_cast<int64_t>(len); m.outside = LOOK ? h->outside_byte : 0u;
    uint32_t T[64][4];
    uint64_t ne = 0, fr = 0;
    uint32_t cur = tile_entry, x_end = 0;
    int nact = 0;
    for (int l = 0; l < own + tail; l++) {
      const int32_t c0 = l * 64;
      for (int q = 0; q < 4; q++) T[l][q] = 0;
      if (c0 >= rend) break;
      nact = l + 1;
      for (int sb = 0; sb < 2; sb++) {
        const int32_t c = c0 + sb * kFsmSub;
        if (c < rend) st[0]++;
        const int32_t cc[1] = {c};
        FsmTraceS ts[1] = {{0u, 0u, 0u}};
        uint32_t entry = 0;
        if (DIRECT) {
          // k_scan_fsmd: 16 bytes from "any state", 64 when the set has not collapsed, then the host's rerun (reason 1)
          const bool origin = tile_lo + static_cast<uint64_t>(c) == 0, from_start = tile_lo + static_cast<uint64_t>(c) == static_cast<uint64_t>(kFsmSub);
          uint32_t e[1] = {h->d_top};
          const int32_t f16[1] = {c - 16};
          fsmd_walk_n<1>(m, tab, f16, 16, e);
          if (origin) e[0] = 0u;
          if (tab.at(prop + e[0]) & 0x80u) {
            st[1]++;
            const int32_t f64[1] = {from_start ? 0 : c - 64};
            e[0] = from_start ? 0u : h->d_top;
            fsmd_walk_n<1>(m, tab, f64, c - f64[0], e);
            if (tab.at(prop + e[0]) & 0x80u) { st[2]++; if (c < rend) return -16 - 1; }
          }
          entry = e[0];
          ts[0].x = entry;
          fsmd_chunk<1>(m, tab, cc, ts);
          if (c < rend) {
            const int32_t to = c + kFsmSub < rend ? c + kFsmSub : rend;
            uint32_t xx = entry & ~3u;
            for (int32_t i = c; i < to; i++) xx = tab.at(fsmd_addr(xx, m.byte(i), 0));
            if ((entry & ~3u) != cur * 4u) return -3;                             // the entry the warm-up found is the true state
            cur = xx >> 2;
          }
          x_end = ts[0].x;
        } else {
        entry = m.origin(v);
        if (tile_lo + static_cast<uint64_t>(c) > 0) {
          const int64_t avail = static_cast<int64_t>(tile_lo) + c;
          const int32_t w1 = static_cast<int32_t>(avail < 16 ? avail : 16), w2 = static_cast<int32_t>(avail < 64 ? avail : 64);
          entry = fsm_walk(v, m, v.top_off, c - w1, c, (w1 % 4) == 0);
          if (entry >= v.u_lo && w2 > w1) { entry = fsm_walk(v, m, v.top_off, c - w2, c, (w2 % 4) == 0); st[1]++; }
          if (entry >= v.u_lo) {
            st[2]++;
            if (c < rend && fsm_member(v, entry, 0) == 0xFFFFu) return -16 - 1;
            entry = cur;
          } else if (c < rend && fsm_canon(v, entry) != cur) return -3;          // a collapsed set holds the true state
        }
        ts[0].x = entry;
        fsm_fast_shallow<1>(v, m, cc, ts);
        if (c < rend) {                                                           // the true state behind the sub-chunk (behind the input's end inside it)
          const int32_t to = c + kFsmSub < rend ? c + kFsmSub : rend;
          cur = fsm_canon(v, fsm_walk(v, m, entry, c, to, false));
        }
        x_end = ts[0].x & ~3u;
        }
        const uint64_t k = ((static_cast<uint64_t>(ts[0].k1) << 32) | ts[0].k0) & fsm_valid_bits(rend - c);
        T[l][2 * sb] = static_cast<uint32_t>(k); T[l][2 * sb + 1] = static_cast<uint32_t>(k >> 32);
      }
      if (l == own - 1) tile_entry = cur;
      if (T[l][0] | T[l][1] | T[l][2] | T[l][3]) { ne |= 1ull << l; if (fsm_first_is_r(T[l])) fr |= 1ull << l; }
    }
    if (nact <= own - 1) tile_entry = cur;
    const uint64_t ln = fsm_lanes_succ_r(ne, fr);
    if (rend > win_end && ne != 0) {
      int top = 63;
      while (!((ne >> top) & 1ull)) top--;
      if (top < own && (DIRECT ? (tab.at(prop + x_end) & 0x7Fu) : fsm_u16(v.tab, x_end + v.ncls2 + 2u)) != 0u) return -16 - 8;   // a match pending past the window's end
    }
    bool first_in_tile = true;
    for (int l = 0; l < own && l < nact; l++) {
      uint32_t Er[4];
      fsm_lane_ends(T[l], static_cast<uint32_t>(ln >> l) & 1u, Er);
      for (int i = 0; i < 4; i++) {
        uint32_t xw = Er[3 - i];
        while (xw) {
          uint32_t q = 0;
          while (!((xw << q) & 0x80000000u)) q++;
          xw &= ~(0x80000000u >> q);
          const int32_t e = l * 64 + 16 * i + 1 + static_cast<int32_t>(q >> 1);
          uint32_t over = 0;
          const bool window_cut = static_cast<int64_t>(tile_lo) + lowest > 0;
          const int32_t rev_lowest = (LOOK && window_cut) ? lowest + 1 : lowest;
          const int32_t bound = first_in_tile ? (window_cut ? lowest - 1 : lowest) : static_cast<int32_t>(prev_end - static_cast<int64_t>(tile_lo));
          const FsmdRev R = {h->d_rstart, h->d_racc_lo, h->d_rdead};
          const bool n8 = ((tile_lo / static_cast<uint64_t>(tile)) & 1u) != 0;   // (the kernel takes 8 steps for tiles with many rows: the answer is the same, every other tile here)
          int32_t s = DIRECT ? (n8 ? fsmd_match_startN<8>(m, tab, R, e, bound, lowest, over) : fsmd_match_startN<16>(m, tab, R, e, bound, lowest, over))
                             : (v.rev_text_col == 0u || tile_lo != 0) ? (n8 ? fsm_match_startN<8>(v, m, e, bound, rev_lowest, over) : fsm_match_startN<16>(v, m, e, bound, rev_lowest, over))
                             : fsm_match_start(v, m, e, bound, rev_lowest, over, tile_lo == 0 ? 0 : kFsmNoStart);
          if (over || (first_in_tile && s != kFsmNoStart && static_cast<int64_t>(tile_lo) + s < prev_end)) {
            st[3]++;
            over = 0;
            const int32_t pb = static_cast<int32_t>(prev_end - static_cast<int64_t>(tile_lo));
            s = DIRECT ? fsmd_match_start(m, tab, R, e, pb, -static_cast<int32_t>(tile_lo), over)
                       : fsm_match_start(v, m, e, pb, -static_cast<int32_t>(tile_lo), over, -static_cast<int32_t>(tile_lo));
          }
          if (over) return -16 - 8;
          if (s == kFsmNoStart) return -2;
          first_in_tile = false;
          res.push_back(static_cast<int64_t>(tile_lo) + s);
          res.push_back(static_cast<int64_t>(tile_lo) + e);
          prev_end = static_cast<int64_t>(tile_lo) + e;
        }
      }
    }
  }
  if (stats) std::memcpy(stats, st, sizeof st);
  const int64_t n = static_cast<int64_t>(res.size());
  if (out && n > 0 && n <= cap_vals) std::memcpy(out, res.data(), static_cast<size_t>(n) * sizeof(int64_t));
  return n;
}

// k_scan_fsmd (direct mode) in the kernel's geometry; -1: the image has no direct section
extern "C" int64_t emu_find_all_fsm_direct(const uint8_t* img, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals,
                                           int tile, int budget_bytes, uint64_t* stats) {
  const FsmHeader* h = reinterpret_cast<const FsmHeader*>(img);
  if (h->magic != kFsmMagic || h->direct_off == 0u || h->depth > 1 || h->nk != 1 || tile % 64 != 0 || budget_bytes % 64 != 0 || budget_bytes <= 0 || (tile + budget_bytes) / 64 > 63) return -1;
  return emu_fsm_shallow<0, true>(img, hay, len, out, cap_vals, tile, budget_bytes, stats);
}

extern "C" int64_t emu_find_all_fsm(const uint8_t* img, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals,
                                    int tile, int chunk, int budget_bytes, uint64_t* stats, int dense) {
  const FsmHeader* h = reinterpret_cast<const FsmHeader*>(img);
  if (h->magic != kFsmMagic || chunk % 4 != 0 || tile % chunk != 0) return -1;
  // the kernel's SHALLOW instantiation in its own geometry (64-byte lanes, a tail of whole lanes)
  if (h->depth <= 1 && chunk == kFsmSub && tile % 64 == 0 && budget_bytes % 64 == 0 && budget_bytes > 0 && (tile + budget_bytes) / 64 <= 63)
    return h->end_col ? emu_fsm_shallow<2, false>(img, hay, len, out, cap_vals, tile, budget_bytes, stats)
         : h->nk > 1  ? emu_fsm_shallow<1, false>(img, hay, len, out, cap_vals, tile, budget_bytes, stats) : emu_fsm_shallow<0, false>(img, hay, len, out, cap_vals, tile, budget_bytes, stats);
  return h->end_col ? emu_fsm<2>(img, hay, len, out, cap_vals, tile, chunk, budget_bytes, stats, dense)
       : h->nk > 1  ? emu_fsm<1>(img, hay, len, out, cap_vals, tile, chunk, budget_bytes, stats, dense)
                    : emu_fsm<0>(img, hay, len, out, cap_vals, tile, chunk, budget_bytes, stats, dense);
}

// ---- round 3: the kernel's way of finding entry states without waiting (scan_fsm.hip "Maps instead of waits") -------------------
// For every tile: each 32-byte sub-chunk is a MAP (entry state -> end state): a constant when its warm-up set collapsed, else the
// end state for each listed member.  A lane folds its two sub-chunks, six Hillis-Steele levels compose the lanes (a constant
// absorbs what lies in front), the last lane's map is the tile's; a group composes its tiles' maps, and a group's entry comes
// from the maps of the groups in front, nearest first, until one with a known exit.  This twin performs exactly those
// compositions (same pair lists, same order, same canonical states) and compares every sub-chunk entry, tile exit and group
// entry it derives with the state a plain left-to-right walk is in at that point.  Returns the number of sub-chunks checked;
// -16 - 1: a set that is not listed (the kernel's fallback); -100 - k: a mismatch of kind k.
namespace {
struct EMap {                      // the kernel's (isc, cv, P[8])
  bool isc = false;
  uint32_t cv = 0;
  uint32_t P[kFsmMembers];         // member | end << 16; 0xFFFFFFFF: unused
  EMap() { for (auto& p : P) p = 0xFFFFFFFFu; }
};
uint32_t emap_look(const EMap& m, uint32_t x) {
  uint32_t r = 0xFFFFu;
  for (int j = 0; j < kFsmMembers; j++) if ((m.P[j] & 0xFFFFu) == x) r = m.P[j] >> 16;
  return r;
}
uint32_t emap_apply(const EMap& m, uint32_t x) { return m.isc ? m.cv : emap_look(m, x); }
EMap emap_then(const EMap& first, const EMap& second) {     // second o first (first lies in front)
  if (second.isc) return second;
  EMap r;
  if (first.isc) { r.isc = true; r.cv = emap_look(second, first.cv); return r; }
  for (int j = 0; j < kFsmMembers; j++) r.P[j] = first.P[j] == 0xFFFFFFFFu ? 0xFFFFFFFFu : ((first.P[j] & 0xFFFFu) | (emap_look(second, first.P[j] >> 16) << 16));
  return r;
}
}  // namespace

template <int LOOK>
static int64_t emu_fsm_maps(const uint8_t* img, const uint8_t* hay, uint64_t len, int tile, int tiles_per_group) {
  const FsmHeader* h = reinterpret_cast<const FsmHeader*>(img);
  const FsmView v = view_of(img);
  const int sub = kFsmSub, lanes = tile / (2 * sub);
  const uint64_t ntiles = (len + static_cast<uint64_t>(tile) - 1) / static_cast<uint64_t>(tile);
  // ground truth: the canonical state in front of every sub-chunk, and behind the last one
  HostMem<LOOK> g;
  g.hay = hay; g.origin_abs = 0; g.len = static_cast<int64_t>(len); g.outside = LOOK ? h->outside_byte : 0u;
  std::vector<uint32_t> truth;
  {
    uint32_t x = g.origin(v);
    for (uint64_t p = 0; p < len; p += static_cast<uint64_t>(sub)) {
      truth.push_back(fsm_canon(v, x));
      const uint64_t to = p + sub < len ? p + sub : len;
      x = fsm_walk(v, g, x, static_cast<int32_t>(p), static_cast<int32_t>(to), false);
    }
    truth.push_back(fsm_canon(v, x));
  }
  int64_t checked = 0;
  std::vector<EMap> tileMap(ntiles);
  for (uint64_t t = 0; t < ntiles; t++) {
    const uint64_t tile_lo = t * static_cast<uint64_t>(tile);
    const int32_t rend = static_cast<int32_t>(len - tile_lo < 0x7FFF0000ull ? len - tile_lo : 0x7FFF0000ull);
    HostMem<LOOK> m;
    m.hay = hay; m.origin_abs = static_cast<int64_t>(tile_lo); m.len = static_cast<int64_t>(len); m.outside = LOOK ? h->outside_byte : 0u;
    std::vector<EMap> sc(2 * lanes), lane_map(lanes), S(lanes);
    std::vector<bool> have(2 * lanes, false), unres(2 * lanes, false);
    for (int k = 0; k < 2 * lanes; k++) {
      const int32_t c0 = k * sub;
      if (c0 >= rend) break;
      have[k] = true;
      const int32_t to = c0 + sub < rend ? c0 + sub : rend;
      uint32_t entry = m.origin(v);
      if (tile_lo + static_cast<uint64_t>(c0) > 0) {
        const int64_t avail = static_cast<int64_t>(tile_lo) + c0;
        const int32_t w1 = static_cast<int32_t>(avail < 16 ? avail : 16), w2 = static_cast<int32_t>(avail < 64 ? avail : 64);
        entry = fsm_walk(v, m, v.top_off, c0 - w1, c0, (w1 % 4) == 0);
        if (entry >= v.u_lo && w2 > w1) entry = fsm_walk(v, m, v.top_off, c0 - w2, c0, (w2 % 4) == 0);
      }
      EMap e;
      if (entry >= v.u_lo) {
        unres[k] = true;
        if (fsm_member(v, entry, 0) == 0xFFFFu) return -16 - 1;
        for (int j = 0; j < kFsmMembers; j++) {
          const uint32_t mj = fsm_member(v, entry, static_cast<uint32_t>(j));
          if (mj != 0xFFFFu) e.P[j] = mj | (fsm_canon(v, fsm_walk(v, m, mj, c0, to, true)) << 16);
        }
      } else {
        if (fsm_canon(v, entry) != truth[(tile_lo + c0) / sub]) return -100 - 1;        // a collapsed set holds the true state
        e.isc = true; e.cv = fsm_canon(v, fsm_walk(v, m, entry, c0, to, true));
      }
      sc[k] = e;
    }
    int last = -1;
    for (int l = 0; l < lanes; l++) {
      if (!have[2 * l]) break;
      last = l;
      lane_map[l] = have[2 * l + 1] ? emap_then(sc[2 * l], sc[2 * l + 1]) : sc[2 * l];
    }
    if (last < 0) break;
    // inclusive scan, the kernel's level order
    S = lane_map;
    for (int d = 1; d < 64; d <<= 1) {
      std::vector<EMap> N = S;
      for (int l = 0; l <= last; l++) if (l - d >= 0) N[l] = emap_then(S[l - d], S[l]);
      S = N;
    }
    const uint32_t tile_entry = truth[tile_lo / sub];
    for (int l = 0; l <= last; l++) {
      const uint32_t before = l == 0 ? tile_entry : emap_apply(S[l - 1], tile_entry);
      if (before != truth[(tile_lo + 2ull * l * sub) / sub]) return -100 - 2;            // a lane's true entry
      if (have[2 * l + 1]) {
        const uint32_t mid = emap_apply(sc[2 * l], before);
        if (mid != truth[(tile_lo + (2ull * l + 1) * sub) / sub]) return -100 - 3;
        checked++;
      }
      checked++;
    }
    const uint64_t end_idx = (tile_lo + static_cast<uint64_t>(tile) < len ? tile_lo + tile : len + sub - 1) / sub;
    if (emap_apply(S[last], tile_entry) != truth[end_idx < truth.size() ? end_idx : truth.size() - 1]) return -100 - 4;   // the tile's exit
    tileMap[t] = S[last];
  }
  // groups: compose the tiles of a group; a group's entry from the groups in front, nearest first, until a known exit
  const uint64_t ngroups = (ntiles + tiles_per_group - 1) / tiles_per_group;
  std::vector<EMap> groupMap(ngroups);
  for (uint64_t gq = 0; gq < ngroups; gq++) {
    EMap c = tileMap[gq * tiles_per_group];
    for (uint64_t q = gq * tiles_per_group + 1; q < ntiles && q < (gq + 1) * tiles_per_group; q++) c = emap_then(c, tileMap[q]);
    groupMap[gq] = c;
  }
  for (uint64_t gq = 1; gq < ngroups; gq++) {
    const uint32_t want = truth[gq * tiles_per_group * static_cast<uint64_t>(tile) / sub];
    // C := identity on this group's candidates; C := C o M for the maps in front
    EMap C;
    const EMap& own = tileMap[gq * tiles_per_group];
    if (own.isc) continue;                                 // the group's first tile has a known entry-independent exit: nothing to look back for
    for (int j = 0; j < kFsmMembers; j++) if (own.P[j] != 0xFFFFFFFFu) C.P[j] = (own.P[j] & 0xFFFFu) | ((own.P[j] & 0xFFFFu) << 16);
    uint32_t got = 0xFFFFu;
    for (int64_t k = static_cast<int64_t>(gq) - 1; k >= 0; k--) {
      const EMap& M = groupMap[k];
      if (M.isc || k == 0) {                                                    // a VALUE: its exit is known (group 0: from the origin)
        const uint32_t e = M.isc ? M.cv : emap_apply(M, truth[0]);
        got = emap_look(C, e);
        break;
      }
      C = emap_then(M, C);
    }
    if (got != want) return -100 - 5;
  }
  return checked;
}

extern "C" int64_t emu_fsm_maps_check(const uint8_t* img, const uint8_t* hay, uint64_t len, int tile, int tiles_per_group) {
  const FsmHeader* h = reinterpret_cast<const FsmHeader*>(img);
  if (h->magic != kFsmMagic || tile % (2 * kFsmSub) != 0 || tile > 64 * 2 * kFsmSub || tiles_per_group < 1) return -1;
  return h->end_col ? emu_fsm_maps<2>(img, hay, len, tile, tiles_per_group) : h->nk > 1 ? emu_fsm_maps<1>(img, hay, len, tile, tiles_per_group) : emu_fsm_maps<0>(img, hay, len, tile, tiles_per_group);
}
