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
template <bool LOOK>
struct HostMem : FsmClassify<HostMem<LOOK>, LOOK> {
  const uint8_t* hay;   // whole haystack
  int64_t origin_abs;   // absolute position of the tile origin
  int64_t len;
  uint32_t outside;     // FsmHeader::outside_byte: what the kernel writes into its window around the haystack
  uint32_t byte(int32_t r) const { const int64_t p = origin_abs + r; return (p >= 0 && p < len) ? hay[p] : outside; }
  uint32_t dword(int32_t r) const { return byte(r) | (byte(r + 1) << 8) | (byte(r + 2) << 16) | (byte(r + 3) << 24); }
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
  v.rev_start_off = h->rev_start_off; v.rev_accept_off = h->rev_accept_off;
  v.create_lo = h->create_lo; v.rematch_lo = h->rematch_lo;
  v.mem = img + h->mem_off; v.row_shift = h->row_shift;
  v.knd = img + h->knd_off;
  v.nk = h->nk;
  return v;
}
}  // namespace

// Returns the number of int64 values written (2 per match) or needed; -16 - reason when a tile would raise the
// fallback flag (reason 1: a lane's entry state did not collapse, 2: more than kFsmLaneRows rows in a chunk,
// 4: level stack overflow, 8: walk budget), -1 on a bad image.  stats (optional, 4 values): chunks, chunks whose entry
// needed the full warm-up, chunks whose entry set did not collapse, rows fixed against the previous row.
template <bool LOOK>
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
        int32_t s = fsm_match_start(v, m, e, bound, rev_lowest, over);
        if (over || (first_in_tile && s != kFsmNoStart && static_cast<int64_t>(tile_lo) + s < prev_end)) {
          // the kernel's epilogue: the walk again, bounded by the previous row's end, bytes from HBM / L2
          st[3]++;
          over = 0;
          s = fsm_match_start(v, m, e, static_cast<int32_t>(prev_end - static_cast<int64_t>(tile_lo)), -static_cast<int32_t>(tile_lo), over);
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


extern "C" int64_t emu_find_all_fsm(const uint8_t* img, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals,
                                    int tile, int chunk, int budget_bytes, uint64_t* stats, int dense) {
  const FsmHeader* h = reinterpret_cast<const FsmHeader*>(img);
  if (h->magic != kFsmMagic || chunk % 4 != 0 || tile % chunk != 0) return -1;
  return h->nk > 1 ? emu_fsm<true>(img, hay, len, out, cap_vals, tile, chunk, budget_bytes, stats, dense)
                   : emu_fsm<false>(img, hay, len, out, cap_vals, tile, chunk, budget_bytes, stats, dense);
}
