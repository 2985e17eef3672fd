// scan_fsml.hip — the lean form of the general-DFA FindAll kernel (see scan_fsm.hip for the design: one left-to-right transducer, fsm.hpp /
// host/fsm.cc, replayed by every lane over its own 64-byte chunk).
#include "scan_fsm_common.hpp"

namespace cxgdev {

// ---- The lean kernel (round 6): SHALLOW machines on input whose entry states collapse — what log text is.  Same geometry, window,
// row derivation, group ordering and look-back as k_scan_fsm<SHALLOW>; what it does not carry is the machinery for entry states that
// do NOT collapse (member maps, deferred tiles, hand-offs between tiles and groups — two thirds of k_scan_fsm's code and, through
// the registers and the control flow they cost, a quarter of its instructions on ordinary input): such input raises fallback
// reason 1 and the host reruns the call — and the program's next calls — on k_scan_fsm (capi_ladder.hip).
// KIND -1: direct mode (fsm.hpp "Direct mode": byte-indexed rows, v_perm_b32 + ds_read_u8 + v_alignbit per byte, machines without
// look-around whose rows fit); KIND 0 / 1 / 2: the class-indexed tables, LOOK = KIND.
namespace {
template <int IMG, int MODE, bool LOOKTAB>
struct FsmdLds {
  uint8_t img[IMG];                                            // direct: rows of 256 bytes at LDS address 0, then the property table; class-indexed: the image behind its header
  uint8_t win[kWavesPerBlock][kFsmWinBytes];
  uint16_t lk16[LOOKTAB ? 256 : 2];                            // look-around: class | kind << 8 of a byte (fsm.hpp FsmView::lk16)
  uint16_t re[kWavesPerBlock][FsmMode<MODE>::kRowsPerWave + 8];
  uint16_t rl[kWavesPerBlock][FsmMode<MODE>::kRowsPerWave];
  uint32_t cnt[kWavesPerBlock][kTilesPerWave];
  int32_t wdst[kWavesPerBlock][kTilesPerWave];                 // epilogue: output index of a tile's first row minus its index in the wave's list
  uint32_t qbase[kWavesPerBlock * kTilesPerWave + 4];
  int64_t tail[kWavesPerBlock * kTilesPerWave];
  uint64_t group;
  uint64_t base;
};
struct FsmdTab {                                               // the image sits at LDS address 0: a step's address goes straight into the ds_read
  lds_bytes_t img;
  __device__ __forceinline__ uint32_t at(uint32_t addr) const { return img[addr]; }
};
}  // namespace

template <int IMG, int MODE, int KIND>
#ifndef CXG_FSML_OCC
#define CXG_FSML_OCC 5          // workgroups per CU the register allocation aims at: 96 VGPRs (round 6, profiles/r06_c19_fsml_occ_*: README IP pattern 0.488 -> 0.466 ms
                                // against 4 / 127 VGPRs on one box; 12 bytes of scratch in the plain instantiation)
#endif
// (the LDS decides how many workgroups fit a CU — window 17 KB, rows 8 / 16 / 33 KB, the image: the register allocation aims at that number)
__global__ __launch_bounds__(kThreads, (((KIND >= 0 && IMG > 10240) || MODE == 3) ? 2 : MODE == 2 ? CXG_FSML_OCC2 : IMG > 6144 ? 4 : CXG_FSML_OCC)) void k_scan_fsml(ScanArgs a) {
  __shared__ __attribute__((aligned(16))) FsmdLds<IMG, MODE, (KIND > 0)> S;
  constexpr int kRowsPerWave = FsmMode<MODE>::kRowsPerWave;
  constexpr int tpw = FsmMode<MODE>::kTpw;
  constexpr bool DIRECT = KIND < 0;
  constexpr int LOOK = KIND < 0 ? 0 : KIND;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) S.group = claim_group(a.static_groups != 0, a.ticket, a.ngroups);
  const FsmHeader* h = reinterpret_cast<const FsmHeader*>(a.blob);
  {
    const uint4* src = reinterpret_cast<const uint4*>(a.blob + (DIRECT ? h->direct_off : static_cast<uint32_t>(sizeof(FsmHeader))));
    const uint32_t nvec = (DIRECT ? h->direct_bytes : h->lds_bytes) >> 4;
    for (uint32_t i = tid; i < nvec; i += kThreads) reinterpret_cast<uint4*>(S.img)[i] = src[i];
  }
  __syncthreads();
  const uint64_t group = S.group;
  if (group >= a.ngroups) return;
  FsmdTab tab;
  tab.img = (lds_bytes_t)S.img;
  FsmView v = view_of(S.img, h);                                                  // (class-indexed kinds; direct: unused)
  if (LOOK) {                                                                     // (kThreads == 256: one entry each)
    S.lk16[tid] = static_cast<uint16_t>(v.cls2[tid] | (static_cast<uint32_t>(v.knd[tid]) << 8));
    __syncthreads();
  }
  v.lk16 = S.lk16;
  const uint32_t top = h->d_top, prop = h->d_slots << 8;                                                  // (direct)
  const FsmdRev R = {h->d_rstart, h->d_racc_lo, h->d_rdead};
  const uint32_t outside = LOOK ? h->outside_byte : 0u;
  uint32_t nrows_w = 0, fallback = 0, long_hit = 0;

  u32x4 x[4];
  uint32_t xbehind = 0;
  // (a tile's offset is carried from tile to tile: the 64-bit products of the tile's number were a dozen scalar instructions per tile)
  constexpr uint64_t kTileStep = static_cast<uint64_t>(kWavesPerBlock) * kWaveTile;
  uint64_t tlo = (group * (kWavesPerBlock * tpw) + static_cast<uint64_t>(wave)) * static_cast<uint64_t>(kWaveTile);
  auto issue_loads = [&](uint64_t lo, bool in_group) {
    int nrec = 0;
    const int pre = lo ? 0 : kFsmLeft;
    const uint64_t from = lo ? lo - kFsmLeft : 0;
    constexpr int kStaged = 4096 + (LOOK ? 4 : 0);
    if (in_group && lo < a.len) {
      const uint64_t rem = a.len - from;
      nrec = rem >= static_cast<uint64_t>(kStaged - pre) ? kStaged - pre : static_cast<int>((rem + 3) & ~3ull);
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.hay) + (nrec ? from : 0), 0, nrec, 0x00020000);
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((lane + 64 * k) << 4) - pre, 0, 0);
    if (LOOK) xbehind = __builtin_amdgcn_raw_buffer_load_b32(rsrc, 4096 - pre, 0, 0);
  };
  issue_loads(tlo, true);
  for (int j = 0; j < tpw; j++, tlo += kTileStep) {
    const uint64_t tile_lo = tlo;
    uint32_t tot = 0;
    if (tile_lo < a.len) {
      uint8_t* win = S.win[wave];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t wo = static_cast<uint32_t>(lane + 64 * k) << 4;
        uint32_t* d = reinterpret_cast<uint32_t*>(win + wo + (wo >> 6) * 4u);
        d[0] = x[k].x; d[1] = x[k].y; d[2] = x[k].z; d[3] = x[k].w;
      }
      if (LOOK && lane == 0) *reinterpret_cast<uint32_t*>(win + 64 * kFsmStride) = xbehind;   // window position 4096
      issue_loads(tlo + kTileStep, j + 1 < tpw);
      const uint64_t remaining = a.len - tile_lo;
      const int32_t rend = remaining > 0x7FFF0000ull ? 0x7FFF0000 : static_cast<int32_t>(remaining);
      if (LOOK && lane == 0) {                          // the byte in front of the haystack and the one behind its end (DS ops of a wave keep their order)
        if (tile_lo == 0) win[kFsmLeft - 1] = static_cast<uint8_t>(outside);
        if (rend <= kFsmWinEnd) { const uint32_t w = static_cast<uint32_t>(rend + kFsmLeft); win[w + (w >> 6) * 4u] = static_cast<uint8_t>(outside); }
      }
      wave_lds_sync();
      const int32_t lowest = tile_lo ? -kFsmLeft : 0;
      const int32_t rev_lowest = (LOOK && tile_lo) ? lowest + 1 : lowest;   // (a reverse step with look-around reads the byte in front of its own)
      FsmMem<LOOK> m;
      m.win = (lds_bytes_t)win;
      m.last_ = rend - 1;
      const int32_t c0 = (lane - 1) * kFsmChunk;
      const bool owned = lane >= 1 && lane <= kWaveTile / kFsmChunk && c0 < rend;
      const bool active = lane >= 1 && c0 < rend;               // the three lanes behind the tile: event bits of the window's tail
      const int32_t cc[2] = {c0, c0 + kFsmSub};
      uint64_t KK[2] = {0ull, 0ull};
      uint32_t xend1 = 0u;
      if (active) {
        // warm-up: 16 bytes from "any state" in front of each sub-chunk (bytes in front of the haystack read as zeros and are
        // overruled below); a set that has not collapsed by then gets 64 bytes, and what is left after that is the host's
        const bool at_origin = tile_lo == 0 && lane == 1;
        const int32_t wfrom[2] = {cc[0] - 16, cc[1] - 16};
        FsmTraceS t[2];
        if constexpr (DIRECT) {
          uint32_t e[2] = {top, top};
          if (!(CXG_FSM_ABL & 1)) fsmd_walk_n<2>(m, tab, wfrom, 16, e);
          if (at_origin) e[0] = 0u;                               // the haystack's first byte: the search starts in state 0
          const uint32_t u0 = tab.at(prop + e[0]) & 0x80u, u1 = tab.at(prop + e[1]) & 0x80u;
          if (u0 | u1) {                                          // rare on text
#pragma unroll
            for (int sb = 0; sb < 2; sb++) {                      // (unrolled: a run-time index into cc[] / e[] would put them into scratch memory)
              if (!(sb ? u1 : u0)) continue;
              const bool from_start = at_origin && sb;            // (the second sub-chunk of the haystack's first chunk: from the true start state)
              const int32_t f1[1] = {from_start ? 0 : cc[sb] - 64};
              uint32_t xx[1] = {from_start ? 0u : top};
              fsmd_walk_n<1>(m, tab, f1, cc[sb] - f1[0], xx);
              e[sb] = xx[0];
              if (tab.at(prop + xx[0]) & 0x80u) fallback |= 1u;
            }
          }
          // the chunk: two chains in lockstep, the flag bits of every step shifted into the masks
          t[0] = {e[0], 0u, 0u}; t[1] = {e[1], 0u, 0u};
          if (!(CXG_FSM_ABL & 2)) fsmd_chunk<2>(m, tab, cc, t);
          xend1 = t[1].x;
        } else {
          uint32_t e[2] = {0u, 0u};
          if (!(CXG_FSM_ABL & 1)) fsm_walk_n<2>(v, m, v.top_off, wfrom, 16, e);
          if (at_origin) e[0] = m.origin(v);
          if ((e[0] >= v.u_lo) | (e[1] >= v.u_lo)) {              // rare on text
#pragma unroll
            for (int sb = 0; sb < 2; sb++) {                      // (unrolled: a run-time index into cc[] / e[] would put them into scratch memory)
              if (e[sb] < v.u_lo) continue;
              const bool from_start = at_origin && sb;
              e[sb] = fsm_walk(v, m, from_start ? m.origin(v) : v.top_off, from_start ? 0 : cc[sb] - 64, cc[sb], true);
              if (e[sb] >= v.u_lo) fallback |= 1u;
            }
          }
          t[0] = {e[0], 0u, 0u}; t[1] = {e[1], 0u, 0u};
          if (!(CXG_FSM_ABL & 2)) fsm_fast_shallow<2>(v, m, cc, t);
          xend1 = t[1].x & ~3u;
        }
        KK[0] = (static_cast<uint64_t>(t[0].k1) << 32) | t[0].k0; KK[1] = (static_cast<uint64_t>(t[1].k1) << 32) | t[1].k0;
      }
      tot = fsm_rows_from_events<kRowsPerWave>(KK, active, owned, rend, c0, lane, S.re[wave], nrows_w, fallback,
                                               [&]() -> uint32_t { if constexpr (DIRECT) return tab.at(prop + xend1) & 0x7Fu; else return fsm_u16(v.tab, xend1 + v.ncls2 + 2u); },
                                               static_cast<uint32_t>(j) << 12);
      wave_lds_sync();
      if (a.out != nullptr || a.max_len != 0) {
        // a tile with 128 rows and more (`\\b\\d+\\b`: 535, nine rounds of this loop) has short matches: 8 branch-free steps in front of the loop
        // instead of 16 (the answer is the same, fsm.hpp fsm_match_startN)
        const bool short_rows = tot >= 128u;
        for (uint32_t q = lane; q < tot && nrows_w + q < static_cast<uint32_t>(kRowsPerWave); q += 64) {
          const int32_t e = S.re[wave][nrows_w + q] & 4095;       // (bits 12..14: the tile's number, for the epilogue)
          const int32_t bound = q ? static_cast<int32_t>(S.re[wave][nrows_w + q - 1] & 4095) : (tile_lo ? lowest - 1 : 0);
          uint32_t over = 0;
          int32_t st;
          if constexpr (DIRECT) {
            if (!CXG_FSM_FAST_STARTS) st = fsmd_match_start(m, tab, R, e, bound, lowest, over);
            else if (short_rows) st = fsmd_match_startN<8>(m, tab, R, e, bound, lowest, over);
            else st = fsmd_match_startN<16>(m, tab, R, e, bound, lowest, over);
          } else {
            // (a text-start anchor is the loop's business: the walk that arrives at position 0 alive asks the state)
            if (!CXG_FSM_FAST_STARTS || (v.rev_text_col != 0u && tile_lo == 0)) st = fsm_match_start(v, m, e, bound, rev_lowest, over, tile_lo == 0 ? 0 : kFsmNoStart);
            else if (short_rows) st = fsm_match_startN<8>(v, m, e, bound, rev_lowest, over);
            else st = fsm_match_startN<16>(v, m, e, bound, rev_lowest, over);
          }
          const uint32_t len = (over || st == kFsmNoStart) ? 0u : static_cast<uint32_t>(e - st);
          if (st == kFsmNoStart && !over) fallback |= 64u;
          S.rl[wave][nrows_w + q] = static_cast<uint16_t>(len);
        }
      }
    }
    if (lane == 0) S.cnt[wave][j] = tot;
    nrows_w += tot;
  }
  if (nrows_w > static_cast<uint32_t>(kRowsPerWave)) fallback |= 32u;
  {
    uint32_t f = fallback;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) f |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(f), d, 64));
    if (f != 0 && lane == 0) raise_err(a.err, 8u | (f << 8));
  }
  __syncthreads();
  // ---- order the group's rows (as k_scan_fsm): wave-tile q = j * 4 + wave
  if (tid < 64) {
    const int q = tid;
    const uint32_t c = (q < kWavesPerBlock * tpw) ? S.cnt[q % kWavesPerBlock][q / kWavesPerBlock] : 0u;
    const uint32_t incl = wave_inclusive_sum(c);
    if (q < kWavesPerBlock * tpw) S.qbase[q] = incl - c;
    if (q == kWavesPerBlock * tpw - 1) S.qbase[kWavesPerBlock * tpw] = incl;
  }
  const int64_t gorigin = static_cast<int64_t>(group * static_cast<uint64_t>(kWaveTile) * kWavesPerBlock * tpw);
  if (tid >= 64 && tid < 64 + kWavesPerBlock * tpw) {
    const int q = tid - 64, w = q % kWavesPerBlock, jj = q / kWavesPerBlock;
    uint32_t st = 0;
    for (int k = 0; k < jj; k++) st += S.cnt[w][k];
    const uint32_t n = S.cnt[w][jj];
    S.tail[q] = (n && st + n <= static_cast<uint32_t>(kRowsPerWave)) ? gorigin + static_cast<int64_t>(q) * kWaveTile + (S.re[w][st + n - 1] & 4095) : -1;
  }
  __syncthreads();
  if (tid >= 128 && tid < 128 + kWavesPerBlock * tpw) {       // (behind the barrier: qbase is there)
    const int q = tid - 128, w = q % kWavesPerBlock, jj = q / kWavesPerBlock;
    uint32_t st = 0;
    for (int k = 0; k < jj; k++) st += S.cnt[w][k];
    S.wdst[w][jj] = static_cast<int32_t>(S.qbase[q]) - static_cast<int32_t>(st);
  }
  __syncthreads();
  const uint32_t total = S.qbase[kWavesPerBlock * tpw];
  tile_lookback(a.status, a.total, a.err, group, a.ngroups, total, &S.base, a.epoch);
  const uint64_t base = S.base;
  // the wave's rows as one list (an end carries its tile's number): five rounds of 64 rows for the 288 of an ordinary group instead of one
  // round per tile with 36 of 64 lanes at work
  const uint32_t nr = nrows_w < static_cast<uint32_t>(kRowsPerWave) ? nrows_w : static_cast<uint32_t>(kRowsPerWave);
  for (uint32_t r = lane; r < nr; r += 64) {
    const uint32_t vr = S.re[wave][r], jr = vr >> 12;
    const int q = static_cast<int>(jr) * kWavesPerBlock + wave;
    const int64_t tb = gorigin + static_cast<int64_t>(q) * kWaveTile;
    int64_t e = tb + static_cast<int64_t>(vr & 4095u), s0 = e - S.rl[wave][r];
    const uint32_t vp = r ? S.re[wave][r - 1] : 0xFFFFu;
    const bool first = (vp >> 12) != jr;                                      // the tile's first row
    if ((first || s0 == e) && (a.out != nullptr || a.max_len != 0)) {
      // the tile's first row was walked without a bound (its predecessor is another wave's row); an unresolved row (s0 == e) left
      // the window.  Previous end: inside the tile, else the nearest earlier tile of the group with rows; the group's first row is
      // checked by k_fsm_fix_heads.
      int64_t prev = -1;
      if (!first) prev = tb + static_cast<int64_t>(vp & 4095u);
      else for (int p = q - 1; p >= 0 && prev < 0; p--) prev = S.tail[p];
      if (prev > s0 || s0 == e) {                                           // rare: walk again from HBM / L2, bounded
        const int64_t lo = prev > 0 ? prev : 0;
        int64_t st = -1, at = e - 1;
        if constexpr (DIRECT) {
          uint32_t sr = R.start;
          for (; at >= lo; at--) {
            if (e - at > kSerialLimit) { raise_err(a.err, kErrSerialLimit); break; }
            sr = tab.at(fsmd_addr(sr, a.hay[at], 0));
            if (sr == R.dead) break;
            if (sr >= R.acc_lo) st = at;
          }
        } else {
          uint32_t sr = v.rev_start_off;
          if (LOOK) sr = fsm_u16(v.knd, 256u + 2u * v.nk + 2u * ((v.knd[a.hay[e - 1]] >> 1) * v.nk + ((static_cast<uint64_t>(e) < a.len ? v.knd[a.hay[e]] : (LOOK == 2 ? v.end_col : v.knd[outside])) >> 1)));
          for (; at >= lo; at--) {
            if (e - at > kSerialLimit) { raise_err(a.err, kErrSerialLimit); break; }
            sr = fsm_u16(v.tab, (sr & ~1u) + v.cls2[a.hay[at]] + (LOOK ? v.knd[at > 0 ? a.hay[at - 1] : outside] : 0u));
            if (sr == v.rev_dead) break;
            if (sr & 1u) st = at;
          }
          if (LOOK && v.rev_text_col != 0u && at < 0 && sr != v.rev_dead && fsm_u16(v.tab, (sr & ~1u) + v.rev_text_col + v.knd[a.hay[0]]) != 0u) st = 0;   // text-start anchor (fsm.hpp fsm_match_start)
        }
        if (st < 0) raise_err(a.err, 8u | (64u << 8)); else s0 = st;
      }
    }
    if (a.max_len != 0 && static_cast<uint64_t>(e - s0) > a.max_len) long_hit = 1;
    const uint64_t di = base + static_cast<uint64_t>(static_cast<int64_t>(S.wdst[wave][jr]) + static_cast<int64_t>(r));
    if (a.out != nullptr && di < a.cap) store_pair_nt(a.out + di * a.row_width, a.base + s0, a.base + e);
  }
  if (long_hit) raise_err(a.err, kErrLongMatch);
}

namespace {
template <int IMG, int KIND>
void launch_fsml_img(const ScanArgs& a, int mode, dim3 grid, dim3 block, hipStream_t stream) {
  if (mode == 0) hipLaunchKernelGGL((k_scan_fsml<IMG, 0, KIND>), grid, block, 0, stream, a);
  else if (mode == 1) hipLaunchKernelGGL((k_scan_fsml<IMG, 1, KIND>), grid, block, 0, stream, a);
  else if (mode == 2) hipLaunchKernelGGL((k_scan_fsml<IMG, 2, KIND>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((k_scan_fsml<IMG, 3, KIND>), grid, block, 0, stream, a);
}
}  // namespace

// direct_bytes != 0: the direct mode (the caller has checked that the image carries the section); look: 0 / 1 / 2 as the kernel's KIND.
// The first rows of the groups are checked by k_fsm_fix_heads behind it (launch_scan_fsm, scan_fsm.hip).
hipError_t launch_scan_fsml(const ScanArgs& a, uint32_t lds_bytes, bool shallow, int look, hipStream_t stream, uint32_t direct_bytes, int mode) {
  const dim3 grid(static_cast<unsigned>(a.ngroups)), block(kThreads);
  if (lds_bytes > 28672 || !shallow) return hipErrorInvalidValue;
  if (direct_bytes != 0u) {
    if (direct_bytes > kFsmdMaxBytes || look) return hipErrorInvalidValue;
    if (direct_bytes <= 6144) launch_fsml_img<6144, -1>(a, mode, grid, block, stream);
    else launch_fsml_img<12288, -1>(a, mode, grid, block, stream);
  }
  else if (look == 2) {                                                                              // end-of-text programs
    if (lds_bytes <= 3072) launch_fsml_img<3072, 2>(a, mode, grid, block, stream);
    else if (lds_bytes <= 10240) launch_fsml_img<10240, 2>(a, mode, grid, block, stream);
    else launch_fsml_img<28672, 2>(a, mode, grid, block, stream);
  }
  else if (look) {
    if (lds_bytes <= 3072) launch_fsml_img<3072, 1>(a, mode, grid, block, stream);
    else if (lds_bytes <= 10240) launch_fsml_img<10240, 1>(a, mode, grid, block, stream);
    else launch_fsml_img<28672, 1>(a, mode, grid, block, stream);
  }
  else if (lds_bytes <= 3072) launch_fsml_img<3072, 0>(a, mode, grid, block, stream);
  else if (lds_bytes <= 10240) launch_fsml_img<10240, 0>(a, mode, grid, block, stream);
  else launch_fsml_img<28672, 0>(a, mode, grid, block, stream);
  return hipGetLastError();
}

}  // namespace cxgdev
