// scan_runs_wave.hip — FindAll for patterns over a small alphabet (device/runs.hpp), the wave as the unit of work.
//
// `(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.…` consumes digits and dots only: a match lies inside one maximal run of such bytes,
// an assertion reads one byte on either side.  The transducer kernel walks every byte of the haystack through its table
// (0.74 ms per GiB of log text); here the bytes are only CLASSIFIED — bit-parallel, as in scan_charclass_wave.hip — and the
// table is walked inside the runs that are long enough to hold a match (on log text: the addresses, 38 runs of 13 bytes per
// tile), one lane per run, position by position as the reference's own digit-prefilter loop does
// (meta/find_indices.go:1050-1088: anchored search at each candidate, first success wins, go on at its end).
//   per wave-tile (3840 B + 256 B halo, 64 bitmap words):
//     window by four buffer_load_dwordx4 per lane -> LDS bytes (the walks read them) and membership bits (SWAR range tests);
//     run starts S = M & ~(M << 1), eroded by the shortest match length (log steps of shifted ANDs): a start whose next
//     min_len - 1 bytes are members too qualifies; the tile owns the runs that START at its bytes [0, 3840) and sees them whole
//     (a run longer than 255 bytes raises the fallback flag);
//     qualifying starts are compacted over the lanes (64 per round); a lane walks its run with runs.hpp's table: attempt at p,
//     on success a row and p = its end, else p + 1; rows go to the wave's pool in LDS in run order;
//   group (4 waves x 4 wave-tiles = 60 KiB): one barrier, one look-back for the global row base, rows out as coalesced
//     nontemporal 16-byte stores.
// Runs are independent, so nothing but the row base crosses a tile boundary: no entry states, no pending matches, no reverse walks.
// Fallback flag (err bit 8, reason bits << 8): 1 a qualifying run longer than 255 bytes, 2 more than 256 qualifying runs in a
// tile or the wave's row pool is full, 4 more than four rows in one run.  The host then takes the transducer (capi.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_common.hpp"
#include "runs.hpp"
#include "scan_dfa.h"
#include "walk.hpp"
#include "wave_common.hpp"

namespace cxgdev {

namespace {
constexpr int kWin = kWaveTile + kWaveHalo;       // 4096
constexpr int kRunsQ = 256;                       // qualifying runs per wave-tile
constexpr int kRunsPool = 512;                    // rows of a wave's four tiles
constexpr int kRunsRowsPerRun = 4;
constexpr int kRunsImgLds = static_cast<int>(kRunsMaxImage - sizeof(RunsHeader));
}

__global__ __launch_bounds__(kThreads, 4) void k_scan_runs_wave(ScanArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t s_img[kRunsImgLds];                          // cls[256], then the table
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kWavesPerBlock][16 + kWin + 16];       // [15] = the byte in front of the tile
  __shared__ __attribute__((aligned(16))) uint64_t s_m[kWavesPerBlock][64];
  __shared__ uint16_t s_q[kWavesPerBlock][kRunsQ];
  __shared__ uint32_t s_tmp[kWavesPerBlock][64 * kRunsRowsPerRun];
  __shared__ uint32_t s_pool[kWavesPerBlock][kRunsPool];                                       // start | end << 16, window coordinates
  __shared__ uint32_t s_cnt[kWavesPerBlock][kCcTilesPerWave];
  __shared__ uint32_t s_qbase[kWavesPerBlock * kCcTilesPerWave + 1];
  __shared__ uint64_t s_group;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = lane0;
  if (tid == 0) s_group = claim_group(a.static_groups != 0, a.ticket, a.ngroups);
  const RunsHeader* h = reinterpret_cast<const RunsHeader*>(a.blob);                           // uniform address: scalar loads
  {
    const uint32_t nd = (h->total_bytes - h->cls_off) >> 2;                                    // total_bytes is a multiple of 16
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.blob + h->cls_off);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_img);
    for (uint32_t i = tid; i < nd && i < static_cast<uint32_t>(kRunsImgLds / 4); i += kThreads) dst[i] = src[i];
  }
  SetRanges rg;
  rg.n = h->nr;
#pragma unroll
  for (int q = 0; q < 4; q++) { rg.lo4[q] = h->lo[q] * 0x01010101u; rg.hi4[q] = (0x7Fu - h->hi[q]) * 0x01010101u; }
  const uint32_t min_len = h->min_len < 64u ? h->min_len : 64u;
  const uint32_t sym_end = h->nsym - 1u;
  const uint32_t start_row[4] = {h->start[0], h->start[1], h->start[2], h->start[3]};
  const uint8_t* const cls = s_img;
  const uint8_t* const tab = s_img + (h->tab_off - h->cls_off);
  __syncthreads();
  const uint64_t group = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_group >> 32))) << 32) |
                         static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_group)));
  if (group >= a.ngroups) return;
  if (limit_reached_skip(a, group, &s_base)) return;                 // FindAll with n > 0 (block_common.hpp)
  uint32_t fallback = 0;

  u32x4 x[4];
  uint32_t xprev = 0;
  auto issue_loads = [&](int jj) {
    const uint64_t wtn = group * (kWavesPerBlock * kCcTilesPerWave) + static_cast<uint64_t>(jj) * kWavesPerBlock + wave;
    const uint64_t lo = wtn * static_cast<uint64_t>(kWaveTile);
    int nrec = 0;
    if (jj < kCcTilesPerWave && lo < a.len) {
      const uint64_t rem = a.len - lo;
      nrec = rem >= static_cast<uint64_t>(kWin) ? kWin : static_cast<int>((rem + 3) & ~3ull);
    }
    const int pre = (nrec && lo) ? 16 : 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.hay) + (nrec ? lo - pre : 0), 0, nrec + pre, 0x00020000);
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (lane + 64 * k) << 4, pre, 0);
    xprev = __builtin_amdgcn_raw_buffer_load_b32(rsrc, 0, pre ? 12 : nrec + pre, 0);
  };
  issue_loads(0);

  uint8_t* const win = s_win[wave] + 16;
  uint32_t pool_used = 0;
  // ---- pass 1: runs, walks, rows into the pool
  for (int j = 0; j < kCcTilesPerWave; j++) {
    lane = lane0;
    asm volatile("" : "+v"(lane));
    const uint64_t wt = group * (kWavesPerBlock * kCcTilesPerWave) + static_cast<uint64_t>(j) * kWavesPerBlock + wave;
    const uint64_t tile_lo = wt * static_cast<uint64_t>(kWaveTile);
    uint32_t rows_here = 0;
    if (tile_lo < a.len) {
      const uint64_t remaining = a.len - tile_lo;
      const int32_t rend = remaining > 0x7FFF0000ull ? 0x7FFF0000 : static_cast<int32_t>(remaining);
      const int32_t stage = rend < kWin ? rend : kWin;
      uint16_t* pieces = reinterpret_cast<uint16_t*>(s_m[wave]);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t lo = __builtin_amdgcn_udot4(notset4(x[k].y, rg), 0x80402010u, __builtin_amdgcn_udot4(notset4(x[k].x, rg), 0x08040201u, 0u, false), false);
        const uint32_t hi = __builtin_amdgcn_udot4(notset4(x[k].w, rg), 0x80402010u, __builtin_amdgcn_udot4(notset4(x[k].z, rg), 0x08040201u, 0u, false), false);
        pieces[lane + 64 * k] = static_cast<uint16_t>(((lo >> 7) | (hi << 1)) ^ 0xFFFFu);
        *reinterpret_cast<u32x4*>(win + ((lane + 64 * k) << 4)) = x[k];
        __builtin_amdgcn_sched_barrier(0);
      }
      const uint32_t pb = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(xprev))) >> 24;
      if (lane == 0) win[-1] = static_cast<uint8_t>(pb);
      issue_loads(j + 1);
      wave_lds_sync();
      uint64_t M = s_m[wave][lane];
      if (stage != kWin) {                                          // short last window: nothing past the data is a member
        const int32_t nv = stage - 64 * lane;
        M &= nv <= 0 ? 0ull : (nv >= 64 ? ~0ull : ((1ull << nv) - 1ull));
      }
      const uint32_t prev_member = tile_lo > 0 ? static_cast<uint32_t>(cls[pb]) >> 7 : 0u;
      uint64_t carry = from_lower64(M) >> 63;                      // DPP outside lane-dependent branches
      if (lane == 0) carry = prev_member;
      uint64_t S = M & ~((M << 1) | carry);
      S &= word_range(lane, 0, kWaveTile - 1);                      // the tile owns the runs that start at its bytes [0, 3840)
      // erosion: bit p of X <=> bytes p .. p + min_len - 1 are members
      uint64_t X = M;
      {
        uint32_t cur = 1;
        auto and_shifted = [&](uint32_t k) {                        // X &= X >> k over the 4096-bit window, 1 <= k <= 32
          uint64_t up = from_upper64(X);
          if (lane == 63) up = 0;
          X &= (X >> k) | (up << (64u - k));
        };
        while (cur * 2u <= min_len) { and_shifted(cur); cur *= 2u; }
        if (cur < min_len) and_shifted(min_len - cur);
      }
      const uint64_t SQ = S & X;
      const uint32_t ns = static_cast<uint32_t>(__popcll(SQ));
      const uint32_t incl = wave_inclusive_sum(ns);
      uint32_t nq = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
      if (nq > static_cast<uint32_t>(kRunsQ)) { fallback |= 2u; nq = kRunsQ; }
      {
        uint32_t r = incl - ns;
        uint64_t sb = SQ;
        while (sb) {
          const int bit = __builtin_ctzll(sb);
          sb &= sb - 1;
          if (r < static_cast<uint32_t>(kRunsQ)) s_q[wave][r] = static_cast<uint16_t>(64 * lane + bit);
          r++;
        }
      }
      wave_lds_sync();
      const bool at_text_start = tile_lo == 0;
      if (a.dbg & 16u) nq = 0;                                      // (CXG_DEBUG=16: timing experiment — no walks, no rows)
      for (uint32_t qb = 0; qb < nq; qb += 64u) {
        const uint32_t idx = qb + static_cast<uint32_t>(lane0);
        uint32_t c = 0;
        if (idx < nq) {
          const int32_t s = s_q[wave][idx];
          int32_t p = s;
          for (;;) {
            if (p >= stage || !(cls[win[p]] & 0x80u)) break;        // the run is over
            if (p - s >= static_cast<int32_t>(kRunsMaxRun)) { fallback |= 1u; break; }
            const uint32_t behind = (at_text_start && p == 0) ? 3u : (static_cast<uint32_t>(cls[win[p - 1]]) >> 5) & 3u;
            uint32_t st = start_row[0];
            if (behind == 1u) st = start_row[1];
            if (behind == 2u) st = start_row[2];
            if (behind == 3u) st = start_row[3];
            int32_t last = -1, q = p;
            while (st != 0u) {
              const uint32_t sym = q < stage ? (cls[win[q]] & 31u) : sym_end;
              const uint32_t e = *reinterpret_cast<const uint16_t*>(tab + st + 2u * sym);
              if (e & 1u) last = q;
              st = e & 0xFFFEu;
              if (q >= stage) break;
              q++;
              if (q - s > static_cast<int32_t>(kRunsMaxRun) + 1) { fallback |= 1u; st = 0u; }
            }
            if (last > p) {
              if (c < static_cast<uint32_t>(kRunsRowsPerRun)) s_tmp[wave][lane0 * kRunsRowsPerRun + static_cast<int>(c)] = static_cast<uint32_t>(p) | (static_cast<uint32_t>(last) << 16);
              c++;
              p = last;
            } else p++;
            if (a.dbg & 32u) break;                                 // (CXG_DEBUG=32: timing experiment — one attempt per run)
          }
        }
        if (c > static_cast<uint32_t>(kRunsRowsPerRun)) { fallback |= 4u; c = kRunsRowsPerRun; }
        const uint32_t ci = wave_inclusive_sum(c);
        const uint32_t tot = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ci), 63));
        const uint32_t off = pool_used + rows_here + ci - c;
        for (uint32_t k = 0; k < c; k++) {
          if (off + k < static_cast<uint32_t>(kRunsPool)) s_pool[wave][off + k] = s_tmp[wave][lane0 * kRunsRowsPerRun + static_cast<int>(k)];
          else fallback |= 2u;
        }
        rows_here += tot;
      }
      if (pool_used + rows_here > static_cast<uint32_t>(kRunsPool)) rows_here = static_cast<uint32_t>(kRunsPool) - pool_used;   // (flagged above)
    } else {
      issue_loads(j + 1);
    }
    if (lane0 == 0) s_cnt[wave][j] = rows_here;
    pool_used += rows_here;
  }
  if (__any(fallback != 0u)) {
    uint32_t f = fallback;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) f |= __shfl_xor(f, d, 64);
    if (lane0 == 0) raise_err(a.err, 8u | (f << 8));
  }
  __syncthreads();

  // ---- group: exclusive prefix over the wave-tiles q = j * 4 + wave, look-back
  if (tid < 64) {
    const int q = tid;
    const uint32_t v = (q < kWavesPerBlock * kCcTilesPerWave) ? s_cnt[q % kWavesPerBlock][q / kWavesPerBlock] : 0u;
    const uint32_t incl = wave_inclusive_sum(v);
    if (q < kWavesPerBlock * kCcTilesPerWave) s_qbase[q] = incl - v;
    if (q == kWavesPerBlock * kCcTilesPerWave - 1) s_qbase[kWavesPerBlock * kCcTilesPerWave] = incl;
  }
  __syncthreads();
  const uint32_t total = s_qbase[kWavesPerBlock * kCcTilesPerWave];
  tile_lookback(a.status, a.total, a.err, group, a.ngroups, total, &s_base, a.epoch, a.limit, a.stop);
  if (a.out == nullptr) return;

  // ---- pass 2: the pool's rows to their places
  const uint64_t base = s_base;
  const bool u32 = a.u32_rows != 0u;
  uint32_t* const out32 = reinterpret_cast<uint32_t*>(a.out);
  const int64_t origin = (u32 ? 0 : a.base) + static_cast<int64_t>(group * static_cast<uint64_t>(kWaveTile) * kWavesPerBlock * kCcTilesPerWave);
  uint32_t seg = 0;
  for (int j = 0; j < kCcTilesPerWave; j++) {
    const uint32_t n = s_cnt[wave][j];
    const uint64_t row0 = base + s_qbase[j * kWavesPerBlock + wave];
    const int64_t tb = origin + static_cast<int64_t>(j * kWavesPerBlock + wave) * kWaveTile;
    for (uint32_t i = lane0; i < n; i += 64) {
      if (row0 + i < a.cap) {
        const uint32_t v = s_pool[wave][seg + i];
        const int64_t st = tb + (v & 0xFFFFu), en = tb + (v >> 16);
        if (u32) store_pair32_nt(out32 + (row0 + i) * 2, static_cast<uint32_t>(st), static_cast<uint32_t>(en));
        else store_pair_nt(a.out + (row0 + i) * 2, st, en);
      }
    }
    seg += n;
  }
}

hipError_t launch_scan_runs_wave(const ScanArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(k_scan_runs_wave, dim3(static_cast<unsigned>(a.ngroups)), dim3(kThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace cxgdev
