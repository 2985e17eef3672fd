// scan_fsm.hip — the general-DFA FindAll kernel: one left-to-right finite-state transducer (fsm.hpp, host/fsm.cc)
// replayed by every lane over its own 64-byte chunk.  Serves what is neither a bit-parallel chain, a literal set nor a
// char class: UseDFA / UseBoth programs with branching DFAs, UseDigitPrefilter programs with alternations
// (`(?:25[0-5]|2[0-4][0-9]|...)`), the span pass of FindAllSubmatchIndex — and it is the fallback of the wave kernels
// for match-dense input and input WITHOUT synchronising bytes, which the table-walking kernels of round 1 refused.
//
// Reference semantics kept (meta/findall.go:216-239 over dfa/lazy/lazy.go:1102-1315,1769-1920): leftmost-first match
// at or after `pos`, its start by the reverse DFA bounded below by `pos`, next search from its end.
//
// One wave64 owns a wave-tile of 60 x 64 B (same geometry as the chain kernel); its LDS window holds 64 chunks: the
// chunk in front of the tile (entry state of the first lane), the 60 owned chunks, three chunks behind them (matches
// that end past the tile; anything longer is read from L2/HBM).  Lane l = 1..60 owns window chunk l:
//   A  window: four coalesced 16-byte buffer loads per lane (one tile ahead), written to LDS with a 68-byte chunk
//      stride (17 dwords: the 32 lanes of a DS access at the same chunk offset land on 32 banks).
//   E  entry state: walk window chunk l-1 from the "any state" row of the table; on text the set of possible states
//      collapses to ONE state within a few bytes.  A lane whose set did not collapse raises the fallback flag.
//   R  replay chunk l from the entry state: one class lookup + one table lookup per byte (LDS); rare events (a match
//      was created, grew, was committed) update the lane's rows; then walk on until nothing can change them any more.
//   S  rows of the tile in lane order; starts by the reverse DFA from each end, bounded by the previous row's end.
// A workgroup (4 waves) takes one group of 32 wave-tiles; after one barrier the rows are ordered, looked back
// (block_common.hpp) and written as coalesced 16-byte stores.  The first row of a GROUP cannot see its predecessor's
// end inside the kernel: k_fsm_fix_heads checks those rows afterwards (one thread per group).
// Roofline: HBM-bound by design (each byte read once, 16 B per match written); in practice LDS-latency / VALU bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "block_common.hpp"
#include "fsm.hpp"
#include "scan_dfa.h"
#include "wave_common.hpp"

namespace cxgdev {

namespace {

constexpr int kFsmStride = 68;                       // LDS bytes per 64-byte chunk
constexpr int kFsmWinBytes = 64 * kFsmStride;        // 4352 per wave
constexpr int kFsmLeft = 64;                         // bytes staged in front of the tile
constexpr int kFsmRowsPerWave = 512;                 // rows buffered per wave and group
constexpr int32_t kFsmAhead = 57344;                 // a lane reads at most this far past its tile origin / before it

struct FsmMem {
  const uint8_t* win;      // this wave's LDS window
  const uint8_t* g;        // hay + tile_lo
  __device__ __forceinline__ uint32_t byte(int32_t r) const {
    const uint32_t w = static_cast<uint32_t>(r + kFsmLeft);
    if (w < 4096u) return win[w + (w >> 6) * 4u];
    return g[r];                                     // past the window: L2 / HBM (rare)
  }
  __device__ __forceinline__ uint32_t dword(int32_t r) const {
    const uint32_t w = static_cast<uint32_t>(r + kFsmLeft);
    return *reinterpret_cast<const uint32_t*>(win + w + (w >> 6) * 4u);
  }
};
struct GlobalMem {
  const uint8_t* g;        // hay - base: rows hold absolute offsets
  __device__ __forceinline__ uint32_t byte(int64_t r) const { return g[r]; }
};
struct LdsRows {
  uint16_t* slot;          // this lane's kFsmLaneRows ends
  __device__ __forceinline__ void set_end(uint32_t r, int32_t e) { slot[r] = static_cast<uint16_t>(e); }
};

__device__ __forceinline__ FsmView view_of(const uint8_t* body, const FsmHeader* h) {   // body = image without its header, in LDS
  FsmView v;
  const uint32_t hs = static_cast<uint32_t>(sizeof(FsmHeader));
  v.cls = body + (h->cls_off - hs);
  v.tab = reinterpret_cast<const uint16_t*>(body + (h->tab_off - hs));
  v.ev = reinterpret_cast<const uint16_t*>(body + (h->ev_off - hs));
  v.lev = body + (h->lev_off - hs);
  v.rev = body + (h->rev_off - hs);
  v.stride = h->stride; v.n_t = h->n_t; v.top_row = h->top_row; v.ncls = h->ncls;
  v.rev_start = h->rev_start; v.rev_first_accept = h->rev_first_accept;
  return v;
}

}  // namespace

__global__ __launch_bounds__(kThreads) void k_scan_fsm(ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_img[];           // tables of the image
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kWavesPerBlock][kFsmWinBytes];
  __shared__ uint16_t s_lrow[kWavesPerBlock][64 * kFsmLaneRows];            // per-lane row ends of the current tile
  __shared__ uint16_t s_re[kWavesPerBlock][kFsmRowsPerWave];                // rows of the group: end inside its wave-tile
  __shared__ uint16_t s_rl[kWavesPerBlock][kFsmRowsPerWave];                // ... and length
  __shared__ uint32_t s_cnt[kWavesPerBlock][kTilesPerWave];
  __shared__ uint32_t s_qbase[kWavesPerBlock * kTilesPerWave + 4];   // + 4: keeps the static LDS a multiple of 16 bytes (dynamic base alignment)
  __shared__ int64_t s_tail[kWavesPerBlock * kTilesPerWave];                // absolute end of a tile's last row, -1: no rows
  __shared__ uint64_t s_group;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) s_group = claim_group(a.static_groups != 0, a.ticket, a.ngroups);
  const FsmHeader* h = reinterpret_cast<const FsmHeader*>(a.blob);
  {
    const uint4* src = reinterpret_cast<const uint4*>(a.blob + sizeof(FsmHeader));
    const uint32_t nvec = h->lds_bytes >> 4;
    for (uint32_t i = tid; i < nvec; i += kThreads) reinterpret_cast<uint4*>(s_img)[i] = src[i];
  }
  __syncthreads();
  const uint64_t group = s_group;
  if (group >= a.ngroups) return;
  const FsmView v = view_of(s_img, h);
  constexpr int tpw = kTilesPerWave;
  uint32_t nrows_w = 0, fallback = 0, long_hit = 0;

  u32x4 x[4];
  auto issue_loads = [&](int jj) {
    const uint64_t wtn = group * (kWavesPerBlock * tpw) + static_cast<uint64_t>(jj) * kWavesPerBlock + wave;
    const uint64_t lo = wtn * static_cast<uint64_t>(kWaveTile);
    int nrec = 0;
    const int pre = lo ? 0 : kFsmLeft;                 // the first tile has nothing in front of it
    const uint64_t from = lo ? lo - kFsmLeft : 0;
    if (jj < tpw && lo < a.len) {
      const uint64_t rem = a.len - from;
      nrec = rem >= static_cast<uint64_t>(4096 - pre) ? 4096 - pre : static_cast<int>((rem + 3) & ~3ull);
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.hay) + (nrec ? from : 0), 0, nrec, 0x00020000);
#pragma unroll
    for (int k = 0; k < 4; k++)                        // window offsets below `pre` wrap around: out of range, read as zero
      x[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((lane + 64 * k) << 4) - pre, 0, 0);
  };
  issue_loads(0);

  for (int j = 0; j < tpw; j++) {
    const uint64_t wt = group * (kWavesPerBlock * tpw) + static_cast<uint64_t>(j) * kWavesPerBlock + wave;
    const uint64_t tile_lo = wt * static_cast<uint64_t>(kWaveTile);
    uint32_t tot = 0;
    if (tile_lo < a.len) {
      // ---- A: window into LDS (chunk stride 68 B)
      uint8_t* win = s_win[wave];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t wo = static_cast<uint32_t>(lane + 64 * k) << 4;
        uint32_t* d = reinterpret_cast<uint32_t*>(win + wo + (wo >> 6) * 4u);
        d[0] = x[k].x; d[1] = x[k].y; d[2] = x[k].z; d[3] = x[k].w;
      }
      issue_loads(j + 1);
      wave_lds_sync();
      const uint64_t remaining = a.len - tile_lo;
      const int32_t rend = remaining > 0x7FFF0000ull ? 0x7FFF0000 : static_cast<int32_t>(remaining);
      const int32_t budget = rend < kFsmAhead ? rend : kFsmAhead;
      const int32_t lowest = tile_lo > static_cast<uint64_t>(kFsmAhead) ? -kFsmAhead : -static_cast<int32_t>(tile_lo);
      FsmMem m{win, a.hay + tile_lo};
      // ---- E + R: entry state, replay
      const int32_t c0 = (lane - 1) * kFsmChunk, c1 = c0 + kFsmChunk;
      const bool active = lane >= 1 && lane <= kWaveTile / kFsmChunk && c0 < rend;
      FsmLane L;
      LdsRows rows{&s_lrow[wave][lane * kFsmLaneRows]};
      if (active) {
        uint32_t entry = 0;
        if (tile_lo + static_cast<uint64_t>(c0) > 0) entry = fsm_walk(v, m, v.top_row, c0 - kFsmChunk, c0, true);
        if (entry >= v.n_t) { fallback |= 1u; entry = 0; }
        fsm_replay(v, m, entry, c0, c1, rend, budget, L, rows);
        fallback |= L.flags << 1;
      }
      // ---- S: rows in lane order, then their starts
      const uint32_t nl = active ? L.nrows : 0u;
      const uint32_t incl = wave_inclusive_sum(nl);
      tot = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
      const uint32_t first = nrows_w + incl - nl;
      for (uint32_t r = 0; r < nl; r++)
        if (first + r < static_cast<uint32_t>(kFsmRowsPerWave)) s_re[wave][first + r] = rows.slot[r];
      wave_lds_sync();
      if (a.out != nullptr || a.max_len != 0) {
        for (uint32_t q = lane; q < tot && nrows_w + q < static_cast<uint32_t>(kFsmRowsPerWave); q += 64) {
          const int32_t e = s_re[wave][nrows_w + q];
          const int32_t bound = q ? static_cast<int32_t>(s_re[wave][nrows_w + q - 1]) : lowest;   // first row of the tile: checked after the barrier
          uint32_t over = 0;
          const int32_t s = fsm_match_start(v, m, e, bound, lowest, over);
          const uint32_t len = (s == kFsmNoStart) ? 0u : static_cast<uint32_t>(e - s);
          if (over || len == 0u || len > 0xFFFFu) fallback |= 16u;
          s_rl[wave][nrows_w + q] = static_cast<uint16_t>(len);
        }
      }
    }
    if (lane == 0) s_cnt[wave][j] = tot;
    nrows_w += tot;
  }
  if (nrows_w > static_cast<uint32_t>(kFsmRowsPerWave)) fallback |= 32u;
  {
    uint32_t f = fallback;                                                   // per-lane reasons -> one atomic per wave
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) f |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(f), d, 64));
    if (f != 0 && lane == 0) raise_err(a.err, 8u | (f << 8));
  }
  __syncthreads();

  // ---- order the group's rows: wave-tile q = j * 4 + wave
  if (tid < 64) {
    const int q = tid;
    const uint32_t c = (q < kWavesPerBlock * tpw) ? s_cnt[q % kWavesPerBlock][q / kWavesPerBlock] : 0u;
    const uint32_t incl = wave_inclusive_sum(c);
    if (q < kWavesPerBlock * tpw) s_qbase[q] = incl - c;
    if (q == kWavesPerBlock * tpw - 1) s_qbase[kWavesPerBlock * tpw] = incl;
  }
  const int64_t gorigin = static_cast<int64_t>(group * static_cast<uint64_t>(kWaveTile) * kWavesPerBlock * tpw);
  if (tid >= 64 && tid < 64 + kWavesPerBlock * tpw) {       // last row end of every tile (absolute), for the cross-tile check
    const int q = tid - 64, w = q % kWavesPerBlock, jj = q / kWavesPerBlock;
    uint32_t st = 0;
    for (int k = 0; k < jj; k++) st += s_cnt[w][k];
    const uint32_t n = s_cnt[w][jj];
    s_tail[q] = (n && st + n <= static_cast<uint32_t>(kFsmRowsPerWave)) ? gorigin + static_cast<int64_t>(q) * kWaveTile + s_re[w][st + n - 1] : -1;
  }
  __syncthreads();
  const uint32_t total = s_qbase[kWavesPerBlock * tpw];
  tile_lookback(a.status, a.total, a.err, group, a.ngroups, total, &s_base, a.epoch);
  const uint64_t base = s_base;
  uint32_t start = 0;
  for (int j = 0; j < tpw; j++) {
    const uint32_t n = s_cnt[wave][j];
    const int q = j * kWavesPerBlock + wave;
    const uint64_t dst = base + s_qbase[q];
    const int64_t tb = gorigin + static_cast<int64_t>(q) * kWaveTile;
    for (uint32_t i = lane; i < n; i += 64) {
      const uint32_t r = start + i;
      if (r >= static_cast<uint32_t>(kFsmRowsPerWave)) continue;
      int64_t e = tb + s_re[wave][r], s = e - s_rl[wave][r];
      if (i == 0 && (a.out != nullptr || a.max_len != 0)) {                 // the tile's first row was walked without a bound
        int64_t prev = -1;
        for (int p = q - 1; p >= 0 && prev < 0; p--) prev = s_tail[p];
        if (prev > s) {                                                     // it reached into the previous match: walk again, bounded (rare)
          GlobalMem gm{a.hay};
          uint32_t sr = v.rev_start;
          int64_t st = -1;
          for (int64_t at = e - 1; at >= prev; at--) {
            sr = v.rev[sr * v.ncls + v.cls[gm.byte(at)]];
            if (sr == 0u) break;
            if (sr >= v.rev_first_accept) st = at;
          }
          if (st < 0) raise_err(a.err, 8u | (16u << 8)); else s = st;
        }
      }
      if (a.max_len != 0 && static_cast<uint64_t>(e - s) > a.max_len) long_hit = 1;
      if (a.out != nullptr && dst + i < a.cap) {
        longlong2 o; o.x = a.base + s; o.y = a.base + e;
        *reinterpret_cast<longlong2*>(a.out + (dst + i) * a.row_width) = o;
      }
    }
    start += n;
  }
  if (long_hit) raise_err(a.err, kErrLongMatch);
}

// The first row of a group was walked back without knowing where the previous row (another workgroup's) ends.  One
// thread per group compares the two once every row is in HBM and, when the start reached into the previous match,
// walks the reverse DFA again with the bound (tables read from the image in HBM; this is rare by construction).
__global__ void k_fsm_fix_heads(ScanArgs a) {
  const uint64_t g = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
  if (g == 0 || g >= a.ngroups || a.out == nullptr) return;
  const uint64_t k = a.status[g - 1] & kValueMask, k1 = a.status[g] & kValueMask;   // inclusive counts after the scan kernel
  if (k1 == k || k == 0 || k >= a.cap) return;
  int64_t* row = a.out + k * a.row_width;
  const int64_t prev = a.out[(k - 1) * a.row_width + 1] - a.base, s0 = row[0] - a.base, e = row[1] - a.base;
  if (s0 >= prev) return;
  const FsmHeader* h = reinterpret_cast<const FsmHeader*>(a.blob);
  const uint8_t* cls = a.blob + h->cls_off;
  const uint8_t* rev = a.blob + h->rev_off;
  uint32_t sr = h->rev_start;
  int64_t st = -1;
  for (int64_t at = e - 1; at >= prev; at--) {
    sr = rev[sr * h->ncls + cls[a.hay[at]]];
    if (sr == 0u) break;
    if (sr >= h->rev_first_accept) st = at;
  }
  if (st < 0) { raise_err(a.err, 8u | (16u << 8)); return; }
  row[0] = a.base + st;
}

hipError_t launch_scan_fsm(const ScanArgs& a, uint32_t lds_bytes, hipStream_t stream) {
  const dim3 grid(static_cast<unsigned>(a.ngroups)), block(kThreads);
  hipLaunchKernelGGL(k_scan_fsm, grid, block, lds_bytes, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || a.out == nullptr || a.ngroups < 2) return e;
  const unsigned fb = 256, fg = static_cast<unsigned>((a.ngroups + fb - 1) / fb);
  hipLaunchKernelGGL(k_fsm_fix_heads, dim3(fg), dim3(fb), 0, stream, a);
  return hipGetLastError();
}

}  // namespace cxgdev
