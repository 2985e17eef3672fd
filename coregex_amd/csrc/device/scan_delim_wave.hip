// scan_delim_wave.hip — FindAll for "delimited" programs  O [^E]+ E  and  O [^E]* E  with two different bytes O and E:
// `\[[^\]]+\]`, `<[^>]+>`, `\([^)]*\)`, `\{[^}]+\}` (round 4).  On the transducer these cost 3.5-8.7 ms per GiB: no byte
// synchronises them (inside or outside the brackets?), every tile goes through the maps.
//
// Reference semantics (the DFA pair's leftmost-first FindAll, dfa/lazy/lazy.go:1102-1315 + 1769-1920 under meta/findall.go:216-283).
// E cuts the haystack into SEGMENTS (the bytes between two consecutive E).  A match that closes at an E starts at the FIRST
// eligible O of that E's segment — eligible: for `+` an O with a non-E byte behind it, for `*` any O — because the class [^E]
// also takes O, and FindAll resumes behind the E.  So:  rows = the E whose segment holds an eligible O, start = the first one.
// As a bit problem, with R = ~E and O' the eligible O (both as multiword integers over the haystack):  sum = R + O'.
// The lowest O' of a segment turns its bit into a carry that runs up the segment's ones and lands ON the closing E (the only
// zero): matched ends = sum & E; starts = O' & ~sum (later O' of the segment find their bit cleared and set it again, without a
// carry).  The k-th start and the k-th matched end of the haystack are row k — the char-class kernel's row machinery
// (scan_charclass_wave.hip: starts and ends owned separately, a row may span any number of tiles).
//
// What a tile cannot know alone is the carry that enters it: "an eligible O of the current segment lies in front of the tile".
// Under a carry-in of 1 the tile differs from the carry-less evaluation only below its first E: the start there (at most one)
// is none, and the first E is matched in any case.  Each tile therefore leaves its carry-less bitmaps, two correction bits and
// its KIND — the exit carry as a function of the entry carry: CONST 0 / CONST 1 (the tile holds an E: what follows the last E
// decides; or no E but an O') or PASS (neither) — and one wave chains the kinds: first across the workgroups (a look-back
// over 2-bit words, ScanArgs::status2, which waits for the predecessors' bitmaps only, not for their rows: every 60 KiB group
// of real text holds an E and ends the walk at distance 1), then over the group's 16 tiles.  Then counts, the usual look-back
// over row counts, and the rows.
// Fallback flag (err bit 8, the host reruns on the transducer kernel): more than 1024 starts or ends in a tile.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_common.hpp"
#include "scan_dfa.h"
#include "walk.hpp"
#include "wave_common.hpp"

namespace cxgdev {

namespace {
constexpr int kDWin = kWaveTile + kWaveHalo;       // 4096
constexpr int kDStage = 1024;                      // rows staged per wave-tile
constexpr int kDTiles = kCcTilesPerWave;           // 4 tiles per wave: 60 KiB groups, as the char-class kernel
constexpr uint32_t kKind0 = 0u, kKind1 = 1u, kKindPass = 2u;

// 0x80 in every byte of x equal to the byte splat in b4
__device__ __forceinline__ uint32_t eq4(uint32_t x, uint32_t b4) {
  const uint32_t y = x ^ b4;
  return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
}
__device__ __forceinline__ uint32_t eqpiece16(const u32x4& x, uint32_t b4) {
  const uint32_t lo = __builtin_amdgcn_udot4(eq4(x.y, b4), 0x80402010u, __builtin_amdgcn_udot4(eq4(x.x, b4), 0x08040201u, 0u, false), false);
  const uint32_t hi = __builtin_amdgcn_udot4(eq4(x.w, b4), 0x80402010u, __builtin_amdgcn_udot4(eq4(x.z, b4), 0x08040201u, 0u, false), false);
  return (lo >> 7) | (hi << 1);
}
// lowest set bit position over the wave's 64 words (word l = bits 64 l ..): 4096 when none
__device__ __forceinline__ uint32_t wave_lowest_bit(uint64_t w) {
  const unsigned long long any = __ballot(w != 0ull);
  if (any == 0ull) return 4096u;
  const int l = __builtin_ctzll(any);
  const uint64_t ww = readlane64(w, l);
  return static_cast<uint32_t>(64 * l + __builtin_ctzll(ww));
}
}  // namespace

__global__ __launch_bounds__(kThreads, 4) void k_scan_delim_wave(ScanArgs a) {
  __shared__ __attribute__((aligned(16))) uint64_t s_o[kWavesPerBlock][64];                 // pieces -> words (O, then E)
  __shared__ __attribute__((aligned(16))) uint64_t s_e[kWavesPerBlock][64];
  __shared__ __attribute__((aligned(16))) uint64_t s_S[kWavesPerBlock][kDTiles][64];        // carry-less starts
  __shared__ __attribute__((aligned(16))) uint64_t s_E[kWavesPerBlock][kDTiles][64];        // carry-less matched ends (bit ON the E)
  __shared__ uint16_t s_rs[kWavesPerBlock][kDStage];
  __shared__ uint16_t s_re[kWavesPerBlock][kDStage];
  __shared__ uint32_t s_info[kWavesPerBlock][kDTiles];     // ns0 | ne0 << 12 | dS << 24 | dE << 25 | kind << 26
  __shared__ uint32_t s_fe[kWavesPerBlock][kDTiles];       // position of the tile's first E (4096: none)
  __shared__ uint32_t s_cin[kWavesPerBlock * kDTiles];     // entry carry of tile q = j * 4 + wave
  __shared__ uint32_t s_qbase[kWavesPerBlock * kDTiles + 1];
  __shared__ uint32_t s_exit;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = lane0;
  const uint64_t group = blockIdx.x;                                // static groups only (capi_ladder.hip)
  if (group >= a.ngroups) return;
  const DelimAux* ax = reinterpret_cast<const DelimAux*>(a.chain);
  const uint32_t o4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ax->open_byte * 0x01010101u)));
  const uint32_t e4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ax->close_byte * 0x01010101u)));
  const bool plus = ax->plus != 0u;
  uint32_t fallback = 0;

  u32x4 x[4];
  auto issue_loads = [&](int jj) {
    const uint64_t wtn = group * (kWavesPerBlock * kDTiles) + static_cast<uint64_t>(jj) * kWavesPerBlock + wave;
    const uint64_t lo = wtn * static_cast<uint64_t>(kWaveTile);
    int nrec = 0;
    if (jj < kDTiles && lo < a.len) {
      const uint64_t rem = a.len - lo;
      nrec = rem >= static_cast<uint64_t>(kDWin) ? kDWin : static_cast<int>((rem + 3) & ~3ull);
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.hay) + (nrec ? lo : 0), 0, nrec, 0x00020000);
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (lane + 64 * k) << 4, 0, 0);
  };
  issue_loads(0);

  // ---- pass 1: carry-less bitmaps, counts, kinds
  for (int j = 0; j < kDTiles; j++) {
    lane = lane0;
    asm volatile("" : "+v"(lane));
    const uint64_t wt = group * (kWavesPerBlock * kDTiles) + static_cast<uint64_t>(j) * kWavesPerBlock + wave;
    const uint64_t tile_lo = wt * static_cast<uint64_t>(kWaveTile);
    uint64_t S = 0, Em = 0;
    uint32_t info = kKindPass << 26, fe = 4096u;
    if (tile_lo < a.len) {
      const uint64_t remaining = a.len - tile_lo;
      const int32_t stage = remaining < static_cast<uint64_t>(kDWin) ? static_cast<int32_t>(remaining) : kDWin;
      uint16_t* po = reinterpret_cast<uint16_t*>(s_o[wave]);
      uint16_t* pe = reinterpret_cast<uint16_t*>(s_e[wave]);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        po[lane + 64 * k] = static_cast<uint16_t>(eqpiece16(x[k], o4));
        pe[lane + 64 * k] = static_cast<uint16_t>(eqpiece16(x[k], e4));
        __builtin_amdgcn_sched_barrier(0);
      }
      issue_loads(j + 1);
      wave_lds_sync();
      uint64_t Ow = s_o[wave][lane], Ew = s_e[wave][lane];
      {                                                             // nothing past the data; (zero bytes read past the end equal neither byte unless the byte is 0)
        const int32_t nv = stage - 64 * lane;
        const uint64_t vf = nv <= 0 ? 0ull : (nv >= 64 ? ~0ull : ((1ull << nv) - 1ull));
        Ow &= vf; Ew &= vf;
      }
      if (plus) Ow &= ~((Ew >> 1) | (from_upper64(Ew) << 63));      // `+`: an O with the E right behind it opens nothing  (lane 63's neighbour: its own word — bit 4095 is not owned)
      const uint64_t own = word_range(lane, 0, kWaveTile - 1);      // events are owned by position: [0, 3840)
      Ow &= own; Ew &= own;
      // sum = R + O' over the 4096-bit window, R = ~E (ones beyond the owned bytes: the carry out of bit 3839 leaves through the top)
      const uint64_t R = ~Ew;
      const uint32_t r0 = static_cast<uint32_t>(R), r1 = static_cast<uint32_t>(R >> 32);
      uint32_t s0, s1;
      unsigned long long GG;
      {
        unsigned long long c0;
        asm("v_add_co_u32_e64 %0, %2, %4, %5\n\tv_addc_co_u32_e64 %1, %3, %6, %7, %2"
            : "=&v"(s0), "=&v"(s1), "=&s"(c0), "=&s"(GG)
            : "v"(r0), "v"(static_cast<uint32_t>(Ow)), "v"(r1), "v"(static_cast<uint32_t>(Ow >> 32)));
      }
      const unsigned long long PP = __builtin_amdgcn_uicmpl(R, ~0ull, 32 /*eq*/) & ~GG;     // words of 64 ones pass a carry on
      const unsigned long long recv = (PP + (GG << 1)) ^ PP;        // lanes that receive a carry
      const uint32_t g0 = static_cast<uint32_t>(((GG | (PP & recv)) >> 63) & 1ull);          // the carry that leaves through the top
      const uint64_t sum = add_carry_mask((static_cast<uint64_t>(s1) << 32) | s0, recv);
      S = Ow & ~sum;
      Em = sum & Ew;
      const uint32_t ns = static_cast<uint32_t>(__popcll(S)), ne = static_cast<uint32_t>(__popcll(Em));
      const uint32_t incl = wave_inclusive_sum(ns | (ne << 16));
      const uint32_t tot = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
      const uint32_t ns0 = tot & 0xFFFFu, ne0 = tot >> 16;
      fe = wave_lowest_bit(Ew);
      const uint32_t fs = wave_lowest_bit(S);
      const uint32_t fm = wave_lowest_bit(Em);
      const uint32_t dS = fs < fe ? 1u : 0u;                        // under a carry-in that start is none (fs < fe also when there is no E: fe = 4096)
      const uint32_t dE = (fe != 4096u && fm != fe) ? 1u : 0u;      // ... and the first E is matched although no O' of this tile precedes it
      const bool anyO = __ballot(Ow != 0ull) != 0ull;
      const uint32_t kind = fe != 4096u ? g0 : (anyO ? kKind1 : kKindPass);
      if (ns0 + 1u > static_cast<uint32_t>(kDStage) || ne0 + 1u > static_cast<uint32_t>(kDStage)) fallback |= 8;
      info = ns0 | (ne0 << 12) | (dS << 24) | (dE << 25) | (kind << 26);
    }
    s_S[wave][j][lane] = S;
    s_E[wave][j][lane] = Em;
    if (lane == 0) { s_info[wave][j] = info; s_fe[wave][j] = fe; }
  }
  if (fallback != 0 && lane0 == 0) raise_err(a.err, 8u | (fallback << 8));
  __syncthreads();

  // ---- the group's kind, the carry that enters it (look-back over kinds), the carries of its tiles, the row counts
  if (wave == 0) {
    const uint64_t etag = static_cast<uint64_t>(a.epoch) << 32;
    uint32_t gk = kKindPass;                                        // composition over the tiles in order: a CONST overrides what came before
    for (int q = 0; q < kWavesPerBlock * kDTiles; q++) {
      const uint32_t k = (s_info[q % kWavesPerBlock][q / kWavesPerBlock] >> 26) & 3u;
      if (k != kKindPass) gk = k;
    }
    if (lane0 == 0) __hip_atomic_store(a.status2 + group, etag | 4u | gk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t cin = 0;
    if (group > 0) {
      int64_t look = static_cast<int64_t>(group) - 1;
      uint32_t spins = 0;
      for (;;) {
        const int64_t idx = look - lane0;
        uint64_t w = etag | 4u | kKind0;                            // in front of the haystack: no carry
        if (idx >= 0) w = __hip_atomic_load(a.status2 + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ready = (w >> 32) == (etag >> 32) && (w & 4u) != 0u;
        if (!__all(ready)) {
          if (++spins > kSpinLimit) { if (lane0 == 0) raise_watchdog(a.err, kWdDelim); break; }
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        const unsigned long long consts = __ballot((w & 3u) != kKindPass);
        if (consts != 0ull) { cin = static_cast<uint32_t>(readlane64(w, __builtin_ctzll(consts))) & 3u; break; }
        look -= 64;
      }
      // a group that only passes its carry on says which one it is, now that it knows: the groups behind stop here instead of
      // walking to the last E of the haystack (`<[^>]+>` on text without a `>`: 3 -> 1 ms per GiB with the walk, less with this)
      if (gk == kKindPass && lane0 == 0) __hip_atomic_store(a.status2 + group, etag | 4u | cin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane0 == 0) {
      uint32_t c = cin, base = 0;
      for (int q = 0; q < kWavesPerBlock * kDTiles; q++) {
        const uint32_t inf = s_info[q % kWavesPerBlock][q / kWavesPerBlock];
        s_cin[q] = c;
        s_qbase[q] = base;
        base += (inf & 0xFFFu) - (c ? ((inf >> 24) & 1u) : 0u);     // starts of the tile under its entry carry
        const uint32_t k = (inf >> 26) & 3u;
        if (k != kKindPass) c = k;
      }
      s_qbase[kWavesPerBlock * kDTiles] = base;
      s_exit = c;
    }
  }
  __syncthreads();
  const uint32_t total = s_qbase[kWavesPerBlock * kDTiles];
  tile_lookback(a.status, a.total, a.err, group, a.ngroups, total, &s_base, a.epoch);
  if (group == a.ngroups - 1 && tid == 0) *a.total = s_base + total - s_exit;   // rows = matched ends: an opening without its E at the end of the haystack is none
  if (a.out == nullptr) return;

  // ---- pass 2: the tiles under their entry carries; starts and ends straight to their rows
  const uint64_t base = s_base;
  const bool u32 = a.u32_rows != 0u;
  uint32_t* const out32 = reinterpret_cast<uint32_t*>(a.out);
  const int64_t origin = (u32 ? 0 : a.base) + static_cast<int64_t>(group * static_cast<uint64_t>(kWaveTile) * kWavesPerBlock * kDTiles);
  for (int j = 0; j < kDTiles; j++) {
    const int q = j * kWavesPerBlock + wave;
    const uint32_t inf = s_info[wave][j];
    const uint32_t open = s_cin[q];
    uint64_t S = s_S[wave][j][lane0], E = s_E[wave][j][lane0];
    if (open) {
      const uint32_t fe = s_fe[wave][j];
      if ((inf >> 24) & 1u) {                                       // the start below the first E is none
        const uint32_t fs = wave_lowest_bit(S);
        if (static_cast<uint32_t>(lane0) == (fs >> 6)) S &= ~(1ull << (fs & 63u));
      }
      if (fe != 4096u && static_cast<uint32_t>(lane0) == (fe >> 6)) E |= 1ull << (fe & 63u);   // the first E closes the row opened in front of the tile
    }
    const uint32_t n = (inf & 0xFFFu) - (open ? ((inf >> 24) & 1u) : 0u);
    const uint32_t n_ends = ((inf >> 12) & 0xFFFu) + (open ? ((inf >> 25) & 1u) : 0u);
    if (n == 0 && n_ends == 0) continue;
    const uint32_t ns = static_cast<uint32_t>(__popcll(S)), ne = static_cast<uint32_t>(__popcll(E));
    const uint32_t incl = wave_inclusive_sum(ns | (ne << 16));
    const uint64_t row0 = base + s_qbase[q];
    uint32_t r = (incl & 0xFFFFu) - ns;
    uint64_t sb = S;
    while (sb) {
      const int bit = __builtin_ctzll(sb);
      sb &= sb - 1;
      if (r < static_cast<uint32_t>(kDStage)) s_rs[wave][r] = static_cast<uint16_t>(64 * lane0 + bit);
      r++;
    }
    r = (incl >> 16) - ne;
    uint64_t eb = E;
    while (eb) {
      const int bit = __builtin_ctzll(eb);
      eb &= eb - 1;
      if (r < static_cast<uint32_t>(kDStage)) s_re[wave][r] = static_cast<uint16_t>(64 * lane0 + bit + 1);   // exclusive end: behind the E
      r++;
    }
    wave_lds_sync();
    const int64_t tb = origin + static_cast<int64_t>(q) * kWaveTile;
    const uint32_t nst = n < static_cast<uint32_t>(kDStage) ? n : static_cast<uint32_t>(kDStage);
    const uint32_t nen = n_ends < static_cast<uint32_t>(kDStage) ? n_ends : static_cast<uint32_t>(kDStage);
    for (uint32_t i = lane0; i < nst; i += 64) {
      if (row0 + i < a.cap) {
        if (i + open < nen) {                                       // both halves of the row are this tile's
          const int64_t vs = tb + s_rs[wave][i], ve = tb + s_re[wave][i + open];
          if (u32) store_pair32_nt(out32 + (row0 + i) * 2, static_cast<uint32_t>(vs), static_cast<uint32_t>(ve));
          else store_pair_nt(a.out + (row0 + i) * 2, vs, ve);
        } else if (u32) out32[(row0 + i) * 2] = static_cast<uint32_t>(tb + s_rs[wave][i]);
        else a.out[(row0 + i) * 2] = tb + s_rs[wave][i];           // closed in a later tile
      }
    }
    if (open && nen != 0 && lane0 == 0 && row0 - 1 < a.cap) {       // a row opened in an earlier tile closes here
      if (u32) out32[(row0 - 1) * 2 + 1] = static_cast<uint32_t>(tb + s_re[wave][0]);
      else a.out[(row0 - 1) * 2 + 1] = tb + s_re[wave][0];
    }
    wave_lds_sync();
  }
}

hipError_t launch_scan_delim_wave(const ScanArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(k_scan_delim_wave, dim3(static_cast<unsigned>(a.ngroups)), dim3(kThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace cxgdev
