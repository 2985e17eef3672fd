// scan_teddy_pair.hip — FindAll for UseTeddy (exact literal alternation, <= 64 prefix-free literals of >= 3 bytes), third
// generation (round 6): the fingerprint is looked up per byte PAIR, on a persistent grid of one 16-wave workgroup per CU.
//
// Reference semantics kept (meta/find_indices.go:925-951 -> prefilter.Teddy.FindMatch, prefilter/teddy.go:391-444,
// verifyBucket :532-550; FindAll advance meta/findall.go:267-275): the next match is at the first fingerprint candidate at or
// after `pos` at which a literal of a hit bucket compares equal; the search resumes at its end.  Any superset of the true
// match starts is a valid candidate set — candidates are verified exactly; the set is prefix-free, so at most one literal
// matches at a position and the order in which literals are tried (here: by first byte, then id) cannot be observed.
//
// Why pairs.  scan_teddy_wave.hip pays four instructions per haystack byte for its filter (address, LDS lookup, two SDWA
// ANDs) and 1.4 more to gather the flags into bits: 350 of a wave-tile's 750 instructions, and the kernel is bound by
// instruction issue.  Here one LDS lookup serves TWO bytes: a table of 65 536 one-byte entries, indexed
// by the 16 bits of an aligned byte pair (b0 | b1 << 8), says everything a literal start at either byte of the pair or at
// the four bytes in front of it needs to know about the pair:
//   bit 0 AB  (b0, b1) are bytes 0, 1 of a literal          bit 1 A2  b1 is byte 0 of a literal
//   bit 2 CD  (b0, b1) are bytes 2, 3 of a literal          bit 3 BC  (b0, b1) are bytes 1, 2 of a literal
//   bit 4 E1  b0 is byte 4 of a literal                     bit 5 DE  (b0, b1) are bytes 3, 4 of a literal
//   bit 6 S1  b0 is outside the literals' alphabet          bit 7 S2  b1 is outside it
// (a literal shorter than the byte asked for: every value qualifies).  With W_k the entry of pair k, a literal can start
//   at the pair's first byte  iff  AB(W_k) & CD(W_k+1) & E1(W_k+2),   at its second byte  iff  A2(W_k) & BC(W_k+1) & DE(W_k+2):
// both at once as W_k & (W_k+1 >> 2) & (W_k+2 >> 4) & 3 — on four pairs per instruction, because the entries of four pairs
// sit in the bytes of one register.  A five-byte fingerprint of exact pairs: on BASELINE config 3 (16 literals) 20.6
// candidates per 3 840-byte tile against 15.8 matches (the three-byte, eight-bucket fingerprint of scan_teddy_wave.hip: 21.8).
// Per 64 bytes of a lane: 32 lookups + 64 address instructions (the bank swizzle, pair_addr below) + 24 packs, then 5
// instructions per 8 bytes to combine and 2 v_dot4 per 8 bytes to make the candidate and synchronising bits dense.
//
// The table is 64 KiB of LDS, so a CU holds ONE workgroup: 16 waves, and the grid is persistent.  GROUPS — 16 units of 8
// consecutive wave-tiles (480 KiB), and small ones of 2-tile units for the haystack's last stretch — are CLAIMED through
// one atomic counter, two claims ahead, the first two groups of a workgroup without an atomic (the ticket is drawn by one lane a group early; this file is compiled with
// -amdgpu-atomic-optimizer-strategy=None: the optimizer's wave reduction waited for the ticket, and for every window load
// in flight, on the spot).  A workgroup that holds group g only ever waits for groups < g, which are held by running
// workgroups: forward progress does not depend on co-residency or on the order of dispatch.  The
// tables come with the program image (built on the host: program.cc buildPairImage).  A wave scans one unit per group behind ONE buffer descriptor, in a two-tile loop body:
// filter(j + 1) runs between the REQUEST of the 16 bytes at each candidate of tile j (global memory; there is no room for an
// LDS copy of the window) and their comparison.  The rows of a group are ordered with the decoupled look-back of
// block_common.hpp, deferred by one group: a group's count is published behind the iteration's one barrier, its base is
// resolved by wave 0 a group later (the status words requested a group earlier), every wave writes its own unit's rows then.
//
// Window, ownership (synchronising bytes), candidate ranking and FindAll order are those of scan_teddy_wave.hip.  The kernel
// is bound by the number of instructions it issues (four waves per SIMD: instruction types hardly overlap) — hence the
// dump slots instead of exec-masked branches, the uniform verification loop, the 32-bit tile arithmetic.
// Fallback flag (err bit 8; capi_ladder.hip reruns the call on scan_teddy_wave.hip): no synchronising byte in a halo,
// > 192 owned candidates in a wave-tile, row buffer overflow, an assertion that needs a byte behind the window.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_common.hpp"
#include "scan_dfa.h"
#include "walk.hpp"
#include "wave_common.hpp"

namespace cxgdev {

namespace {

constexpr int kPWaves = kPairWaves;               // 16
constexpr int kPThreads = kPWaves * 64;           // 1024
constexpr int kPTpw = kPairTilesPerWave;          // 8 consecutive wave-tiles per wave and group: a unit
constexpr int kPRows = 320;                       // rows buffered per wave per group (8 tiles; scan_teddy_wave.hip's figure)
constexpr int kPCands = 192;                      // owned candidates listed per wave-tile
constexpr int kPAuxMax = 2048;
#ifndef CXG_PAIR_WAUX
#define CXG_PAIR_WAUX 0                   // cache policy bits of the window loads / of the verifier's requests (A/B builds: 1 sc0, 2 nt, 16 sc1)
#endif
#ifndef CXG_PAIR_VAUX
#define CXG_PAIR_VAUX 2                   // (nt: the requests miss the L2 either way — measured —, a nontemporal miss fetches half as much: FETCH_SIZE 1.50 -> 1.26 x the haystack, same time)
#endif
#ifndef CXG_PAIR_ABL
#define CXG_PAIR_ABL 0                    // timing experiments (scripts/build_variant.sh; WRONG rows): 1 no verification, 2 no candidate list either, 4 no table lookups, 8 no look-back / row write, 64 no requests of candidate bytes
#endif
constexpr int32_t kPFar = 1 << 20;
constexpr int kPWin = kWaveTile + kWaveHalo;      // 4096
static_assert(kPTpw % 2 == 0 && kWaveTile * kPTpw + kWaveHalo + 256 < 65536, "rows are kept as 16-bit offsets into the unit");

struct PairWaveLds {
  uint32_t w[512];                                // pair entries of the window: piece p (16 bytes) -> dwords 2p, 2p + 1
  uint16_t cpos[2][kPCands + 2];                      // owned candidates of the tile being verified and of the tile being filtered
  uint16_t rs[2][kPRows + 2], re[2][kPRows + 2];          // rows of this unit (relative to its first byte) and of the unit of the group before
  uint16_t ce[64];
  uint8_t em[64];
};
struct PairLds {
  PairImage img;                                  // at LDS address 0 (the pair is the address): pair table, slot ranges by first byte, slot records — copied from the program image
  __attribute__((aligned(16))) uint8_t aux[kPAuxMax];
  uint64_t base[2];                               // output base of the group before (two groups alternate)
  uint32_t gq[4];                                 // ring of claimed groups (three in use)
  uint32_t tot[2];
  uint32_t wcnt[2][kPWaves], woff[2][kPWaves];    // rows of every wave's unit, and their exclusive sums inside the group
  PairWaveLds wv[kPWaves];
};
static_assert(sizeof(PairLds) <= 160 * 1024, "LDS");

// (pair_addr — the table's address swizzle — : walk.hpp)
// ... of the low / high pair of a dword: two SDWA operations each (shift of byte 1 / 3, XOR with word 0 / 1)
#define CXG_PAIR_ADDR(dst, x, BYTE, WORD) \
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #BYTE "\n\t" \
      "v_xor_b32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_" #WORD : "=&v"(dst) : "v"(two), "v"(x))

// status word of the look-back (block_common.hpp): wave 0
__device__ __forceinline__ uint64_t pair_status_load(const uint64_t* status, int64_t idx, uint64_t etag) {
  return idx >= 0 ? __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (kFlagInclusive | etag);
}
// exclusive base of `group` (> 0) from the words in front of it; w = the words of groups group-1-lane, loaded earlier
__device__ __forceinline__ uint64_t pair_resolve(const uint64_t* status, uint32_t* err, uint64_t group, uint64_t w, uint64_t etag, int lane) {
  uint64_t base = 0;
  int64_t look = static_cast<int64_t>(group) - 1;
  uint32_t spins = 0;
  for (;;) {
    const bool ready = (w & kFlagMask) != 0 && (w & kEpochMask) == etag;
    if (!__all(ready)) {
      if (++spins > kSpinLimit) { if (lane == 0) raise_watchdog(err, kWdLookback); break; }
      __builtin_amdgcn_s_sleep(2);
      w = pair_status_load(status, look - lane, etag);
      continue;
    }
    const unsigned long long incl_mask = __ballot((w & kFlagMask) == kFlagInclusive);
    const int first_incl = incl_mask ? __builtin_ctzll(incl_mask) : 64;
    // counts (aggregate words: a unit's rows, far below 2^32) of the units behind the first inclusive word: one DPP sum; the inclusive word itself: two v_readlane
    const uint32_t agg = lane < first_incl ? static_cast<uint32_t>(w) : 0u;
    base += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wave_inclusive_sum(agg)), 63));
    if (first_incl < 64) base += readlane64(w & kValueMask, first_incl);
    if (first_incl < 64) break;
    look -= 64;
    w = pair_status_load(status, look - lane, etag);
  }
  return base;
}

}  // namespace

__global__ __launch_bounds__(kPThreads, 4) void k_scan_teddy_pair(ScanArgs a) {
  __shared__ PairLds S;
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = lane0;
  PairWaveLds& L = S.wv[wave];
  const uint64_t etag = static_cast<uint64_t>(a.epoch) << kEpochShift;

  // ---- once per workgroup: claim two groups, the literal tables, the pair table
  // Groups are claimed from pair_nctr counters — ONE in the product (below); with several (A/B builds of the ladder) counter x hands out groups
  // x, x + nctr, .. and workgroup b asks counter b & (nctr - 1) and, once that is exhausted, the others in turn: that form can deadlock a
  // workgroup on its own look-back.  (One counter serves ~20 returning atomics per microsecond: a GiB has 2 200 groups.)
  const uint32_t nctr = a.pair_nctr, cls = blockIdx.x & (nctr - 1u);
  const uint32_t ngroups32 = static_cast<uint32_t>(a.ngroups);
  const uint64_t ngroups = a.ngroups;
  uint32_t* const ctr = a.pair_ctr + (a.pair_seq & 1u) * (8u * kPairCtrStride);
  if (blockIdx.x == 0 && tid < 8) a.pair_ctr[((a.pair_seq + 1u) & 1u) * (8u * kPairCtrStride) + static_cast<uint32_t>(tid) * kPairCtrStride] = 0u;   // the next launch's set
  auto draw = [&]() -> uint32_t { return __hip_atomic_fetch_add(ctr + cls * kPairCtrStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  // One counter (the product): workgroup b begins with groups b and b + G (G workgroups) without asking anybody — 2 G simultaneous atomics on
  // one address would take the last of them 25 us —, ticket t is group 2 G + t: every workgroup's groups ascend, the order of the groups is
  // the order in which they were handed out, and the smallest uncounted group is always being scanned by a workgroup that waits for nobody.
  const uint32_t G2 = nctr == 1u ? 2u * gridDim.x : 0u;
  auto claimed = [&](uint32_t t) -> uint32_t {                      // ticket of the own counter -> group; 0xFFFFFFFF: no group left anywhere
    if (nctr == 1u) return t + G2 < ngroups32 && t + G2 >= G2 ? t + G2 : 0xFFFFFFFFu;
    uint32_t g = t * nctr + cls;
    for (uint32_t k = 1; g >= ngroups32 && k < nctr; k++) {
      const uint32_t x = (cls + k) & (nctr - 1u);
      g = __hip_atomic_fetch_add(ctr + x * kPairCtrStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * nctr + x;
    }
    return g < ngroups32 ? g : 0xFFFFFFFFu;
  };
  uint32_t t0 = 0, t1 = 0;
  if (tid == 0 && nctr != 1u) { t0 = draw(); t1 = draw(); }       // (consumed behind the image's load)
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(a.blob);
  for (uint32_t i = tid; i < h->aux_len / 4 && i < kPAuxMax / 4; i += kPThreads)
    reinterpret_cast<uint32_t*>(S.aux)[i] = reinterpret_cast<const uint32_t*>(a.blob + h->aux_off)[i];
  if (tid == 0) {                                                   // (the tickets have had the image's load to come back)
    uint32_t g0, g1;
    if (nctr == 1u) { g0 = blockIdx.x < ngroups32 ? blockIdx.x : 0xFFFFFFFFu; g1 = blockIdx.x + gridDim.x < ngroups32 ? blockIdx.x + gridDim.x : 0xFFFFFFFFu; }
    else { g0 = claimed(t0); g1 = g0 == 0xFFFFFFFFu ? g0 : claimed(t1); }
    S.gq[0] = g0; S.gq[1] = g1;
  }
  __syncthreads();
  // (The first window is asked for HERE, in front of the table build: its latency hides behind it.)
  // Window loads: four buffer_load_dwordx4 per lane, one tile ahead — across groups too.  (Two windows in flight were measured: no gain —
  // the kernel is bound by the number of instructions it issues, not by the latency of its loads.)  ONE buffer descriptor per unit
  // (a wave's kPTpw consecutive tiles + the last one's halo, 16 bytes in front for the byte before the unit; zeros past the end of input):
  // a tile costs one add, not a descriptor (the 64-bit tile arithmetic was ~70 scalar instructions of a tile's 730).
  struct UnitGeo { __amdgpu_buffer_rsrc_t rsrc; int32_t pre; int32_t rem; int32_t nrec; int32_t tiles; uint64_t lo; };   // rem: bytes from the unit's first byte to the end of input (0: none; clamped); tiles: 8 in a big group, 2 in a small one
  auto make_unit = [&](uint64_t g) -> UnitGeo {
    UnitGeo u;
    // groups [0, n8) have 8 tiles per wave, the next n6 have 6, the next n4 have 4, the rest 2: one behind the other in the haystack
    const uint64_t n8 = a.pair_nbig, n6 = a.pair_n6, n4 = a.pair_n4;
    const uint64_t g6 = g > n8 ? g - n8 : 0ull, g4 = g6 > n6 ? g6 - n6 : 0ull, g2 = g4 > n4 ? g4 - n4 : 0ull;
    u.tiles = g < n8 ? 8 : (g6 < n6 ? 6 : (g4 < n4 ? 4 : 2));
    // tile rows (16 tiles: one per wave) in front of the group
    const uint64_t rows = 8ull * (g < n8 ? g : n8) + 6ull * (g6 < n6 ? g6 : n6) + 4ull * (g4 < n4 ? g4 : n4) + 2ull * g2;
    const uint64_t ulo = (rows * kPWaves + static_cast<uint64_t>(wave) * static_cast<uint64_t>(u.tiles)) * static_cast<uint64_t>(kWaveTile);
    u.lo = ulo;
    const bool any = g < ngroups && ulo < a.len;
    const uint64_t rem64 = any ? a.len - ulo : 0ull;
    u.rem = rem64 > 0x7FFF0000ull ? 0x7FFF0000 : static_cast<int32_t>(rem64);
    const int32_t span = u.rem < kWaveTile * u.tiles + kWaveHalo ? u.rem : kWaveTile * u.tiles + kWaveHalo;
    u.pre = (any && ulo) ? 16 : 0;
    u.nrec = ((span + 3) & ~3) + u.pre;
    u.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.hay) + (any ? ulo - u.pre : 0), 0, u.nrec, 0x00020000);
    return u;
  };
  u32x4 x[4];
  uint32_t xprev = 0;
  auto issue_loads = [&](const UnitGeo& u, int jj) {
    const int32_t soff = u.pre + jj * kWaveTile;
    const int32_t vo = (lane << 4) + soff;
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = __builtin_amdgcn_raw_buffer_load_b128(u.rsrc, vo + (k << 10), 0, CXG_PAIR_WAUX);
    xprev = __builtin_amdgcn_raw_buffer_load_b32(u.rsrc, soff ? soff - 4 : u.nrec, 0, 0);   // (the haystack's first tile: no byte in front, an offset outside the descriptor reads 0)
  };
  UnitGeo cur = make_unit(S.gq[0]), nxt = cur;
  issue_loads(cur, 0);

  const TeddyAux* ax = reinterpret_cast<const TeddyAux*>(S.aux);
  const uint16_t* t_off = reinterpret_cast<const uint16_t*>(S.aux + ax->off_off);
  const uint8_t* t_bytes = S.aux + ax->bytes_off;
  const uint32_t look_pre = ax->looks & 0xFFu, look_post = (ax->looks >> 8) & 0xFFu;
  const bool fold = (ax->looks & kTeddyFold) != 0u;
  auto same = [&](uint32_t b, uint32_t c) { return b == c || (fold && c >= 'a' && c <= 'z' && (b | 0x20u) == c); };   // byte b stands for literal byte c
  {                                                                 // the tables, built on the host once per program (program.cc buildPairImage): 67 KiB per workgroup from L2
    const u32x4* src = reinterpret_cast<const u32x4*>(a.blob + ax->pair_off);
    u32x4* dst = reinterpret_cast<u32x4*>(&S.img);
    for (uint32_t i = tid; i < sizeof(PairImage) / 16; i += kPThreads) dst[i] = src[i];
  }
  __syncthreads();

  uint32_t fallback = 0, edge_hit = 0;
  const uint32_t maxrun = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(S.img.maxrun)));
  const bool short_lits = ax->maxlen <= 12u;                         // every literal is compared whole by the three masked dwords

  uint64_t prev = ~0ull, prev_lo = 0;                               // the group whose rows wait to be written, and where this wave's unit of it begins
  for (uint32_t it = 0;; it++) {
    const uint32_t b = it & 1u;
    const uint64_t group = S.gq[it % 3u];
    const uint64_t next_group = S.gq[(it + 1u) % 3u];
    const bool live = group < ngroups;
    if (it) cur = nxt;                                              // (built when this unit's first window was asked for)
    uint32_t n2 = 0;
    uint64_t lw = 0;
    if (wave == kPWaves - 1 && live && lane0 == 0) n2 = draw();      // the group after next: the ticket is read in front of the barrier
    if (wave == 0 && prev != ~0ull && prev > 0) lw = pair_status_load(a.status, static_cast<int64_t>(prev) - 1 - lane0, etag);   // look-back of the group before: words requested now, read behind the tiles
    uint32_t nrows_w = 0;                                           // wave-uniform
    // A wave-tile in two stages.  filter(j): pair lookups, candidate and synchronising bits, ownership, the owned candidates listed
    // in LDS.  verify(j): the candidates against the literals, FindAll order, rows.  The 16 bytes at each of the first 64 candidates
    // are REQUESTED (global memory: an L2 round trip) in front of filter(j + 1) and compared behind it.
    struct TileCtx { int32_t soff; int32_t prevb; int32_t rend; uint32_t ncand; };   // soff: the tile's first byte in the unit's descriptor
    auto filter = [&](int j, uint32_t cb_, TileCtx& cx) {
      lane = lane0;
      asm volatile("" : "+v"(lane));                                // (scan_chain_wave.hip: no hoisted-and-spilled lane constants)
      // the loads of the next tile (the unit behind this one may lie anywhere in the haystack — a stolen claim —, also when this tile lies behind its end)
      auto issue_next = [&]() { if (j + 1 < cur.tiles) issue_loads(cur, j + 1); else { nxt = make_unit(next_group); issue_loads(nxt, 0); } };
      const int32_t rend = cur.rem - j * kWaveTile;
      const bool first_tile = j == 0 && cur.pre == 0;               // the haystack's first tile (or a unit behind its end)
      cx.ncand = 0; cx.rend = rend; cx.soff = cur.pre + j * kWaveTile; cx.prevb = -1;
      if (rend <= 0) { issue_next(); return; }
      const int32_t stage = rend < kPWin ? rend : kPWin;
      const uint32_t two = 2u;
      // ---- A: one lookup per byte pair, the entries of a piece's eight pairs in two registers, transposed through LDS
      uint32_t ia[32];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32x4 v = x[k];
        CXG_PAIR_ADDR(ia[8 * k + 0], v.x, 1, 0); CXG_PAIR_ADDR(ia[8 * k + 1], v.x, 3, 1); CXG_PAIR_ADDR(ia[8 * k + 2], v.y, 1, 0); CXG_PAIR_ADDR(ia[8 * k + 3], v.y, 3, 1);
        CXG_PAIR_ADDR(ia[8 * k + 4], v.z, 1, 0); CXG_PAIR_ADDR(ia[8 * k + 5], v.z, 3, 1); CXG_PAIR_ADDR(ia[8 * k + 6], v.w, 1, 0); CXG_PAIR_ADDR(ia[8 * k + 7], v.w, 3, 1);
      }
      cx.prevb = first_tile ? -1 : static_cast<int32_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(xprev))) >> 24);
      issue_next();                                                // x[] is free from here on
      // (ds_read_u8_d16 / _d16_hi would put two entries into the halves of one register without VALU packing — measured: with SRAM ECC
      // on, a D16 load clears the other half of its register on this device, the kernel found nothing.)
      uint32_t ea[32];
#pragma unroll
      for (int q = 0; q < 32; q++) ea[q] = (CXG_PAIR_ABL & 4) ? (ia[q] & 0x3Fu) : S.img.tab[ia[q]];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint2 o;
        // three instructions per dword: v_perm_b32 takes the low bytes of two registers, v_lshl_or joins the halves
        o.x = __builtin_amdgcn_perm(ea[8 * k + 1], ea[8 * k], 0x0C0C0400u) | (__builtin_amdgcn_perm(ea[8 * k + 3], ea[8 * k + 2], 0x0C0C0400u) << 16);
        o.y = __builtin_amdgcn_perm(ea[8 * k + 5], ea[8 * k + 4], 0x0C0C0400u) | (__builtin_amdgcn_perm(ea[8 * k + 7], ea[8 * k + 6], 0x0C0C0400u) << 16);
        *reinterpret_cast<uint2*>(&L.w[2 * (lane + 64 * k)]) = o;
      }
      wave_lds_sync();
      uint32_t W[9];
      {
        const u32x4 wa = *reinterpret_cast<const u32x4*>(&L.w[8 * lane]);
        const u32x4 wb = *reinterpret_cast<const u32x4*>(&L.w[8 * lane + 4]);
        W[0] = wa.x; W[1] = wa.y; W[2] = wa.z; W[3] = wa.w; W[4] = wb.x; W[5] = wb.y; W[6] = wb.z; W[7] = wb.w;
        W[8] = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(W[0]), 0x130 /*wave_shl:1*/, 0xF, 0xF, true));   // lane 63: nothing behind the window
      }
      uint32_t cd[8], zd[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint32_t a1 = __builtin_amdgcn_alignbit(W[q + 1], W[q], 10);   // entry of the next pair >> 2
        const uint32_t a2 = __builtin_amdgcn_alignbit(W[q + 1], W[q], 20);   // entry of the pair behind it >> 4
        cd[q] = __builtin_amdgcn_udot4(W[q] & a1 & a2 & 0x03030303u, 0x40100401u, 0u, false);   // eight bits: the dword's eight byte positions
        zd[q] = __builtin_amdgcn_udot4(W[q] & 0xC0C0C0C0u, 0x40100401u, 0u, false);             // the same for S1 / S2, << 6
      }
      const uint64_t C = (static_cast<uint64_t>(cd[4] | (cd[5] << 8) | (cd[6] << 16) | (cd[7] << 24)) << 32) | (cd[0] | (cd[1] << 8) | (cd[2] << 16) | (cd[3] << 24));
      uint64_t Z = (static_cast<uint64_t>(((zd[4] | (zd[5] << 8)) >> 6) | (((zd[6] | (zd[7] << 8)) >> 6) << 16)) << 32) |
                   (((zd[0] | (zd[1] << 8)) >> 6) | (((zd[2] | (zd[3] << 8)) >> 6) << 16));
      if (stage != kPWin) {                                        // short last window: bytes past the data read as 0
        const int32_t nv = stage - 64 * lane;
        Z &= nv <= 0 ? 0ull : (nv >= 64 ? ~0ull : ((1ull << nv) - 1ull));
      }
      // ---- O: ownership bounds (scan_teddy_wave.hip)
      int32_t zA = -1, zB = kPFar;
      if (!first_tile) {
        if (!(S.img.FB[cx.prevb] & 0x1000000u)) {                          // the segment at the tile's first byte began earlier
          const unsigned long long bz = __ballot(Z != 0ull);
          if (bz) {
            const int Lz = __builtin_ctzll(bz);
            zA = 64 * Lz + static_cast<int32_t>(__builtin_ctzll(readlane64(Z, Lz)));
            if (Lz >= 16) fallback |= 1;                            // (the same budget as scan_teddy_wave.hip: a synchronising byte in the first KiB)
          }
          else zA = kPFar;
        }
      }
      {
        // (bits 3839 .. 4095 of the window: the top bit of lane 59's word, all of the lanes behind it)
        const uint64_t Zb = lane > 59 ? Z : (lane == 59 ? Z & (1ull << 63) : 0ull);
        const unsigned long long bzb = __ballot(Zb != 0ull);
        if (bzb) { const int Lz = __builtin_ctzll(bzb); zB = 64 * Lz + static_cast<int32_t>(__builtin_ctzll(readlane64(Zb, Lz))); }
        else if (stage != rend) { zB = -2; fallback |= 1; }
      }
      const uint64_t Co = C & word_range(lane, zA + 1, zB);
      // ---- list the owned candidates
      const uint32_t nc_lane = static_cast<uint32_t>(__popcll(Co));
      const uint32_t incl = wave_inclusive_sum(nc_lane);
      uint32_t ncand = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
      if (ncand > static_cast<uint32_t>(kPCands)) { fallback |= 8; ncand = kPCands; }
      if (CXG_PAIR_ABL & 2) ncand = 0;
      cx.ncand = (CXG_PAIR_ABL & 1) ? 0u : ncand;
      if (ncand) {
        uint32_t idx = incl - nc_lane;
        uint64_t cb = Co;
        while (__ballot(cb != 0ull) != 0ull) {                      // a uniform loop: a lane without a candidate left writes the dump slot
          const bool has = cb != 0ull && idx < static_cast<uint32_t>(kPCands);
          const int bit = __builtin_ctzll(cb | (1ull << 63));
          L.cpos[cb_][has ? idx : static_cast<uint32_t>(kPCands)] = static_cast<uint16_t>(64 * lane + bit);
          idx += cb != 0ull ? 1u : 0u;
          cb &= cb - 1;
        }
        wave_lds_sync();
      }
    };
    // the 16 bytes at the dword in front of candidate r of the list (requested; compared in verify)
    auto request = [&](uint32_t cb_, const TileCtx& cx, uint32_t r, int32_t& c, u32x4& d) {
      c = L.cpos[cb_][r < static_cast<uint32_t>(kPCands) ? r : static_cast<uint32_t>(kPCands)];   // (a lane behind the list reads a stale entry: its load stays inside the descriptor, verify() ignores it)
      if (CXG_PAIR_ABL & 64) d = u32x4{0u, 0u, 0u, 0u}; else d = __builtin_amdgcn_raw_buffer_load_b128(cur.rsrc, (c & ~3) + cx.soff, 0, CXG_PAIR_VAUX);
    };
    auto verify = [&](int j, uint32_t cb_, const TileCtx& cx, int32_t c0, u32x4 d0) {
      uint32_t emitted_here = 0;
      const uint32_t ncand = cx.ncand;
      const int32_t rend = cx.rend;
      if (ncand) {
        auto wbyte = [&](int32_t i) -> uint32_t { return __builtin_amdgcn_raw_buffer_load_b8(cur.rsrc, i + cx.soff, 0, 0); };   // window byte i (0 past the data)
        int32_t cur_end = -1;                                       // wave-uniform: end of the last emitted match
        for (uint32_t r0 = 0; r0 < ncand; r0 += 64) {
          int32_t c = c0, mlen = 0;
          u32x4 d = d0;
          if (r0) request(cb_, cx, r0 + static_cast<uint32_t>(lane), c, d);
          {
            // the 12 bytes at the candidate as three dwords (v_alignbit over the 16 loaded)
            const bool have = r0 + static_cast<uint32_t>(lane) < ncand;
            const uint32_t sh = (static_cast<uint32_t>(c) & 3u) * 8u;
            const uint32_t w0 = __builtin_amdgcn_alignbit(d.y, d.x, sh), w1 = __builtin_amdgcn_alignbit(d.z, d.y, sh), w2 = __builtin_amdgcn_alignbit(d.w, d.z, sh);
            // The slots of the literals that begin with the candidate's first byte, without a divergent branch: maxrun (uniform) steps, a lane
            // past its range compares slot 0 in vain.  (The reference verifies bucket by bucket, ids ascending — verifyBucket; the set is
            // prefix-free: at most one literal matches at a position, the order cannot be observed.)
            const uint32_t fb = S.img.FB[w0 & 0xFFu];
            const uint32_t kbeg = fb & 0xFFu, kend = have ? (fb >> 8) & 0xFFu : 0u;
            bool tail = false;                                       // a literal longer than 12 bytes agreed on its first 12
            for (uint32_t i = 0; i < maxrun; i++) {
              const uint32_t k = kbeg + i;
              const bool valid = k < kend;
              const uint32_t* lx = S.img.litx[valid ? k : 0u];
              const u32x4 la = *reinterpret_cast<const u32x4*>(lx), lb4 = *reinterpret_cast<const u32x4*>(lx + 4);
              const uint32_t diff = ((w0 ^ la.x) & la.w) | ((w1 ^ la.y) & lb4.x) | ((w2 ^ la.z) & lb4.y);
              const int32_t len = static_cast<int32_t>(lb4.z);
              const bool hit = valid && diff == 0u && c + len <= rend;
              if (short_lits) mlen = hit ? len : mlen;
              else { mlen = (hit && len <= 12) ? len : mlen; tail = tail || (hit && len > 12); }
            }
            if (!short_lits && __ballot(tail && mlen == 0) != 0ull) {   // the bytes behind the twelfth, of every literal of the range that agreed so far
              if (tail && !mlen)
                for (uint32_t k = kbeg; k < kend && !mlen; k++) {
                  const uint32_t* lx = S.img.litx[k];
                  const int32_t len = static_cast<int32_t>(lx[6]);
                  if (len <= 12 || c + len > rend) continue;
                  if ((((w0 ^ lx[0]) & lx[3]) | ((w1 ^ lx[1]) & lx[4]) | ((w2 ^ lx[2]) & lx[5])) != 0u) continue;
                  const uint8_t* lit = t_bytes + t_off[lx[7]];
                  int32_t q = 12;
                  while (q < len && same(wbyte(c + q), lit[q])) q++;
                  if (q == len) mlen = len;
                }
            }
            if ((look_pre | look_post) != 0u && mlen) {             // the assertions around the occurrence (checkLook, nfa/pikevm.go:1646-1674)
              const int pbv = c > 0 ? static_cast<int>(wbyte(c - 1)) : cx.prevb;
              const int nb = c + mlen < rend ? (c + mlen < kPWin ? static_cast<int>(wbyte(c + mlen)) : -2) : -1;
              if (nb == -2) { edge_hit = 1; mlen = 0; }               // the byte behind the occurrence lies behind the window: hand the scan over
              else if (!teddy_look_holds(look_pre, pbv, static_cast<int>(w0 & 0xFFu)) || !teddy_look_holds(look_post, static_cast<int>(wbyte(c + mlen - 1)), nb)) mlen = 0;
            }
          }
          // ---- D: FindAll order inside the round (candidates ascend with the lane)
          const int32_t e = mlen ? c + mlen : 0;
          const int32_t pmax = static_cast<int32_t>(wave_inclusive_max(static_cast<uint32_t>(e)));   // inclusive prefix max of the ends
          int32_t before = static_cast<int32_t>(dpp_from_lower(static_cast<uint32_t>(pmax)));
          if (lane == 0) before = 0;
          if (cur_end > before) before = cur_end;
          uint32_t emit = mlen ? 1u : 0u;
          if (__ballot(mlen && c < before) != 0ull) {               // some verified candidate lies inside an earlier match
            L.ce[lane] = static_cast<uint16_t>(e);
            wave_lds_sync();
            if (lane == 0) {
              int32_t ce = cur_end;
              for (uint32_t k = 0; k < 64; k++) {                   // all 64: lanes past ncand hold e = 0 and must read em = 0
                const int32_t ek = L.ce[k];
                uint8_t em = 0;
                if (ek && static_cast<int32_t>(L.cpos[cb_][(r0 + k) < static_cast<uint32_t>(kPCands) ? r0 + k : 0u]) >= ce) { em = 1; ce = ek; }
                L.em[k] = em;
              }
            }
            wave_lds_sync();
            emit = L.em[lane];
          }
          const unsigned long long em_mask = __ballot(emit != 0);
          if (em_mask) {
            const int last = 63 - __builtin_clzll(em_mask);
            cur_end = __builtin_amdgcn_readlane(e, last);
            const uint32_t n_em = static_cast<uint32_t>(__popcll(em_mask));
            {
              const uint32_t r = nrows_w + emitted_here + static_cast<uint32_t>(__popcll(em_mask & ((1ull << lane) - 1ull)));
              const uint32_t slot = (emit && r < static_cast<uint32_t>(kPRows)) ? r : static_cast<uint32_t>(kPRows);   // (dump slot)
              L.rs[b][slot] = static_cast<uint16_t>(j * kWaveTile + c); L.re[b][slot] = static_cast<uint16_t>(j * kWaveTile + e);
            }
            emitted_here += n_em;
          }
        }
      }
      nrows_w += emitted_here;
    };
    if (live) {
      TileCtx cxa, cxb;                                             // even / odd tiles (the loop body is two tiles: no indexed registers)
      filter(0, 0u, cxa);
      for (int j = 0; j < cur.tiles; j += 2) {
        int32_t c0; u32x4 d0;
        request(0u, cxa, static_cast<uint32_t>(lane), c0, d0);
        filter(j + 1, 1u, cxb);
        verify(j, 0u, cxa, c0, d0);
        request(1u, cxb, static_cast<uint32_t>(lane), c0, d0);
        if (j + 2 < cur.tiles) filter(j + 2, 0u, cxa);
        verify(j + 1, 1u, cxb, c0, d0);
      }
    }
    if (nrows_w > static_cast<uint32_t>(kPRows)) fallback |= 16;

    // ---- the group before: its base (wave 0), then — behind the barrier — its rows; this group: its count
    if (live && lane0 == 0) S.wcnt[b][wave] = nrows_w;
    if (wave == 0 && prev != ~0ull && !(CXG_PAIR_ABL & 8)) {
      const uint64_t base = prev > 0 ? pair_resolve(a.status, a.err, prev, lw, etag, lane0) : 0ull;
      if (lane0 == 0) {
        const uint64_t incl = base + S.tot[b ^ 1u];
        if (prev > 0) __hip_atomic_store(a.status + prev, kFlagInclusive | etag | incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        S.base[b ^ 1u] = base;
        if (prev == ngroups - 1) *a.total = incl;
      }
    }
    if (wave == kPWaves - 1 && live && lane0 == 0) S.gq[(it + 2u) % 3u] = claimed(n2);
    __syncthreads();
    if (live && wave == 0) {                                        // exclusive sums over the group's 16 units; the group's count is published
      const uint32_t v = lane0 < kPWaves ? S.wcnt[b][lane0] : 0u;
      const uint32_t incl = wave_inclusive_sum(v);
      if (lane0 < kPWaves) S.woff[b][lane0] = incl - v;
      if (lane0 == 63) {
        S.tot[b] = incl;
        __hip_atomic_store(a.status + group, (group == 0 ? kFlagInclusive : kFlagAggregate) | etag | static_cast<uint64_t>(incl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (prev != ~0ull && a.out != nullptr && !(CXG_PAIR_ABL & 8)) {   // rows of the group before (buffers b ^ 1): every wave its own unit's
      const uint32_t pbuf = b ^ 1u;
      const uint64_t base = S.base[pbuf] + S.woff[pbuf][wave];
      const int64_t origin = a.base + static_cast<int64_t>(prev_lo);
      const uint32_t nr = S.wcnt[pbuf][wave];
      const uint32_t n = nr < static_cast<uint32_t>(kPRows) ? nr : static_cast<uint32_t>(kPRows);
      for (uint32_t i = lane0; i < n; i += 64)
        if (base + i < a.cap) store_pair_nt(a.out + (base + i) * a.row_width, origin + L.rs[pbuf][i], origin + L.re[pbuf][i]);
    }
    if (!live) break;
    prev = group;
    prev_lo = cur.lo;
  }
  if (__ballot(edge_hit != 0) != 0ull) fallback |= 32;
  if (fallback != 0 && lane0 == 0) raise_err(a.err, 8u | (fallback << 8));
}

hipError_t launch_scan_teddy_pair(const ScanArgs& a, uint32_t workgroups, hipStream_t stream) {
  const dim3 grid(workgroups), block(kPThreads);
  hipLaunchKernelGGL(k_scan_teddy_pair, grid, block, 0, stream, a);
  return hipGetLastError();
}

}  // namespace cxgdev
