// Launch interface of the DFA scan kernels (scan_dfa.hip); shared with capi_internal.hpp and tests/emu.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace cxgdev {

constexpr int kThreads = 256;                 // 4 waves of 64
constexpr int kChunk = 64;                    // bytes owned per lane
constexpr int kTile = kThreads * kChunk;      // 16 KiB per workgroup
constexpr int kHaloChunks = 4;
constexpr int kHalo = kHaloChunks * kChunk;   // 256 B staged past the tile for lanes that overrun
constexpr int kGroupTiles = 8;                // tiles per workgroup in the grouped kernels (one ticket / look-back per 128 KiB)
// wave-tile geometry of scan_digit_wave.hip: one wave64 per 3840 B (+256 B halo = 64 bitmap words)
constexpr int kWaveTile = 3840;
constexpr int kWaveHalo = 256;
constexpr int kWavesPerBlock = 4;
#ifndef CXG_TPW
#define CXG_TPW 8                         // -DCXG_TPW=n: experiments with the group size (scripts/build_variant.sh), all sources
#endif
constexpr int kTilesPerWave = CXG_TPW;
constexpr int kDenseTilesPerWave = 2;          // chain kernel on match-dense input: 256 rows of buffer per wave-tile instead of 64
constexpr uint64_t kWaveGroupBytes = static_cast<uint64_t>(kWaveTile) * kWavesPerBlock * kTilesPerWave;   // 120 KiB per workgroup
// scan_teddy_pair.hip: one workgroup of 16 waves per CU (its pair table takes 64 KiB of LDS), groups of 16 x 8 wave-tiles = 480 KiB
constexpr int kPairWaves = 16;
constexpr int kPairTilesPerWave = 8;
constexpr uint32_t kPairCtrStride = 32;       // uint32 between two group counters (128 bytes)
constexpr uint64_t kPairGroupBytes = static_cast<uint64_t>(kWaveTile) * kPairWaves * kPairTilesPerWave;
#ifndef CXG_PAIR_SMALL_TPW
#define CXG_PAIR_SMALL_TPW 2               // (A/B builds: 4)
#endif
constexpr int kPairSmallTilesPerWave = CXG_PAIR_SMALL_TPW;     // ... and SMALL groups of 16 x 2 wave-tiles = 120 KiB behind them: the last stretch of a haystack (one big group per CU), so that the
constexpr uint64_t kPairSmallGroupBytes = static_cast<uint64_t>(kWaveTile) * kPairWaves * kPairSmallTilesPerWave;   // launch does not end with half of the CUs idle for a big group's time
#ifndef CXG_CC_TILES
#define CXG_CC_TILES 4
#endif
constexpr int kCcTilesPerWave = CXG_CC_TILES;             // scan_charclass_wave.hip: 4 waves x 4 wave-tiles = 60 KiB per workgroup
constexpr uint64_t kCcGroupBytes = static_cast<uint64_t>(kWaveTile) * kWavesPerBlock * kCcTilesPerWave;
constexpr int kRecCap = 1024;                 // LDS match records per tile before the direct-write path
// Serial-walk budget.  A lane that owns a stretch without synchronising bytes walks it alone at ~1.6 us per byte
// (dependent byte loads from L2/HBM; scripts/time_nosync.py): a mebibyte takes seconds.  Every walk is therefore cut
// kSerialLimit bytes behind its staged window — the cut plays end of input — and a lane that reads the last byte
// before the cut raises error bit 32: the host returns CXG_E_INPUT and the caller keeps its CPU loop for this
// haystack.  (The cut also keeps lane-relative offsets inside int32 for haystacks beyond 2 GiB.)
constexpr int32_t kSerialLimit = 128 * 1024;
constexpr uint32_t kSerialReads = 1u << 20;   // ... and at most this many one-byte reads beyond the staged window per lane and launch (scan_dfa.hip LdsMem::byte).  (Round 5 tried 3 << 17 to end a quadratic program's call sooner; the ownership searches of input without synchronising bytes legitimately read more than that per lane.)
constexpr uint32_t kErrSerialLimit = 32u;
constexpr uint32_t kErrLongMatch = 64u;
struct WalkLimit { int32_t rend; int32_t flag_at; };
#if defined(__HIPCC__)
__host__ __device__
#endif
inline WalkLimit walk_limit(uint64_t remaining, int32_t window) {
  const uint64_t cut = static_cast<uint64_t>(window) + static_cast<uint64_t>(kSerialLimit);
  if (remaining > cut) return WalkLimit{static_cast<int32_t>(cut), static_cast<int32_t>(cut) - 1};
  return WalkLimit{static_cast<int32_t>(remaining), 0x7FFFFFFF};
}

// k_scan_fields_pers (scan_fields_wave.hip): pf_rec holds 384 words of 8 bytes per round — 128 records of two words, 128 block sums
constexpr int kPfBlocks = 128;
constexpr uint32_t kPfCtrStride = 32;           // uint32 between two ticket counters (128 bytes: one cache line each); ring of 32 launch epochs x 64 counters
constexpr uint64_t kPfRecStride = 3ull * kPfBlocks + 8ull;   // (+ the round's inclusive row count, round 6)

// A union of <= 4 ASCII ranges prepared for the SWAR class tests (wave_common.hpp "class plans"): nx X terms (xc, xk), nr R terms
// (ra, rb; the first one over the case-folded bytes when fold).  ok == 0: a bound >= 0x80 — the kernel keeps notset4 / its table.
struct ClassPlan { uint32_t nx, nr, fold, ok; uint32_t xc[4], xk[4], ra[4], rb[4]; };

struct ScanArgs {
  const uint8_t* hay;   // device, 16-byte aligned
  uint64_t len;
  int64_t base;         // added to every reported offset (shard origin)
  const uint8_t* blob;  // device copy of the program image (walk.hpp BlobHeader)
  int64_t* out;         // [cap][2] or nullptr to count only
  uint64_t cap;
  uint64_t* status;     // [ntiles] look-back words, zeroed before launch
  uint64_t* status2;    // scan_fsm.hip: [ngroups] exit state of every group (epoch << 32 | valid | state), same life cycle as status
  uint32_t* ticket;     // zeroed before launch
  uint64_t* total;      // match count (written by the last tile); wave kernels: pinned host memory, read without a copy
  uint32_t* err;        // bit0 lane overflow, bit1 look-back watchdog, bit2 capture table, bit3 fallback; host memory likewise
  uint64_t ntiles;
  uint64_t ngroups;     // look-back units: == ntiles, except for kernels that process kGroupTiles tiles per workgroup
  uint64_t* prof;       // optional [8] phase cycle counters (CXG_PROF=1), else nullptr
  uint32_t row_width;   // int64 per output row: 2, or 2*groups when a capture pass follows
  uint8_t chain[224];   // scan_chain_wave.hip: copy of the program's ChainAux (walk.hpp) — kernel arguments are read with
                        // scalar loads before the first instruction needs them, the blob would cost two dependent global loads per workgroup
  uint32_t epoch;          // wave kernels: launch epoch 1..1023 tagging the status words (0: array was zeroed, legacy)
  uint32_t static_groups;  // wave kernels: group = blockIdx.x instead of an atomic ticket (block_common.hpp claim_group)
  uint8_t caps[40];     // scan_chain_wave.hip CAP instantiations: the program's ChainCaps (walk.hpp); caps[0] == 0: spans only
  uint32_t tiles_per_wave;  // scan_chain_wave.hip: kTilesPerWave, or kDenseTilesPerWave after a row-buffer overflow
  uint32_t max_len;     // != 0: a match longer than this raises error bit 64 (UseBoth programs, walk.hpp kFlagBothRestart)
  uint32_t dbg;         // CXG_DEBUG bit0: skip the lane walk, bit1: skip the look-back (timing experiments only)
  uint32_t count_sum;   // scan_fields_wave.hip, count-only call: groups leave their totals in status[], k_sum_counts adds them up (no look-back)
  uint64_t* fsm_maps;   // scan_fsm.hip: three epoch-tagged words per group (fsm_group_entry)
  uint64_t limit;       // FindAll's n when > 0, else 0: rows beyond it are not wanted (block_common.hpp tile_lookback: early stop)
  uint32_t* stop;       // device word: == epoch + 1 once `limit` rows have been counted
  // scan_fields_wave.hip k_scan_fields_pers (persistent grid, ordering deferred by a round); pf_status == nullptr: grouped kernel
  uint32_t* pf_status;  // [pf_cap] one word per unit (a wave's 8 wave-tiles of a round): pf_epoch << 16 | rows of the unit
  uint64_t pf_cap;
  uint32_t pf_epoch;    // 1..65535, own counter (the words are 4 bytes: block_common.hpp's 10-bit epoch words do not fit)
  uint32_t pf_full, pf_tpw_last, pf_units_last;   // filled in by the launcher — round 6 (tickets): units of kPfTiles tiles, of three tiles, of one tile (round 5: full rounds, tiles per unit / units of the tapered last round)
  uint64_t* pf_rec;     // [pf_rec_rounds][384] per round: 128 records {rows of the round in front of the block, rows of the round} + 128 block sums, each word tagged pf_epoch << 48
  uint64_t pf_rec_rounds;
  uint64_t* pf_stats;   // [8192] per wave: units that waited << 32 | polls (CXG_VERBOSE)
  uint32_t* pf_ticket;  // [32][64] ticket counters, kPfCtrStride words apart; block pf_epoch & 31 is this launch's (round 6: units are claimed, not assigned)
  uint32_t pf_ncounters;   // counters in use: a power of two <= min(64, workgroups)
  ClassPlan plan;       // scan_charclass_wave.hip: the class; k_scan_trio_wave: the field class (filled on the host per launch)
  uint32_t plan_shape;  // wave_common.hpp plan_shape(plan); 0: the generic range tests
  uint32_t cc_nr, cc_neg, cc_pairs;   // scan_charclass_wave.hip: walk.hpp CharClassAux copied by the host (kernel arguments: no dependent
  uint8_t cc_lo[4], cc_hi[4];         // loads from the program image before the first window can be requested)
  uint32_t* pair_ctr;   // scan_teddy_pair.hip: [2][8] group counters, kPairCtrStride words apart; set pair_seq & 1 is this launch's, the kernel zeroes the other
  uint32_t pair_nbig;   // groups [0, pair_nbig) are big (kPairGroupBytes: 8 tiles per wave), the next pair_n6 have 6 tiles per wave, the next pair_n4 have 4,
  uint32_t pair_n6, pair_n4;   // the rest 2 (kPairSmallGroupBytes), laid out one behind the other: the groups of a launch taper off, so that its CUs end together
  uint32_t pair_seq, pair_nctr;   // pair_nctr: 8 (workgroup b claims from counter b & 7 first), or 1: strict ticket order (after a watchdog hit)
  uint32_t u32_rows;    // cxg_find_all_device_u32: `out` holds rows of two uint32 relative to `hay` (kernels with the compact epilogue only)
};

}  // namespace cxgdev
