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
#include "scan_fsm_common.hpp"

namespace cxgdev {

namespace {

// All LDS of the kernel is ONE struct, the image first: the transition table then sits at LDS address 0 and a walk
// step's address  (entry & ~3) | 2 * class  goes straight into the ds_read — no base add on the dependent chain
// (with the table anywhere else the compiler adds the base in a VALU op: it cannot prove that base + offset does not
// wrap, so it does not use the instruction's immediate offset).  IMG = bytes reserved for the image.
// MODE: buffer geometry by match density, 0..3 (scan_fsm_common.hpp FsmMode).
template <bool SHALLOW, int IMG, int MODE, bool LOOKTAB>
struct FsmLds {
  uint8_t img[IMG];
  uint8_t win[kWavesPerBlock][kFsmWinBytes];
  uint16_t lk16[LOOKTAB ? 256 : 2];                            // look-around: class | kind << 8 of a byte (fsm.hpp FsmView::lk16)
  uint16_t lrow[kWavesPerBlock][SHALLOW ? 4 : 64 * 2 * FsmMode<MODE>::kRows];   // per-lane row ends of the current tile (two sub-chunks); shallow machines: rows come from the event bits
  uint16_t lev[kWavesPerBlock][SHALLOW ? 4 : 64 * 2 * FsmMode<MODE>::kEvents];   // per-lane recorded events (alias rows); machines with depth > 1 only
  uint16_t re[kWavesPerBlock][FsmMode<MODE>::kRowsPerWave + 8];      // rows of the group: end inside its wave-tile (+ a dump slot for the branch-free row loop)
  uint16_t rl[kWavesPerBlock][FsmMode<MODE>::kRowsPerWave];          // ... and length (0: unresolved)
  uint32_t cnt[kWavesPerBlock][kTilesPerWave];
  union {                                                      // (the 10 KiB-image instantiations sit 320 bytes below 4 workgroups per CU)
    struct {                                                   // after the tile loop
      uint32_t qbase[kWavesPerBlock * kTilesPerWave + 4];
      int64_t tail[kWavesPerBlock * kTilesPerWave];            // absolute end of a tile's last row, -1: no rows
    };
    uint16_t mapx[kWavesPerBlock * kTilesPerWave][kFsmMembers];   // inside it — deferred tiles: exit for the j-th member of the tile's first set (fsm_resolve_exits)
  };
  uint64_t group;
  uint64_t base;
  uint32_t exit[kWavesPerBlock * kTilesPerWave];               // exit state of every tile of the group | 0x80000000 once known
 uint16_t mapu[kWavesPerBlock * kTilesPerWave];               // ... and that set (uncertainty row)
  uint32_t ndefer;                                             // some wave left tiles to the second pass
};

// Exit state of a tile, handed to the next tile (needed only when that tile's first set of possible states does not
// collapse): inside the workgroup through LDS, to the next group through one epoch-tagged word in HBM (relaxed
// agent-scope accesses, self-contained word: block_common.hpp).
constexpr uint32_t kExitValid = 0x80000000u;
__device__ __forceinline__ void publish_tile_exit(uint32_t* s_exit, uint64_t* status2, uint64_t group, int q, int ntiles_group, uint32_t epoch, uint32_t ex) {
  __hip_atomic_store(s_exit + q, ex | kExitValid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (q == ntiles_group - 1)
    __hip_atomic_store(status2 + group, (static_cast<uint64_t>(epoch) << 32) | kExitValid | ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Wave-uniform: the exit state of the tile in front of tile q of this group.
__device__ __forceinline__ uint32_t wait_tile_exit(uint32_t* s_exit, uint64_t* status2, uint32_t* err, uint64_t group, int q, uint32_t epoch, int lane) {
  uint32_t w = 0;
  for (uint32_t spins = 0;; spins++) {
    if (q > 0) w = __hip_atomic_load(s_exit + (q - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else {
      const uint64_t g = __hip_atomic_load(status2 + (group - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      w = (static_cast<uint32_t>(g >> 32) == epoch) ? static_cast<uint32_t>(g) : 0u;
    }
    w = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w)));
    if (w & kExitValid) break;
    if (spins > (kSpinLimit << 6)) { if (lane == 0) raise_watchdog(err, kWdFsmExit); break; }   // (a hand-off chain over every tile of the input is legitimate)
    // Input without synchronising structure makes this a serial chain over every tile (17 476 hops for 64 MiB): the poll
    // interval IS the hop latency.  s_sleep 4 (256 cycles) gave 323 ns per tile = 5.65 ms; the LDS hop inside a workgroup
    // is polled back to back, the HBM hop between workgroups with the shortest sleep.
    if (q == 0) __builtin_amdgcn_s_sleep(1);
  }
  return w & 0xFFFFu;
}
// Wave-uniform: the same word without insisting — kExitValid clear after `tries` looks.
__device__ __forceinline__ uint32_t peek_tile_exit(uint32_t* s_exit, uint64_t* status2, uint64_t group, int q, uint32_t epoch, uint32_t tries) {
  uint32_t w = 0;
  if (q == 0 && group == 0) return kExitValid;
  for (uint32_t spins = 0; spins < tries; spins++) {
    if (q > 0) w = __hip_atomic_load(s_exit + (q - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else {
      const uint64_t g = __hip_atomic_load(status2 + (group - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      w = (static_cast<uint32_t>(g >> 32) == epoch) ? static_cast<uint32_t>(g) : 0u;
    }
    w = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w)));
    if (w & kExitValid) break;
  }
  return w;
}

// The exit of the group in front of `group` when that group's own exit depends on ITS entry, and so on: a decoupled look-back
// over MAPS instead of sums.  Every group whose first tile left a map publishes the group's map (first set u, exit for each
// of its <= 8 members: three epoch-tagged words behind the exit words in status2); a group with a known exit publishes the
// exit VALUE (status2[group]).  Lane l reads group look - l; the maps in front are composed nearest first —
// C := C o M, starting from the identity on this group's candidates — until a VALUE ends the walk: 64 groups per round
// trip, and no group waits for the one in front to finish (a hop from workgroup to workgroup costs ~5 us across XCDs:
// 546 of them for 64 MiB were 2.8 ms).  keys / vals: lanes 0..7, this group's candidate entries and its exits for them.
__device__ __forceinline__ uint32_t fsm_group_entry(const FsmView& v, const ScanArgs& a, uint64_t group, uint32_t u_own, uint32_t keys, uint32_t vals,
                                                    bool publish, int lane) {
  constexpr uint64_t kF = 0xFFFFull;
  const uint64_t tag = static_cast<uint64_t>(a.epoch + 1u) << 48;
  uint64_t* const maps = a.fsm_maps;
  uint32_t sv[kFsmMembers], ck[kFsmMembers], cv[kFsmMembers];
#pragma unroll
  for (int i = 0; i < kFsmMembers; i++) {
    sv[i] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(vals), i));
    ck[i] = cv[i] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(keys), i));   // identity on the candidates
  }
  if (publish && lane < 3) {
    const uint64_t w = lane == 0 ? (u_own | (static_cast<uint64_t>(sv[0]) << 16) | (static_cast<uint64_t>(sv[1]) << 32))
                     : lane == 1 ? (sv[2] | (static_cast<uint64_t>(sv[3]) << 16) | (static_cast<uint64_t>(sv[4]) << 32))
                                 : (sv[5] | (static_cast<uint64_t>(sv[6]) << 16) | (static_cast<uint64_t>(sv[7]) << 32));
    __hip_atomic_store(maps + 3 * group + lane, tag | w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // this lane's field of a broadcast map: member j = lane & 7 is field j + 1 of the 9 (u, v0..v7), three per word
  // (opaque lane id: with the plain one the compiler evaluates every readlane(res, i) below as scalar code for lane i — 64
  // compare-and-select chains of five SALU operations per step, 2 us; as eight lanes of VALU work a step is ~50 instructions)
  int lane_o = lane;
  asm volatile("" : "+v"(lane_o));
  const int fld = (lane_o & (kFsmMembers - 1)) + 1, fw = fld / 3, fs = 16 * (fld % 3);
  int64_t look = static_cast<int64_t>(group) - 1;
  uint32_t spins = 0;
  for (;;) {
    const int64_t idx = look - lane;
    uint64_t wv = (static_cast<uint64_t>(a.epoch) << 32) | kExitValid, m0 = 0, m1 = 0, m2 = 0;   // (in front of group 0: never reached, group 0 has a VALUE)
    if (idx >= 0) {
      wv = __hip_atomic_load(a.status2 + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      m0 = __hip_atomic_load(maps + 3 * idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      m1 = __hip_atomic_load(maps + 3 * idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      m2 = __hip_atomic_load(maps + 3 * idx + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool isval = static_cast<uint32_t>(wv >> 32) == a.epoch && (wv & kExitValid) != 0;
    const bool ismap = (m0 >> 48) == (tag >> 48) && (m1 >> 48) == (tag >> 48) && (m2 >> 48) == (tag >> 48);
    const unsigned long long notready = ~__ballot(isval || ismap);
    const int nready = notready ? __builtin_ctzll(notready) : 64;
    if (nready == 0) {
      if (++spins > kSpinLimit) { if (lane == 0) raise_watchdog(a.err, kWdFsmEntry); return 0u; }
      __builtin_amdgcn_s_sleep(2);
      continue;
    }
    const unsigned long long vm = __ballot(isval) & (nready == 64 ? ~0ull : ((1ull << nready) - 1ull));
    const int nsteps = vm ? __builtin_ctzll(vm) : nready;
    uint32_t kk[kFsmMembers];                                      // this lane's map: its candidate entries
    const uint32_t u_l = static_cast<uint32_t>(m0 & kF);
#pragma unroll
    for (int i = 0; i < kFsmMembers; i++) kk[i] = (ismap && !isval) ? fsm_member(v, u_l, static_cast<uint32_t>(i)) : 0xFFFFu;
    for (int l = 0; l < nsteps; l++) {                             // C := C o M_l  (uniform loop)
      const uint32_t a0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(m0)), l)), a1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(m0 >> 32)), l));
      const uint32_t b0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(m1)), l)), b1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(m1 >> 32)), l));
      const uint32_t c0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(m2)), l)), c1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(m2 >> 32)), l));
      const uint64_t w = fw == 0 ? ((static_cast<uint64_t>(a1) << 32) | a0) : fw == 1 ? ((static_cast<uint64_t>(b1) << 32) | b0) : ((static_cast<uint64_t>(c1) << 32) | c0);
      const uint32_t x = static_cast<uint32_t>((w >> fs) & kF);       // M_l's exit for its member lane & 7
      uint32_t res = 0xFFFFu;
#pragma unroll
      for (int i = 0; i < kFsmMembers; i++) res = (x == ck[i]) ? cv[i] : res;
      if (x == 0xFFFFu) res = 0xFFFFu;
#pragma unroll
      for (int i = 0; i < kFsmMembers; i++) {
        cv[i] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(res), i));
        ck[i] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(kk[i]), l));
      }
    }
    if (vm) {
      const uint32_t e = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(wv)), nsteps)) & 0xFFFFu;
      uint32_t r = 0xFFFFu;
#pragma unroll
      for (int i = 0; i < kFsmMembers; i++) r = (e == ck[i] && e != 0xFFFFu) ? cv[i] : r;
      if (r == 0xFFFFu && lane == 0) raise_err(a.err, 8u | (1u << 8));                 // the true state is not in the listed set: cannot happen
      return r;
    }
    look -= nsteps;
  }
}

// Input without synchronising structure (`1.1.1.1...` for the IPv4 pattern): a tile's first set of possible states does not
// collapse, its true entry is the exit of the tile in front, and so on back to the start of the input — a serial chain over
// every tile.  Round 2 walked that chain with a blocked wave per tile (323 ns per tile: the wave in front had to finish ITS
// tile and stage the next before the chain moved on).  Now a tile that would have to wait leaves its MAP (candidate entry ->
// exit, <= 8 pairs) in LDS and the wave goes on to its other tiles, which do the same; after the group's barrier one wave
// (a) composes the group's maps into one, (b) waits for the exit of the group in front, (c) publishes its own group's exit —
// ONE lookup behind the arrival, the only serial step between groups — and (d) fills in the exit of every tile; the deferred
// tiles then run as usual (second pass), their waits already answered.  NT = tiles per group.
template <int NT>
__device__ __forceinline__ void fsm_resolve_exits(const FsmView& v, uint32_t* s_exit, const uint16_t (*s_mapx)[kFsmMembers], const uint16_t* s_mapu, const ScanArgs& a, uint64_t group, int lane) {
  const uint64_t tiles_all = (a.len + kWaveTile - 1) / static_cast<uint64_t>(kWaveTile), first = group * static_cast<uint64_t>(NT);
  const int nlive = tiles_all - first >= static_cast<uint64_t>(NT) ? NT : static_cast<int>(tiles_all - first);
  const int l8 = lane & (kFsmMembers - 1);
  const bool open = (s_exit[0] & kExitValid) == 0u;             // the group's first tile left a map: its exit depends on the group in front
  auto pair_of = [&](int q) -> uint32_t {                       // lanes 0..7: member | exit << 16 of tile q's map
    return (lane < kFsmMembers ? fsm_member(v, s_mapu[q], static_cast<uint32_t>(l8)) : 0xFFFFu) | (static_cast<uint32_t>(s_mapx[q][l8]) << 16);
  };
  uint32_t cur = open ? (pair_of(0) & 0xFFFFu) : 0u;
  const uint32_t cur0 = cur;
#pragma unroll 1
  for (int q = 0; q < NT; q++) {
    const uint32_t ev = q < nlive ? s_exit[q] : 0u;
    if (q >= nlive) {}
    else if (ev & kExitValid) cur = ev & 0xFFFFu;               // a tile whose exit does not depend on its entry: the chain restarts
    else {
      const uint32_t p = pair_of(q);
      uint32_t nxt = 0xFFFFu;
#pragma unroll
      for (int jm = 0; jm < kFsmMembers; jm++) {
        const uint32_t pj = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(p), jm));
        if ((pj & 0xFFFFu) == cur) nxt = pj >> 16;
      }
      cur = cur == 0xFFFFu ? 0xFFFFu : nxt;
    }
  }
  uint32_t entry = 0u, gx;
  if (open) {
    if (group == 0) { if (lane == 0) raise_err(a.err, 8u | (1u << 8)); return; }      // cannot happen: the input's first state is known
    entry = (a.dbg & 8u) ? 0u : fsm_group_entry(v, a, group, s_mapu[0], cur0, cur, nlive == NT, lane);   // (CXG_DEBUG=8: timing experiment, wrong rows)
    const unsigned long long hit = __ballot(lane < kFsmMembers && cur0 == entry);
    gx = hit ? static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cur), __builtin_ctzll(hit))) : 0xFFFFu;
  } else gx = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cur), 0));
  if (gx == 0xFFFFu) { if (lane == 0) raise_err(a.err, 8u | (1u << 8)); }              // the true state is not in the listed set: cannot happen
  else if (lane == 0 && nlive == NT)
    __hip_atomic_store(a.status2 + group, (static_cast<uint64_t>(a.epoch) << 32) | kExitValid | gx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t c = entry;
#pragma unroll 1
  for (int q = 0; q < NT; q++) {
    const uint32_t ev = q < nlive ? s_exit[q] : 0u;
    if (q >= nlive) {}
    else if (ev & kExitValid) c = ev & 0xFFFFu;
    else {
      const uint32_t p = pair_of(q);
      const unsigned long long hit = __ballot(lane < kFsmMembers && (p & 0xFFFFu) == c && c != 0xFFFFu);
      c = hit ? static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(p >> 16), __builtin_ctzll(hit))) : 0xFFFFu;
      if (c == 0xFFFFu) { if (lane == 0) raise_err(a.err, 8u | (1u << 8)); c = 0u; }
      if (lane == 0) s_exit[q] = c | kExitValid;
    }
  }
}

}  // namespace

// SHALLOW: the machine never holds more than one pending match (FsmHeader::depth <= 1): rows from two event bitmaps
// per sub-chunk instead of a recorded event list (fsm.hpp).  MODE: buffer geometry by match density (FsmMode).
// LOOK: 1 = the image has assertions (nk > 1): a step's class also reads the next byte (fsm.hpp "Look-around"); 2 = ... and an
// end-of-text anchor: the step over the haystack's last byte takes the column of the kind no byte has (fsm.hpp "End of text").
template <bool SHALLOW, int IMG, int MODE, int LOOK>
__global__ __launch_bounds__(kThreads, ((IMG > 10240 || MODE == 3 || (MODE == 2 && !SHALLOW)) ? 2 : (MODE == 2 ? 3 : 4))) void k_scan_fsm(ScanArgs a) {
  __shared__ __attribute__((aligned(16))) FsmLds<SHALLOW, IMG, MODE, (LOOK != 0)> S;
  constexpr int kLaneRows = FsmMode<MODE>::kRows, kLaneEvents = FsmMode<MODE>::kEvents, kRowsPerWave = FsmMode<MODE>::kRowsPerWave;
  uint8_t* const s_img = S.img;
  auto& s_win = S.win; auto& s_lrow = S.lrow; auto& s_lev = S.lev; auto& s_re = S.re; auto& s_rl = S.rl;
  auto& s_cnt = S.cnt; auto& s_qbase = S.qbase; auto& s_tail = S.tail;
  uint64_t& s_group = S.group;
  uint64_t& s_base = S.base;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) { s_group = claim_group(a.static_groups != 0, a.ticket, a.ngroups); S.ndefer = 0u; }
  if (tid < kWavesPerBlock * kTilesPerWave) S.exit[tid] = 0u;
  const FsmHeader* h = reinterpret_cast<const FsmHeader*>(a.blob);
  {
    const uint4* src = reinterpret_cast<const uint4*>(a.blob + sizeof(FsmHeader));
    const uint32_t nvec = h->lds_bytes >> 4;
    for (uint32_t i = tid; i < nvec; i += kThreads) reinterpret_cast<uint4*>(s_img)[i] = src[i];
  }
  __syncthreads();
  const uint64_t group = s_group;
  if (group >= a.ngroups) return;
  FsmView v = view_of(s_img, h);
  if (LOOK) {                                                     // (kThreads == 256: one entry each; every thread of the group is still here)
    S.lk16[tid] = static_cast<uint16_t>(v.cls2[tid] | (static_cast<uint32_t>(v.knd[tid]) << 8));
    __syncthreads();
  }
  v.lk16 = S.lk16;
  const uint32_t outside = LOOK ? h->outside_byte : 0u;           // what the positions around the haystack read as
  constexpr int tpw = FsmMode<MODE>::kTpw;
  uint32_t nrows_w = 0, fallback = 0, long_hit = 0;

  u32x4 x[4];
  uint32_t xbehind = 0;                                // look-around: the dword behind the window (the last step of the window's tail reads its first byte)
  auto issue_loads = [&](int jj) {
    const uint64_t wtn = group * (kWavesPerBlock * tpw) + static_cast<uint64_t>(jj) * kWavesPerBlock + wave;
    const uint64_t lo = wtn * static_cast<uint64_t>(kWaveTile);
    int nrec = 0;
    const int pre = lo ? 0 : kFsmLeft;                 // the first tile has nothing in front of it
    const uint64_t from = lo ? lo - kFsmLeft : 0;
    constexpr int kStaged = 4096 + (LOOK ? 4 : 0);
    if (jj < tpw && lo < a.len) {
      const uint64_t rem = a.len - from;
      nrec = rem >= static_cast<uint64_t>(kStaged - pre) ? kStaged - pre : static_cast<int>((rem + 3) & ~3ull);
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.hay) + (nrec ? from : 0), 0, nrec, 0x00020000);
#pragma unroll
    for (int k = 0; k < 4; k++)                        // window offsets below `pre` wrap around: out of range, read as zero
      x[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((lane + 64 * k) << 4) - pre, 0, 0);
    if (LOOK) xbehind = __builtin_amdgcn_raw_buffer_load_b32(rsrc, 4096 - pre, 0, 0);   // (every lane the same dword: one request)
  };
  issue_loads(0);
#if CXG_FSM_PROF
  uint64_t pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tlast = __builtin_readcyclecounter();
#endif

  // Two passes over the wave's tiles; the second only for tiles left behind by the first (fsm_resolve_exits), which is none
  // on input whose sets of possible states collapse.
  int jd = tpw;                                        // first tile of this wave left to the second pass (wave-uniform)
  for (int it = 0; it < 2 * tpw; it++) {
    if (it == tpw) {
      __syncthreads();
      if (S.ndefer == 0u) break;                       // uniform
      if (wave == 0) fsm_resolve_exits<kWavesPerBlock * tpw>(v, S.exit, S.mapx, S.mapu, a, group, lane);
      __syncthreads();
      if (a.dbg & 4u) break;                           // (CXG_DEBUG=4: timing experiment — no second pass, rows missing)
      if (jd < tpw) issue_loads(jd);
    }
    const int j = it < tpw ? it : it - tpw;
    if (it >= tpw && j < jd) continue;
    const bool map_only = it < tpw && jd < tpw;        // a tile in front (same wave) was deferred: rows must stay in order, so only
    bool deferred = false;                             // this tile's exit or map now, the tile itself in the second pass
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
      if (LOOK && lane == 0) *reinterpret_cast<uint32_t*>(win + 64 * kFsmStride) = xbehind;   // window position 4096
      issue_loads(j + 1);
      const uint64_t remaining = a.len - tile_lo;
      const int32_t rend = remaining > 0x7FFF0000ull ? 0x7FFF0000 : static_cast<int32_t>(remaining);
      if (LOOK && lane == 0) {                          // the byte in front of the haystack and the one behind its end (DS ops of a wave keep their order)
        if (tile_lo == 0) win[kFsmLeft - 1] = static_cast<uint8_t>(outside);
        if (rend <= kFsmWinEnd) { const uint32_t w = static_cast<uint32_t>(rend + kFsmLeft); win[w + (w >> 6) * 4u] = static_cast<uint8_t>(outside); }
      }
      wave_lds_sync();
      FSM_MARK(0);                                      // window staged
      // walks stay inside the window; with look-around a step also reads the byte behind its own (past the end of input
      // the window holds zeros: "not a word byte", as the reference treats the end of the text)
      constexpr int32_t kWinEnd = kFsmWinEnd - (LOOK ? 1 : 0);
      const int32_t budget = rend < kWinEnd ? rend : kWinEnd;
      const int32_t lowest = tile_lo ? -kFsmLeft : 0;
      // ... and a reverse step the byte in front (in front of the haystack's first byte the window holds zeros as well)
      const int32_t rev_lowest = (LOOK && tile_lo) ? lowest + 1 : lowest;
      FsmMem<LOOK> m;
      m.win = (lds_bytes_t)win;
      m.last_ = rend - 1;
      // ---- E + R: entry states, replay.  A lane's 64 bytes are two sub-chunks of 32 walked in lockstep (two
      // independent chains of dependent LDS reads per lane).
      const int32_t c0 = (lane - 1) * kFsmChunk;
      const bool owned = lane >= 1 && lane <= kWaveTile / kFsmChunk && c0 < rend;
      // shallow machines: the three lanes behind the tile walk the window's tail for nothing but its event bits (fsm.hpp "Round 6")
      const bool active = SHALLOW ? (lane >= 1 && c0 < rend) : owned;
      const int32_t cc[2] = {c0, c0 + kFsmSub};
      uint64_t KK[2] = {0ull, 0ull};                   // shallow: the event bits of the two sub-chunks (two per byte)
      uint32_t xend[2] = {0u, 0u};                     // ... and the row behind each of them
      const bool second = active && cc[1] < rend;
      const bool whole = c0 + kFsmChunk <= rend && c0 + kFsmChunk <= budget;   // both sub-chunks are staged data
      const int q_tile = j * kWavesPerBlock + wave;    // index of the tile inside the group
      FsmLane L[2];
      L[0].nrows = L[1].nrows = 0;
      L[0].xc1 = L[1].xc1 = 0;
      L[0].max_rows = L[1].max_rows = kLaneRows;
      L[0].max_events = L[1].max_events = kLaneEvents;
      LdsRows rows[2] = {{&s_lrow[wave][(2 * lane) * kLaneRows]}, {&s_lrow[wave][(2 * lane + 1) * kLaneRows]}};
      LdsEvents evs[2] = {{&s_lev[wave][SHALLOW ? 0 : (2 * lane) * kLaneEvents]}, {&s_lev[wave][SHALLOW ? 0 : (2 * lane + 1) * kLaneEvents]}};
      // entry states: from "any state" over the 16 bytes in front of each sub-chunk; when a set has not collapsed by
      // then, over 64 bytes (rare on text)
      uint32_t entry[2] = {0u, 0u};
      if (active) {
        const bool at_origin = tile_lo + static_cast<uint64_t>(c0) == 0;
        const int32_t from[2] = {at_origin ? cc[1] - 16 : cc[0] - 16, cc[1] - 16};     // (the haystack's first chunk starts in state 0)
        if (!(CXG_FSM_ABL & 1)) fsm_walk_n<2>(v, m, v.top_off, from, 16, entry);
        if (at_origin) entry[0] = m.origin(v);
        if (entry[0] >= v.u_lo) entry[0] = fsm_walk(v, m, v.top_off, cc[0] - 64, cc[0], true);
        if (entry[1] >= v.u_lo) entry[1] = fsm_walk(v, m, v.top_off, at_origin ? 0 : cc[1] - 64, cc[1], true);
        if (at_origin && entry[1] >= v.u_lo) entry[1] = fsm_walk(v, m, m.origin(v), 0, cc[1], true);   // from the true start state
      }
      FSM_MARK(1);                                      // entry states
      const bool unres0 = active && entry[0] >= v.u_lo, unres1 = second && entry[1] >= v.u_lo;
      const unsigned long long um0 = __ballot(unres0), um1 = __ballot(unres1);
      if (map_only && (um0 | um1) == 0ull) {            // only the exit: the last sub-chunk of the tile from its known entry
        const unsigned long long am = __ballot(owned);
        const int ll = 63 - __builtin_clzll(am | 1ull);
        const int sbl = second ? 1 : 0;
        const int32_t to = cc[sbl] + kFsmSub < rend ? cc[sbl] + kFsmSub : rend;
        const uint32_t xe = (owned && lane == ll) ? fsm_canon(v, fsm_walk(v, m, sbl ? entry[1] : entry[0], cc[sbl], to, true)) : 0u;
        const uint32_t ex = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(xe), ll));
        if (lane == 0) publish_tile_exit(S.exit, a.status2, group, q_tile, kWavesPerBlock * tpw, a.epoch, ex);
        deferred = true;
      } else if ((um0 | um1) == 0ull) {
        if (active) {
          if (SHALLOW) {                               // (bytes behind the end of the input read as zeros: their bits are masked below)
            FsmTraceS t[2] = {{entry[0], 0u, 0u}, {entry[1], 0u, 0u}};
            if (!(CXG_FSM_ABL & 2)) fsm_fast_shallow<2>(v, m, cc, t);
            FSM_MARK(2);                                // lockstep walk
            KK[0] = (static_cast<uint64_t>(t[0].k1) << 32) | t[0].k0; KK[1] = (static_cast<uint64_t>(t[1].k1) << 32) | t[1].k0;
            xend[0] = t[0].x & ~3u; xend[1] = t[1].x & ~3u;
          } else if (whole) {
            FsmTrace t[2] = {{entry[0], 0u, 0u, kLaneEvents}, {entry[1], 0u, 0u, kLaneEvents}};
            fsm_fast<2>(v, m, cc, t, evs);
            fsm_finish(v, m, entry[0], &t[0], cc[0], cc[1], rend, budget, L[0], rows[0], evs[0]);
            fsm_finish(v, m, entry[1], &t[1], cc[1], cc[1] + kFsmSub, rend, budget, L[1], rows[1], evs[1]);
          } else {                                     // the input ends inside this lane's bytes
            fsm_finish(v, m, entry[0], static_cast<const FsmTrace*>(nullptr), cc[0], cc[1], rend, budget, L[0], rows[0], evs[0]);
            if (second) fsm_finish(v, m, entry[1], static_cast<const FsmTrace*>(nullptr), cc[1], cc[1] + kFsmSub, rend, budget, L[1], rows[1], evs[1]);
          }
        }
      } else {
        // ---- Some sub-chunk's set of possible entry states did not collapse: input without synchronising structure
        // (`1.1.1.1...` for the IPv4 pattern).  Exact, in four steps.  (1) Sub-chunks with a known entry are replayed;
        // their end state is the next sub-chunk's true entry.  (2) Every unresolved sub-chunk walks its 32 bytes once
        // from each member of its set (<= 8, listed in the image): a MAP member -> end state.  (3) One pass in order over
        // the unresolved sub-chunks chains the maps: true entry = end state of the sub-chunk in front (a replayed one,
        // the previous link of the chain, or — for the tile's first — the exit the previous tile published).
        // (4) The unresolved sub-chunks are replayed from their true entries.  Serial only in step 3: ~20 scalar
        // operations per unresolved sub-chunk.
        auto replay_one = [&](int sb, uint32_t from_state) {   // one sub-chunk from a known entry state
          const int32_t c1s = cc[sb] + kFsmSub;
          if (SHALLOW) {
            const int32_t c1a[1] = {cc[sb]};
            FsmTraceS ts[1] = {{from_state, 0u, 0u}};
            fsm_fast_shallow<1>(v, m, c1a, ts);
            KK[sb] = (static_cast<uint64_t>(ts[0].k1) << 32) | ts[0].k0;
            xend[sb] = ts[0].x & ~3u;
          } else {
            fsm_replay(v, m, from_state, cc[sb], c1s, rend, budget, L[sb], rows[sb], evs[sb]);
          }
        };
        uint32_t xc[2] = {0u, 0u};                     // end state of a replayed sub-chunk (own row of the state)
        // F: member | end state << 16 — the serial steps below (one per unresolved sub-chunk of the tile, up to 126) take both
        // from the lane's registers with one v_readlane; looking the member up in the image there (a dependent LDS read
        // per member and step) was most of a tile's time on input without synchronising structure.
        uint32_t F[2][kFsmMembers];
#pragma unroll
        for (int sb = 0; sb < 2; sb++) {
#pragma unroll
          for (int jm = 0; jm < kFsmMembers; jm++) F[sb][jm] = 0xFFFFFFFFu;
          const bool has = sb ? second : active;
          const bool unres = sb ? unres1 : unres0;
          const int32_t c1s = cc[sb] + kFsmSub;
          if (has && !unres && map_only) {
            xc[sb] = fsm_canon(v, fsm_walk(v, m, entry[sb], cc[sb], c1s < rend ? c1s : rend, true));
          } else if (has && !unres) {
            replay_one(sb, entry[sb]);
            xc[sb] = fsm_canon(v, SHALLOW ? xend[sb] : L[sb].xc1);
          } else if (has) {
            if (fsm_member(v, entry[sb], 0) == 0xFFFFu) fallback |= 1u;      // more than 8 possible states: not listed
            const int32_t to = c1s < rend ? c1s : rend;
            for (int jm = 0; jm < kFsmMembers; jm++) {
              const uint32_t mj = fsm_member(v, entry[sb], static_cast<uint32_t>(jm));
              if (mj != 0xFFFFu) F[sb][jm] = mj | (fsm_canon(v, fsm_walk(v, m, mj, cc[sb], to, true)) << 16);
            }
          }
        }
        // (2b) + (3), round 3: ONE inclusive scan over the sub-chunk maps instead of two serial walks over them.  Every lane
        // folds its two sub-chunks into one map (a sub-chunk with a known entry is a constant map: its end state whatever
        // came before); six Hillis-Steele levels compose them — S_l = tile entry -> state behind lane l, keys = the members
        // of the tile's first set, or a constant as soon as a known sub-chunk lies in front.  S of the last lane is the tile's
        // map (published at once when the exit in front is not known yet: fsm_resolve_exits) resp. its exit; S of lane l - 1
        // applied to the exit in front gives lane l its true entry — all lanes at once.  The serial versions (a chase of the
        // <= 8 candidates and a chain of true entries, one step per unresolved sub-chunk, up to 126 per tile and ~35
        // instructions each) were 13 k of the 31 k instructions of a tile on input without synchronising structure.
        FSM_MARK(2);                                    // (unresolved tiles: replays of the known sub-chunks + member maps)
        const unsigned long long am_all = __ballot(owned);             // (the tile ends behind its last OWNED lane; the tail lanes take part in the scan for their entries only)
        const int last_lane = 63 - __builtin_clzll(am_all | 1ull);
        uint32_t pred_exit = 0u;
        const bool first_open = (um0 >> 1) & 1ull;
        const uint32_t u1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(entry[0]), 1));
        auto look = [](const uint32_t (&M)[kFsmMembers], uint32_t x) -> uint32_t {     // member -> end state; 0xFFFF: not listed
          uint32_t r = 0xFFFFu;
#pragma unroll
          for (int jm = 0; jm < kFsmMembers; jm++) r = ((M[jm] & 0xFFFFu) == x) ? (M[jm] >> 16) : r;
          return r;
        };
        bool isc = !unres0;                              // this lane's map is a constant ...
        uint32_t cv = xc[0];                             // ... this one
        uint32_t P[kFsmMembers];
#pragma unroll
        for (int jm = 0; jm < kFsmMembers; jm++) P[jm] = F[0][jm];
        if (second) {
          if (!unres1) { isc = true; cv = xc[1]; }
          else if (isc) cv = look(F[1], cv);
          else {
#pragma unroll
            for (int jm = 0; jm < kFsmMembers; jm++) P[jm] = (P[jm] & 0xFFFFu) | (look(F[1], P[jm] >> 16) << 16);
          }
        }
#pragma unroll 1
        for (int d = 1; d < 64; d <<= 1) {               // S_l := own o S_{l-d}
          const int src = lane - d;
          const bool pisc = __shfl(static_cast<int>(isc), src, 64) != 0;
          const uint32_t pcv = static_cast<uint32_t>(__shfl(static_cast<int>(cv), src, 64));
          uint32_t N[kFsmMembers];
          const uint32_t ncv = look(P, pcv);
#pragma unroll
          for (int jm = 0; jm < kFsmMembers; jm++) {
            const uint32_t q = static_cast<uint32_t>(__shfl(static_cast<int>(P[jm]), src, 64));
            N[jm] = (q & 0xFFFFu) | (look(P, q >> 16) << 16);
          }
          if (src >= 1 && !isc) {                        // (a constant absorbs whatever lies in front of it; lanes 1.. are the active ones)
            if (pisc) { isc = true; cv = ncv; }
            else {
#pragma unroll
              for (int jm = 0; jm < kFsmMembers; jm++) P[jm] = N[jm];
            }
          }
        }
        // the tile as a whole
        const bool t_isc = __builtin_amdgcn_readlane(static_cast<int>(isc), last_lane) != 0;
        const uint32_t t_cv = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cv), last_lane));
        if (t_isc) {                                     // its exit does not depend on its entry: successors need not wait for anything
          if (t_cv == 0xFFFFu) fallback |= 1u;
          else if (lane == 0) publish_tile_exit(S.exit, a.status2, group, q_tile, kWavesPerBlock * tpw, a.epoch, t_cv);
        }
        if (map_only) {
          if (!t_isc && lane == last_lane) {
#pragma unroll
            for (int jm = 0; jm < kFsmMembers; jm++) S.mapx[q_tile][jm] = static_cast<uint16_t>(P[jm] >> 16);
            S.mapu[q_tile] = static_cast<uint16_t>(u1);
          }
          deferred = true;
        } else if (first_open) {
          // The exit in front: a short look in the first pass (the wave of the tile in front runs beside this one), then
          // rather the map than a blocked wave; in the second pass it is there.
          const uint32_t pw = it < tpw ? peek_tile_exit(S.exit, a.status2, group, q_tile, a.epoch, q_tile ? 64u : 2u)
                                       : (wait_tile_exit(S.exit, a.status2, a.err, group, q_tile, a.epoch, lane) | kExitValid);
          if (!(pw & kExitValid)) {
            if (!t_isc && lane == last_lane) {
#pragma unroll
              for (int jm = 0; jm < kFsmMembers; jm++) S.mapx[q_tile][jm] = static_cast<uint16_t>(P[jm] >> 16);
              S.mapu[q_tile] = static_cast<uint16_t>(u1);
            }
            jd = j;
            if (lane == 0) S.ndefer = 1u;
            deferred = true;
          } else {
            pred_exit = pw & 0xFFFFu;
            if (!t_isc) {
              uint32_t TP[kFsmMembers];
#pragma unroll
              for (int jm = 0; jm < kFsmMembers; jm++) TP[jm] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(P[jm]), last_lane));
              const uint32_t ex = look(TP, pred_exit);
              if (ex == 0xFFFFu) fallback |= 1u;           // the true state is not in the listed set: cannot happen
              else if (lane == 0) publish_tile_exit(S.exit, a.status2, group, q_tile, kWavesPerBlock * tpw, a.epoch, ex);
            }
          }
        }
        FSM_MARK(6);                                    // (scan over the maps, look at the exit in front)
        // (3) true entries: the state behind lane l - 1 is this lane's entry
        uint32_t true_entry[2] = {entry[0], entry[1]};
        if (!deferred) {
          const bool eisc = __shfl(static_cast<int>(isc), lane - 1, 64) != 0;
          const uint32_t ecv = static_cast<uint32_t>(__shfl(static_cast<int>(cv), lane - 1, 64));
          uint32_t EP[kFsmMembers];
#pragma unroll
          for (int jm = 0; jm < kFsmMembers; jm++) EP[jm] = static_cast<uint32_t>(__shfl(static_cast<int>(P[jm]), lane - 1, 64));
          const uint32_t before = lane <= 1 ? pred_exit : (eisc ? ecv : look(EP, pred_exit));
          if (unres0) { true_entry[0] = before; if (before == 0xFFFFu) { fallback |= 1u; true_entry[0] = 0u; } }
          if (unres1) {
            const uint32_t mid = unres0 ? look(F[0], true_entry[0]) : xc[0];
            true_entry[1] = mid;
            if (mid == 0xFFFFu) { fallback |= 1u; true_entry[1] = 0u; }
          }
        }
        // (4) replay of the unresolved sub-chunks
#pragma unroll
        for (int sb = 0; sb < 2; sb++)
          if (!deferred && (sb ? unres1 : unres0)) replay_one(sb, true_entry[sb]);
      }
      if (!deferred) {
      if (!SHALLOW && active) fallback |= (L[0].flags | L[1].flags) << 1;
      {  // this tile's exit state, for a following tile whose first set does not collapse
        const unsigned long long am = __ballot(owned);
        const int ll = 63 - __builtin_clzll(am | 1ull);
        const uint32_t xl = SHALLOW ? (second ? xend[1] : xend[0]) : (second ? L[1].xc1 : L[0].xc1);
        const uint32_t ex = fsm_canon(v, static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(xl), ll)));
        if (lane == 0) publish_tile_exit(S.exit, a.status2, group, q_tile, kWavesPerBlock * tpw, a.epoch, ex);
      }
      FSM_MARK(3);                                      // rows of the lanes (+ divergence of the whole E/R block)
      // ---- S: rows in lane order, then their starts
      if (SHALLOW) {
        tot = fsm_rows_from_events<kRowsPerWave>(KK, active, owned, rend, c0, lane, s_re[wave], nrows_w, fallback,
                                                 [&]() { return fsm_u16(v.tab, xend[1] + v.ncls2 + 2u); });
      } else {
      const uint32_t nl0 = (active && !(CXG_FSM_ABL & 4)) ? L[0].nrows : 0u, nl = nl0 + ((active && !(CXG_FSM_ABL & 4)) ? L[1].nrows : 0u);
      const uint32_t incl = wave_inclusive_sum(nl);
      tot = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
      const uint32_t first = nrows_w + incl - nl;
      for (uint32_t r = 0; r < nl; r++)
        if (first + r < static_cast<uint32_t>(kRowsPerWave)) s_re[wave][first + r] = r < nl0 ? rows[0].slot[r] : rows[1].slot[r - nl0];
      }
      wave_lds_sync();
      FSM_MARK(4);                                      // rows gathered
      if (a.out != nullptr || a.max_len != 0) {
        for (uint32_t q = lane; q < tot && nrows_w + q < static_cast<uint32_t>(kRowsPerWave); q += 64) {
          const int32_t e = s_re[wave][nrows_w + q];
          // first row of the tile: no bound known here (checked after the barrier); one byte below the window makes a
          // reverse DFA that is still alive there report `over`
          const int32_t bound = q ? static_cast<int32_t>(s_re[wave][nrows_w + q - 1]) : (tile_lo ? lowest - 1 : 0);
          uint32_t over = 0;
          // (a text-start anchor is the loop's business: the walk that arrives at position 0 alive asks the state)
          const int32_t s = (CXG_FSM_FAST_STARTS && (v.rev_text_col == 0u || tile_lo != 0)) ? fsm_match_start16(v, m, e, bound, rev_lowest, over)
                                                                                              : fsm_match_start(v, m, e, bound, rev_lowest, over, tile_lo == 0 ? 0 : kFsmNoStart);
          // over: the reverse DFA was still alive at the window's first byte and the haystack goes on in front of it
          // (a match longer than the 64 bytes staged there): length 0 = unresolved, finished in the epilogue
          const uint32_t len = (over || s == kFsmNoStart) ? 0u : static_cast<uint32_t>(e - s);
          if (s == kFsmNoStart && !over) fallback |= 64u;           // internal: the reverse DFA rejects a match the transducer reported
          s_rl[wave][nrows_w + q] = static_cast<uint16_t>(len);
        }
      }
      }                                                 // !deferred
    }
    FSM_MARK(5);                                        // starts
    if (deferred) continue;
    if (lane == 0) s_cnt[wave][j] = tot;
    nrows_w += tot;
  }
#if CXG_FSM_PROF
  if (a.prof && lane == 1) {                          // lane 1 takes every branch of an ordinary tile (lane 0 only stages)
    for (int i = 0; i < 7; i++) atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 8 + i), static_cast<unsigned long long>(pacc[i]));
    atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 15), 1ull);
  }
#endif
  if (nrows_w > static_cast<uint32_t>(kRowsPerWave)) fallback |= 32u;
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
    s_tail[q] = (n && st + n <= static_cast<uint32_t>(kRowsPerWave)) ? gorigin + static_cast<int64_t>(q) * kWaveTile + s_re[w][st + n - 1] : -1;
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
      if (r >= static_cast<uint32_t>(kRowsPerWave)) continue;
      int64_t e = tb + s_re[wave][r], s = e - s_rl[wave][r];
      if ((i == 0 || s == e) && (a.out != nullptr || a.max_len != 0)) {
        // the tile's first row was walked without a bound (its predecessor is another wave's row); an unresolved row
        // (s == e) left the window.  Previous end: inside the tile, else the nearest earlier tile of the group with rows;
        // the group's first row is checked by k_fsm_fix_heads.
        int64_t prev = -1;
        if (i > 0) prev = tb + s_re[wave][r - 1];
        else for (int p = q - 1; p >= 0 && prev < 0; p--) prev = s_tail[p];
        if (prev > s || s == e) {                                           // rare: walk again from HBM / L2, bounded
          const int64_t lo = prev > 0 ? prev : 0;
          uint32_t sr = v.rev_start_off;
          if (LOOK) sr = fsm_u16(v.knd, 256u + 2u * v.nk + 2u * ((v.knd[a.hay[e - 1]] >> 1) * v.nk + ((static_cast<uint64_t>(e) < a.len ? v.knd[a.hay[e]] : (LOOK == 2 ? v.end_col : v.knd[outside])) >> 1)));
          int64_t st = -1, at = e - 1;
          for (; at >= lo; at--) {
            if (e - at > kSerialLimit) { raise_err(a.err, kErrSerialLimit); break; }
            sr = fsm_u16(v.tab, (sr & ~1u) + v.cls2[a.hay[at]] + (LOOK ? v.knd[at > 0 ? a.hay[at - 1] : outside] : 0u));
            if (sr == v.rev_dead) break;
            if (sr & 1u) st = at;
          }
          if (LOOK && v.rev_text_col != 0u && at < 0 && sr != v.rev_dead && fsm_u16(v.tab, (sr & ~1u) + v.rev_text_col + v.knd[a.hay[0]]) != 0u) st = 0;   // text-start anchor (fsm.hpp fsm_match_start)
          if (st < 0) raise_err(a.err, 8u | (64u << 8)); else s = st;
        }
      }
      if (a.max_len != 0 && static_cast<uint64_t>(e - s) > a.max_len) long_hit = 1;
      if (a.out != nullptr && dst + i < a.cap) {
        longlong2 o; o.x = a.base + s; o.y = a.base + e;
        store_pair_nt(a.out + (dst + i) * a.row_width, o.x, o.y);
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
  const uint8_t* cls2 = a.blob + h->cls_off;
  const uint8_t* knd = a.blob + h->knd_off;
  const uint8_t* rev = a.blob + sizeof(FsmHeader);                 // (reverse entries count from the end of the header)
  const uint32_t rdead = h->rev_off - static_cast<uint32_t>(sizeof(FsmHeader));
  const bool look = h->nk > 1;
  const uint32_t outside = h->outside_byte;
  const uint32_t right = static_cast<uint64_t>(e) < a.len ? knd[a.hay[e]] : (h->end_col ? h->end_col : knd[outside]);   // (fsm.hpp "End of text")
  uint32_t sr = look ? fsm_u16(knd, 256u + 2u * h->nk + 2u * ((knd[a.hay[e - 1]] >> 1) * h->nk + (right >> 1))) : h->rev_start_off;
  int64_t st = -1;
  for (int64_t at = e - 1; at >= prev; at--) {
    sr = fsm_u16(rev, (sr & ~1u) + cls2[a.hay[at]] + (look ? knd[at > 0 ? a.hay[at - 1] : outside] : 0u));
    if (sr == rdead) break;
    if (sr & 1u) st = at;
  }
  if (st < 0) { raise_err(a.err, 8u | (64u << 8)); return; }
  row[0] = a.base + st;
}

namespace {
template <int IMG, int LOOK>
void launch_fsm_img(const ScanArgs& a, bool shallow, int mode, dim3 grid, dim3 block, hipStream_t stream) {
  if (shallow) {
    if (mode == 0) hipLaunchKernelGGL((k_scan_fsm<true, IMG, 0, LOOK>), grid, block, 0, stream, a);
    else if (mode == 1) hipLaunchKernelGGL((k_scan_fsm<true, IMG, 1, LOOK>), grid, block, 0, stream, a);
    else if (mode == 2) hipLaunchKernelGGL((k_scan_fsm<true, IMG, 2, LOOK>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_scan_fsm<true, IMG, 3, LOOK>), grid, block, 0, stream, a);
  } else {
    if (mode == 0) hipLaunchKernelGGL((k_scan_fsm<false, IMG, 0, LOOK>), grid, block, 0, stream, a);
    else if (mode == 1) hipLaunchKernelGGL((k_scan_fsm<false, IMG, 1, LOOK>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_scan_fsm<false, IMG, 3, LOOK>), grid, block, 0, stream, a);   // (machines with event lists: their per-lane buffers dominate the LDS, modes 2 and 3 are one)
  }
}
}  // namespace

hipError_t launch_scan_fsml(const ScanArgs& a, uint32_t lds_bytes, bool shallow, int look, hipStream_t stream, uint32_t direct_bytes, int mode);   // scan_fsml.hip

// lean != 0: the lean kernel (the machine is shallow); direct_bytes != 0 with it: its direct mode (the caller has checked that the image
// carries the section).  look: 0 / 1 / 2 as the kernels' LOOK (2: FsmHeader::end_col != 0 — rare programs, one image size only)
// mode: 0..3 (scan_fsm_common.hpp FsmMode; a.tiles_per_wave = 8, 2, 1, 1 goes with it)
hipError_t launch_scan_fsm(const ScanArgs& a, uint32_t lds_bytes, bool shallow, int look, hipStream_t stream, uint32_t direct_bytes, bool lean, int mode) {
  const dim3 grid(static_cast<unsigned>(a.ngroups)), block(kThreads);
  if (mode < 0 || mode > 3 || a.tiles_per_wave != static_cast<uint32_t>(mode == 0 ? kTilesPerWave : (mode == 1 ? kDenseTilesPerWave : 1))) return hipErrorInvalidValue;
  if (lds_bytes > 28672) return hipErrorInvalidValue;
  if (lean) { const hipError_t el = launch_scan_fsml(a, lds_bytes, shallow, look, stream, direct_bytes, mode); if (el != hipSuccess) return el; }
  else if (look == 2) launch_fsm_img<28672, 2>(a, shallow, mode, grid, block, stream);             // end-of-text programs
  else if (look) {                                                                                   // word-boundary programs
    if (lds_bytes <= 3072) launch_fsm_img<3072, 1>(a, shallow, mode, grid, block, stream);
    else if (lds_bytes <= 10240) launch_fsm_img<10240, 1>(a, shallow, mode, grid, block, stream);
    else launch_fsm_img<28672, 1>(a, shallow, mode, grid, block, stream);
  }
  else if (lds_bytes <= 3072) launch_fsm_img<3072, 0>(a, shallow, mode, grid, block, stream);   // instantiations by image size
  else if (lds_bytes <= 10240) launch_fsm_img<10240, 0>(a, shallow, mode, grid, block, stream);
  else launch_fsm_img<28672, 0>(a, shallow, mode, grid, block, stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || a.out == nullptr || a.ngroups < 2) return e;
  const unsigned fb = 256, fg = static_cast<unsigned>((a.ngroups + fb - 1) / fb);
  hipLaunchKernelGGL(k_fsm_fix_heads, dim3(fg), dim3(fb), 0, stream, a);
  return hipGetLastError();
}

}  // namespace cxgdev
