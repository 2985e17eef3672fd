// scan_chain_wave.hip — FindAll for programs whose anchored DFA is ONE complete, ordered chain of
// run(class+) / byte(class) steps (walk.hpp ChainAux, program.cc extractChain): `\d+\.\d+\.\d+\.\d+`
// (UseDigitPrefilter), a literal such as `error`, `[a-z]+=\d+` (UseDFA).  Sixth kernel generation: the whole
// search — candidates, match ends, segment ownership — is bit-parallel on class bitmaps; no DFA table,
// no per-match walk, no workgroup barrier on the data path.
//
// Reference semantics kept (meta/findall.go:176-283 over the DFA searches of dfa/lazy/lazy.go): leftmost-first
// match at or after `pos`, non-empty, next search from its end.  For a complete chain "a match starts at p" is
// "the chain can be taken from p" and its end is where the (greedy, deterministic) steps lead.
//
// One wave64 owns a wave-tile of 60 x 64 B = 3840 B and reads 256 B of halo behind it: 4096 B = 64 bitmap
// words, one per lane.
//   A  4 coalesced 16-byte loads per lane (software-pipelined one tile ahead); per class a 16-bit mask per
//      vector (SWAR compare + shift-or gather) into the wave's LDS scratch; lane l reads forward word l and,
//      bit-reversed, word 63-l (the reversed bitmap: bit i <-> byte 4095-i).
//   B  chain right to left on the REVERSED words: a run step is one multiword addition (carries move towards
//      earlier bytes), resolved across lanes with two ballots: recv = (P + (G<<1)) ^ P.  Result: the starts S.
//   O  ownership, wave-uniform: with zA = first synchronising byte at >= -1 and zB = first one at >= 3839, the
//      tile owns exactly the starts in (zA, zB] (a match never crosses a synchronising byte).
//   F  chain left to right on the FORWARD words from the owned starts: run step M = (M + C) & ~C, byte step
//      M <<= 1.  Result: the ends E; the k-th start pairs with the k-th end ("ordered" chains).
//   P  ranks of starts and ends by one packed DPP prefix sum; both compactions keep the order, so start k and
//      end k land in row k of the wave's row buffer.  FindAll drops a match that begins inside the previous
//      emitted one: detected as "rank of the start != number of ends at or before it", resolved serially (rare).
// A workgroup (4 waves) takes ONE ticket per 32 wave-tiles (120 KiB); after a single barrier the group's
// rows are ordered, looked back (block_common.hpp) and written as coalesced 16-byte stores.
// Fallback flag (err bit 8: the host reruns the scan with the table-walking kernels): no synchronising byte
// in a halo, row buffer overflow (> 512 matches per wave and group), or a violated pairing invariant.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "block_common.hpp"
#include "scan_dfa.h"
#include "walk.hpp"
#include "wave_common.hpp"

#ifndef CXG_CAP_WAVES
#define CXG_CAP_WAVES 5       // SETS / CAP / BND instantiations: 90 VGPRs without scratch (6 waves: 80 VGPRs + 7 spilled, 3 % slower on config 5)
#endif
#ifndef CXG_CHAIN_WAVES
#define CXG_CHAIN_WAVES 8
#endif
#ifndef CXG_CHAIN_SCHED_BARRIER
#define CXG_CHAIN_SCHED_BARRIER 1
#endif
// -DCXG_CHAIN_PROF=1 (experiments only): s_memtime at the phase boundaries, cycles summed into ScanArgs::prof[8..15]
#ifndef CXG_CHAIN_PROF
#define CXG_CHAIN_PROF 0
#endif
#if CXG_CHAIN_PROF
#define PHASE_MARK(i) do { const uint64_t t_ = __builtin_readcyclecounter(); pacc[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define PHASE_MARK(i) do { } while (0)
#endif
// -DCXG_ABL=n (experiments only, results WRONG): 1 = no chain passes and no rows, 2 = no forward pass and no rows,
// 3 = no rows, 4 = no class masks either (with 1).  Used to attribute instruction counts to the phases.
#ifndef CXG_ABL
#define CXG_ABL 0
#endif
#ifndef CXG_CHAIN_PREFETCH
#define CXG_CHAIN_PREFETCH 1
#endif

namespace cxgdev {

namespace {

constexpr int kWRows = 512;                       // rows buffered per wave per group
constexpr int32_t kFar = 1 << 20;                 // "no such byte" position

// 0x80 flag in every byte of x that is NOT in the class (inverted once per 16 bytes by the caller).
// KIND is a template parameter so that the per-class switch is taken once per wave-tile, not once per dword.
template <int KIND>
__device__ __forceinline__ uint32_t notcls4(uint32_t x, uint32_t lo4, uint32_t hi4) {   // lo4/hi4: class bounds splat over the bytes
  // (x ^ c) has the top bit of x in every byte (c < 0x80), so the final "| top bit" can take x itself: three
  // ternary-logic / add ops per dword.
  if (KIND == kClsDigit) return ((((x ^ 0x30303030u) & 0x7F7F7F7Fu) + 0x76767676u) | x) & 0x80808080u;
  if (KIND == kClsByte) return ((((x ^ lo4) & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
  const uint32_t ge = ((x | 0x80808080u) - lo4) & 0x80808080u;
  const uint32_t gt = ((x & 0x7F7F7F7Fu) + hi4) & 0x80808080u;     // hi4 = (0x7F - hi) splat
  return ~(ge & ~gt & ~x) & 0x80808080u;
}
// 16 class bits of a 16-byte vector.  The four 0x80 flags of a dword are gathered by one v_dot4_u32_u8 against
// the weights (1,2,4,8) resp. (16,32,64,128): 128 x the byte of flags accumulates over a dword pair.
template <int KIND>
__device__ __forceinline__ uint32_t cls16(const u32x4& x, uint32_t lo4, uint32_t hi4) {
  const uint32_t lo = __builtin_amdgcn_udot4(notcls4<KIND>(x.y, lo4, hi4), 0x80402010u, __builtin_amdgcn_udot4(notcls4<KIND>(x.x, lo4, hi4), 0x08040201u, 0u, false), false);
  const uint32_t hi = __builtin_amdgcn_udot4(notcls4<KIND>(x.w, lo4, hi4), 0x80402010u, __builtin_amdgcn_udot4(notcls4<KIND>(x.z, lo4, hi4), 0x08040201u, 0u, false), false);
  return ((lo >> 7) | (hi << 1)) ^ 0xFFFFu;
}
// One class of one FULL wave-tile window (4096 bytes present): 4 vectors per lane -> 4 16-bit pieces of the
// forward bitmap in LDS.
template <int KIND>
__device__ __forceinline__ void classify_tile(const u32x4 (&x)[4], uint32_t lo, uint32_t hi, int lane, uint16_t* pieces) {
  const uint32_t lo4 = lo * 0x01010101u, hi4 = (0x7Fu - hi) * 0x01010101u;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    pieces[lane + 64 * k] = static_cast<uint16_t>(cls16<KIND>(x[k], lo4, hi4));
#if CXG_CHAIN_SCHED_BARRIER
    __builtin_amdgcn_sched_barrier(0);                              // one vector at a time: fewer live temporaries
#endif
  }
}
__device__ __forceinline__ void classify_tile_set(const u32x4 (&x)[4], const SetRanges& r, int lane, uint16_t* pieces) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t lo = __builtin_amdgcn_udot4(notset4(x[k].y, r), 0x80402010u, __builtin_amdgcn_udot4(notset4(x[k].x, r), 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(notset4(x[k].w, r), 0x80402010u, __builtin_amdgcn_udot4(notset4(x[k].z, r), 0x08040201u, 0u, false), false);
    pieces[lane + 64 * k] = static_cast<uint16_t>(((lo >> 7) | (hi << 1)) ^ 0xFFFFu);
#if CXG_CHAIN_SCHED_BARRIER
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
}
}  // namespace

// The chain description held in scalar registers for the whole kernel (no LDS/VGPR traffic in the op loops).
template <int NCLS, bool SETS>
struct ChainRegs {
  uint32_t nops;
  uint64_t op_is_run;      // bit k: step k is a run
  uint64_t op_cls2, op_cls2_hi;   // 2 bits per step: its class (steps 0..31, 32..63)
  __device__ __forceinline__ bool is_run(uint32_t k) const { return ((op_is_run >> k) & 1ull) != 0; }
  __device__ __forceinline__ uint32_t cls(uint32_t k) const { return static_cast<uint32_t>(((k < 32u ? op_cls2 >> (2u * k) : op_cls2_hi >> (2u * (k - 32u)))) & 3ull); }
  uint32_t kind[NCLS], lo[NCLS], hi[NCLS];
  const ChainAux* aux;     // uniform address: the ranges of kClsSet classes are read through the scalar cache at use
  __device__ __forceinline__ bool has(int c, uint32_t b) const {
    if (SETS && kind[c] == kClsSet) {
      bool in = false;
      for (uint32_t r = 0; r < aux->cls_nr[c]; r++) in = in || (b >= aux->cls_rlo[c][r] && b <= aux->cls_rhi[c][r]);
      return in;
    }
    return kind[c] == kClsDigit ? (b - 0x30u) < 10u : (b >= lo[c] && b <= hi[c]);
  }
  __device__ __forceinline__ SetRanges ranges(int c) const {
    SetRanges r;
    r.n = aux->cls_nr[c];
#pragma unroll
    for (int q = 0; q < kChainMaxRanges; q++) { r.lo4[q] = aux->cls_rlo[c][q] * 0x01010101u; r.hi4[q] = (0x7Fu - aux->cls_rhi[c][q]) * 0x01010101u; }
    return r;
  }
  __device__ __forceinline__ bool in_alphabet(uint32_t b) const {
    bool r = false;
#pragma unroll
    for (int c = 0; c < NCLS; c++) r = r || has(c, b);
    return r;
  }
};
// Up to four class words as NAMED registers (an array here ends up in scratch memory, indexed by the scalar class
// id, for NCLS >= 3).  Wave-uniform select by compare chain.
struct Words4 {
  uint64_t w0, w1, w2, w3;
  template <int NCLS>
  __device__ __forceinline__ uint64_t pick(uint32_t ci) const {
    uint64_t v = w0;
    if (NCLS > 1) v = (ci == 1u) ? w1 : v;
    // the empty asm keeps the compiler from folding a longer select chain into an indexed load from scratch memory
    if (NCLS > 2) { asm volatile("" : "+v"(v)); v = (ci == 2u) ? w2 : v; }
    if (NCLS > 3) { asm volatile("" : "+v"(v)); v = (ci == 3u) ? w3 : v; }
    return v;
  }
  __device__ __forceinline__ void and_all(uint64_t m) { w0 &= m; w1 &= m; w2 &= m; w3 &= m; }
};

// SETS: some class is a union of ranges (kClsSet); a separate instantiation keeps that code (and its register
// pressure) out of the kernels of single-range programs.
// CAP: the rows carry capture slots (ScanArgs::caps, walk.hpp ChainCaps): the ends of up to four runs are compacted
// next to the starts and ends (dynamic LDS, 4 KiB per run), and every slot is one of those positions plus a constant.
// DENSE: two tiles per wave instead of eight (four times the row-buffer room per tile) for match-dense input; the host
// switches after a row-buffer overflow (capi_ladder.hip).  A template parameter: a run-time tile count cost the default 1.2 %.
// ALTK > 0: the chain is run(class 0) (byte(separator) run(class 0)){ALTK-1} — fields of one class with single-byte
// separators (`\d+\.\d+\.\d+\.\d+`, `\d+:\d+:\d+`, `\w+@\w+\.\w+`): the step loops unroll with constant step kinds, the
// runs use class 0 without a select (with two classes the separator is class 1, with more it is read from the
// description), no loop control; 0: any chain, steps read from the description.
// BND: bounded repetition (`\d{1,3}\.\d{1,3}`...).  The chain evaluated is the surrogate with every run unbounded; the
// rows are then filtered by field length with the run ends compacted as for CAP (ScanArgs::caps carries the bounds):
// a middle field outside its bounds drops the row, a longer first field moves the start to run end - max, a longer
// last field would truncate the match and let FindAll resume inside the run: the tile hands the scan over.
template <int NCLS, bool SETS, bool CAP, bool DENSE, int ALTK, bool BND>
__global__ __launch_bounds__(kThreads, ((SETS || CAP || BND) ? CXG_CAP_WAVES : (NCLS >= 3 ? 7 : CXG_CHAIN_WAVES))) void k_scan_chain_wave(ScanArgs a) {
  __shared__ __attribute__((aligned(16))) uint64_t s_cls[kWavesPerBlock][NCLS][64];   // forward class bitmaps
  __shared__ __attribute__((aligned(16))) uint64_t s_x[kWavesPerBlock][64];       // starts, reversed -> forward
  __shared__ uint16_t s_rs[kWavesPerBlock][kWRows];               // rows of the group, per wave: start / end inside their wave-tile
  __shared__ uint16_t s_re[kWavesPerBlock][kWRows];
  extern __shared__ __attribute__((aligned(16))) uint16_t s_rb[];   // CAP: [nruns][kWavesPerBlock][kWRows] ends of the captured runs
  __shared__ uint32_t s_cnt[kWavesPerBlock][kTilesPerWave];
  __shared__ uint32_t s_qbase[kWavesPerBlock * kTilesPerWave + 1];
  __shared__ uint64_t s_group;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: per-wave LDS bases live in SGPRs
  int lane = lane0;
  if (tid == 0) s_group = claim_group(a.static_groups != 0, a.ticket, a.ngroups);
  static_assert(sizeof(ChainAux) <= sizeof(a.chain), "ScanArgs::chain too small");
  static_assert(sizeof(ChainCaps) <= sizeof(a.caps), "ScanArgs::caps too small");
  const ChainCaps* gcp = reinterpret_cast<const ChainCaps*>(a.caps);   // kernel argument segment: scalar loads
  uint32_t cap_op[kCapMaxRuns];
#pragma unroll
  for (int x = 0; x < kCapMaxRuns; x++) cap_op[x] = (CAP || BND) ? gcp->run_op[x] : 0xFFu;
  const uint32_t bnd_nruns = BND ? gcp->nruns : 0u;                // fields that end at a compacted run end (all but the last)
  const ChainAux* gch = reinterpret_cast<const ChainAux*>(a.chain);   // kernel argument segment: scalar loads
  ChainRegs<NCLS, SETS> ch;
  ch.aux = gch;
  ch.nops = gch->nops;
  const bool restart_check = __builtin_amdgcn_readfirstlane(static_cast<int>(gch->restart_check)) != 0;
  constexpr int tpw = DENSE ? kDenseTilesPerWave : kTilesPerWave;
  ch.op_is_run = gch->run_bits; ch.op_cls2 = gch->cls2_lo; ch.op_cls2_hi = gch->cls2_hi;
#pragma unroll
  for (int c = 0; c < NCLS; c++) { ch.kind[c] = gch->cls_kind[c]; ch.lo[c] = gch->cls_lo[c]; ch.hi[c] = gch->cls_hi[c]; }
  // the blob is read with vector loads (its address comes out of a loaded header field): make the description
  // provably wave-uniform so that everything derived from it stays on the scalar unit
  ch.nops = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ch.nops)));
  auto uniform64 = [](uint64_t v) -> uint64_t {
    return (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32)))) << 32) |
           static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
  };
  ch.op_is_run = uniform64(ch.op_is_run);
  ch.op_cls2 = uniform64(ch.op_cls2);
  ch.op_cls2_hi = uniform64(ch.op_cls2_hi);
#pragma unroll
  for (int c = 0; c < NCLS; c++) {
    ch.kind[c] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ch.kind[c])));
    ch.lo[c] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ch.lo[c])));
    ch.hi[c] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ch.hi[c])));
  }
  __syncthreads();
  const uint64_t group = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_group >> 32))) << 32) |
                         static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_group)));
  if (group >= a.ngroups) return;
  if (limit_reached_skip(a, group, &s_base)) return;                 // FindAll with n > 0 (block_common.hpp)
  if (ALTK > 0) {                                                  // compile-time shape: everything derived from these folds
    ch.nops = 2u * ALTK - 1u;
    ch.op_is_run = 0x55555555ull & ((1ull << (2 * ALTK - 1)) - 1ull);   // steps 0, 2, 4, ... are runs
    if (NCLS == 2) ch.op_cls2 = 0x44444444ull & ((1ull << (2 * (2 * ALTK - 1))) - 1ull);   // step k uses class k & 1
    else ch.op_cls2 &= 0xCCCCCCCCull;                              // runs: class 0; separators: as described
    ch.op_cls2_hi = 0;
  }
  const uint32_t nops = ch.nops;
  const bool lead_run = ch.is_run(0);
  const uint32_t lead_cls = ch.cls(0);
  uint32_t nrows_w = 0;                                            // wave-uniform
  uint32_t fallback = 0;

  // Window loads go through a buffer resource sized to the bytes that exist (rounded up to a dword): lanes past
  // the end of the input read zeros, so there is no tail path and no branch around a load.  The up to 3 bytes
  // between len and the dword boundary are masked out of the bitmaps below (`stage` test).
  u32x4 x[4];
  uint32_t xprev = 0;                                              // the dword that ends with the byte in front of the tile
  auto issue_loads = [&](int jj) {
    const uint64_t wtn = group * (kWavesPerBlock * tpw) + static_cast<uint64_t>(jj) * kWavesPerBlock + wave;
    const uint64_t lo = wtn * static_cast<uint64_t>(kWaveTile);
    int nrec = 0;
    if (jj < tpw && lo < a.len) {
      const uint64_t rem = a.len - lo;
      nrec = rem >= static_cast<uint64_t>(kWaveTile + kWaveHalo) ? kWaveTile + kWaveHalo : static_cast<int>((rem + 3) & ~3ull);
    }
    const int pre = (nrec && lo) ? 16 : 0;                          // the resource starts 16 bytes early (keeps alignment)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.hay) + (nrec ? lo - pre : 0), 0, nrec + pre, 0x00020000);
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (lane + 64 * k) << 4, pre, 0);
    xprev = __builtin_amdgcn_raw_buffer_load_b32(rsrc, 0, pre ? 12 : nrec + pre, 0);   // out of range (zero) at the haystack start
  };
  if (CXG_CHAIN_PREFETCH) issue_loads(0);

#if CXG_CHAIN_PROF
  uint64_t pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tlast = __builtin_readcyclecounter();
#endif
  for (int j = 0; j < tpw; j++) {
    // Opaque copy of the lane id per wave-tile: lane-derived masks and LDS addresses are recomputed (a few ALU
    // ops) instead of being hoisted out of the loop and spilled — a scratch reload waits on vmcnt, which would
    // also drain the prefetched tile.
    lane = lane0;
    asm volatile("" : "+v"(lane));
    if (!CXG_CHAIN_PREFETCH) issue_loads(j);
    const uint64_t wt = group * (kWavesPerBlock * tpw) + static_cast<uint64_t>(j) * kWavesPerBlock + wave;
    const uint64_t tile_lo = wt * static_cast<uint64_t>(kWaveTile);
    uint32_t emitted_here = 0;
    if (tile_lo < a.len) {
      const uint64_t remaining = a.len - tile_lo;
      const int32_t rend = remaining > 0x7FFF0000ull ? 0x7FFF0000 : static_cast<int32_t>(remaining);
      const int32_t stage = rend < kWaveTile + kWaveHalo ? rend : kWaveTile + kWaveHalo;

      // ---- A: class masks of the vectors loaded one iteration ago, transposed through the wave's LDS scratch
#pragma unroll
      for (int c = 0; c < NCLS; c++) {
        const uint32_t kind = ch.kind[c], lo = ch.lo[c], hi = ch.hi[c];
        uint16_t* pieces = reinterpret_cast<uint16_t*>(s_cls[wave][c]);
        if (CXG_ABL == 4) {
#pragma unroll
          for (int q = 0; q < 4; q++) pieces[lane + 64 * q] = static_cast<uint16_t>(x[q].x ^ x[q].y ^ x[q].z ^ x[q].w);
        } else if (kind == kClsDigit) classify_tile<kClsDigit>(x, lo, hi, lane, pieces);
        else if (kind == kClsByte) classify_tile<kClsByte>(x, lo, hi, lane, pieces);
        else if (SETS && kind == kClsSet) classify_tile_set(x, ch.ranges(c), lane, pieces);
        else classify_tile<kClsRange>(x, lo, hi, lane, pieces);
      }
      PHASE_MARK(0);                                                // A: wait for the window + class masks
      const uint32_t xprev_cur = xprev;                             // arrived with x[] (same vmcnt)
      if (CXG_CHAIN_PREFETCH) issue_loads(j + 1);                   // x[] is free from here on
      wave_lds_sync();
      Words4 F{0, 0, 0, 0}, R{0, 0, 0, 0};                          // forward / reversed words
      F.w0 = s_cls[wave][0][lane]; R.w0 = brev64(s_cls[wave][0][63 - lane]);
      if (NCLS > 1) { F.w1 = s_cls[wave][NCLS > 1 ? 1 : 0][lane]; R.w1 = brev64(s_cls[wave][NCLS > 1 ? 1 : 0][63 - lane]); }
      if (NCLS > 2) { F.w2 = s_cls[wave][NCLS > 2 ? 2 : 0][lane]; R.w2 = brev64(s_cls[wave][NCLS > 2 ? 2 : 0][63 - lane]); }
      if (NCLS > 3) { F.w3 = s_cls[wave][NCLS > 3 ? 3 : 0][lane]; R.w3 = brev64(s_cls[wave][NCLS > 3 ? 3 : 0][63 - lane]); }
      if (stage != kWaveTile + kWaveHalo) {                         // short last window: nothing past the data is in a class
        const int32_t nf = stage - 64 * lane, nr = stage - 64 * (63 - lane);
        const uint64_t vf = nf <= 0 ? 0ull : (nf >= 64 ? ~0ull : ((1ull << nf) - 1ull));
        const uint64_t vr = brev64(nr <= 0 ? 0ull : (nr >= 64 ? ~0ull : ((1ull << nr) - 1ull)));
        F.and_all(vf); R.and_all(vr);
      }
      const uint64_t U = F.w0 | F.w1 | F.w2 | F.w3;                 // class union

      PHASE_MARK(1);                                                // loads issued, LDS transpose, words read
      // the byte in front of the tile, once, through the scalar cache
      bool prev_in_alphabet = false, prev_in_lead = false;
      if (tile_lo > 0) {
        const uint32_t pb = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(xprev_cur))) >> 24;
        prev_in_alphabet = ch.in_alphabet(pb);
#pragma unroll
        for (int c = 0; c < NCLS; c++) prev_in_lead = prev_in_lead || (lead_cls == static_cast<uint32_t>(c) && ch.has(c, pb));
      }
      // ---- O: ownership bounds from the synchronising bytes (complement of the class union, inside the data)
      int32_t zA = -1, zB = kFar;
      {
        uint64_t Z = ~U;
        if (stage != kWaveTile + kWaveHalo) {                       // short last window: bytes past the data are not synchronising
          const int32_t nv = stage - 64 * lane;
          Z &= nv <= 0 ? 0ull : (nv >= 64 ? ~0ull : ((1ull << nv) - 1ull));
        }
        if (tile_lo > 0) {
          if (prev_in_alphabet) {                                           // the segment at the tile's first byte began earlier
            const unsigned long long bz = __ballot(Z != 0ull);
            if (bz) { const int L = __builtin_ctzll(bz); zA = 64 * L + static_cast<int32_t>(__builtin_ctzll(readlane64(Z, L))); }
            else zA = kFar;
          }
        }
        const uint64_t Zb = Z & word_range(lane, kWaveTile - 1, kWaveTile + kWaveHalo - 1);
        const unsigned long long bzb = __ballot(Zb != 0ull);
        if (bzb) { const int L = __builtin_ctzll(bzb); zB = 64 * L + static_cast<int32_t>(__builtin_ctzll(readlane64(Zb, L))); }
        else if (stage != rend) { zB = -2; fallback |= 1; }         // no synchronising byte in the halo: ownership unknown
      }

      PHASE_MARK(2);                                                // previous byte + ownership bounds
      // ---- B: chain, right to left, on the reversed words
      uint64_t G = ~0ull;
      const bool at_eoi_edge = (stage == rend) && (stage == kWaveTile + kWaveHalo);   // byte 4095 is the last of the input
      auto bwd_step = [&](const int k) {
        const uint32_t ci = ch.cls(static_cast<uint32_t>(k));
        const uint64_t Ck = R.pick<NCLS>(ci);
        const uint64_t inject = (at_eoi_edge && k == static_cast<int>(nops) - 1) ? 1ull : 0ull;   // G_{n+1} holds at end of input
        if (!ch.is_run(static_cast<uint32_t>(k))) {
          uint64_t low = from_lower64(G) >> 63;                     // DPP outside any lane-dependent branch: a
          if (lane == 0) low = inject;                              // disabled source lane would not be read
          G = Ck & ((G << 1) | low);
        } else {
          // T: positions outside the class from which the rest of the chain holds (the byte after a run).  Adding T
          // to (Ck | T) turns every T position into a carry that ripples up through the class run above it and
          // clears it: one multiword addition, no shift and no neighbour move.  The run of the window's last byte
          // at the end of input gets its carry from the virtual position behind it (inject).
          const uint64_t T = G & ~Ck;
          const uint64_t s1 = (Ck | T) + T;
          const unsigned long long GG = __builtin_amdgcn_uicmpl(s1, T, 36 /*ult*/);
          const unsigned long long PP = __builtin_amdgcn_uicmpl(s1, ~0ull, 32 /*eq*/);
          const unsigned long long G1 = (GG << 1) | static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(inject)));
          const unsigned long long recv = (PP + G1) ^ PP;           // lanes that receive a carry
          G = Ck & ~add_carry_mask(s1, recv);
        }
      };
      if constexpr (ALTK > 0) {
#pragma unroll
        for (int k = 2 * ALTK - 2; k >= 0; k--) bwd_step(k);
      } else {
        for (int k = (CXG_ABL == 1 || CXG_ABL == 4) ? -1 : static_cast<int>(nops) - 1; k >= 0; k--) bwd_step(k);
      }
      PHASE_MARK(3);                                                // backward chain
      // starts, reversed orientation: with a leading run only the first byte of the run is a candidate
      uint64_t surv = G;
      if (lead_run) {
        const uint64_t D = R.pick<NCLS>(lead_cls);
        uint64_t dup = from_upper64(D);
        if (lane == 63) dup = prev_in_lead ? 1ull : 0ull;
        surv = D & ~((D >> 1) | (dup << 63)) & G;
      }
      if (CXG_ABL == 1 || CXG_ABL == 2 || CXG_ABL == 4) surv = 0;
      if (__ballot(surv != 0ull) != 0ull) {
        // ---- to forward orientation, restricted to the owned range (zA, zB]
        s_x[wave][lane] = surv;                                     // the kernel is VALU-bound: two LDS instructions beat the
        wave_lds_sync();                                            // ~20 VALU of a register lane reversal (lane_reverse32)
        const uint64_t S = brev64(s_x[wave][63 - lane]) & word_range(lane, zA + 1, zB);
        PHASE_MARK(4);                                              // starts, moved to forward orientation
        // ---- F: chain left to right on the forward words
        uint64_t M = S;
        uint32_t capcnt[kCapMaxRuns] = {0, 0, 0, 0};                  // CAP: run ends compacted (wave-uniform)
        const bool fixed_len = ch.op_is_run == 0ull;                   // a literal: every match is nops bytes long
        if (fixed_len) {                                            // ends = starts shifted by the length (< 64)
          const uint64_t lower = from_lower64(S);
          M = (S << nops) | ((lane == 0) ? 0ull : (lower >> (64u - nops)));
        }
        auto fwd_step = [&](const uint32_t k) {
          if (!ch.is_run(static_cast<uint32_t>(k))) {
            uint64_t low = from_lower64(M) >> 63;
            if (lane == 0) low = 0ull;
            M = (M << 1) | low;
          } else {
            const uint64_t Ck = F.pick<NCLS>(ch.cls(static_cast<uint32_t>(k)));
            const uint64_t s1 = Ck + M;
            const unsigned long long GG = __builtin_amdgcn_uicmpl(s1, M, 36 /*ult*/);
            const unsigned long long PP = __builtin_amdgcn_uicmpl(s1, ~0ull, 32 /*eq*/);
            const unsigned long long recv = (PP + (GG << 1)) ^ PP;
            M = add_carry_mask(s1, recv) & ~Ck;
            if (CAP || BND) {
#pragma unroll
              for (int x = 0; x < kCapMaxRuns; x++) {
                if (k != cap_op[x]) continue;                          // a captured run: its ends, in match order
                const uint32_t nm = static_cast<uint32_t>(__popcll(M));
                const uint32_t incm = wave_inclusive_sum(nm);
                capcnt[x] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incm), 63));
                uint32_t im = incm - nm;
                uint64_t mb = M;
                while (mb) {
                  const int bit = __builtin_ctzll(mb);
                  mb &= mb - 1;
                  const uint32_t r = nrows_w + im;
                  if (r < static_cast<uint32_t>(kWRows)) s_rb[(x * kWavesPerBlock + wave) * kWRows + r] = static_cast<uint16_t>(64 * lane + bit);
                  im++;
                }
              }
            }
          }
        };
        if constexpr (ALTK > 0) {
#pragma unroll
          for (uint32_t k = 0; k < static_cast<uint32_t>(2 * ALTK - 1); k++) fwd_step(k);
        } else {
          for (uint32_t k = 0; k < (fixed_len ? 0u : nops); k++) fwd_step(k);
        }
        if (restart_check && lead_run) {                            // a match that ends inside a run of the first class: FindAll would
          const uint64_t A = F.pick<NCLS>(lead_cls);                 // resume there, mid-run — not a run start.  Rare; the table kernel takes over.
          uint64_t lowA = from_lower64(A) >> 63;
          if (lane == 0) lowA = 0ull;
          if (__ballot((M & A & ((A << 1) | lowA)) != 0ull) != 0ull) fallback |= 64;
        }
        PHASE_MARK(5);                                              // forward chain
        // ---- P: ranks of starts and ends, (start, end) pairs
        const uint32_t ns = static_cast<uint32_t>(__popcll(S)), ne = static_cast<uint32_t>(__popcll(M));
        const uint32_t packed = ns | (ne << 16);
        const uint32_t incl = wave_inclusive_sum(packed);
        const uint32_t tot = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
        uint32_t n = CXG_ABL == 3 ? 0u : (tot & 0xFFFFu);
        // A marker that left the window is a match ending exactly at byte 4096: possible only when that is the
        // end of the input, and then only for the last match.  Anything else breaks the pairing and falls back.
        const uint32_t cout = (at_eoi_edge && (tot & 0xFFFFu) == (tot >> 16) + 1u) ? 1u : 0u;
        const uint32_t n_ends = (tot >> 16) + cout;
        if (n != n_ends && CXG_ABL != 3) {                          // pairing invariant violated
          if (a.prof && lane == 0 && atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 7), 1ull) == 0ull) {
            a.prof[0] = tile_lo; a.prof[1] = n; a.prof[2] = n_ends; a.prof[3] = cout; a.prof[4] = static_cast<uint64_t>(static_cast<int64_t>(zA));
            a.prof[5] = static_cast<uint64_t>(static_cast<int64_t>(zB)); a.prof[6] = static_cast<uint64_t>(stage);
          }
          fallback |= 4; n = 0;
        }
        if ((CAP || BND) && n != 0) {
#pragma unroll
          for (int x = 0; x < kCapMaxRuns; x++) if (cap_op[x] != 0xFFu && capcnt[x] != n) fallback |= 32;   // a marker merged or left the window
          if (fallback & 32) n = 0;
        }
        // Both compactions keep the order, so start k and end k land in the same row without ever meeting in
        // a register.  FindAll skips a match that begins inside the previous emitted one (findall.go:267-275):
        // that shows as a start whose rank differs from the number of ends at or before it.
        uint32_t ov = 0;
        if (n) {
          uint32_t is = (incl & 0xFFFFu) - ns, ie = (incl >> 16) - ne;
          const uint32_t ie0 = ie;
          uint64_t sb = S, eb = M;
          while (sb) {
            const int bit = __builtin_ctzll(sb);
            sb &= sb - 1;
            const uint32_t r = nrows_w + is;
            if (r < static_cast<uint32_t>(kWRows)) {
              s_rs[wave][r] = static_cast<uint16_t>(64 * lane + bit);
              if (fixed_len) s_re[wave][r] = static_cast<uint16_t>(64 * lane + bit + nops);
            }
            const uint32_t ends_le = ie0 + static_cast<uint32_t>(__popcll(M & ((2ull << bit) - 1ull)));
            ov |= (ends_le != is) ? 1u : 0u;
            is++;
          }
          while (eb && !fixed_len) {
            const int bit = __builtin_ctzll(eb);
            eb &= eb - 1;
            const uint32_t r = nrows_w + ie;
            if (r < static_cast<uint32_t>(kWRows)) s_re[wave][r] = static_cast<uint16_t>(64 * lane + bit);
            ie++;
          }
          if (cout && !fixed_len && lane == 0 && nrows_w + (tot >> 16) < static_cast<uint32_t>(kWRows)) s_re[wave][nrows_w + (tot >> 16)] = static_cast<uint16_t>(kWaveTile + kWaveHalo);
        }
        emitted_here = n;
        if (BND) {
          ov = 0;                                                   // overlaps are judged on the filtered rows below
          if (n != 0 && nrows_w + n <= static_cast<uint32_t>(kWRows)) {
            wave_lds_sync();
            uint32_t kept = 0, trunc = 0;                           // kept: wave-uniform
            for (uint32_t r0 = 0; r0 < n; r0 += 64) {
              const uint32_t q = r0 + static_cast<uint32_t>(lane);
              const bool have = q < n;
              const uint32_t r = nrows_w + (have ? q : 0u);
              int32_t s = s_rs[wave][r];
              const int32_t e = s_re[wave][r];
              bool valid = have;
              int32_t field_lo = s;
#pragma unroll
              for (int x = 0; x < kCapMaxRuns; x++) {
                if (static_cast<uint32_t>(x) >= bnd_nruns) continue;
                const int32_t re = s_rb[(x * kWavesPerBlock + wave) * kWRows + r];
                const int32_t mn = gcp->src[x], mx = gcp->src[8 + x];      // scalar loads (kernel arguments)
                int32_t len = re - field_lo;
                if (x == 0 && mx != 0 && len > mx) { s = re - mx; len = mx; }   // the match starts inside the first run
                valid = valid && len >= mn && (mx == 0 || len <= mx);
                field_lo = re + 1;                                  // past the single-byte separator
              }
              {
                const int32_t mn = gcp->src[bnd_nruns], mx = gcp->src[8 + bnd_nruns];
                const int32_t len = e - field_lo;
                if (have && mx != 0 && len > mx) trunc = 1;         // FindAll would resume inside this run
                valid = valid && len >= mn;
              }
              const unsigned long long vm = __ballot(valid);
              if (valid) {                                          // in place: the target row is never behind the rows still to read
                const uint32_t w = nrows_w + kept + static_cast<uint32_t>(__popcll(vm & ((1ull << lane) - 1ull)));
                s_rs[wave][w] = static_cast<uint16_t>(s);
                s_re[wave][w] = static_cast<uint16_t>(e);
              }
              kept += static_cast<uint32_t>(__popcll(vm));
              wave_lds_sync();
            }
            if (__ballot(trunc != 0) != 0ull) fallback |= 64;
            uint32_t ovr = 0;
            for (uint32_t r0 = 0; r0 < kept; r0 += 64) {
              const uint32_t q = r0 + static_cast<uint32_t>(lane);
              if (q > 0 && q < kept) ovr |= (s_rs[wave][nrows_w + q] < s_re[wave][nrows_w + q - 1]) ? 1u : 0u;
            }
            n = kept;
            emitted_here = kept;
            ov = ovr;
          }
        }
        if (CAP && __ballot(ov != 0) != 0ull) fallback |= 32;        // dropped rows would have to drop their run ends too: two-kernel path
        if (__ballot(ov != 0) != 0ull && nrows_w + n <= static_cast<uint32_t>(kWRows)) {   // rare: resolve serially, in place
          wave_lds_sync();
          uint32_t kept = 0;
          if (lane == 0) {
            int32_t cur_end = -1;
            for (uint32_t q = 0; q < n; q++) {
              const uint16_t sq = s_rs[wave][nrows_w + q], eq = s_re[wave][nrows_w + q];
              if (static_cast<int32_t>(sq) >= cur_end) { s_rs[wave][nrows_w + kept] = sq; s_re[wave][nrows_w + kept] = eq; kept++; cur_end = eq; }
            }
          }
          emitted_here = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(kept)));
          wave_lds_sync();
        }
      }
    }
    PHASE_MARK(6);                                                  // ranks + rows
    if (lane == 0) s_cnt[wave][j] = emitted_here;
    nrows_w += emitted_here;
  }
#if CXG_CHAIN_PROF
  if (a.prof && lane == 0) {
    for (int i = 0; i < 7; i++) atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 8 + i), static_cast<unsigned long long>(pacc[i]));
    atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 15), 1ull);
  }
#endif
  if (nrows_w > static_cast<uint32_t>(kWRows)) fallback |= 16;
  if (fallback != 0 && lane == 0) raise_err(a.err, 8u | (fallback << 8));   // bits 8.. = reason (diagnostics, CXG_VERBOSE)
  if (a.max_len != 0) {                                             // UseBoth programs only (uniform): no match may exceed the restart span
    wave_lds_sync();
    bool long_hit = false;
    for (uint32_t r = lane0; r < nrows_w && r < static_cast<uint32_t>(kWRows); r += 64)
      long_hit = long_hit || (static_cast<uint32_t>(s_re[wave][r]) - static_cast<uint32_t>(s_rs[wave][r]) > a.max_len);
    if (__ballot(long_hit) != 0ull && lane0 == 0) raise_err(a.err, kErrLongMatch);
  }
  __syncthreads();

  // ---- order the group's rows: wave-tile q = j*4 + wave; exclusive prefix over q
  if (tid < 64) {
    const int q = tid;
    const uint32_t v = (q < kWavesPerBlock * tpw) ? s_cnt[q % kWavesPerBlock][q / kWavesPerBlock] : 0u;
    const uint32_t incl = wave_inclusive_sum(v);
    if (q < kWavesPerBlock * tpw) s_qbase[q] = incl - v;
    if (q == kWavesPerBlock * tpw - 1) s_qbase[kWavesPerBlock * tpw] = incl;
  }
  __syncthreads();
  const uint32_t total = s_qbase[kWavesPerBlock * tpw];
  tile_lookback(a.status, a.total, a.err, group, a.ngroups, total, &s_base, a.epoch, a.limit, a.stop);
  if (a.out == nullptr) return;
  const uint64_t base = s_base;
  const int64_t origin = a.base + static_cast<int64_t>(group * static_cast<uint64_t>(kWaveTile) * kWavesPerBlock * tpw);
  uint32_t start = 0;
  if (CAP) {
    // Rows of 2 * groups int64 (64 B for three groups): PARTS lanes share a row, each writes one 16-byte slot pair, so a
    // store instruction covers 64 / PARTS whole rows = 1 KiB of contiguous output (one lane per row wrote 64 pieces of 16 B
    // at a 64-byte stride per instruction: four times the requests into L2 for the same bytes).
    const uint32_t parts = a.row_width >> 1;                          // 2..8, uniform
    const uint32_t inv = (65536u + parts - 1u) / parts;               // i / parts == (i * inv) >> 16 for i < 4096
    for (int j = 0; j < tpw; j++) {
      const uint32_t n = s_cnt[wave][j];
      const uint64_t dst = base + s_qbase[j * kWavesPerBlock + wave];
      const int64_t tb = origin + static_cast<int64_t>(j * kWavesPerBlock + wave) * kWaveTile;
      for (uint32_t i = lane; i < n * parts; i += 64) {
        const uint32_t ri = (i * inv) >> 16, part = i - ri * parts;
        const uint32_t r = start + ri;
        if (r < static_cast<uint32_t>(kWRows) && dst + ri < a.cap) {
          const int64_t ms = tb + s_rs[wave][r], me = tb + s_re[wave][r];
          uint32_t sc0 = gcp->src[0], sc1 = gcp->src[1];
          int32_t of0 = gcp->off[0], of1 = gcp->off[1];
#pragma unroll
          for (uint32_t q = 1; q < 8u; q++)                            // the lane's slot pair: uniform table, per-lane selects
            if (part == q) { sc0 = gcp->src[2 * q]; sc1 = gcp->src[2 * q + 1]; of0 = gcp->off[2 * q]; of1 = gcp->off[2 * q + 1]; }
          auto slot = [&](uint32_t sc, int32_t of) -> int64_t {
            if (sc == kCapSrcUnset) return -1;
            const int64_t ps = sc == kCapSrcStart ? ms : sc == kCapSrcEnd ? me : tb + s_rb[((sc - kCapSrcRun0) * kWavesPerBlock + wave) * kWRows + r];
            return ps + of;
          };
          longlong2 o;
          if (part == 0) { o.x = ms; o.y = me; }
          else { o.x = slot(sc0, of0); o.y = slot(sc1, of1); }
          store_pair_nt(a.out + (dst + ri) * a.row_width + 2u * part, o.x, o.y);
        }
      }
      start += n;
    }
    return;
  }
  for (int j = 0; j < tpw; j++) {
    const uint32_t n = s_cnt[wave][j];
    const uint64_t dst = base + s_qbase[j * kWavesPerBlock + wave];
    for (uint32_t i = lane; i < n; i += 64) {
      const uint32_t r = start + i;
      if (r < static_cast<uint32_t>(kWRows) && dst + i < a.cap) {
        const int64_t tb = origin + static_cast<int64_t>(j * kWavesPerBlock + wave) * kWaveTile;
        longlong2 v; v.x = tb + s_rs[wave][r]; v.y = tb + s_re[wave][r];
        store_pair_nt(a.out + (dst + i) * a.row_width, v.x, v.y);   // row_width > 2 without CAP: a capture pass fills the rest
      }
    }
    start += n;
  }
}

namespace {
template <int NCLS, bool SETS>
void launch_chain(const ScanArgs& a, bool caps, bool dense, dim3 grid, dim3 block, hipStream_t stream) {
  const size_t dyn = caps ? static_cast<size_t>(reinterpret_cast<const ChainCaps*>(a.caps)->nruns) * kWavesPerBlock * kWRows * sizeof(uint16_t) : 0;
  if (caps) {
    if (dense) hipLaunchKernelGGL((k_scan_chain_wave<NCLS, SETS, true, true, 0, false>), grid, block, dyn, stream, a);
    else hipLaunchKernelGGL((k_scan_chain_wave<NCLS, SETS, true, false, 0, false>), grid, block, dyn, stream, a);
  } else {
    if (dense) hipLaunchKernelGGL((k_scan_chain_wave<NCLS, SETS, false, true, 0, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_scan_chain_wave<NCLS, SETS, false, false, 0, false>), grid, block, 0, stream, a);
  }
}
}  // namespace

// Number of runs of a chain run(0) (byte(sep) run(0))* — every run of class 0, single-byte separators — else 0.
int alternating_runs(const ChainAux& c) {
  if (c.ncls < 2 || (c.nops & 1u) == 0 || c.nops < 3 || c.nops > 7) return 0;
  for (uint32_t k = 0; k < c.nops; k++) {
    if (c.op_kind[k] != ((k & 1u) ? kChainByte : kChainRun)) return 0;
    if (!(k & 1u) && c.op_cls[k] != 0) return 0;
    if ((k & 1u) && (c.op_cls[k] == 0 || (c.ncls == 2 && c.op_cls[k] != 1))) return 0;
  }
  return static_cast<int>((c.nops + 1) / 2);
}

hipError_t launch_scan_chain_wave(const ScanArgs& a, uint32_t ncls, bool sets, bool caps, hipStream_t stream) {
  const dim3 grid(static_cast<unsigned>(a.ngroups)), block(kThreads);
  const bool dense = a.tiles_per_wave == static_cast<uint32_t>(kDenseTilesPerWave);
  static const bool altOk = getenv("CXG_NO_SHAPE_KERNELS") == nullptr;
  const int altk = (altOk && !dense) ? alternating_runs(*reinterpret_cast<const ChainAux*>(a.chain)) : 0;
  if (reinterpret_cast<const ChainCaps*>(a.caps)->on == 2) {         // bounded repetition: only the unrolled two-class shapes
    if (ncls != 2 || sets || caps || dense) return hipErrorInvalidValue;
    const int k = alternating_runs(*reinterpret_cast<const ChainAux*>(a.chain));
    const size_t dyn = static_cast<size_t>(reinterpret_cast<const ChainCaps*>(a.caps)->nruns) * kWavesPerBlock * kWRows * sizeof(uint16_t);
    switch (k) {
      case 2: hipLaunchKernelGGL((k_scan_chain_wave<2, false, false, false, 2, true>), grid, block, dyn, stream, a); return hipGetLastError();
      case 3: hipLaunchKernelGGL((k_scan_chain_wave<2, false, false, false, 3, true>), grid, block, dyn, stream, a); return hipGetLastError();
      case 4: hipLaunchKernelGGL((k_scan_chain_wave<2, false, false, false, 4, true>), grid, block, dyn, stream, a); return hipGetLastError();
      default: return hipErrorInvalidValue;
    }
  }
  if (altk && ncls == 2 && !sets && !caps) {                          // unrolled instantiations for the alternating shapes
    switch (altk) {
      case 2: hipLaunchKernelGGL((k_scan_chain_wave<2, false, false, false, 2, false>), grid, block, 0, stream, a); return hipGetLastError();
      case 3: hipLaunchKernelGGL((k_scan_chain_wave<2, false, false, false, 3, false>), grid, block, 0, stream, a); return hipGetLastError();
      case 4: hipLaunchKernelGGL((k_scan_chain_wave<2, false, false, false, 4, false>), grid, block, 0, stream, a); return hipGetLastError();
      default: break;
    }
  }
  if (altk == 3 && ncls == 3) {                                       // field@field.field: three fields, two different separators
    const size_t dyn = caps ? static_cast<size_t>(reinterpret_cast<const ChainCaps*>(a.caps)->nruns) * kWavesPerBlock * kWRows * sizeof(uint16_t) : 0;
    if (sets && caps) hipLaunchKernelGGL((k_scan_chain_wave<3, true, true, false, 3, false>), grid, block, dyn, stream, a);
    else if (sets) hipLaunchKernelGGL((k_scan_chain_wave<3, true, false, false, 3, false>), grid, block, 0, stream, a);
    else if (caps) hipLaunchKernelGGL((k_scan_chain_wave<3, false, true, false, 3, false>), grid, block, dyn, stream, a);
    else hipLaunchKernelGGL((k_scan_chain_wave<3, false, false, false, 3, false>), grid, block, 0, stream, a);
    return hipGetLastError();
  }
  switch (ncls * 2 + (sets ? 1 : 0)) {
    case 2: launch_chain<1, false>(a, caps, dense, grid, block, stream); break;
    case 3: launch_chain<1, true>(a, caps, dense, grid, block, stream); break;
    case 4: launch_chain<2, false>(a, caps, dense, grid, block, stream); break;
    case 5: launch_chain<2, true>(a, caps, dense, grid, block, stream); break;
    case 6: launch_chain<3, false>(a, caps, dense, grid, block, stream); break;
    case 7: launch_chain<3, true>(a, caps, dense, grid, block, stream); break;
    case 8: launch_chain<4, false>(a, caps, dense, grid, block, stream); break;
    case 9: launch_chain<4, true>(a, caps, dense, grid, block, stream); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace cxgdev
