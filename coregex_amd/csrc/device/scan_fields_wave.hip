// scan_fields_wave.hip — FindAll for "fields" programs: the complete ordered chain  run(F) (byte(S) run(F)){K-1}  with ONE
// field class F and ONE separator class S, F and S disjoint: `\d+\.\d+\.\d+\.\d+` (BASELINE configs[1], the headline),
// `\d+:\d+:\d+`, `\d+\.\d+`, `[a-f]+-[a-f]+`.  Seventh generation of the headline kernel (round 3): forward only.
//
// Reference semantics kept (meta/findall.go:176-283 over findIndicesDigitPrefilterAtWithState, meta/find_indices.go:1050-1088,
// resp. the DFA searches of dfa/lazy/lazy.go for UseDFA programs): leftmost-first, non-empty, next search from the match end.
//
// Why no backward pass is needed.  Call a separator byte with a field byte on BOTH sides a LINK, and a maximal stretch of
// field bytes and links a SUPER-RUN: f+ (s f+)*, n fields joined by n-1 links.  A match lies inside one super-run (all its
// separators are links) and consists of K consecutive fields.  Leftmost-first from the super-run's first byte takes fields
// 1..K; FindAll resumes at the end of field K, where a link follows (no match can start on it), so the next match is fields
// K+1..2K, and so on: FindAll over a super-run = its fields in groups of K, from its START.  Nothing outside the super-run
// matters, so the scan needs no synchronising byte and no ownership search: a wave-tile owns the super-runs that START in its
// 3840 bytes.  (scan_chain_wave.hip proved every start by a right-to-left chain first and searched the ownership bounds
// (zA, zB] in the synchronising bytes: 184 of its 515 VALU instructions per tile; this kernel has neither.)
//
// One wave64 per wave-tile; window = 64 bytes in front of the tile + 3840 + 192 behind = 4096 bytes = 64 bitmap words, word l
// in lane l, lanes 1..60 own.  Per tile:
//   A  class bitmaps D (field) and P (separator): SWAR compare + v_dot4_u32_u8 gather per dword, 16-bit pieces transposed
//      through the wave's LDS scratch.  Each 16-byte vector is classified as soon as it has arrived and its register is
//      refilled with the NEXT tile's vector at once: up to four loads per lane stay in flight through the whole tile.
//   B  L = P & (D << 1) & (D >> 1) (links), WS = D & ~(D << 1) & ~(L << 1) (super-run starts), restricted to the owned lanes.
//   C  markers M = WS hop over K fields: s = D + M (a multiword addition: the carry ripples through the field and lands behind
//      it), then K-1 times M' = s & L, s = (D | M') + M' (the link bit is added to itself: the carry enters the next field).
//      Carries between lanes: generate = the v_addc carry-out (an SGPR pair, no compare), propagate = words of 64 field
//      bytes, resolved on the scalar unit, fed back as the carry-in of a second v_addc.  E = s & ~D = the match ends.
//   D  an end that sits on a link (a super-run with more than K fields): the byte behind it starts the next group — rare
//      loop, same hop.  B = all group starts.
//   E  rows: per end bit, start = highest bit of B below it (this lane's word or the previous lane's), one packed
//      (start | end << 16) store into the wave's row buffer at its rank (DPP prefix sum over the lanes' end counts).
// A workgroup (4 waves x 8 tiles = 120 KiB) orders its rows after ONE barrier, looks back (block_common.hpp) and writes
// coalesced int64 pairs, as scan_chain_wave.hip does.
// Fallback flag (err bit 8; the host reruns the scan on scan_chain_wave.hip's dense mode resp. the transducer kernel): a
// super-run that reaches past its window (> 192 bytes behind its tile), a match longer than its start search (64..127
// bytes), row-buffer overflow (reason 0x10: match-dense input).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>

#include "block_common.hpp"
#include "scan_dfa.h"
#include "walk.hpp"
#include "wave_common.hpp"

#ifndef CXG_FIELDS_WAVES
#define CXG_FIELDS_WAVES 8
#endif
// -DCXG_FABL=n (experiments only, results WRONG): 1 = no rows, 2 = no chain and no rows, 3 = no class masks either,
// 4 = no LDS transpose either (the window is only read), 5 = 4 without barrier / look-back / epilogue
#ifndef CXG_FABL
#define CXG_FABL 0
#endif
// cache policy of the haystack loads (buffer intrinsic aux bits on gfx950: 1 = sc0, 2 = nt, 16 = sc1): -DCXG_HAY_LOAD_AUX=2 for A/B
#ifndef CXG_HAY_LOAD_AUX
#define CXG_HAY_LOAD_AUX 2                                   // nt: the haystack is read once (count-only 0.187 -> 0.176 ms, rows 0.234 -> 0.231; profiles/r04_time_wrapped.txt)
#endif

namespace cxgdev {

namespace {

constexpr int kFPre = 64;                                  // window bytes in front of the tile
constexpr int kFWin = kWaveTile + kWaveHalo;               // 4096
constexpr int kFRows = 64 * kTilesPerWave;                                // rows buffered per wave and group
constexpr unsigned long long kFOwn = 0x1FFFFFFFFFFFFFFEull;   // lanes 1..60 own their words (64 window bytes in front of the tile)
// The persistent kernel's window: 128 bytes in front of the tile, lanes 2..61 own, 128 bytes behind — every window starts and
// ends on a 128-byte line (tiles are 30 lines), and inside a unit the words 0..3 of a tile ARE the words 60..63 of the tile
// in front of it: they are carried over in LDS instead of being fetched again (round 5; round 4 read 33 lines per 30-line tile).
constexpr int kFPrePers = 128;
constexpr unsigned long long kFOwnPers = 0x3FFFFFFFFFFFFFFCull;

// 0x80 in every byte of x that IS in the class.  t = x & 0x7F7F7F7F is shared by the classes of a dword; (t ^ c) + k is one
// v_xad_u32, the final ~(u | x) & 0x80808080 one v_bitop3_b32: two instructions per class and dword behind the shared AND.
__device__ __forceinline__ uint32_t xad(uint32_t t, uint32_t c, uint32_t k) {   // (t ^ c) + k; c uniform (the one scalar operand), k in a register
  uint32_t r;
  asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(t), "s"(c), "v"(k));
  return r;
}
template <int KIND>
__device__ __forceinline__ uint32_t incls4(uint32_t x, uint32_t t, uint32_t lo4, uint32_t hi4) {   // lo4 / hi4: bounds splat over the bytes (hi4 = 0x7F - hi)
  if (KIND == kClsDigit) return ~(xad(t, 0x30303030u, 0x76767676u) | x) & 0x80808080u;
  if (KIND == kClsByte) return ~(xad(t, lo4, 0x7F7F7F7Fu) | x) & 0x80808080u;
  const uint32_t ge = (x | 0x80808080u) - lo4;
  const uint32_t gt = t + hi4;
  return ge & ~gt & ~x & 0x80808080u;
}
// 16 class bits of a 16-byte vector: the four flags of a dword are gathered by one v_dot4_u32_u8 (weights 1,2,4,8 resp.
// 16,32,64,128: 128 x the byte of flags accumulates over a dword pair).  Bits above 15 are garbage (ds_write_b16 drops them).
template <int KIND>
__device__ __forceinline__ uint32_t piece16(const u32x4& x, const u32x4& t, uint32_t lo4, uint32_t hi4) {
  const uint32_t lo = __builtin_amdgcn_udot4(incls4<KIND>(x.y, t.y, lo4, hi4), 0x80402010u, __builtin_amdgcn_udot4(incls4<KIND>(x.x, t.x, lo4, hi4), 0x08040201u, 0u, false), false);
  const uint32_t hi = __builtin_amdgcn_udot4(incls4<KIND>(x.w, t.w, lo4, hi4), 0x80402010u, __builtin_amdgcn_udot4(incls4<KIND>(x.z, t.z, lo4, hi4), 0x08040201u, 0u, false), false);
  return (lo >> 7) | (hi << 1);
}
// (a1:a0) + (b1:b0) -> (s1:s0), carry-out of the 64-bit addition of every lane as a wave mask (the v_addc's own carry
// output: no compare instruction).
__device__ __forceinline__ void add64_co(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t& s0, uint32_t& s1, unsigned long long& cout) {
  unsigned long long c0;
  asm("v_add_co_u32_e64 %0, %2, %4, %5\n\tv_addc_co_u32_e64 %1, %3, %6, %7, %2"
      : "=&v"(s0), "=&v"(s1), "=&s"(c0), "=&s"(cout)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}
// (s1:s0) += bit `lane` of mask (carry-in straight from the scalar mask)
__device__ __forceinline__ void add64_cin(uint32_t& s0, uint32_t& s1, unsigned long long mask) {
  uint32_t lo, hi;
  unsigned long long c;
  asm("v_addc_co_u32_e64 %0, %2, %3, 0, %5\n\tv_addc_co_u32_e64 %1, %2, %4, 0, %2"
      : "=&v"(lo), "=&v"(hi), "=&s"(c)
      : "v"(s0), "v"(s1), "s"(mask));
  s0 = lo; s1 = hi;
}
__device__ __forceinline__ uint32_t sel_lanes(uint32_t v, unsigned long long mask) {   // v in the lanes of mask, else 0
  uint32_t r;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(mask));
  return r;
}
__device__ __forceinline__ uint32_t ffbh_raw(uint32_t v) {   // leading zeros; 0xFFFFFFFF for v == 0 (the instruction's own convention)
  uint32_t r;
  asm("v_ffbh_u32_e32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ uint32_t dpp_from_lower_z(uint32_t v) {   // lane i <- lane i-1, lane 0 <- 0
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x138 /*wave_shr:1*/, 0xF, 0xF, true));
}
__device__ __forceinline__ uint32_t dpp_from_upper_ones(uint32_t v) {   // lane i <- lane i+1, lane 63 <- all ones
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(-1, static_cast<int>(v), 0x130 /*wave_shl:1*/, 0xF, 0xF, false));
}

// Inclusive prefix sum over the 64 lanes with the addition inside the DPP instruction (wave_common.hpp's version costs a
// v_mov_dpp + v_add per step: the compiler does not fuse them).  s_nop 1: two wait states between a VALU write and a DPP read.
__device__ __forceinline__ uint32_t wave_inclusive_sum_fused(uint32_t v) {
  asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0"
      : "+v"(v));
  return v;
}

// ---- the tile mathematics shared by both kernels ------------------------------------------------------------------------
struct FieldsTile { uint32_t e0, e1, b0, b1; bool ovf; };   // ends / group starts of the lane's word; ovf (uniform): a marker left the window

// Phases B-D for one window: (d1:d0) / (p1:p0) = field / separator bitmap word of this lane.
template <int K, unsigned long long OWN = kFOwn>
__device__ __forceinline__ FieldsTile fields_core(uint32_t d0, uint32_t d1, uint32_t p0, uint32_t p1) {
  // words of 64 field bytes pass a carry on (with no marker of their own; a word that generates needs no propagate)
  const unsigned long long PPd = __builtin_amdgcn_uicmpl((static_cast<uint64_t>(d1) << 32) | d0, ~0ull, 32 /*eq*/);
  // ---- B: links and super-run starts
  const uint32_t prev_d1 = dpp_from_lower(d1);                    // lane 0: its own word — that lane owns nothing
  const uint32_t next_d0 = dpp_from_upper_ones(d0);               // lane 63: "a field byte follows the window": a link there sends its marker out of the window (fallback)
  const uint32_t Dl0 = __builtin_amdgcn_alignbit(d0, prev_d1, 31), Dl1 = __builtin_amdgcn_alignbit(d1, d0, 31);   // D << 1
  const uint32_t Dr0 = __builtin_amdgcn_alignbit(d1, d0, 1), Dr1 = __builtin_amdgcn_alignbit(next_d0, d1, 1);     // D >> 1
  const uint32_t L0 = p0 & Dl0 & Dr0, L1 = p1 & Dl1 & Dr1;
  const uint32_t prev_l1 = dpp_from_lower(L1);
  const uint32_t Ll0 = __builtin_amdgcn_alignbit(L0, prev_l1, 31), Ll1 = __builtin_amdgcn_alignbit(L1, L0, 31);   // L << 1
  uint32_t b0 = sel_lanes(d0 & ~Dl0 & ~Ll0, OWN), b1 = sel_lanes(d1 & ~Dl1 & ~Ll1, OWN);   // B: group starts (first: the owned super-run starts)
  if (CXG_FABL >= 2) { b0 = 0; b1 = 0; }
  // ---- C: hop over K fields
  unsigned long long ovf = 0;                                     // bit 63: a marker left the window (scalar)
  auto carry_in = [&](unsigned long long GG) -> unsigned long long {
    const unsigned long long Pe = PPd & ~GG;
    const unsigned long long recv = (Pe + (GG << 1)) ^ Pe;        // lanes that receive a carry
    ovf |= GG | (Pe & recv);                                      // lane 63 generates, or passes one on
    return recv;
  };
  auto hop = [&](uint32_t m0, uint32_t m1, uint32_t& r0, uint32_t& r1) {
    uint32_t s0, s1;
    unsigned long long GG;
    add64_co(d0, d1, m0, m1, s0, s1, GG);
    add64_cin(s0, s1, carry_in(GG));
#pragma unroll
    for (int i = 1; i < K; i++) {
      const uint32_t q0 = s0 & L0, q1 = s1 & L1;                  // markers that stand on a link
      add64_co(d0 | q0, d1 | q1, q0, q1, s0, s1, GG);
      add64_cin(s0, s1, carry_in(GG));
    }
    r0 = s0; r1 = s1;
  };
  uint32_t r0, r1;
  hop(b0, b1, r0, r1);
  uint32_t e0 = r0 & ~d0, e1 = r1 & ~d1;                          // ends (exclusive) of the first group of every owned super-run
  uint32_t el0 = r0 & L0, el1 = r1 & L1;
  // ---- D: super-runs with more than K fields (`1.2.3.4.5.6.7.8`): the byte behind an end that sits on a link starts the next group
  while (__builtin_amdgcn_uicmpl((static_cast<uint64_t>(el1) << 32) | el0, 0ull, 33 /*ne*/) != 0ull) {
    const uint32_t prev_e1 = dpp_from_lower_z(el1);
    if ((static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(el1), 63)) >> 31) != 0u) ovf |= 1ull << 63;
    const uint32_t n0 = __builtin_amdgcn_alignbit(el0, prev_e1, 31), n1 = __builtin_amdgcn_alignbit(el1, el0, 31);
    b0 |= n0; b1 |= n1;
    hop(n0, n1, r0, r1);
    e0 |= r0 & ~d0; e1 |= r1 & ~d1;
    el0 = r0 & L0; el1 = r1 & L1;
  }
  return FieldsTile{e0, e1, b0, b1, (ovf >> 63) != 0ull};
}

// Phase E: one packed row (start | end << 16, window bit indices) per end bit of the lane's word, rows[index(r)], r counting
// up from r0.  Start of the match that ends at bit b: the highest bit of B below b — in this word, else in the previous
// lane's (a start further back: the row comes out with start >= end and is caught when the rows are written).
template <typename IndexFn>
__device__ __forceinline__ void fields_rows(const FieldsTile& t, int lane, uint32_t* rows, uint32_t r, IndexFn index, uint32_t shift = 0u) {   // shift: added to the packed row (same offset in both halves)
  const uint32_t pb0 = dpp_from_lower_z(t.b0), pb1 = dpp_from_lower_z(t.b1);
  const uint32_t lane64 = static_cast<uint32_t>(lane) << 6;
  {
    const uint32_t tp = min(ffbh_raw(pb1) | 32u, ffbh_raw(pb0) | 64u);
    uint32_t xx = t.e0;
    while (xx) {
      const uint32_t b = static_cast<uint32_t>(__builtin_ctz(xx));
      xx &= xx - 1u;
      const uint32_t d = min(ffbh_raw(t.b0 & ((1u << b) - 1u)), tp);
      rows[index(r)] = ((lane64 + 31u - d) | ((lane64 + b) << 16)) + shift;
      r++;
    }
  }
  {
    const uint32_t tp = min(min(ffbh_raw(t.b0) | 32u, ffbh_raw(pb1) | 64u), ffbh_raw(pb0) | 96u);
    uint32_t xx = t.e1;
    while (xx) {
      const uint32_t b = static_cast<uint32_t>(__builtin_ctz(xx));
      xx &= xx - 1u;
      const uint32_t d = min(ffbh_raw(t.b1 & ((1u << b) - 1u)), tp);
      rows[index(r)] = ((lane64 + 63u - d) | ((lane64 + 32u + b) << 16)) + shift;
      r++;
    }
  }
}

// Window of the wave-tile that starts at haystack byte lo: bytes [lo - 64, lo + 4032) through a buffer resource sized to the
// bytes that exist (rounded up to a dword): lanes past the end of the input read zeros, no tail path.  nvalid = window bytes
// that are data (or lie in front of the haystack); the window of the haystack's first tile starts 64 bytes in front of the
// haystack: those lanes are sent out of range by the caller.
template <int PRE = kFPre>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fields_window(const uint8_t* hay, uint64_t len, uint64_t lo, bool live, int32_t& nvalid) {
  int nrec = 0;
  uint64_t wlo = 0;
  nvalid = 0;
  if (live && lo < len) {
    wlo = lo >= static_cast<uint64_t>(PRE) ? lo - PRE : 0;
    const uint64_t rem = len - wlo;
    const uint64_t full = static_cast<uint64_t>(kFWin) - (lo - wlo == 0 ? PRE : 0);
    nrec = rem >= full ? static_cast<int>(full) : static_cast<int>((rem + 3) & ~3ull);
    const uint64_t nv = len - lo + PRE;
    nvalid = nv >= static_cast<uint64_t>(kFWin) ? kFWin : static_cast<int32_t>(nv);
  }
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(hay) + wlo, 0, nrec, 0x00020000);
}
// Phase A for one window held in x[]: class pieces into the wave's LDS scratch; every vector's register is refilled from
// `rnext` right behind its last use.  Returns the lane's words (d1:d0), (p1:p0), masked to the valid bytes of a short window.
// CARRY (the persistent kernel): carry_cur — the words 0..3 of this window were left in sd / sp [0..3] by the tile in front (the
// pieces of the first 16 lanes' first vector, which was not loaded, go to the dump words 64..67); carry_next — the next window
// follows this one in the same unit: its first 256 bytes are not loaded, and this tile's words 60..63 are left behind for it.
template <int KD, int KP, bool CARRY = false>
__device__ __forceinline__ void fields_words(u32x4 (&x)[4], __amdgpu_buffer_rsrc_t rnext, int lane, uint64_t* sd, uint64_t* sp, int32_t nvalid,
                                             uint32_t dlo4, uint32_t dhi4, uint32_t plo4, uint32_t phi4, uint32_t& sink,
                                             uint32_t& d0, uint32_t& d1, uint32_t& p0, uint32_t& p1, bool carry_cur = false, bool carry_next = false) {
  {
    uint16_t* pd = reinterpret_cast<uint16_t*>(sd);
    uint16_t* pp = reinterpret_cast<uint16_t*>(sp);
    const uint32_t voff = static_cast<uint32_t>(lane) << 4;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int at = lane + 64 * k;
      if (CARRY && k == 0) at = (carry_cur && lane < 16) ? lane + 256 : lane;
      if (CXG_FABL >= 4) {
        sink ^= x[k].x ^ x[k].y ^ x[k].z ^ x[k].w;
      } else if (CXG_FABL == 3) {
        pd[at] = static_cast<uint16_t>(x[k].x ^ x[k].z ^ x[k].y ^ x[k].w);
        pp[at] = 0;
      } else {
        const u32x4 t = x[k] & 0x7F7F7F7Fu;
        pd[at] = static_cast<uint16_t>(piece16<KD>(x[k], t, dlo4, dhi4));
        pp[at] = static_cast<uint16_t>(piece16<KP>(x[k], t, plo4, phi4));
      }
      uint32_t off = voff + 1024u * k;
      if (CARRY && k == 0 && carry_next && lane < 16) off = 0x7FFFFFF0u;   // out of range: zeros, nothing fetched
      x[k] = __builtin_amdgcn_raw_buffer_load_b128(rnext, off, 0, CXG_HAY_LOAD_AUX);
      __builtin_amdgcn_sched_barrier(0);                            // keep the refill right behind its vector's last use
    }
  }
  d0 = d1 = p0 = p1 = 0;
  if (CXG_FABL >= 4) return;
  wave_lds_sync();
  int lw = lane;                                                  // second opaque copy: word address = base + 8 * lane by shift, not (piece address) + 6 * lane by v_mul_lo
  asm volatile("" : "+v"(lw));
  const uint64_t Dw = sd[lw], Pw = sp[lw];
  d0 = static_cast<uint32_t>(Dw); d1 = static_cast<uint32_t>(Dw >> 32);
  p0 = static_cast<uint32_t>(Pw); p1 = static_cast<uint32_t>(Pw >> 32);
  if (nvalid != kFWin) {                                          // short last window: the up to 3 bytes behind the input in its last dword are not data
    const int32_t nf = nvalid - 64 * lane;
    const uint64_t vf = nf <= 0 ? 0ull : (nf >= 64 ? ~0ull : ((1ull << nf) - 1ull));
    d0 &= static_cast<uint32_t>(vf); d1 &= static_cast<uint32_t>(vf >> 32);
    p0 &= static_cast<uint32_t>(vf); p1 &= static_cast<uint32_t>(vf >> 32);
  }
  if (CARRY && carry_next && lane >= 60) {                          // words 60..63 are the next window's words 0..3
    sd[lane - 60] = (static_cast<uint64_t>(d1) << 32) | d0;
    sp[lane - 60] = (static_cast<uint64_t>(p1) << 32) | p0;
  }
}
// The first loads of a wave (window of the tile at `lo`); `first`: the haystack's first tile, window bytes 0..63 do not exist.
template <int PRE = kFPre>
__device__ __forceinline__ void fields_first_loads(u32x4 (&x)[4], __amdgpu_buffer_rsrc_t r0, int lane, bool first) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t off = static_cast<uint32_t>(lane + 64 * k) << 4;
    if (first) off = off >= static_cast<uint32_t>(PRE) ? off - PRE : 0x7FFFFFF0u;
    x[k] = __builtin_amdgcn_raw_buffer_load_b128(r0, off, 0, CXG_HAY_LOAD_AUX);
    __builtin_amdgcn_sched_barrier(0);                              // issue order = use order: the tile loop waits for x[0] with vmcnt(3), not for all four
  }
}


// ---- literal programs on the persistent kernel (round 5) -----------------------------------------------------------------
// A chain of m single-byte steps over NC <= 4 distinct bytes with no border (no proper prefix of the literal is a suffix of it:
// occurrences cannot overlap, so FindAll = all occurrences): `error`, `GET`, `HTTP/1.1`.  One bitmap per distinct byte (phase A as
// for the fields programs), then  O = AND_j (B_c(j) >> j)  — an occurrence starts where byte j of the literal stands j places on,
// for every j —, ends = O << m.  No candidate is verified against memory and nothing is walked: simd.Memmem's rare-byte pair scan
// (simd/memmem.go:53-152) with every byte of the needle in the filter.
constexpr int kLitClsStride = kWavesPerBlock * (64 + 4);     // words between the bitmaps of two classes (k_scan_fields_pers s_c)
struct LitRegs { uint32_t m, nc; uint32_t b4[4]; uint64_t cls2_lo, cls2_hi; };   // m steps; byte of class c splat; 2 bits per step: its class

template <int NC, bool CARRY>
__device__ __forceinline__ void lit_words(u32x4 (&x)[4], __amdgpu_buffer_rsrc_t rnext, int lane, uint64_t* sc0, int32_t nvalid, const LitRegs& lr,   // sc0: this wave's words of class 0; class c at + c * kLitClsStride
                                          bool carry_cur, bool carry_next) {
  {
    const uint32_t voff = static_cast<uint32_t>(lane) << 4;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int at = lane + 64 * k;
      if (CARRY && k == 0) at = (carry_cur && lane < 16) ? lane + 256 : lane;
      const u32x4 t = x[k] & 0x7F7F7F7Fu;
#pragma unroll
      for (int c = 0; c < NC; c++) reinterpret_cast<uint16_t*>(sc0 + c * kLitClsStride)[at] = static_cast<uint16_t>(piece16<kClsByte>(x[k], t, lr.b4[c], 0u));
      uint32_t off = voff + 1024u * k;
      if (CARRY && k == 0 && carry_next && lane < 16) off = 0x7FFFFFF0u;
      x[k] = __builtin_amdgcn_raw_buffer_load_b128(rnext, off, 0, CXG_HAY_LOAD_AUX);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  wave_lds_sync();
  int lw = lane;
  asm volatile("" : "+v"(lw));
  if (nvalid != kFWin) {                                            // short last window (rare): what lies behind the input is no byte of the literal
    const int32_t nf = nvalid - 64 * lane;
    const uint64_t vf = nf <= 0 ? 0ull : (nf >= 64 ? ~0ull : ((1ull << nf) - 1ull));
#pragma unroll
    for (int c = 0; c < NC; c++) sc0[c * kLitClsStride + lw] &= vf;
    wave_lds_sync();
  }
  (void)lw;
}
// words 60..63 are the next window's words 0..3 — BEHIND lit_core, which reads the bitmaps from LDS
template <int NC>
__device__ __forceinline__ void lit_carry(uint64_t* sc0, int lane) {
  if (lane >= 60) {
#pragma unroll
    for (int c = 0; c < NC; c++) sc0[c * kLitClsStride + lane - 60] = sc0[c * kLitClsStride + lane];
  }
  wave_lds_sync();
}

// The bitmaps stay in LDS: step j reads the words of its class at this lane and the next one (one ds_read2_b64, a scalar base per
// class) and funnels them right by j.  No register arrays, no branch inside the step.
template <unsigned long long OWN>
__device__ __forceinline__ FieldsTile lit_core(const uint64_t* sc0, int lane, const LitRegs& lr) {
  uint32_t o0 = ~0u, o1 = ~0u;
  const uint64_t* base = sc0 + lane;
  auto cls_of = [&](uint32_t j) -> uint32_t { return static_cast<uint32_t>((j < 32u ? lr.cls2_lo >> (2u * j) : lr.cls2_hi >> (2u * (j - 32u))) & 3ull); };
  const uint64_t* w = base + cls_of(0) * kLitClsStride;
  uint64_t cur = w[0], nxt = w[1];                                    // (lane 63 reads a dump word: it owns nothing)
  for (uint32_t j = 0; j < lr.m; j++) {                               // uniform trip count, shifts and classes; the next step's words are on their way
    const uint32_t h0 = static_cast<uint32_t>(cur), h1 = static_cast<uint32_t>(cur >> 32), n0 = static_cast<uint32_t>(nxt), n1 = static_cast<uint32_t>(nxt >> 32);
    if (j + 1u < lr.m) { w = base + cls_of(j + 1u) * kLitClsStride; cur = w[0]; nxt = w[1]; }
    uint32_t r0, r1;
    if (j < 32u) { r0 = __builtin_amdgcn_alignbit(h1, h0, j); r1 = __builtin_amdgcn_alignbit(n0, h1, j); }
    else { r0 = __builtin_amdgcn_alignbit(n0, h1, j - 32u); r1 = __builtin_amdgcn_alignbit(n1, n0, j - 32u); }
    o0 &= r0; o1 &= r1;
  }
  o0 = sel_lanes(o0, OWN); o1 = sel_lanes(o1, OWN);                   // occurrences that START in the tile
  // ends (exclusive) = starts << m, m < 64: the bits that leave this lane's word arrive in the next one's
  const uint32_t p0 = dpp_from_lower_z(o0), p1 = dpp_from_lower_z(o1);
  uint32_t e0, e1;
  const uint32_t m = lr.m;
  if (m < 32u) { e0 = __builtin_amdgcn_alignbit(o0, p1, 32u - m); e1 = __builtin_amdgcn_alignbit(o1, o0, 32u - m); }
  else if (m == 32u) { e0 = p1; e1 = o0; }
  else { e0 = __builtin_amdgcn_alignbit(p1, p0, 64u - m); e1 = __builtin_amdgcn_alignbit(o0, p1, 64u - m); }
  return FieldsTile{e0, e1, o0, o1, false};
}
}  // namespace

// K: number of fields (2..4).  KD / KP: kind of the field / separator class (walk.hpp ChainClassKind; kClsRange also
// serves single bytes and digits as separators).
//
// A workgroup takes 32 consecutive wave-tiles (120 KiB), orders its rows after one barrier and looks back
// (block_common.hpp), as the other wave kernels do.  (Round 3 also built and measured three other orderings of the same tile
// mathematics — a persistent grid with a dense window and a scan server, a persistent grid with deferred look-back, a
// two-level look-back: all slower; DESIGN section 5 and profiles/r03_fields_ablation.txt have the numbers, git history the code.)
template <int K, int KD, int KP>
__global__ __launch_bounds__(kThreads, CXG_FIELDS_WAVES) void k_scan_fields_wave(ScanArgs a) {
  __shared__ __attribute__((aligned(16))) uint64_t s_d[kWavesPerBlock][64];     // field-class bitmap of the wave's window
  __shared__ __attribute__((aligned(16))) uint64_t s_p[kWavesPerBlock][64];     // separator-class bitmap
  __shared__ uint32_t s_row[kWavesPerBlock][kFRows];                            // start | end << 16, window bit indices
  __shared__ uint32_t s_cnt[kWavesPerBlock][kTilesPerWave];
  __shared__ uint32_t s_qbase[kWavesPerBlock * kTilesPerWave + 1];
  __shared__ uint64_t s_group;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = lane0;
  uint64_t group = blockIdx.x;
  if (!a.static_groups) {                                            // uniform: kernel argument
    if (tid == 0) s_group = claim_group(false, a.ticket, a.ngroups);
    __syncthreads();
    group = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_group >> 32))) << 32) |
            static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_group)));
  }
  if (group >= a.ngroups) return;
  const ChainAux* gch = reinterpret_cast<const ChainAux*>(a.chain);   // kernel argument segment: scalar loads
  const uint32_t dlo4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gch->cls_lo[0] * 0x01010101u)));
  const uint32_t dhi4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((0x7Fu - gch->cls_hi[0]) * 0x01010101u)));
  const uint32_t plo4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gch->cls_lo[1] * 0x01010101u)));
  const uint32_t phi4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((0x7Fu - gch->cls_hi[1]) * 0x01010101u)));
  constexpr int tpw = kTilesPerWave;
  uint32_t nrows_w = 0;                                              // wave-uniform
  uint32_t fallback = 0;
  // FindAll with n > 0: enough rows counted in front of this group — it contributes nothing (block_common.hpp tile_lookback)
  if (limit_reached_skip(a, group, &s_base)) return;                 // FindAll with n > 0 (block_common.hpp)
  constexpr int ntile = tpw;
  auto tile_lo_of = [&](int jj) { return (group * (kWavesPerBlock * tpw) + static_cast<uint64_t>(jj) * kWavesPerBlock + wave) * static_cast<uint64_t>(kWaveTile); };
  const uint64_t pt0 = a.prof ? __builtin_readcyclecounter() : 0ull;   // CXG_PROF=1: cycles of wave 0 per phase, summed over the workgroups

  u32x4 x[4];
  uint32_t sink = 0;                                                   // ablations only
  int32_t nvalid_cur = 0;
  fields_first_loads(x, fields_window(a.hay, a.len, tile_lo_of(0), true, nvalid_cur), lane, group == 0 && wave == 0);

  for (int j = 0; j < ntile; j++) {
    // Opaque copy of the lane id per wave-tile: lane-derived values are recomputed (a few ALU ops) instead of being hoisted
    // out of the loop and spilled — a scratch reload waits on vmcnt and would drain the loads in flight.
    lane = lane0;
    asm volatile("" : "+v"(lane));
    int32_t nvalid_next = 0;
    const __amdgpu_buffer_rsrc_t rnext = fields_window(a.hay, a.len, tile_lo_of(j + 1), j + 1 < tpw, nvalid_next);
    uint32_t d0, d1, p0, p1;
    fields_words<KD, KP>(x, rnext, lane, s_d[wave], s_p[wave], nvalid_cur, dlo4, dhi4, plo4, phi4, sink, d0, d1, p0, p1);
    nvalid_cur = nvalid_next;
    if (CXG_FABL >= 4) continue;
    const FieldsTile t = fields_core<K>(d0, d1, p0, p1);
    if (t.ovf) fallback |= 1u;
    const uint32_t c = static_cast<uint32_t>(__popc(t.e0)) + static_cast<uint32_t>(__popc(t.e1));
    const uint32_t incl = wave_inclusive_sum_fused(c);
    uint32_t tot = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
    if (CXG_FABL >= 1) tot = 0;
    if (tot != 0 && (a.out != nullptr || a.max_len != 0))
      fields_rows(t, lane, s_row[wave], nrows_w + incl - c, [](uint32_t r) { return min(r, static_cast<uint32_t>(kFRows - 1)); });   // overflow: flagged below, rows void
    if (lane == 0) s_cnt[wave][j] = tot;
    nrows_w += tot;
  }
  if (CXG_FABL >= 4 && sink == 0x12345u) fallback |= 4u;              // keeps the loads alive
  if (CXG_FABL == 5) { if (fallback != 0 && lane0 == 0) raise_err(a.err, 8u | (fallback << 8)); return; }
  if (CXG_FABL == 4 && lane0 == 0) { for (int j = 0; j < tpw; j++) s_cnt[wave][j] = 0; }
  if (nrows_w > static_cast<uint32_t>(kFRows)) fallback |= 16u;
  wave_lds_sync();
  {                                                                   // rows whose start was not found (start >= end), UseBoth restart span
    bool bad = false, long_hit = false;
    if (a.out != nullptr || a.max_len != 0) {
      for (uint32_t r = lane0; r < nrows_w && r < static_cast<uint32_t>(kFRows); r += 64) {
        const uint32_t v = s_row[wave][r];
        const uint32_t s = v & 0xFFFFu, e = v >> 16;
        bad = bad || s >= e;
        long_hit = long_hit || (a.max_len != 0 && e - s > a.max_len);
      }
    }
    if (__ballot(bad) != 0ull) fallback |= 2u;
    if (__ballot(long_hit) != 0ull && lane0 == 0) raise_err(a.err, kErrLongMatch);
  }
  if (fallback != 0 && lane0 == 0) raise_err(a.err, 8u | (fallback << 8));
  const uint64_t pt1 = a.prof ? __builtin_readcyclecounter() : 0ull;
  __syncthreads();
  const uint64_t pt2 = a.prof ? __builtin_readcyclecounter() : 0ull;

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
  // Count (no rows wanted, no limit): nobody needs this group's place in the output, only the grand total — the look-back, in
  // which a workgroup spends more time than scanning, is replaced by a plain store and a one-workgroup sum behind the kernel
  if (a.count_sum) { if (tid == 0) a.status[group] = total; return; }
  if (a.dbg & 2u) { if (tid == 0) s_base = 0; __syncthreads(); }     // CXG_DEBUG=2 (timing experiments, rows land in the wrong places): no look-back
  else tile_lookback(a.status, a.total, a.err, group, a.ngroups, total, &s_base, a.epoch, a.limit, a.stop);
  if (a.prof && tid == 0) {
    const uint64_t pt3 = __builtin_readcyclecounter();
    atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 1), static_cast<unsigned long long>(pt1 - pt0));   // tile loop
    atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 2), static_cast<unsigned long long>(pt2 - pt1));   // first barrier (waiting for the slowest wave)
    atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 3), static_cast<unsigned long long>(pt3 - pt2));   // prefix + look-back
    atomicAdd(reinterpret_cast<unsigned long long*>(a.prof + 4), 1ull);
  }
  if (a.out == nullptr) return;
  const uint64_t base = s_base;
  const int64_t origin = a.base + static_cast<int64_t>(group * static_cast<uint64_t>(kWaveTile) * kWavesPerBlock * tpw) - kFPre;
  uint32_t start = 0;
  for (int j = 0; j < tpw; j++) {
    const uint32_t n = s_cnt[wave][j];
    const uint64_t dst = base + s_qbase[j * kWavesPerBlock + wave];
    const int64_t tb = origin + static_cast<int64_t>(j * kWavesPerBlock + wave) * kWaveTile;
    for (uint32_t i = lane0; i < n; i += 64) {
      const uint32_t r = start + i;
      if (r < static_cast<uint32_t>(kFRows) && dst + i < a.cap) {
        const uint32_t v = s_row[wave][r];
        longlong2 o; o.x = tb + (v & 0xFFFFu); o.y = tb + (v >> 16);
        store_pair_nt(a.out + (dst + i) * a.row_width, o.x, o.y);
      }
    }
    start += n;
  }
}

// ---- tile mathematics of the run-a-run-b-run programs (k_scan_trio_wave below, and the persistent kernel in its TRIO mode) ----
namespace {
#ifndef CXG_TRIO_ROWS
#define CXG_TRIO_ROWS (64 * kTilesPerWave)
#endif
#ifndef CXG_TRIO_WAVES
#define CXG_TRIO_WAVES 8
#endif
#ifndef CXG_TRIO_SWAR
#define CXG_TRIO_SWAR 0                                      // measured (profiles/r05_c2_configs.txt, config 5): SWAR + class plan 0.429 ms, the byte table 0.397
#endif
constexpr int kTRows = CXG_TRIO_ROWS;                     // rows buffered per wave and group

struct TrioTile { uint32_t e0, e1; bool ovf; };

// K fields, K - 1 links: lk[i] = bitmap of the bytes of the i-th separator class (word of this lane).  For K >= 3 the separator
// classes are pairwise different (trio_shape): a candidate can then only share the LAST run of an earlier match.
// EQ: one separator for every link (K >= 3).  Two candidates can then share up to K - 1 runs, but the selection needs no
// resolution at all: the matches of a super-run are its fields K at a time from its start, as in the fields kernel.
template <int K, bool EQ, unsigned long long OWN = kFOwn>
__device__ __forceinline__ TrioTile trio_core(uint32_t d0, uint32_t d1, const uint32_t (&c0)[K - 1], const uint32_t (&c1)[K - 1]) {
  const uint32_t prev_d1 = dpp_from_lower(d1);
  const uint32_t next_d0 = dpp_from_upper_ones(d0);
  const uint32_t Dl0 = __builtin_amdgcn_alignbit(d0, prev_d1, 31), Dl1 = __builtin_amdgcn_alignbit(d1, d0, 31);   // D << 1
  const uint32_t Dr0 = __builtin_amdgcn_alignbit(d1, d0, 1), Dr1 = __builtin_amdgcn_alignbit(next_d0, d1, 1);     // D >> 1
  uint32_t lk0[K - 1], lk1[K - 1];
  uint32_t l0 = 0, l1 = 0;
#pragma unroll
  for (int i = 0; i < K - 1; i++) { lk0[i] = c0[i] & Dl0 & Dr0; lk1[i] = c1[i] & Dl1 & Dr1; l0 |= lk0[i]; l1 |= lk1[i]; }
  const uint32_t la0 = lk0[0], la1 = lk1[0];                       // the first link of a match
  const uint32_t x0 = d0 | l0, x1 = d1 | l1;                       // super-runs
  const uint32_t prev_l1 = dpp_from_lower(l1);
  const uint32_t Ll0 = __builtin_amdgcn_alignbit(l0, prev_l1, 31), Ll1 = __builtin_amdgcn_alignbit(l1, l0, 31);   // L << 1
  const uint32_t ws0 = sel_lanes(d0 & ~Dl0 & ~Ll0, OWN), ws1 = sel_lanes(d1 & ~Dl1 & ~Ll1, OWN);   // owned super-run starts
  const unsigned long long PPd = __builtin_amdgcn_uicmpl((static_cast<uint64_t>(d1) << 32) | d0, ~0ull, 32 /*eq*/);
  const unsigned long long PPx = __builtin_amdgcn_uicmpl((static_cast<uint64_t>(x1) << 32) | x0, ~0ull, 32 /*eq*/);
  unsigned long long ovf = 0;
  auto carry_in = [&](unsigned long long GG, unsigned long long PP) -> unsigned long long {
    const unsigned long long Pe = PP & ~GG;
    const unsigned long long recv = (Pe + (GG << 1)) ^ Pe;
    ovf |= GG | (Pe & recv);
    return recv;
  };
  (void)PPx;
  // hop over one run from link bits q (subset of L): the carry of q + q runs through the F bytes behind the link
  auto hop = [&](uint32_t q0, uint32_t q1, uint32_t& r0, uint32_t& r1) {
    unsigned long long G2;
    add64_co(d0 | q0, d1 | q1, q0, q1, r0, r1, G2);
    add64_cin(r0, r1, carry_in(G2, PPd));
  };
  auto hops = [&](uint32_t q0, uint32_t q1, uint32_t& e0, uint32_t& e1) {   // ends of the candidates whose first link is in q
    uint32_t r0, r1;
    hop(q0, q1, r0, r1);
#pragma unroll
    for (int i = 1; i < K - 1; i++) {                                       // ... every further run must end on the link of its place
      const uint32_t m0 = r0 & lk0[i], m1 = r1 & lk1[i];
      hop(m0, m1, r0, r1);
    }
    e0 = r0 & ~d0; e1 = r1 & ~d1;
  };
  uint32_t e0, e1;
  if (EQ) {
    uint32_t r0, r1, q0, q1, sel0 = 0, sel1 = 0;
    hop(ws0, ws1, r0, r1);                                                  // over the first run of every owned super-run:
    q0 = r0 & l0; q1 = r1 & l1;                                             // its link, if it has one
    for (int guard = 0; guard < 64; guard++) {
      hops(q0, q1, e0, e1);
      sel0 |= e0; sel1 |= e1;
      const uint32_t n0 = e0 & l0, n1 = e1 & l1;                            // an end on a link: more fields behind it
      if (__builtin_amdgcn_uicmpl((static_cast<uint64_t>(n1) << 32) | n0, 0ull, 33 /*ne*/) == 0ull) break;
      hop(n0, n1, r0, r1);                                                  // over the next match's first run
      q0 = r0 & l0; q1 = r1 & l1;
      if (guard == 63) ovf |= 1ull << 63;
    }
    return TrioTile{sel0, sel1, (ovf >> 63) != 0ull};
  }
  // owned span: the bits of X that the addition of the owned starts clears
  uint32_t s0, s1;
  unsigned long long GG;
  add64_co(x0, x1, ws0, ws1, s0, s1, GG);
  add64_cin(s0, s1, carry_in(GG, PPx));
  const uint32_t own0 = x0 & ~s0, own1 = x1 & ~s1;
  hops(la0 & own0, la1 & own1, e0, e1);
  // ends that sit on a first link: the candidate that begins with that link (if it is one) shares a run with this match
  uint32_t xa0 = e0 & la0, xa1 = e1 & la1;
  if (__builtin_amdgcn_uicmpl((static_cast<uint64_t>(xa1) << 32) | xa0, 0ull, 33 /*ne*/) != 0ull) {
    uint32_t r0 = e0, r1 = e1, sel0 = 0, sel1 = 0;                          // undecided / selected (by their ends)
    for (int guard = 0; guard < 64; guard++) {
      uint32_t k0, k1;
      hops(r0 & la0, r1 & la1, k0, k1);                                     // ends of candidates blocked by undecided ones
      const uint32_t h0 = r0 & ~k0, h1 = r1 & ~k1;                          // heads: undecided, not blocked by an undecided one
      sel0 |= h0; sel1 |= h1;
      hops(h0 & la0, h1 & la1, k0, k1);                                     // what the heads block
      r0 &= ~(h0 | k0); r1 &= ~(h1 | k1);
      if (__builtin_amdgcn_uicmpl((static_cast<uint64_t>(r1) << 32) | r0, 0ull, 33 /*ne*/) == 0ull) break;
      if (guard == 63) ovf |= 1ull << 63;
    }
    e0 = sel0; e1 = sel1;
  }
  return TrioTile{e0, e1, (ovf >> 63) != 0ull};
}

#ifndef CXG_TRIO_ROWS_FAST
#define CXG_TRIO_ROWS_FAST 1                                  // trio_rows: the K highest bits of one 64-bit word when the row lies within 64 bytes (0: always the two-word search; A/B)
#endif
// highest set bit of the 128-bit value (h : l), or -1; and the value with that bit cleared
__device__ __forceinline__ int32_t take_top(uint64_t& l, uint64_t& h) {
  if (h) { const int32_t k = 63 - __builtin_clzll(h); h &= ~(1ull << k); return 64 + k; }
  if (l) { const int32_t k = 63 - __builtin_clzll(l); l &= ~(1ull << k); return k; }
  return -1;
}

// rows of the lane's end bits: start | end << 16 (window bit indices) at rows[r], and the links as distances from the start,
// (la - start) | (lb - start) << 8, at links[r] (all three lie within two words: < 128) — 6 bytes per row keep the kernel at 8
// workgroups per CU with 512 rows per wave; r counts up from r0
template <int K, typename LinkT>
__device__ __forceinline__ void trio_rows(const TrioTile& t, uint32_t d0, uint32_t d1, int lane, uint32_t* rows, LinkT* links, uint32_t r, uint32_t cap, uint32_t shift = 0u) {   // shift: added to both halves of a row
  const uint64_t z = ~((static_cast<uint64_t>(d1) << 32) | d0);             // bytes outside F, this lane's word
  const uint64_t pz = (static_cast<uint64_t>(dpp_from_lower_z(static_cast<uint32_t>(z >> 32))) << 32) | dpp_from_lower_z(static_cast<uint32_t>(z));   // previous lane's (lane 0: none)
  const int32_t base = (lane << 6) - 64;                                    // window index of bit 0 of (pz : z)
  uint64_t ee = (static_cast<uint64_t>(t.e1) << 32) | t.e0;
  while (ee) {
    const int32_t b = __builtin_ctzll(ee);
    ee &= ee - 1ull;
    uint32_t w0 = 0, w1 = 0;                                                // start not found within two words: row void (start >= end), caught below
    // the 64 bytes below the end as ONE word (bit i = index b + i of (z : pz)): when the K bytes outside F that make up the row — the
    // links, last first, then the byte in front of the match — lie in it (every match shorter than 64 bytes), they are its K highest
    // bits: K x (count leading zeros, clear) without a branch (round 5: 0.402 -> 0.389 ms on config 5, profiles/r05_c19_cfg5.txt).
    uint64_t W = (pz >> b) | (b ? (z << (64 - b)) : 0ull);
    if (CXG_TRIO_ROWS_FAST && __popcll(W) >= K) {
      int32_t idx[K];                                                       // indices in (z : pz), nearest first
#pragma unroll
      for (int i = 0; i < K; i++) { const int32_t c = __builtin_clzll(W); idx[i] = b + 63 - c; W &= ~(0x8000000000000000ull >> c); }
      const int32_t ps = idx[K - 1];
      w0 = (static_cast<uint32_t>(base + ps + 1) | (static_cast<uint32_t>(base + 64 + b) << 16)) + shift;
#pragma unroll
      for (int i = 0; i < K - 1; i++) w1 |= static_cast<uint32_t>(idx[K - 2 - i] - ps - 1) << (8 * i);
    } else {
      uint64_t l = pz, h = b ? (z & ((1ull << b) - 1ull)) : 0ull;           // bytes outside F below the end, nearest first:
      int32_t pl[K - 1];                                                    // the links, last first
#pragma unroll
      for (int i = K - 2; i >= 0; i--) pl[i] = take_top(l, h);
      const int32_t ps = take_top(l, h);                                    // the byte in front of the match
      if (ps >= 0) {
        w0 = (static_cast<uint32_t>(base + ps + 1) | (static_cast<uint32_t>(base + 64 + b) << 16)) + shift;
#pragma unroll
        for (int i = 0; i < K - 1; i++) w1 |= static_cast<uint32_t>(pl[i] - ps - 1) << (8 * i);
      }
    }
    const uint32_t rr = r < cap ? r : cap - 1u;
    rows[rr] = w0; links[rr] = static_cast<LinkT>(w1);
    r++;
  }
}
}  // namespace


namespace {
// Phase A of the TRIO mode of the persistent kernel: the flags of K classes (field class F, K - 1 separator bytes) from ONE byte table in
// LDS, 16 bits per class and vector through the wave's bitmaps — k_scan_trio_wave's phase A with the halo carry of fields_words.
// sc0: this wave's words of class 0 (F); class c at + c * kLitClsStride.
template <int K, bool CARRY>
__device__ __forceinline__ void trio_words(u32x4 (&x)[4], __amdgpu_buffer_rsrc_t rnext, int lane, uint64_t* sc0, const uint8_t* s_cls, int32_t nvalid,
                                           bool carry_cur, bool carry_next, uint32_t& d0, uint32_t& d1, uint32_t (&c0)[K - 1], uint32_t (&c1)[K - 1]) {
  {
    const uint32_t voff = static_cast<uint32_t>(lane) << 4;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int at = lane + 64 * k;
      if (CARRY && k == 0) at = (carry_cur && lane < 16) ? lane + 256 : lane;
      const uint32_t w[4] = {x[k].x, x[k].y, x[k].z, x[k].w};
      uint32_t fd = 0, fc[K - 1];
#pragma unroll
      for (int i = 0; i < K - 1; i++) fc[i] = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t f = static_cast<uint32_t>(s_cls[w[q] & 0xFFu]) | (static_cast<uint32_t>(s_cls[(w[q] >> 8) & 0xFFu]) << 8) |
                           (static_cast<uint32_t>(s_cls[(w[q] >> 16) & 0xFFu]) << 16) | (static_cast<uint32_t>(s_cls[w[q] >> 24]) << 24);
        const uint32_t wt = (q & 1) ? 0x80402010u : 0x08040201u;
        if (q < 2) {
          fd = __builtin_amdgcn_udot4(f & 0x01010101u, wt, fd, false);
#pragma unroll
          for (int i = 0; i < K - 1; i++) fc[i] = __builtin_amdgcn_udot4((f >> (i + 1)) & 0x01010101u, wt, fc[i], false);
        } else {
          fd += __builtin_amdgcn_udot4(f & 0x01010101u, wt, 0u, false) << 8;
#pragma unroll
          for (int i = 0; i < K - 1; i++) fc[i] += __builtin_amdgcn_udot4((f >> (i + 1)) & 0x01010101u, wt, 0u, false) << 8;
        }
      }
      reinterpret_cast<uint16_t*>(sc0)[at] = static_cast<uint16_t>(fd);
#pragma unroll
      for (int i = 0; i < K - 1; i++) reinterpret_cast<uint16_t*>(sc0 + (i + 1) * kLitClsStride)[at] = static_cast<uint16_t>(fc[i]);
      uint32_t off = voff + 1024u * k;
      if (CARRY && k == 0 && carry_next && lane < 16) off = 0x7FFFFFF0u;
      x[k] = __builtin_amdgcn_raw_buffer_load_b128(rnext, off, 0, CXG_HAY_LOAD_AUX);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  wave_lds_sync();
  int lw = lane;
  asm volatile("" : "+v"(lw));
  uint64_t vf = ~0ull;
  if (nvalid != kFWin) {
    const int32_t nf = nvalid - 64 * lane;
    vf = nf <= 0 ? 0ull : (nf >= 64 ? ~0ull : ((1ull << nf) - 1ull));
  }
  uint64_t W[K];
#pragma unroll
  for (int c = 0; c < K; c++) W[c] = sc0[c * kLitClsStride + lw] & vf;
  d0 = static_cast<uint32_t>(W[0]); d1 = static_cast<uint32_t>(W[0] >> 32);
#pragma unroll
  for (int i = 0; i < K - 1; i++) { c0[i] = static_cast<uint32_t>(W[i + 1]); c1[i] = static_cast<uint32_t>(W[i + 1] >> 32); }
  if (CARRY && carry_next && lane >= 60) {                          // words 60..63 are the next window's words 0..3
#pragma unroll
    for (int c = 0; c < K; c++) sc0[c * kLitClsStride + lane - 60] = W[c];
  }
}
}  // namespace

// =====================================================================================================================
// k_scan_fields_pers — the same tile mathematics on a PERSISTENT grid of autonomous waves, the ordering of the rows DEFERRED
// by one round (round 4).  Why: the grouped kernel above spends more of a workgroup's life waiting in the look-back than
// scanning (profiles/r03_fields_ablation.txt section 4): workgroups that wait hold their slots, the next generation starts
// when the convoy in front resolves, finishes together and queues up again.  Here W = 4 x CUs x occupancy waves stay for the
// whole launch and never meet at a barrier.  In round r wave v scans unit r * W + v — 8 consecutive wave-tiles = 30 KiB, the
// window of the next unit's first tile in flight behind the last tile of this one — parks the rows in LDS as offsets from
// the unit's first byte and publishes the unit's row count in its own word (launch epoch << 16 | count).  No atomics:
//   * the LEADER of a block of 64 units (its last unit) owes the block's sum: while it scans round r + 1 it loads the 64
//     unit words of round r once per tile (issued behind the tile's refills, looked at behind the tile's mathematics — it
//     never spins while it has tiles to scan) and stores the sum when all are there;
//   * the leader of the round (its last unit) likewise owes the round's RECORDS: one look per tile at the <= 128 block sums,
//     then one 16-byte record per block: rows of the round in front of the block, rows of the round.
// At the end of round r + 1 every wave orders and writes its rows of round r: the record of its block and the words of the
// units of its block in front of it — one 16-byte and one 4-byte load per lane, issued one tile before they are needed:
//   base = rows in front of the round (carried in a register) + record.base + units of my block below me
// with no chain of look-back windows.  Rows of two rounds live in LDS (2 x 2 KiB per wave).
// (Measured on the way, profiles/r04_pers_*: whole workgroups as units with every thread reading all G words of a round —
// 6 KiB at the same addresses from 1 536 workgroups — 0.30 ms against 0.25 for the grouped kernel, better with FEWER
// workgroups; block sums by plain atomics on 3 cache lines: 0.66 ms; by returning atomics, one counter per 4 KiB, last
// arriver sums: 0.27-0.28, every unit waiting ~15 us per round for its record — 64 serialised atomics under a streaming
// load take most of a round; statistics by an atomic per unit on ONE word: +0.8 ms.  Rows written with no protocol: 0.22-0.23.)
// The last round is tapered: what is left of the haystack behind the full rounds is spread over all waves (units of 1..8
// tiles).
// Co-residency: every wave of the grid must be resident (a waiting wave waits for words of waves that run at the same
// time).  The grid comes from the occupancy query; should a device admit fewer, the spin watchdog raises error bit 1 and
// the host reruns with the grouped kernel (capi_ladder.hip staticGroupsOk).
#ifndef CXG_PF_OCC
#define CXG_PF_OCC 6
#endif
// -DCXG_PFABL=n (experiments only, rows land in the wrong places): 1 = nothing read, no duties; 2 = no unit word either; 3 = no rows written
#ifndef CXG_PFABL
#define CXG_PFABL 0
#endif
#ifndef CXG_PF_PRIO
#define CXG_PF_PRIO 1
#endif
#ifndef CXG_PF_PRIO_SHIFT
#define CXG_PF_PRIO_SHIFT 14
#endif
#ifndef CXG_PF_SLEEP
#define CXG_PF_SLEEP 16
#endif
// Tiles per unit.  ODD on purpose: the waves start in lockstep, so at any moment they read at (v * unit + progress) — with a
// unit of 8 tiles = 120 x 256 bytes the offsets v * 120 mod 128 take 16 values, i.e. an eighth of the channels of any
// power-of-two interleave; 7 tiles = 105 x 256 bytes is coprime to it and spreads the waves over all of them.  Round 5: 9 tiles
// (135 x 256 bytes): with the halo carried inside a unit a longer unit re-reads less (1 GiB: 0.2264 -> 0.2218 ms mean of 20,
// 11 tiles 0.2200 but 768 parked rows per wave cost a workgroup per CU; profiles/r05_c1_headline_ab.txt).
#ifndef CXG_PF_TILES
#define CXG_PF_TILES 9
#endif
// 1: line-aligned windows, the 256 bytes two neighbouring tiles of a unit share carried in LDS (round 5); 0: round 4's windows
#ifndef CXG_PF_CARRY
#define CXG_PF_CARRY 1
#endif
constexpr int kPfTiles = CXG_PF_TILES;
static_assert((kPfTiles - 1) * kWaveTile + kWaveTile + kWaveHalo < 65536, "a parked row holds two 16-bit offsets from the unit's first window byte");
constexpr int kPfRows = 64 * (kPfTiles + 1);                 // rows parked per unit and wave (twice: rounds r and r - 1); 64 per tile + 64 as in the grouped kernel
constexpr int kPfMaxWaves = 8192;                            // 128 blocks of 64 units per round
// A wave that has waited this long for a word of another wave gives up (capi_ladder.hip reruns the call one mode down and demotes the
// mode for a term).  Legitimate waits are microseconds; what a missing co-resident wave costs is this limit per call.
// The clock (round 6, scripts/microbench/clocks.hip, profiles/r06_c16_clocks.txt): s_memtime (__builtin_readcyclecounter, clock64) counts
// the shader clock — 2 389 ticks per microsecond at the 2.4 GHz these boxes hold, asleep or busy, one wave or a full grid — so the
// 5 M ticks of round 5 were 2.1 ms, not the "2 to 50 ms" its comment guessed (the 51 ms stalls it saw were a sequence of expiries and
// reruns); s_memrealtime (wall_clock64) counts a constant 100 MHz (99.96 - 99.99 ticks per microsecond measured,
// hipDeviceAttributeWallClockRate = 100 000 kHz).  The limit is stated in that clock: 5 ms.
constexpr uint64_t kPfWaitTicks = 500000ull;                  // s_memrealtime ticks: 5 000 us
#ifdef CXG_PF_WD_MEMTIME                                        // (A/B: the shader clock of round 5, 2.1 ms at 2.4 GHz for the same number)
__device__ __forceinline__ uint64_t pf_wait_clock() { return __builtin_readcyclecounter() / 24u; }
#else
__device__ __forceinline__ uint64_t pf_wait_clock() { return __builtin_amdgcn_s_memrealtime(); }
#endif
// One expiry ends every wait of the launch: the wave that gives up leaves the launch's epoch in a word all waiters look at beside their
// clock.  Without it the waits expired one after another along the chains of rounds and duties — a foreign kernel beside the grid cost
// 51 ms (round 5, 2.1 ms limit) and 87 ms (5 ms limit, profiles/r06_c17_foreign_kernel.txt) per hit, ~20 limits in a row.
constexpr uint32_t kPfAbortWord = 32u * 64u * kPfCtrStride - 1u;       // the last word of pf_ticket (its counters sit at multiples of their stride; unused by the static grid)
__device__ __forceinline__ bool pf_aborted(const ScanArgs& a) { return __hip_atomic_load(a.pf_ticket + kPfAbortWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.pf_epoch; }
__device__ __forceinline__ void pf_abort(const ScanArgs& a) { __hip_atomic_store(a.pf_ticket + kPfAbortWord, a.pf_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool pf_wait_over(const ScanArgs& a, uint64_t t0) { return pf_aborted(a) || pf_wait_clock() - t0 > kPfWaitTicks; }

#ifndef CXG_PF_EAGER
#define CXG_PF_EAGER 0
#endif
#ifndef CXG_PF_TAIL3
#define CXG_PF_TAIL3 1
#endif
#ifndef CXG_PF_TAIL1
#define CXG_PF_TAIL1 2
#endif
// Round 6 built the kernel with CLAIMED units (tickets; below, -DCXG_PF_TICKETS=1) — a wave that holds unit u then only ever waits for smaller
// units, which removes the need for a co-resident grid — and measured it against the static assignment (unit = round * W + wave):
//   * a returning atomic on ONE address is served ~20 times per microsecond; a 6 TB/s scan in 34 KiB units asks ~190 times: 8 counters
//     cost a pure streaming kernel 25 %, 64 counters a cache line apart 2 % (scripts/microbench/ticket_atomics.hip; scalar s_atomic_add
//     works on gfx950 and behaves the same);
//   * with the counters fast, the ORDERING is what costs: rows can only be written once every smaller unit has been counted, and with one
//     unit of slack (two parked row lists fill the LDS that six workgroups per CU leave) every unit ends up waiting for the slowest of the
//     ~6 000 units in flight — 16 GiB: 5.9 ms with a counter per workgroup, 4.8 with the counters asked in rotation, 3.9 with every unit
//     doing its own decoupled look-back over block aggregates, against 3.3 static; 1 GiB 0.27-0.35 ms against 0.22 (profiles/r06_c4..c7_*).
// The static grid stays the product (its spin watchdog and the demotion ladder of capi_ladder.hip cover a grid that is not co-resident); the
// ticketed form stays buildable for whoever finds the second unit of slack.
#ifndef CXG_PF_TICKETS
#define CXG_PF_TICKETS 0
#endif
#if !CXG_PF_TICKETS
// LIT: 0 = a fields program (K, KD, KP as above); 2..4 = a literal over that many distinct bytes (lit_core; K, KD, KP unused);
// 16 + K' + 8 EQ = TRIO mode (round 5): run(F) (byte(c_i) run(F)){K'-1} programs with their capture slots, k_scan_trio_wave's tile
// mathematics (trio_core<K', EQ>) and row epilogue on this kernel's grid and protocol (BASELINE configs[4], `(\w+)@(\w+)\.(\w+)`).
template <int K, int KD, int KP, int LIT = 0>
__global__ __launch_bounds__(kThreads, (LIT >= 16 ? 4 : LIT >= 3 ? 5 : CXG_PF_OCC)) void k_scan_fields_pers(ScanArgs a) {   // (three / four bitmaps: 27 / 29 KB of LDS per workgroup — six do not fit a CU)
  constexpr bool kTrio = LIT >= 16;
  constexpr int TK = kTrio ? ((LIT - 16) & 7) : 2;                    // fields of a TRIO program
  constexpr bool TEQ = kTrio && ((LIT - 16) >> 3) != 0;
  constexpr bool kLit = LIT >= 2 && LIT <= 4;
  typedef typename std::conditional<TK == 4, uint32_t, uint16_t>::type LinkT;
  constexpr int kNBitmaps = kTrio ? TK : (kLit ? LIT : 2);
  __shared__ LinkT s_lnk[kTrio ? 2 : 1][kWavesPerBlock][kTrio ? kPfRows : 1];   // TRIO: the links of a parked row as distances from its start
  __shared__ uint8_t s_cls[kTrio ? 256 : 4];                          // TRIO: byte -> class flags (bit 0 F, bit i + 1 the separator of link i)
  __shared__ __attribute__((aligned(16))) uint64_t s_c[kNBitmaps][kWavesPerBlock][64 + 4];   // class bitmaps of the wave's window (+ 4 dump words: CARRY)
  uint64_t (*const s_d)[64 + 4] = s_c[0];
  uint64_t (*const s_p)[64 + 4] = s_c[1];
  __shared__ uint32_t s_row[2][kWavesPerBlock][kPfRows];                         // rows of round r and r - 1: start | end << 16, offsets from the unit's first byte - kPre

  constexpr bool kCarry = CXG_PF_CARRY != 0;
  constexpr int kPre = kCarry ? kFPrePers : kFPre;
  constexpr unsigned long long kOwn = kCarry ? kFOwnPers : kFOwn;
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = lane0;
  const uint32_t W = gridDim.x * static_cast<uint32_t>(kWavesPerBlock);
  const uint32_t wv = blockIdx.x * static_cast<uint32_t>(kWavesPerBlock) + static_cast<uint32_t>(wave);
  const uint32_t full = a.pf_full, tpw_last = a.pf_tpw_last, units_last = a.pf_units_last;
  if (kTrio) {                                                        // (in front of the first early return: every wave of the workgroup reaches the barrier)
    const ChainAux* tch = reinterpret_cast<const ChainAux*>(a.chain);
    const uint32_t b = static_cast<uint32_t>(tid);
    uint32_t f = chain_class_has(*tch, 0, b) ? 1u : 0u;
    for (int i = 0; i < TK - 1; i++) f |= chain_class_has(*tch, tch->op_cls[2 * i + 1], b) ? (2u << i) : 0u;
    s_cls[tid] = static_cast<uint8_t>(f);
    __syncthreads();
  }
  const uint32_t R_me = full + (wv < units_last ? 1u : 0u);           // rounds in which this wave has a unit
  if (R_me == 0) { if (a.count_sum != 0u && lane0 == 0) a.status[wv] = 0; return; }   // (short input, tapered round with fewer units than waves)
  const ChainAux* gch = reinterpret_cast<const ChainAux*>(a.chain);
  const uint32_t dlo4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gch->cls_lo[0] * 0x01010101u)));
  const uint32_t dhi4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((0x7Fu - gch->cls_hi[0]) * 0x01010101u)));
  const uint32_t plo4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gch->cls_lo[1] * 0x01010101u)));
  const uint32_t phi4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((0x7Fu - gch->cls_hi[1]) * 0x01010101u)));
  LitRegs lr;
  if (kLit) {
    lr.m = gch->nops; lr.nc = gch->ncls; lr.cls2_lo = gch->cls2_lo; lr.cls2_hi = gch->cls2_hi;
#pragma unroll
    for (int c = 0; c < 4; c++) lr.b4[c] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gch->cls_lo[c] * 0x01010101u)));
  }
  const uint32_t ep = a.pf_epoch;
  const uint32_t tag = ep << 16;
  const bool want_rows = a.out != nullptr || a.max_len != 0;
  const bool order = a.count_sum == 0u;                               // count-only calls need no place in the output
  // TRIO: which two slots of a row this lane writes and what they are made of (k_scan_trio_wave's epilogue): sel 0 start, 1 end,
  // 2 + i the end of run i (link i), 7 unset
  uint32_t t_lsh = 0, t_pr = 0, t_sel0 = 0, t_sel1 = 1; int32_t t_off0 = 0, t_off1 = 0; bool t_lane_on = true;
  if (kTrio) {
    const ChainCaps* cp = reinterpret_cast<const ChainCaps*>(a.caps);
    const uint32_t npairs = a.row_width >> 1;
    t_lsh = npairs <= 1u ? 0u : 32u - static_cast<uint32_t>(__builtin_clz(npairs - 1u));
    t_pr = static_cast<uint32_t>(lane0) & ((1u << t_lsh) - 1u);
    t_lane_on = t_pr < npairs;
    auto slot_of = [&](uint32_t k, uint32_t& sel, int32_t& off) {
      const uint32_t src = cp->src[k];
      sel = src == kCapSrcStart ? 0u : src == kCapSrcEnd ? 1u : 7u;
      if (src >= kCapSrcRun0 && src < kCapSrcRun0 + kCapMaxRuns) { const uint32_t op = cp->run_op[src - kCapSrcRun0]; sel = (op >> 1) < static_cast<uint32_t>(TK - 1) ? 2u + (op >> 1) : 1u; }
      off = cp->off[k];
    };
    if (cp->on == 1u && t_lane_on) { slot_of(2u * t_pr, t_sel0, t_off0); slot_of(2u * t_pr + 1u, t_sel1, t_off1); }
  }
  const uint32_t myblk = wv >> 6, myidx = wv & 63u;
  const uint32_t hw_wave = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (3 << 11)) & 15u;   // wave slot on its SIMD

  auto tpw_of = [&](uint32_t r) -> uint32_t { return r < full ? static_cast<uint32_t>(kPfTiles) : tpw_last; };
  auto unit_tile = [&](uint32_t r) -> uint64_t { return static_cast<uint64_t>(r) * kPfTiles * W + static_cast<uint64_t>(wv) * tpw_of(r); };
  auto round_units = [&](uint32_t rr) -> uint32_t { return rr < full ? W : units_last; };
  // pf_rec, 384 words of 8 bytes per round: records [128][2] = {tag << 32 | rows of the round in front of the block, tag << 32 |
  // rows of the round}, then the block sums [128] = tag << 32 | rows of the block
  auto rec_of = [&](uint32_t rr) -> uint64_t* { return a.pf_rec + static_cast<uint64_t>(rr) * kPfRecStride; };
  auto units_rsrc = [&](uint32_t rr) -> __amdgpu_buffer_rsrc_t {
    return __builtin_amdgcn_make_buffer_rsrc(a.pf_status + static_cast<uint64_t>(rr) * W, 0, static_cast<int>(round_units(rr) * 4u), 0x00020000);
  };
  // what this wave needs of round rr: the record of its block (same address in every lane) and the word of unit 64 * myblk + lane
  // (both past the L1: sc0 sc1)
  auto status_load = [&](uint32_t rr, u32x4& vr, uint32_t& vs) {
    if (CXG_PFABL >= 1) return;
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(rec_of(rr) + myblk * 2u, 0, 16, 0x00020000);
    vr = __builtin_amdgcn_raw_buffer_load_b128(rb, 0u, 0, 17);
    vs = __builtin_amdgcn_raw_buffer_load_b32(units_rsrc(rr), (myblk * 64u + static_cast<uint32_t>(lane0)) * 4u, 0, 17);
  };
  // all there?  then pre = rows of the round in front of this unit, tot = rows of the round
  auto status_reduce = [&](const u32x4& vr, uint32_t vs, uint32_t& pre, uint32_t& tot) -> bool {
    if (CXG_PFABL >= 1) { pre = wv * 64u; tot = W * 64u; return true; }
    const bool mine = static_cast<uint32_t>(lane0) < myidx;           // the units of my block in front of me
    const bool ok = vr.y == tag && vr.w == tag && (!mine || (vs >> 16) == ep);
    if (__ballot(!ok) != 0ull) return false;
    const uint32_t p = mine ? (vs & 0xFFFFu) : 0u;
    pre = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(vr.x))) +
          static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wave_inclusive_sum_fused(p)), 63));
    tot = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(vr.z)));
    return true;
  };
  // ---- duties of a leader.  duty_round = the round this wave still owes something for (block sum first, then — the round's
  // leader only — the records); duty_stage 0 = nothing owed, 1 = block sum, 2 = records.
  uint32_t duty_round = 0, duty_stage = 0;
  auto blk_leader = [&](uint32_t rr) -> bool { const uint32_t U = round_units(rr); return wv + 1u == U || (myidx == 63u && wv < U); };
  auto duty_load = [&](u32x4& dv) {
    if (duty_stage == 1u) dv.x = __builtin_amdgcn_raw_buffer_load_b32(units_rsrc(duty_round), (myblk * 64u + static_cast<uint32_t>(lane0)) * 4u, 0, 17);
    else {
      const uint32_t nblocks = (round_units(duty_round) + 63u) >> 6;
      const __amdgpu_buffer_rsrc_t rsum = __builtin_amdgcn_make_buffer_rsrc(rec_of(duty_round) + 256, 0, static_cast<int>(nblocks * 8u), 0x00020000);
      dv = __builtin_amdgcn_raw_buffer_load_b128(rsum, static_cast<uint32_t>(lane0) * 16u, 0, 17);
    }
  };
  auto duty_check = [&](const u32x4& dv) {                            // looks at what duty_load brought; publishes and moves on when complete
    const uint32_t U = round_units(duty_round);
    if (duty_stage == 1u) {
      const uint32_t first = myblk * 64u;
      const uint32_t expect = U - first < 64u ? U - first : 64u;
      const bool in = static_cast<uint32_t>(lane0) < expect;
      if (__ballot(in && (dv.x >> 16) != ep) != 0ull) return;
      const uint32_t sum = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wave_inclusive_sum_fused(in ? (dv.x & 0xFFFFu) : 0u)), 63));
      if (lane0 == 0) __hip_atomic_store(rec_of(duty_round) + 256 + myblk, (static_cast<uint64_t>(tag) << 32) | sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      duty_stage = (wv + 1u == U) ? 2u : 0u;                          // the last unit of the round also owes the records
      return;
    }
    const uint32_t nblocks = (U + 63u) >> 6;
    const uint32_t b0 = static_cast<uint32_t>(lane0) * 2u;
    const bool ok = (b0 >= nblocks || dv.y == tag) && (b0 + 1u >= nblocks || dv.w == tag);
    if (__ballot(!ok) != 0ull) return;
    const uint32_t s0 = b0 < nblocks ? dv.x : 0u, s1 = b0 + 1u < nblocks ? dv.z : 0u;
    const uint32_t incl = wave_inclusive_sum_fused(s0 + s1);
    const uint32_t total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
    const __amdgpu_buffer_rsrc_t rrec = __builtin_amdgcn_make_buffer_rsrc(rec_of(duty_round), 0, static_cast<int>(nblocks * 16u), 0x00020000);
    const u32x4 r0 = {incl - s0 - s1, tag, total, tag}, r1 = {incl - s1, tag, total, tag};
    __builtin_amdgcn_raw_buffer_store_b128(r0, rrec, static_cast<uint32_t>(lane0) * 32u, 0, 17);
    __builtin_amdgcn_raw_buffer_store_b128(r1, rrec, static_cast<uint32_t>(lane0) * 32u + 16u, 0, 17);
    duty_stage = 0u;
  };
  auto duty_finish = [&]() {                                          // no tiles left to hide behind (end of a round with the duty still open, end of the launch)
    uint32_t spins = 0;
    uint64_t t_wait = 0;
    while (duty_stage != 0u) {
      u32x4 dv = {0u, 0u, 0u, 0u};
      duty_load(dv);
      const uint32_t before = duty_stage;
      duty_check(dv);
      if (duty_stage == before) {
        if (spins++ == 0u) t_wait = pf_wait_clock();
        else if ((spins & 15u) == 0u && pf_wait_over(a, t_wait)) { if (lane0 == 0 && !pf_aborted(a)) raise_watchdog(a.err, kWdPersDuty); pf_abort(a); break; }
        __builtin_amdgcn_s_sleep(CXG_PF_SLEEP);
      }
    }
  };

  u32x4 x[4];
  uint32_t sink = 0;
  int32_t nvalid_cur = 0;
  fields_first_loads<kPre>(x, fields_window<kPre>(a.hay, a.len, unit_tile(0) * static_cast<uint64_t>(kWaveTile), true, nvalid_cur), lane, wv == 0);
  uint64_t running = 0;                                               // rows in front of the round being ordered (uniform)
  uint64_t my_total = 0;                                              // count-only: rows of this wave's units
  uint32_t fallback = 0;
  uint32_t nrows_prev = 0;
  uint32_t st_waits = 0, st_polls = 0;
  const uint64_t st_t0 = __builtin_readcyclecounter();
  uint64_t st_scan = 0;

  for (uint32_t r = 0; r <= R_me; r++) {
    const bool scan = r < R_me;
    const uint32_t par = r & 1u;
    u32x4 vr = {0u, 0u, 0u, 0u};
    uint32_t vs = 0;
    uint32_t nrows_w = 0;
    if (scan) {
      const uint32_t tpw = tpw_of(r);
      const uint64_t t0 = unit_tile(r);
      const uint64_t st_a = __builtin_readcyclecounter();
      for (uint32_t j = 0; j < tpw; j++) {
        lane = lane0;
        asm volatile("" : "+v"(lane));
        if (CXG_PF_PRIO) {
          // VALU issue on a SIMD goes to the highest priority, then to the OLDEST wave: with equal priorities the first-dispatched
          // of the six waves of a SIMD runs nearly unimpeded and the youngest gets what is left — measured: wave lives between 117
          // and 220 us for the same work (profiles/r04_pers_wave_times.txt), and a statically partitioned launch ends with its
          // slowest wave.  Rotate the priorities: (time slice + wave slot) mod 4, the same clock for all waves of a SIMD.
          const uint32_t slice = static_cast<uint32_t>(__builtin_readcyclecounter() >> CXG_PF_PRIO_SHIFT);
          switch ((slice + hw_wave) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
          }
        }
        int32_t nvalid_next = 0;
        const bool last = j + 1 == tpw;
        const bool more = !last || r + 1 < R_me;
        const uint64_t lo_next = (last ? unit_tile(r + 1) : t0 + j + 1) * static_cast<uint64_t>(kWaveTile);
        const __amdgpu_buffer_rsrc_t rnext = fields_window<kPre>(a.hay, a.len, lo_next, more, nvalid_next);
        uint32_t d0 = 0, d1 = 0, p0 = 0, p1 = 0;
        uint32_t tc0[TK - 1], tc1[TK - 1];
        if (kTrio) trio_words<TK, kCarry>(x, rnext, lane, &s_c[0][wave][0], s_cls, nvalid_cur, j != 0u, !last, d0, d1, tc0, tc1);
        else if (kLit) lit_words<kNBitmaps, kCarry>(x, rnext, lane, &s_c[0][wave][0], nvalid_cur, lr, j != 0u, !last);
        else fields_words<KD, KP, kCarry>(x, rnext, lane, s_d[wave], s_p[wave], nvalid_cur, dlo4, dhi4, plo4, phi4, sink, d0, d1, p0, p1, j != 0u, !last);
        nvalid_cur = nvalid_next;
        const bool duty = duty_stage != 0u;                           // a leader's look of this tile
        u32x4 dv = {0u, 0u, 0u, 0u};
        if (duty) duty_load(dv);
        if (last && order && r > 0) status_load(r - 1, vr, vs);       // consumed behind this tile's mathematics
        FieldsTile t;
        if (kTrio) { const TrioTile tt = trio_core<TK, TEQ, kOwn>(d0, d1, tc0, tc1); t = FieldsTile{tt.e0, tt.e1, 0u, 0u, tt.ovf}; }
        else t = kLit ? lit_core<kOwn>(&s_c[0][wave][0], lane, lr) : fields_core<K, kOwn>(d0, d1, p0, p1);
        if (kLit && kCarry && !last) lit_carry<kNBitmaps>(&s_c[0][wave][0], lane);
        if (t.ovf) fallback |= 1u;
        const uint32_t c = static_cast<uint32_t>(__popc(t.e0)) + static_cast<uint32_t>(__popc(t.e1));
        const uint32_t incl = wave_inclusive_sum_fused(c);
        const uint32_t tot = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
        if (tot != 0 && want_rows) {
          if (kTrio) trio_rows<TK, LinkT>(TrioTile{t.e0, t.e1, t.ovf}, d0, d1, lane, s_row[par][wave], s_lnk[kTrio ? par : 0][wave], nrows_w + incl - c, static_cast<uint32_t>(kPfRows),
                                          j * static_cast<uint32_t>(kWaveTile) * 0x10001u);
          else fields_rows(t, lane, s_row[par][wave], nrows_w + incl - c, [](uint32_t rr) { return min(rr, static_cast<uint32_t>(kPfRows - 1)); },
                           j * static_cast<uint32_t>(kWaveTile) * 0x10001u);
        }
        nrows_w += tot;
        if (duty) duty_check(dv);
      }
      st_scan += __builtin_readcyclecounter() - st_a;
      if (nrows_w > static_cast<uint32_t>(kPfRows)) fallback |= 16u;
      wave_lds_sync();
      bool bad = false, long_hit = false;
      if (want_rows) {
        for (uint32_t q = lane0; q < nrows_w && q < static_cast<uint32_t>(kPfRows); q += 64) {
          const uint32_t v = s_row[par][wave][q];
          const uint32_t sb = v & 0xFFFFu, eb = v >> 16;
          bad = bad || sb >= eb;
          long_hit = long_hit || (a.max_len != 0 && eb - sb > a.max_len);
        }
      }
      if (__ballot(bad) != 0ull) fallback |= 2u;
      if (__ballot(long_hit) != 0ull && lane0 == 0) raise_err(a.err, kErrLongMatch);
      my_total += nrows_w;
      if (order && CXG_PFABL < 2) {                                   // publish the unit's row count; a leader takes on the round's duty
        if (lane0 == 0) __hip_atomic_store(a.pf_status + static_cast<uint64_t>(r) * W + wv, tag | nrows_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (CXG_PFABL < 1 && blk_leader(r)) {
          duty_finish();                                              // (still owing for round r - 1: everybody waits for that)
          duty_round = r; duty_stage = 1u;
        }
      }
    }
    if (order && !(scan && r + 1 < R_me)) duty_finish();              // no further round to hide behind: the records may be waiting for this wave
    if (order && r > 0) {
      if (!scan) status_load(r - 1, vr, vs);
      // ---- order and write the rows of round r - 1
      const uint32_t rp = r - 1u, parp = rp & 1u;
      uint32_t pre = 0, tot = 0, spins = 0;
      uint64_t t_wait = 0;
      while (!status_reduce(vr, vs, pre, tot)) {                      // something of the round was not there yet
        if (spins++ == 0u) t_wait = pf_wait_clock();
        else if ((spins & 15u) == 0u && pf_wait_over(a, t_wait)) { if (lane0 == 0 && !pf_aborted(a)) raise_watchdog(a.err, kWdPersRecord); pf_abort(a); break; }
        __builtin_amdgcn_s_sleep(CXG_PF_SLEEP);
        status_load(rp, vr, vs);
      }
      st_waits += spins != 0u ? 1u : 0u; st_polls += spins;           // CXG_VERBOSE statistics, left per wave at the end (an atomic per unit on two words cost 0.8 ms)
      const uint64_t base = running + pre;
      running += tot;
      if (a.out != nullptr && CXG_PFABL < 3) {
        const int64_t origin = (a.u32_rows ? 0 : a.base) + static_cast<int64_t>(unit_tile(rp) * static_cast<uint64_t>(kWaveTile)) - kPre;
        const uint32_t n = nrows_prev < static_cast<uint32_t>(kPfRows) ? nrows_prev : static_cast<uint32_t>(kPfRows);
        if (kTrio) {                                                  // capture rows (or spans): k_scan_trio_wave's epilogue, a power of two of lanes per row
          for (uint32_t i = lane0; i < (n << t_lsh); i += 64) {
            const uint32_t rr = i >> t_lsh;
            if (t_lane_on && base + rr < a.cap) {
              const uint32_t w0 = s_row[parp][wave][rr], w1 = s_lnk[kTrio ? parp : 0][wave][rr];
              const int64_t ps = origin + (w0 & 0xFFFFu), pe = origin + (w0 >> 16);
              auto pos_of = [&](uint32_t sel) -> int64_t { return sel == 0u ? ps : sel == 1u ? pe : ps + ((w1 >> (8u * (sel - 2u))) & 0xFFu); };
              store_pair_nt(a.out + (base + rr) * a.row_width + 2u * t_pr, t_sel0 == 7u ? -1 : pos_of(t_sel0) + t_off0, t_sel1 == 7u ? -1 : pos_of(t_sel1) + t_off1);
            }
          }
        } else
        for (uint32_t i = lane0; i < n; i += 64) {
          if (base + i < a.cap) {
            const uint32_t v = s_row[parp][wave][i];
            if (a.u32_rows) store_pair32_nt(reinterpret_cast<uint32_t*>(a.out) + (base + i) * 2u, static_cast<uint32_t>(origin + (v & 0xFFFFu)), static_cast<uint32_t>(origin + (v >> 16)));
            else store_pair_nt(a.out + (base + i) * a.row_width, origin + (v & 0xFFFFu), origin + (v >> 16));
          }
        }
      }
    }
    nrows_prev = nrows_w;
  }
  if (fallback != 0 && lane0 == 0) raise_err(a.err, 8u | (fallback << 8));
  if (!order) { if (lane0 == 0) a.status[wv] = my_total; return; }    // k_sum_counts adds the W words up
  if (lane0 == 0) {
    a.pf_stats[wv] = (static_cast<uint64_t>(st_waits) << 32) | st_polls;
    a.pf_stats[8192 + wv] = __builtin_readcyclecounter() - st_t0;     // s_memtime ticks of this wave's life (~2.2 GHz in a busy kernel)
    a.pf_stats[16384 + wv] = st_scan;                                 // ... of which inside the tile loops
    a.pf_stats[24576 + wv] = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11)) | (static_cast<uint64_t>(__builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11))) << 32);
  }
  if (wv == 0 && lane0 == 0) *a.total = running;                      // wave 0 has a unit in every round
}

#else
// LIT: 0 = a fields program (K, KD, KP as above); 2..4 = a literal over that many distinct bytes (lit_core; K, KD, KP unused);
// 16 + K' + 8 EQ = TRIO mode (round 5): run(F) (byte(c_i) run(F)){K'-1} programs with their capture slots, k_scan_trio_wave's tile
// mathematics (trio_core<K', EQ>) and row epilogue on this kernel's grid and protocol (BASELINE configs[4], `(\w+)@(\w+)\.(\w+)`).
template <int K, int KD, int KP, int LIT = 0>
__global__ __launch_bounds__(kThreads, (LIT >= 16 ? 4 : LIT >= 3 ? 5 : CXG_PF_OCC)) void k_scan_fields_pers(ScanArgs a) {   // ticketed (round 6)   // (three / four bitmaps: 27 / 29 KB of LDS per workgroup — six do not fit a CU)
  constexpr bool kTrio = LIT >= 16;
  constexpr int TK = kTrio ? ((LIT - 16) & 7) : 2;                    // fields of a TRIO program
  constexpr bool TEQ = kTrio && ((LIT - 16) >> 3) != 0;
  constexpr bool kLit = LIT >= 2 && LIT <= 4;
  typedef typename std::conditional<TK == 4, uint32_t, uint16_t>::type LinkT;
  constexpr int kNBitmaps = kTrio ? TK : (kLit ? LIT : 2);
  __shared__ LinkT s_lnk[kTrio ? 2 : 1][kWavesPerBlock][kTrio ? kPfRows : 1];   // TRIO: the links of a parked row as distances from its start
  __shared__ uint8_t s_cls[kTrio ? 256 : 4];                          // TRIO: byte -> class flags (bit 0 F, bit i + 1 the separator of link i)
  __shared__ __attribute__((aligned(16))) uint64_t s_c[kNBitmaps][kWavesPerBlock][64 + 4];   // class bitmaps of the wave's window (+ 4 dump words: CARRY)
  uint64_t (*const s_d)[64 + 4] = s_c[0];
  uint64_t (*const s_p)[64 + 4] = s_c[1];
  __shared__ uint32_t s_row[2][kWavesPerBlock][kPfRows];                         // rows of round r and r - 1: start | end << 16, offsets from the unit's first byte - kPre

  constexpr bool kCarry = CXG_PF_CARRY != 0;
  constexpr int kPre = kCarry ? kFPrePers : kFPre;
  constexpr unsigned long long kOwn = kCarry ? kFOwnPers : kFOwn;
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = lane0;
  const uint32_t W = gridDim.x * static_cast<uint32_t>(kWavesPerBlock);
  const uint32_t wv = blockIdx.x * static_cast<uint32_t>(kWavesPerBlock) + static_cast<uint32_t>(wave);   // (the wave's slot in the statistics and the count-only sums — NOT what it scans)
  // Units, in haystack order (launch_pers_inst): U9 of kPfTiles tiles (the last of them may be shorter), then U3 of three, then U1 of
  // one — the tail in small units so that the waves end within a tile of each other.  Unit u belongs to round u / W, block
  // (u % W) / 64 of that round: the words and records of the ordering protocol are indexed by the unit, whoever scans it.
  const uint32_t U9 = a.pf_full, U3 = a.pf_tpw_last, U1 = a.pf_units_last, U = U9 + U3 + U1;
  const uint64_t nwt_all = (a.len + static_cast<uint64_t>(kWaveTile) - 1) / static_cast<uint64_t>(kWaveTile);
  const uint64_t T9 = nwt_all - 3ull * U3 - U1;
  auto unit_first = [&](uint32_t u) -> uint64_t {
    return u < U9 ? static_cast<uint64_t>(u) * kPfTiles : (u < U9 + U3 ? T9 + 3ull * (u - U9) : T9 + 3ull * U3 + (u - U9 - U3));
  };
  auto unit_tiles = [&](uint32_t u) -> uint32_t {
    return u < U9 ? (u + 1u == U9 ? static_cast<uint32_t>(T9 - static_cast<uint64_t>(U9 - 1u) * kPfTiles) : static_cast<uint32_t>(kPfTiles)) : (u < U9 + U3 ? 3u : 1u);
  };
  // ---- tickets.  A wave takes the next unit from a counter — so whoever holds unit u knows that every unit below u of that counter is
  // held by a wave that RUNS or has finished, and every wait of the protocol below is for a smaller unit: forward progress without
  // the whole grid being resident, beside any other kernel.  One counter serialises at ~20 returning atomics per microsecond (a
  // 64-GiB scan asks for 190: scripts/microbench/ticket_atomics.hip — 8 counters cost 25 % of a streaming kernel, 64 cost 2 %), so
  // there are NC <= 64 of them, a cache line apart, counter c = workgroup & (NC - 1) handing out the units c, c + NC, ...: the
  // workgroups dispatched first cover all counters.  The counters of launch epoch e are block e & 31 of a ring; this launch clears
  // the block 16 epochs ahead (nothing in flight uses it).
  const uint32_t ncnt = a.pf_ncounters;                               // a power of two <= the number of workgroups
  // (Round 6, second form: a wave asks the counters IN ROTATION — with a fixed counter per workgroup the counters advance at the speeds
  // of their workgroups, the order of the units drifts apart from the order in time, and every wave waits for the slowest counter
  // at every unit: 16 GiB 5.9 ms against 3.3, profiles/r06_c5_tail_ab.txt.)
  uint32_t cme = (blockIdx.x * static_cast<uint32_t>(kWavesPerBlock) + static_cast<uint32_t>(wave)) & (ncnt - 1u);
  uint32_t* const ctr0 = a.pf_ticket + (a.pf_epoch & 31u) * (64u * kPfCtrStride);
  auto per_ctr = [&](uint32_t c) -> uint32_t { return (U + ncnt - 1u - c) / ncnt; };
  uint32_t casked = cme;                                              // the counter of the ticket in flight
  auto claim_issue = [&]() -> uint32_t {                              // lane 0's ticket of the next counter of the wave's rotation (valid in lane 0)
    cme = (cme + 1u) & (ncnt - 1u);
    casked = cme;
    uint32_t t = 0;
    if (lane0 == 0) t = __hip_atomic_fetch_add(ctr0 + cme * kPfCtrStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t;
  };
  auto claim_resolve = [&](uint32_t t_raw) -> uint32_t {              // the unit of that ticket; that counter exhausted: two neighbours are asked, then the wave is done; U: nothing left
    const uint32_t t = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(t_raw)));
    if (t < per_ctr(casked)) return t * ncnt + casked;
    for (uint32_t kx = 1; kx < 3u && kx < ncnt; kx++) {
      const uint32_t c = (casked + kx) & (ncnt - 1u);
      uint32_t t2 = 0;
      if (lane0 == 0) t2 = __hip_atomic_fetch_add(ctr0 + c * kPfCtrStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t2 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(t2)));
      if (t2 < per_ctr(c)) return t2 * ncnt + c;
    }
    return U;
  };
  const uint32_t tk_first = claim_issue();
  if (blockIdx.x == 0 && tid < 64) a.pf_ticket[(((a.pf_epoch + 16u) & 31u) * 64u + static_cast<uint32_t>(tid)) * kPfCtrStride] = 0u;
  if (kTrio) {                                                        // (in front of the first early return: every wave of the workgroup reaches the barrier)
    const ChainAux* tch = reinterpret_cast<const ChainAux*>(a.chain);
    const uint32_t b = static_cast<uint32_t>(tid);
    uint32_t f = chain_class_has(*tch, 0, b) ? 1u : 0u;
    for (int i = 0; i < TK - 1; i++) f |= chain_class_has(*tch, tch->op_cls[2 * i + 1], b) ? (2u << i) : 0u;
    s_cls[tid] = static_cast<uint8_t>(f);
  }
  __syncthreads();
  uint32_t u_cur = claim_resolve(tk_first);
  if (u_cur >= U) { if (a.count_sum != 0u && lane0 == 0) a.status[wv] = 0; return; }   // (more waves than units)
  const ChainAux* gch = reinterpret_cast<const ChainAux*>(a.chain);
  const uint32_t dlo4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gch->cls_lo[0] * 0x01010101u)));
  const uint32_t dhi4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((0x7Fu - gch->cls_hi[0]) * 0x01010101u)));
  const uint32_t plo4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gch->cls_lo[1] * 0x01010101u)));
  const uint32_t phi4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((0x7Fu - gch->cls_hi[1]) * 0x01010101u)));
  LitRegs lr;
  if (kLit) {
    lr.m = gch->nops; lr.nc = gch->ncls; lr.cls2_lo = gch->cls2_lo; lr.cls2_hi = gch->cls2_hi;
#pragma unroll
    for (int c = 0; c < 4; c++) lr.b4[c] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gch->cls_lo[c] * 0x01010101u)));
  }
  const uint32_t ep = a.pf_epoch;
  const uint32_t tag = ep << 16;
  const bool want_rows = a.out != nullptr || a.max_len != 0;
  const bool order = a.count_sum == 0u;                               // count-only calls need no place in the output
  // TRIO: which two slots of a row this lane writes and what they are made of (k_scan_trio_wave's epilogue): sel 0 start, 1 end,
  // 2 + i the end of run i (link i), 7 unset
  uint32_t t_lsh = 0, t_pr = 0, t_sel0 = 0, t_sel1 = 1; int32_t t_off0 = 0, t_off1 = 0; bool t_lane_on = true;
  if (kTrio) {
    const ChainCaps* cp = reinterpret_cast<const ChainCaps*>(a.caps);
    const uint32_t npairs = a.row_width >> 1;
    t_lsh = npairs <= 1u ? 0u : 32u - static_cast<uint32_t>(__builtin_clz(npairs - 1u));
    t_pr = static_cast<uint32_t>(lane0) & ((1u << t_lsh) - 1u);
    t_lane_on = t_pr < npairs;
    auto slot_of = [&](uint32_t k, uint32_t& sel, int32_t& off) {
      const uint32_t src = cp->src[k];
      sel = src == kCapSrcStart ? 0u : src == kCapSrcEnd ? 1u : 7u;
      if (src >= kCapSrcRun0 && src < kCapSrcRun0 + kCapMaxRuns) { const uint32_t op = cp->run_op[src - kCapSrcRun0]; sel = (op >> 1) < static_cast<uint32_t>(TK - 1) ? 2u + (op >> 1) : 1u; }
      off = cp->off[k];
    };
    if (cp->on == 1u && t_lane_on) { slot_of(2u * t_pr, t_sel0, t_off0); slot_of(2u * t_pr + 1u, t_sel1, t_off1); }
  }
  const uint32_t hw_wave = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (3 << 11)) & 15u;   // wave slot on its SIMD

  // ---- ordering.  Units are grouped in BLOCKS of 64 consecutive units (block g = units 64 g .. 64 g + 63, whoever scanned them).  The
  // wave that holds a block's LAST unit owes the block: first its sum (the 64 unit words), published as an aggregate, then — a decoupled
  // look-back over the block words in front, 64 per look — the rows up to the block's end, published as an inclusive word.  A unit's
  // rows start at  inclusive(block in front) + the units of its own block below it.  Every word anybody waits for belongs to a SMALLER
  // unit, and a leader looks at its duty once per tile of the unit it scans meanwhile: nobody spins while it has tiles.
  // pf_rec[g] = epoch << 48 | inclusive << 47 | rows (47 bits)
  const uint64_t ep48 = static_cast<uint64_t>(a.pf_epoch) << 48;
  constexpr uint64_t kBwIncl = 1ull << 47, kBwRows = kBwIncl - 1ull;
  uint64_t* const bw = a.pf_rec;
  const uint32_t nblk = (U + 63u) >> 6;
  const __amdgpu_buffer_rsrc_t units_rs = __builtin_amdgcn_make_buffer_rsrc(a.pf_status, 0, static_cast<int>(U * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t blocks_rs = __builtin_amdgcn_make_buffer_rsrc(bw, 0, static_cast<int>(nblk * 8u), 0x00020000);
  struct WordPair { uint32_t lo, hi; };
  // what the rows of unit (block g, index idx) wait for: the inclusive word of block g - 1 (same address in every lane) and the words of
  // the units of block g below it (both past the L1: sc0 sc1)
  auto status_load = [&](uint32_t g, WordPair& vr, uint32_t& vs) {
    if (CXG_PFABL >= 1) return;
    vs = __builtin_amdgcn_raw_buffer_load_b32(units_rs, (g * 64u + static_cast<uint32_t>(lane0)) * 4u, 0, 17);
    if (g != 0u) {                                                    // ONE 8-byte load: the word changes from aggregate to inclusive, two dword loads could see half of each
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(blocks_rs, (g - 1u) * 8u, 0, 17);
      vr.lo = w.x; vr.hi = w.y;
    }
  };
  // all there?  then base = rows in front of the unit
  auto status_reduce = [&](const WordPair& vr, uint32_t vs, uint32_t g, uint32_t idx, uint64_t& base) -> bool {
    if (CXG_PFABL >= 1) { base = static_cast<uint64_t>(wv) * 64u; return true; }
    const bool mine = static_cast<uint32_t>(lane0) < idx;             // the units of the block in front of this one
    const bool ok = (g == 0u || (vr.hi >> 15) == ((ep << 1) | 1u)) && (!mine || (vs >> 16) == ep);
    if (__ballot(!ok) != 0ull) return false;
    const uint32_t p = mine ? (vs & 0xFFFFu) : 0u;
    const uint64_t inc = g == 0u ? 0ull : (((static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(vr.hi)))) << 32) |
                                             static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(vr.lo)))) & kBwRows);
    base = inc + static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wave_inclusive_sum_fused(p)), 63));
    return true;
  };
  // ---- duties of a block's leader.  duty_stage 0 = nothing owed, 1 = the block's sum, 2 = the look-back.
  uint32_t duty_g = 0, duty_stage = 0, duty_look = 0;
  uint64_t duty_acc = 0;                                              // own sum + the aggregates looked back over so far
  auto blk_leader = [&](uint32_t u) -> bool { return (u & 63u) == 63u || u + 1u == U; };   // the LAST unit of a block: it waits for smaller units only
  struct DutyWords { uint32_t lo, hi; };
  auto duty_load = [&](DutyWords& dw) {
    if (duty_stage == 1u) dw.lo = __builtin_amdgcn_raw_buffer_load_b32(units_rs, (duty_g * 64u + static_cast<uint32_t>(lane0)) * 4u, 0, 17);
    else if (static_cast<uint32_t>(lane0) <= duty_look) {             // lane l: block duty_look - l (one 8-byte load each: see status_load)
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(blocks_rs, (duty_look - static_cast<uint32_t>(lane0)) * 8u, 0, 17);
      dw.lo = w.x; dw.hi = w.y;
    }
  };
  auto duty_publish = [&](uint64_t rows_incl) {                       // the look-back is through: rows up to the end of block duty_g
    if (lane0 == 0) {
      __hip_atomic_store(bw + duty_g, ep48 | kBwIncl | rows_incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (duty_g + 1u == nblk) *a.total = rows_incl;                  // the whole haystack's rows
    }
    duty_stage = 0u;
  };
  auto duty_check = [&](const DutyWords& dw) {                        // looks at what duty_load brought; publishes and moves on when complete
    if (duty_stage == 1u) {
      const uint32_t first = duty_g * 64u;
      const uint32_t expect = U - first < 64u ? U - first : 64u;
      const bool in = static_cast<uint32_t>(lane0) < expect;
      if (__ballot(in && (dw.lo >> 16) != ep) != 0ull) return;
      const uint32_t sum = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wave_inclusive_sum_fused(in ? (dw.lo & 0xFFFFu) : 0u)), 63));
      duty_acc = sum;
      if (duty_g == 0u) { duty_publish(sum); return; }
      if (lane0 == 0) __hip_atomic_store(bw + duty_g, ep48 | sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // aggregate: later leaders need not wait for this look-back
      duty_stage = 2u; duty_look = duty_g - 1u;
      return;
    }
    // lanes 0 .. min(63, duty_look): the blocks duty_look, duty_look - 1, ...; the nearest inclusive word ends the look-back
    const bool have = static_cast<uint32_t>(lane0) <= duty_look;
    const bool ready = !have || (dw.hi >> 16) == ep;
    const unsigned long long nr = __ballot(!ready);
    const uint32_t nready = nr ? static_cast<uint32_t>(__builtin_ctzll(nr)) : 64u;
    const unsigned long long im = __ballot(have && ready && ((dw.hi >> 15) & 1u) != 0u) & (nready >= 64u ? ~0ull : ((1ull << nready) - 1ull));
    const uint32_t ntake = im ? static_cast<uint32_t>(__builtin_ctzll(im)) + 1u : (nready == 64u || nready > duty_look ? (duty_look + 1u < 64u ? duty_look + 1u : 64u) : 0u);
    if (ntake == 0u) return;                                          // a block in front has not published anything yet: next tile
    const uint64_t v = (have && static_cast<uint32_t>(lane0) < ntake) ? (((static_cast<uint64_t>(dw.hi) << 32) | dw.lo) & kBwRows) : 0ull;
    uint32_t vlo = static_cast<uint32_t>(v), vhi = static_cast<uint32_t>(v >> 32);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {                               // (64-bit wave sum: block words hold up to 47 bits)
      const uint32_t olo = static_cast<uint32_t>(__shfl_xor(static_cast<int>(vlo), d, 64)), ohi = static_cast<uint32_t>(__shfl_xor(static_cast<int>(vhi), d, 64));
      const uint64_t sum2 = ((static_cast<uint64_t>(vhi) << 32) | vlo) + ((static_cast<uint64_t>(ohi) << 32) | olo);
      vlo = static_cast<uint32_t>(sum2); vhi = static_cast<uint32_t>(sum2 >> 32);
    }
    duty_acc += (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(vhi)))) << 32) | static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(vlo)));
    if (im != 0ull || ntake > duty_look) { duty_publish(duty_acc); return; }   // reached an inclusive word, or the haystack's first block
    duty_look -= ntake;                                               // 64 aggregates and no inclusive word among them: further back on the next look
  };
  auto duty_finish = [&]() {                                          // no tiles left to hide behind (end of a round with the duty still open, end of the launch)
    uint32_t spins = 0;
    uint64_t t_wait = 0;
    while (duty_stage != 0u) {
      DutyWords dv = {0u, 0u};
      duty_load(dv);
      const uint32_t before = duty_stage, before_look = duty_look;
      duty_check(dv);
      if (duty_stage == before && duty_look == before_look) {
        if (spins++ == 0u) t_wait = pf_wait_clock();
        else if ((spins & 15u) == 0u && pf_wait_over(a, t_wait)) { if (lane0 == 0 && !pf_aborted(a)) raise_watchdog(a.err, kWdPersDuty); pf_abort(a); break; }
        __builtin_amdgcn_s_sleep(CXG_PF_SLEEP);
      }
    }
  };

  u32x4 x[4];
  uint32_t sink = 0;
  int32_t nvalid_cur = 0;
  fields_first_loads<kPre>(x, fields_window<kPre>(a.hay, a.len, unit_first(u_cur) * static_cast<uint64_t>(kWaveTile), true, nvalid_cur), lane, u_cur == 0u);
  uint64_t my_total = 0;                                              // count-only: rows of this wave's units
  uint32_t fallback = 0;
  uint32_t st_waits = 0, st_polls = 0;
  const uint64_t st_t0 = __builtin_readcyclecounter();
  uint64_t st_scan = 0;
  // the unit whose rows are parked (written one unit later: their base is ready by then, nobody waits)
  bool have_prev = false;
  uint32_t prev_blk = 0, prev_idx = 0, prev_rows = 0, prev_par = 0;
  uint64_t prev_first = 0;
  uint32_t par = 0;
  auto wave_sum64 = [&](uint64_t v) -> uint64_t {                      // (block words hold up to 47 bits)
    uint32_t vlo = static_cast<uint32_t>(v), vhi = static_cast<uint32_t>(v >> 32);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint32_t olo = static_cast<uint32_t>(__shfl_xor(static_cast<int>(vlo), d, 64)), ohi = static_cast<uint32_t>(__shfl_xor(static_cast<int>(vhi), d, 64));
      const uint64_t s2 = ((static_cast<uint64_t>(vhi) << 32) | vlo) + ((static_cast<uint64_t>(ohi) << 32) | olo);
      vlo = static_cast<uint32_t>(s2); vhi = static_cast<uint32_t>(s2 >> 32);
    }
    return (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(vhi)))) << 32) | static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(vlo)));
  };
  auto write_prev = [&](WordPair vr, uint32_t vs) {                     // order and write the rows of the parked unit
    uint64_t base = 0;
    uint32_t spins = 0;
    if (!status_reduce(vr, vs, prev_blk, prev_idx, base)) {
      // The inclusive word of the block in front was not there when this unit's last tile asked for it.  Do not wait for that block's
      // leader: look back over the block words in front oneself, 64 per look — aggregates add up, the nearest inclusive word ends the
      // walk (decoupled look-back; the leaders' inclusive words only keep these walks short).  What this still waits for are units
      // that have not been scanned yet: the words of the own block below this unit, and the aggregate of a block whose units are
      // still being scanned.
      uint64_t t_wait = 0;
      uint64_t acc = 0;
      uint32_t own = 0, look = prev_blk;                               // blocks 0 .. look - 1 are not accounted for yet
      bool have_own = false;
      for (;;) {
        if (!have_own) {
          const bool mine = static_cast<uint32_t>(lane0) < prev_idx;
          if (__ballot(mine && (vs >> 16) != ep) == 0ull) {
            own = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wave_inclusive_sum_fused(mine ? (vs & 0xFFFFu) : 0u)), 63));
            have_own = true;
          }
        }
        if (look != 0u) {
          typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
          const bool have = static_cast<uint32_t>(lane0) < look;      // lane l: block look - 1 - l
          u32x2 w = {0u, 0u};
          if (have) w = __builtin_amdgcn_raw_buffer_load_b64(blocks_rs, (look - 1u - static_cast<uint32_t>(lane0)) * 8u, 0, 17);
          const bool ready = !have || (w.y >> 16) == ep;
          const unsigned long long nr = __ballot(!ready);
          const uint32_t nready = nr ? static_cast<uint32_t>(__builtin_ctzll(nr)) : 64u;
          const unsigned long long im = __ballot(have && ready && ((w.y >> 15) & 1u) != 0u) & (nready >= 64u ? ~0ull : ((1ull << nready) - 1ull));
          const uint32_t ntake = im ? static_cast<uint32_t>(__builtin_ctzll(im)) + 1u : (nready < look ? nready : look);
          if (ntake != 0u) {
            acc += wave_sum64((have && static_cast<uint32_t>(lane0) < ntake) ? (((static_cast<uint64_t>(w.y) << 32) | w.x) & kBwRows) : 0ull);
            look = im ? 0u : look - ntake;
          }
        }
        if (have_own && look == 0u) break;
        if (spins++ == 0u) t_wait = pf_wait_clock();
        else if ((spins & 15u) == 0u && pf_wait_over(a, t_wait)) { if (lane0 == 0 && !pf_aborted(a)) raise_watchdog(a.err, kWdPersRecord); pf_abort(a); break; }
        if (duty_stage != 0u) { DutyWords dv = {0u, 0u}; duty_load(dv); duty_check(dv); }   // (a leader that waits keeps looking at its own duty)
        __builtin_amdgcn_s_sleep(CXG_PF_SLEEP);
        if (!have_own) vs = __builtin_amdgcn_raw_buffer_load_b32(units_rs, (prev_blk * 64u + static_cast<uint32_t>(lane0)) * 4u, 0, 17);
      }
      base = acc + own;
    }
    st_waits += spins != 0u ? 1u : 0u; st_polls += spins;             // CXG_VERBOSE statistics, left per wave at the end
    if (a.out != nullptr && CXG_PFABL < 3) {
      const int64_t origin = (a.u32_rows ? 0 : a.base) + static_cast<int64_t>(prev_first * static_cast<uint64_t>(kWaveTile)) - kPre;
      const uint32_t n = prev_rows < static_cast<uint32_t>(kPfRows) ? prev_rows : static_cast<uint32_t>(kPfRows);
      if (kTrio) {                                                    // capture rows (or spans): k_scan_trio_wave's epilogue, a power of two of lanes per row
        for (uint32_t i = lane0; i < (n << t_lsh); i += 64) {
          const uint32_t rr = i >> t_lsh;
          if (t_lane_on && base + rr < a.cap) {
            const uint32_t w0 = s_row[prev_par][wave][rr], w1 = s_lnk[kTrio ? prev_par : 0][wave][rr];
            const int64_t ps = origin + (w0 & 0xFFFFu), pe = origin + (w0 >> 16);
            auto pos_of = [&](uint32_t sel) -> int64_t { return sel == 0u ? ps : sel == 1u ? pe : ps + ((w1 >> (8u * (sel - 2u))) & 0xFFu); };
            store_pair_nt(a.out + (base + rr) * a.row_width + 2u * t_pr, t_sel0 == 7u ? -1 : pos_of(t_sel0) + t_off0, t_sel1 == 7u ? -1 : pos_of(t_sel1) + t_off1);
          }
        }
      } else
      for (uint32_t i = lane0; i < n; i += 64) {
        if (base + i < a.cap) {
          const uint32_t v = s_row[prev_par][wave][i];
          if (a.u32_rows) store_pair32_nt(reinterpret_cast<uint32_t*>(a.out) + (base + i) * 2u, static_cast<uint32_t>(origin + (v & 0xFFFFu)), static_cast<uint32_t>(origin + (v >> 16)));
          else store_pair_nt(a.out + (base + i) * a.row_width, origin + (v & 0xFFFFu), origin + (v >> 16));
        }
      }
    }
  };

  for (;;) {
    const uint32_t blk = u_cur >> 6, idx = u_cur & 63u;
    WordPair vr = {0u, 0u};
    uint32_t vs = 0;
    uint32_t nrows_w = 0;
    const uint32_t tpw = unit_tiles(u_cur);
    const uint64_t t0 = unit_first(u_cur);
    const uint32_t tk_raw = claim_issue();                            // the ticket behind this unit: asked for now, looked at in front of the unit's last tile
    uint32_t u_next = U;
    {
      const uint64_t st_a = __builtin_readcyclecounter();
      for (uint32_t j = 0; j < tpw; j++) {
        lane = lane0;
        asm volatile("" : "+v"(lane));
        if (CXG_PF_PRIO) {
          // VALU issue on a SIMD goes to the highest priority, then to the OLDEST wave: with equal priorities the first-dispatched
          // of the six waves of a SIMD runs nearly unimpeded and the youngest gets what is left (profiles/r04_pers_wave_times.txt).
          // Rotate the priorities: (time slice + wave slot) mod 4, the same clock for all waves of a SIMD.
          const uint32_t slice = static_cast<uint32_t>(__builtin_readcyclecounter() >> CXG_PF_PRIO_SHIFT);
          switch ((slice + hw_wave) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
          }
        }
        int32_t nvalid_next = 0;
        const bool last = j + 1 == tpw;
        if (last) u_next = claim_resolve(tk_raw);
        const bool more = !last || u_next < U;
        const uint64_t lo_next = (last ? unit_first(u_next < U ? u_next : 0u) : t0 + j + 1) * static_cast<uint64_t>(kWaveTile);
        const __amdgpu_buffer_rsrc_t rnext = fields_window<kPre>(a.hay, a.len, lo_next, more, nvalid_next);
        uint32_t d0 = 0, d1 = 0, p0 = 0, p1 = 0;
        uint32_t tc0[TK - 1], tc1[TK - 1];
        if (kTrio) trio_words<TK, kCarry>(x, rnext, lane, &s_c[0][wave][0], s_cls, nvalid_cur, j != 0u, !last, d0, d1, tc0, tc1);
        else if (kLit) lit_words<kNBitmaps, kCarry>(x, rnext, lane, &s_c[0][wave][0], nvalid_cur, lr, j != 0u, !last);
        else fields_words<KD, KP, kCarry>(x, rnext, lane, s_d[wave], s_p[wave], nvalid_cur, dlo4, dhi4, plo4, phi4, sink, d0, d1, p0, p1, j != 0u, !last);
        nvalid_cur = nvalid_next;
        const bool duty = duty_stage != 0u;                           // a leader's look of this tile
        DutyWords dv = {0u, 0u};
        if (duty) duty_load(dv);
        if (last && order && have_prev) status_load(prev_blk, vr, vs);   // consumed behind this tile's mathematics
        FieldsTile t;
        if (kTrio) { const TrioTile tt = trio_core<TK, TEQ, kOwn>(d0, d1, tc0, tc1); t = FieldsTile{tt.e0, tt.e1, 0u, 0u, tt.ovf}; }
        else t = kLit ? lit_core<kOwn>(&s_c[0][wave][0], lane, lr) : fields_core<K, kOwn>(d0, d1, p0, p1);
        if (kLit && kCarry && !last) lit_carry<kNBitmaps>(&s_c[0][wave][0], lane);
        if (t.ovf) fallback |= 1u;
        const uint32_t c = static_cast<uint32_t>(__popc(t.e0)) + static_cast<uint32_t>(__popc(t.e1));
        const uint32_t incl = wave_inclusive_sum_fused(c);
        const uint32_t tot = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
        if (tot != 0 && want_rows) {
          if (kTrio) trio_rows<TK, LinkT>(TrioTile{t.e0, t.e1, t.ovf}, d0, d1, lane, s_row[par][wave], s_lnk[kTrio ? par : 0][wave], nrows_w + incl - c, static_cast<uint32_t>(kPfRows),
                                          j * static_cast<uint32_t>(kWaveTile) * 0x10001u);
          else fields_rows(t, lane, s_row[par][wave], nrows_w + incl - c, [](uint32_t rr) { return min(rr, static_cast<uint32_t>(kPfRows - 1)); },
                           j * static_cast<uint32_t>(kWaveTile) * 0x10001u);
        }
        nrows_w += tot;
        if (duty) duty_check(dv);
      }
      st_scan += __builtin_readcyclecounter() - st_a;
      if (nrows_w > static_cast<uint32_t>(kPfRows)) fallback |= 16u;
      wave_lds_sync();
      bool bad = false, long_hit = false;
      if (want_rows) {
        for (uint32_t q = lane0; q < nrows_w && q < static_cast<uint32_t>(kPfRows); q += 64) {
          const uint32_t v = s_row[par][wave][q];
          const uint32_t sb = v & 0xFFFFu, eb = v >> 16;
          bad = bad || sb >= eb;
          long_hit = long_hit || (a.max_len != 0 && eb - sb > a.max_len);
        }
      }
      if (__ballot(bad) != 0ull) fallback |= 2u;
      if (__ballot(long_hit) != 0ull && lane0 == 0) raise_err(a.err, kErrLongMatch);
      my_total += nrows_w;
      if (order && CXG_PFABL < 2) {                                   // publish the unit's row count; the last unit of a block takes on its duty
        if (lane0 == 0) __hip_atomic_store(a.pf_status + u_cur, tag | nrows_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (CXG_PFABL < 1 && blk_leader(u_cur)) {
          duty_finish();                                              // (still owing for an earlier block: everybody behind it waits for that)
          duty_g = blk; duty_stage = 1u;
          if (CXG_PF_EAGER) duty_finish();                            // (A/B: the leader settles its block before it scans on)
          else {                                                      // the block's AGGREGATE at once — every look-back over this block needs it; the inclusive word behind the coming tiles
            uint32_t sp2 = 0;
            uint64_t tw2 = 0;
            while (duty_stage == 1u) {
              DutyWords dv = {0u, 0u};
              duty_load(dv);
              duty_check(dv);
              if (duty_stage != 1u) break;
              if (sp2++ == 0u) tw2 = pf_wait_clock();
              else if ((sp2 & 15u) == 0u && pf_wait_over(a, tw2)) { if (lane0 == 0 && !pf_aborted(a)) raise_watchdog(a.err, kWdPersDuty); pf_abort(a); break; }
              __builtin_amdgcn_s_sleep(CXG_PF_SLEEP);
            }
          }
        }
      }
    }
    if (order && u_next >= U) duty_finish();                          // no further unit to hide behind: a record may be waiting for this wave
    if (order && have_prev) write_prev(vr, vs);
    have_prev = true;
    prev_blk = blk; prev_idx = idx; prev_rows = nrows_w; prev_par = par; prev_first = t0;
    par ^= 1u;
    if (u_next >= U) break;
    u_cur = u_next;
  }
  if (order) {                                                        // the rows of the wave's last unit
    WordPair vr = {0u, 0u};
    uint32_t vs = 0;
    status_load(prev_blk, vr, vs);
    write_prev(vr, vs);
  }
  if (fallback != 0 && lane0 == 0) raise_err(a.err, 8u | (fallback << 8));
  if (!order) { if (lane0 == 0) a.status[wv] = my_total; return; }    // k_sum_counts adds the W words up
  if (lane0 == 0) {
    a.pf_stats[wv] = (static_cast<uint64_t>(st_waits) << 32) | st_polls;
    a.pf_stats[8192 + wv] = __builtin_readcyclecounter() - st_t0;     // s_memtime ticks of this wave's life (~2.2 GHz in a busy kernel)
    a.pf_stats[16384 + wv] = st_scan;                                 // ... of which inside the tile loops
    a.pf_stats[24576 + wv] = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11)) | (static_cast<uint64_t>(__builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11))) << 32);
  }
}

#endif
// Does the chain have the shape this kernel evaluates?  run(0) (byte(1) run(0)){K-1}, two classes of one range each,
// disjoint, K = 2..4, no restart check.  Returns K, else 0.
int fields_shape(const ChainAux& c) {
  if (c.ncls != 2 || (c.nops & 1u) == 0 || c.nops < 3 || c.nops > 7 || c.restart_check) return 0;
  for (uint32_t k = 0; k < c.nops; k++) {
    if (c.op_kind[k] != ((k & 1u) ? kChainByte : kChainRun)) return 0;
    if (c.op_cls[k] != (k & 1u)) return 0;
  }
  for (int q = 0; q < 2; q++) if (c.cls_kind[q] == kClsSet || c.cls_hi[q] > 0x7Fu || c.cls_lo[q] > c.cls_hi[q]) return 0;
  if (c.cls_lo[0] <= c.cls_hi[1] && c.cls_lo[1] <= c.cls_hi[0]) return 0;   // the classes meet
  return static_cast<int>((c.nops + 1) / 2);
}

// Is the chain a literal the persistent kernel's literal mode evaluates?  2..63 single-byte steps over 2..4 distinct ASCII bytes, and
// no border: no proper prefix of the literal is also its suffix (`abab`, `aa`: occurrences overlap and FindAll keeps every
// other one — a sequential rule; those stay on the chain kernel).  Returns the number of distinct bytes, else 0.
int literal_shape(const ChainAux& c) {
  if (c.nops < 2 || c.nops > 63 || c.ncls < 2 || c.ncls > 4 || c.restart_check) return 0;
  uint8_t lit[64];
  for (uint32_t k = 0; k < c.nops; k++) {
    const uint32_t q = c.op_cls[k];
    if (c.op_kind[k] != kChainByte || q >= c.ncls || c.cls_kind[q] == kClsSet || c.cls_kind[q] == kClsDigit || c.cls_lo[q] != c.cls_hi[q] || c.cls_lo[q] > 0x7Fu) return 0;
    lit[k] = c.cls_lo[q];
  }
  for (uint32_t q = 0; q < c.ncls; q++) for (uint32_t r = q + 1; r < c.ncls; r++) if (c.cls_lo[q] == c.cls_lo[r]) return 0;
  for (uint32_t b = 1; b < c.nops; b++) {                            // a border of length b
    bool same = true;
    for (uint32_t i = 0; i < b && same; i++) same = lit[i] == lit[c.nops - b + i];
    if (same) return 0;
  }
  return static_cast<int>(c.ncls);
}

__global__ void k_sum_counts(const uint64_t* counts, uint64_t n, uint64_t* total);

namespace {
// Resident workgroups per CU of a persistent instantiation.  The grid must be co-resident, so this errs on the low side: the
// runtime's occupancy query, and our own count from the kernel's attributes — LDS in 1 280-byte granules of the CU's 160 KiB,
// VGPRs in granules of 8 of the SIMD's 512.  (Round 5: the query answered 6 for the three-bitmap literal instantiation, 27 008
// bytes of LDS; five fit — the sixth workgroup of every CU never started and the launch ran into its watchdog.)
template <int K, int KD, int KP, int LIT = 0>
int pers_occupancy() {
  static int occ = -1;
  if (occ < 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_scan_fields_pers<K, KD, KP, LIT>, kThreads, 0) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_scan_fields_pers<K, KD, KP, LIT>)) == hipSuccess) {
      const int lds = static_cast<int>((fa.sharedSizeBytes + 1279) / 1280 * 1280);
      const int by_lds = lds > 0 ? (160 * 1024) / lds : 8;
      const int regs = (fa.numRegs + 7) / 8 * 8;
      const int by_regs = regs > 0 ? 512 / regs : 8;                // waves per SIMD = workgroups per CU (four waves, four SIMDs)
      if (by_lds < n) n = by_lds;
      if (by_regs < n) n = by_regs;
    } else (void)hipGetLastError();
    int want = CXG_PF_OCC;
    if (const char* e = getenv("CXG_PF_OCC")) want = atoi(e);
    occ = n < want ? n : want;
    if (occ < 0) occ = 0;
    if (getenv("CXG_VERBOSE")) fprintf(stderr, "[cxg] persistent kernel (LIT %d): %d workgroups per CU\n", LIT, occ);
  }
  return occ;
}
template <int K, int KD, int KP, int LIT = 0>
bool launch_pers_inst(ScanArgs a, hipStream_t stream) {
  const int occ = pers_occupancy<K, KD, KP, LIT>();
  static const bool verbose = getenv("CXG_VERBOSE") != nullptr;
  if (occ <= 0) { if (verbose) fprintf(stderr, "[cxg] persistent kernel (LIT %d): the occupancy query says %d workgroups per CU — not launched\n", LIT, occ); return false; }
  static int cus = 0;
  if (cus == 0) { int dev = 0; cus = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); }
  const uint64_t nwt = (a.len + kWaveTile - 1) / kWaveTile;
  uint64_t G = static_cast<uint64_t>(cus) * static_cast<uint64_t>(occ);
  if (G * kWavesPerBlock > static_cast<uint64_t>(kPfMaxWaves)) G = kPfMaxWaves / kWavesPerBlock;
  if (G > (nwt + 3) / 4) G = (nwt + 3) / 4;                           // short input: one tile per wave
  const uint64_t W = G * kWavesPerBlock;
#if CXG_PF_TICKETS
  // units in haystack order: the bulk in units of kPfTiles tiles, then W units of three tiles, then 2 W single tiles — the waves
  // end within a tile of each other whatever their speeds were (round 5's static rounds ended with their slowest wave: ~7 % of a
  // 1 GiB launch).  -DCXG_PF_TAIL3 / CXG_PF_TAIL1: those two counts in units of W, for A/B.
  const uint64_t t1 = std::min<uint64_t>(nwt, static_cast<uint64_t>(CXG_PF_TAIL1) * W);
  const uint64_t t3 = std::min<uint64_t>(nwt - t1, 3ull * CXG_PF_TAIL3 * W) / 3ull * 3ull;
  const uint64_t t9 = nwt - t1 - t3;
  const uint64_t u9 = (t9 + kPfTiles - 1) / kPfTiles, u3 = t3 / 3ull, u1 = t1, units = u9 + u3 + u1;
  const uint64_t rounds = (units + W - 1) / W;
  if (units + 64 > a.pf_cap || rounds + 1 > a.pf_rec_rounds || units > 0x7FFFFFFFull || a.pf_ticket == nullptr) {
    if (verbose) fprintf(stderr, "[cxg] persistent kernel: %llu units in %llu rounds of %llu waves do not fit the status arrays (%llu words, %llu rounds) — not launched\n",
                         (unsigned long long)units, (unsigned long long)rounds, (unsigned long long)W, (unsigned long long)a.pf_cap, (unsigned long long)a.pf_rec_rounds);
    return false;
  }
  a.pf_full = static_cast<uint32_t>(u9); a.pf_tpw_last = static_cast<uint32_t>(u3); a.pf_units_last = static_cast<uint32_t>(u1);
  { uint32_t nc = 1; while (nc * 2u <= G && nc < 64u) nc *= 2u; a.pf_ncounters = nc; }   // (every counter has workgroups asking it, the first-dispatched ones among them)
#else
  const uint64_t per_round = static_cast<uint64_t>(kPfTiles) * W;
  const uint64_t full = nwt / per_round, rem = nwt - full * per_round;
  const uint64_t tpw_last = (rem + W - 1) / W;
  const uint64_t units_last = tpw_last ? (rem + tpw_last - 1) / tpw_last : 0;
  if ((full + 1) * W > a.pf_cap || full + 1 > a.pf_rec_rounds || full > 0xFFFFull) {
    if (verbose) fprintf(stderr, "[cxg] persistent kernel: %llu rounds of %llu waves do not fit the status arrays (%llu words, %llu rounds) — not launched\n",
                         (unsigned long long)(full + 1), (unsigned long long)W, (unsigned long long)a.pf_cap, (unsigned long long)a.pf_rec_rounds);
    return false;
  }   // (the round number is part of a block sum's tag: 16 bits = 64 Ki rounds of 180 MiB)
  a.pf_full = static_cast<uint32_t>(full); a.pf_tpw_last = static_cast<uint32_t>(tpw_last); a.pf_units_last = static_cast<uint32_t>(units_last);
#endif
  hipLaunchKernelGGL((k_scan_fields_pers<K, KD, KP, LIT>), dim3(static_cast<unsigned>(G)), dim3(kThreads), 0, stream, a);
  if (a.count_sum) hipLaunchKernelGGL(k_sum_counts, dim3(1), dim3(1024), 0, stream, a.status, W, a.total);
  return true;
}
template <int K>
bool launch_pers_k(const ScanArgs& a, uint32_t kd, uint32_t kp, hipStream_t stream) {
  const bool dd = kd == kClsDigit, pb = kp == kClsByte;
  if (dd && pb) return launch_pers_inst<K, kClsDigit, kClsByte>(a, stream);
  if (dd) return launch_pers_inst<K, kClsDigit, kClsRange>(a, stream);
  if (pb) return launch_pers_inst<K, kClsRange, kClsByte>(a, stream);
  return launch_pers_inst<K, kClsRange, kClsRange>(a, stream);
}
template <int K>
void launch_fields_k(const ScanArgs& a, uint32_t kd, uint32_t kp, dim3 grid, dim3 block, hipStream_t stream) {
  const bool dd = kd == kClsDigit, pb = kp == kClsByte;
  if (dd && pb) hipLaunchKernelGGL((k_scan_fields_wave<K, kClsDigit, kClsByte>), grid, block, 0, stream, a);
  else if (dd) hipLaunchKernelGGL((k_scan_fields_wave<K, kClsDigit, kClsRange>), grid, block, 0, stream, a);
  else if (pb) hipLaunchKernelGGL((k_scan_fields_wave<K, kClsRange, kClsByte>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((k_scan_fields_wave<K, kClsRange, kClsRange>), grid, block, 0, stream, a);
}
}  // namespace

__global__ __launch_bounds__(1024) void k_sum_counts(const uint64_t* counts, uint64_t n, uint64_t* total) {
  __shared__ uint64_t s_part[16];
  uint64_t v = 0;
  for (uint64_t i = threadIdx.x; i < n; i += 1024) v += counts[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { uint64_t t = 0; for (int i = 0; i < 16; i++) t += s_part[i]; *total = t; }
}

// a.ngroups = number of 120 KiB groups (one workgroup each).
hipError_t launch_scan_fields_wave(const ScanArgs& a, hipStream_t stream, bool* persistent) {
  if (persistent) *persistent = false;
  const ChainAux& c = *reinterpret_cast<const ChainAux*>(a.chain);
  if (const int nc = literal_shape(c)) {                              // literal mode: the persistent kernel or nothing (the caller keeps the chain kernel)
    if (a.pf_status == nullptr) return hipErrorInvalidValue;
    const bool done = nc == 2 ? launch_pers_inst<2, kClsByte, kClsByte, 2>(a, stream) : nc == 3 ? launch_pers_inst<2, kClsByte, kClsByte, 3>(a, stream)
                                                                                                : launch_pers_inst<2, kClsByte, kClsByte, 4>(a, stream);
    if (!done) return hipErrorInvalidValue;
    if (persistent) *persistent = true;
    return hipGetLastError();
  }
  const int k = fields_shape(c);
  const dim3 grid(static_cast<unsigned>(a.ngroups)), block(kThreads);
  if (a.pf_status != nullptr) {                                      // persistent grid, deferred ordering (k_scan_fields_pers)
    bool done = false;
    switch (k) {
      case 2: done = launch_pers_k<2>(a, c.cls_kind[0], c.cls_kind[1], stream); break;
      case 3: done = launch_pers_k<3>(a, c.cls_kind[0], c.cls_kind[1], stream); break;
      case 4: done = launch_pers_k<4>(a, c.cls_kind[0], c.cls_kind[1], stream); break;
      default: return hipErrorInvalidValue;
    }
    if (done) { if (persistent) *persistent = true; return hipGetLastError(); }
  }
  switch (k) {
    case 2: launch_fields_k<2>(a, c.cls_kind[0], c.cls_kind[1], grid, block, stream); break;
    case 3: launch_fields_k<3>(a, c.cls_kind[0], c.cls_kind[1], grid, block, stream); break;
    case 4: launch_fields_k<4>(a, c.cls_kind[0], c.cls_kind[1], grid, block, stream); break;
    default: return hipErrorInvalidValue;
  }
  if (a.count_sum) hipLaunchKernelGGL(k_sum_counts, dim3(1), dim3(1024), 0, stream, a.status, a.ngroups, a.total);
  return hipGetLastError();
}

// =====================================================================================================================
// k_scan_trio_wave — FindAll / FindAllSubmatch for the chain  run(F) byte(a) run(F) byte(b) run(F)  with a != b single bytes
// outside F: `(\w+)@(\w+)\.(\w+)` (BASELINE configs[4]), `\d+-\d+:\d+`.  Same window, ownership and group structure as the
// fields kernel above; what differs:
//   A  all three class bitmaps from ONE byte table in LDS (T[b] = F | a << 1 | b << 2, built from the chain's class list):
//      F may be a union of ranges (`\w`), where a compare per range and dword costs more than a lookup per byte.
//   B  links LA = A & (D<<1) & (D>>1), LB likewise; super-run = maximal stretch of F bytes and links of either kind; the tile
//      owns the super-runs that start in its bytes: OWN = X & ~(X + WS), X = D | LA | LB (one multiword addition).
//   C  a match is a run, an LA link, a run, an LB link, a run; it starts at the START of its first run (leftmost-first: `\w+`
//      takes the run from its first byte) and every owned LA link is a candidate: hop over the run behind it — it must end
//      on an LB link — and over the run behind that: two additions give the ends of all candidates.
//   D  FindAll resumes at a match's end, so a candidate whose first run is the third run of an earlier selected match is
//      dropped (`a@b.c@d.e`: c belongs to the first match).  That is the case exactly when an end sits on the LA link of a
//      candidate; rare on text — resolved by a loop (heads of such chains are selected, what they block is removed, repeat).
//   E  rows: per end bit the nearest bytes outside F below it are the LB link, then the LA link, then the byte in front of the
//      match — three bit scans over (previous lane's word : this word); a match that reaches further back hands the scan over.
//      A row is four positions (start, LA link, LB link, end); the epilogue turns them into the program's capture slots
//      (walk.hpp ChainCaps: start / end / end of the first / second run, plus a constant) or into [start, end).
// Fallback flag: as for the fields kernel (the host reruns on scan_chain_wave.hip).
template <int K, bool EQ>
__global__ __launch_bounds__(kThreads, (K == 4 ? 6 : CXG_TRIO_WAVES)) void k_scan_trio_wave(ScanArgs a) {
  typedef typename std::conditional<K == 4, uint32_t, uint16_t>::type LinkT;   // K - 1 byte distances per row
  __shared__ __attribute__((aligned(16))) uint64_t s_d[kWavesPerBlock][64];
  __shared__ __attribute__((aligned(16))) uint64_t s_c[K - 1][kWavesPerBlock][64];   // separator classes
  __shared__ uint32_t s_row[kWavesPerBlock][kTRows];
  __shared__ LinkT s_lnk[kWavesPerBlock][kTRows];
  __shared__ uint8_t s_cls[256];
  __shared__ uint32_t s_cnt[kWavesPerBlock][kTilesPerWave];
  __shared__ uint32_t s_qbase[kWavesPerBlock * kTilesPerWave + 1];
  __shared__ uint64_t s_group;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = lane0;
  uint64_t group = blockIdx.x;
  if (!a.static_groups) {
    if (tid == 0) s_group = claim_group(false, a.ticket, a.ngroups);
    __syncthreads();
    group = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_group >> 32))) << 32) |
            static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_group)));
  }
  if (group >= a.ngroups) return;
  if (limit_reached_skip(a, group, &s_base)) return;
  {
    const ChainAux* gch = reinterpret_cast<const ChainAux*>(a.chain);
    const uint32_t b = static_cast<uint32_t>(tid);
    uint32_t f = chain_class_has(*gch, 0, b) ? 1u : 0u;             // bit 0: F; bit i + 1: the separator of link i
    for (int i = 0; i < K - 1; i++) f |= chain_class_has(*gch, gch->op_cls[2 * i + 1], b) ? (2u << i) : 0u;
    s_cls[tid] = static_cast<uint8_t>(f);
  }
  SetRanges frg;                                                     // class 0 as ranges (trio_shape: ASCII), the separators as splat bytes
  uint32_t sep4[K - 1];
  bool swar_ok = CXG_TRIO_SWAR != 0;                                  // (uniform) every bound below 0x80: the SWAR tests see bytes >= 0x80 as outside
  {
    const ChainAux* gch = reinterpret_cast<const ChainAux*>(a.chain);
    const uint32_t kind0 = gch->cls_kind[0];
    frg.n = kind0 == kClsSet ? gch->cls_nr[0] : 1u;
    for (uint32_t q = 0; q < frg.n && q < 4u; q++) swar_ok = swar_ok && (kind0 == kClsSet ? gch->cls_rhi[0][q] : kind0 == kClsDigit ? 0x39u : gch->cls_hi[0]) < 0x80u;
    for (int i = 0; i < K - 1; i++) swar_ok = swar_ok && gch->cls_lo[gch->op_cls[2 * i + 1]] < 0x80u;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t lo = kind0 == kClsSet ? gch->cls_rlo[0][q] : kind0 == kClsDigit ? 0x30u : gch->cls_lo[0];
      const uint32_t hi = kind0 == kClsSet ? gch->cls_rhi[0][q] : kind0 == kClsDigit ? 0x39u : gch->cls_hi[0];
      frg.lo4[q] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(lo * 0x01010101u)));
      frg.hi4[q] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((0x7Fu - hi) * 0x01010101u)));
    }
#pragma unroll
    for (int i = 0; i < K - 1; i++) sep4[i] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gch->cls_lo[gch->op_cls[2 * i + 1]] * 0x01010101u)));
  }
  __syncthreads();
  constexpr int tpw = kTilesPerWave;
  uint32_t nrows_w = 0, fallback = 0;
  auto tile_lo_of = [&](int jj) { return (group * (kWavesPerBlock * tpw) + static_cast<uint64_t>(jj) * kWavesPerBlock + wave) * static_cast<uint64_t>(kWaveTile); };
  u32x4 x[4];
  int32_t nvalid_cur = 0;
  fields_first_loads(x, fields_window(a.hay, a.len, tile_lo_of(0), true, nvalid_cur), lane, group == 0 && wave == 0);
  const bool want_rows = a.out != nullptr || a.max_len != 0;

  for (int j = 0; j < tpw; j++) {
    lane = lane0;
    asm volatile("" : "+v"(lane));
    int32_t nvalid_next = 0;
    const __amdgpu_buffer_rsrc_t rnext = fields_window(a.hay, a.len, tile_lo_of(j + 1), j + 1 < tpw, nvalid_next);
    // ---- A: class flags, 16 per class and vector through the LDS scratch.  CXG_TRIO_SWAR (round 5): F as a union of <= 4 ASCII ranges
    // by SWAR compares and the separators as byte compares, as the fields kernel does — no LDS lookups (the byte table cost 16
    // ds_read_u8 per vector and lane: the LDS pipe of a CU, shared by its four SIMDs, was as busy as the VALUs).  0: the table.
    if (swar_ok) {
      uint16_t* pd = reinterpret_cast<uint16_t*>(s_d[wave]);
      const uint32_t voff = static_cast<uint32_t>(lane) << 4;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32x4 t = x[k] & 0x7F7F7F7Fu;
        with_shape(static_cast<int>(a.plan_shape), [&]<int SHAPE>() { pd[lane + 64 * k] = static_cast<uint16_t>(notshape16<SHAPE>(x[k], a.plan, frg) ^ 0xFFFFu); });
#pragma unroll
        for (int i = 0; i < K - 1; i++) reinterpret_cast<uint16_t*>(s_c[i][wave])[lane + 64 * k] = static_cast<uint16_t>(piece16<kClsByte>(x[k], t, sep4[i], 0u));
        x[k] = __builtin_amdgcn_raw_buffer_load_b128(rnext, voff + 1024u * k, 0, CXG_HAY_LOAD_AUX);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else
    {
      uint16_t* pd = reinterpret_cast<uint16_t*>(s_d[wave]);
      const uint32_t voff = static_cast<uint32_t>(lane) << 4;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t w[4] = {x[k].x, x[k].y, x[k].z, x[k].w};
        uint32_t fd = 0, fc[K - 1];
#pragma unroll
        for (int i = 0; i < K - 1; i++) fc[i] = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t f = static_cast<uint32_t>(s_cls[w[q] & 0xFFu]) | (static_cast<uint32_t>(s_cls[(w[q] >> 8) & 0xFFu]) << 8) |
                             (static_cast<uint32_t>(s_cls[(w[q] >> 16) & 0xFFu]) << 16) | (static_cast<uint32_t>(s_cls[w[q] >> 24]) << 24);
          const uint32_t wt = (q & 1) ? 0x80402010u : 0x08040201u;
          // (flag bytes are 0/1 after the mask: the weighted sum is the four flags as bits q*4 .. q*4+3)
          if (q < 2) {
            fd = __builtin_amdgcn_udot4(f & 0x01010101u, wt, fd, false);
#pragma unroll
            for (int i = 0; i < K - 1; i++) fc[i] = __builtin_amdgcn_udot4((f >> (i + 1)) & 0x01010101u, wt, fc[i], false);
          } else {
            fd += __builtin_amdgcn_udot4(f & 0x01010101u, wt, 0u, false) << 8;
#pragma unroll
            for (int i = 0; i < K - 1; i++) fc[i] += __builtin_amdgcn_udot4((f >> (i + 1)) & 0x01010101u, wt, 0u, false) << 8;
          }
        }
        pd[lane + 64 * k] = static_cast<uint16_t>(fd);
#pragma unroll
        for (int i = 0; i < K - 1; i++) reinterpret_cast<uint16_t*>(s_c[i][wave])[lane + 64 * k] = static_cast<uint16_t>(fc[i]);
        x[k] = __builtin_amdgcn_raw_buffer_load_b128(rnext, voff + 1024u * k, 0, CXG_HAY_LOAD_AUX);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    wave_lds_sync();
    int lw = lane;
    asm volatile("" : "+v"(lw));
    uint64_t Dw = s_d[wave][lw], Cw[K - 1];
#pragma unroll
    for (int i = 0; i < K - 1; i++) Cw[i] = s_c[i][wave][lw];
    if (nvalid_cur != kFWin) {
      const int32_t nf = nvalid_cur - 64 * lane;
      const uint64_t vf = nf <= 0 ? 0ull : (nf >= 64 ? ~0ull : ((1ull << nf) - 1ull));
      Dw &= vf;
#pragma unroll
      for (int i = 0; i < K - 1; i++) Cw[i] &= vf;
    }
    nvalid_cur = nvalid_next;
    const uint32_t d0 = static_cast<uint32_t>(Dw), d1 = static_cast<uint32_t>(Dw >> 32);
    uint32_t c0[K - 1], c1[K - 1];
#pragma unroll
    for (int i = 0; i < K - 1; i++) { c0[i] = static_cast<uint32_t>(Cw[i]); c1[i] = static_cast<uint32_t>(Cw[i] >> 32); }
    const TrioTile t = trio_core<K, EQ>(d0, d1, c0, c1);
    if (t.ovf) fallback |= 1u;
    const uint32_t c = static_cast<uint32_t>(__popc(t.e0)) + static_cast<uint32_t>(__popc(t.e1));
    const uint32_t incl = wave_inclusive_sum_fused(c);
    const uint32_t tot = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
    if (tot != 0 && want_rows) trio_rows<K, LinkT>(t, d0, d1, lane, s_row[wave], s_lnk[wave], nrows_w + incl - c, static_cast<uint32_t>(kTRows));
    if (lane == 0) s_cnt[wave][j] = tot;
    nrows_w += tot;
    wave_lds_sync();                                                // (the bitmap scratch is rewritten by the next tile)
  }
  if (nrows_w > static_cast<uint32_t>(kTRows)) fallback |= 16u;
  wave_lds_sync();
  {
    bool bad = false, long_hit = false;
    if (want_rows) {
      for (uint32_t r = lane0; r < nrows_w && r < static_cast<uint32_t>(kTRows); r += 64) {
        const uint32_t v = s_row[wave][r];
        const uint32_t st = v & 0xFFFFu, en = v >> 16;
        bad = bad || st >= en;
        long_hit = long_hit || (a.max_len != 0 && en - st > a.max_len);
      }
    }
    if (__ballot(bad) != 0ull) fallback |= 2u;
    if (__ballot(long_hit) != 0ull && lane0 == 0) raise_err(a.err, kErrLongMatch);
  }
  if (fallback != 0 && lane0 == 0) raise_err(a.err, 8u | (fallback << 8));
  __syncthreads();
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
  const ChainCaps* cp = reinterpret_cast<const ChainCaps*>(a.caps);
  const bool caps = cp->on == 1u;
  const uint32_t npairs = a.row_width >> 1;                          // 16-byte pairs per row
  // a row is written by the next power of two >= npairs lanes (<= 64, the host checks), so that a lane always writes the same
  // pair of a row and what its two slots are made of is decided once, outside the loops
  const uint32_t lsh = npairs <= 1u ? 0u : 32u - static_cast<uint32_t>(__builtin_clz(npairs - 1u));
  const uint32_t pr = static_cast<uint32_t>(lane0) & ((1u << lsh) - 1u);
  const bool lane_on = pr < npairs;
  auto slot_of = [&](uint32_t k, uint32_t& sel, int32_t& off) {      // sel: 0 start, 1 end, 2 + i link i (the end of run i), 7 unset
    const uint32_t src = cp->src[k];
    sel = src == kCapSrcStart ? 0u : src == kCapSrcEnd ? 1u : 7u;
    if (src >= kCapSrcRun0 && src < kCapSrcRun0 + kCapMaxRuns) { const uint32_t op = cp->run_op[src - kCapSrcRun0]; sel = (op >> 1) < static_cast<uint32_t>(K - 1) ? 2u + (op >> 1) : 1u; }
    off = cp->off[k];
  };
  uint32_t sel0 = 0, sel1 = 1; int32_t off0 = 0, off1 = 0;
  if (caps && lane_on) { slot_of(2u * pr, sel0, off0); slot_of(2u * pr + 1u, sel1, off1); }
  const int64_t origin = a.base + static_cast<int64_t>(group * static_cast<uint64_t>(kWaveTile) * kWavesPerBlock * tpw) - kFPre;
  // What a lane's two slots are made of, as shifts and masks decided once (the kernels are bound by instruction issue, DESIGN.md
  // section 5: a chain of selects on 64-bit values per slot and row cost as much as the tile mathematics of the rows):
  //   slot = tb + off + ((w0 >> shA) & 0xFFFF) + ((w1 >> shB) & mB)      start: shA 0, mB 0; end: shA 16, mB 0; link i: shA 0, shB 8 i, mB 0xFF
  const uint32_t shA0 = sel0 == 1u ? 16u : 0u, shA1 = sel1 == 1u ? 16u : 0u;
  const bool lk0 = sel0 >= 2u && sel0 != 7u, lk1 = sel1 >= 2u && sel1 != 7u;
  const uint32_t shB0 = lk0 ? 8u * (sel0 - 2u) : 0u, shB1 = lk1 ? 8u * (sel1 - 2u) : 0u, mB0 = lk0 ? 0xFFu : 0u, mB1 = lk1 ? 0xFFu : 0u;
  const bool un0 = sel0 == 7u, un1 = sel1 == 7u;
  const uint32_t per = 64u >> lsh;                                   // rows per iteration of a wave
  uint32_t start = 0;
  for (int j = 0; j < tpw; j++) {
    const uint32_t n = s_cnt[wave][j];
    const uint64_t dst = base + s_qbase[j * kWavesPerBlock + wave];
    const int64_t tb = origin + static_cast<int64_t>(j * kWavesPerBlock + wave) * kWaveTile;
    // rows of this tile that are parked and fit the output array (uniform)
    uint32_t fit = start >= static_cast<uint32_t>(kTRows) ? 0u : (n < static_cast<uint32_t>(kTRows) - start ? n : static_cast<uint32_t>(kTRows) - start);
    fit = dst >= a.cap ? 0u : (a.cap - dst < fit ? static_cast<uint32_t>(a.cap - dst) : fit);
    const int64_t tb0 = tb + off0, tb1 = tb + off1;
    uint32_t rr = static_cast<uint32_t>(lane0) >> lsh;
    int64_t* po = a.out + (dst + rr) * a.row_width + 2u * pr;
    if (lane_on)
      for (; rr < fit; rr += per, po += static_cast<uint64_t>(per) * a.row_width) {   // consecutive lanes write consecutive 16 bytes
        const uint32_t w0 = s_row[wave][start + rr], w1 = s_lnk[wave][start + rr];
        const int64_t v0 = tb0 + static_cast<int64_t>(((w0 >> shA0) & 0xFFFFu) + ((w1 >> shB0) & mB0));
        const int64_t v1 = tb1 + static_cast<int64_t>(((w0 >> shA1) & 0xFFFFu) + ((w1 >> shB1) & mB1));
        store_pair_nt(po, un0 ? -1 : v0, un1 ? -1 : v1);
      }
    start += n;
  }
}

// Does the chain have the shape k_scan_trio_wave evaluates?  run(0) (byte(c_i) run(0)){K-1}, K = 2..4, every c_i a single byte
// outside class 0; for K >= 3 the separators either pairwise different (two candidates then share at most one run, which the
// overlap resolution looks for) or all the same (the EQ instantiation: `(\d+)\.(\d+)\.(\d+)\.(\d+)`; without captures that
// shape is the fields kernel's).  Returns K, | 8 for one separator, else 0.
int trio_shape(const ChainAux& c) {
  if ((c.nops & 1u) == 0 || c.nops < 3 || c.nops > 7 || c.restart_check) return 0;
  const int K = static_cast<int>((c.nops + 1) / 2);
  uint8_t sep[3] = {0, 0, 0};
  for (uint32_t k = 0; k < c.nops; k++) {
    if (c.op_kind[k] != ((k & 1u) ? kChainByte : kChainRun)) return 0;
    if (!(k & 1u)) { if (c.op_cls[k] != 0) return 0; continue; }
    const uint32_t q = c.op_cls[k];
    if (q == 0 || q >= c.ncls || c.cls_kind[q] == kClsSet || c.cls_kind[q] == kClsDigit || c.cls_lo[q] != c.cls_hi[q]) return 0;
    if (chain_class_has(c, 0, c.cls_lo[q])) return 0;
    sep[k >> 1] = c.cls_lo[q];
  }
  if (K >= 3) {
    int same = 0, pairs = 0;
    for (int i = 0; i < K - 1; i++) for (int j = i + 1; j < K - 1; j++) { pairs++; if (sep[i] == sep[j]) same++; }
    if (same == pairs) return K | 8;
    if (same) return 0;
  }
  return K;
}

hipError_t launch_scan_trio_wave(const ScanArgs& a, hipStream_t stream, bool* persistent) {
  if (persistent) *persistent = false;
  const dim3 grid(static_cast<unsigned>(a.ngroups)), block(kThreads);
  if (a.pf_status != nullptr) {                                      // the persistent kernel's TRIO mode (round 5)
    bool done = false;
    switch (trio_shape(*reinterpret_cast<const ChainAux*>(a.chain))) {
      case 2: done = launch_pers_inst<2, kClsByte, kClsByte, 16 + 2>(a, stream); break;
      case 3: done = launch_pers_inst<2, kClsByte, kClsByte, 16 + 3>(a, stream); break;
      case 4: done = launch_pers_inst<2, kClsByte, kClsByte, 16 + 4>(a, stream); break;
      case 3 | 8: done = launch_pers_inst<2, kClsByte, kClsByte, 16 + 3 + 8>(a, stream); break;
      case 4 | 8: done = launch_pers_inst<2, kClsByte, kClsByte, 16 + 4 + 8>(a, stream); break;
      default: return hipErrorInvalidValue;
    }
    if (done) { if (persistent) *persistent = true; return hipGetLastError(); }
  }
  switch (trio_shape(*reinterpret_cast<const ChainAux*>(a.chain))) {
    case 2: hipLaunchKernelGGL((k_scan_trio_wave<2, false>), grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL((k_scan_trio_wave<3, false>), grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL((k_scan_trio_wave<4, false>), grid, block, 0, stream, a); break;
    case 3 | 8: hipLaunchKernelGGL((k_scan_trio_wave<3, true>), grid, block, 0, stream, a); break;
    case 4 | 8: hipLaunchKernelGGL((k_scan_trio_wave<4, true>), grid, block, 0, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace cxgdev
