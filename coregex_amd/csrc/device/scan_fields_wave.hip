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
#include <stdlib.h>
#include <stdint.h>

#include "block_common.hpp"
#include "scan_dfa.h"
#include "walk.hpp"
#include "wave_common.hpp"
#include "stream_common.hpp"

#ifndef CXG_FIELDS_WAVES
#define CXG_FIELDS_WAVES 8
#endif
// -DCXG_FABL=n (experiments only, results WRONG): 1 = no rows, 2 = no chain and no rows, 3 = no class masks either,
// 4 = no LDS transpose either (the window is only read), 5 = 4 without any output ordering (grouped: no barrier / look-back;
// stream: no count words, no scanner)
#ifndef CXG_FABL
#define CXG_FABL 0
#endif
// Streaming kernel: consecutive wave-tiles a wave takes per round (the window of the resident waves stays dense: round r,
// wave W reads tiles (r * NW + W) * CXG_UNIT + 0 .. CXG_UNIT - 1)
#ifndef CXG_UNIT
#define CXG_UNIT 4
#endif

namespace cxgdev {

namespace {

constexpr int kFPre = 64;                                  // window bytes in front of the tile
constexpr int kFWin = kWaveTile + kWaveHalo;               // 4096
constexpr int kFRows = 64 * kTilesPerWave;                                // rows buffered per wave and group
constexpr unsigned long long kFOwn = 0x1FFFFFFFFFFFFFFEull;   // lanes 1..60 own their words

// 0x80 in every byte of x that IS in the class.
template <int KIND>
__device__ __forceinline__ uint32_t incls4(uint32_t x, uint32_t lo4, uint32_t hi4) {   // lo4 / hi4: bounds splat over the bytes (hi4 = 0x7F - hi)
  if (KIND == kClsDigit) return ~((((x ^ 0x30303030u) & 0x7F7F7F7Fu) + 0x76767676u) | x) & 0x80808080u;
  if (KIND == kClsByte) return ~((((x ^ lo4) & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
  const uint32_t ge = (x | 0x80808080u) - lo4;
  const uint32_t gt = (x & 0x7F7F7F7Fu) + hi4;
  return ge & ~gt & ~x & 0x80808080u;
}
// 16 class bits of a 16-byte vector: the four flags of a dword are gathered by one v_dot4_u32_u8 (weights 1,2,4,8 resp.
// 16,32,64,128: 128 x the byte of flags accumulates over a dword pair).  Bits above 15 are garbage (ds_write_b16 drops them).
template <int KIND>
__device__ __forceinline__ uint32_t piece16(const u32x4& x, uint32_t lo4, uint32_t hi4) {
  const uint32_t lo = __builtin_amdgcn_udot4(incls4<KIND>(x.y, lo4, hi4), 0x80402010u, __builtin_amdgcn_udot4(incls4<KIND>(x.x, lo4, hi4), 0x08040201u, 0u, false), false);
  const uint32_t hi = __builtin_amdgcn_udot4(incls4<KIND>(x.w, lo4, hi4), 0x80402010u, __builtin_amdgcn_udot4(incls4<KIND>(x.z, lo4, hi4), 0x08040201u, 0u, false), false);
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
template <int K>
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
  uint32_t b0 = sel_lanes(d0 & ~Dl0 & ~Ll0, kFOwn), b1 = sel_lanes(d1 & ~Dl1 & ~Ll1, kFOwn);   // B: group starts (first: the owned super-run starts)
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
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fields_window(const uint8_t* hay, uint64_t len, uint64_t lo, bool live, int32_t& nvalid) {
  int nrec = 0;
  uint64_t wlo = 0;
  nvalid = 0;
  if (live && lo < len) {
    wlo = lo >= static_cast<uint64_t>(kFPre) ? lo - kFPre : 0;
    const uint64_t rem = len - wlo;
    const uint64_t full = static_cast<uint64_t>(kFWin) - (lo - wlo == 0 ? kFPre : 0);
    nrec = rem >= full ? static_cast<int>(full) : static_cast<int>((rem + 3) & ~3ull);
    const uint64_t nv = len - lo + kFPre;
    nvalid = nv >= static_cast<uint64_t>(kFWin) ? kFWin : static_cast<int32_t>(nv);
  }
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(hay) + wlo, 0, nrec, 0x00020000);
}
// Phase A for one window held in x[]: class pieces into the wave's LDS scratch; every vector's register is refilled from
// `rnext` right behind its last use.  Returns the lane's words (d1:d0), (p1:p0), masked to the valid bytes of a short window.
template <int KD, int KP>
__device__ __forceinline__ void fields_words(u32x4 (&x)[4], __amdgpu_buffer_rsrc_t rnext, int lane, uint64_t* sd, uint64_t* sp, int32_t nvalid,
                                             uint32_t dlo4, uint32_t dhi4, uint32_t plo4, uint32_t phi4, uint32_t& sink,
                                             uint32_t& d0, uint32_t& d1, uint32_t& p0, uint32_t& p1) {
  {
    uint16_t* pd = reinterpret_cast<uint16_t*>(sd);
    uint16_t* pp = reinterpret_cast<uint16_t*>(sp);
    const uint32_t voff = static_cast<uint32_t>(lane) << 4;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (CXG_FABL >= 4) {
        sink ^= x[k].x ^ x[k].y ^ x[k].z ^ x[k].w;
      } else if (CXG_FABL == 3) {
        pd[lane + 64 * k] = static_cast<uint16_t>(x[k].x ^ x[k].z ^ x[k].y ^ x[k].w);
        pp[lane + 64 * k] = 0;
      } else {
        pd[lane + 64 * k] = static_cast<uint16_t>(piece16<KD>(x[k], dlo4, dhi4));
        pp[lane + 64 * k] = static_cast<uint16_t>(piece16<KP>(x[k], plo4, phi4));
      }
      x[k] = __builtin_amdgcn_raw_buffer_load_b128(rnext, voff + 1024u * k, 0, 0);
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
}
// The first loads of a wave (window of the tile at `lo`); `first`: the haystack's first tile, window bytes 0..63 do not exist.
__device__ __forceinline__ void fields_first_loads(u32x4 (&x)[4], __amdgpu_buffer_rsrc_t r0, int lane, bool first) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t off = static_cast<uint32_t>(lane + 64 * k) << 4;
    if (first) off = off >= static_cast<uint32_t>(kFPre) ? off - kFPre : 0x7FFFFFF0u;
    x[k] = __builtin_amdgcn_raw_buffer_load_b128(r0, off, 0, 0);
    __builtin_amdgcn_sched_barrier(0);                              // issue order = use order: the tile loop waits for x[0] with vmcnt(3), not for all four
  }
}

}  // namespace

// K: number of fields (2..4).  KD / KP: kind of the field / separator class (walk.hpp ChainClassKind; kClsRange also
// serves single bytes and digits as separators).
//
// GROUPED variant: a workgroup takes 32 consecutive wave-tiles (120 KiB), orders its rows after one barrier and looks back
// (block_common.hpp), as the other wave kernels do.  Kept as the A/B partner of the streaming kernel below
// (CXG_FIELDS_GROUPED=1) and as its fallback when the persistent grid cannot be resident.
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
  auto tile_lo_of = [&](int jj) { return (group * (kWavesPerBlock * tpw) + static_cast<uint64_t>(jj) * kWavesPerBlock + wave) * static_cast<uint64_t>(kWaveTile); };
  const uint64_t pt0 = a.prof ? __builtin_readcyclecounter() : 0ull;   // CXG_PROF=1: cycles of wave 0 per phase, summed over the workgroups

  u32x4 x[4];
  uint32_t sink = 0;                                                   // ablations only
  int32_t nvalid_cur = 0;
  fields_first_loads(x, fields_window(a.hay, a.len, tile_lo_of(0), true, nvalid_cur), lane, group == 0 && wave == 0);

  for (int j = 0; j < tpw; j++) {
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
  if (a.dbg & 2u) { if (tid == 0) s_base = 0; __syncthreads(); }     // CXG_DEBUG=2 (timing experiments, rows land in the wrong places): no look-back
  else if (a.dbg & 4u) tile_lookback(a.status, a.total, a.err, group, a.ngroups, total, &s_base, a.epoch);   // CXG_DEBUG=4: the flat look-back (A/B)
  else tile_lookback2(a.status, a.status3, a.total, a.err, group, a.ngroups, total, &s_base, a.epoch);
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
        *reinterpret_cast<longlong2*>(a.out + (dst + i) * a.row_width) = o;
      }
    }
    start += n;
  }
}

// PERSISTENT variant (the default): the grouped kernel with its wait taken out of the critical path.  A resident workgroup
// claims UNITS of 4 waves x kPTiles wave-tiles through a ticket (units are handed out in order, so every unit in front of a
// claimed one is being scanned or done: the look-back cannot deadlock, whatever the grid).  Per unit: scan the tiles (rows
// into one half of the wave's LDS buffer), one barrier, publish the unit's row count — and, instead of waiting for the rows
// in front of it, claim and scan the NEXT unit; the base of the earlier unit is resolved behind that, when the units in front
// of it have long been published, and its rows leave from the other half of the buffer.  Only a workgroup's last unit waits
// the way every group of the grouped kernel does.
// Measured (profiles/r03_fields_ablation.txt): grouped kernel 0.264 ms per GiB, the same without any look-back 0.199 — a
// group's slot idles until every group dispatched before it has finished.
// a.ngroups = units, a.status = their look-back words, a.status3 = 256 ticket counters (zeroed by the host).
#ifndef CXG_PTILES
#define CXG_PTILES 4
#endif
constexpr int kPTiles = CXG_PTILES;                  // wave-tiles per wave and unit: 4 x 4 x 3 840 B = 60 KiB per unit
constexpr int kPRows = 64 * kPTiles;                 // rows a wave buffers per unit (as the grouped kernel: 64 per tile)
constexpr uint32_t kPCounters = 256;                 // ticket counters (a.status3: 1 KiB, zeroed by the host)
template <int K, int KD, int KP>
__global__ __launch_bounds__(kThreads, CXG_FIELDS_WAVES) void k_scan_fields_pers(ScanArgs a) {
  __shared__ __attribute__((aligned(16))) uint64_t s_d[kWavesPerBlock][64];
  __shared__ __attribute__((aligned(16))) uint64_t s_p[kWavesPerBlock][64];
  __shared__ uint32_t s_row[2][kWavesPerBlock][kPRows];               // start | end << 16, window bit indices; [unit parity]
  __shared__ uint32_t s_cnt[2][kWavesPerBlock][kPTiles];
  __shared__ uint32_t s_qbase[2][kWavesPerBlock * kPTiles + 1];
  __shared__ uint64_t s_next;                                         // the unit claimed for the next round of the loop
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = lane0;
  const uint64_t nunits = a.ngroups;
  const ChainAux* gch = reinterpret_cast<const ChainAux*>(a.chain);
  const uint32_t dlo4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gch->cls_lo[0] * 0x01010101u)));
  const uint32_t dhi4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((0x7Fu - gch->cls_hi[0]) * 0x01010101u)));
  const uint32_t plo4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gch->cls_lo[1] * 0x01010101u)));
  const uint32_t phi4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((0x7Fu - gch->cls_hi[1]) * 0x01010101u)));
  const bool want_rows = a.out != nullptr || a.max_len != 0;
  // Tickets: kPCounters counters, counter c hands out units c, c + kPCounters, ...; a workgroup draws from counter
  // blockIdx.x mod kPCounters and moves on to the next ones when that is exhausted.  (Returning atomics on ONE address
  // complete at ~14 per microsecond — 73 ns each, scripts/microbench/stream.hip: 65 536 tickets over 8 counters cost 0.6 ms —
  // so 17 000 units per GiB need many counters: with the 8 per-XCD ones this kernel took 0.44 ms per GiB.)  Units are still
  // claimed in roughly ascending order, and the smallest unclaimed unit can always be claimed: the workgroups that draw from
  // its counter only ever wait for units smaller than theirs, all of which are claimed, hence published without a wait.
  // The atomic for the NEXT unit is issued in front of a unit's tiles and looked at behind them: its round trip never shows.
  uint32_t* const tickets = reinterpret_cast<uint32_t*>(a.status3);
  // (blockIdx / 8: consecutive workgroups go to different XCDs, so the eight workgroups that share a counter sit on eight
  // XCDs — an XCD that runs slower for a while draws fewer units from every counter instead of holding its own subsequence
  // of units back, which would stall every look-back behind them)
  const uint32_t myc = (blockIdx.x >> 3) % kPCounters;
  auto ticket_to_unit = [&](uint32_t tk) -> uint64_t {
    const uint64_t per = (nunits + kPCounters - 1 - myc) / kPCounters;
    if (tk < per) return static_cast<uint64_t>(tk) * kPCounters + myc;
    for (uint32_t k = 1; k < kPCounters; k++) {                       // steal (end of the input)
      const uint32_t x = (myc + k) % kPCounters;
      const uint64_t p2 = (nunits + kPCounters - 1 - x) / kPCounters;
      if (__hip_atomic_load(tickets + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p2) continue;   // exhausted: no atomic needed
      const uint32_t t2 = atomicAdd(tickets + x, 1u);
      if (t2 < p2) return static_cast<uint64_t>(t2) * kPCounters + x;
    }
    return nunits;
  };
  if (tid == 0) s_next = ticket_to_unit(atomicAdd(tickets + myc, 1u));
  __syncthreads();
  uint64_t unit = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_next >> 32))) << 32) |
                  static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_next)));
  uint32_t fallback = 0;
  int par = 0;                                                        // half of the buffers the unit being scanned uses
  bool have_prev = false;
  uint64_t prev_unit = 0;
  uint32_t prev_total = 0;
  uint32_t sink = 0;

  // rows of a scanned unit (buffers `pp`) to their places behind `base`
  auto write_rows = [&](int pp, uint64_t u, uint64_t base) {
    const int64_t origin = a.base + static_cast<int64_t>(u * static_cast<uint64_t>(kWaveTile) * kWavesPerBlock * kPTiles) - kFPre;
    uint32_t start = 0;
    bool bad = false, long_hit = false;
    for (int j = 0; j < kPTiles; j++) {
      const uint32_t n = s_cnt[pp][wave][j];
      const uint64_t dst = base + s_qbase[pp][j * kWavesPerBlock + wave];
      const int64_t tb = origin + static_cast<int64_t>(j * kWavesPerBlock + wave) * kWaveTile;
      for (uint32_t i = lane0; i < n; i += 64) {
        const uint32_t r = start + i;
        if (r < static_cast<uint32_t>(kPRows)) {
          const uint32_t v = s_row[pp][wave][r];
          const uint32_t st = v & 0xFFFFu, en = v >> 16;
          bad = bad || st >= en;
          long_hit = long_hit || (a.max_len != 0 && en - st > a.max_len);
          if (a.out != nullptr && dst + i < a.cap) {
            longlong2 o; o.x = tb + st; o.y = tb + en;
            *reinterpret_cast<longlong2*>(a.out + (dst + i) * a.row_width) = o;
          }
        }
      }
      start += n;
    }
    if (__ballot(bad) != 0ull) fallback |= 2u;                        // a start that was not found (fields_rows)
    if (__ballot(long_hit) != 0ull && lane0 == 0) raise_err(a.err, kErrLongMatch);
  };

  while (unit < nunits) {
    uint32_t next_ticket = 0;
    if (tid == 0) next_ticket = atomicAdd(tickets + myc, 1u);         // answer used behind the tiles
    auto tile_lo_of = [&](int jj) { return (unit * (kWavesPerBlock * kPTiles) + static_cast<uint64_t>(jj) * kWavesPerBlock + wave) * static_cast<uint64_t>(kWaveTile); };
    u32x4 x[4];
    int32_t nvalid_cur = 0;
    uint32_t nrows_w = 0;
    lane = lane0;
    fields_first_loads(x, fields_window(a.hay, a.len, tile_lo_of(0), true, nvalid_cur), lane, unit == 0 && wave == 0);
    for (int j = 0; j < kPTiles; j++) {
      lane = lane0;
      asm volatile("" : "+v"(lane));
      int32_t nvalid_next = 0;
      const __amdgpu_buffer_rsrc_t rnext = fields_window(a.hay, a.len, tile_lo_of(j + 1), j + 1 < kPTiles, nvalid_next);
      uint32_t d0, d1, p0, p1;
      fields_words<KD, KP>(x, rnext, lane, s_d[wave], s_p[wave], nvalid_cur, dlo4, dhi4, plo4, phi4, sink, d0, d1, p0, p1);
      nvalid_cur = nvalid_next;
      const FieldsTile t = fields_core<K>(d0, d1, p0, p1);
      if (t.ovf) fallback |= 1u;
      const uint32_t c = static_cast<uint32_t>(__popc(t.e0)) + static_cast<uint32_t>(__popc(t.e1));
      const uint32_t incl = wave_inclusive_sum_fused(c);
      const uint32_t tot = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
      if (tot != 0 && want_rows)
        fields_rows(t, lane, s_row[par][wave], nrows_w + incl - c, [](uint32_t r) { return min(r, static_cast<uint32_t>(kPRows - 1)); });   // overflow: flagged below, rows void
      if (lane == 0) s_cnt[par][wave][j] = tot;
      nrows_w += tot;
    }
    if (nrows_w > static_cast<uint32_t>(kPRows)) fallback |= 16u;
    __syncthreads();
    // ---- order the unit's rows: wave-tile q = j * 4 + wave; exclusive prefix over q (wave 0)
    uint32_t total = 0;
    if (tid < 64) {
      const int q = tid;
      const uint32_t v = (q < kWavesPerBlock * kPTiles) ? s_cnt[par][q % kWavesPerBlock][q / kWavesPerBlock] : 0u;
      const uint32_t incl = wave_inclusive_sum(v);
      if (q < kWavesPerBlock * kPTiles) s_qbase[par][q] = incl - v;
      total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
      lookback_publish(a.status, unit, total, a.epoch);               // the units behind can count on it; nobody waits for us
      if (tid == 0) s_next = ticket_to_unit(next_ticket);
      if (have_prev) {                                                // the unit before: its predecessors were published a unit ago
        const uint64_t b = lookback_resolve(a.status, a.total, a.err, prev_unit, nunits, prev_total, a.epoch);
        if (tid == 0) s_base = b;
      }
    }
    __syncthreads();
    if (have_prev && want_rows) write_rows(par ^ 1, prev_unit, s_base);
    prev_total = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(total)));   // (wave 0's value; the other waves do not use it)
    prev_unit = unit;
    have_prev = true;
    par ^= 1;
    unit = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_next >> 32))) << 32) |
           static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_next)));
  }
  if (have_prev) {                                                    // the workgroup's last unit: the wait of the grouped kernel, once
    if (tid < 64) {
      const uint64_t b = lookback_resolve(a.status, a.total, a.err, prev_unit, nunits, prev_total, a.epoch);
      if (tid == 0) s_base = b;
    }
    __syncthreads();
    if (want_rows) write_rows(par ^ 1, prev_unit, s_base);
  }
  if (sink == 0x12345u) fallback |= 4u;
  if (fallback != 0 && lane0 == 0) raise_err(a.err, 8u | (fallback << 8));
}

// STREAMING variant (the default): a persistent grid.  In round r wave W (of NW producer waves) takes UNIT r * NW + W = the
// kSUnit consecutive wave-tiles from (r * NW + W) * kSUnit, so that at every moment the resident waves read ONE dense window
// of the haystack (6.2 TB/s against 3.7 TB/s for 120 KiB per workgroup; units of 1..8 tiles stream alike,
// profiles/r03_fields_ablation.txt).  No barrier, no look-back: a wave publishes the row count of its unit, keeps the rows
// in its LDS ring, and writes them out a few tiles later when the scan server (stream_common.hpp) has published the unit's
// base.  Why units of several tiles: every word exchanged with the server is an agent-scope (sc1) access that travels to
// the memory side — with one count store and four poll loads per TILE they cost 0.10 ms per GiB, more than the scan.
// a.cnt16 = count words, a.status2 = base words (one each per unit), a.status = the server's super-batch words,
// a.ngroups = wave-tiles.
constexpr int kSUnit = CXG_UNIT;                     // wave-tiles per unit
#ifndef CXG_RING
#define CXG_RING 960
#endif
constexpr int kSRing = CXG_RING;                     // rows a wave can hold back (3 840 bytes: with the bitmaps 19.7 KiB of LDS per workgroup, 8 workgroups per CU)
constexpr int kSPend = 8;                            // units a wave can hold back
template <int K, int KD, int KP>
__global__ __launch_bounds__(kThreads, CXG_FIELDS_WAVES) void k_scan_fields_stream(ScanArgs a) {
  __shared__ __attribute__((aligned(16))) uint64_t s_d[kWavesPerBlock][64];
  __shared__ __attribute__((aligned(16))) uint64_t s_p[kWavesPerBlock][64];
  __shared__ uint32_t s_row[kWavesPerBlock][kSRing];                  // start | end << 16, relative to the window of the unit's first tile
  __shared__ uint32_t s_punit[kWavesPerBlock][kSPend];                // held-back units: index (< 2^31 per launch) ...
  __shared__ uint32_t s_pcnt[kWavesPerBlock][kSPend];                 // ... and row count

  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint64_t ntiles = a.ngroups;
  const uint64_t nunits = (ntiles + kSUnit - 1) / kSUnit;
  if (blockIdx.x < kScanWorkgroups) {                                 // the scan server (its LDS: the hand-off slots, in s_d's place)
    if (CXG_FABL < 5) stream_scanner(a.cnt16, a.status2, a.status, nunits, a.epoch4, a.epoch, a.total, a.err, reinterpret_cast<StreamChain*>(&s_d[0][0]));
    return;
  }
  int lane = lane0;
  const uint64_t NW = static_cast<uint64_t>(gridDim.x - kScanWorkgroups) * kWavesPerBlock;
  const uint64_t W = static_cast<uint64_t>(blockIdx.x - kScanWorkgroups) * kWavesPerBlock + wave;
  const ChainAux* gch = reinterpret_cast<const ChainAux*>(a.chain);
  const uint32_t dlo4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gch->cls_lo[0] * 0x01010101u)));
  const uint32_t dhi4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((0x7Fu - gch->cls_hi[0]) * 0x01010101u)));
  const uint32_t plo4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gch->cls_lo[1] * 0x01010101u)));
  const uint32_t phi4 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((0x7Fu - gch->cls_hi[1]) * 0x01010101u)));
  const bool want_rows = a.out != nullptr || a.max_len != 0;
  uint32_t fallback = 0;
  uint32_t ring_head = 0, ring_tail = 0, ring_used = 0;               // ring_used rows from ring_head on are held back; head and tail in [0, kSRing)
  uint32_t pend_head = 0, pend_tail = 0;                              // units [head, tail) likewise (the unit being scanned is not in the list yet)
  uint32_t unit_rows = 0;                                             // rows of the unit being scanned, ring [ring_tail - unit_rows, ring_tail)
  bool dead = false;                                                  // a watchdog fired: stop waiting, the launch is void
  uint32_t* const rows = s_row[wave];

  // Rows leave the ring in the order of their units.  State of the OLDEST held-back unit: cur_known = the base of its rows
  // (cur_base) has arrived, cur_done = rows of it already written.
  //
  // The memory instructions of a round (= one tile) form a STATIC sequence — 4 window loads, 1 count store, 2 row stores,
  // 2 poll loads, none of them under a branch: lanes that have nothing to store, and polls with nothing to ask, are sent
  // out of range of a buffer resource (the hardware drops them without a memory access).  Reason: the compiler's wait-count
  // pass is path-insensitive; with a store under `if`, the wait for a window vector assumes the path WITHOUT the store
  // and so also waits for the store's acknowledgement (a round trip to the memory side, ~2 us).  With a static sequence
  // every wait names exactly the instruction it needs (vmcnt(8) for the window vectors).
  bool cur_known = false;
  uint64_t cur_base = 0;
  uint32_t cur_done = 0;
  const uint32_t row_bytes = a.row_width * 8u;
  // rows [cur_done, cur_done + 64) of the oldest unit: one 16-byte store per lane, out-of-range lanes dropped by the hardware
  auto ring_at = [](uint32_t r) -> uint32_t { return r >= static_cast<uint32_t>(kSRing) ? r - static_cast<uint32_t>(kSRing) : r; };   // r < 2 * kSRing
  auto flush_step = [&]() {
    const uint32_t slot = pend_head & (kSPend - 1);
    const uint64_t u = s_punit[wave][slot];
    const uint32_t n = cur_known ? s_pcnt[wave][slot] : 0u;
    const int64_t tb = a.base + static_cast<int64_t>(u * static_cast<uint64_t>(kSUnit) * kWaveTile) - kFPre;
    // resource over the rows this step may write: [cur_base + cur_done, min(cur_base + n, cap))
    const uint64_t first = cur_base + cur_done;
    uint64_t room = (a.out != nullptr && cur_known && a.cap > first) ? a.cap - first : 0ull;
    if (room > n - cur_done) room = n - cur_done;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t*>(a.out) + (a.out ? first * row_bytes : 0ull), 0,
                                                                      static_cast<int>(room * row_bytes), 0x00020000);
    const uint32_t i = static_cast<uint32_t>(lane0);
    const uint32_t v = rows[ring_at(ring_head + cur_done + i)];
    const uint32_t s = v & 0xFFFFu, e = v >> 16;
    const bool live = cur_done + i < n;
    const int64_t ms = tb + s, me = tb + e;
    u32x4 o;
    o.x = static_cast<uint32_t>(ms); o.y = static_cast<uint32_t>(ms >> 32); o.z = static_cast<uint32_t>(me); o.w = static_cast<uint32_t>(me >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(o, ro, i * row_bytes, 0, 0);
    if (__ballot(live && s >= e) != 0ull) fallback |= 2u;            // a start that was not found (fields_rows)
    if (a.max_len != 0 && __ballot(live && e - s > a.max_len) != 0ull && lane0 == 0) raise_err(a.err, kErrLongMatch);
    if (cur_known) {
      cur_done = min(n, cur_done + 64u);
      if (cur_done == n) { ring_head = ring_at(ring_head + n); ring_used -= n; pend_head++; cur_done = 0; cur_known = false; }
    }
  };
  // the answer to a poll for the oldest unit's base; `asked`: the word answers a poll for it
  auto take_base = [&](uint64_t bw, bool asked) {
    const uint32_t blo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(bw)));
    const uint32_t bhi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(bw >> 32)));
    const uint64_t u = (static_cast<uint64_t>(bhi) << 32) | blo;
    if (!asked || cur_known || (u & kFlagMask) != kStreamReady || (u & kEpochMask) != (static_cast<uint64_t>(a.epoch) << kEpochShift)) return;
    cur_base = u & kValueMask;
    cur_known = true;
  };
  // the slow way (ring or list full, and at the end): wait for the oldest unit's base and write all its rows
  auto flush_blocking = [&]() {
    const uint64_t u = s_punit[wave][pend_head & (kSPend - 1)];
    uint32_t spins = 0;
    while (!cur_known && !dead) {
      take_base(__hip_atomic_load(a.status2 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), true);
      if (cur_known) break;
      if (++spins > kSpinLimit) { if (lane0 == 0) raise_err(a.err, 2u); dead = true; break; }
      __builtin_amdgcn_s_sleep(4);
    }
    if (dead) { const uint32_t n = s_pcnt[wave][pend_head & (kSPend - 1)]; ring_head = ring_at(ring_head + n); ring_used -= n; pend_head++; cur_done = 0; cur_known = false; return; }
    const uint32_t want = pend_head + 1;
    while (pend_head != want) flush_step();
  };

  u32x4 x[4];
  uint32_t sink = 0;
  int32_t nvalid_cur = 0;
  uint64_t step = 0;                                                  // tiles this wave has taken
  auto tile_of = [&](uint64_t k) -> uint64_t { return ((k / kSUnit) * NW + W) * kSUnit + (k % kSUnit); };
  uint64_t t = tile_of(0);
  // The polls of the previous round — in flight across a whole round — for the oldest held-back unit (A) and the one behind
  // it (B): two units can leave per round, so a wave that fell behind the scan server for a moment catches up again.
  u32x2 pollA = {0u, 0u}, pollB = {0u, 0u};
  bool askedA = false, askedB = false;
  auto u64of = [](const u32x2& w) { return (static_cast<uint64_t>(w.y) << 32) | w.x; };
  // a poll = one 8-byte sc1 load of a base word through a resource of 8 bytes; nothing to ask: a resource of 0 bytes (no access)
  auto poll = [&](uint64_t u, bool ask) -> u32x2 {
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t*>(a.status2 + (ask ? u : 0)), 0, ask ? 8 : 0, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b64(rb, 0, 0, 16 /*sc1: agent scope*/);
  };
  fields_first_loads(x, fields_window(a.hay, a.len, t * static_cast<uint64_t>(kWaveTile), t < ntiles, nvalid_cur), lane, t == 0);
  if (CXG_FABL < 5) {
    // The round's other memory instructions once in front of the loop, all out of range / unasked: the loop is then entered
    // with the same instructions in flight, in the same order, as a round leaves behind — otherwise the first round's
    // shorter list decides the loop's wait counts, and every round waits for its predecessor's polls with its window loads.
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.hay), 0, 0, 0x00020000);
    const u32x4 z = {0u, 0u, 0u, 0u};
    __builtin_amdgcn_raw_buffer_store_b16(static_cast<short>(0), rz, 64, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(z, rz, 64, 0, 0);
    __builtin_amdgcn_sched_barrier(0);                              // (two stores, not one merged)
    __builtin_amdgcn_raw_buffer_store_b128(z, rz, 80 + (lane0 << 4), 0, 0);
    pollA = poll(0, false);
    __builtin_amdgcn_sched_barrier(0);
    pollB = poll(0, false);
  }
  for (; t < ntiles; t = tile_of(++step)) {
    lane = lane0;
    asm volatile("" : "+v"(lane));
    int32_t nvalid_next = 0;
    const uint64_t tn = tile_of(step + 1);
    const __amdgpu_buffer_rsrc_t rnext = fields_window(a.hay, a.len, tn * static_cast<uint64_t>(kWaveTile), tn < ntiles, nvalid_next);
    uint32_t d0, d1, p0, p1;
    fields_words<KD, KP>(x, rnext, lane, s_d[wave], s_p[wave], nvalid_cur, dlo4, dhi4, plo4, phi4, sink, d0, d1, p0, p1);
    nvalid_cur = nvalid_next;
    if (CXG_FABL >= 5) continue;
    const uint32_t uo = static_cast<uint32_t>(step % kSUnit);        // tile of the unit
    const uint64_t unit = t / kSUnit;
    const bool unit_last = uo == static_cast<uint32_t>(kSUnit - 1) || t + 1 >= ntiles;
    uint32_t tot = 0;
    FieldsTile ft{0, 0, 0, 0, false};
    uint32_t c = 0, incl = 0;
    if (CXG_FABL < 4) {
      ft = fields_core<K>(d0, d1, p0, p1);
      if (ft.ovf) fallback |= 1u;
      c = static_cast<uint32_t>(__popc(ft.e0)) + static_cast<uint32_t>(__popc(ft.e1));
      incl = wave_inclusive_sum_fused(c);
      tot = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
      if (CXG_FABL >= 1) tot = 0;
    }
    if (unit_rows + tot > static_cast<uint32_t>(kSRing)) { fallback |= 16u; tot = 0; }   // more rows in one unit than the ring holds: match-dense input
    if (want_rows) {
      // room in the ring (only when the scan server is far behind: the slow way)
      while (pend_tail != pend_head && ring_used + tot > static_cast<uint32_t>(kSRing)) {
        if (a.prof && lane0 == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.prof), 1ull);   // CXG_PROF: rounds that had to wait for the scan server
        askedA = askedB = false;                                      // (the polls in flight answer for units that are about to leave)
        flush_blocking();
      }
      if (tot != 0) {
        const uint32_t shift = uo * static_cast<uint32_t>(kWaveTile) * 0x00010001u;   // rows relative to the unit's first window
        fields_rows(ft, lane, rows, ring_tail + incl - c, ring_at, shift);
      }
      ring_tail = ring_at(ring_tail + tot);
      ring_used += tot;
    }
    unit_rows += tot;
    if (want_rows && unit_last) {                                     // the unit joins the list of held-back units
      while (pend_tail - pend_head >= static_cast<uint32_t>(kSPend)) {
        if (a.prof && lane0 == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.prof), 1ull);
        askedA = askedB = false;
        flush_blocking();
      }
      if (lane == 0) { s_punit[wave][pend_tail & (kSPend - 1)] = static_cast<uint32_t>(unit); s_pcnt[wave][pend_tail & (kSPend - 1)] = unit_rows; }
      pend_tail++;
    }
    wave_lds_sync();
    // ---- the round's memory instructions behind the window loads, always in this order
    {                                                                 // 1. the count of the unit, behind its last tile: the scan server can move on (lane 0 stores, the others — and every lane before the last tile — are out of range)
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t*>(a.cnt16 + unit), 0, unit_last ? 2 : 0, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b16(static_cast<short>((a.epoch4 << 11) | unit_rows), rc, lane == 0 ? 0 : 64, 0, 16 /*sc1: agent scope*/);
    }
    if (unit_last) unit_rows = 0;
    // 2. rows of the two oldest held-back units whose bases last round's polls brought (asked a whole round ago: the words
    //    travel to the memory side and back, ~3 us); 64 rows per step
    {
      const uint32_t before = pend_head;
      take_base(u64of(pollA), askedA);
      flush_step();
      take_base(u64of(pollB), askedB && pend_head != before);          // B answers for the unit behind A: only if A is gone
      flush_step();
    }
    {                                                                 // 3. ask for the bases of the (now) two oldest units; looked at in the next round
      const uint32_t np = pend_tail - pend_head;
      askedA = want_rows && np != 0 && !cur_known;
      askedB = want_rows && np > 1u;
      pollA = poll(s_punit[wave][pend_head & (kSPend - 1)], askedA);
      pollB = poll(s_punit[wave][(pend_head + 1) & (kSPend - 1)], askedB);
    }
  }
  if (CXG_FABL >= 4 && sink == 0x12345u) fallback |= 4u;
  take_base(u64of(pollA), askedA);
  take_base(u64of(pollB), false);
  while (pend_tail != pend_head) flush_blocking();
  if (fallback != 0 && lane0 == 0) raise_err(a.err, 8u | (fallback << 8));
}

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

namespace {
template <int K, int KD, int KP>
int fields_capacity_of(int device, int mode) {                      // workgroups of the streaming (1) / persistent (2) kernel the device holds at once
  int occ = 0, cus = 0;
  const hipError_t e = mode == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_scan_fields_stream<K, KD, KP>, kThreads, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_scan_fields_pers<K, KD, KP>, kThreads, 0);
  if (e != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
  // MI355X_MICROARCH.md "Correctness boundaries": the occupancy API can be one block per CU high for kernels with more than
  // 80 SGPRs; these kernels are built to stay at or below 8 waves per SIMD with <= 80 SGPRs, and 8 workgroups of 4 waves are
  // the hardware's wave limit anyway
  if (occ > 8) occ = 8;
  return occ * cus;
}
template <int K>
void launch_fields_k(const ScanArgs& a, uint32_t kd, uint32_t kp, int mode, int device, int* capacity, dim3 grid, dim3 block, hipStream_t stream) {
  const bool dd = kd == kClsDigit, pb = kp == kClsByte;
  if (capacity) {
    *capacity = dd && pb ? fields_capacity_of<K, kClsDigit, kClsByte>(device, mode) : dd ? fields_capacity_of<K, kClsDigit, kClsRange>(device, mode)
              : pb ? fields_capacity_of<K, kClsRange, kClsByte>(device, mode) : fields_capacity_of<K, kClsRange, kClsRange>(device, mode);
    return;
  }
  if (mode == 1) {
    if (dd && pb) hipLaunchKernelGGL((k_scan_fields_stream<K, kClsDigit, kClsByte>), grid, block, 0, stream, a);
    else if (dd) hipLaunchKernelGGL((k_scan_fields_stream<K, kClsDigit, kClsRange>), grid, block, 0, stream, a);
    else if (pb) hipLaunchKernelGGL((k_scan_fields_stream<K, kClsRange, kClsByte>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_scan_fields_stream<K, kClsRange, kClsRange>), grid, block, 0, stream, a);
    return;
  }
  if (mode == 2) {
    if (dd && pb) hipLaunchKernelGGL((k_scan_fields_pers<K, kClsDigit, kClsByte>), grid, block, 0, stream, a);
    else if (dd) hipLaunchKernelGGL((k_scan_fields_pers<K, kClsDigit, kClsRange>), grid, block, 0, stream, a);
    else if (pb) hipLaunchKernelGGL((k_scan_fields_pers<K, kClsRange, kClsByte>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_scan_fields_pers<K, kClsRange, kClsRange>), grid, block, 0, stream, a);
    return;
  }
  if (dd && pb) hipLaunchKernelGGL((k_scan_fields_wave<K, kClsDigit, kClsByte>), grid, block, 0, stream, a);
  else if (dd) hipLaunchKernelGGL((k_scan_fields_wave<K, kClsDigit, kClsRange>), grid, block, 0, stream, a);
  else if (pb) hipLaunchKernelGGL((k_scan_fields_wave<K, kClsRange, kClsByte>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((k_scan_fields_wave<K, kClsRange, kClsRange>), grid, block, 0, stream, a);
}
void dispatch_fields(const ScanArgs& a, int k, int mode, int device, int* capacity, dim3 grid, hipStream_t stream) {
  const ChainAux& c = *reinterpret_cast<const ChainAux*>(a.chain);
  const dim3 block(kThreads);
  switch (k) {
    case 2: launch_fields_k<2>(a, c.cls_kind[0], c.cls_kind[1], mode, device, capacity, grid, block, stream); break;
    case 3: launch_fields_k<3>(a, c.cls_kind[0], c.cls_kind[1], mode, device, capacity, grid, block, stream); break;
    case 4: launch_fields_k<4>(a, c.cls_kind[0], c.cls_kind[1], mode, device, capacity, grid, block, stream); break;
    default: break;
  }
}
}  // namespace

// Grouped kernel: a.ngroups = number of 120 KiB groups (one workgroup each).
hipError_t launch_scan_fields_wave(const ScanArgs& a, hipStream_t stream) {
  const int k = fields_shape(*reinterpret_cast<const ChainAux*>(a.chain));
  if (!k) return hipErrorInvalidValue;
  dispatch_fields(a, k, 0, 0, nullptr, dim3(static_cast<unsigned>(a.ngroups)), stream);
  return hipGetLastError();
}
// Workgroups of the streaming (mode 1) / persistent (mode 2) kernel that are resident at once on `device` (0: unknown).
int fields_capacity(const ScanArgs& a, int device, int mode) {
  const int k = fields_shape(*reinterpret_cast<const ChainAux*>(a.chain));
  int cap = 0;
  if (k) dispatch_fields(a, k, mode, device, &cap, dim3(1), nullptr);
  return cap;
}
// Streaming kernel: a.ngroups = number of wave-tiles; `producers` workgroups + the scan server's (stream_scan_workgroups()),
// all of which must be resident together (<= fields_capacity).
int stream_scan_workgroups() { return kScanWorkgroups; }
hipError_t launch_scan_fields_stream(const ScanArgs& a, unsigned producers, hipStream_t stream) {
  const int k = fields_shape(*reinterpret_cast<const ChainAux*>(a.chain));
  if (!k || producers == 0) return hipErrorInvalidValue;
  dispatch_fields(a, k, 1, 0, nullptr, dim3(producers + kScanWorkgroups), stream);
  return hipGetLastError();
}
// Persistent kernel: a.ngroups = units of fields_pers_unit_bytes() bytes, a.ticket zeroed; any grid (no residency condition).
uint64_t fields_pers_unit_bytes() { return static_cast<uint64_t>(kWaveTile) * kWavesPerBlock * kPTiles; }
hipError_t launch_scan_fields_pers(const ScanArgs& a, unsigned workgroups, hipStream_t stream) {
  const int k = fields_shape(*reinterpret_cast<const ChainAux*>(a.chain));
  if (!k || workgroups == 0) return hipErrorInvalidValue;
  dispatch_fields(a, k, 2, 0, nullptr, dim3(workgroups), stream);
  return hipGetLastError();
}

}  // namespace cxgdev
