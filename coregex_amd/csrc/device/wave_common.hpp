// wave_common.hpp — wave64-local building blocks shared by the wave kernels (scan_chain_wave.hip,
// scan_teddy_wave.hip): DPP neighbour moves, lane reversal, LDS hand-off inside one wave, carry-mask add,
// prefix sum, per-lane word masks.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cxgdev {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// A class that is a union of up to four ASCII ranges (\w = [0-9A-Z_a-z]): range tests share x|0x80 and x&0x7F.
struct SetRanges { uint32_t n; uint32_t lo4[4], hi4[4]; };   // splat bounds, hi4 = (0x7F - hi) splat
__device__ __forceinline__ uint32_t notset4(uint32_t x, const SetRanges& r) {
  const uint32_t xh = x | 0x80808080u, xl = x & 0x7F7F7F7Fu;
  uint32_t in = (xh - r.lo4[0]) & ~(xl + r.hi4[0]);
  if (r.n > 1) in |= (xh - r.lo4[1]) & ~(xl + r.hi4[1]);
  if (r.n > 2) in |= (xh - r.lo4[2]) & ~(xl + r.hi4[2]);
  if (r.n > 3) in |= (xh - r.lo4[3]) & ~(xl + r.hi4[3]);
  return ~(in & ~x) & 0x80808080u;
}

// ---- class plans (round 5) ---------------------------------------------------------------------------------------------------
// The same union of ASCII ranges evaluated with fewer instructions per dword: with t = x & 0x7F7F7F7F (shared),
//   * an X term — a single byte, or a range [lo, lo + span] whose lo has its low ceil(log2(span + 1)) bits clear (the digits:
//     0x30..0x39): (t ^ lo) + (0x7F - span) carries into bit 7 exactly for the bytes OUTSIDE it — one v_xad_u32;
//   * an R term — any other range: (s + 0x80 - lo) & ~(s + 0x7F - hi) has bit 7 set exactly inside it — two adds;
//   * FOLD — the R terms [lo, hi] and [lo + 0x20, hi + 0x20] with 0x40 <= lo, hi <= 0x5F (`A-Z` and `a-z`) are ONE R term over
//     s = x & 0x5F5F5F5F (bit 5 cleared).
// `\w` = [0-9A-Z_a-z] is two X terms (digits, `_`) and one folded R term: 9 instructions per dword instead of 15.
// (struct ClassPlan: scan_dfa.h, beside ScanArgs)
__host__ __device__ inline ClassPlan plan_class(uint32_t n, const uint8_t* lo, const uint8_t* hi) {   // run on the HOST per launch (ScanArgs::plan: kernel arguments, scalar loads)
  ClassPlan p;
  p.nx = p.nr = p.fold = 0; p.ok = 1;
  uint32_t rl[4], rh[4], nrr = 0;
  for (uint32_t i = 0; i < 4; i++) { p.xc[i] = p.xk[i] = p.ra[i] = p.rb[i] = 0; rl[i] = rh[i] = 0; }
  for (uint32_t i = 0; i < n && i < 4u; i++) {
    const uint32_t a = lo[i], b = hi[i];
    if (b > 0x7Fu || a > b) { p.ok = 0; continue; }
    const uint32_t span = b - a;
    uint32_t bits = 0;
    while ((1u << bits) <= span) bits++;
    if ((a & ((1u << bits) - 1u)) == 0u) { p.xc[p.nx] = a * 0x01010101u; p.xk[p.nx] = (0x7Fu - span) * 0x01010101u; p.nx++; }
    else { rl[nrr] = a; rh[nrr] = b; nrr++; }
  }
  // fold one pair of R ranges that differ by 0x20 (upper / lower case)
  for (uint32_t i = 0; i < nrr && !p.fold; i++)
    for (uint32_t j = 0; j < nrr && !p.fold; j++)
      if (i != j && rl[i] >= 0x40u && rh[i] <= 0x5Fu && rl[j] == rl[i] + 0x20u && rh[j] == rh[i] + 0x20u) {
        p.fold = 1;
        const uint32_t a = rl[i], b = rh[i];
        uint32_t kl[4], kh[4], kn = 0;
        for (uint32_t q = 0; q < nrr; q++) if (q != i && q != j) { kl[kn] = rl[q]; kh[kn] = rh[q]; kn++; }
        rl[0] = a; rh[0] = b;
        for (uint32_t q = 0; q < kn; q++) { rl[q + 1] = kl[q]; rh[q + 1] = kh[q]; }
        nrr = kn + 1;
      }
  for (uint32_t i = 0; i < nrr; i++) { p.ra[i] = (0x80u - rl[i]) * 0x01010101u; p.rb[i] = (0x7Fu - rh[i]) * 0x01010101u; }
  p.nr = nrr;
  return p;
}
__device__ __forceinline__ uint32_t xad_u32(uint32_t t, uint32_t c, uint32_t k) {   // (t ^ c) + k; c uniform (the one scalar operand), k in a register
  uint32_t r;
  asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(t), "s"(c), "v"(k));
  return r;
}
// 0x80 in every byte of x that is NOT a member (notset4's convention), for a plan of exactly NX X terms and NR R terms.
template <int NX, int NR, bool FOLD>
__device__ __forceinline__ uint32_t notplan4(uint32_t x, const ClassPlan& p) {
  const uint32_t t = x & 0x7F7F7F7Fu;
  uint32_t in = 0;
#pragma unroll
  for (int i = 0; i < NX; i++) in |= ~xad_u32(t, p.xc[i], p.xk[i]);
#pragma unroll
  for (int i = 0; i < NR; i++) {
    const uint32_t s = (FOLD && i == 0) ? (x & 0x5F5F5F5Fu) : t;
    in |= (s + p.ra[i]) & ~(s + p.rb[i]);
  }
  return ~(in & ~x) & 0x80808080u;
}

// The shapes the kernels instantiate (a uniform switch around the classification of a whole window); 0: notset4.
__host__ __device__ inline int plan_shape(const ClassPlan& p) {
  if (!p.ok) return 0;
  const uint32_t key = p.nx * 100u + p.nr * 10u + p.fold;
  switch (key) {
    case 211: return 1;   // \w
    case 111: return 2;   // [A-Za-z0-9]
    case 11: return 3;    // [A-Za-z]
    case 100: return 4;   // \d, one byte
    case 10: return 5;    // [a-z]
    case 200: return 6;   // two X terms
    case 110: return 7;   // [a-z0-9]
    case 300: return 8;   // [.,;]
    default: return 0;
  }
}
template <int SHAPE>
__device__ __forceinline__ uint32_t notshape4(uint32_t x, const ClassPlan& p, const SetRanges& r) {
  if (SHAPE == 1) return notplan4<2, 1, true>(x, p);
  if (SHAPE == 2) return notplan4<1, 1, true>(x, p);
  if (SHAPE == 3) return notplan4<0, 1, true>(x, p);
  if (SHAPE == 4) return notplan4<1, 0, false>(x, p);
  if (SHAPE == 5) return notplan4<0, 1, false>(x, p);
  if (SHAPE == 6) return notplan4<2, 0, false>(x, p);
  if (SHAPE == 7) return notplan4<1, 1, false>(x, p);
  if (SHAPE == 8) return notplan4<3, 0, false>(x, p);
  return notset4(x, r);
}
// 16 NOT-member bits of a 16-byte vector (bits above 15 are garbage: ds_write_b16 drops them)
template <int SHAPE>
__device__ __forceinline__ uint32_t notshape16(const u32x4& x, const ClassPlan& p, const SetRanges& r) {
  const uint32_t lo = __builtin_amdgcn_udot4(notshape4<SHAPE>(x.y, p, r), 0x80402010u, __builtin_amdgcn_udot4(notshape4<SHAPE>(x.x, p, r), 0x08040201u, 0u, false), false);
  const uint32_t hi = __builtin_amdgcn_udot4(notshape4<SHAPE>(x.w, p, r), 0x80402010u, __builtin_amdgcn_udot4(notshape4<SHAPE>(x.z, p, r), 0x08040201u, 0u, false), false);
  return (lo >> 7) | (hi << 1);
}
// `f.template operator()<SHAPE>()` for the plan's shape (uniform switch)
template <class F>
__device__ __forceinline__ void with_shape(int shape, F&& f) {
  switch (shape) {
    case 1: f.template operator()<1>(); break;
    case 2: f.template operator()<2>(); break;
    case 3: f.template operator()<3>(); break;
    case 4: f.template operator()<4>(); break;
    case 5: f.template operator()<5>(); break;
    case 6: f.template operator()<6>(); break;
    case 7: f.template operator()<7>(); break;
    case 8: f.template operator()<8>(); break;
    default: f.template operator()<0>(); break;
  }
}

// Neighbour-lane moves as DPP wavefront shifts (one VALU op per dword, no LDS crossbar round trip).
__device__ __forceinline__ uint32_t dpp_from_lower(uint32_t v) {  // lane i <- lane i-1 (lane 0 keeps its own)
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x138 /*wave_shr:1*/, 0xF, 0xF, false));
}
__device__ __forceinline__ uint32_t dpp_from_upper(uint32_t v) {  // lane i <- lane i+1 (lane 63 keeps its own)
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x130 /*wave_shl:1*/, 0xF, 0xF, false));
}
__device__ __forceinline__ uint64_t from_lower64(uint64_t v) {
  return (static_cast<uint64_t>(dpp_from_lower(static_cast<uint32_t>(v >> 32))) << 32) | dpp_from_lower(static_cast<uint32_t>(v));
}
__device__ __forceinline__ uint64_t from_upper64(uint64_t v) {
  return (static_cast<uint64_t>(dpp_from_upper(static_cast<uint32_t>(v >> 32))) << 32) | dpp_from_upper(static_cast<uint32_t>(v));
}
__device__ __forceinline__ uint64_t brev64(uint64_t v) {
  return (static_cast<uint64_t>(__brev(static_cast<uint32_t>(v))) << 32) | __brev(static_cast<uint32_t>(v >> 32));
}
// Value of lane 63-l: DPP row_mirror inside the rows of 16, then v_permlane16_swap / v_permlane32_swap (gfx950)
// to exchange the rows — registers only, no LDS round trip.
__device__ __forceinline__ uint32_t lane_reverse32(uint32_t v, int lane) {
  const uint32_t m = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x140 /*row_mirror*/, 0xF, 0xF, false));
  const auto r16 = __builtin_amdgcn_permlane16_swap(m, m, false, false);
  const uint32_t s16 = ((lane >> 4) & 1) ? r16[0] : r16[1];
  const auto r32 = __builtin_amdgcn_permlane32_swap(s16, s16, false, false);
  return (lane & 32) ? r32[0] : r32[1];
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
  return (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v >> 32), l))) << 32) |
         static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), l));
}
__device__ __forceinline__ void wave_lds_sync() {                // same-wave LDS hand-off: drain, no barrier
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);                            // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// s + (bit `lane` of mask): the per-lane carry-in comes straight from the scalar mask (v_addc with an SGPR-pair
// carry operand), two VALU ops for the 64-bit add instead of shift + and + add.
__device__ __forceinline__ uint64_t add_carry_mask(uint64_t s, unsigned long long mask) {
  uint32_t lo, hi;
  unsigned long long c;
  asm("v_addc_co_u32_e64 %0, %2, %3, 0, %5\n\tv_addc_co_u32_e64 %1, %2, %4, 0, %2"
      : "=&v"(lo), "=&v"(hi), "=&s"(c)
      : "v"(static_cast<uint32_t>(s)), "v"(static_cast<uint32_t>(s >> 32)), "s"(mask));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
// Inclusive prefix sum over the 64 lanes, all DPP (row shifts, then row broadcasts).
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111 /*row_shr:1*/, 0xF, 0xF, true));
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112 /*row_shr:2*/, 0xF, 0xF, true));
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114 /*row_shr:4*/, 0xF, 0xF, true));
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118 /*row_shr:8*/, 0xF, 0xF, true));
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142 /*row_bcast:15*/, 0xA, 0xF, false));
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143 /*row_bcast:31*/, 0xC, 0xF, false));
  return v;
}
// Inclusive prefix maximum of unsigned values over the 64 lanes, the same DPP ladder (lanes outside a shift contribute 0).
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v) {
  auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
  v = mx(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111 /*row_shr:1*/, 0xF, 0xF, true)));
  v = mx(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112 /*row_shr:2*/, 0xF, 0xF, true)));
  v = mx(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114 /*row_shr:4*/, 0xF, 0xF, true)));
  v = mx(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118 /*row_shr:8*/, 0xF, 0xF, true)));
  v = mx(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142 /*row_bcast:15*/, 0xA, 0xF, false)));
  v = mx(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143 /*row_bcast:31*/, 0xC, 0xF, false)));
  return v;
}
// bits [lo, hi] (inclusive, window bit indices) that fall into lane's word [64*lane, 64*lane+63]
__device__ __forceinline__ uint64_t word_range(int lane, int32_t lo, int32_t hi) {
  // bits >= a and <= b of the word, a/b relative to the word and clamped so that the shifts stay in range
  const int32_t a = lo - 64 * lane, b = hi - 64 * lane;
  const uint32_t ac = static_cast<uint32_t>(a < 0 ? 0 : (a > 64 ? 64 : a));        // 0..64: number of low bits to drop
  const uint32_t bc = static_cast<uint32_t>(b < -1 ? 0 : (b > 63 ? 64 : b + 1));   // 0..64: number of low bits to keep
  const uint64_t ge = ac >= 64u ? 0ull : (~0ull << ac);
  const uint64_t le = bc >= 64u ? ~0ull : ~(~0ull << bc);
  return ge & le;
}


}  // namespace cxgdev
