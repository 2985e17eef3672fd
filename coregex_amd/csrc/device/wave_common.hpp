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
