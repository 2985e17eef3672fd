// scan_teddy_wave.hip — FindAll for UseTeddy (exact literal alternation, Slim Teddy: <= 32 prefix-free literals),
// second generation: the wave is the unit of work (same skeleton as scan_chain_wave.hip).
//
// Reference semantics kept (meta/find_indices.go:925-951 -> prefilter.Teddy.FindMatch, prefilter/teddy.go:391-444,
// verifyBucket :532-550; FindAll advance meta/findall.go:267-275): the next match is at the first fingerprint
// candidate at or after `pos` at which a literal of a hit bucket compares equal; the search resumes at its end.
//
// One wave64 owns a wave-tile of 60 x 64 B = 3840 B and reads 256 B of halo behind it: 4096 B = 64 bitmap
// words, one per lane.
//   A  four buffer_load_dwordx4 per lane (zeros past the end of input), one tile ahead; the window is also kept
//      in the wave's LDS scratch for the verifier.  Per byte ONE LDS lookup T[b] = A | B<<8 | C<<16 | sync<<24
//      (A = lo[0]&hi[0], B = lo[1]&hi[1] of the reference's nibble masks, teddy.go:271-311; C = buckets with a
//      literal whose third byte is b — every literal has >= 3 bytes, and any superset of the true match starts
//      is a valid candidate set because candidates are verified exactly): candidate bit i =
//      A(b_i) & B(b_{i+1}) & C(b_{i+2}) != 0, synchronising bit i = b_i outside the literals' alphabet.  The
//      byte extraction, the three-way AND and the packing are SDWA operations (one VALU op each per byte);
//      16-bit pieces go through LDS and come back as one 64-bit word per lane.
//   O  ownership, wave-uniform, as in the chain kernel: with zA = first synchronising byte at >= -1 and zB =
//      first one at >= 3839 the tile owns the candidates in (zA, zB].
//   V  owned candidates are ranked (DPP prefix sum), listed in LDS and verified 64 at a time, one per lane:
//      buckets of the fingerprint low to high, literals of a bucket in id order, bytes from the LDS window.
//   D  FindAll order: a verified candidate that starts inside the previous emitted match is dropped
//      (prefix max of the ends; a rare serial path when a round contains such a candidate).
// Group structure, tickets, look-back and row write-out: block_common.hpp / scan_chain_wave.hip.
// Fallback flag (err bit 8: the host reruns the scan with scan_teddy.hip): no synchronising byte in a halo,
// > 192 owned candidates in a wave-tile, row buffer overflow.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_common.hpp"
#include "scan_dfa.h"
#include "walk.hpp"
#include "wave_common.hpp"

namespace cxgdev {

namespace {

constexpr int kTRows = 320;                       // rows buffered per wave per group
constexpr int kTCands = 192;                      // owned candidates listed per wave-tile
constexpr int kTAuxMax = 2048;
constexpr int32_t kTFar = 1 << 20;
constexpr int kWin = kWaveTile + kWaveHalo;       // 4096

// 16 flags (bytes of four dwords, each 0 or non-zero) -> 16 bits; NZ: test for non-zero first
__device__ __forceinline__ uint32_t nz80(uint32_t t) { return (((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u; }
__device__ __forceinline__ uint32_t gather16_nz(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3) {
  const uint32_t lo = __builtin_amdgcn_udot4(nz80(t1), 0x80402010u, __builtin_amdgcn_udot4(nz80(t0), 0x08040201u, 0u, false), false);
  const uint32_t hi = __builtin_amdgcn_udot4(nz80(t3), 0x80402010u, __builtin_amdgcn_udot4(nz80(t2), 0x08040201u, 0u, false), false);
  return (lo >> 7) | (hi << 1);
}
__device__ __forceinline__ uint32_t gather16_01(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3) {   // bytes are 0 or 1
  const uint32_t lo = __builtin_amdgcn_udot4(t1, 0x80402010u, __builtin_amdgcn_udot4(t0, 0x08040201u, 0u, false), false);
  const uint32_t hi = __builtin_amdgcn_udot4(t3, 0x80402010u, __builtin_amdgcn_udot4(t2, 0x08040201u, 0u, false), false);
  return lo | (hi << 8);
}

// SDWA helpers (sub-dword addressing: byte select on the sources, byte placement with preserve on the destination)
#define CXG_SDWA_ADDR(dst, w, B) \
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #B : "=v"(dst) : "v"(two), "v"(w))
#define CXG_SDWA_AND01(dst, ea, eb) \
  asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1" : "=v"(dst) : "v"(ea), "v"(eb))
#define CXG_SDWA_AND2_PACK(T, d, ec, K) \
  asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_" #K " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2" : "+v"(T) : "v"(d), "v"(ec))
#define CXG_SDWA_MOV3_PACK(S, e, K) \
  asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_" #K " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(S) : "v"(e))
// one byte step: candidate flag of position i (entries ei, ei1, ei2) and its synchronising flag into byte K
// (SYNC: the synchronising flags are only needed where the ownership bounds are searched — the first and the last KiB of the window)
#define CXG_TEDDY_STEP(T, S, ei, ei1, ei2, K) do { uint32_t d_; CXG_SDWA_AND01(d_, ei, ei1); CXG_SDWA_AND2_PACK(T, d_, ei2, K); if (SYNC) CXG_SDWA_MOV3_PACK(S, ei, K); } while (0)

}  // namespace

// VERIFY: the image is a UseDFA program behind its required literal prefix (walk.hpp kFlagPrefixLiteral): an occurrence
// of the literal is extended to the match end by walking the anchored forward DFA (table in dynamic LDS) over the
// window's bytes; a walk still alive at the window edge hands the scan to the DFA-pair kernel.
// DENSE: two tiles per wave instead of eight — four times the row-buffer room per tile — after a row-buffer overflow
// on match-dense input (capi_ladder.hip), as in scan_chain_wave.hip.
template <bool VERIFY, bool DENSE>
__global__ __launch_bounds__(kThreads, VERIFY ? 4 : 5) void k_scan_teddy_wave(ScanArgs a) {
  constexpr int tpw = DENSE ? kDenseTilesPerWave : kTilesPerWave;
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dfa[];   // VERIFY: [dfa_states][256]
  __shared__ __attribute__((aligned(16))) uint8_t s_aux[kTAuxMax];
  __shared__ uint32_t s_T[256];                                    // A | B<<8 | C<<16 | sync<<24 per byte value
  __shared__ uint8_t s_boff[16];                                   // bucket b: its literals are order[s_boff[b] .. s_boff[b+1])
  __shared__ uint32_t s_lit[32][6];                                // literal id (Slim Teddy): first 12 bytes as three dwords, and their masks
  __shared__ __attribute__((aligned(16))) uint8_t s_bytes[kWavesPerBlock][kWin + 16];
  __shared__ __attribute__((aligned(16))) uint64_t s_cw[kWavesPerBlock][2][64];   // candidate / synchronising bitmaps
  __shared__ uint16_t s_cpos[kWavesPerBlock][kTCands];
  __shared__ uint16_t s_rs[kWavesPerBlock][kTRows];
  __shared__ uint16_t s_re[kWavesPerBlock][kTRows];
  __shared__ uint8_t s_em[kWavesPerBlock][64];
  __shared__ uint16_t s_ce[kWavesPerBlock][64];
  __shared__ uint32_t s_cnt[kWavesPerBlock][kTilesPerWave];
  __shared__ uint32_t s_qbase[kWavesPerBlock * kTilesPerWave + 1];
  __shared__ uint64_t s_group;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = lane0;
  if (tid == 0) s_group = claim_group(a.static_groups != 0, a.ticket, a.ngroups);
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(a.blob);
  for (uint32_t i = tid; i < h->aux_len / 4 && i < kTAuxMax / 4; i += kThreads)
    reinterpret_cast<uint32_t*>(s_aux)[i] = reinterpret_cast<const uint32_t*>(a.blob + h->aux_off)[i];
  __syncthreads();
  const TeddyAux* ax = reinterpret_cast<const TeddyAux*>(s_aux);
  const uint16_t* t_ab = reinterpret_cast<const uint16_t*>(s_aux + ax->ab_off);
  const uint8_t* t_order = s_aux + ax->order_off;
  const uint8_t* t_lens = s_aux + ax->lens_off;
  const uint8_t* t_bucket = s_aux + ax->bucket_off;
  const uint16_t* t_off = reinterpret_cast<const uint16_t*>(s_aux + ax->off_off);
  const uint8_t* t_bytes = s_aux + ax->bytes_off;
  const uint32_t nlits = ax->nlits;
  const uint32_t look_pre = ax->looks & 0xFFu, look_post = (ax->looks >> 8) & 0xFFu;   // literals between assertions (walk.hpp TeddyAux::looks)
  const uint32_t dfa_start = VERIFY ? ax->dfa_start : 0u, dfa_fa = VERIFY ? ax->dfa_first_accept : 0u;
  if (VERIFY) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.blob + h->aux_off + ax->dfa_off);
    for (uint32_t i = tid; i < ax->dfa_states * 64u; i += kThreads) reinterpret_cast<uint32_t*>(s_dfa)[i] = src[i];
  }
  s_T[tid] = static_cast<uint32_t>(t_ab[tid]) | (((a.blob + h->info_off)[tid] & kInfoSync) ? 0x1000000u : 0u);
  __syncthreads();
  const bool fold = (ax->looks & kTeddyFold) != 0u;                  // case-insensitive set: lower-case literals, a letter matches both cases
  if (static_cast<uint32_t>(tid) < nlits) {                          // third-byte masks
    const uint32_t b3 = t_bytes[t_off[tid] + 2];
    atomicOr(&s_T[b3], 0x10000u << t_bucket[tid]);
    if (fold && b3 >= 'a' && b3 <= 'z') atomicOr(&s_T[b3 ^ 0x20u], 0x10000u << t_bucket[tid]);
  }
  if (static_cast<uint32_t>(tid) < nlits && tid < 32) {             // verification compares dwords (below)
    const uint8_t* lb = t_bytes + t_off[tid];
    const uint32_t len = t_lens[tid];
    for (uint32_t k = 0; k < 3; k++) {
      uint32_t L = 0, M = 0;
      for (uint32_t b = 0; b < 4; b++) if (4 * k + b < len) {
        const uint32_t c = lb[4 * k + b];
        L |= c << (8 * b);
        M |= ((fold && c >= 'a' && c <= 'z') ? 0xDFu : 0xFFu) << (8 * b);     // a folded letter: bit 5 does not count
      }
      s_lit[tid][k] = L; s_lit[tid][3 + k] = M;
    }
  }
  if (tid < 16) {                                                  // order[] is bucket-major: first index of every bucket
    uint32_t first = nlits;
    for (uint32_t k = nlits; k-- > 0;) if (t_bucket[t_order[k]] >= static_cast<uint32_t>(tid)) first = k;
    s_boff[tid] = static_cast<uint8_t>(first);
  }
  __syncthreads();
  const uint64_t group = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_group >> 32))) << 32) |
                         static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(s_group)));
  if (group >= a.ngroups) return;
  if (limit_reached_skip(a, group, &s_base)) return;                 // FindAll with n > 0 (block_common.hpp)
  uint32_t nrows_w = 0;                                            // wave-uniform
  uint32_t fallback = 0;
  uint32_t long_hit = 0, edge_hit = 0;                             // VERIFY, per lane: match longer than the UseBoth restart span / walk cut by the window

  u32x4 x[4];
  uint32_t xprev = 0;
  auto issue_loads = [&](int jj) {
    const uint64_t wtn = group * (kWavesPerBlock * tpw) + static_cast<uint64_t>(jj) * kWavesPerBlock + wave;
    const uint64_t lo = wtn * static_cast<uint64_t>(kWaveTile);
    int nrec = 0;
    if (jj < tpw && lo < a.len) {
      const uint64_t rem = a.len - lo;
      nrec = rem >= static_cast<uint64_t>(kWin) ? kWin : static_cast<int>((rem + 3) & ~3ull);
    }
    const int pre = (nrec && lo) ? 16 : 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.hay) + (nrec ? lo - pre : 0), 0, nrec + pre, 0x00020000);
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (lane + 64 * k) << 4, pre, 0);
    xprev = __builtin_amdgcn_raw_buffer_load_b32(rsrc, 0, pre ? 12 : nrec + pre, 0);
  };
  issue_loads(0);

  for (int j = 0; j < tpw; j++) {
    lane = lane0;
    asm volatile("" : "+v"(lane));                                  // see scan_chain_wave.hip: no hoisted-and-spilled lane constants
    const uint64_t wt = group * (kWavesPerBlock * tpw) + static_cast<uint64_t>(j) * kWavesPerBlock + wave;
    const uint64_t tile_lo = wt * static_cast<uint64_t>(kWaveTile);
    uint32_t emitted_here = 0;
    if (tile_lo < a.len) {
      const uint64_t remaining = a.len - tile_lo;
      const int32_t rend = remaining > 0x7FFF0000ull ? 0x7FFF0000 : static_cast<int32_t>(remaining);
      const int32_t stage = rend < kWin ? rend : kWin;

      // ---- A: window to LDS, table lookups, candidate and synchronising bits
      const uint32_t two = 2u;
      uint32_t e0[4], e1[4];                                        // entries of the first two bytes of every vector
#pragma unroll
      for (int k = 0; k < 4; k++) {
        *reinterpret_cast<u32x4*>(&s_bytes[wave][(lane + 64 * k) << 4]) = x[k];
        uint32_t a0, a1;
        CXG_SDWA_ADDR(a0, x[k].x, 0); CXG_SDWA_ADDR(a1, x[k].x, 1);
        e0[k] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(s_T) + a0);
        e1[k] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(s_T) + a1);
      }
      uint16_t* pc = reinterpret_cast<uint16_t*>(s_cw[wave][0]);
      uint16_t* ps = reinterpret_cast<uint16_t*>(s_cw[wave][1]);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        // entries of the two bytes behind the vector: lane+1 of the same load, or lane 0 of the next load
        uint32_t n0 = dpp_from_upper(e0[k]), n1 = dpp_from_upper(e1[k]);
        const uint32_t w0 = (k < 3) ? static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e0[k < 3 ? k + 1 : 3]), 0)) : 0u;
        const uint32_t w1 = (k < 3) ? static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e1[k < 3 ? k + 1 : 3]), 0)) : 0u;
        if (lane == 63) { n0 = w0; n1 = w1; }
        const uint8_t* Tb = reinterpret_cast<const uint8_t*>(s_T);
        uint32_t ad, e[18];
        e[0] = e0[k]; e[1] = e1[k]; e[16] = n0; e[17] = n1;
        CXG_SDWA_ADDR(ad, x[k].x, 2); e[2] = *reinterpret_cast<const uint32_t*>(Tb + ad);
        CXG_SDWA_ADDR(ad, x[k].x, 3); e[3] = *reinterpret_cast<const uint32_t*>(Tb + ad);
        CXG_SDWA_ADDR(ad, x[k].y, 0); e[4] = *reinterpret_cast<const uint32_t*>(Tb + ad);
        CXG_SDWA_ADDR(ad, x[k].y, 1); e[5] = *reinterpret_cast<const uint32_t*>(Tb + ad);
        CXG_SDWA_ADDR(ad, x[k].y, 2); e[6] = *reinterpret_cast<const uint32_t*>(Tb + ad);
        CXG_SDWA_ADDR(ad, x[k].y, 3); e[7] = *reinterpret_cast<const uint32_t*>(Tb + ad);
        CXG_SDWA_ADDR(ad, x[k].z, 0); e[8] = *reinterpret_cast<const uint32_t*>(Tb + ad);
        CXG_SDWA_ADDR(ad, x[k].z, 1); e[9] = *reinterpret_cast<const uint32_t*>(Tb + ad);
        CXG_SDWA_ADDR(ad, x[k].z, 2); e[10] = *reinterpret_cast<const uint32_t*>(Tb + ad);
        CXG_SDWA_ADDR(ad, x[k].z, 3); e[11] = *reinterpret_cast<const uint32_t*>(Tb + ad);
        CXG_SDWA_ADDR(ad, x[k].w, 0); e[12] = *reinterpret_cast<const uint32_t*>(Tb + ad);
        CXG_SDWA_ADDR(ad, x[k].w, 1); e[13] = *reinterpret_cast<const uint32_t*>(Tb + ad);
        CXG_SDWA_ADDR(ad, x[k].w, 2); e[14] = *reinterpret_cast<const uint32_t*>(Tb + ad);
        CXG_SDWA_ADDR(ad, x[k].w, 3); e[15] = *reinterpret_cast<const uint32_t*>(Tb + ad);
        uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        constexpr bool SYNC_ALL = false;                              // (A/B: flags of every vector, as in round 2)
        const bool SYNC = SYNC_ALL || k == 0 || k == 3;               // compile-time per unrolled k
        CXG_TEDDY_STEP(t0, s0, e[0], e[1], e[2], 0);   CXG_TEDDY_STEP(t0, s0, e[1], e[2], e[3], 1);
        CXG_TEDDY_STEP(t0, s0, e[2], e[3], e[4], 2);   CXG_TEDDY_STEP(t0, s0, e[3], e[4], e[5], 3);
        CXG_TEDDY_STEP(t1, s1, e[4], e[5], e[6], 0);   CXG_TEDDY_STEP(t1, s1, e[5], e[6], e[7], 1);
        CXG_TEDDY_STEP(t1, s1, e[6], e[7], e[8], 2);   CXG_TEDDY_STEP(t1, s1, e[7], e[8], e[9], 3);
        CXG_TEDDY_STEP(t2, s2, e[8], e[9], e[10], 0);  CXG_TEDDY_STEP(t2, s2, e[9], e[10], e[11], 1);
        CXG_TEDDY_STEP(t2, s2, e[10], e[11], e[12], 2); CXG_TEDDY_STEP(t2, s2, e[11], e[12], e[13], 3);
        CXG_TEDDY_STEP(t3, s3, e[12], e[13], e[14], 0); CXG_TEDDY_STEP(t3, s3, e[13], e[14], e[15], 1);
        CXG_TEDDY_STEP(t3, s3, e[14], e[15], e[16], 2); CXG_TEDDY_STEP(t3, s3, e[15], e[16], e[17], 3);
        pc[lane + 64 * k] = static_cast<uint16_t>(gather16_nz(t0, t1, t2, t3));
        ps[lane + 64 * k] = SYNC ? static_cast<uint16_t>(gather16_01(s0, s1, s2, s3)) : static_cast<uint16_t>(0);
      }
      const uint32_t xprev_cur = xprev;
      issue_loads(j + 1);                                           // x[] is free from here on
      wave_lds_sync();
      const uint64_t C = s_cw[wave][0][lane];
      uint64_t Z = s_cw[wave][1][lane];
      if (stage != kWin) {                                          // short last window: bytes past the data read as 0
        const int32_t nv = stage - 64 * lane;
        Z &= nv <= 0 ? 0ull : (nv >= 64 ? ~0ull : ((1ull << nv) - 1ull));
      }

      // ---- O: ownership bounds
      int32_t zA = -1, zB = kTFar;
      if (tile_lo > 0) {
        const uint32_t pb = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(xprev_cur))) >> 24;
        if (!(s_T[pb] & 0x1000000u)) {                                // the segment at the tile's first byte began earlier
          const unsigned long long bz = __ballot(Z != 0ull);
          if (bz) {
            const int L = __builtin_ctzll(bz);
            zA = 64 * L + static_cast<int32_t>(__builtin_ctzll(readlane64(Z, L)));
            if (L >= 16) fallback |= 1;                               // no synchronising byte in the first KiB: words 16..47 carry no flags (SYNC above)
          }
          else zA = kTFar;
        }
      }
      {
        const uint64_t Zb = Z & word_range(lane, kWaveTile - 1, kWin - 1);
        const unsigned long long bzb = __ballot(Zb != 0ull);
        if (bzb) { const int L = __builtin_ctzll(bzb); zB = 64 * L + static_cast<int32_t>(__builtin_ctzll(readlane64(Zb, L))); }
        else if (stage != rend) { zB = -2; fallback |= 1; }
      }
      const uint64_t Co = C & word_range(lane, zA + 1, zB);

      // ---- V: list the owned candidates, verify 64 at a time
      const uint32_t nc_lane = static_cast<uint32_t>(__popcll(Co));
      const uint32_t incl = wave_inclusive_sum(nc_lane);
      uint32_t ncand = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
      if (ncand > static_cast<uint32_t>(kTCands)) { fallback |= 8; ncand = kTCands; }
      if (ncand) {
        uint32_t idx = incl - nc_lane;
        uint64_t cb = Co;
        while (cb) {
          const int bit = __builtin_ctzll(cb);
          cb &= cb - 1;
          if (idx < static_cast<uint32_t>(kTCands)) s_cpos[wave][idx] = static_cast<uint16_t>(64 * lane + bit);
          idx++;
        }
        wave_lds_sync();
        int32_t cur_end = -1;                                       // wave-uniform: end of the last emitted match
        for (uint32_t r0 = 0; r0 < ncand; r0 += 64) {
          int32_t c = 0, mlen = 0;
          if (r0 + static_cast<uint32_t>(lane) < ncand) {
            c = s_cpos[wave][r0 + lane];
            const uint8_t* wb = s_bytes[wave];
            // the 12 bytes at the candidate as three dwords (aligned LDS reads + v_alignbit): a literal is compared in three
            // masked XORs instead of a byte loop of up to its length — the byte loop was ~200 VALU of a tile's 783
            const uint32_t* wa = reinterpret_cast<const uint32_t*>(wb + (c & ~3));
            const uint32_t sh = (static_cast<uint32_t>(c) & 3u) * 8u;
            const uint32_t d0 = wa[0], d1 = wa[1], d2 = wa[2], d3 = wa[3];
            const uint32_t w0 = __builtin_amdgcn_alignbit(d1, d0, sh), w1 = __builtin_amdgcn_alignbit(d2, d1, sh), w2 = __builtin_amdgcn_alignbit(d3, d2, sh);
            uint32_t mask = (s_T[w0 & 0xFFu] & 0xFFu) & ((s_T[(w0 >> 8) & 0xFFu] >> 8) & 0xFFu) & ((s_T[(w0 >> 16) & 0xFFu] >> 16) & 0xFFu);
            while (mask && !mlen) {                                 // buckets low to high, ids ascending (verifyBucket)
              const uint32_t bk = static_cast<uint32_t>(__builtin_ctz(mask));
              mask &= mask - 1;
              for (uint32_t k = s_boff[bk]; k < s_boff[bk + 1] && !mlen; k++) {
                const uint32_t id = t_order[k];
                const int32_t len = t_lens[id];
                if (c + len > rend) continue;
                if (id < 32u) {
                  const uint32_t diff = ((w0 ^ s_lit[id][0]) & s_lit[id][3]) | ((w1 ^ s_lit[id][1]) & s_lit[id][4]) | ((w2 ^ s_lit[id][2]) & s_lit[id][5]);
                  if (diff != 0u) continue;
                  if (len <= 12) { mlen = len; continue; }
                }
                const uint8_t* lit = t_bytes + t_off[id];
                int32_t q = id < 32u ? 12 : 0;                     // Fat Teddy ids >= 32 and the tail of long literals: bytes
                if (fold) { while (q < len && (wb[c + q] == lit[q] || (lit[q] >= 'a' && lit[q] <= 'z' && (wb[c + q] | 0x20u) == lit[q]))) q++; }
                else while (q < len && wb[c + q] == lit[q]) q++;
                if (q == len) mlen = len;
              }
            }
            if (!VERIFY && mlen && (look_pre | look_post) != 0u) {   // the assertions around the occurrence (checkLook, nfa/pikevm.go:1646-1674)
              const int pb = c > 0 ? static_cast<int>(wb[c - 1]) : (tile_lo > 0 ? static_cast<int>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(xprev_cur))) >> 24) : -1);
              const int nb = c + mlen < rend ? (c + mlen < kWin ? static_cast<int>(wb[c + mlen]) : -2) : -1;
              if (nb == -2) { edge_hit = 1; mlen = 0; }                 // the byte behind the occurrence lies behind the window: hand the scan over
              else if (!teddy_look_holds(look_pre, pb, static_cast<int>(wb[c])) || !teddy_look_holds(look_post, static_cast<int>(wb[c + mlen - 1]), nb)) mlen = 0;
            }
            if (VERIFY && mlen) {                                   // literal found: the anchored DFA gives the match end
              uint32_t q = dfa_start;
              int32_t last = -1, i = c;
              const int32_t lim = rend < kWin ? rend : kWin;
              for (;; i++) {
                if (q >= dfa_fa) last = i;
                if (i >= lim) break;
                q = s_dfa[q * 256u + wb[i]];
                if (q == 0u) break;
              }
              if (q != 0u && i >= kWin && rend > kWin) edge_hit = 1;    // alive at the window edge: the match may go on
              mlen = last > c ? last - c : 0;
              if (a.max_len != 0 && static_cast<uint32_t>(mlen) > a.max_len) long_hit = 1;
            }
          }
          // ---- D: FindAll order inside the round (candidates ascend with the lane)
          const int32_t e = mlen ? c + mlen : 0;
          int32_t pmax = e;                                         // inclusive prefix max of the ends
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            const int32_t o = __shfl_up(pmax, d, 64);
            if (lane >= d && o > pmax) pmax = o;
          }
          int32_t before = static_cast<int32_t>(dpp_from_lower(static_cast<uint32_t>(pmax)));
          if (lane == 0) before = 0;
          if (cur_end > before) before = cur_end;
          uint32_t emit = mlen ? 1u : 0u;
          if (__ballot(mlen && c < before) != 0ull) {               // some verified candidate lies inside an earlier match
            s_ce[wave][lane] = static_cast<uint16_t>(e);
            wave_lds_sync();
            if (lane == 0) {
              int32_t ce = cur_end;
              for (uint32_t k = 0; k < 64; k++) {                   // all 64: lanes past ncand hold e = 0 and must read em = 0
                const int32_t ek = s_ce[wave][k];
                uint8_t em = 0;
                if (ek && static_cast<int32_t>(s_cpos[wave][r0 + k]) >= ce) { em = 1; ce = ek; }
                s_em[wave][k] = em;
              }
            }
            wave_lds_sync();
            emit = s_em[wave][lane];
          }
          const unsigned long long em_mask = __ballot(emit != 0);
          if (em_mask) {
            const int last = 63 - __builtin_clzll(em_mask);
            cur_end = __builtin_amdgcn_readlane(e, last);
            const uint32_t n_em = static_cast<uint32_t>(__popcll(em_mask));
            if (emit) {
              const uint32_t r = nrows_w + emitted_here + static_cast<uint32_t>(__popcll(em_mask & ((1ull << lane) - 1ull)));
              if (r < static_cast<uint32_t>(kTRows)) { s_rs[wave][r] = static_cast<uint16_t>(c); s_re[wave][r] = static_cast<uint16_t>(e); }
            }
            emitted_here += n_em;
          }
        }
      }
    }
    if (lane == 0) s_cnt[wave][j] = emitted_here;
    nrows_w += emitted_here;
  }
  if (nrows_w > static_cast<uint32_t>(kTRows)) fallback |= 16;
  if (__ballot(edge_hit != 0) != 0ull) fallback |= 32;
  if (fallback != 0 && lane == 0) raise_err(a.err, 8u | (fallback << 8));
  if (VERIFY && __ballot(long_hit != 0) != 0ull && lane == 0) raise_err(a.err, kErrLongMatch);
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
  for (int j = 0; j < tpw; j++) {
    const uint32_t n = s_cnt[wave][j];
    const uint64_t dst = base + s_qbase[j * kWavesPerBlock + wave];
    for (uint32_t i = lane0; i < n; i += 64) {
      const uint32_t r = start + i;
      if (r < static_cast<uint32_t>(kTRows) && dst + i < a.cap) {
        const int64_t tb = origin + static_cast<int64_t>(j * kWavesPerBlock + wave) * kWaveTile;
        longlong2 v; v.x = tb + s_rs[wave][r]; v.y = tb + s_re[wave][r];
        store_pair_nt(a.out + (dst + i) * a.row_width, v.x, v.y);
      }
    }
    start += n;
  }
}

hipError_t launch_scan_teddy_wave(const ScanArgs& a, uint32_t verify_dfa_states, hipStream_t stream) {
  const dim3 grid(static_cast<unsigned>(a.ngroups)), block(kThreads);
  const bool dense = a.tiles_per_wave == static_cast<uint32_t>(kDenseTilesPerWave);
  if (verify_dfa_states) {
    if (dense) hipLaunchKernelGGL((k_scan_teddy_wave<true, true>), grid, block, verify_dfa_states * 256u, stream, a);
    else hipLaunchKernelGGL((k_scan_teddy_wave<true, false>), grid, block, verify_dfa_states * 256u, stream, a);
  } else {
    if (dense) hipLaunchKernelGGL((k_scan_teddy_wave<false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_scan_teddy_wave<false, false>), grid, block, 0, stream, a);
  }
  return hipGetLastError();
}

}  // namespace cxgdev
