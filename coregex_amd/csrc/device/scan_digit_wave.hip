// scan_digit_wave.hip — UseDigitPrefilter FindAll, fifth kernel generation: the chain prefilter of
// scan_digit_chain.hip with the WAVE as the unit of work, so that nothing waits on a workgroup barrier.
//
// One wave64 owns a wave-tile of 60 x 64 B = 3840 B and reads 256 B of halo behind it: 4096 B = 64 bitmap
// words, one per lane.  Everything between "load" and "rows in LDS" is wave-local:
//   A  4 coalesced 16-byte loads per lane (1 KiB per instruction); per chain class a 16-bit mask of each
//      vector (SWAR), written bit-reversed into the wave's private LDS scratch; each lane reads back the
//      64-bit word of its own chunk (transpose through LDS, no barrier: same wave).
//   B  chain right-to-left on registers: neighbour words by DPP/shuffle, the multi-word addition of a run
//      step resolved in ONE step with two ballots: receivers = (P + (G << 1)) ^ P, G/P = per-lane
//      generate/propagate flags (a 64-lane carry-lookahead adder on the scalar unit).
//   C  survivors (digit-run starts passing the chain) are compacted with ballot/popcount prefix sums and
//      verified lane-parallel by the anchored DFA walk (bytes from L2) — exactness by construction; each
//      finds its segment start for ownership.
//   D  FindAll order among owned successes (parallel when no two overlap, else one lane walks them);
//      rows appended to the wave's row buffer.
// A workgroup (4 waves) takes one ticket per group of 32 wave-tiles (120 KiB): tables are staged once,
// and after a single barrier the group's rows are ordered, looked back and written (coalesced 16 B).
// Fallback flag (host reruns the scan with the flat kernel): no synchronising byte in a halo, > 64
// survivors in a wave-tile, row buffer overflow.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_common.hpp"
#include "scan_dfa.h"
#include "walk.hpp"

namespace cxgdev {

namespace {

constexpr int kRowStride = 260;
constexpr int kWRows = 512;                       // rows buffered per wave per group

struct GMem {            // bytes of the wave-tile window come from the wave's LDS copy, anything outside from L2/HBM
  const uint8_t* g;
  const uint8_t* lds;
  int32_t lim;
  int32_t flag_at = 0x7FFFFFFF;   // serial-walk cut (scan_dfa.h walk_limit)
  mutable uint32_t over = 0;
  __device__ __forceinline__ uint32_t byte(int32_t r) const {
    if (static_cast<uint32_t>(r) < static_cast<uint32_t>(lim)) return lds[r];
    over |= static_cast<uint32_t>(r >= flag_at);
    return g[r];
  }
  __device__ __forceinline__ uint64_t digits(int32_t) const { return 0; }
  __device__ __forceinline__ int32_t bitmap_limit() const { return 0; }
};

// 0x80-per-byte flags -> the four flags in bits 28..31 (byte 0 lowest); two shift-or steps, no multiply
// (v_mul_lo_u32 is quarter rate).  Bits below 28 are junk.
__device__ __forceinline__ uint32_t gather_top(uint32_t m80) {
  const uint32_t t = m80 | (m80 << 7);
  return t | (t << 14);
}
// 0x80 flag in every byte of x that is NOT in the class (inverted once per 16 bytes by the caller)
__device__ __forceinline__ uint32_t notcls4(uint32_t x, uint32_t kind, uint32_t lo, uint32_t hi) {
  if (kind == kClsDigit) {
    const uint32_t t = x ^ 0x30303030u;
    return (((t & 0x7F7F7F7Fu) + 0x76767676u) | t) & 0x80808080u;
  }
  if (kind == kClsByte) {
    const uint32_t v = x ^ (lo * 0x01010101u);
    return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;
  }
  const uint32_t ge = ((x | 0x80808080u) - lo * 0x01010101u) & 0x80808080u;
  const uint32_t gt = ((x & 0x7F7F7F7Fu) + (0x7Fu - hi) * 0x01010101u) & 0x80808080u;
  return ~(ge & ~gt & ~x) & 0x80808080u;
}
__device__ __forceinline__ uint32_t cls16(const uint4& x, uint32_t kind, uint32_t lo, uint32_t hi) {
  const uint32_t n = (gather_top(notcls4(x.x, kind, lo, hi)) >> 28) | ((gather_top(notcls4(x.y, kind, lo, hi)) >> 24) & 0xF0u) |
                     ((gather_top(notcls4(x.z, kind, lo, hi)) >> 20) & 0xF00u) | ((gather_top(notcls4(x.w, kind, lo, hi)) >> 16) & 0xF000u);
  return n ^ 0xFFFFu;
}
// Neighbour-lane moves as DPP wavefront shifts (one VALU op per dword, no LDS crossbar round trip).
__device__ __forceinline__ uint32_t dpp_from_lower(uint32_t v) {  // lane i <- lane i-1 (lane 0 keeps its own)
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x138 /*wave_shr:1*/, 0xF, 0xF, false));
}
__device__ __forceinline__ uint32_t dpp_from_upper(uint32_t v) {  // lane i <- lane i+1 (lane 63 keeps its own)
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x130 /*wave_shl:1*/, 0xF, 0xF, false));
}
__device__ __forceinline__ uint64_t shfl_up64(uint64_t v) {      // value of lane-1 (lane 0 gets its own)
  return (static_cast<uint64_t>(dpp_from_lower(static_cast<uint32_t>(v >> 32))) << 32) | dpp_from_lower(static_cast<uint32_t>(v));
}
__device__ __forceinline__ uint64_t shfl_down64(uint64_t v) {    // value of lane+1 (lane 63 gets its own)
  return (static_cast<uint64_t>(dpp_from_upper(static_cast<uint32_t>(v >> 32))) << 32) | dpp_from_upper(static_cast<uint32_t>(v));
}
__device__ __forceinline__ void wave_lds_sync() {                // same-wave LDS hand-off: drain, no barrier
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);                            // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace

__global__ __launch_bounds__(kThreads, 4) void k_scan_digit_wave(ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];   // DFA table, info, sflags, ChainAux
  __shared__ __attribute__((aligned(16))) uint64_t s_cls[kWavesPerBlock][kChainMaxCls][64];
  __shared__ __attribute__((aligned(16))) uint8_t s_bytes[kWavesPerBlock][kWaveTile + kWaveHalo];
  __shared__ __attribute__((aligned(16))) uint64_t s_u[kWavesPerBlock][64];   // union of the class bitmaps (complete chains)
  __shared__ uint32_t s_rowpos[kWavesPerBlock][kWRows];
  __shared__ uint16_t s_rowlen[kWavesPerBlock][kWRows];
  __shared__ uint16_t s_spos[kWavesPerBlock][64];
  __shared__ uint16_t s_slen[kWavesPerBlock][64];
  __shared__ uint8_t s_sown[kWavesPerBlock][64];
  __shared__ uint32_t s_cnt[kWavesPerBlock][kTilesPerWave];
  __shared__ uint32_t s_qbase[kWavesPerBlock * kTilesPerWave + 1];
  __shared__ uint64_t s_group;
  __shared__ uint64_t s_base;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_group = claim_tile(a.ticket, a.ngroups);
  const BlobHeader* h = reinterpret_cast<const BlobHeader*>(a.blob);
  const uint32_t fwd_states = h->fwd_states;
  uint8_t* s_fwd = s_dyn;
  uint8_t* s_info = s_fwd + fwd_states * kRowStride;
  uint8_t* s_sfl = s_info + 256;
  ChainAux* s_chain = reinterpret_cast<ChainAux*>(s_sfl + 256);
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.blob + h->fwd_off);
    for (uint32_t i = tid; i < fwd_states * 64u; i += kThreads)
      *reinterpret_cast<uint32_t*>(s_fwd + (i >> 6) * kRowStride + (i & 63u) * 4u) = src[i];
    if (tid < 64) reinterpret_cast<uint32_t*>(s_info)[tid] = reinterpret_cast<const uint32_t*>(a.blob + h->info_off)[tid];
    else if (tid < 128) reinterpret_cast<uint32_t*>(s_sfl)[tid - 64] = reinterpret_cast<const uint32_t*>(a.blob + h->aux_off)[tid - 64];
    else if (tid < 128 + sizeof(ChainAux) / 4)
      reinterpret_cast<uint32_t*>(s_chain)[tid - 128] = reinterpret_cast<const uint32_t*>(a.blob + h->aux_off + 256)[tid - 128];
  }
  __syncthreads();
  const uint64_t group = s_group;
  if (group >= a.ngroups) return;
  const uint32_t ncls = s_chain->ncls, nops = s_chain->nops;
  DfaView fv{s_fwd, kRowStride, h->fwd_start, h->fwd_first_accept};
  uint32_t nrows_w = 0;                                            // wave-uniform
  uint32_t fallback = 0;
  const bool complete = (h->flags & kFlagChainComplete) != 0;

  // Software pipeline: the 4 loads of wave-tile j+1 are issued as soon as tile j's class masks have been
  // taken from the registers, so their HBM latency overlaps phases B-D of tile j.
  uint4 x[4];
  auto issue_loads = [&](int jj) {
    const uint64_t wtn = group * (kWavesPerBlock * kTilesPerWave) + static_cast<uint64_t>(jj) * kWavesPerBlock + wave;
    const uint64_t lo = wtn * static_cast<uint64_t>(kWaveTile);
    int nf = 0;
    if (jj < kTilesPerWave && lo < a.len) {
      const uint64_t rem = a.len - lo;
      const int32_t st = rem > static_cast<uint64_t>(kWaveTile + kWaveHalo) ? kWaveTile + kWaveHalo : static_cast<int32_t>(rem);
      nf = st >> 4;
    }
    const uint8_t* gp = a.hay + lo;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int v = lane + 64 * k;
      x[k] = (v < nf) ? *reinterpret_cast<const uint4*>(gp + (v << 4)) : make_uint4(0, 0, 0, 0);
    }
  };
  issue_loads(0);

  for (int j = 0; j < kTilesPerWave; j++) {
    const uint64_t wt = group * (kWavesPerBlock * kTilesPerWave) + static_cast<uint64_t>(j) * kWavesPerBlock + wave;
    const uint64_t tile_lo = wt * static_cast<uint64_t>(kWaveTile);
    uint32_t emitted_here = 0;
    if (tile_lo < a.len) {
      const uint64_t remaining = a.len - tile_lo;
      const WalkLimit wl = walk_limit(remaining, kWaveTile + kWaveHalo);   // serial-walk budget, scan_dfa.h
      const int32_t rend = wl.rend;
      const int32_t stage = rend < kWaveTile + kWaveHalo ? rend : kWaveTile + kWaveHalo;
      const uint8_t* g = a.hay + tile_lo;

      // ---- A: class masks of the vectors loaded one iteration ago, transpose through the wave's LDS scratch
      const int nfull = stage >> 4;
      if (!complete) {
#pragma unroll
        for (int k = 0; k < 4; k++) {                               // keep the window's bytes for the verify walks
          const int v = lane + 64 * k;
          if (v < nfull) *reinterpret_cast<uint4*>(&s_bytes[wave][v << 4]) = x[k];
        }
      }
      uint64_t C0 = 0, C1 = 0, C2 = 0, C3 = 0;                      // class words (wave-uniform selects, no indexed registers)
      for (uint32_t c = 0; c < ncls; c++) {
        const uint32_t kind = s_chain->cls_kind[c], lo = s_chain->cls_lo[c], hi = s_chain->cls_hi[c];
        uint16_t* pieces = reinterpret_cast<uint16_t*>(s_cls[wave][c]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int v = lane + 64 * k;
          uint32_t mask = 0;
          if (v < nfull) mask = (a.dbg & 0x40000u) ? (x[k].x & 1u) : cls16(x[k], kind, lo, hi);
          else if (v == nfull) {
            const int base = v << 4;
            for (int b = 0; base + b < stage; b++) mask |= (chain_class_has(*s_chain, static_cast<int>(c), g[base + b]) ? 1u : 0u) << b;
          }
          pieces[255 - v] = static_cast<uint16_t>(__brev(mask) >> 16);     // reversed bitmap: bit i <-> byte 4095-i
        }
      }
      // halo check: is there a synchronising byte in [kWaveTile-1, stage)?  (vectors 239..255 live in load slot 3)
      uint32_t sync_here = 0;
      if (!complete) {
        const int v = lane + 192;
        if (v >= 239 && (v << 4) < stage) {
          const uint32_t w[4] = {x[3].x, x[3].y, x[3].z, x[3].w};
          const int first = (v == 239) ? 15 : 0;
#pragma unroll
          for (int b = 0; b < 16; b++)
            if (b >= first && (v << 4) + b < stage) sync_here |= s_info[(w[b >> 2] >> ((b & 3) * 8)) & 0xFFu] & kInfoSync;
        }
      }
      issue_loads(j + 1);                                           // x[] is free from here on
      if (!(a.dbg & 0x400000u)) wave_lds_sync();
      C0 = s_cls[wave][0][lane];                                    // lane l holds reversed word l (chunk 63-l)
      if (ncls > 1) C1 = s_cls[wave][1][lane];
      if (ncls > 2) C2 = s_cls[wave][2][lane];
      if (ncls > 3) C3 = s_cls[wave][3][lane];
      if (complete) {
        // sync bytes = complement of the class union; the halo is reversed words 0..3 plus bit 0 of word 4
        const uint64_t U = C0 | C1 | C2 | C3;
        s_u[wave][lane] = U;
        const int32_t lo_idx = (kWaveTile + kWaveHalo) - stage;     // reversed indices below this are past the data
        uint64_t valid = 0;
        if (lane < 4) { const int32_t sh = lo_idx - 64 * lane; valid = sh <= 0 ? ~0ull : (sh >= 64 ? 0ull : (~0ull << sh)); }
        else if (lane == 4) valid = (lo_idx <= 256) ? 1ull : 0ull;
        sync_here = (~U & valid) != 0ull ? 1u : 0u;
      }
      const bool halo_ok = (stage == rend) || (__ballot(sync_here != 0) != 0ull);

      // ---- B: chain, right to left, in registers
      uint64_t G = ~0ull;
      const bool at_eoi_edge = (stage == rend) && (stage == kWaveTile + kWaveHalo);   // byte 4095 is the last of the input
      for (int k = (a.dbg & 0x10000u) ? -1 : static_cast<int>(nops) - 1; k >= 0; k--) {
        const uint32_t ci = s_chain->op_cls[k];
        const uint64_t Ck = ci == 0 ? C0 : ci == 1 ? C1 : ci == 2 ? C2 : C3;
        const uint64_t inject = (at_eoi_edge && k == static_cast<int>(nops) - 1) ? 1ull : 0ull;   // G_{n+1} holds at end of input
        if (s_chain->op_kind[k] == kChainByte) {
          uint64_t low = shfl_up64(G) >> 63;                        // DPP outside any lane-dependent branch: a
          if (lane == 0) low = inject;                              // disabled source lane would not be read
          G = Ck & ((G << 1) | low);
        } else {
          uint64_t cup = shfl_down64(Ck);
          if (lane == 63) cup = 0;
          const uint64_t K = G & ~Ck & ((Ck >> 1) | (cup << 63));
          uint64_t klow = shfl_up64(K) >> 63;
          if (lane == 0) klow = inject & Ck;                        // virtual marker just beyond the last byte
          const uint64_t M = (K << 1) | klow;
          const uint64_t s1 = Ck + M;
          const unsigned long long GG = __ballot(s1 < M);
          const unsigned long long PP = __ballot(s1 == ~0ull);
          const unsigned long long recv = (PP + (GG << 1)) ^ PP;    // lanes that receive a carry
          G = Ck & ~(s1 + ((recv >> lane) & 1ull));
        }
      }
      // ---- survivors: digit-run starts passing the chain
      const uint64_t D = C0;
      uint64_t dup = shfl_down64(D);
      if (lane == 63) dup = (tile_lo > 0 && is_digit(g[-1])) ? 1ull : 0ull;
      uint64_t surv = D & ~((D >> 1) | (dup << 63)) & G;
      if (a.dbg & 0x30000u) surv = (a.dbg & 0x80000u) ? (surv & 1ull) : 0ull;   // timing experiments: no / few survivors
      if (a.dbg & 0x200000u) { if (surv == 0x123456789ull) emitted_here = 1; }
      else {
      // ---- C: compaction in ascending position (= descending lane, descending bit)
      const uint32_t mine = static_cast<uint32_t>(__popcll(surv));
      uint32_t suffix = mine;                                       // inclusive suffix sum over lanes >= lane
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_down(suffix, d, 64);
        if (lane + d < 64) suffix += o;
      }
      uint32_t nsurv = __shfl(suffix, 0, 64);
      if (nsurv > 64u || !halo_ok) { fallback = 1; nsurv = nsurv > 64u ? 64u : nsurv; }
      {
        uint32_t idx = suffix - mine;                               // survivors in higher lanes come first
        while (surv) {
          const int bit = 63 - __builtin_clzll(surv);
          surv &= ~(1ull << bit);
          if (idx < 64u) s_spos[wave][idx] = static_cast<uint16_t>(4095 - (64 * lane + bit));
          idx++;
        }
      }
      wave_lds_sync();
      // ---- verify + ownership, one survivor per lane
      int32_t c = 0, len = 0;
      uint32_t owned = 0;
      if (static_cast<uint32_t>(lane) < nsurv) {
        c = s_spos[wave][lane];
        GMem m{g, s_bytes[wave], complete ? 0 : (nfull << 4)};
        m.flag_at = wl.flag_at;
        int32_t e = -1;
        bool walked = false;
        if (complete) {
          const uint64_t* cw[kChainMaxCls] = {s_cls[wave][0], s_cls[wave][1], s_cls[wave][2], s_cls[wave][3]};
          const int32_t ie = chain_walk_end(*s_chain, cw, 4095 - c);
          if (ie >= 0 && (4095 - ie < stage || (4095 - ie == stage && stage == rend))) { e = 4095 - ie; walked = true; }   // an end AT a cut window edge is unknown
        }
        if (!walked) e = verify_jump(m, fv, s_sfl, c, rend);
        if (m.over) raise_err(a.err, kErrSerialLimit);
        len = e < 0 ? 0 : e - c;
        if (len > 0xFFFF) { fallback = 1; len = 0; }
        if (len) {
          int32_t seg;
          if (complete) {
            const int32_t jz = rev_scan_up_zero(s_u[wave], 4096 - c, 4096);   // nearest sync byte before c
            if (jz < 4096) seg = 4096 - jz;
            else seg = (tile_lo == 0 || (s_info[g[-1]] & kInfoSync)) ? 0 : -1;
          } else {
            int32_t p = c - 1;
            while (p >= 0 && !(s_info[m.byte(p)] & kInfoSync)) p--;
            if (p >= 0) seg = p + 1;
            else seg = (tile_lo == 0 || (s_info[g[-1]] & kInfoSync)) ? 0 : -1;
          }
          owned = (seg >= 0 && seg < kWaveTile) ? 1u : 0u;
        }
      }
      // ---- D: FindAll order.  Overlaps between successes are rare: resolve in parallel unless one exists.
      int32_t pmax = owned ? c + len : 0;                          // inclusive prefix max of owned match ends
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int32_t o = __shfl_up(pmax, d, 64);
        if (lane >= d && o > pmax) pmax = o;
      }
      const int32_t prev_end = static_cast<int32_t>(dpp_from_lower(static_cast<uint32_t>(pmax)));
      const bool overlap = owned && lane > 0 && c < prev_end;
      uint32_t emit = owned;
      if (__ballot(overlap) != 0ull) {
        s_slen[wave][lane] = static_cast<uint16_t>(len);
        s_sown[wave][lane] = static_cast<uint8_t>(owned);
        wave_lds_sync();
        if (lane == 0) {
          int32_t cur_end = -1;
          for (uint32_t k = 0; k < nsurv; k++) {
            uint8_t em = 0;
            if (s_sown[wave][k]) {
              const int32_t ck = s_spos[wave][k];
              if (ck >= cur_end) { em = 1; cur_end = ck + s_slen[wave][k]; }
            }
            s_sown[wave][k] = em;
          }
        }
        wave_lds_sync();
        emit = s_sown[wave][lane];
      }
      const unsigned long long em_mask = __ballot(emit != 0);
      emitted_here = static_cast<uint32_t>(__popcll(em_mask));
      if (emit) {
        const uint32_t r = nrows_w + static_cast<uint32_t>(__popcll(em_mask & ((1ull << lane) - 1ull)));
        if (r < static_cast<uint32_t>(kWRows)) {
          s_rowpos[wave][r] = static_cast<uint32_t>(j * kWavesPerBlock + wave) * kWaveTile + static_cast<uint32_t>(c);
          s_rowlen[wave][r] = static_cast<uint16_t>(len);
        }
      }
      }
    }
    if (lane == 0) s_cnt[wave][j] = emitted_here;
    nrows_w += emitted_here;
  }
  if (a.dbg & 0x100000u) return;
  if (nrows_w > static_cast<uint32_t>(kWRows)) fallback = 1;
  if (__ballot(fallback != 0) != 0ull && lane == 0) atomicOr(a.err, 8u);
  __syncthreads();

  // ---- order the group's rows: wave-tile q = j*4 + wave; exclusive prefix over q
  if (tid < 64) {
    const int q = tid;
    uint32_t v = (q < kWavesPerBlock * kTilesPerWave) ? s_cnt[q % kWavesPerBlock][q / kWavesPerBlock] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (q < kWavesPerBlock * kTilesPerWave) s_qbase[q] = incl - v;
    if (q == kWavesPerBlock * kTilesPerWave - 1) s_qbase[kWavesPerBlock * kTilesPerWave] = incl;
  }
  __syncthreads();
  uint32_t total = s_qbase[kWavesPerBlock * kTilesPerWave];
  tile_lookback(a.status, a.total, a.err, group, a.ngroups, total, &s_base);
  if (a.out == nullptr) return;
  const uint64_t base = s_base;
  const int64_t origin = a.base + static_cast<int64_t>(group * static_cast<uint64_t>(kWaveTile) * kWavesPerBlock * kTilesPerWave);
  uint32_t start = 0;
  for (int j = 0; j < kTilesPerWave; j++) {
    const uint32_t n = s_cnt[wave][j];
    const uint64_t dst = base + s_qbase[j * kWavesPerBlock + wave];
    for (uint32_t i = lane; i < n; i += 64) {
      const uint32_t r = start + i;
      if (r < static_cast<uint32_t>(kWRows) && dst + i < a.cap) {
        longlong2 v; v.x = origin + s_rowpos[wave][r]; v.y = v.x + s_rowlen[wave][r];
        *reinterpret_cast<longlong2*>(a.out + (dst + i) * 2) = v;
      }
    }
    start += n;
  }
}

hipError_t launch_scan_digit_wave(const ScanArgs& a, uint32_t fwd_states, hipStream_t stream) {
  const size_t dyn = static_cast<size_t>(fwd_states) * kRowStride + 512 + sizeof(ChainAux) + 16;
  hipLaunchKernelGGL(k_scan_digit_wave, dim3(static_cast<unsigned>(a.ngroups)), dim3(kThreads), dyn, stream, a);
  return hipGetLastError();
}

}  // namespace cxgdev
