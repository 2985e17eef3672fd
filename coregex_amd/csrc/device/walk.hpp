// Per-lane FindAll walks — the sequential algorithms of the reference, executed by one GPU lane
// over the sync-delimited segments that START inside its byte chunk [c0, c1).
//
//   lane_digit  = meta.findAllIndicesLoop (meta/findall.go:176-283) over
//                 findIndicesDigitPrefilterAtWithState (meta/find_indices.go:1050-1088):
//                 digit scan (simd/memchr_digit_amd64.s:26) -> anchored table walk
//                 (dfa/lazy/lazy.go:219-324) -> digit-run skip (:1079-1084).
//   lane_bidir  = the useDFADirect loop (meta/findall.go:216-239): unanchored forward walk
//                 (dfa/lazy/lazy.go:1102-1315) then SearchReverse (:1769-1920).
//
// Why a lane may start in the middle of the haystack: a byte outside the pattern's alphabet
// ("sync byte") kills every live DFA state and cannot be part of a match, so the reference's
// sequential state right after it is the same as at position 0 (pos == that index, nothing pending).
// A lane therefore owns every position p in [c0, c1) with p == 0 or hay[p-1] sync, runs the
// reference loop from there, and stops at the first owned-by-someone-else segment start.  The
// concatenation over lanes equals the single-threaded result for non-nullable patterns.
//
// The DFA tables use *immediate* acceptance (a state accepts when its NFA set holds Match); the
// reference's 1-byte match delay (lazy.go:1360,1371) is the same information one transition later:
// "match-tagged after consuming hay[pos]" == "accepting before consuming hay[pos]".
//
// This header is plain C++ so that tests/emu can compile the very same walks for the host and
// compare them with the oracle lane by lane; the product only ever instantiates them in HIP kernels.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CXG_HD __host__ __device__ __forceinline__
#else
#define CXG_HD inline
#endif

namespace cxgdev {

constexpr uint32_t kBlobMagic = 0x43584731u;  // "CXG1"
enum BlobKind : uint32_t { kKindDigit = 1, kKindBidir = 2, kKindCharClass = 3, kKindTeddy = 4,
                           kKindFsmOnly = 5 };   // header + info table only: the program runs on the transducer kernel alone (UseNFA)
constexpr uint32_t kInfoSync = 1u;       // byte is outside the pattern alphabet
constexpr uint32_t kInfoMember = 2u;     // char-class membership (kKindCharClass)
constexpr uint32_t kInfoStartIdle = 4u;  // fwd.start --byte--> fwd.start (skippable while idle)
constexpr uint32_t kFlagRunSkip = 1u;
constexpr uint32_t kFlagFastDigit = 2u;   // run-skip safe and tail closed: candidate-list kernel allowed; aux = sflags[256]

struct BlobHeader {             // device image of a program; all offsets in bytes from the blob start
  uint32_t magic, kind, flags, ngroups;
  uint32_t fwd_states, fwd_start, fwd_first_accept, fwd_off;
  uint32_t rev_states, rev_start, rev_first_accept, rev_off;
  uint32_t info_off, total_bytes, aux_off, aux_len;
};

struct DfaView {
  const uint8_t* T;   // [states][stride] next-state table; state 0 is dead
  uint32_t stride;    // 256 in the blob, 260 when staged in LDS (bank skew)
  uint32_t start;
  uint32_t first_accept;  // states >= first_accept hold Match
};

CXG_HD bool is_digit(uint32_t b) { return (b - 0x30u) < 10u; }
// state * stride: both < 2^24, so the full-rate 24-bit multiply serves (v_mul_lo_u32 is quarter rate)
CXG_HD uint32_t rowmul(uint32_t q, uint32_t stride) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(q, stride);
#else
  return q * stride;
#endif
}

// 0x80 in every byte lane of x that holds an ASCII digit.  x ^ 0x30.. maps '0'..'9' to 0..9 and
// everything else to >= 10; the carry-free add of 0x76 sets bit 7 exactly for low-7 values >= 10.
CXG_HD uint32_t digit_mask4(uint32_t x) {
  const uint32_t t = x ^ 0x30303030u;
  const uint32_t nd = (((t & 0x7F7F7F7Fu) + 0x76767676u) | t) & 0x80808080u;
  return nd ^ 0x80808080u;
}
CXG_HD uint32_t ctz64(uint64_t v) { return static_cast<uint32_t>(__builtin_ctzll(v)); }
CXG_HD uint32_t ctz32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return static_cast<uint32_t>(__builtin_ctz(v));
#else
  return static_cast<uint32_t>(__builtin_ctz(v));
#endif
}

// Mem concept: uint32_t byte(int32_t r) for r in [-1, rend), r relative to the tile base;
//              uint32_t dword(int32_t r) little-endian 4 bytes at r (r % 4 == 0, r + 4 <= wide_limit(x));
//              int32_t wide_limit(int32_t x): largest bound <= x up to which dword() may be used.
// Sink concept: void emit(int32_t s, int32_t e).

// Finds the first owned segment start in [c0, c1); returns -1 if the lane owns none.
template <class Mem>
CXG_HD int32_t first_owned_start(const Mem& m, const uint8_t* info, int32_t c0, int32_t c1, int32_t rend,
                                 bool chunk_at_origin) {
  int32_t pos = c0;
  if (pos >= rend) return -1;
  if (chunk_at_origin) return pos;
  if (info[m.byte(pos - 1)] & kInfoSync) return pos;
  for (;;) {
    if (pos >= c1 || pos >= rend) return -1;
    uint32_t b = m.byte(pos);
    pos++;
    if (info[b] & kInfoSync) break;
  }
  if (pos >= c1 || pos >= rend) return -1;
  return pos;
}

template <class Mem, class Sink>
CXG_HD void lane_digit(const Mem& m, const DfaView& d, const uint8_t* info, bool skip_safe, int32_t c0, int32_t c1,
                       int32_t rend, bool chunk_at_origin, Sink& sink) {
  int32_t pos = first_owned_start(m, info, c0, c1, rend, chunk_at_origin);
  if (pos < 0) return;
  for (;;) {
    // prefilter: next digit at >= pos (prefilter/digit.go:72).  Inside the lane's own chunk no
    // ownership check is needed, so aligned dwords are tested 4 bytes at a time (the GPU twin of
    // the 32 B/iter AVX2 loop, simd/memchr_digit_amd64.s:26).
    {
      const int32_t wide_end = m.wide_limit(c1 < rend ? c1 : rend);
      while (pos < wide_end) {
        if ((pos & 3) == 0 && pos + 4 <= wide_end) {
          const uint32_t dm = digit_mask4(m.dword(pos));
          if (dm) { pos += static_cast<int32_t>(ctz32(dm) >> 3); break; }
          pos += 4;
          continue;
        }
        if (is_digit(m.byte(pos))) break;
        pos++;
      }
    }
    for (;;) {
      if (pos >= rend) return;
      if (pos >= c1 && (info[m.byte(pos - 1)] & kInfoSync)) return;  // next owner's segment
      if (is_digit(m.byte(pos))) break;
      pos++;
    }
    const int32_t dpos = pos;
    // anchored verify (lazy.go:219-324)
    uint32_t q = d.start;
    int32_t last = -1, i = dpos;
    for (;;) {
      if (q >= d.first_accept) last = i;
      if (i >= rend) break;
      q = d.T[rowmul(q, d.stride) + m.byte(i)];
      if (q == 0) break;
      i++;
    }
    if (last >= 0) {
      sink.emit(dpos, last);
      pos = last > pos ? last : pos + 1;  // findall.go:267-275 (matches are non-empty here)
    } else {
      pos = dpos + 1;                      // find_indices.go:1079-1084
      if (skip_safe)
        while (pos < rend && is_digit(m.byte(pos))) pos++;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Flat form of lane_digit for SIMT execution.  Same algorithm, restructured so that a wave's lanes
// stay convergent: ONE loop whose every iteration is one DFA transition for every live lane; the
// prefilter ("next digit at >= pos", "end of this digit run") is answered in O(1) from a bitmap of
// digit positions that the kernel builds while it stages the tile (the bit-parallel counterpart of
// memchrDigitAVX2).  Ownership is resolved once up front: the lane's candidates are exactly the
// digit positions below `stop`, the first segment start at or after c1 (see lane_digit: a verify can
// never step over a sync byte, so the reference's scan always arrives at `stop` and ends there).
//
// Extra Mem concept: uint64_t digits(int32_t w) -> bit k set iff byte 64*w+k is an ASCII digit, valid
// for bytes below bitmap_limit() (bits at or beyond it read 0); int32_t bitmap_limit().

template <class Mem>
CXG_HD int32_t next_digit(const Mem& m, int32_t pos, int32_t limit) {
  const int32_t blim = m.bitmap_limit();
  while (pos < limit) {
    if (pos < blim) {
      const uint64_t w = m.digits(pos >> 6) >> (pos & 63);
      if (w) return pos + static_cast<int32_t>(ctz64(w));
      pos = (pos | 63) + 1;
    } else {
      if (is_digit(m.byte(pos))) return pos;
      pos++;
    }
  }
  return limit;
}

template <class Mem>
CXG_HD int32_t next_nondigit(const Mem& m, int32_t pos, int32_t rend) {
  const int32_t blim = m.bitmap_limit();
  while (pos < rend && pos < blim) {
    const uint64_t nd = (~m.digits(pos >> 6)) >> (pos & 63);   // logical shift: vacated top bits read "digit"
    if (nd) {
      const int32_t p = pos + static_cast<int32_t>(ctz64(nd));
      if (p < blim) return p;          // blim <= rend
      pos = blim;                      // bits at/after blim are not real data: continue byte-wise
      break;
    }
    pos = (pos | 63) + 1;
  }
  while (pos < rend && is_digit(m.byte(pos))) pos++;
  return pos < rend ? pos : rend;
}

template <class Mem, class Sink>
CXG_HD void lane_digit_flat(const Mem& m, const DfaView& d, const uint8_t* info, bool skip_safe, int32_t c0,
                            int32_t c1, int32_t rend, bool chunk_at_origin, Sink& sink) {
  int32_t pos = first_owned_start(m, info, c0, c1, rend, chunk_at_origin);
  if (pos < 0) return;
  int32_t stop = c1 - 1;
  while (stop < rend && !(info[m.byte(stop)] & kInfoSync)) stop++;
  stop = stop < rend ? stop + 1 : rend;
  bool need = true;
  uint32_t q = 0, cur = 0;
  int32_t last = -1, i = 0, dpos = 0;
  for (;;) {
    if (need) {
      dpos = next_digit(m, pos, stop);
      if (dpos >= stop) break;
      q = d.start; last = -1; i = dpos; need = false;
      cur = m.byte(i);
    }
    if (q >= d.first_accept) last = i;
    bool end = i >= rend;
    if (!end) {
      const int32_t ni = i + 1 < rend ? i + 1 : i;     // speculative fetch of the next byte, off the q chain
      const uint32_t nxt = m.byte(ni);
      q = d.T[rowmul(q, d.stride) + cur];
      cur = nxt;
      if (q == 0) end = true; else i++;
    }
    if (end) {
      if (last >= 0) {
        sink.emit(dpos, last);
        pos = last > dpos ? last : dpos + 1;
      } else {
        pos = dpos + 1;
        if (skip_safe) pos = next_nondigit(m, pos, rend);
      }
      need = true;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// UseTeddy: FindAll over an exact literal alternation (meta/find_indices.go:925-951 ->
// prefilter.Teddy.FindMatch, prefilter/teddy.go:391-444).  The reference finds fingerprint
// candidates with nibble-mask shuffles (teddy_ssse3_amd64.s:273: a position is a candidate when
// lo[0][b0&15] & hi[0][b0>>4] & lo[1][b1&15] & hi[1][b1>>4] != 0 for its first two bytes) and then
// compares the literals of each hit bucket, buckets low to high, ids ascending (verifyBucket
// teddy.go:532-550).  lo&hi of one byte depend only on that byte, so the device tables hold the
// products directly: AB[b] = A | B<<8 with A = lo[0]&hi[0], B = lo[1]&hi[1]; the kernel turns them
// into one candidate bit per haystack byte while staging the tile.
// Restriction (checked on the host): the literal set is prefix-free, so at most one literal matches at
// a position and the bucket-major / position-major order difference of the reference's two code
// paths (teddy.go:400-403,447-458) cannot be observed.

struct TeddyView {
  const uint16_t* ab;      // [256]
  const uint8_t* order;    // literal ids, bucket-major then id
  const uint8_t* lens;     // by id
  const uint8_t* bucket;   // by id
  const uint16_t* off;     // by id, into bytes
  const uint8_t* bytes;
  uint32_t nlits;
};

constexpr uint32_t kTeddyFold = 1u << 16;   // TeddyAux::looks

struct TeddyAux {           // layout of the blob's aux section (all offsets relative to aux start)
  uint32_t nlits, nbuckets, minlen, maxlen;
  uint32_t ab_off, order_off, lens_off, bucket_off, off_off, bytes_off, bytes_len;
  // Literals between two assertions (`\berror\b`, `(?m)^(GET|POST)`, round 4): looks = pre | post << 8, each 0 (none) or nfa.Look + 1
  // (3 StartLine, 4 EndLine, 5 WordBoundary, 6 NoWordBoundary); an occurrence counts when both hold around it (teddy_look_holds)
  // bit 16 (kTeddyFold): case-insensitive set — the literals are stored in lower case and a letter matches either case
  // (`(?i)(error|fail|panic)`); only scan_teddy_wave.hip knows it, like the assertions
  uint32_t looks;
  // kFlagPrefixLiteral images (a UseDFA program behind its required literal prefix): the anchored forward DFA,
  // [dfa_states][256] u8, that turns a prefix occurrence into the match end (0 states: plain literal set)
  uint32_t dfa_off, dfa_states, dfa_start, dfa_first_accept;
  // kKindTeddy images (round 6): offset of the PairImage below from the start of the BLOB (16-byte aligned; 0: none) — the tables of
  // scan_teddy_pair.hip, built once on the host (program.cc buildPairImage) and copied into LDS by every workgroup
  uint32_t pair_off;
};

// LDS address of the entry of the byte pair (b0, b1) in scan_teddy_pair.hip's table: (b0 | b1 << 8) ^ (b1 << 2).  The plain index puts a
// pair on bank (b0 >> 2) & 31 whatever b1 is — text whose pairs begin with a digit or a lower-case letter would use 3 + 7 of the 32 banks;
// the XOR spreads them by b1 as well.  A bijection of the 16 bits (b1 keeps its bits 6, 7; b0 is XORed with a function of b1).
CXG_HD constexpr uint32_t pair_addr(uint32_t b0, uint32_t b1) { return ((b0 | (b1 << 8)) ^ (b1 << 2)) & 0xFFFFu; }
// The kernel's tables as they sit in LDS (scan_teddy_pair.hip explains the entry bits):
struct PairImage {
  uint8_t tab[65536];        // entry of every byte pair, at pair_addr(b0, b1)
  uint32_t FB[256];          // by byte value: the verification slots [beg, end) of the literals that BEGIN with it (beg | end << 8); sync << 24
  uint32_t litx[64][8];      // verification slot k (literals ordered by first byte, then id): the first 12 bytes as three dwords, m0 | m1, m2 (their masks), length, id
  uint32_t maxrun;           // most literals sharing a first byte
  uint32_t pad[3];
};

// checkLook (nfa/pikevm.go:1646-1674) for one assertion at a position with the byte in front of it and the byte behind it; outside the
// haystack: -1 (a line edge, not a word byte)
CXG_HD bool teddy_look_holds(uint32_t look1, int prevb, int nextb) {
  if (look1 == 0u) return true;
  const bool pw = prevb >= 0 && ((prevb >= '0' && prevb <= '9') || (prevb >= 'A' && prevb <= 'Z') || prevb == '_' || (prevb >= 'a' && prevb <= 'z'));
  const bool nw = nextb >= 0 && ((nextb >= '0' && nextb <= '9') || (nextb >= 'A' && nextb <= 'Z') || nextb == '_' || (nextb >= 'a' && nextb <= 'z'));
  switch (look1) {
    case 3: return prevb < 0 || prevb == '\n';
    case 4: return nextb < 0 || nextb == '\n';
    case 5: return pw != nw;
    case 6: return pw == nw;
    default: return false;
  }
}

template <class Mem>
CXG_HD uint32_t teddy_mask_at(const Mem& m, const TeddyView& t, int32_t i, int32_t rend) {
  if (i + 1 >= rend) return 0;
  return (t.ab[m.byte(i)] & 0xFFu) & (t.ab[m.byte(i + 1)] >> 8);
}

// Mem concept adds: uint64_t cands(int32_t w), int32_t bitmap_limit().
template <class Mem>
CXG_HD int32_t next_cand(const Mem& m, const TeddyView& t, int32_t pos, int32_t limit, int32_t rend) {
  const int32_t blim = m.bitmap_limit();
  while (pos < limit) {
    if (pos < blim) {
      const uint64_t w = m.cands(pos >> 6) >> (pos & 63);
      if (w) return pos + static_cast<int32_t>(ctz64(w));
      pos = (pos | 63) + 1;
    } else {
      if (teddy_mask_at(m, t, pos, rend)) return pos;
      pos++;
    }
  }
  return limit;
}

template <class Mem, class Sink>
CXG_HD void lane_teddy(const Mem& m, const TeddyView& t, const uint8_t* info, int32_t c0, int32_t c1, int32_t rend,
                       bool chunk_at_origin, Sink& sink) {
  int32_t pos = first_owned_start(m, info, c0, c1, rend, chunk_at_origin);
  if (pos < 0) return;
  int32_t stop = c1 - 1;
  while (stop < rend && !(info[m.byte(stop)] & kInfoSync)) stop++;
  stop = stop < rend ? stop + 1 : rend;
  for (;;) {
    const int32_t c = next_cand(m, t, pos, stop, rend);
    if (c >= stop) return;
    const uint32_t mask = teddy_mask_at(m, t, c, rend);
    int32_t mlen = 0;
    for (uint32_t k = 0; k < t.nlits && !mlen; k++) {
      const uint32_t id = t.order[k];
      if (!((mask >> t.bucket[id]) & 1u)) continue;
      const int32_t len = t.lens[id];
      if (c + len > rend) continue;
      const uint8_t* lit = t.bytes + t.off[id];
      int32_t j = 0;
      while (j < len && m.byte(c + j) == lit[j]) j++;
      if (j == len) mlen = len;
    }
    if (mlen) { sink.emit(c, c + mlen); pos = c + mlen; }
    else pos = c + 1;
  }
}

// ---------------------------------------------------------------------------------------------
// Third form of the UseDigitPrefilter walk: candidates first, owners second.
//
// Preconditions, established on the host (program.cc) and recorded as kFlagFastDigit:
//   (a) digitRunSkipSafe (meta/compile.go:176): a failed candidate skips the rest of its digit run;
//   (b) "tail closed": from every accepting DFA state every digit leads to an accepting state, so a
//       match never ends in front of a digit.
// Under (a)+(b) the positions the reference's loop can ever try are exactly the digit-RUN STARTS
// (find_indices.go:1059 lands on a run start after a failure because of the skip, and after a match
// because of (b); the first candidate of a segment follows a sync byte).  Whether a candidate succeeds
// is a function of the bytes alone (SearchAtAnchored has no history), so all run starts of a tile can be
// verified independently — by any lane, perfectly balanced — and the sequential semantics reduce to:
// walk the owned segment's successful candidates in order and emit those with start >= previous end.
//
// verify_jump also skips digit runs in O(1): in a state that maps every digit to itself the walk
// cannot change state until the run ends, so it jumps to the run end with the digit bitmap.
constexpr uint32_t kStateDigitLoop = 1u;   // sflags[q]: every digit maps q -> q

template <class Mem>
CXG_HD int32_t verify_jump(const Mem& m, const DfaView& d, const uint8_t* sflags, int32_t c, int32_t rend) {
  uint32_t q = d.start;
  int32_t last = -1, i = c;
  for (;;) {
    if (q >= d.first_accept) last = i;
    if (i >= rend) break;
    const uint32_t b = m.byte(i);
    if ((sflags[q] & kStateDigitLoop) && is_digit(b)) { i = next_nondigit(m, i + 1, rend); continue; }
    q = d.T[rowmul(q, d.stride) + b];
    if (q == 0) break;
    i++;
  }
  return last;
}

// Greedy selection over the candidate list for the segments this lane owns.
// cand_pos[k] ascending; cand_len[k] = match length at that candidate, 0 = no match.
// first_idx = number of candidates below c0.  Candidates at or beyond list_limit (the staged range) are
// not in the list: that tail, if the lane's range reaches it, is walked directly.
template <class Mem, class Sink>
CXG_HD void lane_select(const Mem& m, const DfaView& d, const uint8_t* info, const uint8_t* sflags,
                        const uint16_t* cand_pos, const uint8_t* cand_len, uint32_t ncand, uint32_t first_idx,
                        int32_t list_limit, int32_t c0, int32_t c1, int32_t rend, bool chunk_at_origin, Sink& sink) {
  int32_t pos = first_owned_start(m, info, c0, c1, rend, chunk_at_origin);
  if (pos < 0) return;
  int32_t stop = c1 - 1;
  while (stop < rend && !(info[m.byte(stop)] & kInfoSync)) stop++;
  stop = stop < rend ? stop + 1 : rend;
  uint32_t k = first_idx;
  while (k < ncand) {
    const int32_t c = cand_pos[k];
    if (c >= stop) return;
    int32_t len = cand_len[k];
    k++;
    if (c < pos || len == 0) continue;
    if (len == 255) len = verify_jump(m, d, sflags, c, rend) - c;   // stored saturated: recompute the long match
    sink.emit(c, c + len);
    pos = c + len;
  }
  // the owned range runs past the candidate list (lane at the tile edge with no sync byte in the halo)
  if (stop > list_limit) {
    if (pos < list_limit) pos = list_limit;
    // a digit run straddling list_limit started inside the list; it is not a new candidate
    if (pos == list_limit && pos > 0 && pos < rend && is_digit(m.byte(pos - 1))) pos = next_nondigit(m, pos, rend);
    for (;;) {
      const int32_t c = next_digit(m, pos, stop);
      if (c >= stop) return;
      const int32_t e = verify_jump(m, d, sflags, c, rend);
      if (e >= 0) { sink.emit(c, e); pos = e > c ? e : c + 1; }
      else pos = next_nondigit(m, c + 1, rend);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Bit-parallel candidate pruning for the candidate-list kernels ("chain prefilter").
//
// When the anchored DFA is a chain — start -C1-> q1 (loops on C1) -F1-> q2 -C2-> q3 (loops on C2) ... —
// the leading elements of the pattern are known as a sequence of RUN(class+) / BYTE(class) steps.  A
// run start can only succeed if those steps can be taken from it, and that test is evaluated for all
// positions of a tile at once on per-class bitmaps (one bit per haystack byte), right to left:
//     BYTE(F):  G_k = F & (G_{k+1} >> 1)
//     RUN(C):   G_k = all bits of the C-runs whose end position is in G_{k+1}
// The run step is one multi-word addition: in bit-REVERSED space the marker just below a run, shifted
// onto it and added to C, ripples through exactly that run (Parabix-style ScanThru).  Surviving run
// starts (a superset of the successful candidates — the chain may be a prefix of the pattern, and
// class tests may be widened) are then verified with the DFA as before, so the result stays exact.
// All bitmaps of this scheme are stored reversed: bit i of word w describes byte N-1-(64w+i), N = 64*words.
enum ChainOpKind : uint8_t { kChainByte = 0, kChainRun = 1 };
enum ChainClassKind : uint8_t { kClsDigit = 0, kClsByte = 1, kClsRange = 2, kClsSet = 3 };   // set: union of 2..4 ASCII ranges (\w)
constexpr int kChainMaxOps = 64, kChainMaxCls = 4, kChainMaxRanges = 4;   // 63 steps at most (a 64-bit shift by the length): UUIDs, timestamps, MACs

struct ChainAux {             // aux section of a kKindDigit blob when kFlagChain is set (follows sflags[256])
  uint32_t nops, ncls;
  uint8_t op_kind[kChainMaxOps];
  uint8_t op_cls[kChainMaxOps];
  uint8_t cls_kind[kChainMaxCls];
  uint8_t cls_lo[kChainMaxCls];
  uint8_t cls_hi[kChainMaxCls];
  uint8_t cls_nr[kChainMaxCls];                       // kClsSet: number of ranges, then the ranges
  uint8_t cls_rlo[kChainMaxCls][kChainMaxRanges];
  uint8_t cls_rhi[kChainMaxCls][kChainMaxRanges];
  uint64_t run_bits;            // bit k: step k is a run            } the steps once more, packed for the scalar unit
  uint64_t cls2_lo, cls2_hi;    // 2 bits per step: its class (0..31, 32..63) } (scan_chain_wave.hip ChainRegs)
  uint8_t restart_check;        // the chain begins with a run and a match may end INSIDE a run of that class (last class meets the
                                // first): FindAll would resume there, which run-start candidates cannot express — the kernel hands
                                // the scan over when it sees such an end (program.cc extractChain)
  uint8_t pad[7];
};
constexpr uint32_t kFlagChainBounded = 512u;   // the ChainAux in aux describes the SURROGATE with unbounded runs of a bounded-repetition program
                                                // (`\d{1,3}\.\d{1,3}`...); the kernel filters rows by field length (scan_chain_wave.hip BND, cxg_program::chainBounds)
constexpr uint32_t kFlagPrefixLiteral = 256u;  // kKindBidir image whose aux section is a TeddyAux: every match begins with one literal (>= 3 bytes);
                                                // scan_teddy_wave.hip finds the occurrences and walks the anchored DFA from each, the DFA pair stays the fallback
constexpr uint32_t kFlagBothRestart = 128u;    // UseBoth program: the reference restarts its PikeVM 100 bytes before the DFA's match end
                                                // (find_indices.go:425-431) — identical to leftmost-first unless a match is longer than that
constexpr uint32_t kBothRestartSpan = 100u;
constexpr uint32_t kFlagCcRanges = 64u;         // kKindCharClass: membership is a union of <= 4 ASCII ranges (CharClassAux in aux)
struct CharClassAux { uint32_t nr; uint8_t lo[4], hi[4]; uint32_t neg; uint32_t pairs; };
// neg: the class is the COMPLEMENT of the ranges (bytes >= 0x80 are members).  pairs (round 4): the program is `Q[^Q]*Q` for the one
// byte Q of the class — the matches are the occurrences of Q taken two at a time from the start of the haystack (no byte
// synchronises: the parity of the occurrences in front decides whether a Q opens or closes).
// scan_delim_wave.hip: `O [^E]+ E` / `O [^E]* E` programs (`\[[^\]]+\]`, `<[^>]+>`); kept by cxg_program beside the DFA-pair / transducer
// images (the fallback), copied into ScanArgs::chain for the launch
struct DelimAux { uint32_t open_byte, close_byte, plus, on; };
constexpr uint32_t kFlagChainSets = 32u;        // some class is a kClsSet: only scan_chain_wave.hip evaluates those
constexpr uint32_t kFlagChain = 4u;
constexpr uint32_t kFlagChainOrdered = 16u;    // complete, and the k-th match start pairs with the k-th match end (program.cc extractChain)
constexpr uint32_t kFlagChainComplete = 8u;   // the chain is the whole DFA: survivors are matches, classes = alphabet

CXG_HD bool chain_class_has(const ChainAux& c, int k, uint32_t b) {
  if (c.cls_kind[k] == kClsDigit) return is_digit(b);
  if (c.cls_kind[k] == kClsSet) {
    bool in = false;
    for (int r = 0; r < c.cls_nr[k]; r++) in = in || (b >= c.cls_rlo[k][r] && b <= c.cls_rhi[k][r]);
    return in;
  }
  return b >= c.cls_lo[k] && b <= c.cls_hi[k];
}

// Sequential reference form (host emulator and tests): words[] arrays are reversed bitmaps of `nw` words.
// cls[k] = class bitmaps; out = G_1 (positions from which ops[0..nops) can all be taken).
inline void chain_eval_seq(const ChainAux& c, const uint64_t* const* cls, int nw, uint64_t* out, uint64_t* tmp) {
  for (int w = 0; w < nw; w++) out[w] = ~0ull;                      // G_{n+1}: nothing required after the chain
  for (int k = static_cast<int>(c.nops) - 1; k >= 0; k--) {
    const uint64_t* C = cls[c.op_cls[k]];
    if (c.op_kind[k] == kChainByte) {                                 // G_k = F & (G_{k+1} "next byte")
      uint64_t lowbit = 1;                                            // beyond the highest original position: G_{n+1}=1
      // next byte in original order = next LOWER reversed index: shift left, taking bit 63 of word w-1
      uint64_t carry = 0;
      for (int w = 0; w < nw; w++) {
        const uint64_t g = out[w];
        tmp[w] = C[w] & ((g << 1) | (w == 0 ? lowbit : carry));
        carry = g >> 63;
      }
      for (int w = 0; w < nw; w++) out[w] = tmp[w];
    } else {                                                          // run step
      // K = positions just below a reversed run (orig: first byte after the run) that are in G_{k+1}
      uint64_t carry = 0;                                             // multiword add carry
      uint64_t kprev_hi = 0;                                          // bit 63 of K in word w-1
      for (int w = 0; w < nw; w++) {
        const uint64_t cup = (C[w] >> 1) | ((w + 1 < nw ? C[w + 1] : 0ull) << 63);
        const uint64_t K = out[w] & ~C[w] & cup;
        const uint64_t M = (K << 1) | kprev_hi;
        kprev_hi = K >> 63;
        const uint64_t s1 = C[w] + M;
        const uint64_t c1 = s1 < M ? 1u : 0u;
        const uint64_t s2 = s1 + carry;
        const uint64_t c2 = s2 < s1 ? 1u : 0u;
        tmp[w] = C[w] & ~s2;
        carry = c1 | c2;
      }
      // a run whose end lies below reversed bit 0 (beyond the last original position): G_{n+1}=1 there only
      // matters at true end of input, handled by the caller through the virtual position below bit 0
      for (int w = 0; w < nw; w++) out[w] = tmp[w];
    }
  }
}

// ---- complete chains: end of match, ownership and halo test straight from the reversed bitmaps -------------
// W: reversed bitmap of `nw` words (bit i of word w <-> byte N-1-(64w+i), N = 64*nw).
CXG_HD int32_t rev_scan_down_zero(const uint64_t* W, int32_t i) {       // highest j <= i with bit j clear, or -1
  while (i >= 0) {
    const int32_t w = i >> 6, b = i & 63;
    const uint64_t m = ~W[w] & (b == 63 ? ~0ull : ((2ull << b) - 1ull));
    if (m) return (w << 6) + 63 - static_cast<int32_t>(__builtin_clzll(m));
    i = (w << 6) - 1;
  }
  return -1;
}
CXG_HD int32_t rev_scan_up_zero(const uint64_t* W, int32_t i, int32_t nbits) {   // lowest j >= i with bit j clear, or nbits
  while (i < nbits) {
    const int32_t w = i >> 6, b = i & 63;
    const uint64_t m = ~W[w] & (~0ull << b);
    if (m) return (w << 6) + static_cast<int32_t>(ctz64(m));
    i = (w + 1) << 6;
  }
  return nbits;
}
// Walks the chain forward from reversed index i (a surviving digit-run start).  Returns the reversed index of
// the first byte after the match, or -2 when the walk leaves the window (caller falls back to the DFA walk).
CXG_HD int32_t chain_walk_end(const ChainAux& c, const uint64_t* const* cls, int32_t i) {
  for (uint32_t k = 0; k < c.nops; k++) {
    const uint64_t* W = cls[c.op_cls[k]];
    if (i < 0) return -2;
    if (c.op_kind[k] == kChainRun) {
      i = rev_scan_down_zero(W, i);
      if (i < 0) return -2;
    } else {
      i -= 1;
    }
  }
  return i < 0 ? -2 : i;
}

template <class Mem, class Sink>
CXG_HD void lane_bidir(const Mem& m, const DfaView& f, const DfaView& r, const uint8_t* info, int32_t c0, int32_t c1,
                       int32_t rend, bool chunk_at_origin, Sink& sink) {
  int32_t pos = first_owned_start(m, info, c0, c1, rend, chunk_at_origin);
  if (pos < 0) return;
  for (;;) {
    // forward: end of the leftmost-first match at >= pos (lazy.go:1102-1315)
    uint32_t q = f.start;
    int32_t last = -1, i = pos;
    for (;;) {
      if (q >= f.first_accept) last = i;
      if (i >= rend) break;
      if (q == f.start && i >= c1 && last < 0 && i > 0 && (info[m.byte(i - 1)] & kInfoSync)) return;
      q = f.T[rowmul(q, f.stride) + m.byte(i)];
      if (q == 0) break;
      i++;
    }
    if (last < 0) return;  // findall.go:228-230
    // reverse: leftmost start in [pos, last) (lazy.go:1769-1920)
    uint32_t s = r.start;
    int32_t st = -1;
    for (int32_t at = last - 1; at >= pos; at--) {
      s = r.T[rowmul(s, r.stride) + m.byte(at)];
      if (s == 0) break;
      if (s >= r.first_accept) st = at;
    }
    if (st < 0) return;    // findall.go:235-237
    sink.emit(st, last);
    pos = last > pos ? last : pos + 1;
  }
}

// ---------------------------------------------------------------------------------------------
// Capture pass for FindAllSubmatchIndex (meta/findall.go:390-447 -> PikeVM slots, nfa/pikevm.go:2186).
// The span [s, e) of every match comes from the bidirectional DFA kernel (same leftmost-first result
// as the PikeVM's group 0).  For one-pass patterns — at every point of the match at most one
// byte-consuming NFA state accepts the next byte — the path through the NFA is forced, so the slots
// are recovered by ONE anchored walk over the span: `ent` is the current entry point (an NFA state
// whose epsilon closure is about to be taken), next[ent][byte] the entry after consuming the byte, and
// mask[...] the capture slots whose Capture states lie on the (first, in DFS priority order) epsilon
// path taken — they are stamped with the current position, exactly where addSearchThread stamps them
// (pikevm.go:1986).  fin[ent] is the slot set on the path to Match.  Unset groups stay -1.
struct CapView {
  const uint8_t* next;     // [n_entries][256], 0xFF = no transition
  const uint8_t* maskid;   // [n_entries][256]
  const uint8_t* fin;      // [n_entries], 0xFF = Match not reachable
  const uint32_t* masks;   // [n_masks] slot bitmasks (bit k = slot k, k >= 2)
  uint32_t n_entries, start_entry;
};

// Captures of a chain program straight from the chain kernel (scan_chain_wave.hip, CAP instantiations): every
// boundary between chain steps is forced, so slot k of a match is one of the positions the kernel compacts anyway —
// match start, match end, or the end of one of up to four runs — plus a small constant (program.cc deriveChainCaps).
constexpr uint8_t kCapSrcStart = 0, kCapSrcEnd = 1, kCapSrcRun0 = 2, kCapSrcUnset = 7;   // Run0 + i: end of captured run i
constexpr int kCapMaxRuns = 4;
struct ChainCaps {            // 40 bytes, travels in ScanArgs
  uint8_t on;                 // 1: capture slots; 2: bounded repetition — nruns = fields but the last, src[x] = min and src[8 + x] = max
                              //    (0 = unbounded) of field x, no slots
  uint8_t nslots;             // 2 * groups (<= 16)
  uint8_t nruns;              // run ends the kernel compacts (0..4): 4 KiB of LDS each
  uint8_t pad;
  uint8_t run_op[kCapMaxRuns];  // chain step (a run) whose end positions are compacted as run i; 0xFF: unused
  uint8_t src[16];            // per slot: kCapSrc*
  int8_t off[16];             // per slot: added to the source position
};

struct CapHeader {          // device image: header then the arrays, offsets from the header start
  uint32_t magic, n_entries, start_entry, n_masks, nslots;
  uint32_t next_off, maskid_off, fin_off, masks_off, total_bytes;
};

// hay: absolute haystack; row: 2*ngroups int64 with row[0], row[1] already holding s, e.
// Returns false if the table and the span disagree (must not happen; reported as an internal error).
CXG_HD bool capture_walk(const CapView& c, const uint8_t* hay, int64_t* row, uint32_t nslots) {
  const int64_t s = row[0], e = row[1];
  for (uint32_t k = 2; k < nslots; k++) row[k] = -1;
  uint32_t ent = c.start_entry;
  for (int64_t i = s; i < e; i++) {
    const uint32_t b = hay[i];
    const uint32_t nx = c.next[ent * 256 + b];
    if (nx == 0xFFu) return false;
    uint32_t m = c.masks[c.maskid[ent * 256 + b]];
    while (m) { const uint32_t k = ctz32(m); m &= m - 1; row[k] = i; }
    ent = nx;
  }
  const uint32_t f = c.fin[ent];
  if (f == 0xFFu) return false;
  uint32_t m = c.masks[f];
  while (m) { const uint32_t k = ctz32(m); m &= m - 1; row[k] = e; }
  return true;
}

}  // namespace cxgdev
