"""Host restatement (numpy) of the mathematics of `coregex_amd/csrc/device/scan_teddy_pair.hip` (round 6): the pair table's entry
bits as the kernel's workgroups build them, the address swizzle, and the combination W_k & (W_k+1 >> 2) & (W_k+2 >> 4) & 3 on four
pairs per dword followed by v_dot4 with weights 1, 4, 16, 64.  Property: the candidate set is a SUPERSET of the true starts of the
literals (any superset is valid — candidates are verified exactly), at both parities, for literals of 3, 4 and 5+ bytes, with and
without case folding, up to the last byte of the input (bytes past the data read as 0); the synchronising bits are exactly the bytes
outside the literals' alphabet."""
import random

import numpy as np


def pair_addr(b0, b1):
    return ((b0 | (b1 << 8)) ^ (b1 << 2)) & 0xFFFF


def build_table(lits, fold):
    """PairLds::tab as k_scan_teddy_pair fills it: entry(b0, b1) = F[b0] | G[b1] | exact pair bits, at pair_addr(b0, b1)."""
    def variants(c):
        return [c, c ^ 0x20] if fold and ord("a") <= c <= ord("z") else [c]
    alpha = set()
    for l in lits:
        for c in l:
            alpha.update(variants(c))
    F = np.zeros(256, np.uint8)
    G = np.zeros(256, np.uint8)
    for b in range(256):
        f = 0x40 if b not in alpha else 0
        g = 0x80 if b not in alpha else 0
        for l in lits:
            if b in variants(l[0]): g |= 2
            if len(l) == 3:
                if b in variants(l[2]): f |= 4
                f |= 0x30
            elif len(l) == 4:
                if b in variants(l[3]): f |= 0x20
                f |= 0x10
            elif b in variants(l[4]): f |= 0x10
        F[b], G[b] = f, g
    tab = np.zeros(65536, np.uint8)
    for b1 in range(256):
        for b0 in range(256):
            tab[pair_addr(b0, b1)] = F[b0] | G[b1]
    for l in lits:
        for k, bit in ((0, 1), (1, 8), (2, 4), (3, 0x20)):
            if k + 1 < len(l):
                for c0 in variants(l[k]):
                    for c1 in variants(l[k + 1]):
                        tab[pair_addr(c0, c1)] |= bit
    return tab


def device_bits(tab, hay):
    """Candidate and synchronising bitmaps of one haystack, by the kernel's dword arithmetic (dwords of four pair entries)."""
    n = len(hay)
    pad = np.zeros(((n + 7) // 8 + 2) * 8, np.uint8)
    pad[:n] = hay
    p = pad.astype(np.uint32)
    W = tab[pair_addr(p[0::2], p[1::2])].astype(np.uint64)            # one entry per pair
    nd = len(W) // 4
    D = (W[0::4] | (W[1::4] << 8) | (W[2::4] << 16) | (W[3::4] << 24))[:nd]
    nxt = np.concatenate([D[1:], np.zeros(1, np.uint64)])
    both = D | (nxt << 32)
    a1 = (both >> 10) & 0xFFFFFFFF                                     # v_alignbit_b32(next, W, 10)
    a2 = (both >> 20) & 0xFFFFFFFF
    c = D & a1 & a2 & 0x03030303
    z = D & 0xC0C0C0C0

    def dot4(v):                                                      # v_dot4_u32_u8 with weights 0x40100401
        return (v & 0xFF) + 4 * ((v >> 8) & 0xFF) + 16 * ((v >> 16) & 0xFF) + 64 * ((v >> 24) & 0xFF)
    cd, zd = dot4(c), dot4(z) >> 6
    cand = np.zeros(nd * 8, bool)
    sync = np.zeros(nd * 8, bool)
    for bit in range(8):
        cand[bit::8] = (cd >> bit) & 1
        sync[bit::8] = (zd >> bit) & 1
    return cand[:n], sync[:n]


def test_pair_addr_is_a_bijection_that_spreads_banks():
    idx = np.array([pair_addr(b0, b1) for b1 in range(256) for b0 in range(256)])
    assert len(set(idx.tolist())) == 65536
    digits = {(pair_addr(b0, b1) >> 2) & 31 for b0 in range(0x30, 0x3A) for b1 in range(0x30, 0x3A)}
    plain = {((b0 | b1 << 8) >> 2) & 31 for b0 in range(0x30, 0x3A) for b1 in range(0x30, 0x3A)}
    assert len(plain) == 3 and len(digits) == 12                       # digit pairs: three banks without the swizzle
    lower = {(pair_addr(b0, b1) >> 2) & 31 for b0 in range(0x61, 0x7B) for b1 in range(0x61, 0x7B)}
    assert len({((b0 | b1 << 8) >> 2) & 31 for b0 in range(0x61, 0x7B) for b1 in range(0x61, 0x7B)}) == 7 and len(lower) == 32


def test_candidates_are_a_superset_of_the_true_starts():
    rng = random.Random(2026)
    alphabet = b"abcdefgxyzERO01 .\n"
    for trial in range(60):
        fold = trial % 3 == 0
        nl = rng.randrange(2, 20)
        lits = set()
        while len(lits) < nl:
            L = rng.choice([3, 3, 4, 4, 5, 6, 9, 14])
            w = bytes(rng.choice(b"abcdefgxyz01") for _ in range(L))
            if not any(w.startswith(o) or o.startswith(w) for o in lits):
                lits.add(w)
        lits = sorted(lits)
        tab = build_table(lits, fold)
        toks = lits + [l[:-1] for l in lits] + [l[1:] for l in lits] + [bytes([c]) for c in alphabet]
        if fold:
            toks += [l.upper() for l in lits] + [l.title() for l in lits]
        for n in (1, 2, 5, 64, 257, 4000):
            hay = np.frombuffer(b"".join(rng.choice(toks) for _ in range(n))[:n + rng.randrange(0, 4)], dtype=np.uint8)
            cand, sync = device_bits(tab, hay)
            hb = hay.tobytes()
            cmp = hb.lower() if fold else hb
            for l in lits:
                ll = l.lower() if fold else l
                s = cmp.find(ll)
                while s >= 0:
                    if not fold or all((hb[s + i] == l[i]) or (chr(l[i]).isalpha() and chr(l[i]).islower() and (hb[s + i] | 0x20) == l[i]) for i in range(len(l))):
                        assert cand[s], (lits, fold, n, s, l)
                    s = cmp.find(ll, s + 1)
            alpha = set()
            for l in lits:
                for c in l:
                    alpha.add(c)
                    if fold and ord("a") <= c <= ord("z"): alpha.add(c ^ 0x20)
            assert np.array_equal(sync, np.array([b not in alpha for b in hb], bool))


def test_config_3_fingerprint_is_selective():
    """BASELINE config 3's sixteen literals on a log-like text: candidates stay within 1.6 x the matches (20.6 against 15.8 per
    3 840 bytes on synthlog-v1; the three-byte, eight-bucket fingerprint of scan_teddy_wave.hip: 21.8)."""
    import re
    lits = [l.encode() for l in "error|warning|fatal|critical|panic|timeout|refused|denied|googlebot|bingbot|yandexbot|crawler|spider|failure|exception|overflow".split("|")]
    rng = random.Random(3)
    words = lits + [b"GET /index.html HTTP/1.1", b"200 5123", b"127.0.0.1 - -", b"[10/Dec/2024:01:48:01 +0000]", b"worker", b"search", b"item metrics at line", b"parse", b"created", b"writer"]
    hay = np.frombuffer(b" ".join(rng.choice(words) for _ in range(20000)), dtype=np.uint8)
    cand, _ = device_bits(build_table(lits, False), hay)
    m = len(re.findall(b"|".join(lits), hay.tobytes()))
    assert m <= int(cand.sum()) <= int(1.6 * m)


def test_host_built_image_equals_the_restatement():
    """The program image of a literal set ends with the pair kernel's tables (walk.hpp PairImage, built by program.cc buildPairImage):
    the pair table equals build_table() above byte for byte; the slot ranges by first byte and the slot records are consistent."""
    import struct
    import coregex_amd as cx
    cases = [("error|warning|fatal|critical|panic|timeout|refused|denied|googlebot|bingbot|yandexbot|crawler|spider|failure|exception|overflow", False),
             ("abc|wxyz|hello", False), ("abc|xyz", False), ("(?i)(error|fail|panic)", True), ("connection_reset_by_peer|connection_reset_by_pear|connection_refused", False)]
    size = 65536 + 1024 + 64 * 32 + 16
    for pat, fold in cases:
        rx = cx.compile(pat)
        assert rx.supported and rx.strategy in ("UseTeddy", "UseNFA", "UseDFA"), (pat, rx.strategy)
        blob = rx.blob()
        img = blob[-size:]
        lits = [l.encode() for l in (pat[5:-1] if fold else pat).split("|")]
        tab = np.frombuffer(img[:65536], np.uint8)
        want = build_table(lits, fold)
        assert np.array_equal(tab, want), (pat, int((tab != want).sum()))
        FB = np.frombuffer(img[65536:65536 + 1024], np.uint32)
        litx = np.frombuffer(img[65536 + 1024:65536 + 1024 + 2048], np.uint32).reshape(64, 8)
        maxrun = struct.unpack_from("<I", img, 65536 + 1024 + 2048)[0]
        firsts = sorted(l[0] for l in lits)
        assert maxrun == max(firsts.count(c) for c in set(firsts))
        for b in range(256):
            nb = (b | 0x20) if fold and ord("A") <= b <= ord("Z") else b
            beg, end = int(FB[b]) & 0xFF, (int(FB[b]) >> 8) & 0xFF
            assert end - beg == firsts.count(nb) and all(int(litx[k][0]) & 0xFF == nb for k in range(beg, end)), (pat, b)
        ids = sorted(int(litx[k][7]) for k in range(len(lits)))
        assert ids == list(range(len(lits))) and all(int(litx[k][6]) == len(lits[int(litx[k][7])]) for k in range(len(lits)))
