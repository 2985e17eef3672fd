"""The pair kernel for literal sets (round 6, `scan_teddy_pair.hip`): one fingerprint lookup per byte PAIR, persistent 16-wave workgroups
with claimed groups.  Rows against the oracle at every edge of its geometry (wave-tile 3 840 B, window 4 096 B, unit = 8 tiles = 30 720 B,
group = 16 units = 491 520 B; small groups of 2-tile units behind them), for literals of 3 / 4 / 5+ bytes at even and odd offsets, folded sets, assertions, Fat sets and literals
longer than the 12 bytes the verifier compares at once; and against the wave kernel (`scan_teddy_wave.hip`, which FindAll's n keeps a
call on) over hundreds of haystack lengths — the claims of a launch are handed out in an order that depends on timing."""
import random

import numpy as np
import pytest

import coregex_amd as cx
from routing import routed

pytestmark = pytest.mark.gpu

TILE, WIN, UNIT, GROUP = 3840, 4096, 3840 * 8, 3840 * 8 * 16
USMALL, GSMALL = 3840 * 2, 3840 * 2 * 16       # the last stretch of a haystack (one big group per CU: all of a short one) is cut into small groups
K_PAIR, K_WAVE = 21, 7


def _rows(rx, hay, n=-1):
    import torch
    a = np.ascontiguousarray(hay)
    d = torch.from_numpy(a.copy()).cuda() if a.size else torch.zeros(16, dtype=torch.uint8, device="cuda")
    t = cx.Timing()
    cnt = rx.find_all_device(d.data_ptr(), a.size, timing=t, n=n)
    out = torch.zeros((cnt + 4, 2), dtype=torch.int64, device="cuda")
    t2 = cx.Timing()
    got = rx.find_all_device(d.data_ptr(), a.size, out.data_ptr(), cnt + 4, timing=t2, n=n)
    assert got == cnt
    return out[:cnt].cpu().numpy(), t2


def _check(oracle, pat, hay, want_kernel=None):
    rx = cx.compile(pat)
    assert rx.supported, rx.why_unsupported
    exp = oracle.Regex(pat).find_all_index(hay)
    got, t = _rows(rx, hay)
    if got.shape != exp.shape or not np.array_equal(got, exp):
        miss = sorted(set(map(tuple, exp.tolist())) - set(map(tuple, got.tolist())))[:4]
        extra = sorted(set(map(tuple, got.tolist())) - set(map(tuple, exp.tolist())))[:4]
        raise AssertionError((pat, len(hay), len(got), len(exp), "missing", miss, "extra", extra, list(t.kernels)))
    if want_kernel is not None:
        assert routed(list(t.kernels)[:1] == [want_kernel], pat, len(hay), list(t.kernels), t.fallback_reason)
    return t


def test_edges_of_tile_window_unit_and_group(oracle):
    pat = "spider|error|crawler|denied"
    n = GROUP + UNIT + 9000
    base = np.full(n, ord(" "), dtype=np.uint8)
    spots = []
    for edge in (64, TILE, WIN, 2 * TILE, USMALL, USMALL + TILE, 3 * USMALL, GSMALL, GSMALL + USMALL, UNIT, UNIT + TILE, 8 * UNIT, GROUP, GROUP + UNIT):
        spots += list(range(edge - 9, edge + 3))
    for lit in (b"spider", b"error", b"crawler"):
        ip = np.frombuffer(lit, dtype=np.uint8)
        for off in spots:
            hay = base.copy()
            hay[off:off + len(ip)] = ip
            _check(oracle, pat, hay, K_PAIR)
    for cut in (3, 5, 6, 7):                                        # the haystack ends inside / right behind a literal, at every parity
        for m in (TILE, UNIT, GROUP, GROUP + 5 * TILE + 1):
            hay = base[:m + cut].copy()
            hay[m:m + cut] = np.frombuffer(b"crawler"[:cut], dtype=np.uint8)
            _check(oracle, pat, hay)
            hay[m - 7:m] = np.frombuffer(b"crawler", dtype=np.uint8)
            _check(oracle, pat, hay)


def test_short_literals_at_both_parities(oracle):
    """Literals of 3 and 4 bytes: the fingerprint asks for bytes they do not have (every value qualifies); a start at an even offset reads
    AB / CD / E1, at an odd one A2 / BC / DE."""
    for pat in ("abc|wxyz|hello", "abc|xyz", "abcd|wxyz", "ab1|cd2x|ef3yz|gh4uvw"):
        lits = [l.encode() for l in pat.split("|")]
        rng = random.Random(len(pat))
        for n in (7, 63, TILE - 1, TILE + 2, WIN + 1, UNIT + 5, 200001):
            toks = lits + [l[:-1] for l in lits] + [l[1:] for l in lits] + [b" ", b"\n", b"-", b"q", b"ab", b"xy", b"aabc", b"abcabc"]
            hay = np.frombuffer(b"".join(rng.choice(toks) for _ in range(n // 2 + 2))[:n], dtype=np.uint8)
            _check(oracle, pat, hay)
            _check(oracle, pat, hay[1:])                              # every match at the other parity
    hay = np.full(UNIT + 40, ord("."), dtype=np.uint8)
    for off in list(range(TILE - 6, TILE + 3)) + list(range(UNIT - 6, UNIT + 3)) + [UNIT + 37]:
        h = hay.copy()
        h[off:off + 3] = np.frombuffer(b"xyz", dtype=np.uint8)
        _check(oracle, "abc|xyz", h)


def test_folded_sets_assertions_fat_and_long_literals(oracle):
    rng = random.Random(5)
    words = [b"error", b"ERROR", b"Error", b"fail", b"FAIL", b"panic", b"Panic", b"errors", b"xerror", b"error_", b" ", b"\n", b"-", b"_", b"9", b"abc", b"xyz",
             b"connection_reset_by_peer", b"connection_reset_by_pear", b"Connection_Reset_By_Peer", b"session_closed_cleanly", b"GET", b"POST"]
    for n in (5000, UNIT + 77, GROUP + 12345):
        hay = np.frombuffer(b"".join(rng.choice(words + [b" pad pad pad pad pad "] * 6) for _ in range(n // 6))[:n], dtype=np.uint8)
        for pat in ("(?i)(error|fail|panic)", r"\berror\b", r"(?m)^abc$", r"(?m)^(GET|POST)", r"error\B",
                    "connection_reset_by_peer|session_closed_cleanly|error", "(?i)(connection_reset_by_peer|session_closed_cleanly)",
                    "connection_reset_by_peer|connection_reset_by_pear|connection_refused",     # two literals agree on their first 12 bytes and differ behind them
                    "connection_reset_by_pear|connection_reset_by_peer|connection_reset_by_peek|Connection_Reset_By_Peer"):
            _check(oracle, pat, hay)
    fat = ["word%02d" % i for i in range(20)] + ["key%02dx" % i for i in range(12)] + ["val%d" % i for i in range(10)] + ["item", "timeout", "refused", "denied", "ordinal", "keyword"]
    lit = [w.encode() for w in fat]
    hay = np.frombuffer(b"".join(rng.choice(lit + [b"word", b"key1", b"val", b"ite", b" lorem ipsum dolor sit amet "] * 3) for _ in range(60000)), dtype=np.uint8)
    _check(oracle, "|".join(fat), hay)


def test_dense_input_falls_back_to_the_wave_kernel(oracle):
    pat = "spider|error|crawler|denied"
    t = _check(oracle, pat, np.frombuffer(b"error " * 90000, dtype=np.uint8))          # more rows than a unit's buffer
    assert routed(K_WAVE in list(t.kernels) or K_PAIR not in list(t.kernels), list(t.kernels))
    _check(oracle, pat, np.frombuffer(b"errorerrorerror" * 20000, dtype=np.uint8))     # no synchronising byte at all
    _check(oracle, pat, np.frombuffer(b"erro spide crawle denie " * 30000, dtype=np.uint8))   # fingerprints hit, no literal matches


def test_rows_equal_the_wave_kernels_over_many_lengths():
    """The same call with FindAll's n set stays on scan_teddy_wave.hip (capi_ladder.hip): two device kernels, one answer — over haystack
    lengths from one byte to megabytes, so that last groups are partial, workgroups steal claims and units lie behind the end of input
    (round 6: a stolen claim behind such a unit once met a stale window; one test run in twelve saw it)."""
    import torch
    from test_wrapped_cpu import TOKS
    rng = random.Random(99)
    big = np.frombuffer(b"".join(rng.choice(TOKS + [b" pad pad pad pad pad pad pad pad "] * 12) for _ in range(600_000)), dtype=np.uint8)
    d = torch.from_numpy(big.copy()).cuda()
    pats = [r"(?m)^(?:abc|xyz)$", r"\berror\b", "error|warn|fatal|abc", r"(?m)^(GET|POST|PUT|DELETE|PATCH)"]
    out_a = torch.zeros((600000, 2), dtype=torch.int64, device="cuda")
    out_b = torch.zeros_like(out_a)
    on_pair = 0
    for rep in range(1200):
        if rep % 40 == 0:
            rxs = [cx.compile(p) for p in pats]                      # a remembered fallback keeps a program off the pair kernel
        rx = rxs[rep % len(pats)]
        n = rng.choice([rng.randrange(1, 4000), rng.randrange(3000, 700000), rng.randrange(500000, big.size - 64)])
        off = rng.randrange(0, (big.size - n) // 16 + 1) * 16
        t = cx.Timing()
        ca = rx.find_all_device(d.data_ptr() + off, n, out_a.data_ptr(), 600000, timing=t)
        cb = rx.find_all_device(d.data_ptr() + off, n, out_b.data_ptr(), 600000, n=1 << 40)
        on_pair += list(t.kernels)[:1] == [K_PAIR] and t.n_launches == 1
        assert ca == cb and torch.equal(out_a[:ca], out_b[:cb]), (pats[rep % len(pats)], n, off, ca, cb, list(t.kernels))
    assert routed(on_pair >= 600, on_pair)


def test_config_3_on_the_synthetic_corpus_against_the_wave_kernel():
    import torch
    lits = "error|warning|fatal|critical|panic|timeout|refused|denied|googlebot|bingbot|yandexbot|crawler|spider|failure|exception|overflow"
    n = 256 << 20
    buf = cx.DeviceBuffer(n)
    buf.fill_synth(3, 0xC0FFEE03, 0)
    rx = cx.compile(lits)
    cnt = rx.find_all_device(buf.ptr, n)
    out_a = torch.zeros((cnt + 8, 2), dtype=torch.int64, device="cuda")
    out_b = torch.zeros_like(out_a)
    t = cx.Timing()
    assert rx.find_all_device(buf.ptr, n, out_a.data_ptr(), cnt + 8, timing=t) == cnt
    assert rx.find_all_device(buf.ptr, n, out_b.data_ptr(), cnt + 8, n=1 << 40) == cnt
    assert torch.equal(out_a[:cnt], out_b[:cnt]) and cnt > 1_000_000
    assert routed(list(t.kernels) == [K_PAIR], list(t.kernels), t.fallback_reason)


def test_default_routing():
    """Without CXG_PAIR_MIN_BYTES (a border for A/B runs; this tier sets 0 explicitly): literal sets run on the pair kernel at every length,
    FindAll with an n on the wave kernel."""
    import os, subprocess, sys
    code = ("import coregex_amd as cx\n"
            "n = 64 << 20\n"
            "buf = cx.DeviceBuffer(n); buf.fill_synth(3, 0xC0FFEE03, 0)\n"
            "rx = cx.compile('error|warning|fatal|critical')\n"
            "t = cx.Timing(); a = rx.find_all_device(buf.ptr, 4096, timing=t); ka = list(t.kernels)\n"
            "b = rx.find_all_device(buf.ptr, n, timing=t); kb = list(t.kernels)\n"
            "c = rx.find_all_device(buf.ptr, n, n=5, timing=t); kc = list(t.kernels)\n"
            "print('ROUTE', ka, kb, kc, a, b, c)\n")
    env = dict(os.environ)
    env.pop("CXG_PAIR_MIN_BYTES", None)
    env["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("ROUTE")]
    assert line, out.stderr[-2000:]
    assert "[21] [21] [7]" in line[0], line[0]
