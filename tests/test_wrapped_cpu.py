"""Literals between assertions (`\\berror\\b`, `(?m)^(GET|POST)`, `(?m)error$`: round 4): recognised on the NFA (program.cc
wrappedLiterals), served by the literal kernel with the assertions checked in its verification.  CPU tier: the kernel's twin
against the oracle; shapes that are something else keep their transducer-only program."""
import random
import struct

import numpy as np
import pytest

import coregex_amd as cx
import emu

WRAPPED = [r"\berror\b", r"(?m)^(GET|POST|PUT|DELETE|PATCH)", r"\Berr", r"(?m)error$", r"\berror", r"(?m)^abc$", r"error\B", r"\b(?:warn|fatal)\b",
           r"(?m)^(?:abc|xyz)$", r"\bGET\b"]
TOKS = [b"error", b"err", b"GET", b"POST", b"warn", b"fatal", b"abc", b"xyz", b" ", b"\n", b"_", b"x", b"-", b"PUT", b"errors", b"aerror", b"9", b"\xc3\xa9"]


def _kind(rx):
    return struct.unpack_from("<I", rx.blob(), 4)[0]


@pytest.mark.parametrize("pat", WRAPPED)
def test_twin_equals_oracle(pat, oracle):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.supported and rx.strategy == o.strategy and _kind(rx) == 4 and rx.fsm_image() is not None, (pat, rx.strategy)
    rng = random.Random(len(pat) * 5)
    checked = 0
    for n in [0, 1, 5, 100, 3839, 3840, 3841, 9000, 40000] * 4:
        hay = np.frombuffer(b"".join(rng.choice(TOKS) for _ in range(max(1, n // 3)))[:n], dtype=np.uint8)
        exp = o.find_all_index(hay)
        got = emu.find_all_teddy_wave(rx.blob(), hay)
        if isinstance(got, int):
            continue                                                # a tile the kernel hands to the transducer
        checked += 1
        assert np.array_equal(got, exp), (pat, n, bytes(hay[:60]))
        img = emu.find_all_fsm(rx.fsm_image(), hay, 3840, 32)        # the fallback image is the pattern's own transducer
        assert isinstance(img, int) or np.array_equal(img, exp), (pat, n)
    assert checked >= 12


@pytest.mark.parametrize("pat", [r"\b(err|error)\b", r"\bab\b", r"\b\d+\b", r"\berror\w*", r"\b[A-Z]+\b", r"x\berror\b", r"\b(GET|POST)\b /"])
def test_other_shapes_keep_the_transducer(pat):
    rx = cx.compile(pat)
    assert not rx.supported or _kind(rx) != 4


# Case-insensitive literal alternations: too many case variants for the reference's Teddy (its PikeVM answers: UseNFA), one folded
# literal set for the literal kernel (walk.hpp kTeddyFold).
FOLDED = [r"(?i)(error|fail|exception|panic|fatal)", r"(?i)(googlebot|bingbot|yandexbot)", r"(?i)(jan|feb|mar|apr|may|jun|jul|aug|sep|oct|nov|dec)",
          r"(?i)(connection|session)_(reset|closed)", r"(?i)\b(error|fail|exception|panic|fatal)\b"]
FTOKS = [b"error", b"ERROR", b"Error", b"eRRoR", b"fail", b"FAIL", b"exception", b"Exception", b"panic", b"PANIC", b"fatal", b"FaTaL", b"googlebot", b"GoogleBot",
         b"BINGBOT", b"yandexBot", b"timeout", b"TimeOut", b"refused", b"UNREACHABLE", b"denied", b"warning", b"WARNING", b"Critical", b"diskfull", b"RISKFULL",
         b"taskFull", b"di\xc5\xbfkfull", b"ris\xe2\x84\xaafull", b"error: 5", b"ERROR: 7", b"Error: x", b"connection_reset", b"SESSION_CLOSED", b"Connection_Closed",
         b"jan", b"JUN", b"Jul", b"sEp", b"DEC", b"juN", b" ", b"\n", b"_", b"x", b"-", b"9", b"err", b"ERR", b"\xc3\xa9", b"K", b"s"]


@pytest.mark.parametrize("pat", FOLDED)
def test_folded_twin_equals_oracle(pat, oracle):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.supported and rx.strategy == o.strategy == "UseNFA", (pat, rx.strategy, o.strategy)
    blob = rx.blob()
    assert _kind(rx) == 4 and (rx.fsm_image() is not None or "\\b" in pat), pat     # (64+ symbols x kinds: no transducer behind the literal kernel)
    looks = struct.unpack_from("<I", blob, struct.unpack_from("<I", blob, 56)[0] + 44)[0]
    assert looks & 0x10000, (pat, hex(looks))
    rng = random.Random(len(pat) * 11)
    checked = 0
    for n in [0, 1, 5, 100, 3839, 3840, 3841, 9000, 40000] * 4:
        parts, have = [], 0
        while have < n:
            parts.append(rng.choice(FTOKS) if rng.random() < 0.25 else bytes(rng.choices(b"abcdefghijklmnopqrstuvwxyzEORF  \n:_0", k=rng.randint(1, 12))))
            have += len(parts[-1])
        hay = np.frombuffer(b"".join(parts)[:n], dtype=np.uint8)
        exp = o.find_all_index(hay)
        got = emu.find_all_teddy_wave(blob, hay)
        if isinstance(got, int):
            continue
        checked += 1
        assert np.array_equal(got, exp), (pat, n, bytes(hay[:60]), got[:4].tolist(), exp[:4].tolist())
    assert checked >= 12


@pytest.mark.parametrize("pat", [r"x(?i:yz)w|abcd", r"(?i)(err|error)", r"(?i)(ab|cd)", r"(?i)error|eRRor\d"])
def test_mixed_or_overlapping_case_sets_stay_where_they_were(pat):
    rx = cx.compile(pat)
    if rx.supported and _kind(rx) == 4:
        blob = rx.blob()
        assert not struct.unpack_from("<I", blob, struct.unpack_from("<I", blob, 56)[0] + 44)[0] & 0x10000, pat
