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
