"""CPU tier for the persistent kernel's literal mode (scan_fields_wave.hip lit_core, round 5): the sequential twin
(tests/emu/emu_fields.cc emu_find_all_literal) against the oracle — which literals the mode takes (border-free, 2..4 distinct
ASCII bytes), occurrences across word, lane and tile borders, literals longer than a half word and than 32 bytes."""
import random

import numpy as np
import pytest

import coregex_amd as cx
import emu

TAKEN = ["error", "GET", "ab", "abc", "abcd", "HTTP/", "aab", "abb", "fatal", "xyyyyyyyyyyyyyyyyyyyyyyyyyyyyyyyyyz", "ab" + "b" * 40, "a" * 30 + "b", "abccba" + "d"]
NOT_TAKEN = ["denied", "aa", "abab", "aba", "abcab", "a", "warning", "xyzx", "abcde"]   # borders; one byte; more than four distinct bytes


@pytest.mark.parametrize("lit", TAKEN)
def test_taken_literals_give_the_oracle_rows(lit, oracle):
    rx, o = cx.compile(lit), oracle.Regex(lit)
    assert rx.supported
    nc = emu.literal_shape(rx.blob())
    assert nc == len(set(lit)), (lit, nc)
    rng = random.Random(len(lit) * 131 + ord(lit[0]))
    alpha = sorted(set(lit)) + ["x", " "]
    for it in range(120):
        n = rng.choice([0, 1, len(lit), 63, 64, 65, 127, 128, 129, 200, 700, 4100, 9000])
        parts = []
        while sum(map(len, parts)) < n:
            parts.append(lit if rng.random() < 0.25 else lit[: rng.randrange(1, len(lit) + 1)] if rng.random() < 0.4 else "".join(rng.choices(alpha, k=rng.randrange(1, 9))))
        hay = "".join(parts)[:n].encode()
        exp = o.find_all_index(np.frombuffer(hay, dtype=np.uint8) if hay else np.zeros(0, dtype=np.uint8))
        for ow, pw in ((60, 2), (3, 2), (1, 2), (60, 1), (2, 1)):
            got = emu.find_all_literal(rx.blob(), hay, ow, pw)
            assert got.shape == exp.shape and np.array_equal(got, exp), (lit, ow, pw, hay[:100])


@pytest.mark.parametrize("lit", NOT_TAKEN)
def test_literals_left_to_the_other_kernels(lit):
    rx = cx.compile(lit)
    if rx.supported:
        try:
            blob = rx.blob()
        except cx.CoregexError:
            return
        assert emu.literal_shape(blob) == 0, lit


def test_occurrences_at_every_offset_of_a_tile_border(oracle):
    lit = "error"
    rx, o = cx.compile(lit), oracle.Regex(lit)
    for pos in list(range(3830, 3850)) + list(range(3960, 3975)) + list(range(7670, 7690)):
        hay = b"." * pos + lit.encode() + b"." * 300 + lit.encode()
        exp = o.find_all_index(np.frombuffer(hay, dtype=np.uint8))
        for ow, pw in ((60, 2), (60, 1)):
            assert np.array_equal(emu.find_all_literal(rx.blob(), hay, ow, pw), exp), (pos, pw)


def test_synthlog_config1(oracle):
    rx, o = cx.compile("error"), oracle.Regex("error")
    host = cx.synth_pages(1, 0xC0FFEE01, 0, 96)
    assert np.array_equal(emu.find_all_literal(rx.blob(), host, 60, 2), o.find_all_index(host))
