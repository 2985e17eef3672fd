"""CPU tier for scan_fields_wave.hip (round 3 headline kernel): its sequential twin (tests/emu/emu_fields.cc — the same
word-by-word steps, carry resolution and start search as the kernel) against the oracle.  `own_words` < 60 puts a tile
border every 64..320 bytes, so windows, halos and ownership are exercised thousands of times per kilobyte."""
import random

import numpy as np
import pytest

import coregex_amd as cx
import emu

PATS = [r"\d+\.\d+\.\d+\.\d+", r"\d+:\d+:\d+", r"\d+\.\d+", r"a+ba+", r"\d+-\d+-\d+"]
ALPHA = {r"\d+\.\d+\.\d+\.\d+": "0123456789..  x\n", r"\d+:\d+:\d+": "0123:: \n", r"\d+\.\d+": "01..x", r"a+ba+": "aab c", r"\d+-\d+-\d+": "0189--/ "}


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8)


@pytest.mark.parametrize("pat", PATS)
def test_shape_is_served(pat):
    rx = cx.compile(pat)
    assert rx.supported and emu.fields_shape(rx.blob()) == pat.count("+")


@pytest.mark.parametrize("pat", [r"\d+\.\d+x?", r"(\w+)@(\w+)\.(\w+)", r"[\w]+\.[\w]+", r"error", r"\d{1,3}\.\d{1,3}", r"\d+\.\w+"])
def test_other_shapes_stay_on_the_other_kernels(pat):
    rx = cx.compile(pat)
    try:
        blob = rx.blob()
    except cx.UnsupportedPattern:
        return
    assert emu.fields_shape(blob) == 0


@pytest.mark.parametrize("pat", PATS)
def test_random_text(pat, oracle):
    rx = cx.compile(pat)
    o = oracle.Regex(pat)
    rng = random.Random(hash(pat) & 0xFFFF)
    alpha = ALPHA[pat]
    served = 0
    for _ in range(150):
        n = rng.choice([5, 40, 64, 65, 127, 128, 129, 200, 700, 4100, 9000])
        kind = rng.random()
        w = ([3, 3, 1] + [1] * len(alpha) if kind < 0.3 else [1] * len(alpha) if kind < 0.6 else [5] + [1] * len(alpha))[: len(alpha)]
        hay = "".join(rng.choices(alpha, weights=w, k=n)).encode()
        exp = o.find_all_index(_u8(hay))
        for ow, pw in ((60, 1), (5, 1), (2, 1), (1, 1), (60, 2), (6, 2), (1, 2)):
            got = emu.find_all_fields(rx.blob(), hay, ow, pw)
            if got is None:
                continue
            served += 1
            assert got.shape == exp.shape and np.array_equal(got, exp), (pat, ow, pw, hay[:120])
    assert served > 900


def test_super_runs_with_many_fields(oracle):
    """`1.2.3.4.5.6.7.8`: groups of K fields from the super-run's start (the kernel's rare loop), also across words and tiles."""
    pat = r"\d+\.\d+\.\d+\.\d+"
    rx, o = cx.compile(pat), oracle.Regex(pat)
    for hay in (b"1.2.3.4.5.6.7.8", b"x 1.2.3.4.5 y", b"1.2.3.4.5.6.7 1.2.3", b"9." * 40 + b"9 tail 1.1.1.1", b"ab" * 29 + b"11.22.33.44.55.66.77.88.99.00.11.22 z",
                b"12.34.56.78." * 30 + b" end", (b"7." * 90) + b"7 " + b"1.2.3.4\n" * 50):
        exp = o.find_all_index(_u8(hay))
        for ow, pw in ((60, 1), (3, 1), (1, 1), (60, 2), (3, 2)):
            got = emu.find_all_fields(rx.blob(), hay, ow, pw)
            assert got is not None and np.array_equal(got, exp), (hay[:40], ow, pw)


def test_long_fields_and_word_filling_runs(oracle):
    """Fields longer than a 64-bit word: propagate lanes (a word of 64 field bytes) and the start search across lanes; a
    match whose start lies more than one word back makes the kernel hand the scan over (None), never a wrong row."""
    pat = r"\d+\.\d+"
    rx, o = cx.compile(pat), oracle.Regex(pat)
    served = 0
    for pre in (0, 1, 63, 64, 65, 100):
        for n1 in (1, 30, 63, 64, 65, 100, 127, 128, 129, 200):
            for n2 in (1, 64, 130):
                hay = b"x" * pre + b"5" * n1 + b"." + b"6" * n2 + b" 1.5 y"
                exp = o.find_all_index(_u8(hay))
                for ow, pw in ((60, 1), (4, 1), (60, 2), (4, 2)):
                    got = emu.find_all_fields(rx.blob(), hay, ow, pw)
                    if got is None:
                        assert n1 + n2 + 1 > 64          # only a match longer than a word may be handed over
                        continue
                    served += 1
                    assert np.array_equal(got, exp), (pre, n1, n2, ow, pw)
    assert served > 120


def test_window_overrun_hands_over(oracle):
    """A super-run that runs past its window (> 192 bytes behind its tile on the device; here: own_words = 60): fallback."""
    pat = r"\d+\.\d+\.\d+\.\d+"
    rx = cx.compile(pat)
    hay = b"y" * 3800 + b"1." * 300 + b"1"
    assert emu.find_all_fields(rx.blob(), hay, 60) is None
    # the same stretch well inside one window is served, in groups of four
    hay2 = b"y" * 100 + b"1." * 300 + b"1"
    got = emu.find_all_fields(rx.blob(), hay2, 60)
    assert np.array_equal(got, oracle.Regex(pat).find_all_index(_u8(hay2)))
    # the persistent kernel's window ends 128 bytes behind its tile: a super-run from the tile's last bytes to 150 bytes behind it is
    # handed over there and served by the grouped kernel's window (192 behind)
    hay3 = b"y" * 3830 + b"1." * 80 + b"1 z"
    assert emu.find_all_fields(rx.blob(), hay3, 60, 2) is None
    got3 = emu.find_all_fields(rx.blob(), hay3, 60, 1)
    assert np.array_equal(got3, oracle.Regex(pat).find_all_index(_u8(hay3)))


def test_edges(oracle):
    pat = r"\d+\.\d+\.\d+\.\d+"
    rx, o = cx.compile(pat), oracle.Regex(pat)
    cases = [b"", b"1", b"1.2.3.4", b".1.2.3.4.", b"1.2.3.", b"1..2.3.4", b"a1.2.3.4\n5.6.7.8", b"11..2.3.4.5 999.1.1.1.",
             b"x" * 59 + b"1.2.3.4", b"x" * 60 + b"10.20.30.40", b"x" * 63 + b"1.2.3.4", b"x" * 64 + b"1.2.3.4", b"x" * 3839 + b"1.2.3.4",
             b"x" * 3840 + b"1.2.3.4", b"x" * 3835 + b"1.2.3.4 5.6.7.8", b"x" * 4090 + b"1.2.3.4", b"x" * 4089 + b"1.2.3.4", b"1.2.3.4" + b"x" * 4089,
             b"\xb1.\xb2.\xb3.\xb4 1.2.3.4", bytes(range(256)) * 3]
    for hay in cases:
        exp = o.find_all_index(_u8(hay))
        for ow, pw in ((60, 1), (1, 1), (60, 2), (1, 2)):
            got = emu.find_all_fields(rx.blob(), hay, ow, pw)
            assert got is not None and got.shape == exp.shape and np.array_equal(got, exp), (hay[:30], ow, pw)


def test_synthlog_pages(oracle):
    pat = r"\d+\.\d+\.\d+\.\d+"
    rx, o = cx.compile(pat), oracle.Regex(pat)
    host = cx.synth_pages(2, 0xC0FFEE02, 0, 64)
    exp = o.find_all_index(host)
    for ow, pw in ((60, 1), (7, 1), (60, 2), (7, 2)):
        got = emu.find_all_fields(rx.blob(), host, ow, pw)
        assert got is not None and np.array_equal(got, exp)
