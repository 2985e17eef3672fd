"""Alphabet-run programs (round 4: `(?:25[0-5]|…)\\.…`, `\\d+\\.\\d+x?`, `\\b[0-9]{3}\\b`): patterns over a few ASCII ranges match inside the
maximal runs of those bytes only; scan_runs_wave.hip walks the runs that are long enough with the anchored leftmost-first
automaton (device/runs.hpp, host/lookdfa.cc buildRunsImage) in front of the transducer.  CPU tier: the image through the kernel's
sequential twin against the oracle."""
import random
import struct

import numpy as np
import pytest

import coregex_amd as cx
import emu

README_IP = r"(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)"
RUNS = [(README_IP, b"0123456789..  x\n"), (r"\d+\.\d+x?", b"0189..x y\n"), (r"\b[0-9]{3}\b", b"0123 ab_\n"), (r"a+b|b+a", b"aabb c\n"),
        (r"(?m)^\d+:\d+", b"019::\n\n x"), (r"\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}", b"0123456789..  x\n"), (r"[0-9]+(?:\.[0-9]+)+", b"01..  x"),
        (r"x\d+\b", b"0123x ab_\n"), (r"0x[0-9a-f]+\b", b"0x19afg  _"), (r"\b\d+\.\d+\b", b"019..a _\n")]


def header(img):
    h = struct.unpack_from("<10I", img)
    return dict(nstates=h[2], nsym=h[3], start=h[4:8], min_len=h[8], nr=h[9], lo=list(img[40:40 + h[9]]), hi=list(img[44:44 + h[9]]))


@pytest.mark.parametrize("pat,alpha", RUNS)
def test_twin_equals_oracle(pat, alpha, oracle):
    rx, orc = cx.compile(pat), oracle.Regex(pat)
    assert rx.supported, pat
    img = rx.runs_image()
    assert img is not None, pat
    hd = header(img)
    assert hd["min_len"] >= 1 and 1 <= hd["nr"] <= 4
    rng = random.Random(len(pat))
    hays = [b"", b"1", b"1.1.1.1", b"1.1.1.1.1.1.1.1", b"256.1.1.1", b"1234.1.1.1", b"999", b"1.5x", b"ab", b"ba", b"12:30\n1:2", b"0x1f"]
    for n in [3, 17, 64, 300, 3839, 3841, 7681, 30000]:
        for heavy in (1, 4, 12):
            w = [heavy] * (len(alpha) - 3) + [1, 1, 1]
            hays.append(bytes(rng.choices(alpha, weights=w, k=n)))
    for hay in hays:
        a = np.frombuffer(hay, dtype=np.uint8)
        exp = orc.find_all_index(a)
        for tile in (3840, 64):
            got = emu.find_all_runs(img, a, tile, 1 << 20, 1 << 20)
            if isinstance(got, int):
                assert got == -17, (pat, got)                       # a run longer than 255 bytes: the kernel gives up, the transducer takes over
                continue
            assert np.array_equal(got, exp), (pat, len(hay), tile, hay[:60], got[:4].tolist(), exp[:4].tolist())


def test_long_runs_and_crowded_tiles_give_up(oracle):
    rx = cx.compile(README_IP)
    img = rx.runs_image()
    assert emu.find_all_runs(img, b"x" + b"1" * 300 + b"x") == -17
    assert isinstance(emu.find_all_runs(img, b"x" + b"1" * 255 + b"x"), np.ndarray)
    crowded = b"1.1.1.1 " * 400
    assert emu.find_all_runs(img, crowded, 3840, 256, 4) == -18
    assert emu.find_all_runs(img, b" " + b"1.1.1.1." * 6 + b" ", 3840, 256, 4) == -20
    ok = emu.find_all_runs(img, crowded, 3840, 1024, 4)
    assert np.array_equal(ok, oracle.Regex(README_IP).find_all_index(np.frombuffer(crowded, dtype=np.uint8)))


@pytest.mark.parametrize("pat", [r"\w+@\w+\.com", r"[^,]+x", r"a.b", r"é+x", r"(?i)(error|fail|warn)x\d", r"\d*", r"\Afoo|bar"])
def test_programs_outside_the_shape_have_no_image(pat):
    rx = cx.compile(pat)
    assert not rx.supported or rx.runs_image() is None, pat
