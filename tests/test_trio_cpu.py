"""CPU tier for k_scan_trio_wave (scan_fields_wave.hip, round 3): its sequential twin (tests/emu/emu_fields.cc — the same
bitmaps, additions with the kernel's carry resolution, the loop for matches that share a run, the three bit scans per row) against
the oracle: spans from (start, end), capture rows from all four positions.  `own_words` < 60 puts a tile border every 64..320
bytes."""
import random

import numpy as np
import pytest

import coregex_amd as cx
import emu

PATS = [r"(\w+)@(\w+)\.(\w+)", r"([a-c]+)x([a-c]+)y([a-c]+)", r"(\w+)=(\w+);(\w+)", r"(\d+)/(\d+) (\d+)", r"\d+-\d+:\d+",
        r"(\w+)=(\w+)", r"(\d+):(\d+)", r"(\d+)-(\d+):(\d+)/(\d+)", r"(\w+)@(\w+)",
        r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"(\w+)@(\w+)@(\w+)", r"\d+\.\d+\.\d+", r"([a-c]+)-([a-c]+)-([a-c]+)-([a-c]+)"]   # one separator: K | 8
ALPHA = {PATS[0]: "ab_9@@..  x\n", PATS[1]: "abcxy z", PATS[2]: "ab_1==;; \n", PATS[3]: "0189// x", PATS[4]: "0123--:: \n",
         PATS[5]: "ab_1== \n", PATS[6]: "0123:: x", PATS[7]: "019--::// x", PATS[8]: "ab_9@@ x",
         PATS[9]: "0189.. x", PATS[10]: "ab_9@@ x", PATS[11]: "0189.. x\n", PATS[12]: "abc-- x"}


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8)


def _expected(o, rx, hay):
    """(start, LA link, LB link, end) rows from the oracle: captures give the links (group 1 ends on the first, group 2 on the second)."""
    a = _u8(hay)
    if rx.num_groups > 1:
        sub = o.find_all_submatch_index(a)
        k = rx.num_groups - 1                                   # fields; group i ends on link i
        cols = [sub[:, 0]] + [sub[:, 2 * i + 1] for i in range(1, k)] + [sub[:, 1]]
        return np.stack(cols, axis=1) if len(sub) else np.zeros((0, k + 1), dtype=np.int64)
    return o.find_all_index(a)


@pytest.mark.parametrize("pat", PATS)
def test_shape_is_served(pat):
    rx = cx.compile(pat)
    shape = emu.trio_shape(rx.blob())
    assert rx.supported and (shape & 7) == pat.count("+")
    seps = set(pat.replace("\\d+", "").replace("\\w+", "").replace("[a-c]+", "").replace("(", "").replace(")", "").replace("\\", ""))
    assert bool(shape & 8) == (pat.count("+") >= 3 and len(seps) == 1)


@pytest.mark.parametrize("pat", [r"(\d+)-(\d+)-(\d+) (\d+)", r"(\w+)@(\w+)\.(\w+)@(\w+)", r"(\w+)@(\w+)\.(\w+)x", r"error", r"(\w+)w(\w+)\.(\w+)"])
def test_other_shapes_stay_on_the_other_kernels(pat):
    rx = cx.compile(pat)
    try:
        blob = rx.blob()
    except cx.UnsupportedPattern:
        return
    assert not emu.trio_shape(blob)


@pytest.mark.parametrize("pat", PATS)
def test_random_text(pat, oracle):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    rng = random.Random(hash(pat) & 0xFFFF)
    alpha = ALPHA[pat]
    served = 0
    for _ in range(120):
        n = rng.choice([5, 40, 64, 65, 127, 128, 129, 200, 700, 4100, 9000])
        kind = rng.random()
        w = ([3, 3, 1] + [1] * len(alpha) if kind < 0.3 else [1] * len(alpha) if kind < 0.6 else [1] * (len(alpha) - 3) + [6, 6, 6])[: len(alpha)]
        hay = "".join(rng.choices(alpha, weights=w, k=n)).encode()
        exp = _expected(o, rx, hay)
        for ow in (60, 5, 2, 1):
            got = emu.find_all_trio(rx.blob(), hay, ow)
            if got is None:
                continue
            served += 1
            cmp = got if rx.num_groups > 1 else got[:, [0, -1]]
            assert cmp.shape == exp.shape and np.array_equal(cmp, exp), (pat, ow, hay[:120])
    assert served > 350


def test_matches_that_share_a_run(oracle):
    """`a@b.c@d.e`: the second candidate begins with the third run of the first match and is dropped; chains of any length."""
    pat = PATS[0]
    rx, o = cx.compile(pat), oracle.Regex(pat)
    for hay in (b"a@b.c@d.e", b"a@b.c@d.e@f.g", b"a@b.c@d.e@f.g@h.i x@y.z", b"ab@cd.ef@" * 40 + b"gh.ij", b"x a@b.c.d y a@b@c.d", b"q@r.s@t.u " * 30,
                b"a.b@c.d@e.f@g", b"aa@bb.cc@dd.ee@ff.gg@hh.ii@jj.kk@ll.mm"):
        exp = _expected(o, rx, hay)
        for ow in (60, 3, 1):
            got = emu.find_all_trio(rx.blob(), hay, ow)
            assert got is not None and np.array_equal(got, exp), (hay[:40], ow)


def test_edges_and_handover(oracle):
    pat = PATS[0]
    rx, o = cx.compile(pat), oracle.Regex(pat)
    for hay in (b"", b"a", b"a@b.c", b"a@b", b"@b.c", b"a@@b.c", b"a@b..c", b"a.b@c.d", b"_9@Z_.q0 ", b"\xe9a@b.c\xff", bytes(range(256)) * 2,
                b"-" * 59 + b"bob@site.org", b"x " * 30 + b"bob@site.org", b" " * 3839 + b"a@b.c", b" " * 3835 + b"ab@cd.ef gh@ij.kl"):
        exp = _expected(o, rx, hay)
        for ow in (60, 1):
            got = emu.find_all_trio(rx.blob(), hay, ow)
            assert got is not None and got.shape == exp.shape and np.array_equal(got, exp), (hay[:30], ow)
    # a match whose start lies more than one word back hands the scan over (never a wrong row)
    served = 0
    for n1 in (5, 40, 63, 64, 65, 90, 130):
        for n2 in (1, 30, 64, 70):
            hay = b"x " + b"u" * n1 + b"@" + b"h" * n2 + b".org y bob@site.org"
            got = emu.find_all_trio(rx.blob(), hay, 60)
            if got is None:
                assert n1 + n2 + 5 > 64
                continue
            served += 1
            assert np.array_equal(got, _expected(o, rx, hay))
    assert served >= 8
    assert emu.find_all_trio(rx.blob(), b"y" * 3800 + b"@".join([b"ab"] * 300) + b".c", 60) is None     # a super-run past its window


def test_synthlog_pages(oracle):
    pat = PATS[0]
    rx, o = cx.compile(pat), oracle.Regex(pat)
    host = cx.synth_pages(5, 0xC0FFEE05, 0, 48)
    exp = _expected(o, rx, host)
    for ow in (60, 7):
        got = emu.find_all_trio(rx.blob(), host, ow)
        assert got is not None and np.array_equal(got, exp)
