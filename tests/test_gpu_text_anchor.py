"""Text-start anchors on the device: `k_scan_fsm` started in the text-start state, its reverse walks accepting position 0
by the per-state flag — in the window, and in the epilogue for a first match that is longer than the window's reach.  Round 6:
end-of-text anchors (`a$|z`, `x\\z|foo`) — the step over the haystack's last byte takes the column of the kind no byte has (k_scan_fsm<LOOK = 2>)."""
import random

import numpy as np
import pytest

import coregex_amd as cx
from routing import routed
from test_text_anchor_cpu import TEXT, TEXT_END

pytestmark = pytest.mark.gpu


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8)


@pytest.mark.parametrize("pat", TEXT)
def test_rows_and_captures(pat, oracle):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    rng = random.Random(len(pat) * 17)
    toks = [b"foo", b"bar", b"12", b",", b" ", b"error", b"abc", b"x7", b"ab", b"GET", b"POST", b"k=1", b"\n", b":", b"z", b"_", b"pad pad pad pad "]
    hays = [b"bar", b"foo", b"bar foo bar", b"12,12 12", b" error", b"zabc", b"x7", b"xab", b"GET /a POST", b"k=1 k=2", b"\nbar", b"a" * 300 + b",b"]
    for n in (3839, 3841, 61441, 400000, 2_000_000):
        for lead in (b"", b"bar", b"12", b"abc", b"x9", b"GET", b"k=2", b"ab", b"error"):
            hays.append(lead + b"".join(rng.choice(toks) for _ in range(n // 5)))
    for hay in hays:
        a = _u8(hay)
        exp = o.find_all_index(a)
        try:
            got = rx.find_all_index(a)
        except cx.UnsupportedInput:
            continue                                                    # (a dense random haystack past the transducer's budgets)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay), hay[:40], got[:4].tolist(), exp[:4].tolist())
        assert rx.count(a) == len(exp) and np.array_equal(rx.find_all_index(a, 2), exp[:2])
        if rx.submatch_supported and rx.num_groups > 1 and len(hay) < 500000:
            es = o.find_all_submatch_index(a)
            gs = rx.find_all_submatch_index(a)
            assert gs.shape == es.shape and np.array_equal(gs, es), (pat, len(hay), hay[:40], "submatch")


@pytest.mark.parametrize("pat", TEXT_END)
def test_end_of_text_at_every_edge(pat, oracle):
    """The haystack's last byte at every distance from a chunk, tile and group edge (the tile in front sees the end in its window's tail;
    a group's last tile; the launch's only tile), each tail that may or may not match there; rows, count, limit and `base`."""
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.supported and rx.strategy == o.strategy, (pat, rx.why_unsupported)
    rng = random.Random(len(pat) * 29)
    toks = [b"foo", b"bar", b"12", b",", b" ", b"error", b"abc", b"x7", b"ab", b"k=1", b"\n", b"z", b"a", b"warn", b"pad pad pad "]
    tails = (b"a", b"foo", b",12", b"ab", b"k=12", b"x", b"\n", b"bar", b",abc", b"error", b"za", b"warning", b"foo ")
    hays = [b"a", b"za", b"ayyyyy", b"ab", b"b", b"xfoo", b"foo x foo", b",12", b"k=1 k=12", b"foo\n", b"error", b"warn error"]
    for n in (31, 64, 65, 3776, 3839, 3840, 3841, 3904, 4032, 4033, 7680, 7681, 122880, 122881, 122880 + 3840, 491520, 491521, 1 << 20):
        body = b"".join(rng.choice(toks) for _ in range(n // 2 + 8))[:n]
        for tail in tails:
            hays.append(body[: n - len(tail)] + tail)
    t = cx.Timing()
    for hay in hays:
        a = _u8(hay)
        exp = o.find_all_index(a)
        try:
            got = rx.find_all_index(a)
        except cx.UnsupportedInput:
            continue
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay), hay[-24:], got[-3:].tolist(), exp[-3:].tolist())
        assert rx.count(a) == len(exp)
    import torch
    hay = _u8(hays[-3])
    exp = o.find_all_index(hay)
    d = torch.from_numpy(hay.copy()).cuda()
    out = torch.empty((len(exp) + 8, 2), dtype=torch.int64, device="cuda")
    n = rx.find_all_device(d.data_ptr(), hay.size, out.data_ptr(), len(exp) + 8, base=5000, timing=t)
    assert n == len(exp) and np.array_equal(out[:n].cpu().numpy(), exp + 5000)
    routed(10 in t.kernels or 20 in t.kernels, t.kernels)


def test_long_first_match(oracle):
    """The match that starts at the text start and runs on: found while it ends within the transducer's reach behind its tile (190 bytes);
    past that the call is refused for the haystack (CXG_E_INPUT: a look-around program has no table-walking kernel behind the transducer)."""
    for pat in (r"(?:^|,)[a-c]+", r"(?:^|x)[a-c]+"):
        rx, o = cx.compile(pat), oracle.Regex(pat)
        for n in (100, 3000, 3810, 3900, 3990, 9000, 70000):
            for tail in (b"", b",ab ,c", b" x,a"):
                a = _u8(b"abc" * (n // 3) + tail)
                exp = o.find_all_index(a)
                assert exp[0].tolist() == [0, 3 * (n // 3)]
                try:
                    assert np.array_equal(rx.find_all_index(a), exp), (pat, n, tail)
                except cx.UnsupportedInput:
                    assert n > 3840 + 150, (pat, n, tail)
                b = _u8(b"z" + bytes(a))                               # the same behind another byte: the anchor does not hold
                try:
                    assert np.array_equal(rx.find_all_index(b), o.find_all_index(b)), (pat, n, tail, "shifted")
                except cx.UnsupportedInput:
                    assert n > 3840 + 150, (pat, n, tail)


def test_each_call_is_one_text(oracle):
    """`base` moves the rows, not the text: the anchor holds at the first byte of every call's haystack."""
    import torch
    pat = r"(?:^|,)\d+"
    rx, o = cx.compile(pat), oracle.Regex(pat)
    hay = _u8(b"12 34,56 78\n")
    exp = o.find_all_index(hay)
    d = torch.from_numpy(hay.copy()).cuda()
    out = torch.empty((8, 2), dtype=torch.int64, device="cuda")
    t = cx.Timing()
    n = rx.find_all_device(d.data_ptr(), hay.size, out.data_ptr(), 8, base=1000, timing=t)
    assert n == len(exp) and np.array_equal(out[:n].cpu().numpy(), exp + 1000)
    routed(t.kernels in ([10], [20]), t.kernels)


def test_reference_pairs_on_the_device(oracle):
    """tests/golden "text_anchor_compat" (edge_cases_test.go:262-290,320; spans by Python re) through the device."""
    import json, os
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))
    served = 0
    for c in vec["text_anchor_compat"]["cases"]:
        rx = cx.compile(c["pattern"])
        if not rx.supported:
            continue
        served += 1
        assert rx.find_all_index(_u8(c["input"].encode())).tolist() == c["want"], c
    assert served >= 5
    served_end = 0
    for c in vec["text_anchor_compat_oracle_only"]["cases"]:            # `a$|z`, `(a$)b$`, `^a$|^b$` (round 6); the others are reverse / anchored strategies or nullable
        rx = cx.compile(c["pattern"])
        if not rx.supported or not c["input"]:
            continue
        served_end += 1
        assert rx.find_all_index(_u8(c["input"].encode())).tolist() == c["want"], c
    assert served_end == 5, served_end
