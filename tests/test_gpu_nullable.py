"""Nullable patterns on the device (meta/findall.go:216-283): the transducer of the non-empty variant + capi_nullable.hip
scanNullable's two kernels, against the oracle's FindAll loop.  The CPU half (variant through the twin, merge restated in numpy)
is tests/test_nullable_cpu.py."""
import random

import numpy as np
import pytest

import coregex_amd as cx
from test_nullable_cpu import PATTERNS

pytestmark = pytest.mark.gpu


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8)


def test_golden_empty_match_rule_through_the_c_abi(oracle):
    """`a*` on `ab` -> [[0 1] [2 2]] (tests/golden/reference_vectors.json findall_empty_match_rule); the empty haystack matches once."""
    rx = cx.compile(r"a*")
    assert rx.find_all_index(_u8(b"ab")).tolist() == [[0, 1], [2, 2]]
    assert rx.find_all_index(_u8(b"")).tolist() == [[0, 0]] and rx.count(_u8(b"")) == 1
    assert rx.find_all_index(_u8(b"baaab")).tolist() == oracle.Regex(r"a*").find_all_index(_u8(b"baaab")).tolist()


@pytest.mark.parametrize("pat", PATTERNS)
def test_rows_and_counts(pat, oracle):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    rng = random.Random(len(pat) * 13 + 1)
    hays = [b"", b"a", b"ab", b"xyy a y", b"12 345"]
    for _ in range(14):
        n = rng.choice([3, 50, 700, 4100, 9000, 70000, 300000])
        w = rng.choice([[1] * 8, [8, 1, 1, 1, 1, 1, 1, 1], [1, 1, 8, 1, 1, 1, 1, 1]])
        hays.append(bytes(rng.choices(b"ab xy1c\n", weights=w, k=n)))
    served = 0
    for hay in hays:
        a = _u8(hay)
        exp = o.find_all_index(a)
        try:
            got = rx.find_all_index(a)
        except cx.UnsupportedInput:                       # per haystack: non-empty matches denser than one per two bytes (`b*a?b*` on `aaaa`) are
            continue                                      # outside the transducer kernel's budgets — CXG_E_INPUT, the caller keeps its CPU loop
        served += 1
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay), hay[:40], got[:6].tolist(), exp[:6].tolist())
        assert rx.count(a) == len(exp)
        for n in (1, 2, 17):
            assert np.array_equal(rx.find_all_index(a, n), exp[:n]), (pat, len(hay), n)
            assert rx.count(a, n) == min(n, len(exp))
    assert served >= len(hays) - 5, (pat, served)


def test_device_resident_haystack_with_base(oracle):
    import torch
    pat = r"x?y*"
    rx, o = cx.compile(pat), oracle.Regex(pat)
    rng = random.Random(3)
    hay = np.frombuffer(bytes(rng.choices(b"xy ab\n", weights=[1, 1, 12, 2, 2, 1], k=2_000_000)), dtype=np.uint8)
    exp = o.find_all_index(hay)
    d = torch.from_numpy(hay.copy()).cuda()
    t = cx.Timing()
    n = rx.find_all_device(d.data_ptr(), hay.size, timing=t)
    assert n == len(exp)
    out = torch.empty((n + 8, 2), dtype=torch.int64, device="cuda")
    n2 = rx.find_all_device(d.data_ptr(), hay.size, out.data_ptr(), n + 8, base=1 << 33, timing=t)
    assert n2 == n and np.array_equal(out[:n].cpu().numpy(), exp + (1 << 33))
    small = torch.empty((100, 2), dtype=torch.int64, device="cuda")
    with pytest.raises(Exception):
        rx.find_all_device(d.data_ptr(), hay.size, small.data_ptr(), 100)


@pytest.mark.parametrize("pat", [r"(a*)", r"(a*)(b)?", r"(\d*)x?", r"(a)*", r"(a|b)*c?", r"(a*)(b*)", r"(x?)(y*)z?", r"((a)|b)*", r"(a?)(b?)(c?)", r"([a-z]*)(\d*)"])
def test_submatch_of_nullable_patterns(pat, oracle):
    """FindAllSubmatchIndex of nullable patterns (round 5; meta/findall.go:390-447): rows of FindAllIndex + the backtracking capture pass for
    every row, empty ones included; the reference's quirk at the end of the haystack (every group unset, nfa/pikevm.go:2201-2212) kept."""
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.supported and rx.nullable and rx.submatch_supported
    for hay in (b"", b"a", b"ab", b"xaab aaa b", b"aabbcc xyz 123x", b"bbbaac", b"yyz xz", b"ab12 cd345 x" * 300, cx.synth_pages(2, 0xC0FFEE02, 0, 2).tobytes()):
        exp = o.find_all_submatch_index(hay)
        got = rx.find_all_submatch_index(hay)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, hay[:40], got[:6].tolist(), exp[:6].tolist())
        assert rx.find_all_submatch_index(hay, 3).tolist() == exp[:3].tolist(), (pat, hay[:40])


def test_submatch_of_a_nullable_pattern_with_assertions_is_refused():
    rx = cx.compile(r"(\b\w*)")
    assert not rx.submatch_supported
