"""Nullable patterns (`a*`, `x?y*`, `\\d*`: SURVEY row a3, meta/findall.go:216-283), CPU tier.  The device program of such a pattern
is its NON-EMPTY variant (program.cc nonEmptyVariant); capi_nullable.hip scanNullable adds an empty match at every position outside the
closed intervals of the variant's rows.  Here: the variant's transducer through its sequential twin, the merge restated in numpy
(`merge_empty_matches`, the specification the two device kernels are tested against in tests/test_gpu_nullable.py), against the
oracle's FindAll loop."""
import random

import numpy as np
import pytest

import coregex_amd as cx
import emu

PATTERNS = [r"a*", r"x?y*", r"\d*", r"[a-z]*", r"(?:ab)*", r"\w*x?", r"(a|b)*c?", r"a*b*", r"(?:a|bc)*", r"\S*", r"a*?", r"(|a)", r"(?:a*)*", r"b*a?b*"]


merge_empty_matches = emu.merge_empty_matches


def variant_rows(rx, hay: np.ndarray) -> np.ndarray:
    img = rx.fsm_image()
    if img is None:                                       # only empty matches (`a*?`)
        return np.zeros((0, 2), dtype=np.int64)
    got = emu.find_all_fsm(img, hay, 3840, 32)
    if isinstance(got, int) and got in (-18, -32):
        got = emu.find_all_fsm(img, hay, 3840, 32, dense=1)
    if isinstance(got, int) and got in (-18, -32):
        got = emu.find_all_fsm(img, hay, 3840, 32, dense=2)
    assert not isinstance(got, int), got
    return got


@pytest.mark.parametrize("pat", PATTERNS)
def test_variant_plus_empty_matches_is_the_reference_loop(pat, oracle):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.supported and rx.strategy == o.strategy, (pat, rx.strategy, o.strategy)
    rng = random.Random(len(pat) * 31 + 7)
    hays = [b"", b"ab", b"a", b"b", b"aab", b"xyy a y", b"abab ab", b"12 345", b"bab"]
    for _ in range(40):
        n = rng.choice([1, 2, 7, 30, 200, 4000, 9000])
        hays.append(bytes(rng.choice(b"ab xy1c\n") for _ in range(n)))
    for hay in hays:
        a = np.frombuffer(hay, dtype=np.uint8)
        exp = o.find_all_index(a)
        got = merge_empty_matches(variant_rows(rx, a), len(hay))
        assert got.tolist() == exp.tolist(), (pat, hay[:60], got[:8].tolist(), exp[:8].tolist())


def test_golden_empty_match_rule(oracle):
    """`a*` on `ab` -> [[0 1] [2 2]], not [[0 1] [1 1] [2 2]] (meta/findall.go:251-257; tests/golden findall_empty_match_rule)."""
    rx = cx.compile(r"a*")
    a = np.frombuffer(b"ab", dtype=np.uint8)
    assert merge_empty_matches(variant_rows(rx, a), 2).tolist() == [[0, 1], [2, 2]] == oracle.Regex(r"a*").find_all_index(a).tolist()


@pytest.mark.parametrize("pat", [r"\b", r"a*\b", r"(?m)^", r"(?m)$|a"])
def test_nullable_with_assertions_is_refused(pat):
    assert not cx.compile(pat).supported


@pytest.mark.parametrize("pat", [r"(a*)", r"(a*)(b)?", r"(\d*)x?", r"(a)*", r"(a|b)*c?", r"(a*)(b*)", r"(x?)(y*)z?", r"((a)|b)*", r"(a?)(b?)(c?)", r"([a-z]*)(\d*)"])
def test_submatch_of_nullable_patterns_through_the_twins(pat, oracle):
    """FindAllSubmatch of a nullable pattern (round 5, capi_nullable.hip scanNullableSubmatch): the rows of FindAllIndex through the twin of the
    pattern's first kernel + merged empty matches, then the backtracking capture twin (device/bt.hpp compiled for the host) for EVERY row
    — the empty ones too — and the reference's end-of-haystack quirk (every group unset, nfa/pikevm.go:2201-2212) on the last row."""
    from twins import rows_on_twin
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.supported and rx.nullable and rx.submatch_supported
    cb = rx.submatch_blobs()[1]
    w = 2 * rx.num_groups
    for hay in (b"", b"a", b"ab", b"xaab aaa b", b"aabbcc xyz 123x", b"bbbaac", b"yyz xz", b"ab12 cd345 x" * 40):
        spans = rows_on_twin(rx, hay)
        assert not isinstance(spans, int)
        got = emu.captures_bt(cb, hay, spans, w)
        if len(got) and got[-1][0] == got[-1][1] == len(hay):
            got[-1][2:] = -1
        exp = o.find_all_submatch_index(hay)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, hay[:40], got[:6].tolist(), exp[:6].tolist())
