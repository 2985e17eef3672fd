"""The reference's own differential tables THROUGH THE DEVICE: every row of the golden groups
transcribed from /root/reference's tests (tests/golden/reference_vectors.json) whose program the device serves is run through the
C ABI (cxg_find_all / cxg_count / cxg_find_all_submatch over host bytes, i.e. what the cgo shim calls) and compared with the golden
rows.  The number of served programs per group is pinned: a front-end change that starts refusing rows fails here instead of
shrinking the comparison silently, and a haystack refused with CXG_E_INPUT counts as a failure (these inputs are tiny).

Reference tables: stdlib_compat_test.go:79-219 (findTests), fuzz_stdlib_test.go:31-138 (seed matrix) and :369-440 (capture seeds),
edge_cases_test.go:52-250 + anchor_test.go:10-48, meta/find_indices_extended_test.go:11-101, regex_test.go:180-185,
word_boundary_test.go:273-276, meta/findall_coverage_test.go:56-90,134-170."""
import json
import os

import numpy as np
import pytest

import coregex_amd as cx

pytestmark = pytest.mark.gpu

VEC = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))


def _compile(pat):
    try:
        return cx.compile(pat)
    except cx.CoregexError:
        return None                                                   # (a syntax the front-end does not parse: refused, never mis-served)


def _spans(rx, hay, n=-1):
    return rx.find_all_index(hay, n).tolist()


def _go_rule(want):
    """rows as Go's FindAll loop keeps them: an empty match at the end of the previous match is dropped (meta/findall.go:251-257)"""
    return [w for k, w in enumerate(want) if not (w[0] == w[1] and k and want[k - 1][1] == w[0])]


def test_stdlib_find_tests_on_the_device():
    blk = VEC["stdlib_find_tests"]
    served = caps = 0
    for c in blk["cases"]:
        rx = _compile(c["pattern"])
        if rx is None:
            continue
        hay = bytes.fromhex(c["input_hex"])
        if rx.supported:
            served += 1
            want = [w[:2] for w in c["want"]]
            assert _spans(rx, hay) == want, (c, "FindAllIndex")
            assert rx.count(hay) == len(want), (c, "Count")
        if rx.num_groups > 1 and rx.submatch_supported and c["pattern"] not in blk["submatch_not_asserted"]:
            caps += 1
            assert rx.find_all_submatch_index(hay).tolist() == c["want"], (c, "FindAllSubmatchIndex")
    assert served == 45 and caps == 7, (served, caps)


def test_fuzz_seed_matrix_on_the_device():
    blk = VEC["fuzz_seed_matrix"]
    served = rows = 0
    for pi, pat in enumerate(blk["patterns"]):
        rx = _compile(pat)
        if rx is None or not rx.supported:
            continue
        served += 1
        for ii, inp in enumerate(blk["inputs"]):
            want = blk["want"][pi][ii]
            assert _spans(rx, inp.encode()) == want, (pat, inp)
            rows += 1
        assert rx.count(blk["inputs"][-1].encode()) == len(blk["want"][pi][-1]), pat
    assert served == 49 and rows == 49 * len(blk["inputs"]), (served, rows)


def test_fuzz_seed_capture_rows_on_the_device():
    blk = VEC["fuzz_seed_submatch_first"]
    served = rows = 0
    for pi, pat in enumerate(blk["patterns"]):
        rx = _compile(pat)
        if rx is None or not rx.submatch_supported:
            continue
        served += 1
        for ii, inp in enumerate(blk["inputs"]):
            want = blk["want"][pi][ii]
            if want is None:
                continue
            got = rx.find_all_submatch_index(inp.encode()).tolist()
            assert (got[0] if got else []) == want, (pat, inp)
            rows += 1
    assert served == 12 and rows == 322, (served, rows)


@pytest.mark.parametrize("group,served_want", [("edge_case_pairs", 75), ("real_world_compat", 7), ("text_anchor_compat", 7), ("text_anchor_compat_oracle_only", 5),
                                               ("lookaround_compat", 16), ("lookaround_compat_more", 6)])
def test_edge_case_groups_on_the_device(group, served_want):
    served = 0
    for c in VEC[group]["cases"]:
        rx = _compile(c["pattern"])
        if rx is None or not rx.supported:
            continue
        served += 1
        hay = c["input"].encode()
        want = [w[:2] for w in c["want"]]
        if group.startswith("text_anchor_compat"):
            want = _go_rule(want)                                     # (spans by Python re: its finditer keeps the adjacent empty match)
        assert _spans(rx, hay) == want, (group, c)
        assert rx.count(hay) == len(want), (group, c)
        if c.get("table") == "TestCaptureGroupZeroQuantifier" and rx.submatch_supported:
            assert rx.find_all_submatch_index(hay).tolist() == c["want"], (group, c)
    assert served == served_want, (group, served)


def test_findall_string_kat_and_case_folding_on_the_device():
    served = 0
    for group in ("findall_string_kat", "case_folding_find_all_string"):
        for c in VEC[group]["cases"]:
            rx = _compile(c["pattern"])
            if rx is None or not rx.supported:
                continue
            served += 1
            hay = c["input"].encode()
            assert [hay[s:e].decode() for s, e in _spans(rx, hay)] == c["want"], (group, c)
    assert served == 10, served


def test_find_indices_all_strategies_on_the_device():
    served = 0
    for c in VEC["find_indices_all_strategies"]["cases"]:
        rx = _compile(c["pattern"])
        if rx is None or not rx.supported:
            continue
        served += 1
        got = _spans(rx, c["input"].encode())
        assert (got[0] if got else None) == c["want"], c
        first = _spans(rx, c["input"].encode(), 1)                    # FindIndices is FindAll's first iteration (meta/find_indices.go)
        assert (first[0] if first else None) == c["want"], c
    assert served == 14, served


def test_count_edge_cases_and_submatch_counts_on_the_device():
    served = 0
    for c in VEC["count_edge_cases"]["cases"]:
        rx = _compile(c["pattern"])
        if rx is None or not rx.supported:
            continue
        served += 1
        hay = c["input"].encode()
        assert rx.count(hay, c["limit"]) == c["want"], c
        assert len(_spans(rx, hay, c["limit"])) == c["want"], c
    assert served == 8, served
    sub = 0
    for c in VEC["find_all_submatch_count"]["cases"]:
        rx = _compile(c["pattern"])
        if rx is None or not rx.submatch_supported:
            continue
        sub += 1
        assert len(rx.find_all_submatch_index(c["input"].encode())) == c["want"], c
    assert sub == 3, sub
