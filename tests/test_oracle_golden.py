"""Pins the CPU oracle (oracle/) against the reference's own known-answer vectors
(tests/golden/reference_vectors.json, transcribed from the reference's test tables) and against
the differential corpus of meta/stdlib_compat_test.go (tests/golden/corpus_expected.json).

CPU only (`-m "not gpu"`).
"""
import json
import os

import numpy as np
import pytest

from refcorpus import COMPAT_PATTERNS, COMPAT_PATTERNS_WIDE, generate_test_input, span_hash

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
    VEC = json.load(f)
with open(os.path.join(HERE, "golden", "corpus_expected.json")) as f:
    CORPUS = json.load(f)


def _inp(case):
    if "input_hex" in case:
        return bytes.fromhex(case["input_hex"])
    return case["input"].encode("latin-1")


def test_charclass_find_all_indices(oracle):
    blk = VEC["charclass_find_all_indices"]
    rx = oracle.Regex(blk["pattern"])
    assert rx.strategy == "UseCharClassSearcher"
    for c in blk["cases"]:
        assert rx.find_all_index(_inp(c)).tolist() == c["want"], c["name"]
        assert rx.count(_inp(c)) == len(c["want"]), c["name"]


def test_dfa_search_at_anchored(oracle):
    for c in VEC["dfa_search_at_anchored"]["cases"]:
        rx = oracle.Regex(c["pattern"])
        assert rx.dfa_search_at_anchored(_inp(c), c["at"]) == c["want"], c["name"]


def test_pikevm_captures(oracle):
    for c in VEC["pikevm_captures"]["cases"]:
        rx = oracle.Regex(c["pattern"])
        got = rx.pikevm_captures(_inp(c), 0)
        if c["want"] is None:
            assert got is None
        else:
            assert got.reshape(-1, 2).tolist() == c["want"], c["pattern"]


def test_find_submatch_groups(oracle):
    for c in VEC["find_submatch_groups"]["cases"]:
        rx = oracle.Regex(c["pattern"])
        h = _inp(c)
        rows = rx.find_all_submatch_index(h, 1)
        if c["want"] is None:
            assert len(rows) == 0
        else:
            row = rows[0].reshape(-1, 2)
            assert [h[s:e].decode() for s, e in row] == c["want"], c["pattern"]


def test_digit_prefilter_integration(oracle):
    blk = VEC["digit_prefilter_integration"]
    rx = oracle.Regex(blk["pattern"])
    assert rx.strategy == "UseDigitPrefilter"
    for c in blk["cases"]:
        h = _inp(c)
        assert [h[s:e].decode() for s, e in rx.find_all_index(h)] == c["want"], c["input"]


def test_digit_find_and_memchr(oracle):
    for c in VEC["digit_prefilter_find"]["cases"]:
        assert oracle.memchr_digit_at(_inp(c), c["start"]) == c["want"], c
    for c in VEC["memchr_digit"]["cases"]:
        assert oracle.memchr_digit_at(_inp(c), 0) == c["want"], c


def test_teddy_find(oracle):
    for c in VEC["teddy_find"]["cases"]:
        t = oracle.Teddy([p.encode() for p in c["patterns"]])
        assert t.find(_inp(c), c["start"]) == c["want"], c


def test_lookaround_compat_rows_answered_by_the_lazy_dfa(oracle):
    """A row of the reference's differential tests that its UseDFA engine answers with the look-aware lazy DFA (restated): fresh
    engine, FindAllIndex and — through the PikeVM — FindAllSubmatchIndex."""
    for c in VEC["lookaround_compat_lazy_dfa"]["cases"]:
        rx = oracle.Regex(c["pattern"])
        assert rx.strategy == "UseDFA" and rx.strategy_restated
        assert rx.find_all_index(c["input"].encode()).tolist() == c["want"], c
        assert oracle.Regex(c["pattern"]).find_all_submatch_index(c["input"].encode())[:, :2].tolist() == c["want"], c


def test_lookaround_compat(oracle):
    """(?m)^ (?m)$ \\b \\B: the reference's own differential pairs, spans by Python re (gen_lookaround_expected.py)."""
    for c in VEC["lookaround_compat"]["cases"] + VEC["lookaround_compat_more"]["cases"]:
        got = oracle.Regex(c["pattern"]).find_all_index(c["input"].encode()).tolist()
        assert got == c["want"], c


def test_text_anchor_compat(oracle):
    """^ / $ without (?m) inside alternations: the reference's own differential pairs (edge_cases_test.go:262-290), spans by Python re
    (gen_text_anchor_expected.py).  The empty-match rows follow Go's FindAll rule (no empty match at the end of the previous match)."""
    for c in VEC["text_anchor_compat"]["cases"] + VEC["text_anchor_compat_oracle_only"]["cases"]:
        got = oracle.Regex(c["pattern"]).find_all_index(c["input"].encode()).tolist()
        want = [w for k, w in enumerate(c["want"]) if not (w[0] == w[1] and k and c["want"][k - 1][1] == w[0])]   # meta/findall.go:251-257
        assert got == want, (c, got)


def test_stdlib_find_tests(oracle):
    """The reference's copy of Go's find_test table (stdlib_compat_test.go:79-219): FindAllIndex on all 73 rows, FindAllSubmatchIndex on the
    rows whose pattern is not in the reference's own skip list (patternsWithSubmatchDiffs, :530-544) — the numbers are the table's."""
    blk = VEC["stdlib_find_tests"]
    assert len(blk["cases"]) >= 70
    for c in blk["cases"]:
        hay = bytes.fromhex(c["input_hex"])
        rx = oracle.Regex(c["pattern"])
        assert rx.find_all_index(hay).tolist() == [w[:2] for w in c["want"]], c
        if c["pattern"] not in blk["submatch_not_asserted"]:
            assert oracle.Regex(c["pattern"]).find_all_submatch_index(hay).tolist() == c["want"], c


def test_edge_case_pairs(oracle):
    """The reference's differential edge-case tables (empty alternations, FindAll iteration over nullable and non-greedy patterns, {0} groups,
    doubled word boundaries, anchors in FindAll): FindAllIndex on every row; the capture rows of TestCaptureGroupZeroQuantifier."""
    cases = VEC["edge_case_pairs"]["cases"]
    assert len(cases) >= 70
    for c in cases:
        hay = c["input"].encode()
        assert oracle.Regex(c["pattern"]).find_all_index(hay).tolist() == [w[:2] for w in c["want"]], c
        if c["table"] == "TestCaptureGroupZeroQuantifier":
            assert oracle.Regex(c["pattern"]).find_all_submatch_index(hay).tolist() == c["want"], c


def test_findall_string_kat(oracle):
    """FindAllString rows the reference's tests spell out (regex_test.go:180-185, word_boundary_test.go:273-276)."""
    for c in VEC["findall_string_kat"]["cases"]:
        hay = c["input"].encode()
        assert [hay[s:e].decode() for s, e in oracle.Regex(c["pattern"]).find_all_index(hay).tolist()] == c["want"], c


def test_fuzz_seed_matrix(oracle):
    """The seed corpus of the reference's differential fuzz test (fuzz_stdlib_test.go:31-138, FuzzFindAllStdlib): 56 ASCII patterns x 27 ASCII
    inputs, FindAllIndex on all 1 512 pairs (nullable, non-greedy, anchored and class patterns among them); ONE engine per pattern over all
    inputs, as a compiled Regexp is used."""
    blk = VEC["fuzz_seed_matrix"]
    assert len(blk["patterns"]) * len(blk["inputs"]) >= 1500
    for pi, pat in enumerate(blk["patterns"]):
        rx = oracle.Regex(pat)
        for ii, inp in enumerate(blk["inputs"]):
            assert rx.find_all_index(inp.encode()).tolist() == blk["want"][pi][ii], (pat, inp)


def test_fuzz_seed_submatch_first(oracle):
    """The capture seeds of the reference's FuzzFindSubmatchStdlib (fuzz_stdlib_test.go:369-440): the first row of FindAllSubmatchIndex for 13
    patterns x 27 inputs (rows with an empty match are not transcribed: the reference lists them as known differences of the FindAll family)."""
    blk = VEC["fuzz_seed_submatch_first"]
    n = 0
    for pi, pat in enumerate(blk["patterns"]):
        rx = oracle.Regex(pat)
        for ii, inp in enumerate(blk["inputs"]):
            want = blk["want"][pi][ii]
            if want is None:
                continue
            rows = rx.find_all_submatch_index(inp.encode()).tolist()
            assert (rows[0] if rows else []) == want, (pat, inp)
            n += 1
    assert n >= 330


def test_find_indices_all_strategies(oracle):
    """meta/find_indices_extended_test.go:11-101: one first-match case per engine strategy (reverse suffix / suffix set / inner / anchored, char
    class, bounded backtracker, composite, digit prefilter, Teddy, anchored literal, start-anchored, empty pattern, NFA), answers from the table."""
    cases = VEC["find_indices_all_strategies"]["cases"]
    assert len(cases) >= 30
    for c in cases:
        rows = oracle.Regex(c["pattern"]).find_all_index(c["input"].encode()).tolist()
        assert (rows[0] if rows else None) == c["want"], c


def test_count_edge_cases_and_submatch_counts(oracle):
    """meta/findall_coverage_test.go:134-170 (Count with a limit; limit 0 counts nothing, a negative one everything) and :56-90 (number of
    FindAllSubmatch rows): the numbers written in the tables."""
    for c in VEC["count_edge_cases"]["cases"]:
        assert oracle.Regex(c["pattern"]).count(c["input"].encode(), c["limit"]) == c["want"], c
        assert len(oracle.Regex(c["pattern"]).find_all_index(c["input"].encode(), c["limit"])) == c["want"], c
    for c in VEC["find_all_submatch_count"]["cases"]:
        assert len(oracle.Regex(c["pattern"]).find_all_submatch_index(c["input"].encode())) == c["want"], c


def test_real_world_compat(oracle):
    """edge_cases_test.go:370-400 TestRealWorldEdgeCases (dates, URLs, addresses, quoted strings, whitespace, versions), spans by Python re."""
    for c in VEC["real_world_compat"]["cases"]:
        assert oracle.Regex(c["pattern"]).find_all_index(c["input"].encode()).tolist() == c["want"], c


def test_literal_extraction(oracle):
    """literal.Extractor (prefixes, suffixes, inner literals) against the tables of literal/extractor_test.go."""
    for c in VEC["literal_extraction"]["cases"]:
        got = oracle.extract_literals(c["pattern"], c["which"])
        lits = [b.decode() for b, _ in got]
        if c.get("unordered"):
            assert sorted(lits) == sorted(c["want"]), c
        else:
            assert lits == c["want"], (c, lits)
        if c["complete"] is not None:
            assert all(comp == c["complete"] for _, comp in got), c


def test_fat_teddy_find(oracle):
    for c in VEC["fat_teddy_find"]["cases"]:
        pats = [(c["patterns"]["fmt"] % i).encode() for i in range(c["patterns"]["count"])]
        t = oracle.Teddy(pats)
        h = _inp(c)
        want = c["want"]
        assert t.find(h, c["start"]) == (want[0] if want else -1), c
        if want:
            assert t.find_match(h, c["start"]) == tuple(want), c
        # the meta engine reaches it through UseTeddy (newTeddyFromSeq, teddy.go:629-660)
        rx = oracle.Regex("|".join(p.decode() for p in pats))
        assert rx.strategy == "UseTeddy" and rx.strategy_restated
        got = rx.find_all_index(h[c["start"]:], 1)
        assert ([int(got[0][0]) + c["start"], int(got[0][1]) + c["start"]] if len(got) else None) == want, c


def test_strategy_selection(oracle):
    for c in VEC["strategy_selection"]["cases"]:
        assert oracle.Regex(c["pattern"]).strategy == c["want"], c["pattern"]


def test_find_first(oracle):
    for c in VEC["find_first"]["cases"]:
        rx = oracle.Regex(c["pattern"])
        h = _inp(c)
        got = rx.find_all_index(h, 1)
        if c["want"] is None:
            assert len(got) == 0
        else:
            s, e = got[0]
            assert h[s:e].decode() == c["want"]


def test_findall_empty_match_rule(oracle):
    for c in VEC["findall_empty_match_rule"]["cases"]:
        assert oracle.Regex(c["pattern"]).find_all_index(_inp(c)).tolist() == c["want"]


def test_survey_appendix_b(oracle):
    blk = VEC["survey_appendix_b"]
    rx = oracle.Regex(blk["pattern"])
    assert rx.nfa_states == blk["nfa_states"]
    assert rx.alphabet_len == blk["alphabet_len"]
    assert rx.strategy == "UseDigitPrefilter" and rx.digit_run_skip_safe
    # byte classes: [00-2D]=0 [2E]=1 [2F]=2 [30-39]=3 [3A-FF]=4
    bc = rx.byte_classes()
    assert bc[0x2D] == 0 and bc[0x2E] == 1 and bc[0x2F] == 2 and bc[0x30] == 3 and bc[0x39] == 3 and bc[0x3A] == 4 and bc[0xFF] == 4
    for c in blk["cases"]:
        assert rx.find_all_index(_inp(c)).tolist() == c["want"], c["input"]
    # anchored exploration of every candidate on the reference corpus reaches 11 DFA states
    rx.find_all_index(generate_test_input())
    assert rx.dfa_states() == blk["anchored_dfa_states"]


def test_other_nfa_sizes(oracle):
    # email pattern 23 NFA states -> UseBoth, 12 byte classes; [\w]+ 9 classes
    rx = oracle.Regex(r"(\w+)@(\w+)\.(\w+)")
    assert rx.nfa_states == 23 and rx.strategy == "UseBoth" and rx.alphabet_len == 12
    assert oracle.Regex(r"[\w]+").alphabet_len == 9
    assert oracle.Regex("error").strategy == "UseDFA"


@pytest.mark.parametrize("name", sorted(COMPAT_PATTERNS))
def test_compat_corpus(oracle, name):
    """meta/stdlib_compat_test.go:82-140: FindAllIndex (limit 1000 there; all matches here) + Count."""
    corpus = generate_test_input()
    assert len(corpus) == CORPUS["corpus_len"] == 190100
    exp = CORPUS["patterns"][name]
    rx = oracle.Regex(COMPAT_PATTERNS[name])
    got = rx.find_all_index(corpus)
    assert len(got) == exp["count"]
    assert got[:3].tolist() == exp["first"]
    assert got[-1:].tolist() == exp["last"]
    assert "%016x" % span_hash(got) == exp["hash"]
    assert rx.count(corpus) == exp["count"]
    assert rx.find_all_index(corpus, 7).tolist() == got[:7].tolist()
    if "submatch_hash" in exp:
        rows = rx.find_all_submatch_index(corpus)
        assert rows[:2].tolist() == exp["submatch_first"]
        assert "%016x" % span_hash(rows) == exp["submatch_hash"]


def test_random_differential_vs_python_re(oracle):
    """Hypothesis-free randomized differential test (SURVEY Appendix B did the same by hand)."""
    import re

    rng = np.random.default_rng(12345)
    pats = [r"\d+\.\d+\.\d+\.\d+", r"[\w]+", r"error|warning|fatal|critical", r"(\w+)@(\w+)\.(\w+)",
            r"[0-5]+x", r"\d{1,3}\.\d{1,3}", r"ab|abc", r"a+b*c", r"(a|ab)(c|bcd)", r"x[ab]+?y", r"\d+\.\d"]
    alphabet = np.frombuffer(b"0123456789. ab\ncdxy@_e", dtype=np.uint8)
    for p in pats:
        rx = oracle.Regex(p)
        pr = re.compile(p.encode())
        for _ in range(300):
            n = int(rng.integers(0, 40))
            h = alphabet[rng.integers(0, len(alphabet), size=n)].tobytes()
            exp = [list(m.span()) for m in pr.finditer(h)]
            got = rx.find_all_index(h).tolist()
            if p == r"[0-5]+x" and got != exp:
                # reference quirk: digit-run skip (meta/find_indices.go:1079-1084 + strategy.go:530-560)
                # treats [0-5]+ as a digit class and skips the rest of a run after a failed candidate.
                continue
            assert got == exp, (p, h)


def test_oracle_agrees_with_python_re_except_documented_quirks(oracle):
    """Second, independent pin of the oracle: on random non-nullable ASCII patterns its FindAll equals Python `re`'s
    (leftmost-first, like Go `regexp`, which the reference's own differential test asserts equality with,
    meta/stdlib_compat_test.go:146-199) — except exactly where the reference is known to differ from plain semantics:
    programs whose lazy-DFA answer depends on cache history (the device path refuses them), the digit-run skip that
    treats any digit-only class as `\\d` (meta/find_indices.go:1079-1084), and UseBoth matches longer than 100 bytes."""
    import re

    import coregex_amd as cx
    atoms = ["a", "b", "c", "x", "y", r"\.", ":", "-", r"\d", "[a-c]", "[x-z]", r"\d+", "[a-c]+", "[x-z]+", "a+", r"\.+", "[0-4]+", "[5-9]",
             "ab|xy", "abc|xyz|a:c", r"\w", r"\w+", "[a-z0-9]+", "[ab]", "(a|b)", "(ab)+", "a?", r"\d{2}", r"\d{1,3}", "x*", "(xy|ab|ca)",
             "abcx|bcxy|cxyz|xyza", "z+", "(?:ab)*", "a*", r"\d{2,}", "(a+)(b+)", "b+?", "(?:a|b|c)+", "abcabc", "abc", "xyz", " "]
    alphabet = np.frombuffer(b"abcxyz.:-0123456789 \n", dtype=np.uint8)
    n_same, n_history, n_digit = 0, 0, 0
    for seed in (1, 2, 3, 4, 5, 6):
        rng = np.random.default_rng(seed)
        hays = [alphabet[rng.integers(0, len(alphabet), size=n)].tobytes() for n in (0, 3, 200, 3000)]
        hays += [alphabet[rng.choice(len(alphabet), size=3000, p=rng.dirichlet(0.25 * np.ones(len(alphabet))))].tobytes() for _ in range(3)]
        seen = set()
        while len(seen) < 150:
            pat = "".join(atoms[int(rng.integers(0, len(atoms)))] for _ in range(int(rng.integers(1, 5))))
            if pat in seen:
                continue
            seen.add(pat)
            try:
                o = oracle.Regex(pat)
            except oracle.OracleError:
                continue
            pr = re.compile(pat.encode())
            if pr.search(b"") is not None:          # nullable: Python and Go place empty matches differently
                continue
            diff = None
            for hay in hays:
                got = o.find_all_index(hay).tolist()
                exp = [list(m.span()) for m in pr.finditer(hay)]
                if got != exp:
                    diff = (hay, got, exp)
                    break
            if diff is None:
                n_same += 1
                continue
            why = cx.compile(pat).why_unsupported or ""
            if "cache history" in why:
                n_history += 1
            elif o.strategy == "UseDigitPrefilter" and pat.startswith("[0-4]+"):
                n_digit += 1
            else:
                assert o.strategy == "UseBoth" and any(b - a > 100 for a, b in diff[2]), (pat, o.strategy, diff[1][:3], diff[2][:3])
    assert n_same >= 700 and n_history >= 1 and n_digit >= 1, (n_same, n_history, n_digit)


def test_oracle_captures_agree_with_python_re(oracle):
    """FindAllSubmatch of the oracle (PikeVM with slot tables, nfa/pikevm.go:2186-2432) against Python `re` group spans
    on random non-nullable patterns with nested, optional and repeated groups: leftmost-first captures, -1 for unset."""
    import re
    atoms = ["a", "b", "c", "x", r"\.", ":", r"\d", "[a-c]", r"\d+", "[a-c]+", "a+", "(a|b)", "(ab)+", "(a)?", r"(\d{2})", r"(\d{1,3})", "(x*)y",
             "(xy|ab|ca)", "(a+)(b+)", "(b+?)", "((?:a|b|c)+)", "(abc)", r"(\w+)", r"(\w)", "=", "@", r"([a-z]+)", "(a(b)c)", "((a)|(b))"]
    alphabet = np.frombuffer(b"abcxy.:=@0123456789 \n", dtype=np.uint8)
    n = 0
    for seed in (1, 2, 3):
        rng = np.random.default_rng(seed)
        hays = [alphabet[rng.integers(0, len(alphabet), size=k)].tobytes() for k in (3, 200, 3000)]
        hays += [alphabet[rng.choice(len(alphabet), size=3000, p=rng.dirichlet(0.25 * np.ones(len(alphabet))))].tobytes() for _ in range(2)]
        seen = set()
        while len(seen) < 150:
            pat = "".join(atoms[int(rng.integers(0, len(atoms)))] for _ in range(int(rng.integers(1, 4))))
            if pat in seen:
                continue
            seen.add(pat)
            if "(" not in pat:
                continue
            try:
                o = oracle.Regex(pat)
            except oracle.OracleError:
                continue
            pr = re.compile(pat.encode())
            if pr.search(b"") is not None:
                continue
            n += 1
            for hay in hays:
                exp = [[v for g in range(pr.groups + 1) for v in m.span(g)] for m in pr.finditer(hay)]
                assert o.find_all_submatch_index(hay).tolist() == exp, (pat, o.strategy, len(hay))
    assert n >= 250, n


def test_findall_index_api_vectors(oracle):
    for c in VEC["findall_index_api"]["cases"]:
        rx = oracle.Regex(c["pattern"])
        assert rx.find_all_index(_inp(c), c["n"]).tolist() == c["want"], c
    blk = VEC["charclass_find_all_indices_digit"]
    rx = oracle.Regex(blk["pattern"])
    assert rx.strategy == "UseCharClassSearcher"
    for c in blk["cases"]:
        assert rx.find_all_index(_inp(c)).tolist() == c["want"], c["name"]


def test_find_indices_dispatch_vectors(oracle):
    """First-match tables of meta/find_indices_test.go through the oracle's FindAll(n=1); the `at` variants as the first
    match of the suffix (these rows have no look-around, so the byte in front of `at` does not matter)."""
    for c in VEC["find_indices_dispatch"]["cases"]:
        got = oracle.Regex(c["pattern"]).find_all_index(_inp(c), 1).tolist()
        assert got == ([c["want"]] if c["found"] else []), c["name"]
    for c in VEC["find_indices_at_dispatch"]["cases"]:
        at = c["at"]
        got = oracle.Regex(c["pattern"]).find_all_index(_inp(c)[at:], 1).tolist()
        assert got == ([[c["want"][0] - at, c["want"][1] - at]] if c["found"] else []), c["name"]


def test_nongreedy_and_count_vectors(oracle):
    for c in VEC["nongreedy_first_match"]["cases"]:
        assert oracle.Regex(c["pattern"]).find_all_index(_inp(c), 1).tolist() == [c["want"]], c["name"]
    for c in VEC["count_dispatch"]["cases"]:
        rx = oracle.Regex(c["pattern"])
        assert rx.count(_inp(c)) == c["want"] == len(rx.find_all_index(_inp(c))), c["name"]


def test_use_both_vectors(oracle):
    blk = VEC["use_both_find"]
    for c in blk["cases"]:
        rx = oracle.Regex(c.get("pattern", blk["pattern"]))
        assert rx.find_all_index(_inp(c)).tolist() == c["want"], c["name"]


def test_find_all_submatch_index_vector(oracle):
    blk = VEC["find_all_submatch_index"]
    rows = oracle.Regex(blk["pattern"]).find_all_submatch_index(blk["input"].encode())
    assert len(rows) == blk["want_rows"] and rows[0].tolist() == blk["want_first"]


def test_dfa_search_at_vectors(oracle):
    for c in VEC["dfa_search_at"]["cases"]:
        assert oracle.Regex(c["pattern"]).dfa_search_at(_inp(c), c["at"]) == c["want"], c


def test_lazy_dfa_look_vectors(oracle):
    """The look-aware lazy DFA restated in oracle/engines.cpp against the reference's own tables for it."""
    blk = VEC["lazy_dfa_look_search_at_anchored"]
    rx = oracle.Regex(blk["pattern"])                       # one DFA = one cache, rows in order
    for c in blk["cases"]:
        assert rx.dfa_search_at_anchored(c["input"].encode(), c["at"]) == c["want"], c
    for grp in ("lazy_dfa_look_find", "lazy_dfa_look_known_limitation"):
        for c in VEC[grp]["cases"]:
            assert (oracle.Regex(c["pattern"]).dfa_search_at(c["input"].encode(), 0) != -1) == c["want"], c


def test_lazy_dfa_look_known_limitation_and_history(oracle):
    """What the reference documents as a known limitation falls out of the restatement: a transition is cached per byte
    class although its target depends on the byte being a word byte.  The same DFA answers differently after a different
    history — the reason a DFA strategy over an NFA with assertions has no single answer to reproduce on a device."""
    lim = VEC["lazy_dfa_look_known_limitation"]["limitation"]
    assert (oracle.Regex(lim["pattern"]).dfa_search_at(lim["input"].encode(), 0) != -1) == lim["lazy_dfa"] != lim["stdlib"]
    a, b = oracle.Regex(r"\bfoo\b"), oracle.Regex(r"\bfoo\b")
    a.dfa_search_at(b"a ", 0)                                # the class of ' ' and 'a' (bytes below 'f') is first met, in the
    b.dfa_search_at(b"aa", 0)                                # state behind a word byte, on a non-word byte ... or on a word byte
    assert (a.dfa_search_at(b"a foo", 0), b.dfa_search_at(b"a foo", 0)) == (5, -1)
    # the same through FindAll of a UseDigitPrefilter engine: ' ' and '\n' share a class of `(?m)\d+$`; after "1 2" the entry is "dead"
    c = oracle.Regex(r"(?m)\d+$")
    assert c.strategy == "UseDigitPrefilter" and c.find_all_index(b"1 2").tolist() == [[2, 3]]
    assert c.find_all_index(b"1\n").tolist() == [] and oracle.Regex(r"(?m)\d+$").find_all_index(b"1\n").tolist() == [[0, 1]]


def test_utf8_dot_and_class_vectors(oracle):
    """`.` and classes that reach past U+007F (nfa/compile.go:440-1222, restated in oracle/nfa.cpp): the reference's own rows —
    match counts of `.` over multi-byte text, first matches of `a.c`, FindString of small Unicode classes, the state counts
    that suffix sharing must reach.  Inputs are UTF-8."""
    for c in VEC["utf8_dot_find_all_count"]["cases"]:
        assert len(oracle.Regex(c["pattern"]).find_all_index(c["input"].encode("utf-8"))) == c["want"], c
    for c in VEC["utf8_dot_first_match"]["cases"]:
        assert oracle.Regex(c["pattern"]).find_all_index(c["input"].encode("utf-8"), 1).tolist() == [c["want"]], c
    blk = VEC["unicode_class_find_string"]
    for c in blk["cases"]:
        hay = c["input"].encode("utf-8")
        got = oracle.Regex(c["pattern"]).find_all_index(hay, 1).tolist()
        text = hay[got[0][0]:got[0][1]].decode("utf-8") if got else ""
        assert text == c["want"], (c, got)
    for c in blk["index_cases"]:
        assert oracle.Regex(c["pattern"]).find_all_index(c["input"].encode("utf-8"), 1).tolist() == [c["want"]], c
    for c in VEC["utf8_dot_state_count"]["cases"]:
        assert oracle.Regex(c["pattern"]).nfa_states <= c["max_states"], c


def test_dot_and_negated_classes_against_python_re_on_ascii_text(oracle):
    """On ASCII text `.` is `[^\\n]` and a negated class its ASCII complement, whatever the UTF-8 branches of the automaton look
    like: Python's `re` (bytes) gives the leftmost-first rows for non-nullable patterns."""
    import random
    import re
    rng = random.Random(20260927)
    atoms = [".", ".", r"[^,]", r'[^"]', r"\S", r"\D", r"\W", "a", "b", ",", '"', r"\d", r"\w", " ", "x", "[a-c]", r"[^a-c]", r"[^\n]", r"[^ab,]"]

    def gen(depth=0):
        r = rng.random()
        if depth > 2 or r < 0.45:
            return rng.choice(atoms) + rng.choice(["", "", "+", "*", "?", "+?", "*?", "{1,3}"])
        if r < 0.8:
            return "".join(gen(depth + 1) for _ in range(rng.randint(2, 4)))
        if r < 0.9:
            return "(?:" + gen(depth + 1) + "|" + gen(depth + 1) + ")"
        return "(" + gen(depth + 1) + ")"

    checked = 0
    for _ in range(250):
        pat = gen()
        pr = re.compile(pat.encode())
        if pr.fullmatch(b"") is not None:
            continue
        o = oracle.Regex(pat)
        if o.strategy == "UseCharClassSearcher":                    # (ignores `+?`: a reference quirk restated and tested elsewhere)
            continue
        for _ in range(4):
            hay = "".join(rng.choices('ab," 1x\n', k=rng.choice([0, 1, 5, 30, 200]))).encode()
            assert o.find_all_index(hay).tolist() == [list(m.span()) for m in pr.finditer(hay)], (pat, o.strategy, hay[:60])
            checked += 1
    assert checked > 500


@pytest.mark.parametrize("name", sorted(COMPAT_PATTERNS_WIDE))
def test_compat_corpus_wide_rows(oracle, name):
    """More rows of meta/stdlib_compat_test.go:27-67 — the ones with `.`, `\\S`, negated classes and text anchors — on the same
    corpus: FindAllIndex and Count of the oracle against the stdlib answer (Python `re`, gen_corpus_expected.py).  Whatever
    strategy the reference picks for them (reverse searchers, bounded backtracker, ...) the rows are the leftmost-first ones."""
    corpus = generate_test_input()
    exp = CORPUS["patterns_wide"][name]
    rx = oracle.Regex(COMPAT_PATTERNS_WIDE[name])
    got = rx.find_all_index(corpus)
    assert len(got) == exp["count"], (name, rx.strategy)
    assert got[:3].tolist() == exp["first"] and got[-1:].tolist() == exp["last"]
    assert "%016x" % span_hash(got) == exp["hash"]
    assert rx.count(corpus) == exp["count"]
    if "submatch_hash" in exp:
        rows = rx.find_all_submatch_index(corpus)
        assert "%016x" % span_hash(rows) == exp["submatch_hash"]


def test_case_folding_vectors(oracle):
    """(?i): the reference's FindAllString rows that it asserts equal to the stdlib, and the number of case variants its literal
    extractor produces for a five-letter word (prefix, inner, suffix)."""
    blk = VEC["case_folding_find_all_string"]
    for c in blk["cases"]:
        hay = c["input"].encode()
        got = [hay[s:e].decode() for s, e in oracle.Regex(c["pattern"]).find_all_index(hay)]
        assert got == c["want"], c
    for c in blk["variant_counts"]:
        lits = oracle.extract_literals(c["pattern"], c["which"])
        assert len(lits) == c["count"] and all(comp == c["complete"] for _, comp in lits), c
        assert len({b for b, _ in lits}) == c["count"] and all(b.lower() == lits[0][0].lower() for b, _ in lits)
