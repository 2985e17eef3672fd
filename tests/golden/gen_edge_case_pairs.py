"""Golden rows from the reference's differential edge-case tables (edge_cases_test.go: TestEmptyMatchPatterns :52-86, TestFindAllIterationSemantics
:137-168, TestCaptureGroupZeroQuantifier :174-208, TestWordBoundaryCornerCases :214-250; anchor_test.go:10-48 TestAnchorInFindAll; stdlib_compat_test.go:1429-1582 PerlFlags, GreedyVsNonGreedy, Repetition, Count): the
(pattern, input) pairs are PARSED where they lie (build container only), the expected FindAllStringIndex / FindAllStringSubmatchIndex rows —
the reference asserts equality with Go's regexp — are computed by Go's FindAll loop (regexp.go allMatches: an empty match right behind the
previous match is dropped, an empty match advances the search by one byte) over Python `re` on bytes: ASCII inputs, leftmost-first in both,
`^` `$` `\\b` with the real neighbours of the search position in both, no input ends in a newline.  Writes the group "edge_case_pairs".

    python tests/golden/gen_edge_case_pairs.py
"""
import json, os, re, sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_stdlib_find_tests import go_string, strip_comments

TABLES = [("/root/reference/edge_cases_test.go", "func TestEmptyMatchPatterns"), ("/root/reference/edge_cases_test.go", "func TestFindAllIterationSemantics"),
          ("/root/reference/edge_cases_test.go", "func TestCaptureGroupZeroQuantifier"), ("/root/reference/edge_cases_test.go", "func TestWordBoundaryCornerCases"),
          ("/root/reference/anchor_test.go", "func TestAnchorInFindAll"),
          # stdlib_compat_test.go: FindAllString == Go regexp (the row its own known-difference map skips is left out below)
          ("/root/reference/stdlib_compat_test.go", "func TestStdlibCompat_PerlFlags"), ("/root/reference/stdlib_compat_test.go", "func TestStdlibCompat_GreedyVsNonGreedy"),
          ("/root/reference/stdlib_compat_test.go", "func TestStdlibCompat_Repetition"), ("/root/reference/stdlib_compat_test.go", "func TestStdlibCompat_Count")]
SKIP = {"(?im)^HELLO"}   # perlFlagsKnownDiffs, stdlib_compat_test.go:1432-1434


def go_find_all(pat: bytes, hay: bytes):
    """regexp.(*Regexp).allMatches over Python's leftmost-first search: rows of 2 * (groups + 1) indices, -1 for a group that did not take part."""
    rx = re.compile(pat)
    rows, pos, prev_end = [], 0, -1
    while pos <= len(hay):
        m = rx.search(hay, pos)
        if m is None:
            break
        accept = True
        if m.end() == pos and m.start() == m.end():                   # an empty match at the search position
            if m.start() == prev_end:
                accept = False
            nxt = pos + 1
        elif m.end() == m.start():                                    # an empty match further on
            nxt = m.end() + 1 if m.end() == pos else m.end()
        else:
            nxt = m.end()
        if m.end() == m.start() and m.end() > pos:                    # Go: `if matches[1] == pos` only — an empty match behind pos moves pos to its end
            nxt = m.end()
        prev_end = m.end()
        if accept:
            row = [m.start(), m.end()]
            for g in range(1, rx.groups + 1):
                row += [m.start(g), m.end(g)]
            rows.append(row)
        if nxt == pos:
            nxt = pos + 1
        pos = nxt
    return rows


def main():
    lit = r'(`[^`]*`|"(?:[^"\\]|\\.)*")'
    cases = []
    for path, func in TABLES:
        text = open(path, encoding="utf-8").read()
        body = text[text.index(func):]
        body = body[body.index("}{"):]
        body = body[2:body.index("\n\t}\n")]
        body = strip_comments(body)
        for m in re.finditer(r"\{\s*" + lit + r"\s*,\s*" + lit + r"\s*\}", body):
            pat, inp = go_string(m.group(1)), go_string(m.group(2))
            if pat.decode() in SKIP:
                continue
            try:
                re.compile(pat)
            except re.error:
                continue                                              # (`(?-s)` at the start of a pattern: no Python spelling)
            if b"\\B" in pat and inp == b"":
                continue                                              # (Python's \B never holds in an empty string; Go's does: three rows without a stand-in)
            cases.append({"table": func.split()[1], "pattern": pat.decode(), "input": inp.decode(), "want": go_find_all(pat, inp)})
    path = os.path.join(HERE, "reference_vectors.json")
    v = json.load(open(path))
    v["edge_case_pairs"] = {
        "source": "edge_cases_test.go:52-86,137-168,174-208,214-250 and anchor_test.go:10-48 (pairs the reference compares with Go's regexp: FindAllStringIndex, and "
                  "FindAllStringSubmatchIndex for TestCaptureGroupZeroQuantifier); expected rows by Go's FindAll loop over Python re (tests/golden/gen_edge_case_pairs.py)",
        "cases": cases}
    json.dump(v, open(path, "w"), indent=1)
    print(len(cases), "rows")


if __name__ == "__main__":
    main()
