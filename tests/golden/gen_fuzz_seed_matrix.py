"""Golden matrix from the seed corpus of the reference's differential fuzz test (fuzz_stdlib_test.go:31-138: seedPatterns x seedInputs, every
pair added as a seed of FuzzFindAllStdlib, :306-363, which asserts FindAllStringIndex == Go's regexp outside hasUTF8CodepointDifference).  The two
lists are PARSED where they lie (build container only); ASCII patterns and inputs only (the multibyte ones fall under the reference's own
known differences or need Go's rune semantics); expected rows by Go's FindAll loop over Python `re` on bytes (gen_edge_case_pairs.go_find_all)
with `$` written as \\Z (Go's `$` without (?m) is the end of the text; Python's also holds in front of a final newline).
Writes the group "fuzz_seed_matrix" (compact: patterns, inputs, want[pattern][input]) into reference_vectors.json.

    python tests/golden/gen_fuzz_seed_matrix.py
"""
import json, os, re, sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_stdlib_find_tests import go_string, strip_comments
from gen_edge_case_pairs import go_find_all

SRC = "/root/reference/fuzz_stdlib_test.go"


def go_list(text, name):
    body = text[text.index("var " + name + " = []string{"):]
    body = body[body.index("{") + 1:body.index("\n}\n")]
    body = strip_comments(body)
    return [go_string(t) for t in re.findall(r'(`[^`]*`|"(?:[^"\\]|\\.)*")', body)]


def py_pattern(p: bytes) -> bytes:
    out, i = bytearray(), 0
    while i < len(p):                                                 # `$` -> \Z outside escapes and classes
        c = p[i:i + 1]
        if c == b"\\":
            out += p[i:i + 2]; i += 2; continue
        if c == b"[":
            j = p.index(b"]", i + 2)
            out += p[i:j + 1]; i = j + 1; continue
        out += b"\\Z" if c == b"$" else c
        i += 1
    return bytes(out)


def main():
    text = open(SRC, encoding="utf-8").read()
    pats = [p for p in go_list(text, "seedPatterns") if p.isascii() and b"\\p" not in p]
    inps = [s for s in go_list(text, "seedInputs") if s.isascii()]
    want = [[[r[:2] for r in go_find_all(py_pattern(p), s)] for s in inps] for p in pats]
    path = os.path.join(HERE, "reference_vectors.json")
    v = json.load(open(path))
    v["fuzz_seed_matrix"] = {
        "source": "fuzz_stdlib_test.go:31-138 seedPatterns x seedInputs (seeds of FuzzFindAllStdlib, :306-363: FindAllStringIndex == Go regexp); ASCII rows; expected spans by Go's "
                  "FindAll loop over Python re (tests/golden/gen_fuzz_seed_matrix.py)",
        "patterns": [p.decode() for p in pats], "inputs": [s.decode() for s in inps], "want": want}
    # FuzzFindSubmatchStdlib (:369-440): its capture patterns x the same inputs, FindSubmatchIndex == Go regexp outside hasRepeatedCaptureGroupDifference (:198-209)
    body = text[text.index("capturePatterns := []string{"):]
    body = strip_comments(body[body.index("{") + 1:body.index("\n\t}\n")])
    cpats = [go_string(t) for t in re.findall(r'(`[^`]*`|"(?:[^"\\]|\\.)*")', body)]
    rep = text[text.index("repeatedCapturePatterns := map[string]bool{"):]
    rep = [go_string(t) for t in re.findall(r"(`[^`]*`)\s*:\s*true", rep[:rep.index("\n\t}\n")])]
    cpats = [p for p in cpats if p not in rep and p.isascii()]
    first = [[(go_find_all(py_pattern(p), s) or [[]])[0] for s in inps] for p in cpats]
    # FindSubmatchIndex is the single-match API; the FindAll family is this repository's path, and the reference itself lists `(.*)` under
    # "known difference: empty match behavior" for FindAllSubmatchIndex (stdlib_compat_test.go:530-544): rows whose match is empty are left out (null)
    first = [[None if (r and r[0] == r[1]) else r for r in row] for row in first]
    v["fuzz_seed_submatch_first"] = {
        "source": "fuzz_stdlib_test.go:369-440 FuzzFindSubmatchStdlib: capturePatterns (outside hasRepeatedCaptureGroupDifference) x seedInputs, FindSubmatchIndex == Go regexp; "
                  "expected row ([]: no match; null: an empty match, not transcribed) by Python re (tests/golden/gen_fuzz_seed_matrix.py)",
        "patterns": [p.decode() for p in cpats], "inputs": [s.decode() for s in inps], "want": first}
    json.dump(v, open(path, "w"), indent=1)
    print(len(cpats), "capture patterns x", len(inps), "inputs")
    print(len(pats), "patterns x", len(inps), "inputs =", len(pats) * len(inps), "rows")


if __name__ == "__main__":
    main()
