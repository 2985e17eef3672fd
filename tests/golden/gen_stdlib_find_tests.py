"""Golden rows from the reference's copy of Go's find_test table (stdlib_compat_test.go:79-219 `findTests`): pattern, text and the expected
FindAllSubmatchIndex rows written out in the table itself (`build(n, ...)`; nil = no match).  The reference asserts FindAllStringIndex ==
Go's regexp on every row (TestStdlibCompat_FindAllIndex, :407-422) and FindAllSubmatchIndex on the rows outside hasSubmatchDifference
(:616-660); the table's own numbers are Go's answers.  This script PARSES the table where it lies (run in the build container, where
/root/reference exists) and writes the group "stdlib_find_tests" into reference_vectors.json — data only: patterns, texts, index rows.

    python tests/golden/gen_stdlib_find_tests.py
"""
import json, os, re, sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/stdlib_compat_test.go"


def go_string(tok):
    """A Go string literal -> bytes (UTF-8)."""
    if tok[0] == "`":
        return tok[1:-1].encode()
    out, i, s = bytearray(), 0, tok[1:-1]
    esc = {"a": 7, "b": 8, "f": 12, "n": 10, "r": 13, "t": 9, "v": 11, "\\": 92, '"': 34, "'": 39, "`": 96}
    while i < len(s):
        c = s[i]
        if c != "\\":
            out += c.encode(); i += 1; continue
        n = s[i + 1]
        if n == "u":
            out += chr(int(s[i + 2:i + 6], 16)).encode(); i += 6
        elif n == "x":
            out.append(int(s[i + 2:i + 4], 16)); i += 4
        else:
            out.append(esc[n]); i += 2
    return bytes(out)


def strip_comments(body):
    """Go line comments removed — outside string literals only (`https?://` is a pattern, not a comment)."""
    out, i = [], 0
    while i < len(body):
        c = body[i]
        if c == "`":
            j = body.index("`", i + 1)
            out.append(body[i:j + 1]); i = j + 1
        elif c == '"':
            j = i + 1
            while body[j] != '"':
                j += 2 if body[j] == "\\" else 1
            out.append(body[i:j + 1]); i = j + 1
        elif body.startswith("//", i):
            j = body.find("\n", i)
            i = len(body) if j < 0 else j
        else:
            out.append(c); i += 1
    return "".join(out)


def main():
    text = open(SRC, encoding="utf-8").read()
    body = text[text.index("var findTests = []FindTest{"):]
    body = body[body.index("{") + 1:body.index("\n}\n")]
    body = strip_comments(body)                                       # (the KNOWN DIFFERENCE rows are commented out in the table)
    lit = r'(`[^`]*`|"(?:[^"\\]|\\.)*")'
    rows = []
    for m in re.finditer(r"\{\s*" + lit + r"\s*,\s*" + lit + r"\s*,\s*(nil|build\(([^)]*)\))\s*,?\s*\}", body, re.S):
        pat, txt = go_string(m.group(1)), go_string(m.group(2))
        if m.group(3) == "nil":
            want = []
        else:
            nums = [int(x) for x in re.findall(r"-?\d+", m.group(4))]
            n, vals = nums[0], nums[1:]
            w = len(vals) // n
            want = [vals[k * w:(k + 1) * w] for k in range(n)]
        rows.append({"pattern": pat.decode("utf-8"), "input_hex": txt.hex(), "want": want})
    skip = text[text.index("var patternsWithSubmatchDiffs = map[string]bool{"):]
    skip = [go_string(t).decode() for t in re.findall(r"^\s*(`[^`]*`)\s*:\s*true", skip[:skip.index("\n}\n")], re.M)]   # :530-544
    path = os.path.join(HERE, "reference_vectors.json")
    v = json.load(open(path))
    v["stdlib_find_tests"] = {
        "source": "stdlib_compat_test.go:79-219 findTests (pattern, text, FindAllSubmatchIndex rows as written in the table = Go's answers; the reference asserts "
                  "FindAllStringIndex on every row, :407-422, and FindAllSubmatchIndex outside hasSubmatchDifference, :616-660); parsed by tests/golden/gen_stdlib_find_tests.py",
        "submatch_not_asserted": skip,               # patternsWithSubmatchDiffs: the reference skips FindAllSubmatchIndex for these
        "cases": rows}
    json.dump(v, open(path, "w"), indent=1)
    print(len(rows), "rows;", len(skip), "patterns whose submatch rows the reference does not assert")


if __name__ == "__main__":
    sys.exit(main())
