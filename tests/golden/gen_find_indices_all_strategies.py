"""Known-answer rows of meta/find_indices_extended_test.go:11-101 TestFindIndicesAllStrategies: one first-match case per engine strategy with the
expected (start, end, found) written in the table.  PARSED where the table lies (build container only); data only.
Writes the group "find_indices_all_strategies" into reference_vectors.json.

    python tests/golden/gen_find_indices_all_strategies.py
"""
import json, os, re, sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_stdlib_find_tests import go_string, strip_comments

SRC = "/root/reference/meta/find_indices_extended_test.go"


def main():
    text = open(SRC, encoding="utf-8").read()
    body = text[text.index("func TestFindIndicesAllStrategies"):]
    body = strip_comments(body[body.index("}{") + 2:body.index("\n\t}\n")])
    lit = r'(`[^`]*`|"(?:[^"\\]|\\.)*")'
    cases = []
    for m in re.finditer(r"\{\s*" + lit + r"\s*,\s*" + lit + r"\s*,\s*" + lit + r"\s*,\s*(-?\d+)\s*,\s*(-?\d+)\s*,\s*(true|false)\s*\}", body):
        cases.append({"name": go_string(m.group(1)).decode(), "pattern": go_string(m.group(2)).decode(), "input": go_string(m.group(3)).decode(),
                      "want": [int(m.group(4)), int(m.group(5))] if m.group(6) == "true" else None})
    path = os.path.join(HERE, "reference_vectors.json")
    v = json.load(open(path))
    v["find_indices_all_strategies"] = {"source": "meta/find_indices_extended_test.go:11-101 (FindIndices: expected start, end, found as written in the table; null = no match)", "cases": cases}
    json.dump(v, open(path, "w"), indent=1)
    print(len(cases), "rows")


if __name__ == "__main__":
    main()
