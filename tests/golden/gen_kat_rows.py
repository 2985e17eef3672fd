"""Known-answer rows the reference's tests spell out themselves (no stand-in): regex_test.go:180-185 TestFindAll (FindAllString with n = -1) and
word_boundary_test.go:273-276 TestWordBoundaryFindAll.  Writes the group "findall_string_kat" into reference_vectors.json.

    python tests/golden/gen_kat_rows.py
"""
import json, os

HERE = os.path.dirname(os.path.abspath(__file__))
ROWS = [  # (pattern, input, want strings)
    (r"\d", "a1b2c3", ["1", "2", "3"]), (r"\d", "abc", []), (r"\w+", "hello world test", ["hello", "world", "test"]),          # regex_test.go:180-184
    (r"\bword\b", "word word word", ["word", "word", "word"]), (r"\b\w+\b", "hello world", ["hello", "world"]),              # word_boundary_test.go:273-276
    (r"\b[0-9]+\b", "test 123 and 456 end", ["123", "456"]), (r"\bthe\b", "the cat and the dog", ["the", "the"]),
]


def main():
    path = os.path.join(HERE, "reference_vectors.json")
    v = json.load(open(path))
    v["findall_string_kat"] = {"source": "regex_test.go:180-185 (FindAllString, n = -1), word_boundary_test.go:273-276 (FindAllString): the expected strings are the tests' own",
                               "cases": [{"pattern": p, "input": s, "want": w} for p, s, w in ROWS]}
    json.dump(v, open(path, "w"), indent=1)
    print(len(ROWS), "rows")


if __name__ == "__main__":
    main()
