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


COUNT = [  # meta/findall_coverage_test.go:141-148 TestCount_EdgeCases: (name, pattern, input, limit, want) as written
    ("empty_pattern_empty_input", "", "", 1, 1), ("empty_pattern", "", "abc", 4, 4), ("limit_zero", r"\w+", "hello world", 0, 0), ("limit_one", r"\w+", "hello world", 1, 1),
    ("limit_exact", r"\w+", "hello world", 2, 2), ("limit_over", r"\w+", "hello world", 10, 2), ("no_match", r"\d+", "no digits", -1, 0), ("greedy_star", "a*", "aaa", -1, 1),
]
SUBMATCH_N = [  # meta/findall_coverage_test.go:63-66 TestFindAllSubmatch_Multi: number of FindAllSubmatch rows (also == Go regexp there)
    ("word_captures", r"(\w+)", "hello world test", 3), ("digit_pairs", r"(\d+)-(\d+)", "1-2 3-4 5-6", 3), ("no_match", r"(\d+)", "no digits", 0), ("single_match", r"^(\w+)", "hello world", 1),
]


def main():
    path = os.path.join(HERE, "reference_vectors.json")
    v = json.load(open(path))
    v["findall_string_kat"] = {"source": "regex_test.go:180-185 (FindAllString, n = -1), word_boundary_test.go:273-276 (FindAllString): the expected strings are the tests' own",
                               "cases": [{"pattern": p, "input": s, "want": w} for p, s, w in ROWS]}
    v["count_edge_cases"] = {"source": "meta/findall_coverage_test.go:134-170 TestCount_EdgeCases (Engine.Count(haystack, limit) as written in the table)",
                             "cases": [{"name": n, "pattern": p, "input": s, "limit": l, "want": w} for n, p, s, l, w in COUNT]}
    v["find_all_submatch_count"] = {"source": "meta/findall_coverage_test.go:56-90 TestFindAllSubmatch_Multi (len(FindAllSubmatch(haystack, -1)) as written in the table)",
                                    "cases": [{"name": n, "pattern": p, "input": s, "want": w} for n, p, s, w in SUBMATCH_N]}
    json.dump(v, open(path, "w"), indent=1)
    print(len(ROWS), "+", len(COUNT), "+", len(SUBMATCH_N), "rows")


if __name__ == "__main__":
    main()
