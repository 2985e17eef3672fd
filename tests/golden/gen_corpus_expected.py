"""Generates tests/golden/corpus_expected.json with Python `re` in bytes mode.

Secondary differential oracle: for these non-nullable ASCII patterns Python's
leftmost-first semantics coincide with Go's stdlib `regexp`, which is what the reference's
TestStdlibCompatibility asserts coregex equals (meta/stdlib_compat_test.go:82-140).
Run from the repo root:  python tests/golden/gen_corpus_expected.py
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refcorpus import COMPAT_PATTERNS, COMPAT_PATTERNS_WIDE, generate_test_input, span_hash  # noqa: E402

corpus = generate_test_input()
out = {"_generator": "tests/golden/gen_corpus_expected.py (python re, bytes mode)", "corpus_len": len(corpus), "patterns": {}}
out["patterns_wide"] = {}
for name, pat, key in [(n, p, "patterns") for n, p in COMPAT_PATTERNS.items()] + [(n, p, "patterns_wide") for n, p in COMPAT_PATTERNS_WIDE.items()]:
    rx = re.compile(pat.encode())
    groups = rx.groups
    spans = [list(m.span()) for m in rx.finditer(corpus)]
    entry = {"pattern": pat, "count": len(spans), "first": spans[:3], "last": spans[-1:], "hash": "%016x" % span_hash(spans)}
    if groups:
        rows = [[x for g in range(groups + 1) for x in m.span(g)] for m in rx.finditer(corpus)]
        entry["submatch_first"] = rows[:2]
        entry["submatch_hash"] = "%016x" % span_hash(rows)
    out[key][name] = entry
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "corpus_expected.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path)
