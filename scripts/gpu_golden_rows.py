#!/usr/bin/env python3
"""Every FindAll-shaped golden group of tests/golden/reference_vectors.json through the DEVICE (GPU box only; prepared at the end of round 4 for
the first device call of round 5, then to become a test of the GPU tier): the reference's find_test table, the seed matrix of its differential
fuzz test, its edge-case tables and the smaller groups — rows of every served program against the golden rows, counts, and the first capture row
where captures are served.  Prints one line per mismatch and a summary; exit code 1 on any mismatch."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import coregex_amd as cx

vec = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))
bad = served = rows = refused_input = 0


def check(pat, hay, want, what):
    global bad, served, rows, refused_input
    try:
        rx = cx.compile(pat)
    except cx.CoregexError:
        return
    if not rx.supported:
        return
    served += 1
    a = np.frombuffer(hay, dtype=np.uint8)
    try:
        got = rx.find_all_index(a).tolist()
        cnt = rx.count(a)
    except cx.UnsupportedInput:
        refused_input += 1
        return
    rows += 1
    if got != want or cnt != len(want):
        bad += 1
        print("MISMATCH", what, repr(pat), hay[:60], got[:6], want[:6], cnt)


for c in vec["stdlib_find_tests"]["cases"]:
    check(c["pattern"], bytes.fromhex(c["input_hex"]), [w[:2] for w in c["want"]], "stdlib_find_tests")
blk = vec["fuzz_seed_matrix"]
for pi, pat in enumerate(blk["patterns"]):
    for ii, inp in enumerate(blk["inputs"]):
        check(pat, inp.encode(), blk["want"][pi][ii], "fuzz_seed_matrix")
for group in ("edge_case_pairs", "real_world_compat", "text_anchor_compat", "lookaround_compat", "lookaround_compat_more"):
    for c in vec[group]["cases"]:
        check(c["pattern"], c["input"].encode(), [w[:2] for w in c["want"]], group)
for c in vec["find_indices_all_strategies"]["cases"]:
    try:
        rx = cx.compile(c["pattern"])
    except cx.CoregexError:
        continue
    if rx.supported:
        got = rx.find_all_index(np.frombuffer(c["input"].encode(), dtype=np.uint8)).tolist()
        if (got[0] if got else None) != c["want"]:
            bad += 1
            print("MISMATCH find_indices_all_strategies", c, got[:2])
blk = vec["fuzz_seed_submatch_first"]
ncap = 0
for pi, pat in enumerate(blk["patterns"]):
    rx = cx.compile(pat)
    if not rx.submatch_supported:
        continue
    for ii, inp in enumerate(blk["inputs"]):
        want = blk["want"][pi][ii]
        if want is None:
            continue
        try:
            got = rx.find_all_submatch_index(np.frombuffer(inp.encode(), dtype=np.uint8)).tolist()
        except cx.CoregexError as e:
            print("CAPTURE-ERROR", repr(pat), repr(inp), e); bad += 1
            continue
        ncap += 1
        if (got[0] if got else []) != want:
            bad += 1
            print("MISMATCH fuzz_seed_submatch_first", repr(pat), repr(inp), got[:1], want)
print(f"golden rows on the device: {served} program runs served, {rows} compared ({refused_input} haystacks refused with CXG_E_INPUT), {ncap} first capture rows, {bad} bad")
sys.exit(1 if bad else 0)
