#!/usr/bin/env python3
"""CPU fuzz of the k_scan_trio_wave twin (tests/emu/emu_fields.cc emu_find_all_trio) against the oracle: random programs of the
family run(F) (byte(c_i) run(F)){K-1} with and without groups, random few-symbol haystacks, several tile sizes.
Usage: python scripts/cpu_fuzz_trio.py [seed] [n_patterns]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import coregex_amd as cx, emu
from oracle import oracle

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
npat = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = random.Random(seed)
CLASSES = [(r"\w", "ab_9Z"), (r"\d", "0189"), ("[a-c]", "abc"), ("[a-z0-9]", "az09"), ("[x-z]", "xyz"), ("a", "a")]
SEPS = ["@", ".", "=", ";", ":", "-", "/", " ", ","]
served = checked = bad = refused = 0
for it in range(npat):
    cls, members = rng.choice(CLASSES)
    K = rng.choice([2, 2, 3, 3, 3, 4])
    seps = rng.sample([s for s in SEPS if s not in members], K - 1) if K >= 3 else [rng.choice([s for s in SEPS if s not in members])]
    same = K >= 3 and rng.random() < 0.4               # one separator for every link (the shape with bit 3 set)
    if same: seps = [seps[0]] * (K - 1)
    grp = rng.random() < 0.7
    esc = lambda s: "\\" + s if s in ".-/ " and s != " " else s
    parts = []
    for i in range(K):
        parts.append(f"({cls}+)" if grp else f"{cls}+")
        if i < K - 1: parts.append(esc(seps[i]))
    pat = "".join(parts)
    try:
        rx = cx.compile(pat)
        o = oracle.Regex(pat)
    except Exception as e:
        refused += 1; continue
    if not rx.supported or rx.strategy != o.strategy: refused += 1; continue
    try:
        blob = rx.blob()
    except Exception:
        refused += 1; continue
    if emu.trio_shape(blob) != (K | 8 if same else K): refused += 1; continue
    alpha = members + "".join(sorted(set(seps))) + rng.choice(["  \n", " x", "#"]) + rng.choice(SEPS)
    for _ in range(12):
        n = rng.choice([10, 70, 200, 700, 4100, 9000])
        w = [rng.choice([1, 2, 5]) for _ in alpha]
        hay = "".join(rng.choices(alpha, weights=w, k=n)).encode()
        a = np.frombuffer(hay, dtype=np.uint8)
        exp = o.find_all_index(a)
        expp = None
        if grp:                                      # captures come from the PikeVM: plain leftmost-first spans and the links
            sub = o.find_all_submatch_index(a)
            expp = np.stack([sub[:, 0]] + [sub[:, 2 * i + 1] for i in range(1, K)] + [sub[:, 1]], axis=1) if len(sub) else np.zeros((0, K + 1), dtype=np.int64)
        for ow in (60, 4, 1):
            got = emu.find_all_trio(blob, hay, ow)
            if got is None:
                continue
            served += 1
            ok = (got.shape == expp.shape and np.array_equal(got, expp)) if grp else (got.shape[0] == exp.shape[0] and np.array_equal(got[:, [0, -1]], exp))
            checked += 1
            if not ok:
                bad += 1
                print("MISMATCH", pat, o.strategy, ow, hay[:100]); break
print(f"seed {seed} patterns {npat} refused {refused} twin runs {served} checked {checked} bad {bad}")
