#!/usr/bin/env python3
"""Debug: persistent fields kernel rows vs oracle on the random texts of tests/test_gpu_fields.py::test_random_text; prints where rows differ."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coregex_amd as cx
from oracle import oracle as O
for pat, alpha in [(r"\d+\.\d+\.\d+\.\d+", "0123456789..  x\n"), (r"\d+:\d+:\d+", "0123:: \n")]:
    rx = cx.compile(pat); o = O.Regex(pat)
    rng = random.Random(len(pat))
    for it in range(24):
        n = rng.choice([700, 4100, 9000, 40000, 130000, 500000])
        kind = it % 3
        w = ([3, 3, 1] + [1] * len(alpha) if kind == 0 else [1] * len(alpha) if kind == 1 else [5] + [1] * len(alpha))[: len(alpha)]
        hay = np.frombuffer("".join(rng.choices(alpha, weights=w, k=n)).encode(), dtype=np.uint8)
        exp = o.find_all_index(hay)
        d = torch.from_numpy(np.concatenate([hay, np.zeros(64, dtype=np.uint8)])).cuda()
        t = cx.Timing()
        cnt = rx.find_all_device(d.data_ptr(), n, timing=t)
        l1 = (t.kernel, t.n_launches, t.fallback_reason)
        out = torch.full((cnt + 8, 2), -7, dtype=torch.int64, device="cuda")
        n2 = rx.find_all_device(d.data_ptr(), n, out.data_ptr(), cnt + 8, timing=t)
        rows = out[:cnt].cpu().numpy()
        ok = rows.shape == exp.shape and np.array_equal(rows, exp)
        print(pat, "it", it, "n", n, "rows", cnt, len(exp), "count call", l1, "rows call", (t.kernel, t.n_launches, t.fallback_reason, list(t.ladder)[:t.n_ladder]), "ok", ok, flush=True)
        if not ok and rows.shape == exp.shape:
            bad = np.nonzero((rows != exp).any(axis=1))[0]
            print("  first bad rows", bad[:10].tolist(), "count", len(bad))
            for i in bad[:6]:
                print("   row", i, "got", rows[i].tolist(), "exp", exp[i].tolist(), "exp tile", exp[i][1] // 3840, "unit block", (exp[i][1] // 3840) >> 6)
            print("  multiset equal:", sorted(map(tuple, rows.tolist())) == sorted(map(tuple, exp.tolist())), " untouched (-7):", int((rows == -7).any(axis=1).sum()))
