"""Repro helper (GPU box): tests/test_gpu_wrapped.py::test_rows for one pattern, the test's own sequence of calls, repeated; prints what differs."""
import random, sys
import numpy as np, torch
sys.path.insert(0, "tests")
import coregex_amd as cx
from oracle import oracle; oracle.build()
from test_wrapped_cpu import TOKS
pat = sys.argv[2] if len(sys.argv) > 2 else r"(?m)^(?:abc|xyz)$"
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    rng = random.Random(len(pat) * 3 + 1)
    for n in [0, 1, 7, 3839, 3840, 3841, 61441, 500000, 3_000_000]:
        for sparse in (False, True):
            toks = TOKS + ([b" pad pad pad pad pad pad pad pad "] * 12 if sparse else [])
            hay = np.frombuffer(b"".join(rng.choice(toks) for _ in range(max(1, n // 3)))[:n], dtype=np.uint8)
            exp = o.find_all_index(hay)
            t = cx.Timing()
            if hay.size:
                d = torch.from_numpy(hay.copy()).cuda()
                cnt = rx.find_all_device(d.data_ptr(), hay.size, timing=t)
                if cnt != len(exp): print(f"rep {rep} n {n} sparse {sparse}: COUNT {cnt} / {len(exp)} kernels {list(t.kernels)}", flush=True); bad += 1
            got = rx.find_all_index(hay)
            if got.shape != exp.shape or not np.array_equal(got, exp):
                bad += 1
                miss = sorted(set(map(tuple, exp.tolist())) - set(map(tuple, got.tolist())))
                extra = sorted(set(map(tuple, got.tolist())) - set(map(tuple, exp.tolist())))
                print(f"rep {rep} n {n} sparse {sparse}: rows {len(got)} / {len(exp)} missing {miss[:4]} tile {[m[0] // 3840 for m in miss[:4]]} in-tile {[m[0] % 3840 for m in miss[:4]]} unit {[m[0] // 15360 for m in miss[:4]]} of {n // 15360} extra {extra[:4]} tail {bytes(hay[-8:])}", flush=True)
            g2 = rx.find_all_index(hay, 2)
            if not np.array_equal(g2, exp[:2]): print(f"rep {rep} n {n} sparse {sparse}: limit-2 rows differ", flush=True); bad += 1
print("bad", bad)
