import random, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import coregex_amd as cx
from oracle import oracle as O
O.build()
for pat in [r"a*", r"x?y*"]:
    rx, o = cx.compile(pat), O.Regex(pat)
    rng = random.Random(len(pat) * 13 + 1)
    hays = [b"", b"a", b"ab", b"xyy a y", b"12 345"]
    for _ in range(14):
        n = rng.choice([3, 50, 700, 4100, 9000, 70000, 300000])
        w = rng.choice([[1] * 8, [8, 1, 1, 1, 1, 1, 1, 1], [1, 1, 8, 1, 1, 1, 1, 1]])
        hays.append(bytes(rng.choices(b"ab xy1c\n", weights=w, k=n)))
    for hay in hays:
        a = np.frombuffer(hay, dtype=np.uint8)
        exp = o.find_all_index(a)
        try:
            got = rx.find_all_index(a)
            print(pat, len(hay), "ok" if np.array_equal(got, exp) else ("MISMATCH", got[:5].tolist(), exp[:5].tolist()), flush=True)
        except Exception as e:
            print(pat, len(hay), hay[:30], "EXC", str(e)[:100], flush=True)
