import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import coregex_amd as cx
from oracle import oracle as O
pat = sys.argv[1] if len(sys.argv) > 1 else r"\d+\.\d+\.\d+\.\d+"
unit = (sys.argv[2] if len(sys.argv) > 2 else "1.").encode()
rx = cx.compile(pat); o = O.Regex(pat)
for n in (8000, 16000, 40000, 300000, 3000000):
    hay = np.frombuffer((b" x " + unit * (n // len(unit)))[:n - 16] + b" y 1.2.3.4 abcx ", dtype=np.uint8)
    exp = o.find_all_index(hay)
    for rep in range(2):
        try:
            got = rx.find_all_index(hay)
        except cx.CoregexError as e:
            print(n, "ERR", e); break
        if got.shape == exp.shape and np.array_equal(got, exp):
            print(n, "ok", len(exp)); continue
        k = 0
        while k < min(len(got), len(exp)) and np.array_equal(got[k], exp[k]): k += 1
        bad = [i for i in range(min(len(got), len(exp))) if not np.array_equal(got[i], exp[i])]
        print(n, "MISMATCH rows", len(got), "exp", len(exp), "first diff row", k, got[max(0,k-1):k+3].tolist(), exp[max(0,k-1):k+3].tolist(), "n bad", len(bad), "bad idx", bad[:6], bad[-3:])
