import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import coregex_amd as cx
from oracle import oracle as O
cases = [(r"\d+\.\d+\.\d+\.\d+", b"1."), (r"error|warning|fatal|critical", b"error"), (r"[\w]+", b"a"), (r"\d+:\d+:\d+", b"12:"),
         (r"\d+\.\d+x?", b"1."), (r"a+b|b+a", b"ab"), (r"HTTP/\d\.\d", b"HTTP/1.1"), (r"ab+c", b"abbc")]
for pat, unit in cases:
    rx = cx.compile(pat); o = O.Regex(pat)
    for n in (100 * 1024, 600 * 1024, 3 << 20):
        hay = np.frombuffer(b" 1.2.3.4 " + (unit * (n // len(unit) + 1))[:n] + b" error 12:3:4 abbc ab ", dtype=np.uint8)
        exp = o.find_all_index(hay)
        sys.stderr.write(f"--- {pat} {n}\n"); sys.stderr.flush()
        try:
            got = rx.find_all_index(hay)
            print(pat, n, "ok" if got.shape == exp.shape and np.array_equal(got, exp) else f"MISMATCH {len(got)} {len(exp)}", flush=True)
        except cx.CoregexError as e:
            print(pat, n, "ERR", e, flush=True)
        try:
            c = rx.count(hay)
            print(pat, n, "count ok" if c == len(exp) else f"COUNT MISMATCH {c} {len(exp)}", flush=True)
        except cx.CoregexError as e:
            print(pat, n, "COUNT ERR", e, flush=True)
rx = cx.compile(r"(\w+)@(\w+)\.(\w+)"); o = O.Regex(r"(\w+)@(\w+)\.(\w+)")
for hay in (b"x a@b.c y", b"ab" * (400 * 1024) + b" a@b.c ", (b"a@b.c" * 200000)):
    exp = o.find_all_submatch_index(hay)
    sys.stderr.write(f"--- submatch {len(hay)}\n"); sys.stderr.flush()
    try:
        got = rx.find_all_submatch_index(hay)
        print("submatch", len(hay), "ok" if got.shape == exp.shape and np.array_equal(got, exp) else f"MISMATCH {got.shape} {exp.shape}", flush=True)
    except cx.CoregexError as e:
        print("submatch", len(hay), "ERR", e, flush=True)
