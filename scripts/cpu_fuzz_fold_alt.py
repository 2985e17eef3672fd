"""CPU fuzz of the folded literal sets (program.cc wrappedLiterals + walk.hpp kTeddyFold): random case-insensitive alternations of words,
with and without assertions and suffixes, through the literal kernel's sequential twin against the oracle.
python scripts/cpu_fuzz_fold_alt.py [n_patterns] [seed]"""
import sys, os, time, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import coregex_amd as cx
import emu
from oracle import oracle as O

WORDS = ["error", "fail", "exception", "panic", "fatal", "warn", "warning", "timeout", "refused", "denied", "kernel", "system", "task", "risk", "disk", "ask", "jan", "jun", "jul",
         "get", "post", "put", "select", "insert", "k1", "s_9", "a-b", "x.y", "abc", "abd", "xyz", "mask", "desk", "ssl", "kkk", "ok:", "e2e", "i/o"]

def main(n=300, seed=1):
    rng = np.random.default_rng(seed)
    n_fold = n_cmp = n_give = 0
    strat, kinds = {}, {}
    t0 = time.time()
    seen = set()
    while n_fold < n:
        k = int(rng.integers(2, 9))
        ws = [WORDS[int(i)] for i in rng.choice(len(WORDS), size=k, replace=False)]
        body = "|".join(w.replace(".", r"\.") for w in ws)
        form = int(rng.integers(0, 6))
        pat = ["(?i)(%s)", r"(?i)\b(%s)\b", "(?i)(?:%s)", "(?i)(%s):", r"(?i)(?m)^(%s)", "(?i)x(%s)"][form] % body
        if pat in seen: continue
        seen.add(pat)
        try: o = O.Regex(pat)
        except O.OracleError: continue
        try: rx = cx.compile(pat)
        except cx.CoregexError: continue
        if not rx.supported: continue
        blob = rx.blob()
        kind = struct.unpack_from("<I", blob, 4)[0]
        if kind != 4: continue
        looks = struct.unpack_from("<I", blob, struct.unpack_from("<I", blob, 56)[0] + 44)[0]
        if not looks & 0x10000: continue
        n_fold += 1
        strat[rx.strategy] = strat.get(rx.strategy, 0) + 1
        toks = [w.encode() for w in ws] + [w.upper().encode() for w in ws] + [w.capitalize().encode() for w in ws] + [w[:-1].encode() for w in ws] + \
               [b" ", b"\n", b":", b"x", b"_", b"-", b"X", "K".encode(), "ſ".encode(), b"ERR", b"9"]
        for size in (0, 3, 40, 400, 4000, 9000):
            hay = np.frombuffer(b"".join(toks[int(i)] for i in rng.integers(0, len(toks), size=size // 3 + 1))[:size], dtype=np.uint8)
            exp = o.find_all_index(hay)
            for geom in ((3840, 256), (192, 64)):
                got = emu.find_all_teddy_wave(blob, hay, *geom)
                if got is None or isinstance(got, int):
                    n_give += 1
                    continue
                n_cmp += 1
                if got.shape != exp.shape or not np.array_equal(got, exp):
                    print("MISMATCH", repr(pat), rx.strategy, geom, bytes(hay[:100]), got[:5].tolist(), exp[:5].tolist())
                    return 1
    print(f"{n_fold} folded literal programs {strat}, {n_cmp} comparisons clean, {n_give} twin give-ups (dense tiles), {time.time()-t0:.1f}s")
    return 0

if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
