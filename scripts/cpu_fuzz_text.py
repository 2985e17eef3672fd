"""CPU fuzz of text anchors inside unanchored patterns (`(^|,)\\d+`, `foo|\\Abar`; round 4): the front-end's strategy against the
oracle's, and — for the programs the build accepts — the transducer's sequential twin against the oracle on haystacks whose first bytes
matter.  python scripts/cpu_fuzz_text.py [n_patterns] [seed] [end]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import coregex_amd as cx
import emu
from oracle import oracle as O

ATOMS = ["^", "^", r"\A", "(?:^|,)", r"(^|\s)", "(?:^|x)", r"\b", r"\d+", r"\d", "[a-c]+", "[a-c]", "foo", "bar", "ab", "x", ",", " ", r"\s", r"\w+", "(?:foo|bar)", "(?:a|^b)", "(?:^a|b)", ":", r"\.",
         "(a|b)", r"\S+", r"\w", "a+", "x?", "(?:ab)+", "$", r"\z"]

NEED = ("^", r"\A")     # third argument "end": patterns with an end-of-text anchor instead (round 6)

def main(n=300, seed=1):
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"abcfoxr ,:.019\n_Z", dtype=np.uint8)
    words = [b"foo", b"bar", b"ab", b"12", b",", b" ", b"x", b"a", b"b", b"abc", b"foo,bar", b"9,", b"\n", b":", b"."]
    seen, n_pat, n_served, n_cmp, why, strat = set(), 0, 0, 0, {}, {}
    n_strat_refused = n_sub = 0
    t0 = time.time()
    while n_pat < n:
        k = int(rng.integers(2, 6))
        pat = "".join(ATOMS[int(rng.integers(0, len(ATOMS)))] for _ in range(k))
        if rng.random() < 0.4: pat = pat + "|" + "".join(ATOMS[int(rng.integers(0, len(ATOMS)))] for _ in range(int(rng.integers(1, 4))))
        if pat in seen or not any(t in pat for t in NEED): continue
        seen.add(pat)
        try: o = O.Regex(pat)
        except O.OracleError: continue
        try: rx = cx.compile(pat)
        except cx.CoregexError as e:
            print("HOST-REJECTS", repr(pat), e); return 1
        n_pat += 1
        if rx.strategy != o.strategy:
            if rx.supported: print("STRATEGY", repr(pat), rx.strategy, o.strategy); return 1
            n_strat_refused += 1                                  # (a refused pattern whose strategy name differs: `\\z` in the middle, never matches)
        strat[o.strategy] = strat.get(o.strategy, 0) + 1
        if not rx.supported:
            key = rx.why_unsupported[:70]; why[key] = why.get(key, 0) + 1
            continue
        img = rx.fsm_image()
        if img is None:
            continue
        n_served += 1
        hays = [b"", b"f", b"foo", b"bar", b"ab", b"12", b",12", b"foo bar", b"bar foo", b"abfoo", b"a", b"x", b" ", b"\n", b"ab,ab", b"12,12 12", b"bbb", b"foo\nbar", b"\nfoo"]
        for size in (10, 60, 400, 4000, 9000):
            hays.append(bytes(alphabet[rng.integers(0, len(alphabet), size=size)]))
            hays.append(b"".join(words[int(i)] for i in rng.integers(0, len(words), size=size // 2))[:size])
        sub = "(" in pat.replace("(?:", "") and rx.submatch_supported
        if sub:
            sb, cb = rx.submatch_blobs()[:2]
            oc = rx.offset_captures
        for hay in hays:
            a = np.frombuffer(hay, dtype=np.uint8)
            exp = o.find_all_index(a)
            if sub and len(hay) <= 4000:
                es = o.find_all_submatch_index(a)
                if oc is not None:
                    for k2, (src, d) in enumerate(oc):
                        if not np.array_equal(es[:, k2], es[:, 1 if src else 0] + d): print("OFFSET-CAPS", repr(pat), k2, hay[:60]); return 1
                else:
                    try: gs = emu.captures_bt(cb, hay, es[:, :2], es.shape[1])      # the backtracking pass over the spans (capi_captures.hip launchCapturePass)
                    except AssertionError as e:
                        if "error -4" not in str(e): print("SUBMATCH-TWIN", repr(pat), hay[:60], e); return 1
                        gs = None
                    if gs is not None and (gs.shape != es.shape or not np.array_equal(gs, es)): print("SUBMATCH", repr(pat), hay[:60], gs[:3].tolist(), es[:3].tolist()); return 1
                n_sub += 1
            if rx.strategy == "UseBoth":
                plain = O.Regex(pat).find_all_submatch_index(a)[:, :2]
                if len(plain) and int((plain[:, 1] - plain[:, 0]).max()) > 100: continue
            for tile, chunk in ((3840, 32), (64, 8)):
                got = emu.find_all_fsm(img, a, tile, chunk)
                if isinstance(got, int) and got in (-18, -32): got = emu.find_all_fsm(img, a, tile, chunk, dense=1)
                if isinstance(got, int): continue
                n_cmp += 1
                if got.shape != exp.shape or not np.array_equal(got, exp):
                    print("MISMATCH", repr(pat), rx.strategy, tile, chunk, hay[:80], got[:6].tolist(), exp[:6].tolist()); return 1
    print(f"{n_pat} patterns with a text anchor {NEED} {strat}; {n_served} served, {n_cmp} twin comparisons clean, {n_sub} capture comparisons clean, {n_strat_refused} refused patterns with another strategy name, {time.time()-t0:.1f}s")
    for k, v in sorted(why.items(), key=lambda kv: -kv[1]): print(f"  refused {v:5d}: {k}")
    return 0

if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "end": NEED = ("$", r"\z")
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
