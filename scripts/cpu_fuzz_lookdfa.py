"""CPU fuzz of host/lookdfa.cc: look-around programs of the reference's lazy-DFA strategies (UseDFA / UseBoth / UseDigitPrefilter) that the build-time
proof accepts must give, on the transducer's sequential twin, exactly what the oracle's restated look-aware lazy DFA gives — with
ONE oracle engine per pattern reused over all haystacks (the proof also claims independence of cache history).
python scripts/cpu_fuzz_lookdfa.py [n_patterns] [seed] [text]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import coregex_amd as cx
import emu
from oracle import oracle as O

ATOMS = [r"\b", r"\b", r"\B", r"\w", r"\w+", r"\w+", r"\w*", r"\w?", r"[\w.]+", r"[\w-]+", r"\W", r"\W+", " ", " +", "=", ":", "@", r"\.", "-", ";", "foo", "bar", "ab", "x", "err",
         r"\d+", "[a-z]+", "[A-Z]", r"(\w+)", r"(?:\w+ )+", r"(?:foo|bar)", r"(?:ab|\w)", r"\w{2}", r"\w{2,}", r"(?:\b|x)", r"(?:\w+\b)", r"\n", "^", "$", r"[^\n]", r"[^\n]+", r"\w+\n",
         r"(?:=\w+)?", r"(?:\.\w+)+", r"\w+?", r"(?:com|org)", r"\s", r"\s+", r"[ \t]+"]

DIGIT_LEADS = [r"\d+", r"\d+", r"\d", r"\d{2}", r"[0-9]+", r"\d+\.", r"\d+\.\d+", r"[0-5]+", r"\d{2,}", r"\d+-"]

TEXT_ANCHORS = False

def main(n=400, seed=1):
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"abfor xyERZ_0912 .=:@-;\n\t  ", dtype=np.uint8)
    words = [b"12", b"3.4", b"56-", b"007 ", b"foo", b"bar", b"ab", b"err", b"x=y", b"a@b.com", b" ", b"  ", b"\n", b"foo bar", b"user_1", b"k=v;", b"ab.cd.org", b"9", b"_", b"-", b"x"]
    seen, n_dfa, n_ok, n_cmp, why, n_digit = set(), 0, 0, 0, {}, 0
    t0 = time.time()
    tries = 0
    while n_dfa < n and tries < n * 400:
        tries += 1
        pat = "".join(ATOMS[int(rng.integers(0, len(ATOMS)))] for _ in range(int(rng.integers(3, 9))))
        if rng.random() < 0.25: pat = DIGIT_LEADS[int(rng.integers(0, len(DIGIT_LEADS)))] + pat      # digit-lead: UseDigitPrefilter
        if pat in seen or not any(t in pat for t in (r"\b", r"\B", "^", "$")): continue
        seen.add(pat)
        if not TEXT_ANCHORS or rng.random() < 0.5: pat = "(?m)" + pat      # (third argument "text": half of the patterns keep ^ / $ as text anchors, round 6)
        try: o = O.Regex(pat)
        except O.OracleError: continue
        if o.strategy not in ("UseDFA", "UseBoth", "UseDigitPrefilter"): continue
        try: rx = cx.compile(pat)
        except cx.CoregexError: continue
        if rx.strategy != o.strategy:
            if rx.supported or not TEXT_ANCHORS: print("STRATEGY", repr(pat), rx.strategy, o.strategy); return 1
            continue                                              # (a refused pattern with `$` in its middle — it never matches — whose strategy name differs, as in cpu_fuzz_text.py)
        n_dfa += 1
        n_digit += o.strategy == "UseDigitPrefilter"
        if not rx.supported:
            k = rx.why_unsupported[:60]; why[k] = why.get(k, 0) + 1
            continue
        img = rx.fsm_image()
        if img is None: continue
        n_ok += 1
        hays = []
        for k in (0, 1, 5, 40, 200, 900):
            hays.append(alphabet[rng.integers(0, len(alphabet), size=k)])
            parts = [words[int(rng.integers(0, len(words)))] if rng.random() < 0.6 else bytes(alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(1, 4)))]) for _ in range(k // 3 + 1)]
            hays.append(np.frombuffer(b"".join(parts), dtype=np.uint8))
        hays.append(np.frombuffer(b" " * 150 + b"foo bar ab" + b"." * 130 + b"user_1 k=v; a@b.com\n" + b"-" * 101 + b"x", dtype=np.uint8))   # gaps > 100 bytes: the UseBoth restart point
        o_rev = O.Regex(pat)                              # a second engine that meets the haystacks in the opposite order,
        exp_rev = {}                                      # and a fresh one per haystack: three cache histories, one answer
        for i in range(len(hays) - 1, -1, -1): exp_rev[i] = o_rev.find_all_index(hays[i])
        for hi, hay in enumerate(hays):
            exp = o.find_all_index(hay)
            fresh = O.Regex(pat).find_all_index(hay)
            if not (np.array_equal(exp, fresh) and np.array_equal(exp, exp_rev[hi])):
                print("HISTORY", repr(pat), rx.strategy, bytes(hay[:160]), exp[:6].tolist(), fresh[:6].tolist(), exp_rev[hi][:6].tolist())
                return 1
            if rx.strategy == "UseBoth":
                plain = O.Regex(pat).find_all_submatch_index(hay)[:, :2]
                if len(plain) and int((plain[:, 1] - plain[:, 0]).max()) > 100: continue
            for tile, chunk in ((3840, 32), (64, 8), (256, 16)):
                got = emu.find_all_fsm(img, hay, tile, chunk)
                if isinstance(got, int) and got in (-18, -32): got = emu.find_all_fsm(img, hay, tile, chunk, dense=1)
                if isinstance(got, int): continue
                n_cmp += 1
                if got.shape != exp.shape or not np.array_equal(got, exp):
                    np.save("/tmp/lookdfa_fail_hay.npy", hay)
                    print("MISMATCH", repr(pat), rx.strategy, tile, chunk, bytes(hay[:160]), got[:6].tolist(), exp[:6].tolist())
                    return 1
    print(f"{n_dfa} look-around programs of UseDFA/UseBoth/UseDigitPrefilter ({n_digit} digit), {n_ok} accepted by the proof, {n_cmp} comparisons with the restated reference DFA clean, {time.time()-t0:.1f}s")
    for k, v in sorted(why.items(), key=lambda kv: -kv[1]): print(f"  refused {v:5d}: {k}")
    return 0

if __name__ == "__main__":
    TEXT_ANCHORS = len(sys.argv) > 3 and sys.argv[3] == "text"
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 400, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
