#!/usr/bin/env python3
"""Extended GPU fuzz (not part of pytest): random patterns x structured haystacks, device path vs the oracle.
Usage: python scripts/gpu_fuzz.py [seed] [n_patterns].  Prints one line per mismatch and a summary."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import coregex_amd as cx
from oracle import oracle

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
npat = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(seed)
atoms = ["a", "b", "c", "x", "y", r"\.", ":", "-", r"\d", "[a-c]", "[x-z]", r"\d+", "[a-c]+", "[x-z]+", "a+", r"\.+", "[0-4]+", "[5-9]",
         "ab|xy", "abc|xyz|a:c", r"\w", r"\w+", "[a-z0-9]+", "[ab]", "(a|b)", "(ab)+", "a?", r"\d{2}", r"\d{1,3}", "x*", "(xy|ab|ca)",
         "abcx|bcxy|cxyz|xyza", "z+", ".", ".*", r"\s", r"\S+", "(?:ab)*", "a*", r"\d{2,}", "(a+)(b+)", "[^x]", "(?i:ab)", "(?i:x)", "b+?", "a+?", "[a-c]+?",
         "(?:a|b|c)+", "abcabc", "abc", "xyz", "a:c", "xyzxyz|abcabc|a:c:a:", r"[\d.]+", "(x|y|z)", " ", r"\n"]
LOOK = bool(os.environ.get("FUZZ_LOOK"))      # word boundaries / multi-line anchors: every pattern gets (?m) and one assertion at least
if LOOK:
    atoms = atoms[:36] + [r"\b", r"\B", r"\b", "^", "$", "^", "$", "_", "[a-c_]+", r"\w+", " ", "A", r"\n", r"(a|\b)", r"(\bab|xy\b)", r"(^a|b$)", r"(?:$\n^)?", "^ab|xy$",
                         "abc", "xyz", r"\d+", "abcx|bcxy|cxyz|xyza"]
WIDE = bool(os.environ.get("FUZZ_WIDE"))      # `.` and classes past U+007F in every pattern; multi-byte sequences and stray bytes >= 0x80 in the haystacks
WIDE_ATOMS = [".", ".", ".*", ".+", ".?", r"[^x]", r'[^"]', r"\S", r"\S+", r"\D", r"\W", r"[^a-c]+", r"[^\n]*", "(.)", r"(\S+)", r'"[^"]*"', "é", "[aé]", r"[^:]*:", ".+?", r"\D+?", "x.y", "(?s:.)"]
FOLD = bool(os.environ.get("FUZZ_FOLD"))      # a case-insensitive literal in every pattern (literal sets of case variants; never run on a device in round 3)
FOLD_ATOMS = ["(?i:error)", "(?i:warn)", "(?i:k)", "(?i:s1)", "(?i:ok)", "(?i:get)", "(?i:ab|xy)", "(?i:(abc))", "(?i:exception)", "(?i:a)b", "(?i:xyz)+", "(?i:[a-c])", "(?i:[x-z]+)", "(?i:a|b)c"]
TEXT = bool(os.environ.get("FUZZ_TEXT"))      # a text-start anchor (\A, ^ without (?m)) in every pattern (round 4); the haystacks get leads that match at position 0
TEXT_ATOMS = ["^", "^", r"\A", "(?:^|,)", r"(^|\s)", "(?:^|x)", "(?:a|^b)", "(?:^a|b)", r"(?:^|:)"]
if TEXT:
    atoms = atoms[:36] + TEXT_ATOMS * 4 + [",", " ", r"\b", r"\w+", "abc", r"\d+"]
END = bool(os.environ.get("FUZZ_END"))        # an end-of-text anchor (\z, $ without (?m)) in every pattern (round 6); the haystacks get tails that match at the last position
END_ATOMS = ["$", "$", r"\z", "(?:$|,)", r"(\s|$)", "(?:x|$)", "(?:a$|b)", "(?:a|b$)", r"(?:\z|:)"]
if END:
    atoms = atoms[:36] + END_ATOMS * 4 + [",", " ", r"\b", r"\w+", "abc", r"\d+", "|", "|"]
if WIDE:
    atoms = atoms + WIDE_ATOMS * 3
if FOLD:
    atoms = atoms[:34] + FOLD_ATOMS * 3
alphabet = np.frombuffer(b"abcxyz.:-0123456789 \nABX\x00\x7f\x80\xc3\xa9\xff" + (b"_ \n_a \n" if LOOK else b"") + ('"éé日😀'.encode() if WIDE else b""), dtype=np.uint8)
T = 3840
def rnd(n, p=None):
    return alphabet[rng.integers(0, len(alphabet), size=int(n))] if p is None else alphabet[rng.choice(len(alphabet), size=int(n), p=p)]
skew = np.ones(len(alphabet)); skew[:6] = 8; skew /= skew.sum()
sk2 = np.ones(len(alphabet)); sk2[9:19] = 10; sk2 /= sk2.sum()
hays = [rnd(0), rnd(5), rnd(T - 1), rnd(T + 1), rnd(32 * T), rnd(32 * T + 7, skew), rnd(70000, sk2), rnd(200000, skew),
        np.frombuffer((b"xyab" + b"." * 28) * 4000, dtype=np.uint8), np.frombuffer(b"abcxyza:c" * 9000, dtype=np.uint8),
        np.frombuffer((b"1.2.3.4 " * 7 + b"\n") * 3000, dtype=np.uint8), np.frombuffer(b"a" * 9000 + b"b" + b"a" * 70000, dtype=np.uint8)]
if FOLD:
    toks = [w for base in (b"error", b"warn", b"k", b"s1", b"ok", b"get", b"ab", b"xy", b"abc", b"exception", b"xyz", b"a", b"b", b"c") for w in (base, base.upper(), base.capitalize(), base[:1] + base[1:].upper())]
    toks += [b" ", b":", b"1", b"\n", "\u212a".encode(), "\u017f".encode(), b"-"]
    hays += [b"".join(toks[int(i)] for i in rng.integers(0, len(toks), size=n)) for n in (40, 4000, 60000)]
    hays = [np.frombuffer(bytes(h), dtype=np.uint8) if not isinstance(h, np.ndarray) else h for h in hays]
if TEXT:
    hays = [h for h in hays] + [np.concatenate([np.frombuffer(lead, dtype=np.uint8), h]) for lead in (b"abc", b"12", b"xyz,", b"a", b"b:") for h in (hays[2], hays[5], hays[7])]
if END:
    hays = [h for h in hays] + [np.concatenate([h[:cut], np.frombuffer(tail, dtype=np.uint8)]) for tail in (b"abc", b"12", b",xyz", b"a", b":b", b"x")
                                for h, cut in ((hays[2], T - 1), (hays[3], T - 2), (hays[4], 32 * T - 3), (hays[5], 32 * T - 1), (hays[7], 199999))]
if os.environ.get("FUZZ_FEW"):
    # few-symbol haystacks of several groups (120 KiB each): the sets of possible entry states stay unresolved for long
    # stretches, so the transducer kernel's member maps, its serial chain and the tile / group hand-off do the work
    few = [np.frombuffer(a, dtype=np.uint8) for a in (b"ab", b"abc", b"ab ", b"xyz", b"1.", b"a:c", b"abx\n", b"01 .")]
    hays = [f[rng.integers(0, len(f), size=int(n))] for f in few for n in (70000, 260000)]
    # ... and two long ones: hundreds of groups in the kernel's dense modes, i.e. several windows of the look-back over maps (round 3)
    hays += [few[4][rng.integers(0, 2, size=2 << 20)], np.frombuffer((b"1." * (1 << 20)), dtype=np.uint8)]
ORACLE_ONLY = bool(os.environ.get('FUZZ_ORACLE_ONLY'))
seen, n_dev, n_sub, bad = set(), 0, 0, 0
n_refused = 0
n_nosync = 0
n_budget = 0
by_strategy = {}
t0 = time.time()
while len(seen) < npat:
    pat = "".join(atoms[int(rng.integers(0, len(atoms)))] for _ in range(int(rng.integers(1, 5))))
    if LOOK:
        if not any(t in pat for t in ("\\b", "\\B", "^", "$")):
            continue
        pat = "(?m)" + pat
    if FOLD and "(?i" not in pat:
        continue
    if TEXT and not any(t in pat for t in ("^", "\\A")):
        continue
    if END and (not any(t in pat for t in ("$", "\\z")) or pat.startswith("|") or pat.endswith("|") or "||" in pat):
        continue
    if WIDE and not any(a in pat for a in (".", "[^", "\\S", "\\D", "\\W", "é")):
        continue
    if pat in seen:
        continue
    seen.add(pat)
    if os.environ.get("FUZZ_TRACE"):                      # which pattern is running (a hang or a very slow program shows as the last line)
        print("[%6.1fs] %r" % (time.time() - t0, pat), file=sys.stderr, flush=True)
    try:
        rx = cx.compile(pat)
    except cx.CoregexError:
        continue
    try:
        o = oracle.Regex(pat)
    except oracle.OracleError as ex:
        if rx.supported:
            print('ORACLE-REJECTS-BUT-DEVICE-ACCEPTS', repr(pat), ex); bad += 1
        continue
    if rx.strategy != o.strategy and (o.strategy_restated or not LOOK):
        print("STRATEGY", repr(pat), rx.strategy, o.strategy); bad += 1
    if rx.supported:
        n_dev += 1
        by_strategy[rx.strategy] = by_strategy.get(rx.strategy, 0) + 1
        for hi, hay in enumerate(hays):
            exp = o.find_all_index(hay)
            if ORACLE_ONLY:
                continue
            try:
                got = rx.find_all_index(hay)
            except cx.UnsupportedInput as ex:
                if "transducer kernel's budgets" in str(ex):   # look-around programs have no table-walking image: dense / long-pending input is refused
                    n_budget += 1
                    continue
                # legitimate only for a UseBoth program whose plain leftmost-first result holds a match longer than 100 bytes
                if 'serial-walk budget' in str(ex) and (len(hay) > 128 * 1024 or rx.fsm_image() is None):   # (round 4: also the total budget of slow reads of a program without a transducer image)
                    n_nosync += 1          # plausible: every byte of a long periodic haystack is in the pattern's alphabet
                    continue
                plain = o.find_all_submatch_index(hay)[:, :2]
                if not (rx.strategy == 'UseBoth' and len(plain) and int((plain[:, 1] - plain[:, 0]).max()) > 100):
                    print('REFUSED', repr(pat), rx.strategy, 'hay', hi, len(hay), ex); bad += 1
                else:
                    n_refused += 1
                continue
            except cx.CoregexError as ex:
                print('ERROR', repr(pat), rx.strategy, 'hay', hi, len(hay), ex); bad += 1
                continue
            if got.shape != exp.shape or not np.array_equal(got, exp):
                print("MISMATCH", repr(pat), rx.strategy, "hay", hi, len(hay), got.shape, exp.shape); bad += 1
            try:
                c = rx.count(hay)
            except cx.UnsupportedInput:
                c = len(exp)
            if c != len(exp):
                print("COUNT", repr(pat), rx.strategy, "hay", hi, c, len(exp)); bad += 1
    if "(" in pat and rx.submatch_supported:
        n_sub += 1
        for hi, hay in enumerate(hays[:8]):
            exp = o.find_all_submatch_index(hay)
            if ORACLE_ONLY:
                continue
            try:
                got = rx.find_all_submatch_index(hay)
            except cx.UnsupportedInput as ex:                 # a match longer than the serial-walk / per-row budgets (few-symbol haystacks)
                if "transducer kernel's budgets" in str(ex):   # span programs with assertions have no table-walking image either
                    n_budget += 1
                    continue
                if 'serial-walk budget' in str(ex) and len(hay) > 64 * 1024:
                    n_nosync += 1
                    continue
                plain = exp[:, :2]
                if rx.strategy == 'UseBoth' and len(plain) and int((plain[:, 1] - plain[:, 0]).max()) > 100:
                    n_refused += 1
                    continue
                print("SUBMATCH-REFUSED", repr(pat), "hay", hi, len(hay), ex); bad += 1
                continue
            if got.shape != exp.shape or not np.array_equal(got, exp):
                print("SUBMATCH", repr(pat), "hay", hi, len(hay), got.shape, exp.shape); bad += 1
print("seed", seed, "patterns", len(seen), "device", n_dev, "submatch", n_sub, "refused-long-UseBoth", n_refused, "refused-no-sync", n_nosync, "refused-budget", n_budget, "bad", bad, by_strategy, "%.1fs" % (time.time() - t0))
