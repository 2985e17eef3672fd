"""CPU fuzz (not part of pytest): random patterns vs the oracle through the host front-end and the sequential twins of the
kernels (tests/emu).  Usage: python scripts/cpu_fuzz.py SEED0 SEED1 — 150 patterns per seed, one line per mismatch."""
import sys, struct, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import coregex_amd as cx, emu
from oracle import oracle
seed0, seed1 = int(sys.argv[1]), int(sys.argv[2])
atoms = ["a", "b", "c", "x", "y", r"\.", ":", "-", r"\d", "[a-c]", "[x-z]", r"\d+", "[a-c]+", "[x-z]+", "a+", r"\.+", "[0-4]+", "[5-9]",
         "ab|xy", "abc|xyz|a:c", r"\w", r"\w+", "[a-z0-9]+", "[ab]", "(a|b)", "(ab)+", "a?", r"\d{2}", r"\d{1,3}", "x*", "(xy|ab|ca)",
         "abcx|bcxy|cxyz|xyza", "z+", ".", ".*", r"\s", r"\S+", "(?:ab)*", "a*", r"\d{2,}", "(a+)(b+)", "[^x]", "(?i:ab)", "(?i:x)", "b+?", "a+?", "[a-c]+?", "(?:a|b|c)+",
         "abcabc", "abc", "xyz", "a:c", "xyzxyz|abcabc|a:c:a:", r"[\d.]+", "(x|y|z)", " ", r"\n"]
if os.environ.get("FUZZ_FOLD"):    # case-insensitive literals in every pattern (literal/extractor.go:838-941 expandCaseFoldLiteral)
    FOLD = ["(?i:error)", "(?i:warn)", "(?i:k)", "(?i:s1)", "(?i:ok)", "(?i:get)", "(?i:ab|xy)", "(?i:(abc))", "(?i:exception)", "(?i:a)b", "(?i:xyz)+", "(?i:[a-c])", "(?i:[x-z]+)", "(?i:a|b)c"]
    atoms = atoms[:34] + FOLD * 3
alphabet = np.frombuffer(b"abcxyz.:-0123456789 \nABX\x00\x7f\x80\xc3\xa9\xff" + (b"ERRORerrorWarnOKkKsSgetGET\xe2\x84\xaa\xc5\xbf" if os.environ.get("FUZZ_FOLD") else b""), dtype=np.uint8)
bad=0; tot=0; strat={}; n_long=0; n_bt_limit=0; n_cc_unchecked=0; n_cc_fallback=0
t0=time.time()
for seed in range(seed0, seed1):
    rng = np.random.default_rng(seed)
    hays = [alphabet[rng.integers(0, len(alphabet), size=int(n))].tobytes() for n in (0, 3, 200, 5000)]
    hays += [alphabet[rng.choice(len(alphabet), size=5000, p=rng.dirichlet(0.25 * np.ones(len(alphabet))))].tobytes() for _ in range(4)]
    hays += [b"abcxyza:c" * 300, b"a" * 900 + b"b" + b"a" * 900]
    if os.environ.get("FUZZ_FOLD"):   # words of the fold atoms in every casing, glued and separated
        toks = [w for base in (b"error", b"warn", b"k", b"s1", b"ok", b"get", b"ab", b"xy", b"abc", b"exception", b"xyz", b"a", b"b", b"c") for w in (base, base.upper(), base.capitalize(), base[:1] + base[1:].upper())]
        toks += [b" ", b":", b"1", b"\n", "\u212a".encode(), "\u017f".encode(), b"-"]
        hays += [b"".join(toks[int(i)] for i in rng.integers(0, len(toks), size=n)) for n in (40, 400, 2000)]
    seen=set()
    while len(seen) < 150:
        pat = "".join(atoms[int(rng.integers(0, len(atoms)))] for _ in range(int(rng.integers(1, 5))))
        if os.environ.get("FUZZ_FOLD") and "(?i" not in pat: continue
        if pat in seen: continue
        seen.add(pat)
        try: rx = cx.compile(pat)
        except cx.CoregexError as e:
            try: o = oracle.Regex(pat); print('HOST-REJECTS', repr(pat), e); bad+=1
            except oracle.OracleError: pass
            continue
        try: o = oracle.Regex(pat)
        except oracle.OracleError as e:
            if rx.supported: print('ORACLE-REJECTS', repr(pat), e); bad+=1
            continue
        if rx.strategy != o.strategy: print('STRATEGY', repr(pat), rx.strategy, o.strategy); bad+=1
        if not rx.supported:
            if rx.submatch_supported:
                pass
            else: continue
        tot+=1; strat[rx.strategy]=strat.get(rx.strategy,0)+1
        if rx.supported:
            blob = rx.blob(); kind = struct.unpack_from("<I", blob, 4)[0]; fl = struct.unpack_from("<I", blob, 8)[0]
            for hay in hays:
                exp = o.find_all_index(hay).tolist()
                if rx.strategy == 'UseBoth':     # a match longer than 100 bytes: the device refuses the haystack (CXG_E_INPUT, asserted by
                    plain = o.find_all_submatch_index(hay)[:, :2]   # tests/test_gpu_parity.py::test_use_both_programs); nothing to compare here
                    if len(plain) and int((plain[:, 1] - plain[:, 0]).max()) > 100:
                        n_long += 1
                        continue
                if kind == 5:                                # kKindFsmOnly (UseNFA, > 100 NFA states): the transducer twin
                    a = np.frombuffer(hay, dtype=np.uint8)
                    if rx.nullable == 2:                     # round 4: every match is empty, no device program
                        if emu.merge_empty_matches(np.zeros((0, 2), dtype=np.int64), len(hay)).tolist() != exp: print('NULL2', repr(pat), len(hay)); bad+=1; break
                        continue
                    got = emu.find_all_fsm(rx.fsm_image(), a, 3840, 32)
                    if isinstance(got, int) and got in (-18, -32): got = emu.find_all_fsm(rx.fsm_image(), a, 3840, 32, dense=1)
                    if rx.nullable and not isinstance(got, int): got = emu.merge_empty_matches(got, len(hay))   # the image is the non-empty variant's
                    if not isinstance(got, int) and got.tolist() != exp: print('FSM', repr(pat), rx.strategy, len(hay)); bad+=1; break
                    continue
                if kind == 3 and rx.strategy != 'UseCharClassSearcher':   # round 4: `C+` / quote-pair programs of the DFA strategies on the char-class kernels
                    g = emu.find_all_charclass_wave(blob, hay) if (fl & 64) else None
                    if g is None: n_cc_unchecked += 1
                    elif isinstance(g, int): n_cc_fallback += 1
                    elif g.tolist() != exp: print('CCW-RUNS', repr(pat), rx.strategy, len(hay)); bad+=1; break
                    continue
                if kind == 4 and struct.unpack_from("<I", blob, struct.unpack_from("<I", blob, 56)[0] + 44)[0]:   # literals between assertions (TeddyAux::looks): wave twin + the transducer fallback
                    g = emu.find_all_teddy_wave(blob, hay)
                    if g is not None and not isinstance(g, int) and g.tolist() != exp: print('WRAPPED', repr(pat), len(hay)); bad+=1; break
                    continue
                if rx.delimiters is not None:               # round 4: delimiter kernel in front of the images checked below
                    g = emu.find_all_delim(*rx.delimiters, np.frombuffer(hay, dtype=np.uint8))
                    if not isinstance(g, int) and g.tolist() != exp: print('DELIM', repr(pat), len(hay)); bad+=1; break
                if rx.strategy != 'UseCharClassSearcher':   # (scan_charclass.hip has no lane walk in walk.hpp: its wave twin is checked below)
                    got = emu.find_all(blob, hay).tolist()
                    if got != exp: print('LANES', repr(pat), rx.strategy, len(hay), len(got), len(exp)); bad+=1; break
                elif not (fl & 64):
                    n_cc_unchecked += 1
                if fl & 16 and rx.strategy in ('UseDFA','UseDigitPrefilter','UseBoth'):
                    for geom in ((192,64),(3840,256)):
                        g6 = emu.find_all_chain6_bounded(blob, rx.chain_bounds()[0], hay, *geom) if fl & 512 else emu.find_all_chain6(blob, hay, *geom)
                        if not isinstance(g6,int) and g6.tolist()!=exp: print('CHAIN6', repr(pat), len(hay), geom, len(g6), len(exp)); bad+=1
                if kind == 4 or (fl & 256):       # literal image (UseTeddy / one plain literal), or required literal prefix + anchored DFA
                    g = emu.find_all_teddy_wave(blob, hay)
                    if g is not None and not isinstance(g,int) and g.tolist()!=exp: print('TEDDYW', repr(pat), len(hay)); bad+=1
                if rx.strategy=='UseCharClassSearcher' and (fl & 64):
                    g = emu.find_all_charclass_wave(blob, hay)
                    if g is None: print('CCW-NO-TWIN', repr(pat)); bad+=1
                    elif isinstance(g,int): n_cc_fallback += 1
                    elif g.tolist()!=exp: print('CCW', repr(pat), len(hay)); bad+=1
        if "(" in pat and rx.submatch_supported:
            sb, cb = rx.submatch_blobs()[:2]
            w = 2*(o.num_groups if hasattr(o,'num_groups') else rx.num_groups)
            oc = rx.offset_captures
            for hay in hays[:6]:
                exp = o.find_all_submatch_index(hay)
                if oc is not None:                           # round 4: slots at fixed distances from the span's ends (FindAll + expansion kernel)
                    for k, (src, d) in enumerate(oc):
                        if not np.array_equal(exp[:, k], exp[:, 1 if src else 0] + d): print('OFFSET-CAPS', repr(pat), k, len(hay)); bad+=1; break
                    continue
                try: got = emu.find_all_submatch(sb, cb, hay, exp.shape[1])
                except AssertionError as e:
                    if 'error -4' in str(e): n_bt_limit += 1; break      # a long match of a pattern that is not one-pass: the backtracking pass runs out of stack and the call fails (kErrSerialLimit), INTEGRATION.md
                    print('SUBMATCH-TWIN', repr(pat), len(hay), e); bad+=1; break
                if got.shape!=exp.shape or not np.array_equal(got,exp): print('SUBMATCH', repr(pat), len(hay), got.shape, exp.shape); bad+=1; break
print('seeds', seed0, seed1, 'checked', tot, 'bad', bad, 'UseBoth-long-skipped', n_long, 'charclass-without-ranges-unchecked', n_cc_unchecked, 'charclass-wave-fallbacks', n_cc_fallback, 'captures-refused-long-match', n_bt_limit, strat, '%.0fs'%(time.time()-t0))
