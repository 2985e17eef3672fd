"""CPU fuzz of the FindAll transducer (host/fsm.cc tables + device/fsm.hpp lane functions run by tests/emu/emu_fsm.cc)
against the oracle: random patterns, small tile/chunk geometries, few-symbol haystacks.  python scripts/cpu_fuzz_fsm.py [n] [seed] [look|wide|lookwide]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import coregex_amd as cx
import emu
from oracle import oracle as O

ATOMS = ["a", "b", "c", "x", "y", r"\.", ":", "-", r"\d", "[a-c]", "[x-z]", r"\d+", "[a-c]+", "[x-z]+", "a+", r"\.+", "[0-4]+", "[5-9]",
         "ab|xy", "abc|xyz|a:c", r"\w", r"\w+", "[a-z0-9]+", "[ab]", "(a|b)", "(ab)+", "a?", r"\d{2}", r"\d{1,3}", "x*", "(xy|ab|ca)",
         "abcx|bcxy|cxyz|xyza", "z+", "abc", "xyz", "a:c", "(b*c)?", "(a|ab)", "(abc|ab|a)", "b*", "(ab*c|a|bb)", "a+?", "[ab]*?c",
         # (found by the device fuzz: a match created under a pending one that later grows — `(?:ab)*[ab]` on "abb")
         "(?:ab)*", "(?:a|b|c)+", "(?:ab)*[ab]", "(?:xy)*[x-z]", "(?:ab|a)*", "a*", r"\d{2,}", "(a+)(b+)", "b+?", "[a-c]+?", "(?:abc)*",
         "(a|b)*c", "(?:a+b)*", "(a?b)+", "(?:ab|a)(?:c|bc)", "x{2,3}", "(?:a{2})+", "[ab]{2}", "(?:x|xy)+z", "a*b", "(?:ab)+?c", "(?:a|ab|abc)", "(?:b|bc)*c",
         "(?:xy?)+", "[a-c]?[x-z]", "(?:a:)*c", r"(?:\d\.)*\d"]

LOOK_ATOMS = [r"\b", r"\B", r"\b", "_", "[a-c_]+", r"\w+", "ab", " ", "A", r"\d+", r"(a|\b)", r"(\bab|xy\b)", r"\b\b", r"(?:\bx)+",
              "^", "$", "^", "$", r"\n", r"(^a|b$)", r"[a-c\n]+", r"(?:$\n^)?", "^ab|xy$"]

# `.` and classes past U+007F (UTF-8 byte automata): mode "wide"; the haystacks then hold multi-byte sequences and stray bytes >= 0x80
WIDE_ATOMS = [".", ".", ".*", ".+", ".?", r"[^x]", r'[^"]', r"\S", r"\S+", r"\D", r"\W", r"[^a-c]+", r"[^\n]*", "(.)", r"(\S+)", r'"[^"]*"', "é", "[aé]", r"[^:]*:", ".+?", r"\D+?", "x.y", "(?s:.)"]

LEN_RANGE = tuple(int(v) for v in os.environ.get("FUZZ_ATOMS", "1,5").split(","))   # atoms per pattern: [lo, hi)

FOLD_ATOMS = ["(?i:error)", "(?i:warn)", "(?i:k)", "(?i:s1)", "(?i:ok)", "(?i:get)", "(?i:ab|xy)", "(?i:(abc))", "(?i:exception)", "(?i:a)b", "(?i:xyz)+", "(?i:[a-c])", "(?i:[x-z]+)", "(?i:a|b)c"]

def main(n=300, seed=1, look=False, wide=False):
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"abcxyz.:-0123456789 \n" + (b"_A  __" if look else b""), dtype=np.uint8)
    if wide:
        alphabet = np.frombuffer(b'abcxyz.:-019 \n"' + "éé日😀".encode() + b"\x80\xc3\xff", dtype=np.uint8)
    fold = bool(os.environ.get("FUZZ_FOLD"))    # case-insensitive literals in every pattern
    if fold: alphabet = np.frombuffer(b"abcxyzABCXYZ.:-01 \nerrorERRORwarnWARNkKsSokOKgetGET" + "\u212a\u017f".encode(), dtype=np.uint8)
    atoms = (ATOMS[:40] + FOLD_ATOMS * 3) if fold else (ATOMS + LOOK_ATOMS * 3 + (WIDE_ATOMS * 2 if wide else [])) if look else ATOMS + WIDE_ATOMS * 3 if wide else ATOMS
    n_strat = 0
    seen, n_img, n_checked, reasons = set(), 0, 0, {}
    n_caps = 0
    n_direct = 0
    t0 = time.time()
    while len(seen) < n:
        pat = "".join(atoms[int(rng.integers(0, len(atoms)))] for _ in range(int(rng.integers(*LEN_RANGE))))
        if pat in seen: continue
        if fold and "(?i" not in pat: continue
        if wide and not look and not any(a in pat for a in (".", "[^", "\\S", "\\D", "\\W", "é")): continue
        if look:
            if "\\b" not in pat and "\\B" not in pat and "^" not in pat and "$" not in pat: continue
            pat = "(?m)" + pat
        seen.add(pat)
        try: rx = cx.compile(pat)
        except cx.CoregexError: continue
        if look:                                                      # the front-end's strategy for word-boundary patterns == the oracle's
            os_ = O.Regex(pat)
            if os_.strategy_restated and rx.strategy != os_.strategy:
                print("STRATEGY", repr(pat), rx.strategy, os_.strategy)
                return 1
            n_strat += 1
        img = rx.fsm_image()
        simg = rx.fsm_image(True) if rx.num_groups > 1 else None
        cap_bt = None
        if simg is not None and rx.submatch_supported:
            cb = rx.submatch_blobs()[1]
            if cb[:4] == b"TBXC": cap_bt = cb                                 # cxgdev::kBtMagic "CXBT", little endian
        if img is None and simg is None: continue
        n_img += 1
        o = O.Regex(pat)
        hays = [alphabet[rng.integers(0, len(alphabet), size=int(k))] for k in (0, 1, 7, 63, 64, 65, 300, 2000)]
        hays += [alphabet[rng.integers(0, 6, size=700)], alphabet[rng.integers(0, 3, size=500)],
                 np.frombuffer(b"1.2.3.4.5.6.7.8.9 " * 30, dtype=np.uint8), np.frombuffer(b"abcabcabxyzxyz" * 40, dtype=np.uint8)]
        for hay in hays:
            for tile, chunk in ((3840, 32), (64, 8), (32, 4), (256, 16), (128, 32)):
                for which, image in (("idx", img), ("sub", simg)):
                    if image is None: continue
                    exp = o.find_all_index(hay) if which == "idx" else o.find_all_submatch_index(hay)[:, :2]
                    if which == "idx" and rx.strategy == "UseBoth":      # the reference restarts its PikeVM inside matches > 100 bytes: kernel raises CXG_E_INPUT
                        plain = o.find_all_submatch_index(hay)[:, :2]
                        if len(plain) and int((plain[:, 1] - plain[:, 0]).max()) > 100: continue
                    try:
                        got = emu.find_all_fsm(image, hay, tile, chunk)
                    except AssertionError as ex:
                        np.save("/tmp/fsm_fail_hay.npy", hay)
                        print("EMU ERROR", ex, repr(pat), rx.strategy, which, tile, chunk, bytes(hay[:120]), exp[:6].tolist())
                        return 1
                    if isinstance(got, int) and got in (-18, -32):       # a chunk's row / event buffers: the kernel's mode 2
                        try:
                            got = emu.find_all_fsm(image, hay, tile, chunk, dense=1)
                        except AssertionError as ex:
                            np.save("/tmp/fsm_fail_hay.npy", hay)
                            print("EMU ERROR (dense)", ex, repr(pat), which, tile, chunk)
                            return 1
                    if chunk == 32:                                      # round 6: the same rows through the byte-indexed tables (k_scan_fsmd's twin)
                        try:
                            gd = emu.find_all_fsm_direct(image, hay, tile)
                        except AssertionError as ex:
                            np.save("/tmp/fsm_fail_hay.npy", hay)
                            print("EMU ERROR (direct)", ex, repr(pat), rx.strategy, which, tile, bytes(hay[:120]))
                            return 1
                        if gd is not None:
                            n_direct += 1
                            if isinstance(gd, int): reasons[("direct", gd)] = reasons.get(("direct", gd), 0) + 1
                            else:
                                gd2 = emu.merge_empty_matches(gd, len(hay)) if (which == "idx" and rx.nullable) else gd
                                if gd2.shape != exp.shape or not np.array_equal(gd2, exp):
                                    np.save("/tmp/fsm_fail_hay.npy", hay)
                                    print("MISMATCH (direct)", repr(pat), rx.strategy, which, tile, bytes(hay[:120]), gd2[:6].tolist(), exp[:6].tolist())
                                    return 1
                    if isinstance(got, int):
                        reasons[got] = reasons.get(got, 0) + 1
                        continue
                    n_checked += 1
                    if which == "idx" and rx.nullable:                   # the image is the non-empty variant's: the empty matches are merged behind the scan
                        got = emu.merge_empty_matches(got, len(hay))
                    if got.shape != exp.shape or not np.array_equal(got, exp):
                        np.save("/tmp/fsm_fail_hay.npy", hay)
                        print("MISMATCH", repr(pat), rx.strategy, which, tile, chunk, bytes(hay[:120]), got[:6].tolist(), exp[:6].tolist())
                        return 1
                    if which == "sub" and look and tile == 3840 and cap_bt is not None:      # slots by the backtracking pass, assertions included
                        full = o.find_all_submatch_index(hay)
                        try:
                            caps = emu.captures_bt(cap_bt, hay, got, 2 * rx.num_groups)
                        except AssertionError as ex:
                            # a row longer than the visited bitmap allows (states x span > 65 536): the call fails loudly on the device
                            if "error -4" in str(ex) and len(full) and int((full[:, 1] - full[:, 0]).max() + 1) * rx.nfa_states > 65536: continue
                            print("CAPTURE ERROR", repr(pat), rx.strategy, rx.nfa_states, int((full[:, 1] - full[:, 0]).max()), ex)
                            return 1
                        n_caps += 1
                        if caps.shape != full.shape or not np.array_equal(caps, full):
                            np.save("/tmp/fsm_fail_hay.npy", hay)
                            print("CAPTURE MISMATCH", repr(pat), rx.strategy, bytes(hay[:120]), caps[:4].tolist(), full[:4].tolist())
                            return 1
    if look: print(f"{n_strat} strategies compared with the oracle")
    if look: print(f"{n_caps} capture-row comparisons (backtracking pass with assertions) clean")
    print(f"{len(seen)} patterns, {n_img} with a transducer image, {n_checked} comparisons clean ({n_direct} more through the direct tables), fallback reasons {reasons}, {time.time()-t0:.1f}s")
    return 0

if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1, len(sys.argv) > 3 and sys.argv[3] in ("look", "lookwide"), len(sys.argv) > 3 and sys.argv[3] in ("wide", "lookwide")))
