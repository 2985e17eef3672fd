"""CPU fuzz of the alphabet-run image (host/lookdfa.cc buildRunsImage, device/runs.hpp): random patterns over small alphabets — digits,
dots, a few letters, with and without assertions — through the kernel's sequential twin (tests/emu) against the oracle.
python scripts/cpu_fuzz_runs.py [n_patterns] [seed]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import coregex_amd as cx
import emu
from oracle import oracle as O

ATOMS = [r"\d", r"\d+", r"\d+", r"\d{1,3}", r"\d{2}", r"[0-4]", r"[5-9]+", r"\.", r"\.", ":", "x", "ab", "a+", "[ab]+", r"(?:\d|x)", r"(?:25[0-5]|1?\d)", r"(?:\.\d+)+", r"(?:\.\d+)?", r"\d+?", r"\d*",
         r"\b", r"\b", r"\B", "^", "$", r"(?:ab|a)", r"(?:a|ab)", r"(\d+)", r"x?", r"[0-9a-f]+", "0x", "-"]

def main(n=300, seed=1):
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"0123456789..::xab-f _\n  zZ", dtype=np.uint8)
    words = [b"12", b"3.4", b"1.2.3.4", b"255.255.0.1", b"256.1", b"007 ", b"ab", b"aab", b"x1", b"0x1f", b"12:30", b" ", b"\n", b"_", b"-", b"x", b".", b"..", b"1" * 40, b"1.1.1.1.1.1.1.1.1.1"]
    seen, n_img, n_cmp, n_give, strat, why = set(), 0, 0, 0, {}, {}
    t0 = time.time()
    tries = 0
    while n_img < n and tries < n * 200:
        tries += 1
        pat = "".join(ATOMS[int(rng.integers(0, len(ATOMS)))] for _ in range(int(rng.integers(2, 7))))
        if pat in seen: continue
        seen.add(pat)
        if rng.random() < 0.5: pat = "(?m)" + pat
        try: o = O.Regex(pat)
        except O.OracleError: continue
        try: rx = cx.compile(pat)
        except cx.CoregexError: continue
        if not rx.supported: continue
        img = rx.runs_image()
        if img is None: continue
        n_img += 1
        strat[rx.strategy] = strat.get(rx.strategy, 0) + 1
        hays = []
        for k in (0, 1, 2, 7, 40, 300, 4000, 9000):
            hays.append(alphabet[rng.integers(0, len(alphabet), size=k)])
            parts = [words[int(rng.integers(0, len(words)))] if rng.random() < 0.6 else bytes(alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(1, 4)))]) for _ in range(k // 3 + 1)]
            hays.append(np.frombuffer(b"".join(parts), dtype=np.uint8))
        for hay in hays:
            exp = o.find_all_index(hay)
            for tile in (3840, 64):
                got = emu.find_all_runs(img, hay, tile, 1 << 20, 1 << 20)
                if isinstance(got, int):
                    n_give += 1
                    continue
                n_cmp += 1
                if got.shape != exp.shape or not np.array_equal(got, exp):
                    print("MISMATCH", repr(pat), rx.strategy, tile, bytes(hay[:120]), got[:6].tolist(), exp[:6].tolist())
                    return 1
    print(f"{n_img} programs with an alphabet-run image {strat}, {n_cmp} comparisons clean, {n_give} give-ups (runs > 255 bytes), {time.time()-t0:.1f}s")
    return 0

if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
