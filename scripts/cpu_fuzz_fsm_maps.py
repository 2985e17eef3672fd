#!/usr/bin/env python3
"""CPU fuzz of the transducer kernel's map compositions (round 3: tests/emu/emu_fsm.cc emu_fsm_maps_check) — random patterns
(general and look-around), few-symbol haystacks with islands, the kernel's tile and tiny tiles / groups: every entry state, tile
exit and group entry the compositions yield against a plain left-to-right walk.  python scripts/cpu_fuzz_fsm_maps.py [n] [seed] [look]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import coregex_amd as cx, emu
import cpu_fuzz_fsm as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
look = len(sys.argv) > 3 and sys.argv[3] == "look"
rng = np.random.default_rng(seed)
atoms = F.ATOMS + F.LOOK_ATOMS * 3 if look else F.ATOMS
few = [b"ab", b"abc", b"ab ", b"xyz", b"1.", b"a:c", b"abx\n", b"01 .", b"a", b"x_ "]
seen, images, checked, unlisted, bad = set(), 0, 0, 0, 0
while len(seen) < n:
    pat = "".join(atoms[int(rng.integers(0, len(atoms)))] for _ in range(int(rng.integers(1, 5))))
    if pat in seen: continue
    if look:
        if "\\b" not in pat and "\\B" not in pat and "^" not in pat and "$" not in pat: continue
        pat = "(?m)" + pat
    seen.add(pat)
    try:
        rx = cx.compile(pat)
        img = rx.fsm_image()
    except cx.CoregexError:
        continue
    if img is None: continue
    images += 1
    for _ in range(6):
        f = np.frombuffer(few[int(rng.integers(0, len(few)))], dtype=np.uint8)
        m = int(rng.choice([500, 4000, 9000, 40000]))
        hay = f[rng.integers(0, len(f), size=m)].copy()
        for _ in range(int(rng.integers(0, 3))):
            at = int(rng.integers(0, max(1, m - 100))); ln = int(rng.choice([5, 60]))
            hay[at:at + ln] = np.frombuffer((b"GET /x 10.0.0.1 error k=v " * 4)[:ln], dtype=np.uint8)[: len(hay[at:at + ln])]
        for tile, tpg in ((3840, 32), (128, 3), (64, 1), (256, 8)):
            r = emu.fsm_maps_check(bytes(img), hay, tile, tpg)
            if r == -17: unlisted += 1
            elif r < 0:
                bad += 1; print("MISMATCH", repr(pat), r, tile, tpg, bytes(hay[:60]))
            else: checked += r
print(f"seed {seed} patterns {n} look {look} images {images} sub-chunk entries checked {checked} unlisted-set runs {unlisted} bad {bad}")
