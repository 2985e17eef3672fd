"""CPU fuzz of bounded-repetition programs (not part of pytest): lane walks and the BND twin of the chain kernel vs the oracle.
Usage: python scripts/cpu_fuzz_bounded.py SEED0 SEED1"""
import sys, struct
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import coregex_amd as cx, emu
from oracle import oracle
atoms=[r"\d{1,3}", r"\d{2,4}", r"\d+", r"\d{1,2}", r"\d{2,}", r"[a-c]{1,3}", r"[a-c]{2,4}", r"[a-c]+", r"\.", ":", "-", r"(?:\.\d{1,3}){2}", r"(?:\.\d{1,3}){3}", r"(?::\d{2}){2}", r"\d{3}", "x"]
alph=np.frombuffer(b"0123456789..::--abcx ", dtype=np.uint8)
tot=0; bnd=0; bad=0; exact=0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng=np.random.default_rng(seed)
    hays=[alph[rng.choice(len(alph), size=int(n), p=rng.dirichlet(0.4*np.ones(len(alph))))].tobytes() for n in (30, 300, 3000, 3000, 5000)]
    hays += [b"10.0.0.1 192.168.1.254 1.2.3 12:34:56 2026-09-27 1234.5.6.7 1.2.3.4567 abc-12 aa:bb "*40]
    seen=set()
    while len(seen)<120:
        pat="".join(atoms[int(rng.integers(0,len(atoms)))] for _ in range(int(rng.integers(2,6))))
        if pat in seen: continue
        seen.add(pat)
        try: rx=cx.compile(pat); o=oracle.Regex(pat)
        except Exception: continue
        if rx.strategy!=o.strategy: print('STRATEGY', pat); bad+=1
        if not rx.supported: continue
        tot+=1
        b=rx.chain_bounds()
        for hay in hays:
            exp=o.find_all_index(hay).tolist()
            if emu.find_all(rx.blob(), hay).tolist()!=exp: print('LANES', repr(pat), len(hay)); bad+=1; break
            if b is not None:
                g=emu.find_all_chain6_bounded(rx.blob(), b[0], hay)
                if not isinstance(g,int):
                    exact+=1
                    if g.tolist()!=exp: print('BND', repr(pat), b[1], len(hay)); bad+=1; break
        bnd += b is not None
print('programs', tot, 'bounded', bnd, 'exact twin runs', exact, 'bad', bad)
