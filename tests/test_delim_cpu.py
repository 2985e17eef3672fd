"""`O [^E]+ E` / `O [^E]* E` programs (`\\[[^\\]]+\\]`, `<[^>]+>`: round 4): recognised on the anchored DFA (program.cc isDelimited),
served by scan_delim_wave.hip in front of the transducer.  CPU tier: the kernel's sequential twin against the oracle."""
import random

import numpy as np
import pytest

import coregex_amd as cx
import emu

DELIM = [(r"\[[^\]]+\]", "[", "]", True), (r"<[^>]+>", "<", ">", True), (r"\([^)]*\)", "(", ")", False), (r"\{[^}]+\}", "{", "}", True),
         (r"<[^>]*>", "<", ">", False), (r"a[^b]+b", "a", "b", True)]


@pytest.mark.parametrize("pat,o,c,plus", DELIM)
def test_twin_equals_oracle(pat, o, c, plus, oracle):
    rx, orc = cx.compile(pat), oracle.Regex(pat)
    assert rx.supported and rx.strategy == orc.strategy == "UseDFA"
    assert rx.delimiters == (ord(o), ord(c), plus), (pat, rx.delimiters)
    rng = random.Random(ord(o) * 7 + plus)
    alpha = (o + c + "xy \n").encode() + "é".encode() + b"\xff"
    hays = [b"", o.encode(), c.encode(), (o + c).encode(), (o + "x" + c).encode(), (o + o + c + c).encode(), (o + c + "x" + c).encode(), (o + o + c).encode()]
    for n in [5, 64, 3839, 3840, 3841, 7681, 20000, 70000]:
        for wo, wc in ((1, 1), (6, 1), (1, 6), (1, 0), (0, 1), (12, 12)):
            w = [wo, wc] + [10] * (len(alpha) - 2)
            hays.append(bytes(rng.choices(alpha, weights=w, k=n)))
    hays.append(o.encode() + b"z" * 30000 + c.encode() + b"  " + o.encode() + b"y" * 9000)          # a row over many tiles, an opening without its E
    for hay in hays:
        a = np.frombuffer(hay, dtype=np.uint8)
        exp = orc.find_all_index(a)
        for tile in (3840, 64):
            got = emu.find_all_delim(ord(o), ord(c), plus, a, tile)
            if isinstance(got, int):
                assert got == -24 and tile == 3840
                continue
            assert np.array_equal(got, exp), (pat, len(hay), tile, hay[:50], got[:4].tolist(), exp[:4].tolist())


@pytest.mark.parametrize("pat", [r"\[[^\]]+\]x", r"\[[a-z]+\]", r"\[[^\]]{2,}\]", r'"[^"]*"', r"x\[[^\]]+\]", r"\[[^\]\n]+\]", r"\[[^\]]+?\]"])
def test_other_shapes_are_not_delimited(pat):
    rx = cx.compile(pat)
    assert not rx.supported or rx.delimiters is None or pat.endswith("+?\\]")
