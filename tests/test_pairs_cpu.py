"""`Q[^Q]*Q` programs (`"[^"]*"`: round 4): recognised on the anchored DFA (program.cc isQuotePairs), served by the char-class wave
kernel's pairs mode.  CPU tier: the kernel's sequential twin against the oracle; what is NOT such a program stays where it was."""
import random
import struct

import numpy as np
import pytest

import coregex_amd as cx
import emu

PAIRS = [r'"[^"]*"', r"'[^']*'", r"\|[^|]*\|", r'"[^"]*?"', r"x[^x]*x", r"`[^`]*`"]


def _kind(rx):
    return struct.unpack_from("<I", rx.blob(), 4)[0]


@pytest.mark.parametrize("pat", PAIRS)
def test_twin_equals_oracle(pat, oracle):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.supported and rx.strategy == o.strategy == "UseDFA" and _kind(rx) == 3
    q = pat[1] if pat[0] == "\\" else pat[0]
    rng = random.Random(ord(q))
    alpha = (q + "ab \n").encode() + "é".encode() + b"\xff"
    for n in [0, 1, 2, 3, 64, 3839, 3840, 3841, 7681, 20000, 70000]:
        for w in (1, 6, 40):
            hay = np.frombuffer(bytes(rng.choices(alpha, weights=[w] + [10] * (len(alpha) - 1), k=n)), dtype=np.uint8)
            got = emu.find_all_charclass_wave(rx.blob(), hay)
            if isinstance(got, int):
                assert w == 40 and got == -24                        # more than 1024 occurrences in a tile: CXG_E_INPUT on the device
                continue
            assert np.array_equal(got, o.find_all_index(hay)), (pat, n, w)


@pytest.mark.parametrize("pat", [r'"[^"]+"', r'"[a-z]*"', r'"[^"]*"x', r'a"[^"]*"', r'"[^"\n]*"', r'("[^"]*")+'])
def test_other_shapes_are_not_pairs(pat):
    rx = cx.compile(pat)
    if rx.supported:
        blob = rx.blob()
        assert _kind(rx) != 3 or not struct.unpack_from("<I", blob, struct.unpack_from("<I", blob, 9 * 4)[0] + 16)[0]
