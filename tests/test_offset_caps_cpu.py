"""Offset captures (round 4): program.cc deriveOffsetCaps says, per capture slot, "match start + d" or "match end - d" when the slot
sits on the split-free chain behind the pattern's start or in front of its Match.  Here: whatever it says must be what the
oracle's FindAllSubmatch reports on random haystacks — for the patterns it takes, and it must decline where a boundary moves."""
import random

import numpy as np
import pytest

import coregex_amd as cx

TAKEN = [r"user=(\S+)", r'"([^"]*)"', r"\[([^\]]+)\]", r"(\w+)", r"x(a)b", r"(a|b)c", r"(ab)(cd)e+", r"e+(ab)(cd)", r"<(\w+)>", r"k=(\d+);",
         r"((a)b)", r"(\d+)", r"id=(\d+)", r"(GET|POST) /", r"\((\w+)\)", r"(?:ab)+(c)d", r"(a*)b"]
DECLINED = [r"(a)+", r"(?:(a)b)+", r"(a){2}", r"(\w+)@(\w+)\.(\w+)", r"(a+)(b+)", r"(a)(b)?c", r"(?:x|(y))z", r"(a)|b", r"x(a)?"]


@pytest.mark.parametrize("pat", TAKEN)
def test_offsets_are_the_oracles_slots(pat, oracle):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    oc = rx.offset_captures
    assert rx.supported and oc is not None and len(oc) == 2 * rx.num_groups and rx.submatch_supported, (pat, oc)
    rng = random.Random(len(pat))
    alpha = b'user= "ab[]<>()xy@.k1;cdeGETPOS/id\n'
    rows = 0
    for _ in range(60):
        raw = bytes(rng.choice(alpha) for _ in range(rng.choice([5, 60, 800, 5000])))
        raw += rng.choice([b"", b' user=abc "q" [z] <w> k=12; xab abcdee eeabcd id=7 GET / (v) ababcd ']) + raw[:20]
        hay = np.frombuffer(raw, dtype=np.uint8)
        sub = o.find_all_submatch_index(hay)
        rows += len(sub)
        for k, (src, d) in enumerate(oc):
            assert np.array_equal(sub[:, k], sub[:, 1 if src else 0] + d), (pat, k, bytes(hay[:40]))
    assert rows > 0, pat


@pytest.mark.parametrize("pat", DECLINED)
def test_moving_boundaries_are_declined(pat):
    rx = cx.compile(pat)
    assert rx.offset_captures is None, (pat, rx.offset_captures)
