"""Text-start anchors inside unanchored patterns (`(^|,)\\d+`, `foo|\\Abar`: SURVEY a9, round 4).  The reference's lazy DFA has a
start state per kind of the byte in front of the search, Text among them (dfa/lazy/start.go:64-172); the transducer starts the scan
in a state closed with the anchor holding and its reverse walk accepts position 0 by a per-state flag (host/fsm.cc, fsm.hpp
fsm_match_start).  CPU tier: front-end strategy, build-time proofs and the transducer's sequential twin against the oracle."""
import random

import numpy as np
import pytest

import coregex_amd as cx
import emu

TEXT = [r"foo|^bar", r"\Afoo|bar", r"(?:^|,)\d+", r"(^|\s)error", r"(?:^|[^a-z])abc", r"^\d+|x\d+", r"(^|x)[a-c]+", r"(^|\s)(GET|POST)", r"(^|\s)(\w+)=(\d+)",
        r"\bfoo|^bar", r"(?:^|,)[a-c]+", r"(?m)(?:\A|^x)\d+", r"(?:^|:)\w+\b", r"(?:^|x)*a"]
# end-of-text anchors inside unanchored patterns (round 6): the kind "behind the haystack's last byte" of host/fsm.cc
TEXT_END = [r"foo$|x", r"x\z|foo", r"(?:^|,)\d+$", r"a$|z", r"(a$)b$", r"^a$|^b$", r"foo$|bar", r"\d+$|x", r"(?m)foo$|bar\z", r"\bfoo\z|ba+", r"(?:^|,)[a-c]+$|,", r"(\w+)=(\d+)\z|k=",
            r"error$|warn(ing)?"]
TEXT = TEXT + TEXT_END
REFUSED = [r"^foo", r"\Afoo", r"foo|^", r"a+$|a", r"\z(?:^|,)|[a-c]\w+", r"ab?|$"]


def _twin(rx, a):
    img = rx.fsm_image()
    got = emu.find_all_fsm(img, a, 3840, 32)
    if isinstance(got, int) and got in (-18, -32):
        got = emu.find_all_fsm(img, a, 3840, 32, dense=1)
    return got


@pytest.mark.parametrize("pat", TEXT)
def test_twin_equals_oracle(pat, oracle):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.strategy == o.strategy and rx.supported and rx.fsm_image() is not None, (pat, rx.strategy, o.strategy, rx.why_unsupported)
    rng = random.Random(len(pat) * 13)
    hays = [b"", b"bar", b"foo", b"barbar", b"bar foo bar", b"12,12 12", b"error error", b" error", b"abc abc", b"zabc", b"7x7", b"x7", b"abab", b"xab",
            b"GET /a POST", b"k=1 k=2", b"a=b=3", b"\nbar", b"12\n12", b"x1\nx2", b"w:w w", b"a" * 300 + b",b", b"c" * 5000 + b",a"]
    toks = [b"foo", b"bar", b"12", b",", b" ", b"error", b"abc", b"x7", b"ab", b"GET", b"POST", b"k=1", b"\n", b":", b"z", b"_"]
    for n in (50, 500, 3839, 3841, 9000):
        for lead in (b"", b"bar", b"12", b"abc", b"x9", b" ", b"GET", b"k=2", b"ab"):
            hays.append(lead + b"".join(rng.choice(toks) for _ in range(n // 3)))
    if pat in TEXT_END:                                                 # the haystack's LAST bytes matter: every tail at every distance from a tile / chunk edge
        hays += [b"a", b"za", b"ayyyyy", b"ab", b"b", b"xfoo", b"foo x foo", b",12", b"k=1 k=12", b"foo\n", b"error", b"warn error"]
        for n in (31, 32, 33, 63, 64, 65, 3776, 3839, 3840, 3841, 3904, 4031, 4032, 4033, 7679, 7680, 7681):
            for tail in (b"a", b"foo", b",12", b"ab", b"k=12", b"x", b"\n", b"bar", b",abc", b"error", b"za"):
                body = b"".join(rng.choice(toks) for _ in range(n))[:n]
                hays.append(body[: n - len(tail)] + tail)
    for hay in hays:
        a = np.frombuffer(hay, dtype=np.uint8)
        exp = o.find_all_index(a)
        got = _twin(rx, a)
        if isinstance(got, int):
            continue                                                    # a tile past the twin's row buffers (the kernel's denser modes)
        assert np.array_equal(got, exp), (pat, hay[:60], got[:4].tolist(), exp[:4].tolist())
        if rx.submatch_supported and rx.num_groups > 1 and len(hay) < 6000:
            es = o.find_all_submatch_index(a)
            oc = rx.offset_captures
            if oc is not None:
                for k, (src, d) in enumerate(oc):
                    assert np.array_equal(es[:, k], es[:, 1 if src else 0] + d), (pat, k)
            else:
                gs = emu.captures_bt(rx.submatch_blobs()[1], a, es[:, :2], es.shape[1])
                assert np.array_equal(gs, es), (pat, hay[:60], gs[:3].tolist(), es[:3].tolist())


@pytest.mark.parametrize("pat", REFUSED)
def test_anchored_quirky_and_nullable_patterns_stay_refused(pat, oracle):
    """Start-anchored patterns (other strategies), `a+$|a` (the reference's reverse DFA ignores the anchor and reports another start:
    lookdfa.cc proves it), a pattern that matches only the empty text, a nullable pattern with an assertion."""
    rx = cx.compile(pat)
    assert not rx.supported, pat
    assert rx.strategy == oracle.Regex(pat).strategy, pat


def test_reference_pairs_on_the_twin(oracle):
    """tests/golden "text_anchor_compat": the reference's own differential pairs with a text-start anchor inside an alternation
    (edge_cases_test.go:262-290,320), expected spans by Python re; front-end, proofs and transducer twin."""
    import json, os
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))
    served = 0
    for c in vec["text_anchor_compat"]["cases"]:
        rx = cx.compile(c["pattern"])
        assert rx.strategy == oracle.Regex(c["pattern"]).strategy, c
        if not rx.supported:
            continue
        served += 1
        got = _twin(rx, np.frombuffer(c["input"].encode(), dtype=np.uint8))
        assert not isinstance(got, int) and got.tolist() == c["want"], (c, got)
    assert served >= 5
    # the pairs with an end-of-text anchor (oracle-only until round 6): `a$|z`, `(a$)b$`, `^a$|^b$` are UseDFA programs and served; the
    # others are the reference's reverse / anchored strategies (out of scope) or nullable
    served_end = 0
    for c in vec["text_anchor_compat_oracle_only"]["cases"]:
        rx = cx.compile(c["pattern"])
        assert rx.strategy == oracle.Regex(c["pattern"]).strategy, c
        if not rx.supported:
            assert rx.strategy in ("UseReverseAnchored", "UseBoundedBacktracker") or "nullable" in rx.why_unsupported, (c, rx.why_unsupported)
            continue
        served_end += 1
        got = _twin(rx, np.frombuffer(c["input"].encode(), dtype=np.uint8))
        assert not isinstance(got, int) and got.tolist() == c["want"], (c, got)
    assert served_end == 5, served_end
