"""GPU tier, last file on purpose: look-around programs that came in after this round's GPU budget was spent.  Their device path
is the transducer kernel's look-around instantiation that tests/test_gpu_fsm.py validates; each (program, haystack) pair below
was checked on the kernel's sequential twin (tests/emu) against the oracle before it was written down here."""
import numpy as np
import pytest

import coregex_amd as cx
from refcorpus import generate_test_input

pytestmark = pytest.mark.gpu

WIDER = [(r"\d+\.\d+\.\d+\.\d+\b", "UseDigitPrefilter"), (r"\d{4}-\d{2}-\d{2}\b", "UseDigitPrefilter"),      # SearchAtAnchored of the look-aware lazy DFA at each digit
         (r"timeout=\d+\b ms elapsed", "UseDFA"), (r"(GET|POST|PUT|DELETE)\b /[a-z/]+ HTTP", "UseDFA")]     # byte classes that mix word and non-word bytes, harmlessly


@pytest.mark.parametrize("pat,strategy", WIDER)
def test_look_programs_proved_at_build_time(oracle, pat, strategy):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.strategy == o.strategy == strategy and o.strategy_restated and rx.supported, (pat, rx.strategy, o.strategy, rx.why_unsupported)
    line = b"user=bob ip=10 status=ok  a=b;c=d  going moved bars  a b c d! xuser=a ip=b status=c_ k=v;w=x\n"
    hays = [generate_test_input(), cx.synth_pages(2, 0xC0FFEE02, 0, 256).tobytes(), b"", line * 2000,
            b"10.0.0.1 10.0.0.1x 1.2.3.4.5 2024-01-02 2024-01-023 timeout=30 ms elapsed GET /a HTTP GETX /a HTTP " * 300]
    for hay in hays:
        exp = o.find_all_index(hay)
        got = rx.find_all_index(hay)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay))
        assert rx.count(hay) == len(exp)


LOOK_CAPTURES = [r"\b(\w+)=(\w+)\b", r"(?m)^(\d+) (\w+)", r"(\d+)\.(\d+)\b", r"\b(DEBUG|INFO|WARN|ERROR)\b", r"(?m)^(GET|POST) "]


@pytest.mark.parametrize("pat", LOOK_CAPTURES)
def test_captures_of_look_programs(oracle, pat):
    """FindAllSubmatchIndex with assertions: spans by the look-around transducer, slots by the backtracking pass (bt.hpp LOOK states).
    The reference's FindAllSubmatch of these strategies is its PikeVM (meta/findall.go:89-98): served even where FindAllIndex is not."""
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.submatch_supported and rx.strategy == o.strategy, (pat, rx.strategy, o.strategy)
    hays = [generate_test_input(), b"", b"a=b c=d; e=f\n12 ab\nGET /x\nPOST y\n3.4 5.6x me@x.com ERROR error warn_ warn\n7 z\nkey: val\n", b"k=v", b"x=y\n" * 3000,
            b"12 ab", b"1.2"]
    for hay in hays:
        exp = o.find_all_submatch_index(hay)
        got = rx.find_all_submatch_index(hay)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay))


def test_more_lookaround_golden_rows_on_the_device():
    """tests/golden "lookaround_compat_more": word-boundary pairs of the reference's differential tests (edge_cases_test.go:245-249,
    346-347) with spans by Python re — device rows against the fixture, no oracle in between."""
    import json, os, re as pyre
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))
    for c in vec["lookaround_compat_more"]["cases"]:
        rx = cx.compile(c["pattern"])
        assert rx.supported, (c["pattern"], rx.why_unsupported)
        hay = c["input"].encode()
        assert rx.find_all_index(hay).tolist() == c["want"], c
        big = (hay + b"\n") * 3000
        assert rx.find_all_index(big).tolist() == [[m.start(), m.end()] for m in pyre.finditer(c["pattern"].encode(), big)], c["pattern"]


def test_constructor_programs_admitted_late(oracle):
    """cxg_program_from_nfa for the look-around programs of UseDFA / UseBoth in tests/test_boundary.py NFA_PATTERNS (same device
    image as cxg_compile's, checked on the CPU tier)."""
    import test_boundary as B
    corpus = cx.synth_pages(2, 0xC0FFEE02, 0, 64).tobytes() + b" user=a ip=b status=c k=v;w=x a=b;c=d_ GET /a/b HTTP/1.1 k=12 " * 50
    for pat in B.ADMITTED_LATE:
        eng, prog = B.via_constructor(pat)
        assert prog.supported, (pat, prog.why_unsupported)
        assert np.array_equal(prog.find_all_index(corpus), oracle.Regex(pat).find_all_index(corpus)), pat
