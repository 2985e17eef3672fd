"""GPU tier, last file on purpose: (?i) programs were added after the round's GPU minutes were spent, so this is their FIRST device run
(the driver's `pytest -x` stops at a failure; nothing that has passed on a device before comes after this file)."""
import numpy as np
import pytest

import coregex_amd as cx
from refcorpus import generate_test_input

pytestmark = pytest.mark.gpu


def _u8(b):
    return b if isinstance(b, np.ndarray) else np.frombuffer(bytes(b), dtype=np.uint8)


FOLD = [r"(?i)error", r"(?i)hello", r"(?i)(?:login|auth|sign.?in|session)", r"(?i)(error|fail|exception|panic|fatal)", r"(?i)warn(ing)?", r"x(?i:yz)w", r"(?i)k1",
        r"(?i)error: \w+", r"GET (?i:/index)", r"(?i)(googlebot|bingbot|yandexbot)"]


@pytest.mark.parametrize("pat", FOLD)
def test_case_insensitive_literals(oracle, pat):
    """(?i): the literal sets are the reference's case variants (literal/extractor.go:838-941, restated late in round 3; `K` and `S`
    fold to U+212A and U+017F as well), the programs Teddy sets, literal prefixes + DFA, or the transducer — kernels that ran
    before with other tables.  NOT run on a device by the builder (the round's GPU minutes were spent): first device run is the
    driver's."""
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.supported and rx.strategy == o.strategy, (pat, rx.why_unsupported)
    rng = np.random.default_rng(3)
    toks = [w for base in (b"error", b"hello", b"login", b"auth", b"signin", b"sign-in", b"session", b"fail", b"panic", b"fatal", b"exception", b"warn", b"warning",
                           b"xyzw", b"k1", b"googlebot", b"bingbot", b"GET /index", b"error: disk")
            for w in (base, base.upper(), base.capitalize(), base[:1] + base[1:].upper())] + [b" ", b"\n", b": ", "\u212a1".encode(), "\u017fession".encode()]
    words = b"".join(toks[int(i)] for i in rng.integers(0, len(toks), size=60000))
    for hay in (generate_test_input(), cx.synth_pages(3, 0xC0FFEE03, 0, 128), _u8(words), _u8(b""), _u8(b"ERROR")):
        exp = o.find_all_index(hay)
        got = rx.find_all_index(hay)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay), got[:3].tolist(), exp[:3].tolist())
        assert rx.count(hay) == len(exp)
    if rx.num_groups > 1 and rx.submatch_supported:
        hay = _u8(words[:100000])
        assert np.array_equal(rx.find_all_submatch_index(hay), o.find_all_submatch_index(hay)), pat


BOUNDED = [r"\S+", r"[^,]+", r"[0-9a-f]{32}", r"[a-f0-9]{32,}", r"(\w{2,8})+", r"\d{3}", r"[a-c]{2,3}?", r"(?i)[a-z]+", r"\w", r"\d{1,3}", r"[ab]", r"[^ ]+", r"\d"]


@pytest.mark.parametrize("pat", BOUNDED)
def test_bounded_backtracker_programs(oracle, pat):
    """Unanchored UseBoundedBacktracker programs — concatenations / repetitions of character classes — are answered by the reference
    with two leftmost-first engines (its backtracker while the rest of the haystack fits 32 M visited entries, the lazy-DFA pair
    before that): served as the DFA pair's program since the end of round 3 (program.cc), with the existing kernels.  NOT run on a
    device by the builder: first device run is the driver's."""
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.strategy == o.strategy == "UseBoundedBacktracker" and o.strategy_restated
    if not rx.supported:
        pytest.skip(rx.why_unsupported)
    rng = np.random.default_rng(5)
    mixed = np.frombuffer(b"abcdef0123456789 ,.\n-_XYZ" + "é日".encode() + b"\x80\xff", dtype=np.uint8)
    for hay in (generate_test_input(), cx.synth_pages(2, 0xC0FFEE02, 0, 64), mixed[rng.integers(0, len(mixed), size=50000)], mixed[rng.integers(0, 6, size=20000)], _u8(b""), _u8(b"a")):
        exp = o.find_all_index(hay)
        try:
            got = rx.find_all_index(hay)
        except cx.UnsupportedInput:
            continue                                        # (a run beyond the serial-walk budget of a program without a transducer image)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay), got[:3].tolist(), exp[:3].tolist())
        assert rx.count(hay) == len(exp)
