"""GPU tier: `.` and classes that reach past U+007F (UTF-8 byte automata, nfa/compile.go:440-1222) through the C ABI, against the
oracle: log-shaped patterns on the synthetic corpus and on text with multi-byte sequences and stray bytes >= 0x80."""
import numpy as np
import pytest

import coregex_amd as cx
from refcorpus import generate_test_input

pytestmark = pytest.mark.gpu

SPANS = [r'"[^"]*"', r"GET .* HTTP", r"\[[^\]]+\]", r"/[^ ]+\.html", r"user=\S+", r"\d+ .* \d+", r"a.c", r"<[^>]+>", r"https?://[^\s]+", r"(?s)a.b", r"é+", r"k[^k]",
         r"HTTP/1\.1. \d+", r"session_id=[^ ]+ ", r"x.y|a.b", r"/\*.*?\*/", r"ms=\d+."]
CAPS = [r"user=(\S+)", r'"([^"]*)"', r"\[([^\]]+)\]", r"ms=(\d+)(.)", r'"(GET|POST) ([^ ]+) HTTP', r"(a)(.)(c)"]


def _u8(b):
    return b if isinstance(b, np.ndarray) else np.frombuffer(bytes(b), dtype=np.uint8)


def _hays():
    rng = np.random.default_rng(7)
    mixed = np.frombuffer(b'abckxy.:-019 \n"<>[]/=*' + "éé日😀".encode() + b"\x80\xc3\xff", dtype=np.uint8)
    return [cx.synth_pages(2, 0xC0FFEE02, 0, 256), generate_test_input(), mixed[rng.integers(0, len(mixed), size=120000)],
            mixed[rng.integers(0, 9, size=30000)], _u8('a😀c x日y "é" <ü> aéc a\nc [日本語] /*é*/ https://ü.example/é k😀'.encode() * 50), _u8(b""), _u8(b"a"), _u8("é".encode())]


@pytest.mark.parametrize("pat", SPANS)
def test_spans(oracle, pat):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.supported and rx.strategy == o.strategy, (pat, rx.why_unsupported)
    for hay in _hays():
        exp = o.find_all_index(hay)
        if rx.strategy == "UseBoth":
            plain = o.find_all_submatch_index(hay)[:, :2]
            if len(plain) and int((plain[:, 1] - plain[:, 0]).max()) > 100 and len(plain) > 64 * 50:
                continue                                    # more restarts than the device wrapper makes (CXG_E_INPUT)
        got = rx.find_all_index(hay)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay), got[:3].tolist(), exp[:3].tolist())
        assert rx.count(hay) == len(exp)


@pytest.mark.parametrize("pat", CAPS)
def test_capture_rows(oracle, pat):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.submatch_supported, pat
    for hay in _hays()[:1] + _hays()[2:]:
        hay = hay[:200000]
        exp = o.find_all_submatch_index(hay)
        got = rx.find_all_submatch_index(hay)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay), got[:2].tolist(), exp[:2].tolist())


def test_dot_alone_and_every_byte_value(oracle):
    """`.` / `.+` on the reference's own rows (regex_unicode_test.go:112-141) and over every byte value: ill-formed bytes match alone,
    lead bytes C2..F4 without their continuation do not match at all."""
    for pat, text, want in [(".", "日本語", 3), (".+", "日本語", 1), (".", "😀😁", 2), (".", "a日b", 3), (".", "Привет", 6), (".", "a\nb", 2), ("(?s).", "日\n本", 3)]:
        rx = cx.compile(pat)
        assert rx.supported, (pat, rx.why_unsupported)
        assert len(rx.find_all_index(text.encode())) == want, (pat, text)
    hay = _u8(bytes(range(256)) * 3 + "é日😀".encode() + bytes([0xC3, 0x41, 0xE6, 0x97, 0x41, 0xF0, 0x9F, 0x41]))
    for pat in [".", ".+", r"[^a]", r"\S+x?", r"[^\n]y?"]:
        rx, o = cx.compile(pat), oracle.Regex(pat)
        if not rx.supported or rx.strategy == "UseBoundedBacktracker":     # (that strategy's first device run: tests/test_zzz_gpu_fold.py)
            continue
        assert np.array_equal(rx.find_all_index(hay), o.find_all_index(hay)), pat


def test_a_match_longer_than_the_serial_walk_budget_is_refused_not_truncated(oracle):
    """`[^a-c]+` runs across newlines: one match can be the whole haystack.  Up to the serial-walk budget (128 KiB) the UseBoth
    restart rule answers (the reference's PikeVM starts 100 bytes before the end); past it the table-walking kernel cuts its walk
    and flags it — the call must then fail (CXG_E_INPUT) instead of handing the cut row to the restart loop (found by the device
    fuzz, `(.)[^a-c]+` on 171 000 bytes: rows [147612, 171000) instead of [170900, 171000))."""
    for pat in (r"(.)[^a-c]+", r"[^a-c]+x?y?"):
        rx, o = cx.compile(pat), oracle.Regex(pat)
        assert rx.supported and rx.strategy == "UseBoth"
        for n in (200, 3000, 100000, 130000, 171000, 400000):
            hay = _u8(((b"1.2.3.4 " * 7 + b"\n") * 8000)[:n])
            exp = o.find_all_index(hay)
            try:
                got = rx.find_all_index(hay)
            except cx.UnsupportedInput:
                assert n > 128 * 1024, n
                continue
            assert np.array_equal(got, exp), (pat, n, got.tolist(), exp.tolist())


def test_long_capture_rows(oracle):
    """Patterns with `\\S` / `[^"]` are not one-pass (a byte >= 0x80 is accepted by several NFA states): their slots come from the
    backtracking pass.  Its stack stays flat over a long repetition (dead alternatives are not pushed, bt.hpp); what bounds a
    row is the visited bitmap, states x span <= 65 536 — longer rows fail the call (CXG_E_INPUT), never a wrong slot."""
    for pat, hay in [(r'"([^"]*)"', b'x "' + b"z" * 1200 + b'" "a" "' + "é日".encode() * 100 + b'"'),
                     (r"user=(\S+)", b"user=" + b"a" * 700 + b" user=bob\nuser=\xff\xfe z"),
                     (r"(xy|ab|ca)\S+", b"ab" + b"q" * 900 + b" xyzz ca\xc3\xa9"),
                     (r"user=(\S+)", b"x user=" + b"a" * 5000 + b" y user=bob z")]:
        rx, o = cx.compile(pat), oracle.Regex(pat)
        exp = o.find_all_submatch_index(hay)
        try:
            got = rx.find_all_submatch_index(hay)
        except cx.UnsupportedInput:
            assert int((exp[:, 1] - exp[:, 0]).max()) * 40 > 65536, pat
            continue
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, got.tolist(), exp.tolist())

