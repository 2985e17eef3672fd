"""GPU tier: the general-DFA kernel (scan_fsm.hip, FindAll transducer) against the oracle, through the C ABI."""
import numpy as np
import pytest

import coregex_amd as cx
from routing import routed
from coregex_amd import _lib
from refcorpus import COMPAT_PATTERNS, generate_test_input

pytestmark = pytest.mark.gpu

K_FSM = (10, 19, 20)   # CXG_K_FSM, CXG_K_FSM_DIRECT (round 6: the same machine through byte-indexed rows), CXG_K_FSM_LEAN (the lean kernel)


@pytest.fixture(autouse=True)
def _literals_between_assertions_stay_on_the_transducer(monkeypatch):
    """Round 4 serves `\\berror\\b`-style programs by the literal kernel (tests/test_gpu_wrapped.py); this file is the transducer
    kernel's tier and keeps them there."""
    monkeypatch.setenv("CXG_NO_WRAPPED_LITERALS", "1")
README_IP = r"(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)"
GENERAL = [README_IP, COMPAT_PATTERNS["la_peak_hours"], r"\d+\.\d+x?", r"a+b|b+a", r"ab*c|a|bb", r"a[0-9]*b|a\.", r"(foobar|foo)\d*",
           r"[1-9][0-9]*|0", r"a(b*c)?", r"x[ab]+?y", r"[a-f0-9]{8}-[a-f0-9]{4}", r"ab|abc", r"GET|POST /[a-z]+", r"[a-c]x|[b-d]y",
           r"(?:ab)*[ab]", r"(?:ab)*[a-c]"]      # a match created under a pending one that later grows: event-list path (device fuzz, seed 73)


FSM_EXPECTED = {README_IP, COMPAT_PATTERNS["la_peak_hours"], r"\d+\.\d+x?", r"a+b|b+a", r"[a-c]x|[b-d]y"}


def _device_rows(rx, hay, sub=False):
    import torch
    a = np.ascontiguousarray(np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else hay)
    n = (a.size + 15) // 16 * 16
    buf = cx.DeviceBuffer(max(n, 16))
    if a.size:
        buf.upload(a)
    scan = rx.find_all_submatch_device if sub else rx.find_all_device
    w = 2 * rx.num_groups if sub else 2
    cnt = scan(buf.ptr, a.size)
    out = torch.empty((cnt + 4, w), dtype=torch.int64, device="cuda")
    t = cx.Timing()
    assert scan(buf.ptr, a.size, out.data_ptr(), cnt + 4, timing=t) == cnt
    return out[:cnt].cpu().numpy(), t


@pytest.mark.parametrize("pat", GENERAL)
def test_general_dfas_run_on_the_transducer_kernel(oracle, pat):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.supported and rx.fsm_image() is not None, (pat, rx.why_unsupported)
    rng = np.random.default_rng(len(pat))
    alphabet = np.frombuffer(b"abcfoxy.:-0123456789 %=/GETPOS\n", dtype=np.uint8)
    hays = [generate_test_input(), cx.synth_pages(2, 0xC0FFEE02, 3, 300), b"", b"a", b"ab", b"abbbbbc abbbbbd bb",
            alphabet[rng.integers(0, len(alphabet), size=200000)], alphabet[rng.integers(0, 8, size=50000)],
            np.frombuffer((b"foobar12 foo3 foob " * 9000), dtype=np.uint8)]
    for hay in hays:
        exp = o.find_all_index(hay)
        got = rx.find_all_index(hay)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay))
        assert rx.count(hay) == len(exp)
    rows, t = _device_rows(rx, hays[1])
    assert np.array_equal(rows, o.find_all_index(hays[1]))
    # one launch of the transducer kernel for the programs that have no faster kernel and whose matches are not denser
    # than a chunk's row / event buffers on this corpus (every number or every 'a' of a log line is too dense: those
    # hand over to the table-walking kernel — same rows, checked above)
    if pat in FSM_EXPECTED:
        assert routed(t.kernel in K_FSM and t.n_launches == 1, t.kernel, t.n_launches, t.fallback_reason), (pat, t.kernel, t.n_launches, t.fallback_reason)


def test_transducer_kernel_edges(oracle):
    """Matches across chunk, wave-tile and group edges; matches that end with the input; rows whose start lies in the
    previous tile / group (the unbounded reverse walk + the cross-tile and cross-group checks)."""
    group = 3840 * 32
    pat = r"a[0-9]*b|a\."
    rx, o = cx.compile(pat), oracle.Regex(pat)
    lit = np.frombuffer(b"a0123456789012345678901234567890123456789b", dtype=np.uint8)
    base = np.full(2 * group + 5000, ord(" "), dtype=np.uint8)
    offs = list(range(3840 - 45, 3840 + 3, 3)) + list(range(4096 - 45, 4096 + 3, 5)) + list(range(20, 70, 7)) + list(range(group - 45, group + 3, 3)) + [2 * 3840 - 1, group + 3840 - 6]
    for off in offs:
        hay = base.copy()
        hay[off:off + len(lit)] = lit
        hay[off + len(lit) + 1:off + len(lit) + 3] = np.frombuffer(b"a.", dtype=np.uint8)
        exp = o.find_all_index(hay)
        assert np.array_equal(rx.find_all_index(hay), exp), off
        rows, t = _device_rows(rx, hay)
        assert np.array_equal(rows, exp) and routed(t.kernel in K_FSM, t.kernel), off
    for n in (1, 63, 64, 65, 3839, 3840, 3841, 4095, 4096, 4097, group - 1, group, group + 1):
        rep = (b"a12b a. ab x" * (n // 12 + 2))[:n]
        assert np.array_equal(rx.find_all_index(rep), o.find_all_index(rep)), n
        tail = np.full(n, ord(" "), dtype=np.uint8)
        tail[max(0, n - 4):] = np.frombuffer(b"a12b", dtype=np.uint8)[-min(4, n):]
        assert np.array_equal(rx.find_all_index(tail), o.find_all_index(tail)), n
    # a second match found by a search that started inside the unbounded reverse walk of the first row of a tile:
    # `ax|x?b+` on "...axbb": [.., ax] then [bb]; the reverse DFA from the second end accepts "xbb" — the bound must cut it
    pat2 = r"ax|x?b+"
    rx2, o2 = cx.compile(pat2), oracle.Regex(pat2)
    assert rx2.supported and rx2.fsm_image() is not None
    for off in (3837, 3838, 3839, 3840, 3841, group - 3, group - 2, group - 1, group, group + 1):
        hay = base.copy()
        hay[off - 1:off + 3] = np.frombuffer(b"axbb", dtype=np.uint8)
        exp = o2.find_all_index(hay)
        assert len(exp) == 2 and exp[1][0] == exp[0][1]
        rows, t = _device_rows(rx2, hay)
        assert np.array_equal(rows, exp) and routed(t.kernel in K_FSM, t.kernel), (off, rows.tolist(), exp.tolist())
    # matches longer than the 64 bytes staged in front of a tile / the 192 bytes behind it: the start is finished in the
    # epilogue from HBM, a walk past the window's end hands the scan over — the oracle's rows either way
    long1 = np.frombuffer(b"a" + b"7" * 150 + b"b", dtype=np.uint8)
    long2 = np.frombuffer(b"a" + b"7" * 700 + b"b", dtype=np.uint8)
    for lit in (long1, long2):
        for off in (3840 - 100, 3840 - 10, 3840 - 200, group - 100, group - 500, 100):
            hay = base.copy()
            hay[off:off + len(lit)] = lit
            exp = o.find_all_index(hay)
            assert len(exp) == 1
            rows, t = _device_rows(rx, hay)
            assert np.array_equal(rows, exp), (len(lit), off, rows.tolist(), exp.tolist())


def test_captures_take_their_spans_from_the_transducer_kernel(oracle):
    hay = cx.synth_pages(5, 0xC0FFEE05, 0, 256).tobytes() + b" k=12 key=7;x=y GET /a/b POST /c " * 2000
    for pat in (r"(GET|POST|PUT) /([a-z/]+)", r"([a-z]+)=(\d+|[a-z])", r"(\d+)(\.\d+)?"):
        rx, o = cx.compile(pat), oracle.Regex(pat)
        assert rx.submatch_supported, pat
        exp = o.find_all_submatch_index(hay)
        got = rx.find_all_submatch_index(hay)
        assert got.shape == exp.shape and np.array_equal(got, exp), pat


def test_dense_and_unconverged_input_falls_back_exactly(oracle):
    """More than 8 matches in a 64-byte chunk, or a chunk whose entry state does not collapse (few-symbol input without
    synchronising bytes): the kernel raises its flag and the table-walking kernel answers — same rows."""
    pat = r"\d+\.\d+x?"
    rx, o = cx.compile(pat), oracle.Regex(pat)
    for hay in (b"1.2 " * 20000, b"1.1" * 9000, b"1.2.3.4.5.6.7.8.9 " * 500):
        assert np.array_equal(rx.find_all_index(hay), o.find_all_index(hay))


@pytest.mark.parametrize("pat,unit", [(README_IP, b"1."), (r"\d+\.\d+x?", b"1."), (r"\d+\.\d+\.\d+\.\d+", b"12."), (r"\b\d+\.\d+\b", b"1.")])
def test_input_without_synchronising_structure(oracle, pat, unit):
    """`1.1.1.1...`: no tile's first set of possible states collapses, every group leaves a map and finds its entry by the
    look-back over maps (scan_fsm.hip fsm_group_entry) — 137 groups here, more than two look-back windows; islands of other
    text give some groups a known exit (the walk ends there), the tail of the input is a partial group.  Rows == oracle."""
    n = 16 * (1 << 20) + 12345
    body = bytearray((unit * (n // len(unit) + 1))[:n])
    rng = np.random.default_rng(len(pat))
    for at in rng.integers(0, n - 5000, 9):           # islands: a stretch of text with synchronising bytes, some a whole group long
        ln = int(rng.choice([7, 300, 5000, 130000]))
        body[at:at + ln] = (b"GET /index.html 10.0.0.1 x " * (ln // 27 + 1))[:ln]
    hay = np.frombuffer(bytes(body), dtype=np.uint8)
    rx, o = cx.compile(pat), oracle.Regex(pat)
    rows, t = _device_rows(rx, hay)
    exp = o.find_all_index(hay)
    assert rows.shape == exp.shape and np.array_equal(rows, exp), (t.kernel, t.n_launches, t.fallback_reason)
    assert routed(t.kernel in K_FSM, t.kernel)


# ---- word boundaries (\b \B): UseNFA in the reference (PikeVM, nfa/pikevm.go:1646-1674), the transducer kernel here with
# (class of this byte, kind of the next byte) as input symbol (fsm.hpp "Look-around").  No table-walking fallback exists.
LOOK = [r"\berror\b", r"\b\d+\b", r"\bGET\b", r"\b(GET|PUT)\b", r"\Berror", r"error\B", r"\b[A-Z]+\b", r"ab(a|\b)", r"\b\d+\.\d+\b", r"\berror\w*",
        r"\b[a-z]+\b", r"\b0x[0-9a-f]+\b"]


@pytest.mark.parametrize("pat", LOOK)
def test_word_boundary_programs(oracle, pat):
    rx = cx.compile(pat)
    o = oracle.Regex(pat)
    assert o.strategy == "UseNFA" and o.strategy_restated and rx.strategy == "UseNFA", (pat, o.strategy, rx.strategy)
    assert rx.supported, rx.why_unsupported
    hays = [generate_test_input()]
    hays += [cx.synth_pages(c, 0xC0FFEE00 + c, 3, 256).tobytes() for c in (1, 2, 3)]
    hays += [b"", b"error", b"error ", b" error", b"xerror error_ error", b"GET", b"12", b"a", b"POST /x GET", b"aba ab ab_ abab", b"deadbeef 0deadbeef deadbeef0 12345678"]
    for hay in hays:
        exp = o.find_all_index(hay)
        got = rx.find_all_index(hay)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay))
        assert rx.count(hay) == len(exp)
    rows, t = _device_rows(rx, hays[1])
    assert np.array_equal(rows, o.find_all_index(hays[1]))
    assert routed(t.kernel in K_FSM, t.kernel), (pat, t.kernel)


# ---- assertions inside the reference's lazy-DFA strategies: served when host/lookdfa.cc proves that the reference's look-aware
# lazy DFA answers leftmost-first, history-free; the oracle answers with that DFA, restated.
LOOK_DFA = [(r"\b\w+\s+\w+\s+\w+\b", "UseBoth"), (r"\buser=\w+ ip=\w+ status=\w+\b", "UseDFA"), (r"\b\w+=\w+;\w+=\w+\b", "UseBoth"), (r"\w+\b \w+\b \w+\b \w+\b!", "UseBoth")]


@pytest.mark.parametrize("pat,strategy", LOOK_DFA)
def test_look_programs_of_lazy_dfa_strategies(oracle, pat, strategy):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.strategy == o.strategy == strategy and o.strategy_restated and rx.supported, (pat, rx.strategy, o.strategy, rx.why_unsupported)
    line = b"user=bob ip=10 status=ok  a=b;c=d  going moved bars  a b c d! xuser=a ip=b status=c_ k=v;w=x\n"
    hays = [generate_test_input(), b"", line * 4000, b"a b c", b"a=b;c=d", b" " * 150 + line + b"." * 130 + b"q r s t!", b"aaa" + b" " * 120 + b"bbb ccc " + line]
    for hay in hays:
        exp = o.find_all_index(hay)                                   # ONE oracle engine over all haystacks: no cache-history effect
        plain = o.find_all_submatch_index(hay)[:, :2]                 # (PikeVM spans: plain leftmost-first)
        if strategy == "UseBoth" and len(plain) and int((plain[:, 1] - plain[:, 0]).max()) > 100:
            with pytest.raises(cx.CoregexError): rx.find_all_index(hay)   # the 100-byte restart span: CXG_E_INPUT, the caller keeps its CPU loop
            continue
        got = rx.find_all_index(hay)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay))
        assert rx.count(hay) == len(exp)
    rows, t = _device_rows(rx, hays[2])
    assert np.array_equal(rows, o.find_all_index(hays[2])) and routed(t.kernel in K_FSM, t.kernel), (pat, t.kernel)
    assert not cx.compile(r"\b(DEBUG|INFO|WARN|ERROR)\b").supported          # byte class mixes word and non-word bytes: history-dependent reference


def test_word_boundary_edges(oracle):
    """The assertion reads the byte on either side of a position: word / non-word neighbours across chunk (32 / 64 B),
    wave-tile (3840 B), window (4032 B) and group (120 KiB) edges, at the haystack's first and last byte, and for inputs that
    end exactly on such an edge (the byte behind the end counts as "not a word byte")."""
    group = 3840 * 32
    for pat in (r"\berror\b", r"\Berror", r"error\B", r"\b[a-z]+\b"):
        rx, o = cx.compile(pat), oracle.Regex(pat)
        base = np.full(2 * group + 5000, ord(" "), dtype=np.uint8)
        offs = [0, 1, 27, 31, 32, 59, 63, 64, 3835, 3839, 3840, 3841, 4027, 4031, 4032, 4091, 4096, group - 5, group - 1, group, group + 1, 2 * group - 3]
        for off in offs:
            for lit in (b"error", b"xerror", b"errorx", b"_error_", b"error error"):
                hay = base.copy()
                a = np.frombuffer(lit, dtype=np.uint8)
                hay[off:off + len(a)] = a
                exp = o.find_all_index(hay)
                rows, t = _device_rows(rx, hay)
                assert np.array_equal(rows, exp) and routed(t.kernel in K_FSM, t.kernel), (pat, off, lit, rows.tolist(), exp.tolist())
        for n in (1, 5, 6, 31, 32, 33, 63, 64, 65, 3839, 3840, 3841, 4031, 4032, 4033, 4095, 4096, 4097, group - 1, group, group + 1):
            for tail in (b"error", b"xerror", b" error", b"errorx"):
                hay = np.full(n, ord(" "), dtype=np.uint8)
                k = min(len(tail), n)
                hay[n - k:] = np.frombuffer(tail, dtype=np.uint8)[len(tail) - k:]
                assert np.array_equal(rx.find_all_index(hay), o.find_all_index(hay)), (pat, n, tail)
                head = np.full(n, ord(" "), dtype=np.uint8)
                head[:k] = np.frombuffer(tail, dtype=np.uint8)[:k]
                assert np.array_equal(rx.find_all_index(head), o.find_all_index(head)), (pat, n, tail, "head")


def test_word_boundary_long_words_and_budgets(oracle):
    """A word longer than the window: `\\b[a-z]+\\b` creates its match only where the closing assertion holds, the start
    is then found from HBM in the epilogue (exact).  `\\berror\\w*` grows by a rematch per byte while the word goes on (served since
    round 6); `\\berror(?:z*!)?` keeps a match pending WITHOUT events: past the window that is the kernel's walk budget and, with no
    table-walking image to fall back to, CXG_E_INPUT."""
    group = 3840 * 32
    rx, o = cx.compile(r"\b[a-z]+\b"), oracle.Regex(r"\b[a-z]+\b")
    for n, off in ((150, 3800), (700, 3000), (5000, 100), (70000, group - 1000)):
        hay = np.full(2 * group + 5000, ord(" "), dtype=np.uint8)
        hay[off:off + n] = ord("q")
        hay[off + n + 1:off + n + 4] = np.frombuffer(b"abc", dtype=np.uint8)
        exp = o.find_all_index(hay)
        assert len(exp) == 2
        assert np.array_equal(rx.find_all_index(hay), exp), (n, off)
    rx2, o2 = cx.compile(r"\berror\w*"), oracle.Regex(r"\berror\w*")
    hay = np.full(20000, ord(" "), dtype=np.uint8)
    hay[100:105] = np.frombuffer(b"error", dtype=np.uint8)
    hay[105:165] = ord("z")
    assert np.array_equal(rx2.find_all_index(hay), o2.find_all_index(hay))          # 65 bytes: inside the window
    hay[105:5000] = ord("z")
    # round 6: a row belongs to the tile its END lies in and every `z` is a rematch event, so the growing match is simply handed on from
    # tile to tile (its start then comes from HBM in the epilogue) — served, where rounds 2-5 ran out of their walk budget
    assert np.array_equal(rx2.find_all_index(hay), o2.find_all_index(hay))
    # what still is a budget: a match that stays PENDING without events past the window's end (`!` may still come: threads alive, nothing to report)
    rx3, o3 = cx.compile(r"\berror(?:z*!)?"), oracle.Regex(r"\berror(?:z*!)?")
    assert np.array_equal(rx3.find_all_index(hay[:165]), o3.find_all_index(hay[:165]))
    with pytest.raises(cx.CoregexError) as ei:
        rx3.find_all_index(hay)
    assert ei.value.code == _lib.CXG_E_INPUT
    assert np.array_equal(rx3.find_all_index(hay[:90]), o3.find_all_index(hay[:90]))  # the program stays usable


def test_word_boundary_program_from_nfa(oracle):
    """The cgo shim's route: nfa.State with StateLook (cxg_nfa_state.lo = nfa.Look) through cxg_program_from_nfa."""
    pat = r"\bGET\b"
    src = cx.compile(pat)
    nfa, keep = cx.flatten_nfa(src.nfa())
    prog = cx.program_from_nfa(nfa, "UseNFA", 0)
    assert prog.supported, prog.why_unsupported
    hay = cx.synth_pages(1, 0xC0FFEE01, 0, 64).tobytes() + b" GET POSTS xGET GET_ POST"
    assert np.array_equal(prog.find_all_index(hay), oracle.Regex(pat).find_all_index(hay))
    # larger patterns with word boundaries are UseDFA / UseBoth in the reference (its lazy DFA carries the look state,
    # dfa/lazy/builder.go:183-242 — not restated, not served): refused at build time, the caller keeps its CPU loop
    big = cx.compile(r"\b(GET|POST|PUT)\b")
    assert big.strategy == "UseDFA" and not big.supported


# ---- multi-line anchors (?m)^ (?m)$: a third kind of neighbour byte (newline); the positions around the haystack read as a
# line edge (the kernel writes '\n' into its window there)
LINE = [r"(?m)^line", r"(?m)error$", r"(?m)^\w+$", r"(?m)[a-z]+$", r"(?m)^GET|POST$", r"(?m)^\d+", r"(?m)^line\b"]
LINE_TEDDY = [r"(?m)^(GET|POST|PUT|DELETE|PATCH)", r"(?m)^GET|^POST|^PUT"]


@pytest.mark.parametrize("pat", LINE + LINE_TEDDY)
def test_multiline_anchor_programs(oracle, pat):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert o.strategy_restated and rx.strategy == o.strategy and o.strategy == ("UseTeddy" if pat in LINE_TEDDY else "UseNFA"), (pat, o.strategy, rx.strategy)
    assert rx.supported, rx.why_unsupported
    hays = [generate_test_input()] + [cx.synth_pages(c, 0xC0FFEE00 + c, 3, 256).tobytes() for c in (1, 2, 3)]
    hays += [b"", b"line", b"line\n", b"\nline", b"error", b"error\n", b"x error\nerror x\nerror", b"GET /\nPOST\n GET\nPOST", b"12\n34 56\n\n78", b"\n\n\n", b"a\nb\n"]
    for hay in hays:
        exp = o.find_all_index(hay)
        got = rx.find_all_index(hay)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay), hay[:40])
        assert rx.count(hay) == len(exp)
    rows, t = _device_rows(rx, hays[1])
    assert np.array_equal(rows, o.find_all_index(hays[1])) and routed(t.kernel in K_FSM, t.kernel)


def test_multiline_anchor_edges(oracle):
    """Line starts and ends across chunk / wave-tile / window / group edges, at the first and last byte of the haystack,
    and for inputs that end exactly on such an edge."""
    group = 3840 * 32
    for pat in (r"(?m)^line", r"(?m)error$", r"(?m)^\w+$"):
        rx, o = cx.compile(pat), oracle.Regex(pat)
        base = np.full(2 * group + 5000, ord(" "), dtype=np.uint8)
        offs = [0, 1, 27, 31, 32, 59, 63, 64, 3835, 3839, 3840, 3841, 4027, 4031, 4032, 4091, 4096, group - 5, group - 1, group, group + 1, 2 * group - 3]
        for off in offs:
            for lit in (b"\nline\n", b"line\n", b"\nerror\n", b"error\n", b"\nerror x", b"\nline error\n", b"\n\nline\n\n"):
                hay = base.copy()
                a = np.frombuffer(lit, dtype=np.uint8)
                hay[off:off + len(a)] = a
                exp = o.find_all_index(hay)
                rows, t = _device_rows(rx, hay)
                assert np.array_equal(rows, exp) and routed(t.kernel in K_FSM, t.kernel), (pat, off, lit, rows.tolist(), exp.tolist())
        for n in (1, 4, 5, 6, 31, 32, 33, 63, 64, 65, 3839, 3840, 3841, 4031, 4032, 4033, 4095, 4096, 4097, group - 1, group, group + 1):
            for tail in (b"line", b"\nline", b"error", b" error", b"\nerror"):
                hay = np.full(n, ord(" "), dtype=np.uint8)
                k = min(len(tail), n)
                hay[n - k:] = np.frombuffer(tail, dtype=np.uint8)[len(tail) - k:]
                assert np.array_equal(rx.find_all_index(hay), o.find_all_index(hay)), (pat, n, tail)
                head = np.full(n, ord(" "), dtype=np.uint8)
                head[:k] = np.frombuffer(tail, dtype=np.uint8)[:k]
                assert np.array_equal(rx.find_all_index(head), o.find_all_index(head)), (pat, n, tail, "head")


def test_lookaround_golden_rows_on_the_device():
    """tests/golden "lookaround_compat": the reference's own differential pairs for (?m)^ (?m)$ \\b \\B with spans computed by
    Python re — device rows against the fixture, no oracle in between."""
    import json, os
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))
    for c in vec["lookaround_compat"]["cases"]:
        rx = cx.compile(c["pattern"])
        assert rx.supported, (c["pattern"], rx.why_unsupported)
        hay = c["input"].encode()
        assert rx.find_all_index(hay).tolist() == c["want"], c
        big = (hay + b"\n") * 3000                                   # the same lines repeated past several wave-tiles
        exp = [[s + k * (len(hay) + 1), e + k * (len(hay) + 1)] for k in range(3000) for s, e in c["want"]]
        import re as pyre
        assert exp == [[m.start(), m.end()] for m in pyre.finditer(c["pattern"].encode(), big)], "fixture arithmetic"
        got = rx.find_all_index(big)
        assert got.tolist() == exp, c["pattern"]
