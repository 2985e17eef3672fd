"""CPU tier for the FindAll transducer (host/fsm.cc tables + the lane functions of device/fsm.hpp, run sequentially by
tests/emu/emu_fsm.cc): rows == oracle for general DFAs, inputs without synchronising bytes and word-boundary programs,
over several tile / chunk geometries.  The GPU tier (test_gpu_fsm.py) runs the same functions inside scan_fsm.hip."""
import os
import sys

import numpy as np
import pytest

import coregex_amd as cx
import emu
from refcorpus import generate_test_input

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))

GEOMS = ((3840, 32), (256, 16), (64, 8), (32, 4), (128, 32))
GENERAL = [r"\d+\.\d+x?", r"a+b|b+a", r"ab*c|a|bb", r"a[0-9]*b|a\.", r"(foobar|foo)\d*", r"[1-9][0-9]*|0", r"x[ab]+?y", r"ab|abc", r"[a-c]x|[b-d]y",
           r"\d+\.\d+\.\d+\.\d+", r"error|warning|fatal", r"ax|x?b+", r"(?:ab)*[ab]", r"(?:ab)*[a-c]", r"(?:xy)*[x-z]",
           r"(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)"]
LOOK = [r"\berror\b", r"\b\d+\b", r"\bGET\b", r"\b(GET|PUT)\b", r"\Berror", r"error\B", r"\b[A-Z]+\b", r"ab(a|\b)", r"(ab)+(a|\b)\b\b", r"\b\d+\.\d+\b",
        r"\berror\w*", r"\b[a-z]+\b", r"\b0x[0-9a-f]+\b", r"\w+\b", r"a\B", r"(?:\bx)+", r"x\b|\By", r"\b_+\b",
        # multi-line anchors: a third kind of neighbour byte (newline); outside the haystack counts as a line edge
        r"(?m)^line", r"(?m)error$", r"(?m)^\w+$", r"(?m)^line\b", r"(?m)[a-z]+$", r"(?m)^GET|POST$", r"(?m)^\d+", r"(?m)\n^a", r"(?m)a$\n^b"]
LOOK_TEDDY = [r"(?m)^(GET|POST|PUT|DELETE|PATCH)", r"(?m)^(GET|PUT)", r"(?m)^GET|^POST|^PUT"]   # UseTeddy behind prefilter.WrapLineAnchor


def _check(oracle, pat, hays, look):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.supported, (pat, rx.why_unsupported)
    img = rx.fsm_image()
    assert img is not None, pat
    if look:
        assert o.strategy == look and o.strategy_restated and rx.strategy == look, (pat, o.strategy, rx.strategy)
    n_ok = 0
    for hay in hays:
        exp = o.find_all_index(hay)
        for tile, chunk in GEOMS:
            got = emu.find_all_fsm(img, hay, tile, chunk)
            if isinstance(got, int) and got in (-18, -32):           # a chunk's row / event buffers: the kernel's mode 2
                got = emu.find_all_fsm(img, hay, tile, chunk, dense=1)
            if isinstance(got, int):
                assert got in (-17, -18, -20, -24, -32), (pat, tile, chunk, got)   # fallback reasons the kernel raises too
                continue
            assert got.shape == exp.shape and np.array_equal(got, exp), (pat, tile, chunk, len(hay))
            n_ok += 1
            if chunk == 32:                                          # round 6: the same rows through the byte-indexed tables (k_scan_fsmd's twin)
                gd = emu.find_all_fsm_direct(img, hay, tile)
                if gd is not None and not isinstance(gd, int):
                    assert gd.shape == exp.shape and np.array_equal(gd, exp), (pat, tile, "direct", len(hay))
                elif gd is not None:
                    assert gd in (-17, -24), (pat, tile, "direct", gd)
    assert n_ok >= len(hays), (pat, n_ok)                               # most geometries must actually answer


def _hays(seed):
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"abcxyz_A.:-0123456789  \n", dtype=np.uint8)
    hays = [generate_test_input(), cx.synth_pages(2, 0xC0FFEE02, 3, 8).tobytes(), cx.synth_pages(1, 0xC0FFEE01, 5, 8).tobytes(),
            b"", b"a", b"error", b" error", b"error ", b"xerror error_ error", b"GET /a PUT x GETS", b"1.2.3.4.5.6.7.8.9 " * 40,
            b"abcabcabxyzxyz" * 50, b"0x1f 0xZZ x0x12 0x12_ 0xab"]
    hays += [alpha[rng.integers(0, len(alpha), size=int(k))].tobytes() for k in (7, 63, 64, 65, 300, 3000)]
    hays += [alpha[rng.integers(0, 5, size=900)].tobytes(), alpha[rng.integers(6, 12, size=700)].tobytes()]
    return hays


@pytest.mark.parametrize("pat", GENERAL)
def test_transducer_twin_general(oracle, pat):
    _check(oracle, pat, _hays(len(pat)), look=False)


@pytest.mark.parametrize("pat", LOOK)
def test_transducer_twin_word_boundaries(oracle, pat):
    """\\b / \\B: the step over a byte also reads the kind of the next one (fsm.hpp "Look-around"); the reverse automaton is
    built look-aware on the host; the byte on either side of the haystack counts as "not a word byte"."""
    _check(oracle, pat, _hays(100 + len(pat)), look="UseNFA")


@pytest.mark.parametrize("pat", LOOK_TEDDY)
def test_transducer_twin_line_anchored_literals(oracle, pat):
    """`(?m)^(GET|POST|...)`: UseTeddy in the reference, its candidates filtered by a line-start check
    (prefilter.WrapLineAnchor) and, for literals of different lengths, the PikeVM from the first one that passes — the
    pattern's leftmost-first match, which the transducer computes (no literal kernel involved)."""
    hays = _hays(7) + [b"GET /a\nPOST /b\n GET\nPUTS\nDELETE\nPATCH x GET\nGET", b"\nGET\n\nPUT\n"]
    _check(oracle, pat, hays, look="UseTeddy")

def test_lookaround_golden_rows_on_the_twin():
    """The look-around rows of tests/golden (the reference's differential pairs, spans by Python re) through the product's
    front-end, transducer tables and lane functions — no oracle in between."""
    import json
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))
    n = 0
    for c in vec["lookaround_compat"]["cases"] + vec["lookaround_compat_more"]["cases"]:
        rx = cx.compile(c["pattern"])
        assert rx.supported, (c["pattern"], rx.why_unsupported)
        for tile, chunk in ((3840, 32), (64, 8), (32, 4)):
            got = emu.find_all_fsm(rx.fsm_image(), c["input"].encode(), tile, chunk, dense=1)
            assert not isinstance(got, int) and got.tolist() == c["want"], (c, tile, chunk)
            n += 1
    assert n >= 45


def test_match_created_under_a_pending_match_that_later_grows(oracle):
    """`(?:ab)*[ab]` on "abb": [0,1) is pending (the thread inside `ab` lives), [1,2) is created under it — with no threads of
    its own, so the stack stays one level deep — then the pending match grows to [0,3) and the row under it must go.  The
    two-bitmap row derivation cannot express that: such machines carry depth 2 and take the event-list path (found by the
    device fuzz, seed 73)."""
    for pat in (r"(?:ab)*[ab]", r"(?:ab)*[a-c]"):
        rx, o = cx.compile(pat), oracle.Regex(pat)
        img = rx.fsm_image()
        assert np.frombuffer(img[:32], dtype=np.uint32)[7] >= 2          # FsmHeader::depth
        for hay in (b"2x00xa8z\x7fX\xa9y8abbzc\x7fyxccbz2cyxazx", b"abb" + b" " * 29, b" " * 29 + b"abb", b"ababb" * 13, (b"xabb ab abb ababa " * 20)[:320]):
            for geom in GEOMS:
                got = emu.find_all_fsm(img, hay, *geom, dense=1)
                assert not isinstance(got, int) and got.tolist() == o.find_all_index(hay).tolist(), (pat, hay, geom)
    for pat in (r"\d+\.\d+x?", r"a+b|b+a", r"\d+\.\d+\.\d+\.\d+", r"\berror\b"):     # these stay on the two-bitmap path
        assert np.frombuffer(cx.compile(pat).fsm_image()[:32], dtype=np.uint32)[7] <= 1


def test_word_boundary_scope(oracle):
    """Served: UseNFA programs (small patterns; PikeVM semantics in the reference).  Refused at build time: line / text
    anchors, nullable patterns, and those of the larger patterns the reference gives to its look-aware lazy DFA (UseDFA / UseBoth)
    for which that DFA's answer depends on cache history or is not leftmost-first (host/lookdfa.cc; the served ones are below)."""
    for pat, frag in ((r"^error", "anchor"), (r"error$", "anchor"), (r"\b", "nullable"), (r"(?m)^", "nullable"), (r"\b(GET|POST|PUT)\b", "depends on cache history"),
                      (r"\b[a-f0-9]{8}\b", "depends on cache history"), (r"(?m)^foo|barr", "some alternatives only"), (r"(?m)\d+$", "depends on cache history")):
        rx = cx.compile(pat)
        assert not rx.supported and frag in rx.why_unsupported, (pat, rx.why_unsupported)
    img = cx.compile(r"\berror\b").fsm_image()
    hdr = np.frombuffer(img[:4 * 40], dtype=np.uint32)
    assert hdr[0] == 0x43584736                                        # "CXG6"


def test_transducer_fuzz_smoke():
    import cpu_fuzz_fsm
    assert cpu_fuzz_fsm.main(120, 20260927) == 0
    assert cpu_fuzz_fsm.main(250, 20260928, look=True) == 0


# ---- look-around inside the reference's lazy-DFA strategies (host/lookdfa.cc): served only when the build-time proof holds
LOOK_DFA_OK = [(r"status=\w+?\b; code", "UseDFA"),                                       # non-greedy: no reverse DFA, the DFA's IsMatchAt gates the PikeVM
               (r"\d+\.\d+\.\d+\.\d+\b", "UseDigitPrefilter"), (r"\d{4}-\d{2}-\d{2}\b", "UseDigitPrefilter"), (r"\d+\b ms", "UseDigitPrefilter"),   # SearchAtAnchored at each digit
               (r"timeout=\d+\b ms elapsed", "UseDFA"), (r"(GET|POST|PUT|DELETE)\b /[a-z/]+ HTTP", "UseDFA"), (r"[a-z]+=\d+\b; [a-z]+=\d+\b", "UseBoth"),   # classes that mix word and non-word bytes, harmlessly
               (r"\b[\w.]+@[\w.]+\.(com|org|net)\b", "UseBoth"), (r"\bfoo=\w+;bar=\w+\b", "UseDFA"), (r"\b\w+\s+\w+\s+\w+\b", "UseBoth"), (r"\buser=\w+ ip=\w+ status=\w+\b", "UseDFA"), (r"\b\w+=\w+;\w+=\w+\b", "UseBoth"),
               (r"\b\w+@\w+\.\w+\.com\b", "UseBoth"), (r"\b\w+ing\b \b\w+ed\b \b\w+s\b", "UseBoth"), (r"\w+\b \w+\b \w+\b \w+\b!", "UseBoth")]
LOOK_DFA_REFUSED = [(r"\b(DEBUG|INFO|WARN|ERROR)\b", "depends on cache history"),          # class of ' ' and '1': cache history decides
                    (r"(?m)^\w+: \w+ \w+ \w+$", "depends on cache history"),
                    (r"\b(GET|POST|PUT|DELETE|PATCH) /[a-z/]+", "depends on cache history"),
                    (r"\buser=\w+ host=\w+", "does not answer leftmost-first"),           # UseDFA: early return at the first possible end
                    (r"(?m)(\w+)(?:\b|x)$(?:=\w+)?\n", "match end behind the leftmost-first one"),   # UseBoth: the PikeVM restart would skip a match
                    (r"(?m)\Bbar(?:com|org)x(\w+)(\w+)\sbar", "would report an earlier match start"),
                    (r"\d+\B[a-z]+ [a-z]+", "does not answer leftmost-first"),
                    (r"(?m);\w+?[\w-]+\s+\s$\B$", "IsMatchAt can miss a match")]               # UseDFA, non-greedy: a wrong "no" would never reach the PikeVM              # UseDigitPrefilter: early return inside [a-z]+   # UseDFA: the reverse DFA ignores \B


@pytest.mark.parametrize("pat,strategy", LOOK_DFA_OK)
def test_look_programs_of_lazy_dfa_strategies(oracle, pat, strategy):
    """The oracle answers with the reference's look-aware lazy DFA (restated); ONE engine is reused over the haystacks, the
    proof claims independence of cache history too.  The twin runs the transducer the device runs."""
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.strategy == o.strategy == strategy and o.strategy_restated and rx.supported, (pat, rx.strategy, o.strategy, rx.why_unsupported)
    img = rx.fsm_image()
    assert img is not None
    rng = np.random.default_rng(3)
    alphabet = np.frombuffer(b"abingedsuser=ip status_09 .=:@;com\n  ", dtype=np.uint8)
    hays = [alphabet[rng.integers(0, len(alphabet), size=k)] for k in (0, 1, 9, 70, 300, 2000)]
    hays += [np.frombuffer(b"user=bob ip=10 status=ok  a=b;c=d  me@ex.am.com going moved bars  a b c d! xuser=a ip=b status=c_ " * 7, dtype=np.uint8),
             np.frombuffer(b" " * 140 + b"a=b;c=d user=a ip=b status=c one two three" + b"." * 120 + b"q r s t!", dtype=np.uint8),
             np.frombuffer(b"timeout=30 ms elapsed timeout=5ms elapsed GET /a/b HTTP GETX /a HTTP k=1; v=22 k=1;v=2 k=1_; v=2 x@y.z.org " * 9, dtype=np.uint8),
             np.frombuffer(b"10.0.0.1 10.0.0.1x 1.2.3.4.5 1.2.3 2024-01-02 2024-01-023 2024-01-02_ 15 ms 15ms 7 ms. 1234567.8.9.0\n" * 9, dtype=np.uint8),
             np.frombuffer(b"status=ok; code status=ok_; code status=a-b; code status=x;code status=yy; codes\n" * 9, dtype=np.uint8)]
    for hay in hays:
        exp = o.find_all_index(hay)
        for tile, chunk in ((3840, 32), (64, 8), (256, 16)):
            got = emu.find_all_fsm(img, hay, tile, chunk)
            if isinstance(got, int) and got in (-18, -32): got = emu.find_all_fsm(img, hay, tile, chunk, dense=1)
            if isinstance(got, int):                      # a budget of the toy geometries (match pending across too many tiny tiles), or
                assert tile != 3840 or got == -17, (pat, got)   # an entry state that 16 bytes of warm-up do not resolve: CXG_E_INPUT for this haystack
                continue
            assert got.shape == exp.shape and np.array_equal(got, exp), (pat, tile, chunk, got[:5].tolist(), exp[:5].tolist())


@pytest.mark.parametrize("pat,why", LOOK_DFA_REFUSED)
def test_look_programs_the_reference_answers_differently_are_refused(oracle, pat, why):
    rx = cx.compile(pat)
    assert rx.strategy == oracle.Regex(pat).strategy and rx.strategy in ("UseDFA", "UseBoth", "UseDigitPrefilter")
    assert not rx.supported and why in rx.why_unsupported, (pat, rx.why_unsupported)


def test_refused_look_program_really_differs_in_the_reference(oracle):
    """`\\buser=\\w+ host=\\w+` is UseDFA; the restated look-aware lazy DFA returns at the first byte at which its state holds a match
    state (checkWordBoundaryMatch, lazy.go:1262-1264, :1533-1560) — one byte into the greedy tail.  Leftmost-first takes it all."""
    import re
    pat, hay = r"\buser=\w+ host=\w+", b"user=ab host=cde f"
    assert oracle.Regex(pat).find_all_index(hay).tolist() == [[0, 14]]
    assert [list(m.span()) for m in re.finditer(pat.encode(), hay)] == [[0, 16]]
    assert not cx.compile(pat).supported


def test_look_dfa_fuzz_smoke():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import cpu_fuzz_lookdfa
    assert cpu_fuzz_lookdfa.main(120, 4242) == 0


# ---- FindAllSubmatchIndex of programs with assertions: spans by the look-aware transducer, slots by the backtracking pass whose
# LOOK states read the bytes around the position (device/bt.hpp).  The reference's FindAllSubmatch of these strategies is its
# PikeVM over the whole haystack (meta/findall.go:89-98), so `\b(DEBUG|INFO|WARN|ERROR)\b` has rows although its FindAllIndex —
# the lazy DFA's business there — is refused.
LOOK_CAPTURES = [r"\b(\w+)=(\w+)\b", r"\b(error|warn)\b", r"(?m)^(\d+) (\w+)", r"(\d+)\.(\d+)\b", r"\b(\w+)@(\w+)\.com\b", r"(?m)^(GET|POST) ",
                 r"\b(DEBUG|INFO|WARN|ERROR)\b", r"(?m)(\w+)$", r"(a|\b)(b|c)x", r"(\w+)\B(\d)", r"(?m)^(\w+): (\w+)$"]


@pytest.mark.parametrize("pat", LOOK_CAPTURES)
def test_captures_of_look_programs(oracle, pat):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.submatch_supported and rx.strategy == o.strategy, (pat, rx.strategy, o.strategy)
    if pat == r"\b(DEBUG|INFO|WARN|ERROR)\b": assert not rx.supported      # FindAllIndex: the lazy DFA's history-dependent answer; FindAllSubmatch: the PikeVM's
    simg, cap = rx.fsm_image(True), rx.submatch_blobs()[1]
    assert simg is not None and cap[:4] == b"TBXC"                   # cxgdev::kBtMagic: the backtracking image
    hays = [generate_test_input(), b"", b"a=b c=d; e=f\n12 ab\nGET /x\nPOST y\n3.4 5.6x me@x.com ERROR error warn_ warn\n7 z\nkey: val\nkey: val x\n",
            b"k=v", b"x=y\n" * 300, b"bx cx abx a1 ab12 x9\n" * 50, b"\n\n12 ab", b"1.2"]
    for hay in hays:
        h = np.frombuffer(hay, dtype=np.uint8)
        exp = o.find_all_submatch_index(h)
        spans = emu.find_all_fsm(simg, h, 3840, 32)
        if isinstance(spans, int) and spans in (-18, -32): spans = emu.find_all_fsm(simg, h, 3840, 32, dense=1)
        assert not isinstance(spans, int), (pat, spans)
        got = emu.captures_bt(cap, h, spans, 2 * rx.num_groups)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay), got[:4].tolist(), exp[:4].tolist())


# ---- round 3: maps instead of waits (scan_fsm.hip fsm_resolve_exits / fsm_group_entry and the scan over sub-chunk maps) ----------
MAPS_PATS = [r"(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)",
             r"\d+\.\d+x?", r"a+b|b+a", r"[a-c]x|[b-d]y", r"ab*c|a|bb", r"(?:ab)*[a-c]", r"\b\d+\.\d+\b", r"\berror\b", r"(?m)^\d+", r"\b[a-z]+\b"]


@pytest.mark.parametrize("pat", MAPS_PATS)
def test_map_compositions_give_the_true_entry_states(pat):
    """Input without synchronising structure: the kernel composes maps (member of a set of possible states -> end state) over
    sub-chunks, lanes, tiles and groups instead of waiting for the state in front.  The twin performs the same compositions and
    compares every entry state, tile exit and group entry they yield with a plain left-to-right walk — on few-symbol haystacks
    (sets stay unresolved for long stretches), with islands of ordinary text (constant maps in the chain), for the kernel's tile
    and for tiny tiles and groups (many borders)."""
    import random
    rx = cx.compile(pat)
    img = rx.fsm_image()
    if img is None:
        pytest.skip("no transducer image")
    rng = random.Random(len(pat))
    checked = unlisted = 0
    units = [b"1.", b"12.", b"a", b"ab", b"b", b"by", b"1", b"1.1x", b"error", b"ab ", b"01 ."]
    for it in range(40):
        unit = rng.choice(units)
        n = rng.choice([300, 4000, 9000, 40000, 130000])
        body = bytearray((unit * (n // len(unit) + 1))[:n])
        if it % 3 == 0:
            alpha = bytes(set(unit)) + b" "
            body = bytearray(rng.choices(alpha, k=n))
        for _ in range(it % 4):                                   # islands of ordinary text
            at = rng.randrange(0, max(1, n - 200))
            ln = rng.choice([5, 70, 700])
            body[at:at + ln] = (b"GET /index.html 10.0.0.1 error x=1 " * 30)[:ln]
        for tile, tpg in ((3840, 32), (3840, 2), (128, 3), (256, 8), (64, 1)):
            r = emu.fsm_maps_check(bytes(img), bytes(body), tile, tpg)
            if r == -17:
                unlisted += 1
                continue
            assert r >= 0, (pat, bytes(unit), n, tile, tpg, r)
            checked += r
    assert checked > 10000 or unlisted > 0, (checked, unlisted)


def test_direct_tables_exist_for_the_benchmark_patterns():
    """Round 6: shallow machines without look-around carry the byte-indexed section (fsm.hpp "Direct mode") when it fits; the README
    IPv4 pattern's 95 explored stacks minimise to 20 states (host/fsm.cc), which is what makes its section fit."""
    import struct
    for pat, want in ((GENERAL[-1], True), (r"\d+\.\d+x?", True), (r"a+b|b+a", True), (r"\b\d+\b", False)):
        img = cx.compile(pat).fsm_image()
        n_t, depth, nk = struct.unpack_from("<I", img, 4)[0], struct.unpack_from("<I", img, 28)[0], struct.unpack_from("<I", img, 92)[0]
        direct_off, direct_bytes, d_slots = struct.unpack_from("<III", img, 120)
        assert (direct_off != 0) == want, (pat, direct_off)
        if want:
            assert depth <= 1 and nk == 1 and n_t <= 64 and direct_bytes == d_slots * 256 + 256 and direct_bytes <= 12288, (pat, n_t, direct_bytes)
    readme = r"(?:(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.){3}(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)"
    assert struct.unpack_from("<I", cx.compile(readme).fsm_image(), 4)[0] == 20
