"""CPU tier (`-m "not gpu"`): the C-ABI library loads and exports what the header declares, the host
front-end agrees with the oracle (strategy, NFA size), and the device lane walks — compiled for the
host by tests/emu — reproduce the oracle's spans for every chunk geometry.  No GPU compute here."""
import os
import re
import struct

import numpy as np
import pytest

import coregex_amd as cx
from coregex_amd import _lib
from refcorpus import COMPAT_PATTERNS, generate_test_input

import emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, "include", "coregex_hip.h")).read()
    declared = set(re.findall(r"\b(cxg_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"cxg_nfa"}
    L = _lib.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(declared) == sorted(_lib.SYMBOLS)
    assert L.cxg_version().startswith(b"coregex_hip")


def test_header_is_plain_c_and_example_compiles(tmp_path):
    """include/coregex_hip.h is C99 (no C++/torch types): the example host program compiles against it."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    obj = tmp_path / "find_all.o"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                           "-c", os.path.join(root, "examples", "find_all.c"), "-o", str(obj)])
    assert obj.stat().st_size > 0


def test_no_gpu_means_loud_failure():
    if cx.device_count() > 0:
        pytest.skip("a GPU is present")
    rx = cx.compile(r"\d+\.\d+\.\d+\.\d+")
    with pytest.raises(cx.CoregexError) as ei:
        rx.find_all_index(b"1.2.3.4")
    assert ei.value.code == _lib.CXG_E_NO_GPU


PATTERNS = [
    r"\d+\.\d+\.\d+\.\d+", r"[\w]+", r"error", r"error|warning|fatal|critical", r"(\w+)@(\w+)\.(\w+)",
    r"\d+\.\d+\.\d+", r"\d+:\d+:\d+", r"[a-z]+", r"[0-9a-f]+", r"ab", r"a", r"ab|abc", r"((a+)(b+))", r"a+?",
    r"[a-zA-Z]+[0-9]+", r"\w+[0-9]+", r"(\w)+", r"25[0-5]|2[0-4][0-9]|1[0-9][0-9]|[1-9][0-9]|[0-9]",
    r"(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]", r"[1-9][0-9]*|0", r"foo|bar|baz", r"[a-f0-9]{32,}",
    r"apple|banana|cherry|date|elderberry|fig|grape|honeydew|kiwi|lemon|mango|orange", r"x[ab]+?y", r"(a|ab)(c|bcd)",
    r"\d{1,3}\.\d{1,3}", r"[0-5]+x", r"(foo|foobar)\d+", r"\d+[a-z]", r"\d+\.\d", r"\d+\.\d+x?", r"\w+\.txt", r"[a-z]+@[a-z]+", r"ERROR|WARN", r"a{2,4}b", r"(?:ab)*c", r"a||b", r"[a-c]|x|yz",
]


# word boundaries: UseNFA in the reference (PikeVM), the transducer kernel here (fsm.hpp "Look-around")
LOOK_PATTERNS = [r"\berror\b", r"\b\d+\b", r"\bfoo\w+", r"\b(GET|POST)\b", r"\w+\b", r"\Btest", r"ab(a|\b)", r"(ab)+(a|\b)\b\b", r"\b[a-z]+\b",
                 r"(?:\bx)+", r"a\B", r"\b\b", r"x\b|\By", r"(?m)^line", r"(?m)error$", r"(?m)^(GET|POST|PUT|DELETE|PATCH)", r"(?m)^\w+$", r"(?m)a$\n^b"]


# `.` and classes that reach past U+007F: UTF-8 byte automata (nfa/compile.go:440-1222)
WIDE_PATTERNS = [r".", r".*", r".+", r"a.c", r'"[^"]*"', r"\S+", r"GET .* HTTP", r"error: .*", r"[^,]+,", r"(?s)a.b", r"\D+x", r"é+", r"[a-zé]+x", r"[α-ω]+",
                 r"\d+ .* \d+", r"x.y", r"[föd]+", r"[äöü]+", r"(ö|a)+", r"[^\n]+", r"[\x{80}-\x{10FFFF}]x", r"[\x{100}-\x{7FF}]+", r"[\x{800}-\x{FFFF}]y", r"[\x{D000}-\x{E100}]",
                 r"[\x{D800}-\x{DFFF}]", r"[\x{10000}-\x{3FFFF}]", r"[\x{7F0}-\x{20000}]z", r"\W\d", r"k[^k]", r"(.)(.)", r".?x", r"[^\x00-\x7F]+", r"user=(\S+)", r'"([^"]*)"', r"\[([^\]]+)\]"]


@pytest.mark.parametrize("pat", PATTERNS + LOOK_PATTERNS + WIDE_PATTERNS)
def test_frontend_agrees_with_oracle(oracle, pat):
    o = oracle.Regex(pat)
    p = cx.compile(pat)
    assert p.nfa_states == o.nfa_states, pat
    assert p.num_groups == o.num_groups
    if o.strategy_restated:
        assert p.strategy == o.strategy, pat


def test_nfa_view_matches_oracle_dump(oracle):
    """State-by-state comparison of the product NFA with the oracle's (creation order is semantic)."""
    kinds = {0: "Match", 1: "ByteRange", 2: "Sparse", 3: "Split", 4: "Eps", 5: "Cap", 7: "Look"}
    for pat in PATTERNS + LOOK_PATTERNS + WIDE_PATTERNS:
        p = cx.compile(pat)
        v = p.nfa()
        lines = oracle.Regex(pat).dump().strip().split("\n")[2:]
        assert len(lines) == v.n_states, pat
        for i, line in enumerate(lines):
            st = v.states[i]
            body = line.split(": ", 1)[1]
            k = kinds[st.kind]
            if k == "Split":
                assert body.endswith(f"Split({st.left},{st.right})"), (pat, i, body)
            elif k == "ByteRange":
                assert body == f"ByteRange[{st.lo}-{st.hi}]->{st.next}", (pat, i, body)
            elif k == "Sparse":
                exp = "Sparse" + "".join(
                    f" [{v.trans[st.trans_off + j].lo}-{v.trans[st.trans_off + j].hi}]->{v.trans[st.trans_off + j].next}"
                    for j in range(st.trans_len))
                assert body == exp, (pat, i, body)
            elif k == "Match":
                assert body == "Match"
            elif k == "Cap":
                assert body == f"Cap{st.cap_index}{'(' if st.cap_start else ')'}->{st.next}", (pat, i, body)
            elif k == "Look":
                assert body == f"Look{st.lo}->{st.next}", (pat, i, body)      # cxg_nfa_state.lo = nfa.Look
            else:
                nxt = -1 if st.next == 0xFFFFFFFF else st.next
                assert body == f"Eps->{nxt}", (pat, i, body)


EMU_PATTERNS = [p for p in PATTERNS]


def _fast_digit(p) -> bool:
    import struct
    return bool(struct.unpack_from("<I", p.blob(), 8)[0] & 2)


def _table_free(p):
    """Programs of kind 5 run on the transducer kernel alone (word boundaries, (?m)^ literals): no table-walking image.  Round 4:
    literals between assertions are a literal image (kind 4) with TeddyAux::looks set — the wave kernel and the transducer only
    (tests/test_wrapped_cpu.py is their twin tier)."""
    b = p.blob()
    kind = struct.unpack_from("<I", b, 4)[0]
    if kind == 4:
        aux_off = struct.unpack_from("<I", b, 4 * 14)[0]
        return struct.unpack_from("<I", b, aux_off + 4 * 11)[0] != 0     # walk.hpp TeddyAux::looks
    return kind == 5


@pytest.mark.parametrize("chunk", [4, 16, 64])
def test_emulated_lane_walks_match_oracle_on_reference_corpus(oracle, chunk):
    corpus = generate_test_input()
    n_checked = 0
    for name, pat in COMPAT_PATTERNS.items():
        try:
            p = cx.compile(pat)
        except cx.CoregexError:
            continue
        if not p.supported or p.strategy not in ("UseDigitPrefilter", "UseDFA", "UseTeddy") or _table_free(p):
            continue
        got = emu.find_all(p.blob(), corpus, chunk)
        exp = oracle.Regex(pat).find_all_index(corpus)
        assert got.tolist() == exp.tolist(), (name, chunk)
        if p.strategy == "UseDigitPrefilter":
            assert emu.find_all(p.blob(), corpus, chunk, flat=1).tolist() == exp.tolist(), (name, chunk, "flat")
            if _fast_digit(p):
                assert emu.find_all(p.blob(), corpus, chunk, flat=2).tolist() == exp.tolist(), (name, chunk, "list")
        n_checked += 1
    assert n_checked >= 5


def test_emulated_lane_walks_random(oracle):
    rng = np.random.default_rng(7)
    alphabet = np.frombuffer(b"0123456789. ab\ncdxy@_eror:", dtype=np.uint8)
    tried = 0
    for pat in EMU_PATTERNS:
        p = cx.compile(pat)
        if not p.supported or p.strategy not in ("UseDigitPrefilter", "UseDFA", "UseTeddy") or _table_free(p):
            continue
        o = oracle.Regex(pat)
        blob = p.blob()
        tried += 1
        for it in range(120):
            n = int(rng.integers(0, 700))
            hay = alphabet[rng.integers(0, len(alphabet), size=n)].tobytes()
            exp = o.find_all_index(hay).tolist()
            for chunk in (4, 8, 64):
                got = emu.find_all(blob, hay, chunk).tolist()
                assert got == exp, (pat, chunk, hay)
                if p.strategy == "UseDigitPrefilter":
                    assert emu.find_all(blob, hay, chunk, flat=1).tolist() == exp, (pat, chunk, hay, "flat")
                    if _fast_digit(p):
                        assert emu.find_all(blob, hay, chunk, flat=2).tolist() == exp, (pat, chunk, hay, "list")
    assert tried >= 10


def test_chain_prefilter_emulated(oracle):
    """Bit-parallel chain prefilter + DFA verify (4th-generation digit kernel) vs the oracle."""
    import struct
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"0123456789..:: ab\nxy-", dtype=np.uint8)
    corpus = generate_test_input()
    synth = cx.synth_pages(2, 0xC0FFEE02, 3, 40).tobytes()
    n_ok = 0
    for pat in [r"\d+\.\d+\.\d+\.\d+", r"\d+\.\d+\.\d+", r"\d+:\d+:\d+", r"\d+[a-f]+;\d+", r"\d+\.\d+"]:
        p = cx.compile(pat)
        if not p.supported or p.strategy != "UseDigitPrefilter":
            continue
        if (struct.unpack_from("<I", p.blob(), 8)[0] & 6) != 6:
            continue
        o = oracle.Regex(pat)
        n_ok += 1
        for hay in (corpus, synth, b"", b"1.2.3.4", b"9" * 300 + b" 1.2.3.4 " + b"7." * 200):
            got = emu.find_all_chain(p.blob(), hay)
            assert got is not None and got.tolist() == o.find_all_index(hay).tolist(), (pat, len(hay))
            got = emu.find_all_chain(p.blob(), hay, 3840, 256)      # wave-tile geometry (scan_digit_wave.hip)
            assert got is not None and got.tolist() == o.find_all_index(hay).tolist(), (pat, len(hay), "wave")
        for _ in range(150):
            n = int(rng.integers(0, 40000))
            hay = alphabet[rng.integers(0, len(alphabet), size=n)].tobytes()
            for geom in ((16384, 256), (3840, 256)):
                got = emu.find_all_chain(p.blob(), hay, *geom)
                if got is None:
                    continue      # no sync byte in a halo: the kernel would hand the scan to the flat kernel
                assert got.tolist() == o.find_all_index(hay).tolist(), (pat, n, geom)
    assert n_ok >= 3
    # a halo without any synchronising byte must raise the fallback flag, never a wrong answer
    assert emu.find_all_chain(cx.compile(r"\d+\.\d+\.\d+\.\d+").blob(), b"1.2.3.4." * 4000) is None


def _restart_check(blob) -> int:
    """ChainAux.restart_check of a chain program (walk.hpp): the kernel hands the scan over when a match ends inside a
    run of the chain's first class."""
    import struct
    aux_off = struct.unpack_from("<I", blob, 56)[0]
    return blob[aux_off + 256 + 208]                                # after the class tables and the three packed step words


def _fallback_ok(got: int, blob) -> bool:
    """Twin returned -(16 + reason): no synchronising byte in a halo (1) is always legitimate, a match ending inside a
    first-class run (64) only for programs that carry the restart check."""
    reason = -got - 16
    return reason > 0 and (reason & ~(1 | (64 if _restart_check(blob) else 0))) == 0


def test_bitparallel_chain_emulated(oracle):
    """Sixth generation (scan_chain_wave.hip): starts, ownership and ends all from class bitmaps — digit-prefilter
    chains and UseDFA chains (literals, run/byte sequences) vs the oracle, several window geometries."""
    import struct
    rng = np.random.default_rng(23)
    corpus = generate_test_input()
    n_ok = 0
    cases = [(r"\d+\.\d+\.\d+\.\d+", b"0123456789..:: ab\nxy-", 2), (r"\d+:\d+:\d+", b"0123456789..:: ab\nxy-", 2),
             (r"error", b"eror rre\nxE", 1), (r"[a-z]+=\d+", b"abz=09 =\n-", 2), (r"ab+c", b"abc abbc\n", 1), (r"aba", b"ab \n", 1),
             (r"GET", b"GET \nEG", 1), (r"\d{4}-\d{2}-\d{2}", b"0123456789-- \n", 2), (r"\d\d:\d\d", b"0189:: \n", 2)]
    for pat, alpha, cfg in cases:
        p = cx.compile(pat)
        flags = struct.unpack_from("<I", p.blob(), 8)[0] if p.supported else 0
        if not (flags & 16):
            continue
        n_ok += 1
        o = oracle.Regex(pat)
        alphabet = np.frombuffer(alpha, dtype=np.uint8)
        synth = cx.synth_pages(cfg, 0xC0FFEE00 + cfg, 7, 24).tobytes()
        for hay in (corpus, synth, b"", b"1.2.3.4", b"error", b"xerror", b"a=1", b"ababa abbbc"):
            for geom in ((3840, 256), (192, 64), (64, 64)):
                got = emu.find_all_chain6(p.blob(), hay, *geom)
                if isinstance(got, int):
                    assert _fallback_ok(got, p.blob()), (pat, geom, got)
                    continue
                assert got.tolist() == o.find_all_index(hay).tolist(), (pat, len(hay), geom)
        for _ in range(120):
            n = int(rng.integers(0, 20000))
            hay = alphabet[rng.integers(0, len(alphabet), size=n)].tobytes()
            exp = o.find_all_index(hay).tolist()
            for geom in ((3840, 256), (192, 64), (128, 128)):
                got = emu.find_all_chain6(p.blob(), hay, *geom)
                if isinstance(got, int):
                    assert _fallback_ok(got, p.blob()), (pat, geom, got)
                    continue
                assert got.tolist() == exp, (pat, n, geom)
        # exact window edge at the end of input: the last run touches the last byte of a full window
        for total in (4096, 4096 + 3840, 256, 128):
            tail = {"error": b"error", "GET": b"GET"}.get(pat, None)
            if tail is None:
                tail = b"1.2.3.4" if "." in pat else (b"1:2:3" if ":" in pat else (b"ab=12" if "=" in pat else (b"abbc" if "+" in pat else b"aba")))
            hay = b" " * (total - len(tail)) + tail
            got = emu.find_all_chain6(p.blob(), hay, 3840, 256)
            assert not isinstance(got, int) and got.tolist() == o.find_all_index(hay).tolist(), (pat, total)
    assert n_ok >= 5, n_ok
    # FindAllSubmatch programs: the span image carries the chain too, classes may be unions of ranges (\\w)
    for pat, cfg in ((r"(\w+)@(\w+)\.(\w+)", 5), (r"([a-z]+)=(\d+)", 5)):
        p = cx.compile(pat)
        span_blob = p.submatch_blobs()[0]
        assert (struct.unpack_from("<I", span_blob, 8)[0] & 16) == 16, pat        # ordered chain (\\w: a set class)
        o = oracle.Regex(pat)
        synth = cx.synth_pages(cfg, 0xC0FFEE00 + cfg, 3, 48).tobytes()
        for hay in (synth, corpus, b"a@b.c", b"foo@bar.com x@y.z k=1"):
            for geom in ((3840, 256), (192, 64)):
                got = emu.find_all_chain6(span_blob, hay, *geom)
                if isinstance(got, int):
                    assert _fallback_ok(got, span_blob), (pat, geom, got)
                    continue
                assert got.tolist() == o.find_all_index(hay).tolist(), (pat, len(hay), geom)
    # unordered chains (a run whose class meets the class of the step before) must not get the flag
    for pat in (r"a[ab]+", r"\d\d+x"):
        p = cx.compile(pat)
        if p.supported:
            assert not (struct.unpack_from("<I", p.blob(), 8)[0] & 16), pat
    # a chain that begins with a run and can end on a byte of that run's class with more of the class behind it: the
    # next match may start mid-run ("z.azz.bc" -> [0,4] [4,8]), which run-start candidates cannot express
    # such chains carry the restart check: the kernel (and its twin) hands the scan over when a match ends inside a
    # first-class run, and is exact otherwise
    for pat, hay, exp in ((r"z+\.\w\w", b"z.azz.bc", [[0, 4], [4, 8]]), (r"[a-z0-9]+\.+[x-z]", b"ab.xy.z", [[0, 4], [4, 7]])):
        p = cx.compile(pat)
        assert p.supported and (struct.unpack_from("<I", p.blob(), 8)[0] & 16) and _restart_check(p.blob()) == 1, pat
        assert oracle.Regex(pat).find_all_index(hay).tolist() == exp
        assert emu.find_all(p.blob(), hay).tolist() == exp
        assert emu.find_all_chain6(p.blob(), hay) == -(16 + 64)
        calm = hay.replace(b"zz", b"z ").replace(b"xy", b"x ")
        assert emu.find_all_chain6(p.blob(), calm).tolist() == oracle.Regex(pat).find_all_index(calm).tolist() != []
    span_blob = cx.compile(r"(\w+)=(\d+)").submatch_blobs()[0]
    assert (struct.unpack_from("<I", span_blob, 8)[0] & 16) and _restart_check(span_blob) == 1
    assert _restart_check(cx.compile(r"\d+\.\d+\.\d+\.\d+").blob()) == 0 and _restart_check(cx.compile(r"(\w+)@(\w+)\.(\w+)").submatch_blobs()[0]) == 0


def _random_chain_patterns(rng, count):
    """Random concatenations of literal bytes, classes and class+ over a small alphabet (some are chains, some not)."""
    atoms = ["a", "b", "c", "x", r"\.", ":", r"\d", "[a-c]", "[x-z]", r"\d+", "[a-c]+", "[x-z]+", "a+", r"\.+", "[0-4]+", "[5-9]"]
    out = []
    while len(out) < count:
        n = int(rng.integers(1, 6))
        pat = "".join(atoms[int(rng.integers(0, len(atoms)))] for _ in range(n))
        if pat not in out:
            out.append(pat)
    return out


def test_random_chain_patterns_emulated(oracle):
    """Fuzz of the chain extraction (complete / ordered flags) and the bit-parallel evaluation: whatever pattern gets
    the ordered-chain flag must reproduce the oracle on random haystacks, window geometry 192+64 and 3840+256."""
    import struct
    rng = np.random.default_rng(77)
    alphabet = np.frombuffer(b"abcxyz.:0123456789 \n", dtype=np.uint8)
    hays = [alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(0, 6000)))].tobytes() for _ in range(12)]
    # few-symbol haystacks: dense, abutting matches (a match that starts where the previous one ended, inside a run)
    hays += [alphabet[rng.choice(len(alphabet), size=4000, p=rng.dirichlet(0.25 * np.ones(len(alphabet))))].tobytes() for _ in range(10)]
    n_chain = 0
    for pat in _random_chain_patterns(rng, 220):
        try:
            p = cx.compile(pat)
        except cx.CoregexError:
            continue
        if not p.supported:
            continue
        if not (struct.unpack_from("<I", p.blob(), 8)[0] & 16):
            continue
        n_chain += 1
        o = oracle.Regex(pat)
        assert o.strategy == p.strategy, pat
        for hay in hays:
            exp = o.find_all_index(hay).tolist()
            for geom in ((192, 64), (3840, 256)):
                got = emu.find_all_chain6(p.blob(), hay, *geom)
                if isinstance(got, int):
                    assert _fallback_ok(got, p.blob()), (pat, geom, got)
                    continue
                assert got.tolist() == exp, (pat, len(hay), geom)
    assert n_chain >= 40, n_chain


def test_teddy_and_charclass_wave_twins(oracle):
    """Sequential twins of scan_teddy_wave.hip (three-byte fingerprint + exact verification + (zA, zB] ownership) and
    scan_charclass_wave.hip (start/end bitmaps, skipped leading end) vs the oracle, several window geometries."""
    rng = np.random.default_rng(31)
    corpus = generate_test_input()
    lits = "error|warning|fatal|critical|panic|timeout|refused|denied|googlebot|bingbot|yandexbot|crawler|spider|failure|exception|overflow"
    fat = "|".join(FAT_WORDS)      # 48 literals: the reference's Fat Teddy (teddy_fat.go), buckets folded 16 -> 8 on the device
    for pat, alpha, cfg in ((lits, b"erowanigftlcpmusdbyxv  \n.", 3), ("spider|error|crawler|denied", b"spidercawln  \n", 3),
                            (fat, b"wordkeyvalitm0123456789  \n", 3)):
        p = cx.compile(pat)
        assert p.strategy == "UseTeddy" and p.supported
        o = oracle.Regex(pat)
        alphabet = np.frombuffer(alpha, dtype=np.uint8)
        synth = cx.synth_pages(cfg, 0xC0FFEE00 + cfg, 5, 32).tobytes()
        hays = [corpus, synth, b"", b"error", b"xerror", b"spiderror crawlerror", b"err", b"erro"]
        hays += [alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(0, 20000)))].tobytes() for _ in range(60)]
        for hay in hays:
            exp = o.find_all_index(hay).tolist()
            for geom in ((3840, 256), (192, 64), (128, 128)):
                got = emu.find_all_teddy_wave(p.blob(), hay, *geom)
                if isinstance(got, int):
                    assert got in (-17, -24, -25), (pat, geom, got)
                    continue
                assert got.tolist() == exp, (pat, len(hay), geom)
    for pat, alpha in ((r"[\w]+", b"ab_9 Z-\n"), (r"[a-c]+", b"abcd \n"), (r"[0-9a-fA-F]+", b"09afAFgz \n")):
        p = cx.compile(pat)
        assert p.strategy == "UseCharClassSearcher" and p.supported
        o = oracle.Regex(pat)
        alphabet = np.frombuffer(alpha, dtype=np.uint8)
        synth = cx.synth_pages(4, 0xC0FFEE04, 9, 16).tobytes()
        hays = [corpus, synth, b"", b"a", b" a", b"a ", b"w" * 4096, b"w" * 3840 + b" ", b" " + b"w" * 200 + b" "]
        hays += [alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(0, 12000)))].tobytes() for _ in range(60)]
        for hay in hays:
            exp = o.find_all_index(hay).tolist()
            for geom in ((3840, 256), (192, 64), (64, 64)):
                got = emu.find_all_charclass_wave(p.blob(), hay, *geom)
                if isinstance(got, int):
                    assert got in (-17, -24), (pat, geom, got)      # a run longer than the halo / too many runs
                    continue
                assert got.tolist() == exp, (pat, len(hay), geom)


def test_history_dependent_reference_programs_are_refused(oracle):
    """The reference's lazy DFA files states under their sorted NFA set but walks them in first-built order
    (dfa/lazy/state.go:329-373): for `a?(a|b)` its answer depends on what the Regex scanned before.  The host refuses
    exactly those programs (program.cc priorityOrderConflict); for everything it accepts, the oracle's answer must
    not depend on cache history — a Regex reused across haystacks in either order equals a fresh one."""
    fresh, used = oracle.Regex("a?(a|b)"), oracle.Regex("a?(a|b)")
    used.find_all_index(b"b")
    assert fresh.find_all_index(b"aaaa").tolist() == [[0, 2], [2, 4]]
    assert used.find_all_index(b"aaaa").tolist() == [[0, 1], [1, 2], [2, 3], [3, 4]]
    for pat in ("a?(a|b)", r"\wa?[ab]", "a?[ab]", "x*[x-z]"):
        rx = cx.compile(pat)
        assert not rx.supported and "cache history" in rx.why_unsupported, pat
    for pat in ("a?c", r"-?\d+", "x*a", "a?z+", r"https?://\w+", r"\d+\.\d+\.\d+\.\d+"):
        assert cx.compile(pat).supported, pat
    rng = np.random.default_rng(5)
    atoms = ["a", "b", "c", "x", r"\.", r"\d", "[a-c]", "[x-z]", r"\d+", "[a-c]+", "a+", "ab|xy", r"\w", r"\w+", "[ab]", "(a|b)", "(ab)+",
             "a?", "b?", r"\d{1,3}", "x*", "(a|ab)", "(ab|a)", "[ab]*"]
    alphabet = np.frombuffer(b"abcxyz.:-0123456789 \n", dtype=np.uint8)
    skew = np.ones(len(alphabet)); skew[:6] = 8; skew /= skew.sum()
    hays = [alphabet[rng.choice(len(alphabet), size=n, p=skew)] for n in (60, 500, 6000)] + [np.frombuffer(b"a" * 300 + b"b" + b"a" * 300, dtype=np.uint8)]
    seen, n_checked, n_refused = set(), 0, 0
    while len(seen) < 250:
        pat = "".join(atoms[int(rng.integers(0, len(atoms)))] for _ in range(int(rng.integers(1, 5))))
        if pat in seen:
            continue
        seen.add(pat)
        try:
            rx = cx.compile(pat)
        except cx.CoregexError:
            continue
        if rx.strategy not in ("UseDFA", "UseDigitPrefilter"):
            continue
        if not rx.supported:
            n_refused += "cache history" in rx.why_unsupported
            continue
        shared = oracle.Regex(pat)
        n_checked += 1
        for order in (range(len(hays)), reversed(range(len(hays)))):
            for hi in order:
                assert np.array_equal(shared.find_all_index(hays[hi]), oracle.Regex(pat).find_all_index(hays[hi])), (pat, hi)
    assert n_checked >= 60 and n_refused >= 1, (n_checked, n_refused)


def test_wide_fuzz_host_and_twins(oracle):
    """Wide fuzz (optional/star prefixes, non-greedy, case folding, dot, counted repeats, captures): strategy agreement
    with the oracle, and for everything the device path accepts the lane walks and the wave-kernel twins reproduce the
    oracle's spans and captures.  scripts/cpu_fuzz.py is the long-running version."""
    import struct
    atoms = ["a", "b", "c", "x", "y", r"\.", ":", "-", r"\d", "[a-c]", "[x-z]", r"\d+", "[a-c]+", "[x-z]+", "a+", r"\.+", "[0-4]+", "[5-9]",
             "ab|xy", "abc|xyz|a:c", r"\w", r"\w+", "[a-z0-9]+", "[ab]", "(a|b)", "(ab)+", "a?", r"\d{2}", r"\d{1,3}", "x*", "(xy|ab|ca)",
             "abcx|bcxy|cxyz|xyza", "z+", ".", ".*", r"\s", r"\S+", "(?:ab)*", "a*", r"\d{2,}", "(a+)(b+)", "[^x]", "(?i:ab)", "(?i:x)", "b+?",
             "(?:a|b|c)+", "abcabc", "abc", "xyz", "a:c", "xyzxyz|abcabc|a:c:a:", r"[\d.]+", "(x|y|z)", " ", r"\n"]
    alphabet = np.frombuffer(b"abcxyz.:-0123456789 \nABX", dtype=np.uint8)
    n_dev, n_sub, n_chain, n_both, n_both_long, n_bt_limit = 0, 0, 0, 0, 0, 0
    for seed in (300, 301, 302, 303):
        rng = np.random.default_rng(seed)
        hays = [alphabet[rng.integers(0, len(alphabet), size=int(n))].tobytes() for n in (0, 3, 200, 5000)]
        hays += [alphabet[rng.choice(len(alphabet), size=5000, p=rng.dirichlet(0.25 * np.ones(len(alphabet))))].tobytes() for _ in range(4)]
        hays += [b"abcxyza:c" * 300, b"a" * 900 + b"b" + b"a" * 900]
        seen = set()
        while len(seen) < 150:
            pat = "".join(atoms[int(rng.integers(0, len(atoms)))] for _ in range(int(rng.integers(1, 5))))
            if pat in seen:
                continue
            seen.add(pat)
            try:
                rx = cx.compile(pat)
            except cx.CoregexError:
                with pytest.raises(oracle.OracleError):
                    oracle.Regex(pat)
                continue
            try:
                o = oracle.Regex(pat)
            except oracle.OracleError:
                assert not rx.supported, pat
                continue
            assert rx.strategy == o.strategy, pat
            if rx.supported:
                n_dev += 1
                blob = rx.blob()
                flags = struct.unpack_from("<I", blob, 8)[0]
                n_chain += bool(flags & 16)
                for hay in hays:
                    exp = o.find_all_index(hay).tolist()
                    if rx.strategy == "UseBoth":
                        # the reference restarts its PikeVM 100 bytes before the DFA's end: plain leftmost-first unless a
                        # match is longer than that — then the kernels raise error bit 64 (CXG_E_INPUT), checked on the GPU
                        plain = o.find_all_submatch_index(hay)[:, :2]
                        assert (flags & 128), pat
                        if len(plain) and int((plain[:, 1] - plain[:, 0]).max()) > 100:
                            n_both_long += 1
                            assert emu.find_all(blob, hay).tolist() == plain.tolist(), (pat, len(hay))
                            continue
                        n_both += 1
                    if struct.unpack_from("<I", blob, 4)[0] == 5:    # kKindFsmOnly (a UseNFA program: more than 100 NFA states): the transducer alone
                        if rx.nullable == 2:                             # `a*?`: every match is empty, no device program
                            assert emu.merge_empty_matches(np.zeros((0, 2), dtype=np.int64), len(hay)).tolist() == exp, pat
                            continue
                        got = emu.find_all_fsm(rx.fsm_image(), np.frombuffer(hay, dtype=np.uint8), 3840, 32)
                        if isinstance(got, int) and got in (-18, -32):
                            got = emu.find_all_fsm(rx.fsm_image(), np.frombuffer(hay, dtype=np.uint8), 3840, 32, dense=1)
                        if rx.nullable and not isinstance(got, int):     # round 4: the image is the non-empty variant's (program.cc nonEmptyVariant)
                            got = emu.merge_empty_matches(got, len(hay))
                        assert isinstance(got, int) or got.tolist() == exp, (pat, rx.strategy, len(hay))
                        continue
                    is_cc = struct.unpack_from("<I", blob, 4)[0] == 3    # kKindCharClass: UseCharClassSearcher, or (round 4) a `C+` program of a DFA strategy
                    if not is_cc:
                        assert emu.find_all(blob, hay).tolist() == exp, (pat, rx.strategy, len(hay))
                    twins = []
                    if (flags & 16) and rx.strategy in ("UseDFA", "UseDigitPrefilter", "UseBoth"):
                        if flags & 512:                              # bounded repetition: surrogate chain + field bounds
                            raw = rx.chain_bounds()[0]
                            twins = [emu.find_all_chain6_bounded(blob, raw, hay, 192, 64), emu.find_all_chain6_bounded(blob, raw, hay, 3840, 256)]
                        else:
                            twins = [emu.find_all_chain6(blob, hay, 192, 64), emu.find_all_chain6(blob, hay, 3840, 256)]
                    elif struct.unpack_from("<I", blob, 4)[0] == 4 or (flags & 256):   # literal image, or required literal prefix + anchored DFA
                        twins = [emu.find_all_teddy_wave(blob, hay)]
                    elif is_cc and (flags & 64):
                        twins = [emu.find_all_charclass_wave(blob, hay)]
                    for got in twins:
                        if got is not None and not isinstance(got, int):
                            assert got.tolist() == exp, (pat, rx.strategy, len(hay))
            if "(" in pat and rx.submatch_supported:
                n_sub += 1
                sb, cb = rx.submatch_blobs()[:2]
                for hay in hays[:6]:
                    exp = o.find_all_submatch_index(hay)
                    try:
                        got = emu.find_all_submatch(sb, cb, hay, exp.shape[1])
                    except AssertionError as e:
                        # a pattern that is not one-pass (`\S`, `.`: a byte >= 0x80 is accepted by several states) takes the backtracking
                        # pass, whose stack holds one entry per repetition: a match of many hundred bytes fails the call loudly
                        # (kErrSerialLimit) — never a wrong row
                        assert "error -4" in str(e) and len(exp) and int((exp[:, 1] - exp[:, 0]).max()) > 300, (pat, len(hay), str(e))
                        n_bt_limit += 1
                        continue
                    assert got.shape == exp.shape and np.array_equal(got, exp), (pat, "submatch", len(hay))
    assert n_bt_limit <= 12, n_bt_limit
    assert n_dev >= 200 and n_sub >= 30 and n_chain >= 20 and n_both >= 20 and n_both_long >= 1, (n_dev, n_sub, n_chain, n_both, n_both_long)


def _chain_of(span_blob):
    """ChainAux (walk.hpp) of a span program: [(is_run, membership[256])]."""
    import struct
    aux_off = struct.unpack_from("<I", span_blob, 56)[0]
    raw = span_blob[aux_off + 256:aux_off + 256 + 184]
    nops, ncls = struct.unpack_from("<II", raw, 0)
    op_kind, op_cls = raw[8:72], raw[72:136]                       # kChainMaxOps = 64
    cls_kind, cls_lo, cls_hi, cls_nr = raw[136:140], raw[140:144], raw[144:148], raw[148:152]
    rlo, rhi = raw[152:168], raw[168:184]
    members = []
    for c in range(ncls):
        m = np.zeros(256, dtype=bool)
        if cls_kind[c] == 0:
            m[0x30:0x3A] = True
        elif cls_kind[c] == 3:
            for r in range(cls_nr[c]):
                m[rlo[4 * c + r]:rhi[4 * c + r] + 1] = True
        else:
            m[cls_lo[c]:cls_hi[c] + 1] = True
        members.append(m)
    return [(op_kind[k] == 1, members[op_cls[k]]) for k in range(nops)]      # kChainRun == 1


def test_chain_captures_mapping(oracle):
    """Captures straight from the chain kernel (program.cc deriveChainCaps): every slot of every oracle row equals
    match start / match end / the end of one of two runs, plus the derived constant — the boundaries recomputed here
    by walking the chain steps greedily from the match start."""
    corpus = generate_test_input()
    n_on = 0
    cases = [(r"(\w+)@(\w+)\.(\w+)", 5), (r"([a-z]+)=(\d+)", 5), (r"((a+)(b+))", 2), (r"(ab)c(d)", 1), (r"(a)(b)c", 1), (r"x(\d+)y(\d+)z", 2),
             (r"([a-z])+@", 5), (r"(\d+)-(\d+)", 2), (r"(\d+)\.(\d+)\.(\d+)\.(\d+)", 2), (r"(\w+)=(\d+)", 5), (r"(GET|POST) (\w+)", 2)]
    for pat, cfg in cases:
        rx = cx.compile(pat)
        assert rx.submatch_supported, pat
        caps = rx.chain_captures()
        if pat == r"(GET|POST) (\w+)":
            assert caps is None, pat           # not a chain: two-kernel path
            continue
        assert caps is not None, pat
        n_on += 1
        chain = _chain_of(rx.submatch_blobs()[0])
        o = oracle.Regex(pat)
        synth = cx.synth_pages(cfg, 0xC0FFEE00 + cfg, 3, 24).tobytes()
        extra = b"aab abb ab xaab abcd abc x1y22z x12y3z k@l a@ ab@ 10-20 1-2 foo=12 a@b.c aa@bb.cc "
        for hay in (corpus, synth, extra * 3):
            h = np.frombuffer(hay, dtype=np.uint8)
            rows = o.find_all_submatch_index(hay)
            for row in rows[:4000]:
                s, e = int(row[0]), int(row[1])
                pos, run_end = s, {}
                for k, (is_run, member) in enumerate(chain):
                    assert pos < len(h) and member[h[pos]], (pat, s, k)
                    pos += 1
                    while is_run and pos < len(h) and member[h[pos]]:
                        pos += 1
                    run_end[k] = pos
                assert pos == e, (pat, s, e, pos)
                src_pos = {0: s, 1: e}
                src_pos.update({2 + i: run_end[op] for i, op in enumerate(caps["run_op"])})
                for q, (src, off) in enumerate(caps["slots"]):
                    exp = -1 if src == 7 else src_pos[src] + off
                    assert int(row[q]) == exp, (pat, s, q, int(row[q]), exp)
    assert n_on >= 8


def test_program_routing_table():
    """Which device image a pattern gets (README "Which kernel a program gets"): kind and flags of the program blob.
    kind 1 digit / 2 DFA pair / 3 char class / 4 literal set; flags 16 ordered chain, 256 required literal prefix,
    128 UseBoth restart span."""
    import struct

    def image(pat, sub=False):
        rx = cx.compile(pat)
        assert rx.submatch_supported if sub else rx.supported, (pat, rx.why_unsupported)
        blob = rx.submatch_blobs()[0] if sub else rx.blob()
        return rx.strategy, struct.unpack_from("<I", blob, 4)[0], struct.unpack_from("<I", blob, 8)[0]

    chain, prefix, both = 16, 256, 128
    for pat, strategy, kind, must, must_not in [
            (r"\d+\.\d+\.\d+\.\d+", "UseDigitPrefilter", 1, chain, 0),
            (r"\d+:\d+:\d+", "UseDigitPrefilter", 1, chain, 0),
            (r"\d{4}-\d{2}-\d{2}T\d{2}:\d{2}:\d{2}", "UseDigitPrefilter", 1, chain, 0),
            (r"error", "UseDFA", 2, chain, prefix),
            (r"[A-Z]{3}-\d{4}", "UseDFA", 2, chain, prefix),
            (r"[0-9a-f]{8}-[0-9a-f]{4}-[0-9a-f]{4}-[0-9a-f]{4}-[0-9a-f]{12}", "UseBoth", 2, chain | both, prefix),
            (r"warning", "UseDFA", 4, 0, chain),
            (r"error|warning|fatal|critical", "UseTeddy", 4, 0, chain),
            (r"[\w]+", "UseCharClassSearcher", 3, 64, 0),
            (r"HTTP/\d\.\d", "UseDFA", 2, prefix, chain),
            (r"(?:GET|POST|PUT) /[a-z/]+", "UseDFA", 2, prefix, chain),
            (r"a?c", "UseDFA", 2, 0, chain | prefix)]:
        got = image(pat)
        assert got[0] == strategy and got[1] == kind and (got[2] & must) == must and (got[2] & must_not) == 0, (pat, got)
    assert image(r"(\w+)@(\w+)\.(\w+)", sub=True)[2] & chain and cx.compile(r"(\w+)@(\w+)\.(\w+)").chain_captures() is not None
    assert image(r"(GET|POST|PUT) /([a-z/]+)", sub=True)[2] & prefix
    for pat, why in [(r"a?(a|b)", "cache history"), (r"\w+@\w+\.\w+", "has no device kernel"), (r"\b(foo|bar|bazz|quux)\b", "cache history"),
                     (r"^foo", "anchor")]:
        rx = cx.compile(pat)
        assert not rx.supported and why in rx.why_unsupported, (pat, rx.strategy, rx.why_unsupported)
    wb = cx.compile(r"\bfoo\b")                       # a literal between assertions (round 4): UseNFA, literal image with looks + the transducer as fallback
    assert wb.supported and wb.strategy == "UseNFA" and image(r"\bfoo\b")[1] == 4 and wb.fsm_image() is not None
    wb2 = cx.compile(r"\b\w+=\d\b")                  # other small word-boundary patterns: transducer kernel only (kind 5, no tables)
    assert wb2.supported and wb2.strategy == "UseNFA" and image(r"\b\w+=\d\b")[1] == 5 and wb2.fsm_image() is not None


def test_bounded_repetition_chain_emulated(oracle):
    """`\\d{1,3}\\.\\d{1,3}\\.\\d{1,3}\\.\\d{1,3}` and friends: the chain kernel evaluates the surrogate with unbounded runs and filters
    the rows by field length (scan_chain_wave.hip BND; program.cc attachBoundedChain; frontend.cc boundedSurrogate).  The
    twin must equal the oracle whenever it does not hand the scan over (no synchronising byte in a halo, or a last field
    longer than its bound: FindAll would resume inside the run)."""
    import struct
    rng = np.random.default_rng(41)
    alphabet = np.frombuffer(b"0123456789.. x-:ab", dtype=np.uint8)
    w = np.array([1] * 10 + [4, 4, 1, 1, 1, 1, 1, 1], dtype=float)
    w /= w.sum()
    synth = cx.synth_pages(2, 0xC0FFEE02, 5, 24).tobytes()
    n_ok = 0
    for pat, fields in [(r"\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}", [(1, 3)] * 4), (r"\d{1,3}(?:\.\d{1,3}){3}", [(1, 3)] * 4),
                        (r"\d{2,4}-\d{1,2}", [(2, 4), (1, 2)]), (r"\d{1,2}:\d{2,}:\d+", [(1, 2), (2, 0), (1, 0)])]:
        rx, o = cx.compile(pat), oracle.Regex(pat)
        assert rx.supported and rx.strategy == o.strategy
        assert struct.unpack_from("<I", rx.blob(), 8)[0] & 512, pat
        raw, got_fields = rx.chain_bounds()
        assert got_fields == fields, (pat, got_fields)
        hays = [synth, b"", b"1234.5.6.7 1.2.3.4 999.999.999.999x", b"10.0.0.1 - 1.22.333.4 - 1.2.3", b"12-3 1234-56 12345-6 1-2"]
        hays += [alphabet[rng.choice(len(alphabet), size=int(rng.integers(0, 5000)), p=w)].tobytes() for _ in range(40)]
        exact = 0
        for hay in hays:
            exp = o.find_all_index(hay).tolist()
            assert emu.find_all(rx.blob(), hay).tolist() == exp, (pat, len(hay))          # the table-walking fallback
            for geom in ((3840, 256), (192, 64)):
                got = emu.find_all_chain6_bounded(rx.blob(), raw, hay, *geom)
                if isinstance(got, int):
                    assert (-got - 16) & ~(1 | 64) == 0, (pat, got)
                    continue
                exact += 1
                assert got.tolist() == exp, (pat, len(hay), geom)
        assert exact >= 10, (pat, exact)
        n_ok += 1
    assert n_ok == 4
    # the synthetic access-log corpus never needs the fallback: octets have one to three digits
    rx = cx.compile(r"\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}")
    got = emu.find_all_chain6_bounded(rx.blob(), rx.chain_bounds()[0], synth)
    assert not isinstance(got, int) and len(got) > 500
    # shapes the BND kernels do not take keep their table-walking image
    for pat in (r"x\d{1,3}", r"\d{1,3}[a-z]{1,3}\d{1,3}[a-z]{1,2}\d+[a-z]+\d+[a-z]+\d+"):
        rx = cx.compile(pat)
        assert rx.chain_bounds() is None or not rx.supported, pat


def test_emulated_no_sync_bytes_at_all(oracle):
    """A haystack made only of pattern-alphabet bytes: one lane walks everything, results still exact."""
    pat = r"\d+\.\d+\.\d+\.\d+"
    p = cx.compile(pat)
    hay = (b"1.2.3.4.5.6.7.8.9..10.11.12.13" * 40)
    assert emu.find_all(p.blob(), hay, 8).tolist() == oracle.Regex(pat).find_all_index(hay).tolist()
    assert emu.find_all(p.blob(), hay, 8, flat=1).tolist() == oracle.Regex(pat).find_all_index(hay).tolist()
    assert _fast_digit(p)
    assert emu.find_all(p.blob(), hay, 8, flat=2).tolist() == oracle.Regex(pat).find_all_index(hay).tolist()
    # fast-digit preconditions: run-skip safe AND tail closed
    assert not _fast_digit(cx.compile(r"\d+\.\d+x?"))       # a match (ending in x) can end in front of a digit
    assert not _fast_digit(cx.compile(r"\d{1,3}\.\d{1,3}"))   # leading class is bounded: no run skip
    assert _fast_digit(cx.compile(r"\d+\.\d+\.\d+")) and _fast_digit(cx.compile(r"\d+:\d+:\d+"))


SUBMATCH_PATTERNS = [r"(\w+)@(\w+)\.(\w+)", r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"(a+)(b+)", r"([a-z]+)([0-9]+)", r"(\d{4})-(\d{2})-(\d{2})",
                     r"(\w+)", r"(foo|bar)baz", r"(a)(b)?c", r"x(ab)+y", r"(\w+)=(\d+)", r"((a+)(b+))"]


def test_submatch_programs_emulated(oracle):
    """FindAllSubmatchIndex: bidirectional spans + one-pass capture table vs the oracle's PikeVM slots."""
    corpus = generate_test_input()
    synth = cx.synth_pages(5, 0xC0FFEE05, 0, 32).tobytes()
    rng = np.random.default_rng(99)
    alphabet = np.frombuffer(b"0123456789.-= ab\ncxy@_fo", dtype=np.uint8)
    checked = 0
    for pat in SUBMATCH_PATTERNS:
        p = cx.compile(pat)
        if not p.submatch_supported:
            continue
        o = oracle.Regex(pat)
        sb, cb = p.submatch_blobs()
        w = 2 * p.num_groups
        for hay in (corpus, synth):
            exp = o.find_all_submatch_index(hay)
            got = emu.find_all_submatch(sb, cb, hay, w, 64)
            assert got.tolist() == exp.tolist(), pat
        for _ in range(60):
            hay = alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(0, 300)))].tobytes()
            assert emu.find_all_submatch(sb, cb, hay, w, 4).tolist() == o.find_all_submatch_index(hay).tolist(), (pat, hay)
        checked += 1
    assert checked >= 8
    assert cx.compile(r"(\w+)@(\w+)\.(\w+)").submatch_supported
    # not one-pass (two NFA paths accept 'a'; `a+` then `a*`): the general capture pass — bounded backtracking over the NFA
    # per match row, device/bt.hpp — gives the PikeVM's slots
    import struct
    for pat in (r"(a|ab)(c|bcd)", r"(a+)(a*)", r"(a*)(a+)b", r"((a)|(ab))((c)|(bcd))", r"(ab|a)(bc|c)?", r"(a+|b+)*c", r"(\w+)=(\w+|\d+)"):
        p = cx.compile(pat)
        assert p.submatch_supported, pat
        sb, cb = p.submatch_blobs()
        if pat != r"(a+|b+)*c":
            assert struct.unpack_from("<I", cb, 0)[0] == 0x43584254, pat      # "CXBT": the backtracking image
        o = oracle.Regex(pat)
        w = 2 * p.num_groups
        for _ in range(80):
            hay = alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(0, 300)))].tobytes()
            assert emu.find_all_submatch(sb, cb, hay, w, 4).tolist() == o.find_all_submatch_index(hay).tolist(), (pat, hay)
        hay = (b"abcd ab abc abcd aaaa b a=1 ab=cd " * 200)
        assert emu.find_all_submatch(sb, cb, hay, w, 64).tolist() == o.find_all_submatch_index(hay).tolist(), pat


FAT_WORDS = ["word%02d" % i for i in range(20)] + ["key%02dx" % i for i in range(12)] + ["val%d" % i for i in range(10)] + \
            ["item", "timeout", "refused", "denied", "ordinal", "keyword"]


def test_fat_teddy_programs(oracle):
    """33..64 exact literals: UseTeddy through Fat Teddy in the reference (newTeddyFromSeq, prefilter/teddy.go:629-660);
    the device image folds the 16 buckets onto 8 and verifies exactly.  65 literals go to Aho-Corasick: refused."""
    assert len(FAT_WORDS) == 48
    pat = "|".join(FAT_WORDS)
    p = cx.compile(pat)
    o = oracle.Regex(pat)
    assert o.strategy == "UseTeddy" and o.strategy_restated
    assert p.strategy == "UseTeddy" and p.supported, p.why_unsupported
    rng = np.random.default_rng(48)
    pieces = [w.encode() for w in FAT_WORDS] + [b"word", b"wor", b"key1", b"val", b" ", b"\n", b"word9", b"xx", b"ite", b"keywor"]
    for _ in range(20):
        hay = b"".join(pieces[i] for i in rng.integers(0, len(pieces), size=int(rng.integers(0, 3000))))
        exp = o.find_all_index(hay).tolist()
        for chunk in (4, 64):
            assert emu.find_all(p.blob(), hay, chunk).tolist() == exp
    for n in (33, 64):
        words = ["lit%02dz" % i for i in range(n)]
        q = cx.compile("|".join(words))
        assert q.strategy == "UseTeddy" and q.supported, (n, q.why_unsupported)
        hay = (" ".join(words[::-1]) + " lit0 lit00 lit99z").encode()
        assert emu.find_all(q.blob(), hay, 4).tolist() == oracle.Regex("|".join(words)).find_all_index(hay).tolist()
    # 65 literals with a common prefix: regexp/syntax factors it, the NFA passes 200 states: UseNFA (PikeVM) in the reference —
    # the transducer kernel serves it since round 2 (plain leftmost-first), no literal kernel involved
    words = ["lit%02dz" % i for i in range(65)]
    q = cx.compile("|".join(words))
    oq = oracle.Regex("|".join(words))
    assert q.strategy == "UseNFA" and oq.strategy == "UseNFA" and q.supported and q.fsm_image() is not None
    hay = ("    ".join(words[::-1]) + " lit0 lit00 lit99z").encode()
    assert emu.find_all_fsm(q.fsm_image(), hay, 256, 32).tolist() == oq.find_all_index(hay).tolist()


def test_teddy_programs(oracle):
    """Teddy: prefix-free sets are accepted and reproduce the oracle; overlapping sets are refused."""
    lits16 = "error|warning|fatal|critical|panic|timeout|refused|denied|googlebot|bingbot|yandexbot|crawler|spider|failure|exception|overflow"
    p = cx.compile(lits16)
    assert p.strategy == "UseTeddy" and p.supported
    o = oracle.Regex(lits16)
    assert o.strategy == "UseTeddy" and o.strategy_restated
    hay = cx.synth_pages(3, 0xC0FFEE03, 0, 64).tobytes()
    exp = o.find_all_index(hay).tolist()
    assert len(exp) > 500
    for chunk in (4, 64):
        assert emu.find_all(p.blob(), hay, chunk).tolist() == exp
    tricky = b"errorerrorwarningwarnin fatalfatalities criticalcritica panicpanic xerrory erro error"
    assert emu.find_all(p.blob(), tricky, 4).tolist() == o.find_all_index(tricky).tolist()
    q = cx.compile("foobar|bazz|foo")   # not adjacent: regexp/syntax does not factor it, literals stay exact
    assert q.strategy == "UseTeddy" and not q.supported and "prefix-free" in q.why_unsupported


def test_synth_corpus_is_frozen():
    """synthlog-v1 must not drift: checksum of the first 16 pages of every config."""
    import zlib
    sums = {c: zlib.crc32(cx.synth_pages(c, 0xC0FFEE00 + c, 0, 16).tobytes()) for c in (1, 2, 3, 4, 5)}
    frozen = SYNTH_CRC
    assert sums == frozen, sums
    pg = cx.synth_pages(2, 0xC0FFEE02, 5, 2)
    assert pg[4095] == 10 and pg[8191] == 10 and pg.max() < 128


SYNTH_CRC = {1: 1650641072, 2: 4080580921, 3: 4062772577, 4: 2183380067, 5: 32623048}


def test_host_side_under_address_and_ub_sanitizers():
    """tests/emu/san_driver: front-end, program builder, transducer builder and the transducer twin compiled with
    g++ -fsanitize=address,undefined, fed with the golden patterns, random (also malformed) patterns and damaged NFAs."""
    import subprocess, sys as _sys
    r = subprocess.run([_sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "cpu_sanitize.py"), "600", "11"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0 and b"no sanitizer report" in r.stdout, r.stdout.decode()[-3000:]


def test_product_frontend_pinned_on_reference_strategy_rows():
    """`cxg_compile` (the product's own C++ front-end, not the oracle) directly against the rows of the
    reference's strategy table (meta/strategy_selection_test.go:16-63, :323-352, transcribed in tests/golden/reference_vectors.json).
    Every row gives the reference's strategy — since late round 3 also the three end-of-text patterns (UseReverseAnchored: the
    anchor rules at the head of meta.SelectStrategy, frontend.cc textAnchorStrategy), which stay refused for the device as every
    strategy without a kernel."""
    import json
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))
    rows = vec["strategy_selection"]["cases"]
    assert len(rows) >= 27
    for c in rows:
        rx = cx.compile(c["pattern"])
        assert rx.strategy == c["want"], (c, rx.strategy)
        if c["want"] == "UseReverseAnchored":
            assert not rx.supported and "UseReverseAnchored" in rx.why_unsupported
    # the refinements of nfa.IsPatternEndAnchored / IsPatternStartAnchored (nfa/compile.go:1785-1926): an end anchor that is not
    # the pattern's last element, or a start anchor of either kind in any branch, rule UseReverseAnchored out
    for pat, rev in [("(foo$)", True), (r"x\z", True), ("a$|b$", True), ("(a$)b$", False), ("a$|b", False), (r"(?m)^foo\z", False), ("^a?$|^b?$", False)]:
        assert (cx.compile(pat).strategy == "UseReverseAnchored") == rev, pat
    for pat in ("^foo", r"\Afoo", "^foo$", "(^a)b"):
        rx = cx.compile(pat)
        assert rx.strategy == "UseBoundedBacktracker" and not rx.supported, pat      # (or UseAnchoredLiteral / UseBranchDispatch: not restated)


def _teddy_literals_of(blob: bytes):
    """The literal table of a UseTeddy program image (walk.hpp BlobHeader / TeddyAux): literals by id."""
    magic, kind, flags, ngroups, fs, fst, ffa, foff, rs, rst, rfa, roff, info_off, total, aux_off, aux_len = struct.unpack_from("<16I", blob, 0)
    assert kind == 4, kind                                           # kKindTeddy
    nlits, nb, mn, mx, ab_off, order_off, lens_off, bucket_off, off_off, bytes_off, bytes_len, _pad = struct.unpack_from("<12I", blob, aux_off)
    lens = blob[aux_off + lens_off: aux_off + lens_off + nlits]
    offs = struct.unpack_from("<%dH" % nlits, blob, aux_off + off_off)
    return [bytes(blob[aux_off + bytes_off + offs[i]: aux_off + bytes_off + offs[i] + lens[i]]) for i in range(nlits)]


def test_product_literal_sets_pinned_on_reference_rows():
    """The literal sets the product front-end hands to the Teddy kernels (program image of UseTeddy patterns) against the
    reference's extractor table (literal/extractor_test.go, prefix rows of plain alternations): the same literals."""
    import json
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))
    checked = 0
    for c in vec["literal_extraction"]["cases"]:
        if c["which"] != "prefix":
            continue
        try:
            rx = cx.compile(c["pattern"])
            if rx.strategy != "UseTeddy" or not rx.supported:
                continue
            blob = rx.blob()
        except cx.CoregexError:
            continue
        assert sorted(_teddy_literals_of(blob)) == sorted(x.encode("latin-1") for x in c["want"]), c
        checked += 1
    assert checked >= 3, checked


def test_use_both_restart_rule_equals_the_reference_iteration(oracle):
    """capi_ladder.hip scanDevice (round 3) answers a UseBoth program WITHOUT a usable prefilter by iterating plain leftmost-first rows
    and, at the first match longer than 100 bytes, restarting the search at that match's end - 100 (find_indices.go:432-441: the
    DFA's end only picks where the PikeVM starts).  The same loop in Python over the oracle's plain leftmost-first spans (its
    PikeVM: FindAllSubmatch of a UseBoth program does not restart) must reproduce the oracle's UseBoth FindAllIndex.  (This test found that the first row of a
    restarted search must be exempt from the rule: the PikeVM's answer from end - 100 can be the next long match, in full.)"""
    rng = np.random.default_rng(11)

    def by_rule(o, hay):
        rows, off = [], 0
        for _ in range(200):
            plain = o.find_all_submatch_index(hay[off:])[:, :2] + off
            lens = plain[:, 1] - plain[:, 0]
            if off and len(lens):
                lens[0] = 0                                    # the first row of a restarted search is the PikeVM's answer: it stands
            long_ = np.nonzero(lens > 100)[0]
            if len(long_) == 0:
                rows.extend(plain.tolist())
                return np.array(rows, dtype=np.int64).reshape(-1, 2)
            k = int(long_[0])
            rows.extend(plain[:k].tolist())
            nxt = int(plain[k, 1]) - 100
            assert nxt > off
            off = nxt
        raise AssertionError("no progress")

    for pat in (r"(\w+)@(\w+)\.(\w+)", r"(xy|ab|ca)\w+(ab)+", r"(\w+)=(\w+);(\w+)"):
        rx, o = cx.compile(pat), oracle.Regex(pat)
        assert rx.strategy == o.strategy == "UseBoth" and not (rx.flags & 4)
        toks = [b"a@b.c", b"k=v;w", b"xyab", b"u" * 130 + b"@host.example.org", b"k" * 250 + b"@" + b"l" * 250 + b"." + b"m" * 250, b"ab" * 90, b"ca" + b"z" * 140 + b"ab",
                b"key" * 40 + b"=" + b"v" * 70 + b";" + b"w" * 10, b" ", b"\n", b"--", b"x@y", b"q@r.st"]
        for _ in range(60):
            hay = b" ".join(toks[i] for i in rng.integers(0, len(toks), int(rng.integers(1, 25))))
            a = np.frombuffer(hay, dtype=np.uint8)
            exp = o.find_all_index(a)
            got = by_rule(o, a)
            assert got.shape == exp.shape and np.array_equal(got, exp), (pat, hay[:80])


def test_wide_rows_of_the_reference_compat_table(oracle):
    """The rows of meta/stdlib_compat_test.go:27-67 with `.`, `\\S`, negated classes or text anchors (refcorpus.COMPAT_PATTERNS_WIDE):
    the product front-end names the oracle's strategy for each — reverse searchers, the multi-line reverse suffix, the engines of
    start-anchored patterns —, refuses what has no device kernel, and the rows it serves come out of the transducer twin as the
    golden spans (tests/golden/corpus_expected.json, stdlib semantics)."""
    import json
    from refcorpus import COMPAT_PATTERNS_WIDE, generate_test_input, span_hash
    corpus = np.frombuffer(generate_test_input(), dtype=np.uint8)
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "corpus_expected.json")))["patterns_wide"]
    served = 0
    for name, pat in COMPAT_PATTERNS_WIDE.items():
        rx, o = cx.compile(pat), oracle.Regex(pat)
        assert rx.strategy == o.strategy, (name, rx.strategy, o.strategy)
        if not rx.supported:
            assert rx.why_unsupported, name
            continue
        served += 1
        got = emu.find_all_fsm(rx.fsm_image(), corpus, 3840, 32)
        assert not isinstance(got, int) and len(got) == gold[name]["count"] and "%016x" % span_hash(got) == gold[name]["hash"], name
    assert served >= 1


def test_unicode_case_folding_is_refused_not_skipped(oracle):
    """Classes past U+007F are served (late round 3) — but not under (?i): unicode.SimpleFold is restated for the ASCII letters and
    the two runes their orbits reach (U+017F, U+212A) only, and `(?i)[é]` is {É, é} in the reference.  Both parsers refuse a rune
    past U+007F written under (?i) instead of compiling the class unfolded; ASCII classes still fold, Kelvin sign included."""
    for pat in (r"1(?i:[éa])2", r"(?i)é", r"(?i)[^é]a", r"(?i)[à-ü]+\d", r"x(?i:é)"):
        with pytest.raises(cx.CoregexError):
            cx.compile(pat)
        with pytest.raises(oracle.OracleError):
            oracle.Regex(pat)
    rx, o = cx.compile(r"1(?i:[ka])2"), oracle.Regex(r"1(?i:[ka])2")
    assert rx.strategy == o.strategy == "UseTeddy" and rx.supported
    assert sorted(b for b, _ in o.prefix_literals()) == sorted([b"1A2", b"1a2", b"1K2", b"1k2", "1\u212a2".encode()])
    assert cx.compile(r"[éa]+x").supported or True                  # (outside (?i): served or refused by strategy, never a parse error)


def test_case_insensitive_programs_through_the_twins(oracle):
    """The CPU half of tests/test_zzz_gpu_fold.py::test_case_insensitive_literals: the same patterns and the same word-rich text through the
    sequential twins of the kernels their programs name (lane walks, Teddy wave twin, transducer twin, capture twin)."""
    import struct
    from refcorpus import generate_test_input
    from test_zzz_gpu_fold import FOLD
    rng = np.random.default_rng(3)
    toks = [w for base in (b"error", b"hello", b"login", b"auth", b"signin", b"sign-in", b"session", b"fail", b"panic", b"fatal", b"exception", b"warn", b"warning",
                           b"xyzw", b"k1", b"googlebot", b"bingbot", b"GET /index", b"error: disk")
            for w in (base, base.upper(), base.capitalize(), base[:1] + base[1:].upper())] + [b" ", b"\n", b": ", "K1".encode(), "ſession".encode()]
    words = b"".join(toks[int(i)] for i in rng.integers(0, len(toks), size=6000))
    for pat in FOLD:
        rx, o = cx.compile(pat), oracle.Regex(pat)
        assert rx.supported and rx.strategy == o.strategy, (pat, rx.why_unsupported)
        blob = rx.blob()
        kind, flags = struct.unpack_from("<II", blob, 4)
        for hay in (generate_test_input()[:60000], words, b"", b"ERROR"):
            exp = o.find_all_index(hay).tolist()
            a = np.frombuffer(hay, dtype=np.uint8)
            folded = kind == 4 and struct.unpack_from("<I", blob, struct.unpack_from("<I", blob, 56)[0] + 44)[0] != 0   # TeddyAux::looks (round 4: folded sets)
            if kind != 5 and not folded:                              # (the table kernel knows neither assertions nor folded sets: capi_ladder.hip never sends them there)
                assert emu.find_all(blob, hay).tolist() == exp, (pat, "lanes", len(hay))
            if rx.fsm_image() is not None:
                got = emu.find_all_fsm(rx.fsm_image(), a, 3840, 32)
                if isinstance(got, int) and got in (-18, -32):
                    got = emu.find_all_fsm(rx.fsm_image(), a, 3840, 32, dense=1)
                assert isinstance(got, int) or got.tolist() == exp, (pat, "transducer", len(hay))
            if kind == 4 or (flags & 256):
                got = emu.find_all_teddy_wave(blob, hay)
                assert got is None or isinstance(got, int) or got.tolist() == exp, (pat, "teddy wave", len(hay))
        if rx.num_groups > 1 and rx.submatch_supported:
            sb, cb = rx.submatch_blobs()[:2]
            exps = o.find_all_submatch_index(words)
            got = emu.find_all_submatch(sb, cb, words, exps.shape[1])
            assert got.shape == exps.shape and np.array_equal(got, exps), (pat, "captures")


def test_bounded_backtracker_programs_through_the_twins(oracle):
    """The CPU half of tests/test_zzz_gpu_fold.py::test_bounded_backtracker_programs, and the reference's own rows for this strategy:
    `[a-f0-9]{32,}` and `(\\w{2,8})+` on its differential corpus (stdlib answers, tests/golden/corpus_expected.json)."""
    import json
    import struct
    from refcorpus import COMPAT_PATTERNS, generate_test_input, span_hash
    from test_zzz_gpu_fold import BOUNDED
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "corpus_expected.json")))["patterns"]
    corpus = np.frombuffer(generate_test_input(), dtype=np.uint8)
    rng = np.random.default_rng(5)
    mixed = np.frombuffer(b"abcdef0123456789 ,.\n-_XYZ" + "é日".encode() + b"\x80\xff", dtype=np.uint8)
    served = 0
    for pat in BOUNDED + [COMPAT_PATTERNS["la_tokens"], COMPAT_PATTERNS["word_repeat"]]:
        rx, o = cx.compile(pat), oracle.Regex(pat)
        assert rx.strategy == o.strategy == "UseBoundedBacktracker" and o.strategy_restated, pat
        if not rx.supported:
            continue
        served += 1
        blob = rx.blob()
        for hay in (corpus[:50000], mixed[rng.integers(0, len(mixed), size=6000)], mixed[rng.integers(0, 6, size=3000)], np.zeros(0, dtype=np.uint8)):
            exp = o.find_all_index(hay).tolist()
            if struct.unpack_from("<I", blob, 4)[0] == 3:           # round 4: `C+` programs (`\\S+`, `[^,]+`) are the char-class kernels'
                got = emu.find_all_charclass_wave(blob, hay) if struct.unpack_from("<I", blob, 8)[0] & 64 else None
                assert got is None or isinstance(got, int) or got.tolist() == exp, (pat, "class runs", len(hay))
                continue
            assert emu.find_all(blob, hay).tolist() == exp, (pat, "lanes", len(hay))
            if rx.fsm_image() is not None:
                got = emu.find_all_fsm(rx.fsm_image(), hay, 3840, 32)
                if isinstance(got, int) and got in (-18, -32):
                    got = emu.find_all_fsm(rx.fsm_image(), hay, 3840, 32, dense=1)
                assert isinstance(got, int) or got.tolist() == exp, (pat, "transducer", len(hay))
    assert served >= 10
    for name in ("la_tokens", "word_repeat"):
        rx = cx.compile(COMPAT_PATTERNS[name])
        if rx.supported:
            got = emu.find_all_charclass_wave(rx.blob(), corpus) if struct.unpack_from("<I", rx.blob(), 4)[0] == 3 else emu.find_all(rx.blob(), corpus)
            assert len(got) == gold[name]["count"] and "%016x" % span_hash(got) == gold[name]["hash"], name


def test_golden_find_all_rows_through_the_twins(oracle):
    """Every FindAll-shaped group of tests/golden/reference_vectors.json that came from the reference's differential tests — anchors inside
    alternations, look-around, case folding, the real-world patterns of edge_cases_test.go:370-400: the front-end names the oracle's
    strategy, and what it serves gives the golden spans on the sequential twin of the kernel its first launch would run (tests/twins.py)."""
    import json
    from twins import rows_on_twin
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))
    served = 0
    for group in ("real_world_compat", "text_anchor_compat", "text_anchor_compat_oracle_only", "lookaround_compat", "lookaround_compat_more", "case_folding_find_all_string",
                  "edge_case_pairs", "findall_string_kat"):
        for c in vec[group]["cases"]:
            rx, o = cx.compile(c["pattern"]), oracle.Regex(c["pattern"])
            assert rx.strategy == o.strategy, (group, c)
            if not rx.supported:
                continue
            served += 1
            hay = c["input"].encode()
            got = rows_on_twin(rx, hay)
            assert not isinstance(got, int), (group, c, got)
            want = [w[:2] for w in c["want"]] if not c["want"] or isinstance(c["want"][0], list) else None
            if want is None:                                          # (FindAllString rows: the matched strings)
                assert [hay[s:e].decode() for s, e in got.tolist()] == c["want"], (group, c)
            else:
                assert got.tolist() == want, (group, c, got.tolist())
    assert served >= 60, served


def test_stdlib_find_tests_through_the_front_end_and_the_twins(oracle):
    """tests/golden "stdlib_find_tests" (the reference's copy of Go's find_test table): the front-end names the oracle's strategy for every
    row; the rows it serves give the table's spans on the twin of their first kernel, and the table's capture rows on the capture twins."""
    import json
    from twins import rows_on_twin
    blk = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))["stdlib_find_tests"]
    served = caps = 0
    for c in blk["cases"]:
        hay = bytes.fromhex(c["input_hex"])
        try:
            rx = cx.compile(c["pattern"])
        except cx.CoregexError:
            continue                                                  # (a Unicode class the front-end does not parse: refused, never mis-served)
        assert rx.strategy == oracle.Regex(c["pattern"]).strategy, c
        if rx.supported:
            served += 1
            got = rows_on_twin(rx, hay)
            assert not isinstance(got, int) and got.tolist() == [w[:2] for w in c["want"]], (c, got)
        if rx.num_groups > 1 and rx.submatch_supported and c["pattern"] not in blk["submatch_not_asserted"]:
            want = np.array(c["want"], dtype=np.int64).reshape(-1, 2 * rx.num_groups)
            oc = rx.offset_captures
            if oc is not None:
                for k, (src, d) in enumerate(oc):
                    assert np.array_equal(want[:, k], want[:, 1 if src else 0] + d), (c, k)
            elif rx.nullable:                                          # round 5: FindAllIndex rows + the backtracking pass for every row + the end-of-haystack quirk
                got = emu.captures_bt(rx.submatch_blobs()[1], hay, rows_on_twin(rx, hay), want.shape[1])
                if len(got) and got[-1][0] == got[-1][1] == len(hay):
                    got[-1][2:] = -1
                assert np.array_equal(got, want), (c, got.tolist())
            else:
                sb, cb = rx.submatch_blobs()[:2]
                got = emu.find_all_submatch(sb, cb, hay, want.shape[1]) if len(want) else want
                assert np.array_equal(got, want), (c, got.tolist())
            caps += 1
    assert served >= 25 and caps >= 3, (served, caps)


def test_fuzz_seed_matrix_through_the_front_end_and_the_twins(oracle):
    """tests/golden "fuzz_seed_matrix" (the seed corpus of the reference's FuzzFindAllStdlib, 56 patterns x 27 inputs): the front-end names the
    oracle's strategy for every pattern, and every pattern it serves gives the matrix's rows on the twin of its first kernel for all 27 inputs."""
    import json
    from twins import rows_on_twin
    blk = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))["fuzz_seed_matrix"]
    served = rows = 0
    for pi, pat in enumerate(blk["patterns"]):
        rx = cx.compile(pat)
        assert rx.strategy == oracle.Regex(pat).strategy, pat
        if not rx.supported:
            continue
        served += 1
        for ii, inp in enumerate(blk["inputs"]):
            got = rows_on_twin(rx, inp.encode())
            assert not isinstance(got, int) and got.tolist() == blk["want"][pi][ii], (pat, inp, got if isinstance(got, int) else got.tolist())
            rows += 1
    assert served >= 35 and rows >= 900, (served, rows)


def test_fuzz_seed_capture_rows_through_the_capture_twins(oracle):
    """tests/golden "fuzz_seed_submatch_first" (capture seeds of the reference's FuzzFindSubmatchStdlib): for the patterns whose captures the
    device serves, the first row of the capture twin (one-pass table / backtracking pass / offsets) is the golden row."""
    import json
    blk = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))["fuzz_seed_submatch_first"]
    served = rows = 0
    for pi, pat in enumerate(blk["patterns"]):
        rx, o = cx.compile(pat), oracle.Regex(pat)
        assert rx.strategy == o.strategy, pat
        if not rx.submatch_supported or rx.nullable:
            continue
        served += 1
        w = 2 * rx.num_groups
        oc = rx.offset_captures
        sb, cb = rx.submatch_blobs()[:2]
        for ii, inp in enumerate(blk["inputs"]):
            want = blk["want"][pi][ii]
            if want is None:
                continue
            hay = inp.encode()
            if oc is not None:
                if want:
                    assert all(want[k] == want[1 if src else 0] + d for k, (src, d) in enumerate(oc)), (pat, inp)
            else:
                got = emu.find_all_submatch(sb, cb, hay, w)
                assert (got[0].tolist() if len(got) else []) == want, (pat, inp, got[:1].tolist())
            rows += 1
    assert served >= 8 and rows >= 200, (served, rows)


def test_find_indices_all_strategies_rows_served(oracle):
    """tests/golden "find_indices_all_strategies": the front-end names the oracle's strategy for each row, and the rows it serves give the
    table's first match on the twins."""
    import json
    from twins import rows_on_twin
    served = 0
    for c in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))["find_indices_all_strategies"]["cases"]:
        rx = cx.compile(c["pattern"])
        assert rx.strategy == oracle.Regex(c["pattern"]).strategy, c
        if not rx.supported:
            continue
        served += 1
        got = rows_on_twin(rx, c["input"].encode())
        assert not isinstance(got, int) and (got[0].tolist() if len(got) else None) == c["want"], (c, got)
    assert served >= 12
