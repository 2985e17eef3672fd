"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front for ``oracle/liboracle.so``, the CPU restatement of the reference's
FindAll path (see the headers of oracle/*.hpp for the file:line map).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package ``coregex_amd`` never does.

Parity status: pinned against the reference's own known-answer vectors transcribed
under ``tests/golden/`` (tests/test_oracle_golden.py) — since round 4 also the reference's
copy of Go's find_test table (73 rows with their FindAllSubmatchIndex answers), the seed
matrix of its differential fuzz test (56 patterns x 27 inputs), its edge-case tables
(101 pairs) and its per-strategy first-match table — and against the differential
corpus recipe of meta/stdlib_compat_test.go:146-199.  The reference (Go) cannot be
built in this image (no Go toolchain), so there is no ``oracle/_ref``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

STRATEGY_NAMES = [
    "UseNFA", "UseDFA", "UseBoth", "UseReverseAnchored", "UseReverseSuffix", "UseOnePass",
    "UseReverseInner", "UseBoundedBacktracker", "UseTeddy", "UseReverseSuffixSet",
    "UseCharClassSearcher", "UseCompositeSearcher", "UseBranchDispatch", "UseDigitPrefilter",
    "UseAhoCorasick", "UseAnchoredLiteral", "UseMultilineReverseSuffix",
]


def build(force: bool = False) -> str:
    """Compile liboracle.so with make (g++)."""
    if force or not os.path.exists(_LIB_PATH) or _stale():
        subprocess.check_call(["make", "-s", "-j8", "-C", _HERE])
    return _LIB_PATH


def _stale() -> bool:
    t = os.path.getmtime(_LIB_PATH)
    for f in os.listdir(_HERE):
        if f.endswith((".cpp", ".hpp")) and os.path.getmtime(os.path.join(_HERE, f)) > t:
            return True
    return False


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, i64, u8p, i64p = C.c_void_p, C.c_int64, C.c_char_p, C.POINTER(C.c_int64)
        L.orc_last_error.restype = C.c_char_p
        L.orc_compile.restype = vp
        L.orc_compile.argtypes = [C.c_char_p, i64]
        L.orc_free.argtypes = [vp]
        for name in ("orc_strategy", "orc_strategy_restated", "orc_num_groups", "orc_nfa_states",
                     "orc_alphabet_len", "orc_dfa_states", "orc_digit_run_skip_safe",
                     "orc_num_prefix_literals"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [vp]
        L.orc_byte_classes.argtypes = [vp, C.c_void_p]
        L.orc_prefix_literal.restype = C.c_int
        L.orc_prefix_literal.argtypes = [vp, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_find_all.restype = i64
        L.orc_find_all.argtypes = [vp, C.c_void_p, i64, i64, C.c_void_p, i64]
        L.orc_find_all_submatch.restype = i64
        L.orc_find_all_submatch.argtypes = [vp, C.c_void_p, i64, i64, C.c_void_p, i64]
        L.orc_count.restype = i64
        L.orc_count.argtypes = [vp, C.c_void_p, i64, i64]
        L.orc_dfa_search_at_anchored.restype = i64
        L.orc_dfa_search_at_anchored.argtypes = [vp, C.c_void_p, i64, i64]
        L.orc_dfa_search_at.restype = i64
        L.orc_dfa_search_at.argtypes = [vp, C.c_void_p, i64, i64]
        L.orc_memchr_digit_at.restype = i64
        L.orc_memchr_digit_at.argtypes = [C.c_void_p, i64, i64]
        L.orc_pikevm_captures.restype = C.c_int
        L.orc_pikevm_captures.argtypes = [vp, C.c_void_p, i64, i64, C.c_void_p]
        L.orc_teddy_new.restype = vp
        L.orc_teddy_new.argtypes = [C.c_char_p, C.c_int]
        L.orc_teddy_free.argtypes = [vp]
        L.orc_teddy_find_match.restype = C.c_int
        L.orc_teddy_find_match.argtypes = [vp, C.c_void_p, i64, i64, i64p, i64p]
        L.orc_scan_synth.restype = C.c_int
        L.orc_scan_synth.argtypes = [C.c_char_p, i64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p]
        L.orc_dump.restype = C.c_int
        L.orc_dump.argtypes = [vp, C.c_char_p, C.c_int]
        _ = (u8p,)
        _lib = L
    return _lib


def _buf(hay):
    """Return (ctypes pointer value, length, keepalive) for bytes / numpy uint8."""
    if isinstance(hay, np.ndarray):
        a = np.ascontiguousarray(hay, dtype=np.uint8)
        return a.ctypes.data, a.size, a
    b = bytes(hay)
    a = np.frombuffer(b, dtype=np.uint8) if b else np.zeros(1, dtype=np.uint8)
    return a.ctypes.data, len(b), (a, b)


class OracleError(Exception):
    pass


class Regex:
    """Mirror of the reference's coregex.Regex surface on the FindAll path (regex.go:695-1450)."""

    def __init__(self, pattern):
        p = pattern.encode() if isinstance(pattern, str) else bytes(pattern)
        self.pattern = p
        self._h = lib().orc_compile(p, len(p))
        if not self._h:
            raise OracleError(lib().orc_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.orc_free(self._h)
            self._h = None

    @property
    def strategy(self) -> str:
        return STRATEGY_NAMES[lib().orc_strategy(self._h)]

    @property
    def strategy_restated(self) -> bool:
        return bool(lib().orc_strategy_restated(self._h))

    @property
    def num_groups(self) -> int:
        return lib().orc_num_groups(self._h)

    @property
    def nfa_states(self) -> int:
        return lib().orc_nfa_states(self._h)

    @property
    def alphabet_len(self) -> int:
        return lib().orc_alphabet_len(self._h)

    @property
    def digit_run_skip_safe(self) -> bool:
        return bool(lib().orc_digit_run_skip_safe(self._h))

    def byte_classes(self) -> np.ndarray:
        out = np.zeros(256, dtype=np.uint8)
        lib().orc_byte_classes(self._h, out.ctypes.data)
        return out

    def dfa_states(self) -> int:
        return lib().orc_dfa_states(self._h)

    def prefix_literals(self):
        out = []
        for i in range(lib().orc_num_prefix_literals(self._h)):
            buf = C.create_string_buffer(256)
            comp = C.c_int(0)
            n = lib().orc_prefix_literal(self._h, i, buf, 256, C.byref(comp))
            out.append((buf.raw[:n], bool(comp.value)))
        return out

    def _all(self, fn, hay, limit, width):
        ptr, n, keep = _buf(hay)
        cap = 1 << 12
        while True:
            out = np.empty(cap, dtype=np.int64)
            got = fn(self._h, ptr, n, limit, out.ctypes.data, cap)
            if got <= cap:
                del keep
                return out[:got].reshape(-1, width).copy()
            cap = int(got)

    def find_all_index(self, hay, n: int = -1) -> np.ndarray:
        """FindAllIndex(b, n) as an (M, 2) int64 array (n == 0 -> empty, regex.go:696)."""
        if n == 0:
            return np.zeros((0, 2), dtype=np.int64)
        return self._all(lib().orc_find_all, hay, n, 2)

    def find_all_submatch_index(self, hay, n: int = -1) -> np.ndarray:
        if n == 0:
            return np.zeros((0, 2 * self.num_groups), dtype=np.int64)
        return self._all(lib().orc_find_all_submatch, hay, n, 2 * self.num_groups)

    def count(self, hay, n: int = -1) -> int:
        ptr, ln, keep = _buf(hay)
        r = lib().orc_count(self._h, ptr, ln, n)
        del keep
        return int(r)

    def dfa_search_at_anchored(self, hay, at: int) -> int:
        ptr, ln, keep = _buf(hay)
        return int(lib().orc_dfa_search_at_anchored(self._h, ptr, ln, at))

    def dfa_search_at(self, hay, at: int) -> int:
        ptr, ln, keep = _buf(hay)
        return int(lib().orc_dfa_search_at(self._h, ptr, ln, at))

    def pikevm_captures(self, hay, at: int = 0):
        ptr, ln, keep = _buf(hay)
        slots = np.full(2 * self.num_groups, -1, dtype=np.int64)
        ok = lib().orc_pikevm_captures(self._h, ptr, ln, at, slots.ctypes.data)
        return slots if ok else None

    def dump(self) -> str:
        buf = C.create_string_buffer(1 << 20)
        lib().orc_dump(self._h, buf, 1 << 20)
        return buf.value.decode()


def scan_synth(pattern, config: int, seed: int, first_page: int, npages: int, nthreads: int = 0, width: int = 2):
    """The oracle over synthlog-v1 pages, multi-threaded (oracle/scale.cpp): returns dict(rows, sums[width], scan_s, gen_s,
    threads).  sums[j] = sum over rows k of value * (k + 1 + 7 j) mod 2^64, offsets relative to page `first_page`."""
    p = pattern.encode() if isinstance(pattern, str) else bytes(pattern)
    if nthreads <= 0:
        nthreads = max(1, len(os.sched_getaffinity(0)))
    out = np.zeros(19, dtype=np.uint64)
    rc = lib().orc_scan_synth(p, len(p), config, seed, first_page, npages, nthreads, width, out.ctypes.data)
    if rc != 0:
        raise OracleError("orc_scan_synth failed")
    return {"rows": int(out[0]), "sums": [int(x) for x in out[1:1 + width]], "scan_s": float(out[17]) * 1e-9,
            "gen_s": float(out[18]) * 1e-9, "threads": nthreads}


def memchr_digit_at(hay, at: int) -> int:
    ptr, ln, keep = _buf(hay)
    return int(lib().orc_memchr_digit_at(ptr, ln, at))


def extract_literals(pattern: str, which: str = "prefix"):
    """literal.Extractor of the reference on a parsed pattern (no engine): [(bytes, complete), ...]."""
    L = lib()
    L.orc_extract_literals.restype = C.c_int
    L.orc_extract_literals.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_void_p, C.c_int]
    p = pattern.encode()
    buf = (C.c_uint8 * 65536)()
    n = L.orc_extract_literals(p, len(p), {"prefix": 0, "suffix": 1, "inner": 2}[which], buf, 65536)
    if n < 0:
        raise OracleError(L.orc_last_error().decode() if n == -1 else "literal records do not fit")
    out, off = [], 0
    raw = bytes(buf)
    for _ in range(n):
        ln, comp = raw[off], raw[off + 1]
        out.append((raw[off + 2:off + 2 + ln], bool(comp)))
        off += 2 + ln
    return out


class Teddy:
    def __init__(self, patterns):
        packed = b"".join(bytes([len(p)]) + bytes(p) for p in patterns)
        self._h = lib().orc_teddy_new(packed, len(patterns))
        if not self._h:
            raise OracleError("NewTeddy returned nil")

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.orc_teddy_free(self._h)

    def find_match(self, hay, start: int = 0):
        ptr, ln, keep = _buf(hay)
        s, e = C.c_int64(-1), C.c_int64(-1)
        ok = lib().orc_teddy_find_match(self._h, ptr, ln, start, C.byref(s), C.byref(e))
        return (s.value, e.value) if ok else (-1, -1)

    def find(self, hay, start: int = 0) -> int:
        return self.find_match(hay, start)[0]
