"""coregex_amd — MI355X-native bulk FindAll for coregex (host-side mirror of the reference API).

Python stands where the reference's Go API would (no Go toolchain in this image): names and argument
meaning follow ``coregex.Regexp`` on the FindAll path (regex.go:695-1450) and ``meta.Engine``
(meta/findall.go:155,297,390).  All matching happens in ``libcoregex_hip.so`` on the GPU.

    rx = coregex_amd.compile(r"\\d+\\.\\d+\\.\\d+\\.\\d+")
    rx.strategy                    # 'UseDigitPrefilter'  (meta/strategy.go)
    rx.find_all_index(data, -1)    # (M, 2) int64, == Regexp.FindAllIndex
    rx.count(data, -1)             # == meta.Engine.Count
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import Timing  # noqa: F401

STRATEGY_NAMES = [
    "UseNFA", "UseDFA", "UseBoth", "UseReverseAnchored", "UseReverseSuffix", "UseOnePass",
    "UseReverseInner", "UseBoundedBacktracker", "UseTeddy", "UseReverseSuffixSet",
    "UseCharClassSearcher", "UseCompositeSearcher", "UseBranchDispatch", "UseDigitPrefilter",
    "UseAhoCorasick", "UseAnchoredLiteral", "UseMultilineReverseSuffix",
]


class CoregexError(Exception):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class UnsupportedPattern(CoregexError):
    """CXG_E_UNSUPPORTED: the caller keeps its CPU loop (mirrors the reference's degrade-don't-fail)."""


class UnsupportedInput(CoregexError):
    """CXG_E_INPUT: this haystack holds a long stretch without synchronising bytes; the caller keeps its CPU loop for
    this call (the compiled program stays usable for other haystacks)."""


def _check(rc: int):
    if rc == 0:
        return
    msg = _lib.lib().cxg_last_error().decode(errors="replace")
    if rc == _lib.CXG_E_UNSUPPORTED:
        raise UnsupportedPattern(rc, msg)
    if rc == _lib.CXG_E_INPUT:
        raise UnsupportedInput(rc, msg)
    raise CoregexError(rc, msg)


def device_count() -> int:
    return _lib.lib().cxg_device_count()


def set_device(i: int):
    _check(_lib.lib().cxg_set_device(i))


def path_state(device: int = 0) -> dict:
    """cxg_path_state: calls left on the slower launch mode and watchdog hits per mode (static groups, persistent grid, delimiter kernel)."""
    st = _lib.PathState()
    _check(_lib.lib().cxg_path_state(device, C.byref(st)))
    return {n: int(getattr(st, n)) for n, _ in _lib.PathState._fields_ if n != "reserved"}


def path_reset(device: int = 0):
    """cxg_path_reset: forget every launch-mode demotion of the device."""
    _check(_lib.lib().cxg_path_reset(device))


def _host_view(hay):
    if isinstance(hay, np.ndarray):
        a = np.ascontiguousarray(hay, dtype=np.uint8)
    else:
        b = bytes(hay)
        a = np.frombuffer(b, dtype=np.uint8) if b else np.zeros(0, dtype=np.uint8)
    return a


class Regex:
    """Mirror of coregex.Regexp on the FindAll path; every search runs on the GPU."""

    def __init__(self, handle, pattern: bytes):
        self._h = handle
        self.pattern = pattern

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and _lib._lib is not None:
            _lib._lib.cxg_program_destroy(self._h)
            self._h = None

    # --- introspection (meta.Engine.Strategy(), NumSubexp) ---
    @property
    def strategy(self) -> str:
        return STRATEGY_NAMES[_lib.lib().cxg_program_strategy(self._h)]

    @property
    def flags(self) -> int:
        return _lib.lib().cxg_program_flags(self._h)

    @property
    def num_groups(self) -> int:
        return _lib.lib().cxg_program_num_groups(self._h)

    @property
    def nfa_states(self) -> int:
        return _lib.lib().cxg_program_nfa_states(self._h)

    @property
    def dfa_states(self) -> int:
        return _lib.lib().cxg_program_dfa_states(self._h)

    @property
    def supported(self) -> bool:
        return bool(_lib.lib().cxg_program_supported(self._h))

    @property
    def offset_captures(self):
        """[(src, delta)] per capture slot — src 0 = match start, 1 = match end — when every capture boundary sits at a fixed distance
        from one of them (FindAllSubmatch = FindAll + an expansion kernel), else None."""
        src, delta = (C.c_int * 32)(), (C.c_int * 32)()
        n = _lib.lib().cxg_program_offset_captures(self._h, src, delta, 32)
        return [(src[k], delta[k]) for k in range(n)] if n else None

    @property
    def delimiters(self):
        """(open byte, close byte, plus) of an `O [^E]+ E` / `O [^E]* E` program, else None."""
        o, c, pl = C.c_int(0), C.c_int(0), C.c_int(0)
        if not _lib.lib().cxg_program_delimiters(self._h, C.byref(o), C.byref(c), C.byref(pl)):
            return None
        return o.value, c.value, bool(pl.value)

    @property
    def nullable(self) -> int:
        """0: not nullable; 1: the device program is the non-empty variant (empty matches merged behind the scan); 2: only empty matches."""
        return int(_lib.lib().cxg_program_nullable(self._h))

    @property
    def why_unsupported(self) -> str:
        if self.supported:
            return ""
        return _lib.lib().cxg_last_error().decode(errors="replace")

    @property
    def submatch_supported(self) -> bool:
        return bool(_lib.lib().cxg_program_submatch_supported(self._h))

    def submatch_blobs(self):
        sp, sn, cp, cn = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        _check(_lib.lib().cxg_program_submatch_blobs(self._h, C.byref(sp), C.byref(sn), C.byref(cp), C.byref(cn)))
        return C.string_at(sp, sn.value), C.string_at(cp, cn.value)

    def chain_captures(self):
        """None, or the ChainCaps record of a program whose capture slots are written by the chain kernel itself:
        dict(run_op=(a, b, ...), slots=[(source, offset), ...]) with source 0 start, 1 end, 2 + i end of run_op[i], 7 unset."""
        buf = C.create_string_buffer(40)
        if not _lib.lib().cxg_program_chain_captures(self._h, buf):
            return None
        raw = buf.raw
        n = raw[1]
        return {"run_op": tuple(raw[4:4 + raw[2]]), "slots": [(raw[8 + k], int.from_bytes(raw[24 + k:25 + k], "little", signed=True)) for k in range(n)]}

    def chain_bounds(self):
        """None, or (raw 40-byte record, [(min, max), ...] per field; max 0 = unbounded) of a bounded-repetition program
        served by the chain kernel."""
        buf = C.create_string_buffer(40)
        if not _lib.lib().cxg_program_chain_bounds(self._h, buf):
            return None
        raw = buf.raw
        return raw, [(raw[8 + f], raw[16 + f]) for f in range(raw[2] + 1)]

    def blob(self) -> bytes:
        p, n = C.c_void_p(), C.c_size_t()
        _check(_lib.lib().cxg_program_blob(self._h, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value)

    def fsm_image(self, submatch: bool = False):
        """None, or the FindAll transducer image (device/fsm.hpp) the general-DFA kernel runs for this program."""
        p, n = C.c_void_p(), C.c_size_t()
        if _lib.lib().cxg_program_fsm_image(self._h, 1 if submatch else 0, C.byref(p), C.byref(n)) != 0:
            return None
        return C.string_at(p, n.value)

    def nfa(self):
        """Host copy of the NFA as (states ndarray-of-tuples, trans, start_anchored, start_unanchored, captures)."""
        v = _lib.Nfa()
        _check(_lib.lib().cxg_program_nfa(self._h, C.byref(v)))
        return v

    # --- FindAll family over host bytes (what the cgo shim calls) ---
    def find_all_index(self, hay, n: int = -1) -> np.ndarray:
        """Regexp.FindAllIndex(b, n) as an (M, 2) int64 array; n == 0 -> empty (regex.go:696)."""
        if n == 0:
            return np.zeros((0, 2), dtype=np.int64)
        return self._rows(_lib.lib().cxg_find_all, hay, n, 2)

    def find_all_submatch_index(self, hay, n: int = -1) -> np.ndarray:
        w = 2 * self.num_groups
        if n == 0:
            return np.zeros((0, w), dtype=np.int64)
        return self._rows(_lib.lib().cxg_find_all_submatch, hay, n, w)

    def find_index(self, hay):
        """Regexp.FindIndex(b) (regex.go; meta/find.go:29 Engine.Find): (start, end) of the first match, or None."""
        a = _host_view(hay)
        span = (C.c_int64 * 2)(-1, -1)
        found = C.c_int(0)
        _check(_lib.lib().cxg_find(self._h, a.ctypes.data, a.size, span, C.byref(found)))
        return (int(span[0]), int(span[1])) if found.value else None

    def is_match(self, hay) -> bool:
        """Regexp.Match(b) (meta/ismatch.go:27 Engine.IsMatch)."""
        a = _host_view(hay)
        m = C.c_int(0)
        _check(_lib.lib().cxg_is_match(self._h, a.ctypes.data, a.size, C.byref(m)))
        return bool(m.value)

    def find_device(self, d_hay: int, length: int, base: int = 0, stream: int = 0):
        span = (C.c_int64 * 2)(-1, -1)
        found = C.c_int(0)
        _check(_lib.lib().cxg_find_device(self._h, d_hay, length, base, span, C.byref(found), stream or None))
        return (int(span[0]), int(span[1])) if found.value else None

    def is_match_device(self, d_hay: int, length: int, stream: int = 0) -> bool:
        m = C.c_int(0)
        _check(_lib.lib().cxg_is_match_device(self._h, d_hay, length, C.byref(m), stream or None))
        return bool(m.value)

    def count(self, hay, n: int = -1) -> int:
        a = _host_view(hay)
        out = C.c_uint64(0)
        _check(_lib.lib().cxg_count(self._h, a.ctypes.data, a.size, n, C.byref(out)))
        return int(out.value)

    def _rows(self, fn, hay, n, width):
        a = _host_view(hay)
        cap = max(1024, a.size // 64 + 16)
        while True:
            out = np.empty((cap, width), dtype=np.int64)
            got = C.c_uint64(0)
            rc = fn(self._h, a.ctypes.data, a.size, n, out.ctypes.data, cap, C.byref(got))
            if rc == _lib.CXG_E_CAPACITY:
                cap = int(got.value)
                continue
            _check(rc)
            n_rows = int(got.value)
            if n_rows * 2 >= cap:
                return out[:n_rows]                  # a view: copying 100+ MB would cost more than the scan
            return out[:n_rows].copy()

    # --- device-resident haystacks (bench, shards) ---
    def find_all_submatch_device(self, d_hay: int, length: int, d_out: int = 0, cap: int = 0, base: int = 0, n: int = -1,
                                 stream: int = 0, timing: Timing | None = None) -> int:
        got = C.c_uint64(0)
        rc = _lib.lib().cxg_find_all_submatch_device(self._h, d_hay, length, base, n, d_out or None, cap, C.byref(got),
                                                     stream or None, C.byref(timing) if timing is not None else None)
        _check(rc)
        return int(got.value)

    def find_all_device(self, d_hay: int, length: int, d_out: int = 0, cap: int = 0, base: int = 0, n: int = -1,
                        stream: int = 0, timing: Timing | None = None) -> int:
        got = C.c_uint64(0)
        rc = _lib.lib().cxg_find_all_device(self._h, d_hay, length, base, n, d_out or None, cap, C.byref(got),
                                            stream or None, C.byref(timing) if timing is not None else None)
        _check(rc)
        return int(got.value)


class Pending:
    """Handle of cxg_find_all_device_async; wait() on the thread that made the call returns the row count.  Also a context manager
    (`with rx.find_all_device_async(...) as p:` waits on exit) — and a handle that is dropped is waited for when it is collected, so
    that its scratch slot comes back.  (Since round 6 a pending call holds no lock: other threads' scans queue behind it on the device.)"""

    def __init__(self, h):
        self._h = h
        self.rows = None

    def wait(self, timing: "Timing | None" = None) -> int:
        if self._h is None:
            if self.rows is None:
                raise CoregexError(_lib.CXG_E_INVALID, "this asynchronous call was already waited for and failed")
            return self.rows
        got = C.c_uint64(0)
        h, self._h = self._h, None
        rc = _lib.lib().cxg_wait(h, C.byref(got), C.byref(timing) if timing is not None else None)
        if rc == _lib.CXG_E_THREAD:
            self._h = h                                               # wrong thread: the C side has not touched the handle (every other code consumes it)
        _check(rc)
        self.rows = int(got.value)
        return self.rows

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self._h is not None:
            try:
                self.wait()
            except CoregexError:
                if exc[0] is None:
                    raise
        return False

    def __del__(self):
        if getattr(self, "_h", None) is not None:
            try:
                self.wait()
            except Exception:                                         # noqa: BLE001 (collection on another thread, interpreter shutdown)
                pass


def _find_all_device_async(self, d_hay: int, length: int, d_out: int = 0, cap: int = 0, base: int = 0, n: int = -1, stream: int = 0) -> Pending:
    """cxg_find_all_device_async: the launch stays in flight; .wait() completes the call."""
    h = C.c_void_p()
    _check(_lib.lib().cxg_find_all_device_async(self._h, d_hay, length, base, n, d_out or None, cap, stream or None, C.byref(h)))
    return Pending(h)


Regex.find_all_device_async = _find_all_device_async


def _find_all_device_u32(self, d_hay: int, length: int, d_out: int = 0, cap: int = 0, n: int = -1, stream: int = 0,
                         timing: "Timing | None" = None) -> int:
    """cxg_find_all_device_u32: rows of two uint32 relative to d_hay (8 bytes per match); d_out == 0 counts."""
    got = C.c_uint64(0)
    rc = _lib.lib().cxg_find_all_device_u32(self._h, d_hay, length, n, d_out or None, cap, C.byref(got), stream or None,
                                            C.byref(timing) if timing is not None else None)
    _check(rc)
    return int(got.value)


Regex.find_all_device_u32 = _find_all_device_u32


def device_mem_info(device: int = 0):
    """(free, total) bytes of HBM on `device` (cxg_device_mem_info)."""
    f, t = C.c_uint64(0), C.c_uint64(0)
    _check(_lib.lib().cxg_device_mem_info(device, C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


def compile(pattern) -> Regex:  # noqa: A001  (mirrors coregex.Compile)
    p = pattern.encode() if isinstance(pattern, str) else bytes(pattern)
    h = C.c_void_p()
    _check(_lib.lib().cxg_compile(p, len(p), C.byref(h)))
    return Regex(h, p)


def must_compile(pattern) -> Regex:
    return compile(pattern)


# --- the constructors a cgo shim calls once per compiled *meta.Engine ------------------
def flatten_nfa(view: "_lib.Nfa"):
    """flattenNFA of the shim: copies an NFA description into freshly allocated C arrays (what Go does with C.malloc;
    the library must not depend on the source arrays staying alive).  Returns (Nfa, keepalive)."""
    states = (_lib.NfaState * max(1, view.n_states))()
    C.memmove(states, view.states, C.sizeof(_lib.NfaState) * view.n_states)
    trans = (_lib.NfaTrans * max(1, view.n_trans))()
    if view.n_trans:
        C.memmove(trans, view.trans, C.sizeof(_lib.NfaTrans) * view.n_trans)
    out = _lib.Nfa(C.cast(states, C.POINTER(_lib.NfaState)), view.n_states, C.cast(trans, C.POINTER(_lib.NfaTrans)),
                   view.n_trans, view.start_anchored, view.start_unanchored, view.capture_count)
    return out, (states, trans)


def program_from_nfa(nfa: "_lib.Nfa", strategy, flags: int = 0, pattern: bytes = b"") -> Regex:
    """cxg_program_from_nfa: e.nfa + e.strategy + {digitRunSkipSafe, reverseDFA != nil} -> device program."""
    st = STRATEGY_NAMES.index(strategy) if isinstance(strategy, str) else int(strategy)
    h = C.c_void_p()
    _check(_lib.lib().cxg_program_from_nfa(C.byref(nfa), st, flags, C.byref(h)))
    return Regex(h, pattern)


def program_from_literals(lits, pattern: bytes = b"") -> Regex:
    """cxg_program_from_literals: prefilter.Teddy patterns in pattern-ID order (UseTeddy)."""
    lits = [bytes(x) for x in lits]
    arr = (C.c_char_p * max(1, len(lits)))(*lits)
    lens = (C.c_uint32 * max(1, len(lits)))(*[len(x) for x in lits])
    h = C.c_void_p()
    _check(_lib.lib().cxg_program_from_literals(arr, lens, len(lits), C.byref(h)))
    return Regex(h, pattern)


def program_from_charclass(membership, min_match: int = 1, pattern: bytes = b"") -> Regex:
    """cxg_program_from_charclass: nfa.CharClassSearcher.membership[256] (UseCharClassSearcher)."""
    m = bytes(1 if x else 0 for x in membership)
    assert len(m) == 256
    h = C.c_void_p()
    _check(_lib.lib().cxg_program_from_charclass(m, min_match, C.byref(h)))
    return Regex(h, pattern)


class DeviceBuffer:
    """cxg_buffer: a haystack resident in HBM."""

    def __init__(self, length: int):
        h = C.c_void_p()
        _check(_lib.lib().cxg_buffer_alloc(length, C.byref(h)))
        self._h = h
        self.length = length

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and _lib._lib is not None:
            _lib._lib.cxg_buffer_free(self._h)
            self._h = None

    @property
    def ptr(self) -> int:
        return _lib.lib().cxg_buffer_device_ptr(self._h)

    def upload(self, data, off: int = 0):
        a = _host_view(data)
        _check(_lib.lib().cxg_buffer_upload(self._h, off, a.ctypes.data, a.size))

    def download(self, off: int, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.uint8)
        _check(_lib.lib().cxg_buffer_download(self._h, off, out.ctypes.data, n))
        return out

    def fill_synth(self, config: int, seed: int, first_page: int = 0):
        _check(_lib.lib().cxg_buffer_fill_synth(self._h, config, seed, first_page))


def synth_pages(config: int, seed: int, first_page: int, npages: int) -> np.ndarray:
    """CPU twin of the device corpus generator (synthlog-v1)."""
    out = np.empty(npages * 4096, dtype=np.uint8)
    L = _lib.lib()
    for i in range(npages):
        L.cxg_synth_page_host(config, seed, first_page + i, out.ctypes.data + i * 4096)
    return out
