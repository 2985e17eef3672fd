"""ctypes front for tests/emu/libcxg_emu.so (host emulation of the device lane walks; tests only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "emu", "libcxg_emu.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "emu")])
        L = C.CDLL(_PATH)
        L.emu_find_all.restype = C.c_int64
        L.emu_find_all.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_int64, C.c_int]
        L.emu_find_all_chain.restype = C.c_int64
        L.emu_find_all_chain.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.emu_find_all_chain6.restype = C.c_int64
        L.emu_find_all_chain6.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.emu_find_all_chain6_bounded.restype = C.c_int64
        L.emu_find_all_chain6_bounded.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_int, C.c_int]
        for name in ("emu_find_all_teddy_wave", "emu_find_all_charclass_wave"):
            f = getattr(L, name)
            f.restype = C.c_int64
            f.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.emu_find_all_fields.restype = C.c_int64
        L.emu_find_all_fields.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_int]
        L.emu_find_all_fields2.restype = C.c_int64
        L.emu_find_all_fields2.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.emu_find_all_literal.restype = C.c_int64
        L.emu_find_all_literal.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.emu_literal_shape.restype = C.c_int
        L.emu_literal_shape.argtypes = [C.c_char_p]
        L.emu_fields_shape.restype = C.c_int
        L.emu_fields_shape.argtypes = [C.c_char_p]
        L.emu_find_all_trio.restype = C.c_int64
        L.emu_find_all_trio.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_int]
        L.emu_trio_shape.restype = C.c_int
        L.emu_trio_shape.argtypes = [C.c_char_p]
        L.emu_fsm_maps_check.restype = C.c_int64
        L.emu_fsm_maps_check.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
        L.emu_find_all_fsm_direct.restype = C.c_int64
        L.emu_find_all_fsm_direct.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.emu_find_all_fsm.restype = C.c_int64
        L.emu_find_all_fsm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.emu_find_all_submatch.restype = C.c_int64
        L.emu_find_all_submatch.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_int64]
        L.emu_captures_bt.restype = C.c_int64
        L.emu_captures_bt.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p]
        _lib = L
    return _lib


def find_all(blob: bytes, hay, chunk: int = 64, flat: int = 0) -> np.ndarray:
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else np.ascontiguousarray(hay)
    # pad so the emulator's dword reads near the end stay inside the allocation
    padded = np.concatenate([a, np.zeros(8, dtype=np.uint8)])
    cap = 1 << 12
    while True:
        out = np.empty(cap, dtype=np.int64)
        n = lib().emu_find_all(blob, padded.ctypes.data, a.size, chunk, out.ctypes.data, cap, int(flat))
        assert n >= 0, f"emulator error {n}"
        if n <= cap:
            return out[:n].reshape(-1, 2).copy()
        cap = int(n)


def find_all_submatch(span_blob: bytes, cap_blob: bytes, hay, width: int, chunk: int = 64) -> np.ndarray:
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else np.ascontiguousarray(hay)
    padded = np.concatenate([a, np.zeros(8, dtype=np.uint8)])
    cap = 1 << 12
    while True:
        out = np.empty(cap, dtype=np.int64)
        n = lib().emu_find_all_submatch(span_blob, cap_blob, padded.ctypes.data, a.size, chunk, out.ctypes.data, cap)
        assert n >= 0, f"emulator error {n}"
        if n <= cap:
            return out[:n].reshape(-1, width).copy()
        cap = int(n)


def captures_bt(cap_blob: bytes, hay, spans: np.ndarray, width: int) -> np.ndarray:
    """Rows of FindAllSubmatchIndex from given spans: the backtracking capture pass (both tiers), as capi_captures.hip runs it."""
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else np.ascontiguousarray(hay)
    padded = np.concatenate([a, np.zeros(8, dtype=np.uint8)])
    sp = np.ascontiguousarray(spans, dtype=np.int64).reshape(-1, 2)
    out = np.empty((len(sp), width), dtype=np.int64)
    n = lib().emu_captures_bt(cap_blob, padded.ctypes.data, a.size, sp.ctypes.data, len(sp), out.ctypes.data)
    assert n == len(sp) * width, f"emulator error {n}"
    return out


def find_all_chain(blob: bytes, hay, tile: int = 16384, halo: int = 256):
    """Fourth-generation digit kernel, emulated.  Returns None when the tile-level fallback flag would be raised."""
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else np.ascontiguousarray(hay)
    padded = np.concatenate([np.zeros(8, dtype=np.uint8), a, np.zeros(8, dtype=np.uint8)])
    cap = 1 << 12
    while True:
        out = np.empty(cap, dtype=np.int64)
        n = lib().emu_find_all_chain(blob, padded.ctypes.data + 8, a.size, out.ctypes.data, cap, tile, halo)
        if n == -5:
            return None
        assert n >= 0, f"emulator error {n}"
        if n <= cap:
            return out[:n].reshape(-1, 2).copy()
        cap = int(n)


def find_all_chain6(blob: bytes, hay, tile: int = 3840, halo: int = 256):
    """Sixth-generation (bit-parallel) chain kernel, emulated.  Returns the int reason (< 0) when a tile would
    raise the fallback flag."""
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else np.ascontiguousarray(hay)
    padded = np.concatenate([np.zeros(8, dtype=np.uint8), a, np.zeros(8, dtype=np.uint8)])
    cap = 1 << 12
    while True:
        out = np.empty(cap, dtype=np.int64)
        n = lib().emu_find_all_chain6(blob, padded.ctypes.data + 8, a.size, out.ctypes.data, cap, tile, halo)
        if n <= -16:
            return int(n)
        assert n >= 0, f"emulator error {n}"
        if n <= cap:
            return out[:n].reshape(-1, 2).copy()
        cap = int(n)


def find_all_chain6_bounded(blob: bytes, bounds40: bytes, hay, tile: int = 3840, halo: int = 256):
    """Bounded-repetition mode of the chain kernel (BND instantiations), emulated: the surrogate chain of `blob`, rows
    filtered by the field bounds.  Returns the int reason (< 0) when a tile would raise the fallback flag."""
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else np.ascontiguousarray(hay)
    padded = np.concatenate([np.zeros(8, dtype=np.uint8), a, np.zeros(8, dtype=np.uint8)])
    cap = 1 << 12
    while True:
        out = np.empty(cap, dtype=np.int64)
        n = lib().emu_find_all_chain6_bounded(blob, bounds40, padded.ctypes.data + 8, a.size, out.ctypes.data, cap, tile, halo)
        if n <= -16:
            return int(n)
        assert n >= 0, f"emulator error {n}"
        if n <= cap:
            return out[:n].reshape(-1, 2).copy()
        cap = int(n)


def _wave_twin(name: str, blob: bytes, hay, tile: int, halo: int):
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else np.ascontiguousarray(hay)
    padded = np.concatenate([np.zeros(8, dtype=np.uint8), a, np.zeros(8, dtype=np.uint8)])
    cap = 1 << 12
    while True:
        out = np.empty(cap, dtype=np.int64)
        n = getattr(lib(), name)(blob, padded.ctypes.data + 8, a.size, out.ctypes.data, cap, tile, halo)
        if n <= -16:
            return int(n)
        assert n >= 0, f"emulator error {n}"
        if n <= cap:
            return out[:n].reshape(-1, 2).copy()
        cap = int(n)


def find_all_teddy_wave(blob: bytes, hay, tile: int = 3840, halo: int = 256):
    """scan_teddy_wave.hip, emulated; the int reason (< 0) when a tile would raise the fallback flag."""
    return _wave_twin("emu_find_all_teddy_wave", blob, hay, tile, halo)


def find_all_charclass_wave(blob: bytes, hay, tile: int = 3840, halo: int = 256):
    """scan_charclass_wave.hip, emulated; the int reason (< 0) when a tile would raise the fallback flag."""
    return _wave_twin("emu_find_all_charclass_wave", blob, hay, tile, halo)


def find_all_fsm_direct(image: bytes, hay, tile: int = 3840, budget: int = 192, stats=None):
    """scan_fsm.hip k_scan_fsmd (round 6: the transducer through byte-indexed rows), emulated; None when the image has no direct
    section, the int reason (< 0) when a tile would raise the fallback flag."""
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else np.ascontiguousarray(hay)
    padded = np.concatenate([np.zeros(8, dtype=np.uint8), a, np.zeros(8, dtype=np.uint8)])
    cap = 1 << 12
    st = np.zeros(4, dtype=np.uint64)
    while True:
        out = np.empty(cap, dtype=np.int64)
        n = lib().emu_find_all_fsm_direct(image, padded.ctypes.data + 8, a.size, out.ctypes.data, cap, tile, budget, st.ctypes.data)
        if n == -1:
            return None
        if n <= -16:
            return int(n)
        assert n >= 0, f"emulator error {n}"
        if n <= cap:
            if stats is not None:
                stats[:] = st
            return out[:n].reshape(-1, 2).copy()
        cap = int(n)


def find_all_fsm(image: bytes, hay, tile: int = 3840, chunk: int = 32, budget: int = 192, stats=None, dense: int = 0):
    """scan_fsm.hip (FindAll transducer), emulated; the int reason (< 0) when a tile would raise the fallback flag."""
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else np.ascontiguousarray(hay)
    padded = np.concatenate([np.zeros(8, dtype=np.uint8), a, np.zeros(8, dtype=np.uint8)])
    cap = 1 << 12
    st = np.zeros(4, dtype=np.uint64)
    while True:
        out = np.empty(cap, dtype=np.int64)
        n = lib().emu_find_all_fsm(image, padded.ctypes.data + 8, a.size, out.ctypes.data, cap, tile, chunk, budget, st.ctypes.data, dense)
        if n <= -16:
            return int(n)
        assert n >= 0, f"emulator error {n}"
        if n <= cap:
            if stats is not None:
                stats[:] = st
            return out[:n].reshape(-1, 2).copy()
        cap = int(n)


def fields_shape(blob: bytes) -> int:
    """Number of fields when scan_fields_wave.hip serves the program (run(F) (byte(S) run(F))*, two disjoint classes), else 0."""
    return int(lib().emu_fields_shape(blob))


def find_all_fields(blob: bytes, hay, own_words: int = 60, pre_words: int = 1):
    """Sequential twin of scan_fields_wave.hip.  Returns None where a tile would raise the fallback flag.  pre_words = 1: the grouped
    kernel's window (64 bytes in front of the tile); 2: the persistent kernel's (128 in front, 128 behind at own_words = 60)."""
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else np.ascontiguousarray(hay)
    padded = np.concatenate([a, np.zeros(8, dtype=np.uint8)])
    cap = 1 << 12
    while True:
        out = np.empty(cap, dtype=np.int64)
        n = lib().emu_find_all_fields2(blob, padded.ctypes.data, a.size, out.ctypes.data, cap, int(own_words), int(pre_words))
        if n <= -16:
            return None
        assert n >= 0, f"emulator error {n}"
        if n <= cap:
            return out[:n].reshape(-1, 2).copy()
        cap = int(n)


def literal_shape(blob: bytes) -> int:
    """Number of distinct bytes (2..4) when the persistent kernel's literal mode serves the program (a border-free literal), else 0."""
    return int(lib().emu_literal_shape(blob))


def find_all_literal(blob: bytes, hay, own_words: int = 60, pre_words: int = 2):
    """Sequential twin of k_scan_fields_pers's literal mode (lit_core)."""
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else np.ascontiguousarray(hay)
    padded = np.concatenate([a, np.zeros(8, dtype=np.uint8)])
    cap = 1 << 12
    while True:
        out = np.empty(cap, dtype=np.int64)
        n = lib().emu_find_all_literal(blob, padded.ctypes.data, a.size, out.ctypes.data, cap, int(own_words), int(pre_words))
        assert n >= 0, f"emulator error {n}"
        if n <= cap:
            return out[:n].reshape(-1, 2).copy()
        cap = int(n)


def trio_shape(blob: bytes) -> int:
    """Number of fields K (2..4) when k_scan_trio_wave serves the program — run(F) (byte(c_i) run(F)){K-1} — else 0; | 8 when
    every link has the same separator (K >= 3)."""
    return int(lib().emu_trio_shape(blob))


def find_all_trio(blob: bytes, hay, own_words: int = 60):
    """Sequential twin of k_scan_trio_wave: rows (start, link 1 .. link K-1, end).  None where a tile would raise the fallback flag."""
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else np.ascontiguousarray(hay)
    padded = np.concatenate([a, np.zeros(8, dtype=np.uint8)])
    cap = 1 << 12
    while True:
        out = np.empty(cap, dtype=np.int64)
        n = lib().emu_find_all_trio(blob, padded.ctypes.data, a.size, out.ctypes.data, cap, int(own_words))
        if n <= -16:
            return None
        assert n >= 0, f"emulator error {n}"
        if n <= cap:
            return out[:n].reshape(-1, (trio_shape(blob) & 7) + 1).copy()
        cap = int(n)


def fsm_maps_check(image: bytes, hay, tile: int = 3840, tiles_per_group: int = 32) -> int:
    """Round 3 (scan_fsm.hip "Maps instead of waits"): the kernel's compositions of sub-chunk / tile / group maps against a plain
    left-to-right walk.  Returns the number of sub-chunk entries checked (>= 0), -17 where the kernel would raise its fallback flag
    (a set that is not listed), -100 - k on a mismatch of kind k."""
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else np.ascontiguousarray(hay)
    padded = np.concatenate([np.zeros(8, dtype=np.uint8), a, np.zeros(8, dtype=np.uint8)])
    return int(lib().emu_fsm_maps_check(image, padded.ctypes.data + 8, a.size, int(tile), int(tiles_per_group)))


def merge_empty_matches(rows: np.ndarray, n: int) -> np.ndarray:
    """FindAll of a NULLABLE pattern from the rows of its non-empty variant over a haystack of n bytes (meta/findall.go:216-283;
    what capi_nullable.hip scanNullable computes on the device): an empty match at every position 0..n outside the closed intervals [s, e]."""
    covered = np.zeros(n + 2, dtype=bool)
    for s, e in np.asarray(rows).reshape(-1, 2).tolist():
        covered[s:e + 1] = True
    out = [(int(s), int(e)) for s, e in np.asarray(rows).reshape(-1, 2).tolist()] + [(p, p) for p in range(n + 1) if not covered[p]]
    out.sort()
    return np.array(out, dtype=np.int64).reshape(-1, 2)


def find_all_delim(open_byte: int, close_byte: int, plus: bool, hay, tile: int = 3840):
    """scan_delim_wave.hip (`O [^E]+ E` / `O [^E]* E`), emulated; the int reason (< 0) when a tile would raise the fallback flag."""
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else np.ascontiguousarray(hay)
    L = lib()
    L.emu_find_all_delim.restype = C.c_int64
    L.emu_find_all_delim.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_int]
    cap = 1 << 12
    while True:
        out = np.empty(cap, dtype=np.int64)
        n = L.emu_find_all_delim(open_byte, close_byte, int(plus), a.ctypes.data if a.size else None, a.size, out.ctypes.data, cap, tile)
        if n < 0:
            return int(n)
        if n <= cap:
            return out[:n].reshape(-1, 2).copy()
        cap = int(n)
