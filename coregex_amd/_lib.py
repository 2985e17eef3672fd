"""Loader for the C-ABI library ``libcoregex_hip.so`` (include/coregex_hip.h).

The library is the product: hand-written HIP kernels for gfx950 plus the host glue.  There is no
Python or CPU search path behind it — if the shared object is missing, or no MI355X is visible, every
search raises; nothing falls back to another engine.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CXG_LIB_PATH") or os.path.join(_HERE, "libcoregex_hip.so")   # CXG_LIB_PATH: A/B builds (scripts/)
_lib = None

CXG_OK, CXG_E_INVALID, CXG_E_UNSUPPORTED, CXG_E_CAPACITY = 0, -1, -2, -3
CXG_E_DEVICE, CXG_E_NO_GPU, CXG_E_SYNTAX, CXG_E_INTERNAL, CXG_E_INPUT, CXG_E_THREAD = -4, -5, -6, -7, -8, -9

# every symbol declared in include/coregex_hip.h (tests check the exports against the header)
SYMBOLS = [
    "cxg_last_error", "cxg_version", "cxg_device_count", "cxg_set_device", "cxg_thread_release", "cxg_compile", "cxg_program_flags",
    "cxg_program_from_nfa", "cxg_program_from_literals", "cxg_program_from_charclass",
    "cxg_program_destroy", "cxg_program_strategy", "cxg_strategy_name", "cxg_kernel_name", "cxg_program_num_groups",
    "cxg_program_nfa_states", "cxg_program_dfa_states", "cxg_program_supported", "cxg_program_nullable", "cxg_program_delimiters", "cxg_program_offset_captures", "cxg_program_blob",
    "cxg_program_nfa", "cxg_program_fsm_image", "cxg_program_submatch_blobs", "cxg_program_chain_captures", "cxg_program_chain_bounds", "cxg_program_submatch_supported", "cxg_find_all", "cxg_count", "cxg_find_all_submatch", "cxg_buffer_alloc",
    "cxg_buffer_free", "cxg_buffer_upload", "cxg_buffer_download", "cxg_buffer_len",
    "cxg_buffer_device_ptr", "cxg_buffer_fill_synth", "cxg_synth_page_host", "cxg_find_all_device", "cxg_find_all_device_u32",
    "cxg_find_all_submatch_device", "cxg_abi_version", "cxg_timing_size", "cxg_path_state", "cxg_debug_demote", "cxg_path_reset", "cxg_find_all_device_async", "cxg_wait",
    "cxg_device_mem_info", "cxg_find", "cxg_is_match", "cxg_find_device", "cxg_is_match_device",
]


class Timing(C.Structure):
    _fields_ = [("kernel_ms", C.c_float), ("total_ms", C.c_float), ("n_launches", C.c_uint32),
                ("grid", C.c_uint32), ("block", C.c_uint32), ("tiles", C.c_uint64), ("kernel", C.c_uint32),
                ("fallback_reason", C.c_uint32), ("n_ladder", C.c_uint32), ("ladder", C.c_uint8 * 12)]

    @property
    def kernels(self):
        """cxg_kernel id of every span launch of the call, in order (a fallback adds a rung)."""
        return [int(self.ladder[i]) for i in range(min(int(self.n_ladder), 12))]


class PathState(C.Structure):
    """cxg_path_state_t: launch-mode demotions of a device (watchdog hygiene, include/coregex_hip.h)."""
    _fields_ = [("static_penalty", C.c_uint32), ("static_hits", C.c_uint32), ("persistent_penalty", C.c_uint32), ("persistent_hits", C.c_uint32),
                ("delim_penalty", C.c_uint32), ("delim_hits", C.c_uint32), ("order_waiters", C.c_uint32), ("reserved", C.c_uint32)]


class NfaTrans(C.Structure):
    _fields_ = [("lo", C.c_uint8), ("hi", C.c_uint8), ("_pad", C.c_uint16), ("next", C.c_uint32)]


class NfaState(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("lo", C.c_uint8), ("hi", C.c_uint8), ("cap_start", C.c_uint8),
                ("next", C.c_uint32), ("left", C.c_uint32), ("right", C.c_uint32), ("cap_index", C.c_uint32),
                ("trans_off", C.c_uint32), ("trans_len", C.c_uint32)]


class Nfa(C.Structure):
    _fields_ = [("states", C.POINTER(NfaState)), ("n_states", C.c_uint32), ("trans", C.POINTER(NfaTrans)),
                ("n_trans", C.c_uint32), ("start_anchored", C.c_uint32), ("start_unanchored", C.c_uint32),
                ("capture_count", C.c_uint32)]


def build(force: bool = False) -> str:
    """Compile libcoregex_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force or not os.path.exists(LIB_PATH) or _stale(src):
        subprocess.check_call(["make", "-s", "-j8", "-C", src])
    return LIB_PATH


def _stale(src: str) -> bool:
    t = os.path.getmtime(LIB_PATH)
    for root, _, files in os.walk(src):
        for f in files:
            if f.endswith((".hip", ".cc", ".h", ".hpp")) and os.path.getmtime(os.path.join(root, f)) > t:
                return True
    return os.path.getmtime(os.path.join(os.path.dirname(_HERE), "include", "coregex_hip.h")) > t


def lib():
    """Load the library.  torch (if installed) is imported first so that both share one HIP runtime:
    PyTorch-ROCm bundles libamdhip64 with the soname this library needs."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950).  coregex_amd has no fallback engine.")
    try:
        import torch  # noqa: F401  (loads torch/lib/libamdhip64.so first)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, u64, i64, u32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_uint32
    L.cxg_last_error.restype = C.c_char_p
    L.cxg_version.restype = C.c_char_p
    L.cxg_strategy_name.restype = C.c_char_p
    L.cxg_strategy_name.argtypes = [C.c_int]
    L.cxg_kernel_name.restype = C.c_char_p
    L.cxg_kernel_name.argtypes = [C.c_int]
    L.cxg_set_device.argtypes = [C.c_int]
    L.cxg_compile.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(vp)]
    L.cxg_program_from_nfa.argtypes = [C.POINTER(Nfa), C.c_int, u32, C.POINTER(vp)]
    L.cxg_program_from_literals.argtypes = [C.POINTER(C.c_char_p), C.POINTER(u32), u32, C.POINTER(vp)]
    L.cxg_program_from_charclass.argtypes = [C.c_char_p, u32, C.POINTER(vp)]
    L.cxg_program_destroy.argtypes = [vp]
    L.cxg_program_destroy.restype = None
    for n in ("cxg_program_strategy", "cxg_program_num_groups", "cxg_program_nfa_states",
              "cxg_program_dfa_states", "cxg_program_supported", "cxg_program_nullable"):
        getattr(L, n).argtypes = [vp]
    L.cxg_program_delimiters.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.cxg_program_offset_captures.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.cxg_program_flags.argtypes = [vp]
    L.cxg_program_flags.restype = u32
    L.cxg_thread_release.argtypes = []
    L.cxg_thread_release.restype = None
    L.cxg_program_blob.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.cxg_program_nfa.argtypes = [vp, C.POINTER(Nfa)]
    L.cxg_program_fsm_image.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.cxg_program_submatch_blobs.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.cxg_program_chain_captures.argtypes = [vp, C.c_char_p]
    L.cxg_program_chain_captures.restype = C.c_int
    L.cxg_program_chain_bounds.argtypes = [vp, C.c_char_p]
    L.cxg_program_chain_bounds.restype = C.c_int
    L.cxg_program_submatch_supported.argtypes = [vp]
    L.cxg_find_all.argtypes = [vp, vp, u64, i64, vp, u64, C.POINTER(u64)]
    L.cxg_count.argtypes = [vp, vp, u64, i64, C.POINTER(u64)]
    L.cxg_find_all_submatch.argtypes = [vp, vp, u64, i64, vp, u64, C.POINTER(u64)]
    L.cxg_buffer_alloc.argtypes = [u64, C.POINTER(vp)]
    L.cxg_buffer_free.argtypes = [vp]
    L.cxg_buffer_free.restype = None
    L.cxg_buffer_upload.argtypes = [vp, u64, vp, u64]
    L.cxg_buffer_download.argtypes = [vp, u64, vp, u64]
    L.cxg_buffer_len.argtypes = [vp]
    L.cxg_buffer_len.restype = u64
    L.cxg_buffer_device_ptr.argtypes = [vp]
    L.cxg_buffer_device_ptr.restype = vp
    L.cxg_buffer_fill_synth.argtypes = [vp, u32, u64, u64]
    L.cxg_synth_page_host.argtypes = [u32, u64, u64, vp]
    L.cxg_find_all_device.argtypes = [vp, vp, u64, i64, i64, vp, u64, C.POINTER(u64), vp, C.POINTER(Timing)]
    L.cxg_find_all_device_u32.argtypes = [vp, vp, u64, i64, vp, u64, C.POINTER(u64), vp, C.POINTER(Timing)]
    L.cxg_find_all_submatch_device.argtypes = [vp, vp, u64, i64, i64, vp, u64, C.POINTER(u64), vp, C.POINTER(Timing)]
    L.cxg_timing_size.restype = C.c_size_t
    L.cxg_path_state.argtypes = [C.c_int, C.POINTER(PathState)]
    L.cxg_debug_demote.argtypes = [C.c_int, C.c_int]
    L.cxg_path_reset.argtypes = [C.c_int]
    L.cxg_find_all_device_async.argtypes = [vp, vp, u64, i64, i64, vp, u64, vp, C.POINTER(vp)]
    L.cxg_wait.argtypes = [vp, C.POINTER(u64), C.POINTER(Timing)]
    L.cxg_device_mem_info.argtypes = [C.c_int, C.POINTER(u64), C.POINTER(u64)]
    L.cxg_find.argtypes = [vp, vp, u64, C.POINTER(i64), C.POINTER(C.c_int)]
    L.cxg_is_match.argtypes = [vp, vp, u64, C.POINTER(C.c_int)]
    L.cxg_find_device.argtypes = [vp, vp, u64, i64, C.POINTER(i64), C.POINTER(C.c_int), vp]
    L.cxg_is_match_device.argtypes = [vp, vp, u64, C.POINTER(C.c_int), vp]
    if L.cxg_abi_version() != 3 or L.cxg_timing_size() != C.sizeof(Timing):
        raise RuntimeError(f"{LIB_PATH}: ABI {L.cxg_abi_version()} / cxg_timing of {L.cxg_timing_size()} bytes, this binding expects 3 / {C.sizeof(Timing)}")
    _lib = L
    return L
