#!/usr/bin/env python3
"""cxg_find_all on a HOST haystack through the C ABI with a caller-owned, already touched row array (what a Go caller that reuses
its [][2]int does): separates the seam's own cost from first-touch page faults of a fresh array (scripts/time_host_path.py)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import coregex_amd as cx
L = cx._lib.lib()
rx = cx.compile(r"\d+\.\d+\.\d+\.\d+")
for mib in (64, 256, 1024, 4096):
    npages = (mib << 20) // 4096
    hay = cx.synth_pages(2, 0xC0FFEE02, 0, npages)
    cap = hay.size // 64 + 16
    out = np.zeros((cap, 2), dtype=np.int64)                 # touched once
    got = C.c_uint64(0)
    assert L.cxg_find_all(rx._h, hay.ctypes.data, hay.size, -1, out.ctypes.data, cap, C.byref(got)) == 0
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        L.cxg_find_all(rx._h, hay.ctypes.data, hay.size, -1, out.ctypes.data, cap, C.byref(got))
    dt = (time.perf_counter() - t0) / reps
    print("%8.0f MiB  cxg_find_all (reused row array) %9.3f ms  %7.2f GB/s   rows %d" % (mib, dt * 1e3, hay.size / dt / 1e9, got.value), flush=True)
