#!/usr/bin/env python3
"""PCIe-inclusive rate: cxg_find_all on a HOST haystack (H2D copy + scan + D2H of the rows), what the cgo shim sees.
Also the small-haystack latency that sets the shim's threshold.  Usage: python scripts/time_host_path.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import coregex_amd as cx

rx = cx.compile(r"\d+\.\d+\.\d+\.\d+")
page = 4096
for mib in (0.0625, 0.25, 1, 4, 16, 64, 256, 1024):
    npages = int(mib * (1 << 20)) // page
    hay = cx.synth_pages(2, 0xC0FFEE02, 0, npages)          # pageable host memory (numpy)
    rows = rx.find_all_index(hay)                            # warm: scratch buffers sized
    reps = 3 if mib >= 256 else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        rows = rx.find_all_index(hay)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        n = rx.count(hay)
    dc = (time.perf_counter() - t0) / reps
    print("%8.2f MiB  find_all %9.3f ms  %7.2f GB/s   count %9.3f ms  %7.2f GB/s   rows %d" %
          (mib, dt * 1e3, hay.size / dt / 1e9, dc * 1e3, hay.size / dc / 1e9, len(rows)), flush=True)
