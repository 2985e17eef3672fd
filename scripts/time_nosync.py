#!/usr/bin/env python3
"""How slow is the serial walk over input without synchronising bytes?  One lane owns the whole stretch.
Usage: python scripts/time_nosync.py [max_mib]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import coregex_amd as cx

max_mib = float(sys.argv[1]) if len(sys.argv) > 1 else 4
cases = [(r"[a-c]+x", b"abc"), (r"\d+\.\d+\.\d+\.\d+", b"1"), (r"\d+\.\d+\.\d+\.\d+", b"1."), (r"[\w]+", b"a"), (r"error|warning|fatal|critical", b"error"),
         (r"(\w+)@(\w+)\.(\w+)", b"ab")]
for pat, unit in cases:
    rx = cx.compile(pat)
    mib = 0.25
    while mib <= max_mib:
        n = int(mib * (1 << 20))
        hay = np.frombuffer((unit * (n // len(unit) + 1))[:n], dtype=np.uint8)
        t0 = time.time()
        try:
            if "(" in pat:
                rows = len(rx.find_all_submatch_index(hay))
            else:
                rows = len(rx.find_all_index(hay))
            err = ""
        except cx.CoregexError as ex:
            rows, err = -1, str(ex)
        dt = time.time() - t0
        print("%-32s %-8s %6.2f MiB rows %9d  %8.1f ms  %s" % (pat, unit.decode(), mib, rows, dt * 1e3, err), flush=True)
        if dt > 5:
            break
        mib *= 4
