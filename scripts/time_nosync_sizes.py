#!/usr/bin/env python3
"""Kernel time of the transducer kernel on `1.1.1...` by size (GPU box only): linear growth = serial cost per group,
quadratic = the look-back over maps walking back to the start."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coregex_amd as cx
pat = sys.argv[1] if len(sys.argv) > 1 else r"\d+\.\d+x?"
rx = cx.compile(pat)
for mib in ([int(a) for a in sys.argv[2:]] or (1, 2, 4, 8, 16, 32, 64, 128, 256)):
    n = mib << 20
    hay = np.frombuffer((b" x " + b"1." * (n // 2))[:n - 16] + b" y 1.2.3.4 abcx ", dtype=np.uint8)
    buf = cx.DeviceBuffer(n); buf.upload(hay)
    t = cx.Timing()
    cnt = rx.find_all_device(buf.ptr, n)
    out = torch.empty((cnt + 8, 2), dtype=torch.int64, device="cuda")
    best = 1e9
    for i in range(4):
        rx.find_all_device(buf.ptr, n, out.data_ptr(), cnt + 8, timing=t)
        best = min(best, t.kernel_ms)
    rx.find_all_device(buf.ptr, n, timing=t)
    print(f"{mib:4d} MiB rows {cnt:9d} kernel_ms {best:8.3f} count-only {t.kernel_ms:8.3f} launches {t.n_launches} reason {t.fallback_reason}", flush=True)
