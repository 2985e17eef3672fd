#!/usr/bin/env python3
"""Kernel time of assorted chain-shaped patterns on 1 GiB of synthlog config 2 (device-resident).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coregex_amd as cx
n = 1 << 30
buf = cx.DeviceBuffer(n)
buf.fill_synth(2, 0xC0FFEE02, 0)
pats = sys.argv[1:] or [r"\d+\.\d+\.\d+\.\d+", r"error", r"\d{4}-\d{2}-\d{2}", r"\d{2}:\d{2}:\d{2}", r"\d+:\d+:\d+", r"GET|POST|PUT", r"[0-9a-f]{2}(:[0-9a-f]{2}){5}",
                        r"[A-Z][a-z]+", r"HTTP/\d\.\d", r"(\d+)\.(\d+)\.(\d+)\.(\d+)",
                        r"(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)",
                        r"(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]", r"\d+\.\d+x?", r"a+b|b+a", r"GET|POST /[a-z]+", r"\d+(\.\d+)?%"]
for pat in pats:
    rx = cx.compile(pat)
    sub = "(" in pat and rx.submatch_supported and rx.num_groups > 1
    if not (rx.supported or sub):
        print(f"{pat:40s} {rx.strategy:20s} unsupported: {rx.why_unsupported}")
        continue
    t = cx.Timing()
    w = 2 * rx.num_groups if sub else 2
    f0 = (lambda: rx.find_all_submatch_device(buf.ptr, n)) if sub else (lambda: rx.find_all_device(buf.ptr, n))
    cnt = f0()
    out = torch.empty((cnt + 8, w), dtype=torch.int64, device="cuda")
    f = (lambda: rx.find_all_submatch_device(buf.ptr, n, out.data_ptr(), cnt + 8, timing=t)) if sub else (lambda: rx.find_all_device(buf.ptr, n, out.data_ptr(), cnt + 8, timing=t))
    best = 1e9
    for i in range(6):
        f()
        if i:
            best = min(best, t.kernel_ms)
    print(f"{pat:40s} {rx.strategy:20s} matches {cnt:10d} kernel_ms {best:8.4f} launches {t.n_launches} {cx._lib.lib().cxg_kernel_name(t.kernel).decode():22s} {n / best / 1e6:8.1f} GB/s", flush=True)
    del out
