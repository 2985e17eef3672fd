#!/usr/bin/env python3
"""Kernel time of one pattern on 1 GiB of synthlog config 2: count-only call and row-writing call (best of 5 each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coregex_amd as cx
pat = sys.argv[1] if len(sys.argv) > 1 else r"\d+\.\d+\.\d+\.\d+"
n = 1 << 30
buf = cx.DeviceBuffer(n); buf.fill_synth(2, 0xC0FFEE02, 0)
rx = cx.compile(pat); t = cx.Timing()
cnt = rx.find_all_device(buf.ptr, n)
out = torch.empty((max(cnt, 10_000_000) + 8, 2), dtype=torch.int64, device="cuda")
best_c = best_r = 1e9
for i in range(6):
    rx.find_all_device(buf.ptr, n, timing=t)
    if i: best_c = min(best_c, t.kernel_ms)
for i in range(6):
    rx.find_all_device(buf.ptr, n, out.data_ptr(), out.shape[0], timing=t)
    if i: best_r = min(best_r, t.kernel_ms)
print(f"{os.environ.get('CXG_LIB_PATH', 'product').split('_')[-1]:12s} matches {cnt:9d} count-only {best_c:.4f} ms  rows {best_r:.4f} ms  kernel {cx._lib.lib().cxg_kernel_name(t.kernel).decode()} launches {t.n_launches}", flush=True)
