"""Literal-set kernels on small device-resident haystacks (GPU box): wall time per call and kernel time, pair kernel against wave kernel
(CXG_NO_TEDDY_PAIR=1 in a second process) — the pair kernel builds a 64 KiB table per workgroup before its first byte."""
import os, sys, time
os.environ.setdefault("CXG_PAIR_MIN_BYTES", "0")      # (by default the pair kernel serves haystacks from 320 MiB on)
import torch
import coregex_amd as cx
LITS16 = "error|warning|fatal|critical|panic|timeout|refused|denied|googlebot|bingbot|yandexbot|crawler|spider|failure|exception|overflow"
buf = cx.DeviceBuffer(64 << 20); buf.fill_synth(3, 0xC0FFEE03, 0)
rx = cx.compile(LITS16)
out = torch.zeros((400000, 2), dtype=torch.int64, device="cuda")
for n in (1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26):
    t = cx.Timing()
    for _ in range(5): rx.find_all_device(buf.ptr, n, out.data_ptr(), 400000, timing=t)
    reps = 200 if n <= (1 << 20) else 40
    km = 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        rx.find_all_device(buf.ptr, n, out.data_ptr(), 400000, timing=t); km += t.kernel_ms
    dt = (time.perf_counter() - t0) / reps
    print(f"n {n:9d} wall {dt * 1e6:8.1f} us kernel {km / reps * 1e3:8.1f} us  {cx._lib.lib().cxg_kernel_name(t.kernel).decode()} launches {t.n_launches}", flush=True)
