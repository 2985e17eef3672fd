"""Literal-set kernels on 64 MiB .. 1 GiB of synthlog-v1 config 3 (GPU box): kernel time by haystack length (CXG_PAIR_TAPER: the pair kernel's
small-group stretch; CXG_NO_TEDDY_PAIR=1: the wave kernel)."""
import sys, os
os.environ.setdefault("CXG_PAIR_MIN_BYTES", "0")
import torch
import coregex_amd as cx
LITS16 = "error|warning|fatal|critical|panic|timeout|refused|denied|googlebot|bingbot|yandexbot|crawler|spider|failure|exception|overflow"
buf = cx.DeviceBuffer(1 << 30); buf.fill_synth(3, 0xC0FFEE03, 0)
rx = cx.compile(LITS16)
out = torch.zeros((4500000, 2), dtype=torch.int64, device="cuda")
res = []
for n in (64 << 20, 128 << 20, 256 << 20, 512 << 20, 1 << 30):
    t = cx.Timing(); best = 1e9
    for i in range(8):
        rx.find_all_device(buf.ptr, n, out.data_ptr(), 4500000, timing=t)
        if i: best = min(best, t.kernel_ms)
    res.append(f"{n >> 20} MiB {best * 1e3:.1f}")
print(os.environ.get("CXG_PAIR_TAPER", "default"), "| us:", " | ".join(res), flush=True)
