"""Literal-set kernels on 1 GiB of synthlog-v1 config 3 resident in HBM (GPU box only): kernel time, row count and an
order-sensitive checksum of the rows.  Run once as is (scan_teddy_pair.hip) and once with CXG_NO_TEDDY_PAIR=1
(scan_teddy_wave.hip): the two checksum lines must be equal.  argv: [pattern-name ...] from PATS."""
import sys
import torch
import coregex_amd as cx

LITS16 = "error|warning|fatal|critical|panic|timeout|refused|denied|googlebot|bingbot|yandexbot|crawler|spider|failure|exception|overflow"
PATS = {
    "cfg3": LITS16,
    "four": "spider|error|crawler|denied",
    "fold": "(?i)(error|fail|exception|panic|fatal)",
    "word": r"\b(error|warning|fatal)\b",
    "short": "GET|POST|HTTP|html",
    "fat": "|".join(["error", "warning", "fatal", "critical", "panic", "timeout", "refused", "denied", "googlebot", "bingbot", "yandexbot", "crawler", "spider", "failure", "exception", "overflow"] +
                    ["metrics", "session", "upload", "export", "assets", "index", "worker", "search", "item%02d" % 0] + ["zq%02dxx" % i for i in range(15)]),
}
names = sys.argv[1:] or ["cfg3"]
n = 1 << 30
buf = cx.DeviceBuffer(n)
buf.fill_synth(3, 0xC0FFEE03, 0)
for name in names:
    rx = cx.compile(PATS[name])
    t = cx.Timing()
    cnt = rx.find_all_device(buf.ptr, n)
    out = torch.zeros((cnt + 8, 2), dtype=torch.int64, device="cuda")
    best = 1e9
    for i in range(8):
        got = rx.find_all_device(buf.ptr, n, out.data_ptr(), cnt + 8, timing=t)
        assert got == cnt
        if i:
            best = min(best, t.kernel_ms)
    idx = torch.arange(1, cnt + 1, dtype=torch.int64, device="cuda")
    ck = int(((out[:cnt, 0] * 1000003 + out[:cnt, 1] * 7919) * idx).sum().item()) & 0xFFFFFFFFFFFFFFFF
    srt = bool((out[1:cnt, 0] >= out[:cnt - 1, 1]).all().item())
    kn = cx._lib.lib().cxg_kernel_name(t.kernel).decode()
    print(f"{name:6s} {rx.strategy:10s} rows {cnt:9d} checksum {ck:016x} ordered {srt} | {kn} launches {t.n_launches} kernel_ms {best:.4f} "
          f"frac {(n + 16 * cnt) / best / 1e6 / 8000:.4f}", flush=True)
    del out
