#!/usr/bin/env python3
"""Input without synchronising structure (`1.1.1.1...` for the IPv4 pattern): rows vs the oracle (small sizes), kernel time
and the kernel that ran, up to 64 MiB.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coregex_amd as cx
from oracle import oracle as O
README_IP = r"(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)"
cases = [(r"\d+\.\d+\.\d+\.\d+", b"1."), (README_IP, b"1."), (r"\d+:\d+:\d+", b"12:"), (r"\d+\.\d+x?", b"1."),
         (r"a+b|b+a", b"a"), (r"[a-c]x|[b-d]y", b"b"), (r"error", b"error"), (r"error|warning|fatal|critical", b"error")]
for pat, unit in cases:
    rx = cx.compile(pat)
    o = O.Regex(pat)
    for n in (1 << 20, 64 << 20):
        hay = np.frombuffer((b" x " + unit * (n // len(unit)))[:n - 16] + b" y 1.2.3.4 abcx ", dtype=np.uint8)
        buf = cx.DeviceBuffer(n); buf.upload(hay)
        t = cx.Timing()
        try:
            cnt = rx.find_all_device(buf.ptr, n)
            out = torch.empty((cnt + 8, 2), dtype=torch.int64, device="cuda")
            t0 = time.time(); cnt2 = rx.find_all_device(buf.ptr, n, out.data_ptr(), cnt + 8, timing=t); dt = time.time() - t0
        except cx.CoregexError as e:
            print(f"{pat[:24]:24s} n={n>>20:3d} MiB  ERROR {e}"); continue
        ok = ""
        if n <= (1 << 20):
            exp = o.find_all_index(hay)
            ok = "rows==oracle" if np.array_equal(out[:cnt].cpu().numpy(), exp) else f"MISMATCH got {cnt} exp {len(exp)}"
        else:
            rows = out[:cnt]
            ok = "sorted,disjoint" if bool((rows[1:, 0] >= rows[:-1, 1]).all()) and bool((rows[:, 1] > rows[:, 0]).all()) else "BROKEN ORDER"
        print(f"{pat[:24]:24s} n={n>>20:3d} MiB rows {cnt:9d} kernel_ms {t.kernel_ms:9.3f} launches {t.n_launches} {cx._lib.lib().cxg_kernel_name(t.kernel).decode():18s} reason {t.fallback_reason} wall {dt*1e3:8.1f} ms {ok}", flush=True)
