#!/usr/bin/env python3
"""Headline kernel (IP regex, 1 GiB synthlog config 2 in HBM): count-only and row-writing calls, best and mean of N launches, plus an
order-sensitive checksum of the rows (equal across library variants = same rows).  GPU box only; CXG_LIB_PATH selects an A/B build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coregex_amd as cx
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n = int(gib * (1 << 30)) // 4096 * 4096
buf = cx.DeviceBuffer(n); buf.fill_synth(2, 0xC0FFEE02, 0)
rx = cx.compile(r"\d+\.\d+\.\d+\.\d+"); t = cx.Timing()
cnt = rx.find_all_device(buf.ptr, n)
out = (torch.zeros if os.environ.get("TH_ZEROS") else torch.empty)((cnt + 8, 2), dtype=torch.int64, device="cuda")   # TH_ZEROS=1: a fill kernel of torch's stream runs beside the first scan
for _ in range(30):
    rx.find_all_device(buf.ptr, n, out.data_ptr(), out.shape[0])
tc, tr = [], []
for i in range(reps):
    rx.find_all_device(buf.ptr, n, timing=t); tc.append(t.kernel_ms)
for i in range(reps):
    rx.find_all_device(buf.ptr, n, out.data_ptr(), out.shape[0], timing=t); tr.append(t.kernel_ms)
k = torch.arange(1, cnt + 1, dtype=torch.int64, device="cuda")
ck = (int((out[:cnt, 0] * k).sum().item()) + 3 * int((out[:cnt, 1] * k).sum().item())) & ((1 << 64) - 1)
alg = n + 16 * cnt
name = os.path.basename(os.environ.get("CXG_LIB_PATH", "product")).replace("libcoregex_hip_", "").replace(".so", "")
st = cx.path_state(0)
print(f"{name:10s} hits(static/pers)={st['static_hits']}/{st['persistent_hits']} occ={os.environ.get('CXG_PF_OCC', '-')} {gib:g} GiB rows {cnt} count-only best {min(tc):.4f} mean {sum(tc)/len(tc):.4f} | rows best {min(tr):.4f} mean {sum(tr)/len(tr):.4f} ms"
      f" -> frac best {alg / min(tr) / 8e9:.4f} mean {alg / (sum(tr)/len(tr)) / 8e9:.4f} | {cx._lib.lib().cxg_kernel_name(t.kernel).decode()} x{t.n_launches} ck {ck:016x}", flush=True)
