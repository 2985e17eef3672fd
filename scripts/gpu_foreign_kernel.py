#!/usr/bin/env python3
"""What a foreign kernel beside the persistent scan does (round 5): thread A scans 1 GiB with the headline program in a loop, thread B keeps
torch fill kernels (default stream) or synthlog fills (null stream) running on the same device.  Prints per-call wall times of A (median, p99,
max), watchdog hits and the kernels A's calls ran on.  GPU box only."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coregex_amd as cx

mode = sys.argv[1] if len(sys.argv) > 1 else "torchfill"
# argv[2] == "literals" (round 6): BASELINE config 3's literal set on config 3's corpus — the pair kernel (scan_teddy_pair.hip), whose
# groups are claimed: it needs neither co-residency nor dispatch in index order
lits = len(sys.argv) > 2 and sys.argv[2] == "literals"
n = 1 << 30
buf = cx.DeviceBuffer(n); buf.fill_synth(3 if lits else 2, 0xC0FFEE03 if lits else 0xC0FFEE02, 0)
rx = cx.compile("error|warning|fatal|critical|panic|timeout|refused|denied|googlebot|bingbot|yandexbot|crawler|spider|failure|exception|overflow" if lits else r"\d+\.\d+\.\d+\.\d+")
cnt = rx.find_all_device(buf.ptr, n)
out = torch.empty((cnt + 8, 2), dtype=torch.int64, device="cuda")
for _ in range(20):
    rx.find_all_device(buf.ptr, n, out.data_ptr(), cnt + 8)
stop = False


def foreign():
    cx.set_device(0)
    if mode == "torchfill":
        x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        i = 0
        while not stop:
            x.fill_(i & 255); i += 1
            if i % 8 == 0:
                torch.cuda.synchronize()
    elif mode == "synthfill":
        b2 = cx.DeviceBuffer(256 << 20)
        i = 0
        while not stop:
            b2.fill_synth(2, 0xC0FFEE02, i); i += 1
    elif mode == "none":
        while not stop:
            time.sleep(0.01)


th = threading.Thread(target=foreign); th.start()
time.sleep(0.2)
times, kernels = [], {}
t = cx.Timing()
for i in range(200):
    t0 = time.perf_counter()
    assert rx.find_all_device(buf.ptr, n, out.data_ptr(), cnt + 8, timing=t) == cnt
    times.append((time.perf_counter() - t0) * 1e3)
    kernels[int(t.kernel)] = kernels.get(int(t.kernel), 0) + 1
stop = True; th.join()
times.sort()
st = cx.path_state(0)
print(f"{mode:10s} lib={os.path.basename(os.environ.get('CXG_LIB_PATH', 'product'))} calls 200: wall ms median {times[100]:.3f} p90 {times[180]:.3f} p99 {times[198]:.3f} max {times[-1]:.3f} | kernels {kernels} | "
      f"hits static/pers {st['static_hits']}/{st['persistent_hits']} penalty {st['static_penalty']}/{st['persistent_penalty']}", flush=True)
