import torch, coregex_amd as cx
rx = cx.compile(r"\d+\.\d+\.\d+\.\d+")
t = cx.Timing()
for mib in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
    n = mib << 20
    buf = cx.DeviceBuffer(n); buf.fill_synth(2, 0xC0FFEE02, 0)
    cnt = rx.find_all_device(buf.ptr, n)
    out = torch.empty((cnt + 8, 2), dtype=torch.int64, device="cuda")
    best = 1e9
    for i in range(6):
        rx.find_all_device(buf.ptr, n, out.data_ptr(), cnt + 8, timing=t)
        if i: best = min(best, t.kernel_ms)
    print(f"{mib:6d} MiB kernel_ms {best:8.4f}  ms/GiB {best * 1024 / mib:7.4f}", flush=True)
    del buf, out
