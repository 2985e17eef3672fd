import torch, coregex_amd as cx, numpy as np
rx = cx.compile(r"\d+\.\d+\.\d+\.\d+")
n = 1 << 30
buf = cx.DeviceBuffer(n); buf.fill_synth(2, 0xC0FFEE02, 0)
cnt = rx.find_all_device(buf.ptr, n)
out = torch.empty((cnt + 16, 2), dtype=torch.int64, device="cuda")
t = cx.Timing(); ks = []
for i in range(43):
    rx.find_all_device(buf.ptr, n, out.data_ptr(), cnt + 16, timing=t); ks.append(round(t.kernel_ms, 4))
print(ks)
print("mean", np.mean(ks[3:23]), "min", min(ks), "median", np.median(ks))
