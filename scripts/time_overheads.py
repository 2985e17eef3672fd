import time, torch, coregex_amd as cx
rx = cx.compile(r"\d+\.\d+\.\d+\.\d+")
n = 1 << 30
buf = cx.DeviceBuffer(n); buf.fill_synth(2, 0xC0FFEE02, 0)
cnt = rx.find_all_device(buf.ptr, n)
out = torch.empty((cnt + 8, 2), dtype=torch.int64, device="cuda")
t = cx.Timing()
for _ in range(3): rx.find_all_device(buf.ptr, n, out.data_ptr(), cnt + 8, timing=t)
torch.cuda.synchronize()
K = 50; ks = []; ts = []
t0 = time.perf_counter()
for _ in range(K):
    rx.find_all_device(buf.ptr, n, out.data_ptr(), cnt + 8, timing=t); ks.append(t.kernel_ms); ts.append(t.total_ms)
wall = (time.perf_counter() - t0) / K * 1e3
print(f"wall/step {wall:.4f} ms  total_ms(ev0->ev2) {sum(ts)/K:.4f}  kernel_ms {sum(ks)/K:.4f}")
