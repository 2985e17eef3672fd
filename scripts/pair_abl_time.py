"""Kernel time of the literal-set kernel on config 3 (1 GiB), rows NOT checked: for the ablated builds of scripts/build_variant.sh."""
import sys, os
import torch
import coregex_amd as cx
LITS16 = "error|warning|fatal|critical|panic|timeout|refused|denied|googlebot|bingbot|yandexbot|crawler|spider|failure|exception|overflow"
n = 1 << 30
buf = cx.DeviceBuffer(n); buf.fill_synth(3, 0xC0FFEE03, 0)
rx = cx.compile(LITS16)
out = torch.zeros((4440808 + 64, 2), dtype=torch.int64, device="cuda")
t = cx.Timing(); best = 1e9; cnt = -1
for i in range(8):
    try:
        cnt = rx.find_all_device(buf.ptr, n, out.data_ptr(), out.shape[0], timing=t)
    except Exception as e:
        print("call failed:", str(e)[:100]); break
    if i: best = min(best, t.kernel_ms)
print(f"{os.environ.get('CXG_LIB_PATH', 'product').split('_')[-1]:12s} rows {cnt} kernel {cx._lib.lib().cxg_kernel_name(t.kernel).decode()} launches {t.n_launches} kernel_ms {best:.4f}", flush=True)
