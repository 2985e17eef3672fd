import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coregex_amd as cx
pat = r"\D+?xyzabcx|bcxy|cxyz|xyza[a-c]"
rx = cx.compile(pat)
import struct
b = rx.blob()
print("strategy", rx.strategy, "kind", struct.unpack_from("<I", b, 4)[0], "flags", hex(struct.unpack_from("<I", b, 8)[0]), "fsm image", None if rx.fsm_image() is None else len(rx.fsm_image()), flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
hay = np.frombuffer(b"abcxyza:c" * n, dtype=np.uint8)
d = torch.from_numpy(hay.copy()).cuda()
t = cx.Timing()
t0 = time.time()
try:
    c = rx.find_all_device(d.data_ptr(), hay.size, timing=t)
    print("count", c, "kernels", t.kernels, "launches", t.n_launches, "reason", t.fallback_reason, "%.2fs" % (time.time() - t0), flush=True)
except Exception as e:
    print("EXC", str(e)[:200], "kernels", t.kernels, "%.2fs" % (time.time() - t0), flush=True)
