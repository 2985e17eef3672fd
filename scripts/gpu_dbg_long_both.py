import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coregex_amd as cx
from oracle import oracle
for pat in ['(.)[^a-c]+', '[^a-c]+x?y?', r'(\d)[\d. \n]+']:
    rx = cx.compile(pat); o = oracle.Regex(pat)
    print(pat, rx.strategy, rx.supported, flush=True)
    if not rx.supported: continue
    for n in (200, 3000, 100000, 130000, 171000, 400000):
        hay = np.frombuffer(((b"1.2.3.4 " * 7 + b"\n") * 8000)[:n], dtype=np.uint8)
        exp = o.find_all_index(hay).tolist()
        try:
            got = rx.find_all_index(hay).tolist()
        except cx.CoregexError as e:
            got = "ERR " + str(e)[:80]
        d = torch.from_numpy(np.concatenate([hay, np.zeros(64, dtype=np.uint8)])).cuda()
        t = cx.Timing()
        out = torch.full((16, 2), -7, dtype=torch.int64, device="cuda")
        try:
            cnt = rx.find_all_device(d.data_ptr(), hay.size, out.data_ptr(), 16, timing=t)
            dev = out[:cnt].cpu().numpy().tolist()
        except cx.CoregexError as e:
            dev = "ERR " + str(e)[:80]
        print(" n", n, "exp", exp[:3], "host", got[:3] if isinstance(got, list) else got, "dev", dev[:3] if isinstance(dev, list) else dev, "kernel", t.kernel, "launches", t.n_launches, "reason", hex(t.fallback_reason), flush=True)
