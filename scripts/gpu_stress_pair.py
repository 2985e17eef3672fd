"""Stress (GPU box): the pair kernel against the wave kernel (the same call with FindAll's n set: capi_ladder.hip keeps such calls on
scan_teddy_wave.hip) on haystacks of many lengths, thousands of launches; prints every disagreement."""
import os, random, sys, time
os.environ.setdefault("CXG_PAIR_MIN_BYTES", "0")      # (by default the pair kernel serves haystacks from 320 MiB on)
import numpy as np, torch
sys.path.insert(0, "tests")
import coregex_amd as cx
from test_wrapped_cpu import TOKS
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
pats = [r"(?m)^(?:abc|xyz)$", r"\berror\b", "error|warn|fatal|abc", r"(?m)^(GET|POST|PUT|DELETE|PATCH)"]
rng = random.Random(77)
big = np.frombuffer(b"".join(rng.choice(TOKS + [b" pad pad pad pad pad pad pad pad "] * 12) for _ in range(1_500_000)), dtype=np.uint8)
d = torch.from_numpy(big.copy()).cuda()
bad = 0; runs = 0; t0 = time.time(); on_pair = 0
rxs = [cx.compile(p) for p in pats]
out_a = torch.zeros((1500000, 2), dtype=torch.int64, device="cuda"); out_b = torch.zeros_like(out_a)
for rep in range(reps):
    if rep % 50 == 0: rxs = [cx.compile(p) for p in pats]      # fresh programs: a remembered fallback keeps a program off the pair kernel
    rx = rxs[rep % len(rxs)]
    n = rng.choice([rng.randrange(1, 4000), rng.randrange(3000, 300000), rng.randrange(200000, big.size - 64)])
    off = rng.randrange(0, (big.size - n) // 16 + 1) * 16
    t = cx.Timing()
    ca = rx.find_all_device(d.data_ptr() + off, n, out_a.data_ptr(), 1500000, timing=t)
    cb = rx.find_all_device(d.data_ptr() + off, n, out_b.data_ptr(), 1500000, n=1 << 40)
    runs += 1; on_pair += list(t.kernels)[:1] == [21] and t.n_launches == 1
    if ca != cb or not torch.equal(out_a[:ca], out_b[:cb]):
        bad += 1
        A = set(map(tuple, out_a[:ca].cpu().tolist())); B = set(map(tuple, out_b[:cb].cpu().tolist()))
        miss = sorted(B - A)[:3]; extra = sorted(A - B)[:3]
        print(f"rep {rep} pat {pats[rep % len(pats)]!r} n {n} off {off}: pair {ca} wave {cb} kernels {list(t.kernels)} missing {miss} (unit {[m[0] // 15360 for m in miss]} of {n // 15360}, in tile {[m[0] % 3840 for m in miss]}) extra {extra}", flush=True)
print(f"launch pairs {runs} (pair kernel alone: {on_pair}) disagreements {bad} in {time.time() - t0:.1f}s", flush=True)
