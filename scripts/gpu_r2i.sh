cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "submatch or config5 or full_size or boundary or random_patterns" > gpurun_out/pytest_sub.log 2>&1; echo pytest=$?; tail -5 gpurun_out/pytest_sub.log
timeout 300 python bench.py --config 5 --steps 20 --warmup 5 > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err; echo bench=$?; python -c "
import json; d=json.load(open('gpurun_out/bench_cfg5.json')); print(d['value'], d['ms_per_step'], d['roofline'])"
python - <<'PY'
import torch, time
# write-bandwidth ceiling: fill 3.17 GB of int64 (config 4 writes 198M rows of 16 B per GiB) while reading 1 GiB
n = 198_000_000 * 2
out = torch.empty(n, dtype=torch.int64, device="cuda")
src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for name, f in (("fill 3.17 GB", lambda: out.fill_(7)), ("copy 1 GiB (r+w)", lambda: src.clone())):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): f()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    print(name, f"{dt*1e3:.3f} ms")
PY
