cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cat > /tmp/abl.py <<'PY'
import os, sys, torch, coregex_amd as cx
from coregex_amd import Timing
pat = sys.argv[1] if len(sys.argv) > 1 else r"\d+\.\d+\.\d+\.\d+"
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = 1 << 30
buf = cx.DeviceBuffer(n); buf.fill_synth(cfg, 1, 0)
re_ = cx.compile(pat)
out = torch.empty((24_000_000, 2), dtype=torch.int64, device="cuda")
t = Timing(); best = 1e9; cnt = 0
for i in range(4):
    cnt = re_.find_all_device(buf.ptr, n, out.data_ptr(), out.shape[0], timing=t)
    if i: best = min(best, t.kernel_ms)
print(pat, "count", cnt, "kernel_ms", round(best, 4), "launches", t.n_launches)
PY
CXG_PROF=1 CXG_VERBOSE=1 PYTHONPATH=$GRAFT_REPO_ROOT timeout 120 python /tmp/abl.py 2>&1 | grep -v "^$" | grep -v amdgpu.ids | tail -3
