cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
FUZZ_WIDE=1 FUZZ_TRACE=1 timeout 200 python scripts/gpu_fuzz.py 82 300 > gpurun_out/r04_gpu_fuzz_wide.txt 2> gpurun_out/r04_gpu_fuzz_wide_trace.txt; tail -3 gpurun_out/r04_gpu_fuzz_wide.txt | cut -c1-400
python - <<'PY'
import re
prev=None; rows=[]
for l in open("gpurun_out/r04_gpu_fuzz_wide_trace.txt"):
    m=re.match(r"\[\s*([0-9.]+)s\] (.*)", l)
    if m:
        t=float(m.group(1))
        if prev: rows.append((round(t-prev[0],1), prev[1]))
        prev=(t, m.group(2))
print("patterns started:", len(rows)+1, "last:", prev)
rows.sort(reverse=True)
print("slowest patterns (s):", rows[:10])
PY
