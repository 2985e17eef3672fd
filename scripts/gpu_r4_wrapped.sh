cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_wrapped.py tests/test_gpu_offset_caps.py tests/test_gpu_fsm.py tests/test_zz_gpu_look_wider.py tests/test_zzzz_gpu_routing.py -m gpu -q -x > gpurun_out/r04_wrapped_pytest.log 2>&1; echo pytest=$?; tail -6 gpurun_out/r04_wrapped_pytest.log | cut -c1-500
{ timeout 200 python scripts/time_patterns.py '\berror\b' '(?m)^(?:GET|POST|PUT|DELETE|PATCH)' '\bGET\b' '(?m)error$' 2>&1 | grep -v amdgpu.ids | tail -4
  echo "config-1 corpus:"; timeout 200 python bench.py --pattern '\berror\b' --synth-config 1 --steps 10 --warmup 2 --no-pmc 2>/dev/null | grep -o '"roofline": {[^}]*}' | cut -c1-400
  for v in sc1ld; do echo "headline, haystack loads variant $v"; CXG_LIB_PATH=$GRAFT_REPO_ROOT/coregex_amd/variants/libcoregex_hip_$v.so timeout 120 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1; done
  echo "headline product"; timeout 120 python scripts/time_modes.py 2>&1 | grep -v amdgpu.ids | tail -1
} > gpurun_out/r04_time_wrapped.txt 2>&1; cat gpurun_out/r04_time_wrapped.txt
FUZZ_WIDE=1 FUZZ_TRACE=1 timeout 500 python scripts/gpu_fuzz.py 82 300 > gpurun_out/r04_gpu_fuzz_wide_trace.txt 2>&1; tail -5 gpurun_out/r04_gpu_fuzz_wide_trace.txt | cut -c1-300
python - <<'PY'
import re
prev=None; rows=[]
for l in open("gpurun_out/r04_gpu_fuzz_wide_trace.txt"):
    m=re.match(r"\[\s*([0-9.]+)s\] (.*)", l)
    if m:
        t=float(m.group(1))
        if prev: rows.append((t-prev[0], prev[1]))
        prev=(t, m.group(2))
rows.sort(reverse=True)
print("slowest patterns (s):", rows[:8])
PY
