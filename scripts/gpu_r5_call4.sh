# Round 5, fourth device call: a foreign kernel beside the persistent scan; char-class swizzle; literal mode at 5 workgroups; async with one event; the tier.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; V=$R/coregex_amd/variants
{ for m in none torchfill synthfill; do CXG_VERBOSE=1 timeout 120 python scripts/gpu_foreign_kernel.py $m 2>&1 | grep -v "XCD\|waves;\|units waited\|amdgpu.ids" | tail -4; done
  CXG_LIB_PATH=$V/libcoregex_hip_noprio.so CXG_VERBOSE=1 timeout 120 python scripts/gpu_foreign_kernel.py torchfill 2>&1 | grep -v "XCD\|waves;\|units waited\|amdgpu.ids" | tail -3
} > gpurun_out/r05_c4_foreign_kernel.txt 2>&1; cat gpurun_out/r05_c4_foreign_kernel.txt | cut -c1-300
{ echo "product"; CXG_VERBOSE=1 timeout 200 python scripts/time_configs.py 1 4 2>&1 | grep -v "XCD\|waves;\|units waited"
  echo "literals on 1 GiB of config 2"; timeout 100 python scripts/time_patterns.py 'GET' 'HTTP/' 'error' 2>&1 | sed 's/  */ /g'
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c4_configs.txt; cat gpurun_out/r05_c4_configs.txt | cut -c1-330
timeout 900 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r05_c4_pytest_gpu.log 2>&1; echo pytest=$?; tail -15 gpurun_out/r05_c4_pytest_gpu.log | cut -c1-400
timeout 300 python bench.py --no-north-star --no-pmc --no-cpu-baseline > gpurun_out/r05_c4_bench.json 2> gpurun_out/r05_c4_bench.err; echo bench=$?; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_c4_bench.json"))
print("1 GiB", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], "async", d.get("async"))
PY
