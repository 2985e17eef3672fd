# Round 5, third device call: async entry, order-slot mutex (two threads, eight shard threads), literal core from LDS, trio table back.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; V=$R/coregex_amd/variants
timeout 500 python -m pytest tests/test_gpu_async.py tests/test_gpu_watchdog.py tests/test_gpu_shards_threads.py tests/test_gpu_literal.py tests/test_gpu_u32_rows.py tests/test_boundary.py tests/test_gpu_trio.py -m gpu -q > gpurun_out/r05_c3_pytest_new.log 2>&1; echo pytest_new=$?; tail -30 gpurun_out/r05_c3_pytest_new.log | cut -c1-500
{ echo "product"; timeout 200 python scripts/time_configs.py 1 5
  echo "literals on 1 GiB of config 2"; timeout 100 python scripts/time_patterns.py 'GET' 'HTTP/' 'error' 2>&1 | sed 's/  */ /g'
  timeout 60 python scripts/time_headline.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c3_configs.txt; cat gpurun_out/r05_c3_configs.txt | cut -c1-330
timeout 300 python bench.py --no-north-star --no-pmc > gpurun_out/r05_c3_bench.json 2> gpurun_out/r05_c3_bench.err; echo bench=$?; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_c3_bench.json"))
print("1 GiB", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], "async", d.get("async"), "anchor", d["cpu_baseline"].get("anchor_sparse_GBps"), d["cpu_baseline"]["value"])
PY
tail -3 gpurun_out/r05_c3_bench.err
