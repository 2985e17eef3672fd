# Round 5, second device call: literal mode (config 1), watchdog hygiene, class plans (configs 4, 5), trio SWAR, 9-tile units;
# the 16 GiB watchdog hit of call 1 looked at (fill kernel beside the scan or not, which wait gave up).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; V=$R/coregex_amd/variants
timeout 400 python -m pytest tests/test_gpu_literal.py tests/test_gpu_watchdog.py tests/test_gpu_fields.py tests/test_gpu_trio.py tests/test_gpu_u32_rows.py tests/test_gpu_pairs.py -m gpu -q -x > gpurun_out/r05_c2_pytest_new.log 2>&1; echo pytest_new=$?; tail -25 gpurun_out/r05_c2_pytest_new.log | cut -c1-600
{ timeout 60 python scripts/time_headline.py
  for v in t7 t11; do CXG_LIB_PATH=$V/libcoregex_hip_$v.so timeout 60 python scripts/time_headline.py; done
  timeout 60 python scripts/time_headline.py
  echo "16 GiB, torch.empty"; CXG_VERBOSE=1 timeout 120 python scripts/time_headline.py 10 16 2>&1 | grep -v "XCD\|waves;\|units waited"
  echo "16 GiB, torch.zeros (a fill kernel beside the first scan)"; TH_ZEROS=1 CXG_VERBOSE=1 timeout 120 python scripts/time_headline.py 10 16 2>&1 | grep -v "XCD\|waves;\|units waited"
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c2_headline.txt; cat gpurun_out/r05_c2_headline.txt | cut -c1-330
{ echo "product"; timeout 200 python scripts/time_configs.py 1 3 4 5
  echo "CXG_NO_LITERAL_KERNEL=1 (chain kernel)"; CXG_NO_LITERAL_KERNEL=1 timeout 100 python scripts/time_configs.py 1
  echo "CXG_NO_PLANS=1"; CXG_NO_PLANS=1 timeout 100 python scripts/time_configs.py 4 5
  echo "variant triolut (byte table)"; CXG_LIB_PATH=$V/libcoregex_hip_triolut.so timeout 100 python scripts/time_configs.py 5
  echo "variant ccnt (nt loads)"; CXG_LIB_PATH=$V/libcoregex_hip_ccnt.so timeout 100 python scripts/time_configs.py 4
  echo "other literals on 1 GiB of config 2"; timeout 100 python scripts/time_patterns.py 'GET' 'HTTP/' 'error' 'Mozilla' 2>&1 | sed 's/  */ /g'
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c2_configs.txt; cat gpurun_out/r05_c2_configs.txt | cut -c1-200
timeout 600 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r05_c2_pytest_gpu.log 2>&1; echo pytest=$?; tail -12 gpurun_out/r05_c2_pytest_gpu.log | cut -c1-400
timeout 400 python bench.py > gpurun_out/r05_c2_bench_default.json 2> gpurun_out/r05_c2_bench_default.err; echo bench=$?; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_c2_bench_default.json"))
print("1 GiB", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["roofline"]["traffic"], "| 64 GiB", d["north_star"]["roofline"]["kernel_ms_avg"], d["north_star"]["roofline"]["frac"], d["north_star"]["all_rows_check"])
PY
