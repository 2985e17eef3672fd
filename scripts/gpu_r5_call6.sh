# Round 5, sixth device call: TRIO mode of the persistent kernel (config 5): parity, then A/B against the grouped kernel.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_gpu_trio.py tests/test_gpu_parity.py tests/test_boundary.py tests/test_gpu_golden_rows.py -m gpu -q -x -k "not 8gib or 5-" > gpurun_out/r05_c6_pytest_trio.log 2>&1; echo pytest_trio=$?; tail -12 gpurun_out/r05_c6_pytest_trio.log | cut -c1-500
{ echo "product (TRIO mode)"; CXG_VERBOSE=1 timeout 100 python scripts/time_configs.py 5 2>&1 | grep -v "XCD\|waves;\|units waited"
  echo "CXG_NO_PERSIST=1 (grouped k_scan_trio_wave)"; CXG_NO_PERSIST=1 timeout 100 python scripts/time_configs.py 5
  echo "other trio programs on 1 GiB of config 2"; timeout 100 python scripts/time_patterns.py '(\d+)\.(\d+)\.(\d+)\.(\d+)' '\d+-\d+:\d+' '(\w+)=(\w+)' 2>&1 | sed 's/  */ /g'
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c6_trio.txt; cat gpurun_out/r05_c6_trio.txt | cut -c1-300
