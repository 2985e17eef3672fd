cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_gpu_runs.py tests/test_zzzz_gpu_routing.py -m gpu -q -x > gpurun_out/r04_runs_pytest.log 2>&1; echo pytest=$?; tail -12 gpurun_out/r04_runs_pytest.log | cut -c1-600
{ timeout 250 python scripts/time_patterns.py '(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)' '\d+\.\d+x?' '\b[0-9]{3}\b' '0x[0-9a-f]+\b' 2>&1 | grep -v amdgpu.ids | tail -5
  echo "same, CXG_NO_RUNS_KERNEL=1:"
  CXG_NO_RUNS_KERNEL=1 timeout 250 python scripts/time_patterns.py '(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)' '\d+\.\d+x?' '\b[0-9]{3}\b' '0x[0-9a-f]+\b' 2>&1 | grep -v amdgpu.ids | tail -5
} > gpurun_out/r04_time_runs.txt 2>&1; cat gpurun_out/r04_time_runs.txt | cut -c1-400
