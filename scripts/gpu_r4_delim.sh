cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_gpu_delim.py tests/test_gpu_pairs.py tests/test_gpu_wide.py -m gpu -q -x > gpurun_out/r04_delim_pytest.log 2>&1; echo pytest=$?; tail -8 gpurun_out/r04_delim_pytest.log | cut -c1-600
{ timeout 200 python scripts/time_patterns.py '"[^"]*"' '\S+' '[^,]+' '\[[^\]]+\]' '<[^>]+>' '\([^)]*\)' 'user=(\S+)' 2>&1 | grep -v amdgpu.ids | tail -7; } > gpurun_out/r04_time_wide.txt; cat gpurun_out/r04_time_wide.txt
