cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_gpu_offset_caps.py tests/test_gpu_wide.py tests/test_gpu_trio.py -m gpu -q -x > gpurun_out/r04_offcaps_pytest.log 2>&1; echo pytest=$?; tail -6 gpurun_out/r04_offcaps_pytest.log | cut -c1-600
{ timeout 200 python scripts/time_patterns.py 'user=(\S+)' '"([^"]*)"' '\[([^\]]+)\]' 'user=\S+' '(\S+)' 2>&1 | grep -v amdgpu.ids | tail -5; } > gpurun_out/r04_time_offset_caps.txt; cat gpurun_out/r04_time_offset_caps.txt
